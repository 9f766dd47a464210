cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for i in 1 2; do python $R/bench.py --cpu-seconds 0 --no-ba --no-bow 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_us_per_step'])"; done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fx_fetch -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-ba --no-bow > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, os
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/fx_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for name in ("k_fast_cells","k_blur<","k_describe","k_resize"):
            if name in k: acc[name]+=float(r["Counter_Value"]); n[name].add(r["Dispatch_Id"])
for k in acc: print(k, round(acc[k]/len(n[k])/1024,1), "MB/launch")
PY
