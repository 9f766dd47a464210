#!/bin/bash
# blur alone, folded (k_blur_fold) vs the round-2 pair (k_blur + k_blur_edge_cols): kernel stats of the serial headline run with the
# separate blur kernels forced, and the FETCH_SIZE / WRITE_SIZE passes of the same command.  bash scratch/blur_ab.sh <tag>
TAG=${1:-blur}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
HEAD="--cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api"
for f in 1 0; do
  DCS_BLUR_FOLD=$f DCS_ORB_FUSED_BLUR=0 DCS_ORB_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo_$f -- python $R/bench.py $HEAD --serial --steps 30 > /dev/null 2>&1
  cp $(ls $O/solo_$f/*/*kernel_stats.csv | head -1) $O/solo_fold${f}_kernel_stats.csv
  DCS_BLUR_FOLD=$f DCS_ORB_FUSED_BLUR=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$f -- python $R/bench.py --steps 5 --warmup 1 $HEAD > /dev/null 2>&1
  DCS_BLUR_FOLD=$f DCS_ORB_FUSED_BLUR=0 timeout 300 python $R/bench.py $HEAD > $O/bench_fold$f.json 2>/dev/null
  echo "== fold=$f"; grep k_blur $O/solo_fold${f}_kernel_stats.csv
  python - <<PY
import csv,glob,collections
tot=collections.defaultdict(lambda:[set(),0.0])
for fn in glob.glob("$O/fetch_$f/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if 'blur' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
            k=r['Kernel_Name'].split('(')[0]; tot[k][0].add(r['Dispatch_Id']); tot[k][1]+=float(r['Counter_Value'])
for k,(ids,v) in tot.items():
    n=len(ids); print(k, 'launches', n, 'FETCH_SIZE KB per launch', v/n, '= MB', v/n/1024)
PY
  python -c "import json;d=json.load(open('$O/bench_fold$f.json'));print('headline separate blur', d['value'], d['ms_per_step'])"
done
rm -rf $O/solo_? $O/fetch_?
