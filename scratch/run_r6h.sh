cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_track.py tests/test_gpu_match.py tests/test_gpu_cpp_mirror.py tests/test_gpu_threads.py -m gpu -q 2>&1 | tail -3
( timeout 200 python scratch/stress_track.py 70 821 2>/dev/null | tail -1 ) &
( timeout 200 python scratch/stress_track_dev.py 70 822 2>/dev/null | tail -1 ) &
( timeout 200 python scratch/stress_parity2.py 70 823 2>/dev/null | tail -1 ) &
( timeout 200 python scratch/stress_parity3.py 70 824 2>/dev/null | tail -1 ) &
wait
python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-two-lanes --steps 3 --warmup 1 > gpurun_out/pf.log 2>/dev/null
python - <<'PY'
import json
lines=[l for l in open("gpurun_out/pf.log").read().splitlines() if l.startswith("{")]; d=json.loads(lines[-1]); p=d["per_frame_total"]; print({k:p[k] for k in p if k.startswith("ms_")}); print(d["per_frame_chain"]["chained_ms_per_frame_batch_of_1"], d["per_frame_chain"]["chained_ms_per_frame_batch_of_16"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf -o pf -- python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-two-lanes --steps 3 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/pf/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'resolve' in r['Name']: print(r['Name'][:58], r['Calls'], r['AverageNs'], r['MinNs'])
PY
