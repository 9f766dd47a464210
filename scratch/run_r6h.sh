cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_track.py tests/test_gpu_match.py tests/test_gpu_cpp_mirror.py tests/test_gpu_threads.py -m gpu -q 2>&1 | tail -3
( timeout 200 python scratch/stress_track.py 70 861 2>/dev/null | tail -1 ) &
( timeout 200 python scratch/stress_track_dev.py 70 862 2>/dev/null | tail -1 ) &
( timeout 200 python scratch/stress_parity2.py 70 863 2>/dev/null | tail -1 ) &
( timeout 200 python scratch/stress_parity3.py 70 864 2>/dev/null | tail -1 ) &
wait
for np_ in 2000 6000 12000; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tqn$np_ -o tq -- python scratch/time_track.py $np_ 2000 10 > gpurun_out/tqn$np_.log 2>&1
echo "$np_ $(grep 'ms per frame' gpurun_out/tqn$np_.log | sed 's/.*edges/edges/')"
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/tqn$np_/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'resolve' in r['Name']: print('   ', r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'])
PY
done
python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-two-lanes --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); p=d['per_frame_total']; print({k:p[k] for k in p if k.startswith('ms_')})"
