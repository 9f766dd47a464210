cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_track.py tests/test_gpu_cpp_mirror.py tests/test_gpu_threads.py -m gpu -q 2>&1 | tail -3
( timeout 200 python scratch/stress_track_dev.py 60 832 2>/dev/null | tail -1 ) 
for i in 1 2; do python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-two-lanes --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); p=d['per_frame_total']; print({k:p[k] for k in p if k.startswith('ms_')})"; done
