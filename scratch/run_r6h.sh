cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ba.py tests/test_gpu_track.py -m gpu -q 2>&1 | tail -3
python scratch/pose_flip_stats.py 400 1 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tt -o tt -- python scratch/time_track.py > gpurun_out/tt.log 2>&1
grep DCS_POSE gpurun_out/tt.log
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/tt/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'pose' in r['Name']: print(r['Name'][:40], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
python scratch/time_track.py 2>/dev/null | tail -1
DCS_LIB_PATH=$GRAFT_REPO_ROOT/scratch/ab/pose_prof/libdcs_hip.so python tools/pose_timeline.py 2>&1 | tail -12
