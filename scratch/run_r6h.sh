cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ba.py tests/test_gpu_track.py tests/test_gpu_ba_variants.py -m gpu -q 2>&1 | tail -3
python scratch/pose_flip_stats.py 400 1 2>/dev/null | tail -1
DCS_POSE_EXACT_EDGE=1 python scratch/pose_flip_stats.py 400 1 2>/dev/null | tail -1
python scratch/time_track.py 2>/dev/null | tail -1
DCS_POSE_EXACT_EDGE=1 python scratch/time_track.py 2>/dev/null | tail -1
DCS_LIB_PATH=$GRAFT_REPO_ROOT/scratch/ab/pose_prof/libdcs_hip.so python tools/pose_timeline.py 2>&1 | tail -12
