cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_track.py tests/test_gpu_ba_threads.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6a/tests.log
timeout 200 python scratch/time_track.py > gpurun_out/r6a/track.log 2>&1
timeout 200 python scratch/time_track.py 900 1000 >> gpurun_out/r6a/track.log 2>&1
DCS_LIB_PATH=$GRAFT_REPO_ROOT/scratch/ab/pose_prof/libdcs_hip.so timeout 120 python tools/pose_timeline.py 2000 2000 > gpurun_out/r6a/pose_timeline.txt 2>&1
DCS_LIB_PATH=$GRAFT_REPO_ROOT/scratch/ab/pose_prof/libdcs_hip.so timeout 120 python tools/pose_timeline.py 4000 4000 >> gpurun_out/r6a/pose_timeline.txt 2>&1
cat gpurun_out/r6a/tests.log gpurun_out/r6a/track.log gpurun_out/r6a/pose_timeline.txt
