cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ba.py tests/test_gpu_track.py tests/test_gpu_ba_threads.py -q -m gpu 2>&1 | tail -25 > $O/tests.log
cat $O/tests.log
