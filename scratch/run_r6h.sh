cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_threads.py tests/test_gpu_ba_variants.py tests/test_gpu_match.py -q -m gpu 2>&1 | tail -5 > $O/tests.log
python scratch/time_ba_large.py 5 2>/dev/null | grep "it/s" > $O/large.log
python scratch/time_ba_batch.py 8 20 2>/dev/null | grep "B=1\|B=8" >> $O/large.log
cat $O/tests.log $O/large.log
