cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_threads.py -q -m gpu 2>&1 | tail -15 > $O/tests.log
timeout 900 python scratch/time_ba_large.py 5 > $O/ba_large.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/scratch/time_ba_large.py 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/ba_large_kernel_stats.csv; rm -rf $O/prof
cat $O/tests.log; grep -v amdgpu.ids $O/ba_large.log; head -10 $O/ba_large_kernel_stats.csv | cut -c1-150
