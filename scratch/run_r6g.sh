cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/tests.log
for i in 1 2; do timeout 300 python scratch/time_ba_batch.py 8 20 2>/dev/null | tail -4 >> $O/ba.log; done
cat $O/tests.log $O/ba.log
