cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
for i in 1 2; do
timeout 300 python scratch/time_ba_batch.py 8 20 2>/dev/null | tail -4 >> $O/ba_new.log
DCS_LIB_PATH=$GRAFT_REPO_ROOT/scratch/ab/r5/libdcs_hip.so timeout 300 python scratch/time_ba_batch.py 8 20 2>/dev/null | tail -4 >> $O/ba_r5.log
done
echo NEW; cat $O/ba_new.log; echo R5; cat $O/ba_r5.log
