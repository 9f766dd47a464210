#!/bin/bash
# GPU box: VALU / SALU / LDS instruction counts of k_describe with phase A or phase C compiled out (side builds -DDCS_DESCRIBE_SKIP=1 / 2 in
# scratch/ab/dskip1, dskip2) next to the product build: A = full - skip1, C = full - skip2, set-up + B = skip1 + skip2 - full
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/descsec; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export DCS_ORB_NO_OVERLAP=1
for v in full dskip1 dskip2; do
  lib=$R/orb-slam2-dualcam_amd/lib/libdcs_hip.so; [ $v != full ] && lib=$R/scratch/ab/$v/libdcs_hip.so
  DCS_LIB_PATH=$lib timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_MFMA --output-format csv -d $O/pmc_$v -- python $R/scratch/time_extract.py 128 > $O/run_$v.log 2>&1
  echo "$v $(python $R/scratch/pmc_sum.py $O/pmc_$v | grep describe)" >> $O/summary.txt
done
cat $O/summary.txt
