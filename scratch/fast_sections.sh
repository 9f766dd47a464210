#!/bin/bash
# GPU box: VALU / SALU / LDS instruction counts of k_fast_cells per section. Needs the profiling side build:
#   DCS_OUT_DIR=scratch/ab/sec DCS_OBJ_DIR=/tmp/objsec DCS_EXTRA_FLAGS=-DDCS_FAST_SECTIONS bash build.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fastsec; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export DCS_ORB_NO_OVERLAP=1 DCS_LIB_PATH=$R/scratch/ab/sec/libdcs_hip.so
for s in ${SECTIONS:-1 2 3 4 0}; do
  DCS_FAST_STOP=$s timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc$s -- python $R/scratch/time_extract.py 128 > $O/run$s.log 2>&1
  echo "stop=$s $(python $R/scratch/pmc_sum.py $O/pmc$s | grep fast_cells)" >> $O/summary.txt
done
cat $O/summary.txt
