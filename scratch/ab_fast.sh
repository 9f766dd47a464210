#!/bin/bash
# GPU box: A/B of two builds of the library (scratch/ab/libdcs_hip_base.so vs the in-tree one): stage times solo and overlapped,
# then PMC instruction counts of k_fast_cells for both
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abfast; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BASE=${BASE:-$R/scratch/ab/base/libdcs_hip.so}
for i in 1 2; do
  for L in $BASE ""; do
    echo "lib=${L:-new} solo: $(DCS_LIB_PATH=$L DCS_ORB_NO_OVERLAP=1 python $R/scratch/time_extract.py 2>&1 | tail -1 | cut -c1-200)"
  done
done
for L in $BASE ""; do echo "lib=${L:-new} overlapped: $(DCS_LIB_PATH=$L python $R/scratch/time_extract.py 2>&1 | tail -1 | cut -c1-220)"; done
if [ -z "$NO_PMC" ]; then
for L in $BASE ""; do
  n=$( [ -z "$L" ] && echo new || echo base )
  DCS_LIB_PATH=$L DCS_ORB_NO_OVERLAP=1 timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc_$n -- python $R/scratch/time_extract.py 128 > $O/run_$n.log 2>&1
  echo "$n: $(python $R/scratch/pmc_sum.py $O/pmc_$n | grep fast_cells)"
done
fi
