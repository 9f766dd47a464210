#!/bin/bash
# per-kernel average durations of real LM steps of the C4 problem for one build: usage ba_kernel_avgs.sh <lib.so> <tag>
cd /tmp; export TMPDIR=/tmp
DCS_LIB_PATH=$1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kavg_$2 -- python $GRAFT_REPO_ROOT/scratch/time_ba_batch.py 1 20 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
echo "== $2"; python scratch/kstats.py $(ls gpurun_out/kavg_$2/*/*kernel_stats.csv | head -1) 9; rm -rf gpurun_out/kavg_$2
