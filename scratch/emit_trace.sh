#!/bin/bash
# GPU box: per-(kernel, grid) durations of k_resize / k_fast_cells with the emitting FAST on / off, in the pipeline and alone
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
HEAD="--cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api --no-two-lanes --steps 6 --warmup 2"
for e in ${EMITS:-15 0}; do
  rm -rf $R/gpurun_out/tr_e$e; echo "== pipeline DCS_ORB_EMIT=$e"
  DCS_ORB_EMIT=$e rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_e$e -o t -- python $R/bench.py $HEAD > /dev/null 2>&1
  python $R/scratch/trace_by_grid.py $R/gpurun_out/tr_e$e k_resize k_fast_cells
  rm -rf $R/gpurun_out/tr_e$e
done
for e in ${EMITS:-15 0}; do
  rm -rf $R/gpurun_out/tr_solo$e; echo "== alone DCS_ORB_EMIT=$e"
  DCS_ORB_NO_OVERLAP=1 DCS_ORB_EMIT=$e rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_solo$e -o t -- python $R/bench.py $HEAD --serial > /dev/null 2>&1
  python $R/scratch/trace_by_grid.py $R/gpurun_out/tr_solo$e k_resize k_fast_cells
  rm -rf $R/gpurun_out/tr_solo$e
done
