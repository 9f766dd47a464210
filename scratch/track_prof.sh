#!/bin/bash
# GPU box: rocprofv3 kernel stats of the per-frame tracking chain alone (batch 1 and 16), new and round-4 pose kernel
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for f in ${MODES:-1 0}; do
  rm -rf $R/gpurun_out/track_f$f
  DCS_POSE_FAST=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/track_f$f -o t -- python $R/scratch/time_track.py ${ARGS:-} > $R/gpurun_out/track_f$f.log 2>&1
  tail -1 $R/gpurun_out/track_f$f.log
  python $R/scratch/kstats.py $R/gpurun_out/track_f$f 12
  python $R/scratch/trace_by_grid.py $R/gpurun_out/track_f$f k_pose_opt
done
