#!/bin/bash
# kernel stats of the BA batch at B = 1 and B = 8 (run on the GPU box from the repo root)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ba_prof
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for B in 1 8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$B -- python $R/scratch/time_ba_batch.py $B 10 > $O/b$B.log 2>&1
  f=$(ls $O/b$B/*/*kernel_stats.csv | head -1); cp $f $O/b${B}_kernel_stats.csv
  echo "== B=$B"; tail -3 $O/b$B.log; python $R/scratch/kstats.py $O/b${B}_kernel_stats.csv 20
done
