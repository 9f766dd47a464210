#!/bin/bash
# alternating rounds of the serial separate-blur headline: blur kernels' average time, folded vs the round-2 pair.  bash scratch/blur_ab2.sh <tag> [rounds]
TAG=${1:-blurab}; N=${2:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
HEAD="--cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api"
for i in $(seq $N); do for f in 1 0; do
  DCS_BLUR_FOLD=$f DCS_ORB_FUSED_BLUR=0 DCS_ORB_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python $R/bench.py $HEAD --serial --steps 30 > /dev/null 2>&1
  echo "round $i fold=$f: $(grep -h 'k_blur\|k_fast_cells\|k_resize' $O/s/*/*kernel_stats.csv | awk -F'",' '{split($2,a,","); split($1,n,"::"); split(n[2],m,"("); printf "%s %.1f us x %d   ", m[1], a[3]/1000, a[1]}')"
  rm -rf $O/s
done; done | tee $O/ab.txt
