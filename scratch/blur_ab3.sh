#!/bin/bash
# alternating rounds: blur kernel time of the serial separate-blur headline, default build vs a side build (DCS_LIB_PATH).  bash scratch/blur_ab3.sh <lib> [rounds]
LIB=$1; N=${2:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/blurab3
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
HEAD="--cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api"
for i in $(seq $N); do for f in default side; do
  if [ $f = side ]; then export DCS_LIB_PATH=$R/$LIB; else unset DCS_LIB_PATH; fi
  DCS_ORB_FUSED_BLUR=0 DCS_ORB_NO_OVERLAP=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python $R/bench.py $HEAD --serial --steps 30 > /dev/null 2>&1
  echo "round $i $f: $(grep -h 'k_blur' $O/s/*/*kernel_stats.csv | awk -F'",' '{split($2,a,","); split($1,n,"::"); split(n[2],m,"("); printf "%s %.1f us x %d   ", m[1], a[3]/1000, a[1]}')"
  rm -rf $O/s
done; done | tee $O/ab.txt
