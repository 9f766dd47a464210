#!/bin/bash
# GPU box: the headline bench (timed region only) under different settings of ONE environment variable, alternating rounds
# usage: scratch/ab_envvar.sh VAR "v1 v2 v3" [rounds]
VAR=$1; VALS=$2; ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do for v in $VALS; do
  env $VAR=$v python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api --no-two-lanes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', round(d['value']), d['ms_per_step'], {k:round(x,1) for k,x in d['stage_us_per_step'].items() if x})"
done; done
