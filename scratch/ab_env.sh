#!/bin/bash
# A/B of environment switches in the headline bench: scratch/ab_env.sh "VAR=a" "VAR=b" ...
for rep in 1 2; do
for e in "$@"; do
env $e python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s' % '$e', round(d['value']), d['ms_per_step'], {k:round(v) for k,v in d['stage_us_per_step'].items() if k in ('fast_us','describe_us','pyramid_us','total_us')})"
done; done
