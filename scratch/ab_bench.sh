#!/bin/bash
# A/B of library builds in the headline bench (as launched): scratch/ab_bench.sh <fused 0|1> <lib dir> [<lib dir> ...]
MODE=$1; shift
for rep in 1 2; do
for l in "$@"; do
DCS_ORB_FUSED_BLUR=$MODE DCS_LIB_PATH=$l/libdcs_hip.so python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s' % '$l', round(d['value']), d['ms_per_step'], {k:round(v) for k,v in d['stage_us_per_step'].items() if k in ('fast_us','describe_us','pyramid_us','total_us')})"
done; done
