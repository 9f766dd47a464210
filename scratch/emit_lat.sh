#!/bin/bash
# GPU box: the latency / host-api / c3 / c5 legs with the emitting FAST on and off
for e in ${VALS:-0 7 0 7}; do
DCS_ORB_EMIT=$e python bench.py --cpu-seconds 0 --no-ba --no-bow --no-two-lanes --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EMIT=$e value', round(d['value']), 'latency', d.get('latency'), '\n   with_transfers', d.get('with_transfers'), d.get('with_transfers_page_locked'), '\n   c3', d.get('c3'), '\n   c5', {k:v for k,v in (d.get('c5_one_gpu') or {}).items() if not isinstance(v,(dict,list))})"
done
