#!/bin/bash
# GPU box: extraction parity, then the headline under DCS_ORB_EMIT settings (alternating), then per-grid traces
timeout 900 python -m pytest tests/test_gpu_extract.py -x -q 2>&1 | tail -3
bash scratch/ab_envvar.sh DCS_ORB_EMIT "${VALS:-0 7 4 3}" ${ROUNDS:-2}
EMITS="${TRACE:-7}" bash scratch/emit_trace.sh
