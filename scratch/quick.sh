#!/bin/bash
# quick loop on the GPU box: extraction parity + the headline bench line (timed region only), separate-blur vs fused pipeline
# usage: scratch/quick.sh <tag> [pytest args...]
TAG=${1:-q}; shift
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest ${@:-tests/test_gpu_extract.py} -x -q 2>&1 | tail -4
for mode in 0 1 0 1; do
DCS_ORB_FUSED_BLUR=$mode python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-host-api > gpurun_out/$TAG/bench_headline_f$mode.json 2> gpurun_out/$TAG/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench_headline_f$mode.json").read().strip().splitlines()[-1])
print("fused=$mode value", round(d["value"]), "ms/step", d["ms_per_step"], "stages", {k:round(v,1) for k,v in d["stage_us_per_step"].items()})
PY
done
