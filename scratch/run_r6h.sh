cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ba.py tests/test_gpu_track.py -m gpu -q 2>&1 | tail -3
python scratch/pose_flip_stats.py 200 1 2>/dev/null | tail -1
python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-two-lanes --steps 3 --warmup 1 > gpurun_out/pf.log 2>/dev/null
python - <<'PY'
import json
lines=[l for l in open("gpurun_out/pf.log").read().splitlines() if l.startswith("{")]; d=json.loads(lines[-1]); p=d["per_frame_total"]; print({k:p[k] for k in p if k.startswith("ms_")}); print(d["per_frame_chain"]["chained_ms_per_frame_batch_of_1"], d["per_frame_chain"]["chained_ms_per_frame_batch_of_16"])
PY
