cd $GRAFT_REPO_ROOT
python scratch/time_seam.py 2>/dev/null | tail -1
python - <<'PY' 2>/dev/null | tail -1
import torch
torch.zeros(4, device="cuda")
exec(open("scratch/time_seam.py").read())
PY
python bench.py --cpu-seconds 0 --no-ba --no-bow --no-c3 --no-c5 --no-two-lanes --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['seam_latency'])"
