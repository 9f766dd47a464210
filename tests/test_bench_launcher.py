"""bench.py's N > 1 entry on CPU: `python bench.py --gpus 2 --selftest-launch` must spawn its own two ranks (re-exec under
torch.distributed.run, the way `--gpus 2` does on the GPU node), bring up the process group (gloo here, RCCL there), run
the feature all-gather of orb-slam2-dualcam_amd/sharding.py and print ONE JSON line with n_gpus = 2."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-launch"] + extra, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_its_ranks():
    out = _run(["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["allgather_ok"] is True and out["units_of_rank0"] == [0, 2, 4]


def test_bench_single_rank_selftest():
    out = _run([])
    assert out["n_gpus"] == 1 and out["allgather_ok"] is True
