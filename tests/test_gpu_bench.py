"""bench.py as ONE rank of a distributed run on the single GPU of the test box: RANK / WORLD_SIZE / LOCAL_RANK set the way
torch.distributed.run sets them, so the process initialises RCCL ("nccl"), brings up the library's own communicator (dcs_comm_*) and
runs the per-step feature exchange INSIDE the timed region -- the N > 1 code path of the scaling bench at world size 1. N = 2, 4, 8 on
real GPUs is the driver's SCALE run; the world-2 exchange itself runs on CPU / gloo (tests/test_sharding_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench_as_rank(extra):
    env = dict(os.environ)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--pairs", "8", "--cpu-seconds", "0",
           "--no-bow", "--no-c3", "--no-c5", "--no-host-api"] + (["--no-ba"] if "--rank-legs-with-ba" not in extra else []) + [e for e in extra if e != "--rank-legs-with-ba"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-3000:]
    return json.loads(lines[0])


def test_bench_under_rank_env_exchanges_through_the_c_abi():
    out = _bench_as_rank(["--no-rank-legs"])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["exchange"].startswith("dcs_features_allgather"), out["exchange"]
    assert out["allgather_us"] > 0 and out["allgather_bytes_per_rank"] > 100000
    assert out["exchange_crosscheck"]["slot_arrays_equal"] is True
    assert out["config"]["input_sets_rotated"] == 3


def test_bench_under_rank_env_torch_exchange():
    out = _bench_as_rank(["--exchange", "torch", "--no-rank-legs"])
    assert out["exchange"].startswith("torch.distributed") and out["allgather_us"] > 0 and out["value"] > 0


def test_bench_rank_legs_c3_scaled_and_c5_node():
    """what an N-GPU run measures after the headline -- BASELINE configs[2] (c3_scaled) and configs[4] (c5_node) -- on RCCL at world size 1:
    the real extraction / matching / exchange / solver behind the same leg code tests/test_bench_launcher.py drives at world size 2 on gloo"""
    out = _bench_as_rank(["--rank-legs-with-ba", "--rank-leg-seconds", "0.5"])
    assert out["config"]["workload"].startswith("configs[1]")
    c3, c5 = out["c3_scaled"], out["c5_node"]
    assert isinstance(c3, dict) and isinstance(c5, dict), (c3, c5)
    assert c3["n_gpus"] == 1 and c3["kfeatures_s"] > 0 and c3["exchange"].startswith("dcs_features_allgather")
    assert 1500 * 128 < c3["per_rank_features_per_step"][0] <= 2096 * 128            # 64 dual frames = 128 images x ~2000 features
    assert c3["allgather_us"] > 0 and c3["allgather_bytes_per_rank"] == 2 * (2096 * 60 + 64)
    assert c5["concurrent"]["dual_frames_s"] > 0 and c5["concurrent"]["ba_iters_s"] > 0 and c5["concurrent"]["ba_solves_per_rank"][0] > 0
    assert c5["alone"]["dual_frames_s"] > 0 and c5["alone"]["ba_iters_s"] > 0
    assert "watchdog" not in out and out["exchange_crosscheck"]["slot_arrays_equal"] is True
