"""bench.py's N > 1 entry on CPU: `python bench.py --gpus 2 --selftest-launch` must spawn its own two ranks (re-exec under
torch.distributed.run, the way `--gpus 2` does on the GPU node), bring up the process group (gloo here, RCCL there), run
the feature all-gather of orb-slam2-dualcam_amd/sharding.py and print ONE JSON line with n_gpus = 2.

The two rank-path legs that follow the headline in an N-GPU run -- c3_scaled (BASELINE configs[2]) and c5_node (configs[4]) -- run here too, at
world size 2, with the GPU work replaced by stand-ins of the same shapes: the collectives, the Tracking || LocalMapping thread pair, the node
sums and the JSON keys are the code bench.py executes on the GPU node (tests/test_gpu_bench.py runs the real legs at world size 1 on RCCL)."""
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


C3_KEYS = {"workload", "n_gpus", "steps", "warmup", "kfeatures_s", "dual_frames_s", "ms_per_step", "per_rank_kfeatures_s", "per_rank_features_per_step",
           "allgather_us", "allgather_bytes_per_rank", "exchange"}
C5_KEYS = {"workload", "n_gpus", "window_s", "concurrent", "alone", "concurrent_vs_alone"}


def _check_legs(out, world):
    c3, c5 = out["c3_scaled"], out["c5_node"]
    assert set(c3) == C3_KEYS and set(c5) == C5_KEYS
    assert c3["n_gpus"] == world and len(c3["per_rank_kfeatures_s"]) == world and len(c3["per_rank_features_per_step"]) == world
    assert c3["workload"].startswith("configs[2]") and "1280x720" in c3["workload"]
    # node rate = all ranks' features / the slowest rank's time: never above the sum of the per-rank rates
    assert 0 < c3["kfeatures_s"] <= sum(c3["per_rank_kfeatures_s"]) * 1.001
    assert c3["dual_frames_s"] > 0 and c3["allgather_us"] > 0
    assert c3["allgather_bytes_per_rank"] == 2 * (2096 * 60 + 64)          # 2 camera slots of the 2000-feature capacity: the C3 payload of SURVEY 8(e)
    assert c5["workload"].startswith("configs[4]") and c5["n_gpus"] == world
    for side in ("concurrent", "alone"):
        assert len(c5[side]["per_rank_dual_frames_s"]) == world and len(c5[side]["per_rank_ba_iters_s"]) == world
        assert abs(c5[side]["dual_frames_s"] - sum(c5[side]["per_rank_dual_frames_s"])) < 0.1 * world
        assert abs(c5[side]["ba_iters_s"] - sum(c5[side]["per_rank_ba_iters_s"])) < 0.1 * world
    assert all(n > 0 for n in c5["concurrent"]["ba_solves_per_rank"])     # the solver thread really ran next to the front end on every rank
    assert set(c5["concurrent_vs_alone"]) == {"front_end", "ba"}


def test_bench_gpus_2_spawns_its_ranks():
    out = _run(["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["allgather_ok"] is True and out["units_of_rank0"] == [0, 2, 4]
    _check_legs(out, 2)


def test_bench_selftest_without_rank_legs():
    out = _run(["--gpus", "2", "--no-rank-legs"])
    assert out["n_gpus"] == 2 and "c3_scaled" not in out and "c5_node" not in out


def test_bench_single_rank_selftest():
    out = _run([])
    assert out["n_gpus"] == 1 and out["allgather_ok"] is True
    _check_legs(out, 1)
