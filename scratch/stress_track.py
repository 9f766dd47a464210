"""Randomised sweep of dcs_track_local_map against the oracle's three stages composed on the host (tests/test_gpu_track.py holds the fixed cases).
usage: python scratch/stress_track.py [seconds] [seed]"""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
pkg, O = e.load_package(), e.load_oracle()
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_track.py"))
t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n_cfg, n_frames, bad, worst = time.time(), 0, 0, 0, 0.0
while time.time() - t0 < budget:
    kw = dict(n_frames=int(rng.integers(1, 7)), n_points=int(rng.integers(50, 3000)), n_features=int(rng.integers(60, 2600)), seed=int(rng.integers(0, 1 << 30)),
              th=float(rng.choice([1.0, 1.0, 3.0, 5.0])), pre_matched=float(rng.uniform(0, 0.8)))
    frames, prm = pkg.synth.tracking_problem(**kw)
    prm["nn_ratio"] = float(rng.choice([0.8, 0.8, 0.9, 0.6])); prm["th_high"] = int(rng.choice([100, 100, 50, 150]))
    t._with_grid(pkg, frames)
    got = pkg.abi.PreparedTracking(frames, prm).track()
    for k, fr in enumerate(frames):
        exp = t._oracle_chain(O, fr, prm)
        g = got[k]
        ok = (np.array_equal(g["match_of_point"], exp["match_of_point"]) and np.array_equal(g["point_of_feature"], exp["point_of_feature"]) and
              g["n_matches"] == exp["n_matches"])
        dp = float(np.abs(g["pose"] - exp["pose"]).max())
        soft = g["n_inliers"] != exp["n_inliers"] or not np.array_equal(g["outlier"], exp["outlier"]) or dp > 1e-9
        worst = max(worst, dp)
        if not ok or (soft and dp > 1e-6):
            bad += 1
            print("MISMATCH", kw, "frame", k, "matching equal:", ok, "pose diff", dp, "inliers", g["n_inliers"], exp["n_inliers"], flush=True)
        n_frames += 1
    n_cfg += 1
print("tracking-chain configs %d (%d frames), mismatches %d, largest pose difference %.3g" % (n_cfg, n_frames, bad, worst))
