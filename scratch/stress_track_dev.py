"""Randomised sweep of dcs_track_frame_device: mode 0 against dcs_track_local_map on the same frames (bit for bit), mode 1 against the oracle's
stages composed (tests/test_gpu_track.py holds the fixed cases). usage: python scratch/stress_track_dev.py [seconds] [seed]"""
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
t0, n0, n1, f0, f1, bad, worst = time.time(), 0, 0, 0, 0, 0, 0.0
while time.time() - t0 < budget:
    nf, npts, nfeat, seed = int(rng.integers(1, 5)), int(rng.integers(50, 2600)), int(rng.integers(60, 2200)), int(rng.integers(0, 1 << 30))
    cap = nfeat + int(rng.integers(0, 300))
    if rng.random() < 0.5:
        frames, prm = pkg.synth.tracking_problem(n_frames=nf, n_points=npts, n_features=nfeat, seed=seed, th=float(rng.choice([1.0, 3.0])), pre_matched=float(rng.uniform(0, 0.8)))
        t._with_grid(pkg, frames)
        ref = pkg.abi.PreparedTracking(frames, prm).track()
        dfr, keep = t._device_frames(frames, cap=cap, mode=0)
        got = pkg.abi.PreparedTrackingDevice(dfr, prm, mode=0).track()
        for a, b in zip(ref, got):
            ok = all(np.array_equal(a[k], b[k]) for k in ("match_of_point", "point_of_feature", "outlier", "pose")) and a["n_matches"] == b["n_matches"] and a["n_inliers"] == b["n_inliers"]
            bad += not ok; f0 += 1
            if not ok: print('MISMATCH mode 0', nf, npts, nfeat, seed, cap, [k for k in ('match_of_point', 'point_of_feature', 'outlier', 'pose') if not np.array_equal(a[k], b[k])], a['n_matches'], b['n_matches'], a['n_inliers'], b['n_inliers'])
        n0 += 1
    else:
        check = bool(rng.integers(0, 2))
        frames, prm = pkg.synth.motion_model_problem(n_frames=nf, n_points=npts, n_features=nfeat, seed=seed, th=float(rng.choice([7.0, 14.0, 3.0])), seen=float(rng.uniform(0.3, 1.0)))
        prm["th_high"] = int(rng.choice([100, 100, 50]))
        t._with_grid(pkg, frames)
        dfr, keep = t._device_frames(frames, cap=cap, mode=1)
        got = pkg.abi.PreparedTrackingDevice(dfr, prm, mode=1, check_orientation=check).track()
        for k, fr in enumerate(frames):
            exp = t._oracle_motion_model(O, fr, prm, check)
            g = got[k]
            ok = np.array_equal(g["match_of_point"], exp["match_of_point"]) and np.array_equal(g["point_of_feature"], exp["point_of_feature"]) and g["n_matches"] == exp["n_matches"]
            dp = float(np.abs(g["pose"] - exp["pose"]).max())
            worst = max(worst, dp)
            soft = g["n_inliers"] != exp["n_inliers"] or not np.array_equal(g["outlier"], exp["outlier"]) or dp > 1e-9
            if not ok or (soft and dp > 1e-6):
                bad += 1
                print("MISMATCH mode 1", nf, npts, nfeat, seed, check, k, ok, dp)
            f1 += 1
        n1 += 1
print("device chain: mode 0 %d configs (%d frames) bit-equal to the host-buffer chain, mode 1 %d configs (%d frames) vs oracle; mismatches %d, largest mode-1 pose difference %.1e" % (n0, f0, n1, f1, bad, worst))
