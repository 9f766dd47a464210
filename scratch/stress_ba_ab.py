"""Random local-BA problems of stress_parity2.py's family, GPU vs oracle, BA only (for A/B of two library builds: DCS_LIB_PATH).
usage: stress_ba_ab.py [count] [seed]; prints every problem beyond 1e-4 and the totals."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle(); synth = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = plateau = 0; worst = 0.0
for it in range(n):
    seed = int(rng.integers(1 << 30))
    P = int(rng.integers(3, 44)); F = int(rng.integers(1, max(2, P // 3))); L = int(rng.integers(20, 500)); O = int(rng.integers(2, min(P, 10) + 1))
    args = dict(n_poses=P, n_fixed=F, n_points=L, obs_per_point=O, seed=seed % 100000, outlier_frac=float(rng.choice([0.0, 0.05, 0.2])), exact_adjoint=bool(rng.integers(0, 2)))
    pb = synth.ba_problem(**args)
    if it % 3 == 2:                                   # sparse covisibility: clusters of poses without a common point / a chain (round 4: device-built pair lists)
        ep, el = np.asarray(pb["edge_pose"]), np.asarray(pb["edge_point"])
        keep = ((ep < P // 2) == (el % 2 == 0)) if it % 2 else (np.abs((ep * 7) % P - (el % P)) <= max(2, P // 6))
        if keep.sum() >= 30:
            pb = dict(pb)
            for k in ("edge_pose", "edge_point", "edge_cam", "obs", "inv_sigma2"): pb[k] = np.ascontiguousarray(np.asarray(pb[k])[keep])
            args = dict(args, sparse=int(it % 2))
    prob = dict(pb); prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
    got, exp = pkg.Optimizer.LocalBundleAdjustment(pb), oracle.ba_local(prob)
    dt = np.abs(got["poses"][:, :3] - exp["poses"][:, :3]).max()
    if got["n_trials"] != exp["n_trials"]: plateau += 1
    if not dt < 1e-4: bad += 1; print("BEYOND", args, got["n_trials"], exp["n_trials"], dt, flush=True)
    else: worst = max(worst, dt)
print("problems", n, "beyond 1e-4:", bad, "different trial counts:", plateau, "worst of the rest %.2e" % worst)
