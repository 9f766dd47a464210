import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle(); synth = pkg.synth
rng = np.random.default_rng(5)
for obs in (2, 6):
    for exact in (False, True):
        dts = []
        for i in range(120):
            P = int(rng.integers(15, 34)); args = dict(n_poses=P, n_fixed=int(rng.integers(1, 6)), n_points=int(rng.integers(150, 450)), obs_per_point=obs,
                                                       seed=int(rng.integers(100000)), outlier_frac=0.2, exact_adjoint=exact)
            pb = synth.ba_problem(**args)
            prob = dict(pb); prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
            g, o = pkg.Optimizer.LocalBundleAdjustment(pb), oracle.ba_local(prob)
            if g["n_iters"] == o["n_iters"] and g["n_trials"] == o["n_trials"]:
                dts.append(np.abs(g["poses"][:, :3] - o["poses"][:, :3]).max())
        dts = np.array(dts)
        print("obs", obs, "exact", exact, "n", len(dts), "median %.2e  p90 %.2e  max %.2e" % (np.median(dts), np.quantile(dts, 0.9), dts.max()), flush=True)
