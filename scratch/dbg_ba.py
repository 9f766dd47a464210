import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'oracle'); sys.path.insert(0,'tests')
import __graft_entry__ as e
pkg = e.load_package(); O = e.load_oracle(); synth = pkg.synth
pb = synth.ba_problem(n_poses=12, n_fixed=3, n_points=150, obs_per_point=6, seed=7)
prob = dict(pb); prob["cams"] = [O.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
exp = O.ba_local(prob)
for mode in ("blocked", "reg"):
    if mode == "blocked": os.environ["DCS_BA_FORCE_BLOCKED_LDLT"] = "1"
    else: os.environ.pop("DCS_BA_FORCE_BLOCKED_LDLT", None)
    got = pkg.Optimizer.LocalBundleAdjustment(pb)
    print(mode, "max dt", np.abs(got["poses"][:, :3]-exp["poses"][:, :3]).max(), got["n_iters"], got["n_trials"], exp["n_iters"], exp["n_trials"])
    print("  chi", got["chi2_trace"][:6], exp["chi2_trace"][:6], "ms", got["gpu_ms"])
