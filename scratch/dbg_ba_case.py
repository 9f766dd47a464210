import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle(); synth = pkg.synth
args = dict(n_poses=31, n_fixed=1, n_points=326, obs_per_point=3, seed=8141, outlier_frac=0.2, exact_adjoint=True)
pb = synth.ba_problem(**args)
prob = dict(pb); prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
o = oracle.ba_local(prob)
np.set_printoptions(precision=12, linewidth=200)
print("oracle iters", o["n_iters"], o["n_trials"], "lambda", o["lambda_"]); print(o["chi2_trace"][:16])
try:
    g = pkg.Optimizer.LocalBundleAdjustment(pb)
    print("gpu    iters", g["n_iters"], g["n_trials"], "lambda", g["lambda_"]); print(g["chi2_trace"][:16])
    print("rel diff of traces", (g["chi2_trace"][:15] - o["chi2_trace"][:15]) / np.maximum(o["chi2_trace"][:15], 1e-300))
    print("max dt", np.abs(g["poses"][:, :3] - o["poses"][:, :3]).max(), "level1 flips", int((g["edge_level1"] != o["edge_level1"]).sum()))
except pkg.DcsError as ex:
    print("no gpu:", ex)
