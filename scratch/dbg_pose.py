import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import oracle
pkg = load_pkg(); oracle.build(); oracle.lib()
synth = pkg.synth
pb = synth.pose_problem(n_frames=24, obs_per_frame=350, seed=8)
prob = dict(pb); prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
exp = oracle.pose_optimization(prob); got = pkg.Optimizer.PoseOptimization(pb)
d = np.nonzero((got["n_iters"] != exp["n_iters"]).any(axis=1))[0]
print("frames with different iteration counts", d, got["n_iters"][d], exp["n_iters"][d])
print("max |dt|", np.abs(got["poses"][:, :3] - exp["poses"][:, :3]).max(axis=1).max(), "max |dq|", np.abs(got["poses"][:, 3:] - exp["poses"][:, 3:]).max())
print("per-frame dt", np.abs(got["poses"][:, :3] - exp["poses"][:, :3]).max(axis=1))
print("outlier flips", int(np.sum(got["outlier"] != exp["outlier"])), "inliers diff", np.abs(got["n_inliers"] - exp["n_inliers"]).max())
import time
t0=time.perf_counter(); 
for _ in range(5): pkg.Optimizer.PoseOptimization(pb)
print("gpu ms per batch of 24 frames", (time.perf_counter()-t0)/5*1e3)
t0=time.perf_counter(); oracle.pose_optimization(prob); print("oracle ms", (time.perf_counter()-t0)*1e3)
