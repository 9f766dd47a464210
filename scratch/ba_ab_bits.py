"""GPU box: results of dcs_ba_local / dcs_ba_local_batch under two builds of the library, compared bit for bit.
usage: python scratch/ba_ab_bits.py dump OUT.npz   (run once per DCS_LIB_PATH), then: python scratch/ba_ab_bits.py cmp A.npz B.npz"""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("arrays", len(a.files), "different", bad)
    sys.exit(1 if bad else 0)
pkg = importlib.import_module("orb-slam2-dualcam_amd")
synth = pkg.synth
out = {}
cases = [dict(), dict(seed=5), dict(n_poses=12, n_fixed=3, n_points=300, seed=9), dict(n_poses=30, n_fixed=29, n_points=500, obs_per_point=4, seed=2), dict(n_poses=60, n_fixed=8, n_points=2500, obs_per_point=6, seed=3)]
for i, kw in enumerate(cases):
    try:
        pb = synth.ba_problem(**kw)
    except TypeError:
        continue
    r = pkg.Optimizer.prepare(pb).solve()
    for k in ("poses", "points", "edge_outlier", "n_iters", "n_trials", "lambda", "chi2_trace"):
        if k in r: out["c%d_%s" % (i, k)] = np.asarray(r[k])
np.savez(sys.argv[2], **out)
print("dumped", len(out), "arrays;", {k: v.tolist() for k, v in out.items() if k.endswith("n_iters")})
