"""Random local / global BA problems (the family of stress_ba_ab.py) solved by ONE build of the library (DCS_LIB_PATH); writes a digest of
every result (poses, points, flags, chi2, iteration / trial counts, chi2 trace) so that two builds can be compared BIT FOR BIT.
usage: stress_ba_bits.py <count> <seed> <out.txt>;  then: diff a.txt b.txt"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
n, seed0, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(seed0)
lines = []
probs = []
for it in range(n):
    seed = int(rng.integers(1 << 30))
    P = int(rng.integers(3, 44)); F = int(rng.integers(1, max(2, P // 3))); L = int(rng.integers(20, 500)); O = int(rng.integers(2, min(P, 10) + 1))
    args = dict(n_poses=P, n_fixed=F, n_points=L, obs_per_point=O, seed=seed % 100000, outlier_frac=float(rng.choice([0.0, 0.05, 0.2])), exact_adjoint=bool(rng.integers(0, 2)))
    pb = synth.ba_problem(**args)
    if it % 5 == 4: pb = dict(pb); pb["iters1"], pb["iters2"] = int(rng.integers(0, 6)), int(rng.integers(0, 6))
    probs.append(pb)
    r = pkg.Optimizer.LocalBundleAdjustment(pb)
    h = hashlib.sha1()
    for k in ("poses", "points", "edge_outlier", "edge_level1", "edge_chi2", "chi2_trace"): h.update(np.ascontiguousarray(r[k]).tobytes())
    lines.append("%d %s %s %s" % (it, h.hexdigest(), r["n_iters"], r["n_trials"]))
    if len(probs) == 8:                                   # the same problems once more as a batch
        rb = pkg.Optimizer.LocalBundleAdjustmentBatch([pkg.Optimizer.prepare(q) for q in probs])
        h = hashlib.sha1()
        for x in rb:
            for k in ("poses", "points", "edge_outlier", "edge_level1", "edge_chi2", "chi2_trace"): h.update(np.ascontiguousarray(x[k]).tobytes())
        lines.append("batch@%d %s" % (it, h.hexdigest()))
        probs = []
open(out, "w").write("\n".join(lines) + "\n")
print("wrote", len(lines), "digests to", out)
