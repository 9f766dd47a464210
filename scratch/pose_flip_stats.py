"""How often does k_pose_opt2 end a round of Optimizer::PoseOptimization one LM iteration away from the oracle? (GPU box)
usage: pose_flip_stats.py [n_problems] [seed0]     DCS_LIB_PATH selects a side build, DCS_POSE_FAST=0 the general kernel k_pose_opt,
DCS_POSE_EXACT_EDGE=1 the build of k_pose_opt2 with the oracle's per-edge arithmetic.
Random batches as scratch/stress_parity2.py draws them (2..19 frames, 10..599 observations per frame, 0 / 10 / 30 % outliers) plus larger
frames (600..1900). Reports: batches with any round apart, (frame, round) entries apart, largest pose difference, outlier-flag flips."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth; oracle = e.load_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
batches = bad_batches = entries = bad_entries = two_apart = flips = frames = 0
dmax = 0.0
for i in range(n):
    big = i % 4 == 3
    pp = synth.pose_problem(n_frames=int(rng.integers(2, 8 if big else 20)), obs_per_frame=int(rng.integers(600, 1900) if big else rng.integers(10, 600)),
                            seed=int(rng.integers(0, 100000)), outlier_frac=float(rng.choice([0.0, 0.1, 0.3])))
    prob = dict(pp); prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pp["cams"]]
    exp, got = oracle.pose_optimization(prob), pkg.Optimizer.PoseOptimization(pp)
    dn = np.abs(got["n_iters"] - exp["n_iters"])
    batches += 1; bad_batches += int(dn.max() > 0); entries += dn.size; bad_entries += int(np.count_nonzero(dn)); two_apart += int(np.count_nonzero(dn > 1))
    frames += got["poses"].shape[0]
    flips += int(np.sum(got["outlier"] != exp["outlier"]))
    dmax = max(dmax, float(np.abs(got["poses"] - exp["poses"]).max()))
print("%s DCS_POSE_FAST=%s DCS_POSE_EXACT_EDGE=%s: %d batches (%d frames): %d batches with a round apart (%.1f %%), %d of %d (frame, round) counts apart (%.2f %%), %d by more than one, %d outlier flags differ, largest pose difference %.2e"
      % (os.path.basename(os.path.dirname(os.environ.get("DCS_LIB_PATH", "default/lib"))), os.environ.get("DCS_POSE_FAST", "1"), os.environ.get("DCS_POSE_EXACT_EDGE", "0"), batches, frames, bad_batches, 100.0 * bad_batches / batches,
         bad_entries, entries, 100.0 * bad_entries / entries, two_apart, flips, dmax))
