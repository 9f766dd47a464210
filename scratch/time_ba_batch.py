"""Times dcs_ba_local (1 problem) and dcs_ba_local_batch (B problems) on C4-shaped problems."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_pkg
pkg = load_pkg()
import importlib
synth = importlib.import_module("orb_slam2_dualcam_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
preps = [pkg.Optimizer.prepare(synth.ba_problem(seed=42 + s)) for s in range(B)]
for nb in sorted({1, 2, 4, B}):
    sub = preps[:nb]
    for _ in range(3):
        r = pkg.Optimizer.LocalBundleAdjustmentBatch(sub)
    t0 = time.perf_counter()
    for _ in range(reps):
        r = pkg.Optimizer.LocalBundleAdjustmentBatch(sub)
    dt = (time.perf_counter() - t0) / reps
    its = sum(sum(x["n_iters"]) for x in r)
    tr = sum(sum(x["n_trials"]) for x in r)
    print("B=%d: %.3f ms per call, %d LM iterations (%d trials) -> %.0f it/s aggregate; gpu_ms %.3f" % (nb, dt * 1e3, its, tr, its / dt, r[0]["gpu_ms"]))
