"""Aggregate LM iterations/s when T host threads each solve groups of G C4 problems concurrently (own stream each)."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_pkg
pkg = load_pkg()
import importlib
synth = importlib.import_module("orb_slam2_dualcam_amd.synth")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
preps = [pkg.Optimizer.prepare(synth.ba_problem(seed=42 + s)) for s in range(8)]
for T, G in [(1, 1), (1, 8), (2, 4), (4, 2), (8, 1), (4, 1), (2, 1)]:
    its = [0] * T
    def worker(t):
        sub = preps[t * G:(t + 1) * G]
        for _ in range(3):
            pkg.Optimizer.LocalBundleAdjustmentBatch(sub)
        bar.wait()
        for _ in range(reps):
            r = pkg.Optimizer.LocalBundleAdjustmentBatch(sub)
        its[t] = sum(sum(x["n_iters"]) for x in r) * reps
    bar = threading.Barrier(T + 1)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th: x.start()
    bar.wait(); t0 = time.perf_counter()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    print("threads %d x group %d: %.3f ms per round of %d problems -> %.0f it/s aggregate" % (T, G, dt / reps * 1e3, T * G, sum(its) / dt))
