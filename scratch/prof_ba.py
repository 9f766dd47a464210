import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
pb = synth.ba_problem()
pkg.Optimizer.LocalBundleAdjustment(pb)
t0 = time.perf_counter()
for _ in range(5):
    r = pkg.Optimizer.LocalBundleAdjustment(pb)
print("wall per solve ms", (time.perf_counter()-t0)/5*1e3, "opt ms", r["gpu_ms"], r["n_iters"], r["n_trials"])
