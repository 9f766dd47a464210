import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
pb = synth.ba_problem()
prep = pkg.Optimizer.prepare(pb)
prep.solve()
import ctypes as C
L = pkg.abi.lib()
t0 = time.perf_counter()
for _ in range(10):
    r = prep.solve()
t1 = time.perf_counter()
for _ in range(10):
    L.dcs_ba_local(C.byref(prep.pb), None, C.byref(prep.res))
t2 = time.perf_counter()
print("wall per solve ms (wrapper)", (t1-t0)/10*1e3, "(bare ctypes)", (t2-t1)/10*1e3, "opt ms", r["gpu_ms"], r["n_iters"], r["n_trials"])
