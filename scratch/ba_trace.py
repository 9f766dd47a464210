"""DCS_BA_TRACE breakdown of single and batch-of-8 local BA calls (run on the GPU box)."""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DCS_BA_TRACE"] = "1"
import numpy as np
pkg = importlib.import_module("orb-slam2-dualcam_amd")
synth = pkg.synth
prep = pkg.Optimizer.prepare(synth.ba_problem())
for _ in range(4): prep.solve()
preps8 = [prep] + [pkg.Optimizer.prepare(synth.ba_problem(seed=43 + s)) for s in range(7)]
import ctypes as C
def batch(preps):
    n = len(preps)
    pbs = (C.c_void_p * n)(*[C.addressof(p.pb) for p in preps])
    ress = (C.c_void_p * n)(*[C.addressof(p.res) for p in preps])
    rc = pkg.abi.lib().dcs_ba_local_batch(n, C.cast(pbs, C.c_void_p), None, C.cast(ress, C.c_void_p))
    assert rc == 0
for g in (os.environ.get("BA_GROUP_LIST", "2").split(",")):
    pkg.abi.set_option("DCS_BA_GROUPS", int(g))
    print("groups", g, file=sys.stderr)
    for _ in range(5):
        batch(preps8)
    t0 = time.perf_counter()
    for _ in range(10): batch(preps8)
    print("ms per call %.3f" % ((time.perf_counter() - t0) * 100), file=sys.stderr)
