import os, sys, faulthandler, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle(); synth = pkg.synth
faulthandler.dump_traceback_later(25, exit=True)
pb = synth.ba_problem(n_poses=15, n_fixed=4, n_points=300, obs_per_point=6, seed=46366, outlier_frac=0.0, exact_adjoint=False)
prob = dict(pb); prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
print("local gpu", flush=True); g = pkg.Optimizer.LocalBundleAdjustment(pb)
print("local oracle", flush=True); o = oracle.ba_local(prob)
print(g["n_iters"], o["n_iters"], flush=True)
prob.update(iters1=1, iters2=0, huber_delta=0.0)
print("global gpu", flush=True); g = pkg.Optimizer.BundleAdjustment(pb, nIterations=1, bRobust=False)
print("global oracle", flush=True); o = oracle.ba_local(prob)
print(g["n_iters"], o["n_iters"], flush=True)
