"""Times dcs_ba_local on windows beyond the one-workgroup factorisation: 60 free poses (n = 360) and a global-BA shape (200 KF / 20 000 MP);
prints LM iterations per second and the factorisation's share (dcs_ba_timing). usage: time_ba_large.py [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cases = [("p60", dict(n_poses=67, n_fixed=6, n_points=2400, obs_per_point=8, seed=61), None),
         ("global_ba", dict(n_poses=200, n_fixed=1, n_points=20000, obs_per_point=8, seed=5), 5)]
for name, kw, gba_iters in cases:
    pb = synth.ba_problem(**kw)
    if gba_iters:
        pb = dict(pb); pb["iters1"], pb["iters2"] = gba_iters, 0; pb["huber_delta"] = float(np.float32(np.sqrt(3.99)))
    prep = pkg.Optimizer.prepare(pb)
    for _ in range(2): r = prep.solve(None)
    pkg.Optimizer.timing(True)
    t0 = time.perf_counter()
    for _ in range(reps): r = prep.solve(None)
    dt = (time.perf_counter() - t0) / reps
    tm = pkg.Optimizer.timing(False)
    its = sum(r["n_iters"]); tr = sum(r["n_trials"])
    print("%s: n = %d, %d edges: %.2f ms per solve, %d LM iterations (%d trials) -> %.1f it/s; factorisations %.2f ms per solve in %d launches (%.0f us per trial)"
          % (name, 6 * int((pb["pose_fixed"] == 0).sum()), len(pb["edge_pose"]), dt * 1e3, its, tr, its / dt, tm["ldlt_us"] / reps / 1e3, tm["ldlt_launches"] / reps, tm["ldlt_us"] / reps / max(tr, 1)))
