"""dcs_ba_local_batch of 8 C4-shaped problems under different k_schur workgroup widths (options DCS_BA_SCHUR_WIDE / _MID) and group counts"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
preps = [pkg.Optimizer.prepare(synth.ba_problem(seed=42 + s)) for s in range(8)]
for rnd in range(2):
    for g, wide, mid in ((0, 2, 4), (0, 4, 4), (0, 0, 4), (0, 0, 0), (0, 8, 8), (4, 2, 4), (4, 2, 2), (4, 0, 0)):
        pkg.abi.set_option("DCS_BA_GROUPS", g); pkg.abi.set_option("DCS_BA_SCHUR_WIDE", wide); pkg.abi.set_option("DCS_BA_SCHUR_MID", mid)
        for _ in range(3): r = pkg.Optimizer.LocalBundleAdjustmentBatch(preps)
        t0 = time.perf_counter()
        for _ in range(reps): r = pkg.Optimizer.LocalBundleAdjustmentBatch(preps)
        dt = (time.perf_counter() - t0) / reps
        its = sum(sum(x["n_iters"]) for x in r)
        print("groups %d wide %d mid %d: %.3f ms per call -> %.0f it/s" % (g, wide, mid, dt * 1e3, its / dt))
