"""The solver's host / device protocol under concurrency (config C5: eight LocalMapping threads each call
Optimizer::LocalBundleAdjustment, src/LocalMapping.cc:97-104, while Tracking calls PoseOptimization, src/Tracking.cc:236-269, and
LocalMapping::InterruptBA raises the stop flag asynchronously, src/LocalMapping.cc:141 / Optimizer.cc:582-600).
Each host thread owns a BaContext (thread_local: arena, pinned progress words, streams); a call returns with its queued-ahead steps
still in flight and the next call on the thread drains them. Results of concurrent calls must equal the isolated solves BIT FOR BIT."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = ("poses", "points", "edge_chi2", "edge_outlier", "edge_level1", "chi2_trace")


def _snap(r):
    return {k: np.array(r[k], copy=True) for k in KEYS}, list(r["n_iters"]), list(r["n_trials"])


def _same(r, ref, tag):
    for k in KEYS:
        assert np.array_equal(r[k], ref[0][k]), (tag, k)
    assert list(r["n_iters"]) == ref[1] and list(r["n_trials"]) == ref[2], tag


def _run_threads(fns):
    errors = []
    start = threading.Barrier(len(fns))

    def guard(fn):
        def run():
            try:
                start.wait()
                fn()
            except BaseException as ex:        # noqa: BLE001
                errors.append(repr(ex))
        return run
    ths = [threading.Thread(target=guard(f)) for f in fns]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors


def test_four_threads_each_solving_its_own_problem(pkg, synth):
    shapes = [dict(n_poses=14, n_fixed=3, n_points=220, obs_per_point=6, seed=401), dict(n_poses=18, n_fixed=4, n_points=300, obs_per_point=5, seed=402),
              dict(n_poses=12, n_fixed=2, n_points=150, obs_per_point=7, seed=403), dict(n_poses=24, n_fixed=4, n_points=500, obs_per_point=8, seed=404)]
    preps = [pkg.Optimizer.prepare(synth.ba_problem(**kw)) for kw in shapes]
    refs = [_snap(p.solve()) for p in preps]
    reps, done = 25, [0] * 4

    def worker(i):
        def run():
            for it in range(reps):
                _same(preps[i].solve(), refs[i], ("thread", i, "round", it))
                done[i] += 1
        return run
    _run_threads([worker(i) for i in range(4)])
    assert done == [reps] * 4


def test_batch_and_pose_optimization_interleaved_on_two_threads(pkg, synth):
    preps = [pkg.Optimizer.prepare(synth.ba_problem(n_poses=16, n_fixed=3, n_points=260, obs_per_point=6, seed=500 + s)) for s in range(4)]
    refs = [_snap(p.solve()) for p in preps]
    pose_pb = synth.pose_problem(n_frames=12, obs_per_frame=300, seed=9)
    pose_ref = pkg.Optimizer.PoseOptimization(pose_pb)
    counts = [0, 0]

    def batches():
        for it in range(15):
            out = pkg.Optimizer.LocalBundleAdjustmentBatch(preps)
            for i, r in enumerate(out):
                _same(r, refs[i], ("batch", it, i))
            counts[0] += 1

    def poses():
        for it in range(40):
            got = pkg.Optimizer.PoseOptimization(pose_pb)
            for k in ("poses", "outlier", "n_inliers", "edge_chi2", "n_iters"):
                assert np.array_equal(got[k], pose_ref[k]), ("pose optimisation", it, k)
            counts[1] += 1
    _run_threads([batches, poses])
    assert counts == [15, 40]


def test_stop_flag_raised_mid_solve_then_back_to_back(pkg, synth):
    """InterruptBA from another thread while the solve runs: the call must end early in a consistent state (what Optimizer.cc:597-600
    leaves: no second round, no level-1 set) and the NEXT call on the thread -- whose context still holds the stopped call's queued
    steps -- must equal the undisturbed solve bit for bit. When the stop fell between two accepted iterations the stopped result
    equals a solve limited to that many iterations."""
    pb = synth.ba_problem()                                    # C4: 50 KF / 2000 MP, ~2 ms per solve
    prep = pkg.Optimizer.prepare(pb)
    ref = _snap(prep.solve())
    total_ref = sum(ref[1])
    mid, checked_equiv = 0, 0
    rng = np.random.default_rng(5)
    for it in range(24):
        stop = np.zeros(1, np.uint8)
        delay = float(rng.uniform(0.1e-3, 2.2e-3))
        go = threading.Event()

        def raiser():
            go.wait()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < delay:
                pass
            stop[0] = 1
        th = threading.Thread(target=raiser)
        th.start()
        go.set()
        r = prep.solve(stop)
        th.join()
        s_it, s_tr = list(r["n_iters"]), list(r["n_trials"])
        assert np.all(np.isfinite(r["poses"])) and np.all(np.isfinite(r["points"]))
        assert 0 <= s_it[0] <= ref[1][0] and 0 <= s_it[1] <= ref[1][1]
        if sum(s_it) < total_ref:                               # the flag took effect
            if s_it[0] < ref[1][0] or s_it[1] == 0:             # ... during round 0 (or right at its end): Optimizer.cc:597-600
                assert s_it[1] == 0 and not r["edge_level1"].any()
            if 0 < sum(s_it):
                mid += 1
            if s_it[1] == 0 and 0 < s_it[0] == s_tr[0]:          # every trial accepted: the same estimates as a solve limited to s_it[0] iterations
                lim = dict(pb)
                lim["iters1"], lim["iters2"] = s_it[0], 0
                r2 = pkg.Optimizer.LocalBundleAdjustment(lim)
                assert np.array_equal(r2["poses"], r["poses"]) and np.array_equal(r2["points"], r["points"]) and np.array_equal(r2["edge_outlier"], r["edge_outlier"])
                checked_equiv += 1
        _same(prep.solve(), ref, ("undisturbed solve after a stopped one", it))
    assert mid >= 1, "no stop flag landed inside a solve in 24 attempts"
