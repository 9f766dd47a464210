"""The reference constructs an ORBmatcher on the stack of three threads at once (Tracking, LocalMapping, LoopClosing:
SURVEY.md 8(b) "Match" row), so the host-buffer entry points must be re-entrant. Three host threads hammer
dcs_match_bf, dcs_search_by_projection and dcs_search_by_bow concurrently (each on its thread's own stream and scratch
arena: no hipMalloc, no null stream); every result must equal the oracle's, every time."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_three_matcher_threads(pkg, oracle, synth):
    # --- thread A: brute-force match of two real feature sets (Tracking: relocalisation-style BF)
    img0, img1 = synth.frame_pair(640, 480, 0, 0)
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    kp0, d0 = o.extract(img0)
    kp1, d1 = o.extract(img1)
    bi, bd, sd = oracle.knn2(d0, d1)
    exp_bf = oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, kp0["angle"], kp1["angle"])
    # small problem too: below the matrix-core threshold the popcount kernel serves the call
    bis, bds, sds = oracle.knn2(d0[:40], d1[:50])
    exp_small = oracle.ratio_rot_filter(bis, bds, sds, 50, False, 0.75, True, kp0["angle"][:40], kp1["angle"][:50])
    # --- thread B: projection-guided search (Tracking::SearchLocalPoints)
    frame, q = synth.projection_problem(n_per_cam=900, n_queries=700, seed=13)
    frame["grid_off"], frame["grid_idx"] = oracle.frame_grid(frame["cam_off"], frame["kp_x"], frame["kp_y"], frame["min_x"], frame["min_y"],
                                                             frame["grid_w_inv"], frame["grid_h_inv"])
    exp_proj = oracle.search_by_projection(frame, q, 100, 0.8, False)
    # --- thread C: BoW-guided search (LocalMapping / relocalisation)
    fv_kf, fv_f = synth.csr_buckets(len(d0), 100, seed=100), synth.csr_buckets(len(d1), 100, seed=101)
    valid = (np.random.default_rng(3).random(len(d0)) < 0.8).astype(np.uint8)
    exp_bow = oracle.search_by_bow_crosscam(d0, kp0["angle"], valid, d1, kp1["angle"], fv_kf, fv_f, 0.75, True)

    errors, counts = [], [0, 0, 0]
    start = threading.Barrier(3)
    reps = 40

    def guard(fn):
        def run():
            try:
                start.wait()
                fn()
            except Exception as ex:            # noqa: BLE001
                errors.append(repr(ex))
        return run

    def bf():
        m = pkg.ORBmatcher(0.75, True)
        for i in range(reps):
            got, n = m.match_bf(d0, kp0, d1, kp1, 50)
            assert np.array_equal(got, exp_bf[0]) and n == exp_bf[1], "match_bf differs in round %d" % i
            got, n = m.match_bf(d0[:40], kp0[:40], d1[:50], kp1[:50], 50)
            assert np.array_equal(got, exp_small[0]) and n == exp_small[1]
            counts[0] += 1

    def proj():
        m = pkg.ORBmatcher(0.8, True)
        for i in range(reps):
            mq, qf, n = m.SearchByProjection(frame, q, 100, use_ratio=True, check_orientation=False)
            assert np.array_equal(mq, exp_proj[0]) and np.array_equal(qf, exp_proj[1]) and n == exp_proj[2], "projection search differs in round %d" % i
            counts[1] += 1

    def bow():
        m = pkg.ORBmatcher(0.75, True)
        for i in range(reps):
            got, n = m.SearchByBoWCrossCam(d0, kp0["angle"], valid, d1, kp1["angle"], fv_kf, fv_f)
            assert np.array_equal(got, exp_bow[0]) and n == exp_bow[1], "BoW search differs in round %d" % i
            counts[2] += 1

    ths = [threading.Thread(target=guard(f)) for f in (bf, proj, bow)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    assert counts == [reps, reps, reps]
    assert exp_bf[1] > 100 and exp_proj[2] > 100 and exp_bow[1] > 5
