"""GPU parity: Hamming knn2 / grouped knn2 / ratio + rotation filter through the C ABI vs the oracle
(bit-exact: indices, distances, tie-breaking), golden fixture, edge cases."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _eq3(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("nq,nt", [(1000, 1000), (2000, 2037), (1, 1), (63, 257), (65, 255), (300, 1)])
def test_knn2_random(pkg, oracle, synth, nq, nt):
    q = synth.random_descriptors(nq, seed=7)
    t = synth.random_descriptors(nt, seed=8)
    _eq3(pkg.ORBmatcher.knn2(q, t), oracle.knn2(q, t))


def test_knn2_index_field_boundary_and_i8_form(pkg, oracle, synth, opts):
    """the FP4 matrix-core kernel keeps a 14-bit train index inside its float keys: 16 383 descriptors per slot is its largest problem,
    anything above runs the i8 form (22-bit index); the i8 form on the ordinary sizes (option DCS_KNN2_I8 = 1) gives the same arrays"""
    q = synth.random_descriptors(300, seed=11)
    for nt in (16383, 16384, 16500):
        t = synth.random_descriptors(nt, seed=12)
        t[nt - 1] = q[5]; t[nt - 2] = q[5]                       # the best match sits in the last rows, with a tie
        _eq3(pkg.ORBmatcher.knn2(q, t), oracle.knn2(q, t))
    import hashlib
    digests = []
    for i8 in (0, 1):                                            # option DCS_KNN2_I8, read per call
        opts("DCS_KNN2_I8", i8)
        h = hashlib.sha1()
        for nq, nt in ((1000, 1000), (2000, 2037), (65, 255), (257, 4000)):
            for a in pkg.ORBmatcher.knn2(synth.random_descriptors(nq, seed=7), synth.random_descriptors(nt, seed=8)):
                h.update(np.ascontiguousarray(a).tobytes())
        digests.append(h.hexdigest())
    assert digests[0] == digests[1]


def test_knn2_ties_masks_and_empty(pkg, oracle, synth):
    q = synth.random_descriptors(200, seed=1)
    t = np.concatenate([synth.noisy_copy(q, 12, seed=2), q[::-1], q])        # exact duplicates: distance-0 ties
    _eq3(pkg.ORBmatcher.knn2(q, t), oracle.knn2(q, t))
    few = np.tile(synth.random_descriptors(3, seed=3), (100, 1))             # heavy ties across LDS tiles / waves
    _eq3(pkg.ORBmatcher.knn2(q, few), oracle.knn2(q, few))
    mask = (np.random.default_rng(4).random(len(t)) < 0.5).astype(np.uint8)
    _eq3(pkg.ORBmatcher.knn2(q, t, mask), oracle.knn2(q, t, mask))
    _eq3(pkg.ORBmatcher.knn2(q, t, np.ones(len(t), np.uint8)), oracle.knn2(q, t, np.ones(len(t), np.uint8)))
    bi, bd, sd = pkg.ORBmatcher.knn2(q, t[:0])
    assert np.all(bi == -1) and np.all(bd == 256) and np.all(sd == 256)
    assert len(pkg.ORBmatcher.knn2(q[:0], t)[0]) == 0


def test_knn2_grouped(pkg, oracle, synth):
    q = synth.random_descriptors(1000, seed=3)
    t = synth.noisy_copy(q, 20, seed=4)[np.random.default_rng(5).permutation(1000)]
    qn, qo, qi = synth.csr_buckets(1000, 100, seed=5)
    tn, to, ti = synth.csr_buckets(1000, 100, seed=6)
    assert np.array_equal(qn, tn)
    _eq3(pkg.ORBmatcher.knn2_grouped(q, t, qo, qi, to, ti), oracle.knn2_grouped(q, t, qo, qi, to, ti))
    # queries outside every group keep the sentinel
    qo2, qi2 = qo[:11].copy(), qi[:qo[10]].copy()
    got = pkg.ORBmatcher.knn2_grouped(q, t, qo2, qi2, to[:11], ti[:to[10]])
    _eq3(got, oracle.knn2_grouped(q, t, qo2, qi2, to[:11], ti[:to[10]]))
    assert np.sum(got[0] == -1) == 1000 - len(qi2)


def test_filter_and_fused_bf_on_real_features(pkg, oracle, synth):
    img0, img1 = synth.frame_pair(640, 480, 0, 0)
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    kp0, d0 = o.extract(img0)
    kp1, d1 = o.extract(img1)
    bi, bd, sd = oracle.knn2(d0, d1)
    for ratio, ori, th, strict in [(0.75, True, 50, False), (0.6, True, 50, True), (0.9, False, 100, False), (0.75, True, 30, False)]:
        exp_m, exp_n = oracle.ratio_rot_filter(bi, bd, sd, th, strict, ratio, ori, kp0["angle"], kp1["angle"])
        m = pkg.ORBmatcher(ratio, ori)
        got_m, got_n = m.filter(bi, bd, sd, th, strict, kp0["angle"], kp1["angle"])
        assert np.array_equal(got_m, exp_m) and got_n == exp_n
        if not strict:
            fm, fn = m.match_bf(d0, kp0, d1, kp1, th)
            assert np.array_equal(fm, exp_m) and fn == exp_n
    assert exp_n > 20


def test_match_golden(pkg):
    g = np.load(os.path.join(GOLDEN, "match_small.npz"))
    bi, bd, sd = pkg.ORBmatcher.knn2(g["q"], g["t"])
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_d"]) and np.array_equal(sd, g["second_d"])
    m, n = pkg.ORBmatcher(0.75, True).filter(bi, bd, sd, 50, False, g["q_angle"], g["t_angle"])
    assert np.array_equal(m, g["match"]) and n == int(g["n_matches"])


def test_batch_device_pairs(pkg, oracle, synth):
    import torch
    B = 4
    imgs = list(synth.frame_pair(640, 480, 0, 0)) + list(synth.frame_pair(640, 480, 0, 1))
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=B)
    cap = e.default_cap()
    d_img = torch.from_numpy(np.stack(imgs)).cuda()
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=torch.cuda.current_stream().cuda_stream)
    pairs = np.array([[0, 1], [2, 3], [2, 0], [3, 1]], np.int32)        # cam0-cam1 at t0, t1; cam_c(t1) vs cam_c(t0)
    d_pairs = torch.from_numpy(pairs).cuda()
    P = len(pairs)
    d_match = torch.full((P, cap), -7, dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
    d_b = torch.zeros((P, cap), dtype=torch.int32, device="cuda")
    d_s = torch.zeros((P, cap), dtype=torch.int32, device="cuda")
    pkg.ORBmatcher(0.75, True).match_bf_batch_device(d_desc, d_kp, d_n, cap, d_pairs, P, d_match, d_nm, d_b, d_s, 50,
                                                     stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    feats = [oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(im) for im in imgs]
    for p, (a, b) in enumerate(pairs):
        (kq, dq), (kt, dt) = feats[a], feats[b]
        bi, bd, sd = oracle.knn2(dq, dt)
        m, n = oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, kq["angle"], kt["angle"])
        assert int(d_nm[p]) == n
        assert np.array_equal(d_match[p, :len(dq)].cpu().numpy(), m)
        assert np.array_equal(d_b[p, :len(dq)].cpu().numpy(), bd) and np.array_equal(d_s[p, :len(dq)].cpu().numpy(), sd)
    e.close()


def test_distinctive_descriptors(pkg, oracle, synth):
    """MapPoint::ComputeDistinctiveDescriptors batched: lists of 0..90 observations, shared descriptors, ties."""
    rng = np.random.default_rng(21)
    pool = synth.random_descriptors(600, seed=5)
    pool[100:140] = pool[100]                   # identical rows: all-zero distances, the first must win
    pool[140:180] = synth.noisy_copy(np.repeat(pool[140:141], 40, 0), flip_bits=6, seed=3)   # tight cluster
    sizes = [0, 1, 2, 3, 4, 5, 8, 13, 21, 34, 64, 65, 90] + list(rng.integers(1, 30, 80))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    idx = np.concatenate([rng.choice(600, n, replace=False) for n in sizes if n] or [[]]).astype(np.int32)
    idx[off[5]:off[6]] = np.arange(100, 105)    # the all-identical case
    idx[off[8]:off[9]] = np.arange(140, 161)    # the clustered case
    got = pkg.ComputeDistinctiveDescriptors(pool, off, idx)
    exp = oracle.distinctive_descriptors(pool, off, idx)
    assert np.array_equal(got, exp)
    assert got[0] == -1 and got[1] == 0 and got[5] == 0
    with pytest.raises(pkg.DcsError):
        pkg.ComputeDistinctiveDescriptors(pool, off, idx + 1000)


def _proj_problem(pkg, oracle, synth, **kw):
    frame, q = synth.projection_problem(**kw)
    off, idx = pkg.frame_grid(frame["cam_off"], frame["kp_x"], frame["kp_y"], frame["min_x"], frame["min_y"],
                              frame["grid_w_inv"], frame["grid_h_inv"])
    ooff, oidx = oracle.frame_grid(frame["cam_off"], frame["kp_x"], frame["kp_y"], frame["min_x"], frame["min_y"],
                                   frame["grid_w_inv"], frame["grid_h_inv"])
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
    frame["grid_off"], frame["grid_idx"] = off, idx
    return frame, q


@pytest.mark.parametrize("kw", [dict(n_per_cam=900, n_queries=700, seed=13),
                                dict(n_per_cam=2000, n_queries=1500, seed=2, th=3.0),
                                dict(n_per_cam=400, n_queries=300, seed=5, big_windows=25),
                                dict(n_per_cam=2000, n_queries=2600, seed=31),
                                dict(n_per_cam=1800, n_queries=3300, seed=32, th=3.0, big_windows=12)])
def test_search_by_projection(pkg, oracle, synth, kw):
    """ORBmatcher::SearchByProjection (ratio rule) and SearchByProjectionOnCam (best only + rotation histogram): the
    order-dependent greedy result must equal the sequential oracle exactly, including windows beyond the candidate cap -- and queries
    beyond the 2 048 whose state the parallel resolver keeps on chip (round 6: the later ones go through its global-memory path, next to
    on-chip ones whose lists are longer than the 8 words held in LDS)."""
    frame, q = _proj_problem(pkg, oracle, synth, **kw)
    m = pkg.ORBmatcher(0.8, True)
    for use_ratio, ori in ((True, False), (False, True), (False, False)):
        mq, qf, n = m.SearchByProjection(frame, q, 100, use_ratio=use_ratio, check_orientation=ori)
        emq, eqf, en = oracle.search_by_projection(frame, q, 100, 0.8 if use_ratio else 0.0, ori)
        assert np.array_equal(mq, emq) and np.array_equal(qf, eqf) and n == en
        assert n > 0.3 * len(mq)
    # invalid queries and an empty query set
    q0 = {k: v[:0] for k, v in q.items()}
    mq, qf, n = m.SearchByProjection(frame, q0)
    assert n == 0 and len(mq) == 0 and (qf == -1).all()


@pytest.mark.parametrize("kw", [dict(n_per_cam=900, n_queries=700, seed=21), dict(n_per_cam=1500, n_queries=1200, seed=4, th=3.0),
                                dict(n_per_cam=400, n_queries=300, seed=6, big_windows=25)])
def test_search_by_projection_kf(pkg, oracle, synth, kw):
    """SearchByProjection(KF, query, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:416-536, loop closing): the key frame's own window
    -- KeyFrame::GetFeaturesInArea tests the position of mvTotalKeysUn[camera-LOCAL index] (KeyFrame.cc:756), so for the second camera
    another key point's position decides -- with the octave gate in the loop and the sequential vpMatched chain: exact vs the oracle,
    and different from the Frame-window search on the two-camera problem (otherwise the test would not see the quirk)."""
    frame, q = _proj_problem(pkg, oracle, synth, **kw)
    assert len(frame["cam_off"]) == 3 and (q["cam"] == 1).sum() > 50
    q = dict(q)
    q["min_level"] = np.maximum(q["max_level"] - 1, 0).astype(np.int32)                  # nPredictedLevel - 1 .. nPredictedLevel
    m = pkg.ORBmatcher(0.8, True)
    for th in (50, 100):
        mq, qf, n = m.SearchByProjectionKF(frame, q, th)
        emq, eqf, en = oracle.search_by_projection_kf(frame, q, th)
        assert np.array_equal(mq, emq) and np.array_equal(qf, eqf) and n == en
    assert n > 0
    fq, _, _ = m.SearchByProjection(frame, q, 100, use_ratio=False, check_orientation=False)
    cam1 = q["cam"] == 1
    assert not np.array_equal(fq[cam1], mq[cam1])
    # already matched features are skipped and an empty query set clears the map
    fr2 = dict(frame); fr2["taken"] = frame["taken"].copy(); fr2["taken"][emq[emq >= 0][::2]] = 1
    mq2, qf2, n2 = m.SearchByProjectionKF(fr2, q, 100)
    e2 = oracle.search_by_projection_kf(fr2, q, 100)
    assert np.array_equal(mq2, e2[0]) and np.array_equal(qf2, e2[1]) and n2 == e2[2] and not set(mq2[mq2 >= 0]) & set(np.nonzero(fr2["taken"])[0])
    q0 = {k: v[:0] for k, v in q.items()}
    mq, qf, n = m.SearchByProjectionKF(frame, q0)
    assert n == 0 and len(mq) == 0 and (qf == -1).all()


def test_is_in_frustum_and_projection_chain(pkg, oracle, synth):
    """Frame::isInFrustum + PredictScale + window (dcs_is_in_frustum) vs the oracle, float outputs bit for bit; then the chain of
    Tracking::SearchLocalPoints: frustum gate -> projection queries -> SearchByProjection, GPU vs oracle end to end."""
    frame, pts = synth.frustum_problem(n_points=20000, seed=3)
    for cos_limit, th in ((0.5, 1.0), (0.5, 3.0), (0.8, 5.0)):
        got, exp = pkg.isInFrustum(frame, pts, cos_limit, th), oracle.is_in_frustum(frame, pts, cos_limit, th)
        for k in ("in_view", "cam"):
            assert np.array_equal(got[k], exp[k]), k
        for k in ("u", "v", "view_cos"):
            assert got[k].tobytes() == exp[k].tobytes(), k
        flips = int(np.sum(got["level"] != exp["level"]))        # logf (glibc) vs float(log(double)) on the GPU: an ulp can tip the ceil (Q13)
        assert flips <= 1
        same = got["level"] == exp["level"]
        assert got["radius"][same].tobytes() == exp["radius"][same].tobytes()
    assert exp["in_view"].sum() > 500 and (exp["cam"] == 1).sum() > 200
    one = {k: (v[:1] if k not in ("scale_factors", "log_scale_factor") else v) for k, v in frame.items()}      # bForAllCam = false
    g1, e1 = pkg.isInFrustum(one, pts), oracle.is_in_frustum(one, pts)
    assert np.array_equal(g1["in_view"], e1["in_view"]) and (g1["cam"] <= 0).all() and g1["in_view"].sum() < exp["in_view"].sum()
    empty = {k: v[:0] for k, v in pts.items()}
    assert len(pkg.isInFrustum(frame, empty)["in_view"]) == 0
    # chain: features of a synthetic frame placed where visible points project; queries built from the frustum outputs
    fr = pkg.isInFrustum(frame, pts, 0.5, 1.0)
    vis = np.nonzero(fr["in_view"])[0]
    rng = np.random.default_rng(1)
    n_cams = 2
    per_cam = [vis[fr["cam"][vis] == c] for c in range(n_cams)]
    cam_off = np.array([0, len(per_cam[0]), len(per_cam[0]) + len(per_cam[1])], np.int32)
    order = np.concatenate(per_cam)
    N = len(order)
    desc_mp = synth.random_descriptors(len(pts["pos"]), seed=12)
    kp_x = (fr["u"][order] + rng.normal(0, 1.0, N)).astype(np.float32)
    kp_y = (fr["v"][order] + rng.normal(0, 1.0, N)).astype(np.float32)
    octave = np.clip(fr["level"][order] + rng.integers(-1, 2, N), 0, 7).astype(np.int32)
    fdesc = synth.noisy_copy(desc_mp[order], flip_bits=12, seed=4)
    min_x, max_x, min_y, max_y = frame["min_x"], frame["max_x"], frame["min_y"], frame["max_y"]
    pf = dict(cam_off=cam_off, kp_x=kp_x, kp_y=kp_y, kp_octave=octave, kp_angle=np.zeros(N, np.float32), desc=fdesc, taken=np.zeros(N, np.uint8),
              min_x=min_x, min_y=min_y, grid_w_inv=(np.float32(64) / (max_x - min_x)).astype(np.float32),
              grid_h_inv=(np.float32(48) / (max_y - min_y)).astype(np.float32))
    pf["grid_off"], pf["grid_idx"] = pkg.frame_grid(cam_off, kp_x, kp_y, min_x, min_y, pf["grid_w_inv"], pf["grid_h_inv"])
    q = pkg.projection_queries(fr, desc_mp)
    qo = pkg.projection_queries(oracle.is_in_frustum(frame, pts, 0.5, 1.0), desc_mp)
    mq, qf, n = pkg.ORBmatcher(0.8, False).SearchByProjection(pf, q, 100, use_ratio=True, check_orientation=False)
    emq, eqf, en = oracle.search_by_projection(pf, qo, 100, 0.8, False)
    assert np.array_equal(mq, emq) and np.array_equal(qf, eqf) and n == en and n > 0.6 * N


def test_gpu_hamming_equals_reference_descriptor_distance(pkg):
    """Both GPU Hamming kernels (i8 matrix cores, VALU popcount) against distances produced by the reference's own
    ORBmatcher::DescriptorDistance loop (oracle/_ref; golden: tests/golden/ref_dbow2.npz): one query per train row."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_dbow2.npz"))
    a, b, d = g["ham_a"], g["ham_b"], g["ham_orbmatcher"]
    for i in range(0, len(a), 37):                                   # row i of b alone as train set: best distance = d(a_i, b_i)
        bi, bd, sd = pkg.ORBmatcher.knn2(a[i:i + 1], b[i:i + 1])
        assert int(bd[0]) == int(d[i]) and int(bi[0]) == 0
    bi, bd, sd = pkg.ORBmatcher.knn2(a, b)                           # full table: minimum over the reference-checked popcounts
    full = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
    assert np.array_equal(np.diag(full), d) and np.array_equal(bd, full.min(1))


def test_gpu_hamming_against_the_reference_library_live(pkg, oracle):
    """The binary compiled from the reference's own statements (oracle/_ref/libref.so: ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:2015-2031,
    and FORB::distance) travels to the GPU box with the snapshot: fresh random descriptors on every run, the matrix-core matcher's best distances
    against the reference's distances themselves (no golden file in between). Skipped where the binary was not built."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref/libref.so is not on this machine")
    import os
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    a = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    b = a ^ (rng.integers(0, 256, (300, 32), dtype=np.uint8) & rng.integers(0, 256, (300, 32), dtype=np.uint8) & rng.integers(0, 256, (300, 32), dtype=np.uint8))
    d_ref, d_forb = oracle.ref_distances(a, b)                      # d(a_i, b_i) by the reference's two loops
    assert np.array_equal(d_ref, d_forb)
    for i in range(0, len(a), 7):
        bi, bd, sd = pkg.ORBmatcher.knn2(a[i:i + 1], b[i:i + 1])
        assert int(bd[0]) == int(d_ref[i]) and int(bi[0]) == 0, i
    # the full table: every entry of row i is a reference distance d(a_i, b_j) -- computed pair by pair through the library
    bi, bd, sd = pkg.ORBmatcher.knn2(a, b)
    for i in range(0, len(a), 29):
        row = oracle.ref_distances(np.repeat(a[i:i + 1], len(b), 0), b)[0]
        order = np.sort(row)
        assert int(bd[i]) == int(order[0]) and int(sd[i]) == int(order[1]) and int(row[bi[i]]) == int(order[0]), i


def _desc_buckets(desc, n_bits=6):
    """feature vector in which similar descriptors share a node (first descriptor bits), ascending node ids, ascending indices"""
    node = (desc[:, 0] >> (8 - n_bits)).astype(np.int64)
    ids = np.unique(node)
    off, idx = [0], []
    for k in ids:
        idx.extend(np.nonzero(node == k)[0].tolist()); off.append(len(idx))
    return ids.astype(np.int32), np.asarray(off, np.int32), np.asarray(idx, np.int32)


def test_kfkf_bow_and_triangulation_vs_oracle(pkg, oracle, synth):
    """The two remaining BoW-guided matchers with their in-loop state -- SearchByBoWCrossCam(KF1, c1, KF2, c2) (ORBmatcher.cc:297-414:
    strict threshold, sticky vbMatched2, MapPoint masks) and SearchForTriangulation (:1253-1427: best only, last tie, epipole and
    epipolar-line gates of :74-91) -- on real ORB features of consecutive frames, GPU vs oracle, exact."""
    o = oracle.OrbOracle(1500, 1.2, 8, 20, 7)
    f0, f1 = synth.frame_pair(640, 480, 2, 0)[0], synth.frame_pair(640, 480, 2, 1)[0]       # same camera, frame t and t + 1 ((3, 1) px apart)
    kp0, d0 = o.extract(f0)
    kp1, d1 = o.extract(f1)
    rng = np.random.default_rng(5)
    a0, a1 = kp0["angle"].copy(), kp1["angle"].copy()
    tested = 0
    for n_bits, ratio, ori in ((6, 0.75, True), (4, 0.9, True), (8, 0.6, False), (2, 0.75, True)):
        fv0, fv1 = _desc_buckets(d0, n_bits), _desc_buckets(d1, n_bits)
        v0, v1 = (rng.random(len(d0)) < 0.8).astype(np.uint8), (rng.random(len(d1)) < 0.7).astype(np.uint8)
        em, en = oracle.search_by_bow_kfkf(d0, a0, v0, d1, a1, v1, fv0, fv1, ratio, ori)
        gm, gn = pkg.abi.SearchByBoWCrossCamKF(d0, a0, v0, d1, a1, v1, fv0, fv1, ratio, ori)
        assert np.array_equal(gm, em) and gn == en, (n_bits, ratio, ori)
        tested += en
        # triangulation: pure image translation (3, 1) between the frames -> F12 = [t]x, plus noise-level slack from sigma2
        for (tx, ty, ex, ey) in ((3.0, 1.0, -500.0, 240.0), (3.0, 1.2, 320.0, 240.0)):
            F = np.float32([0, 0, ty, 0, 0, -tx, -ty, tx, 0])
            lvl = np.arange(8)
            epi = dict(F12=F, ex=ex, ey=ey, kp1_x=kp0["x"], kp1_y=kp0["y"], kp2_x=kp1["x"], kp2_y=kp1["y"], kp2_octave=kp1["octave"],
                       level_sigma2=(np.float32(1.2) ** lvl).astype(np.float32) ** 2, scale_factors=(np.float32(1.2) ** lvl).astype(np.float32))
            em, en = oracle.search_for_triangulation(d0, a0, 1 - v0, d1, a1, 1 - v1, fv0, fv1, epi, ori)
            gm, gn = pkg.abi.SearchForTriangulation(d0, a0, 1 - v0, d1, a1, 1 - v1, fv0, fv1, epi, ori)
            assert np.array_equal(gm, em) and gn == en, (n_bits, tx, ty, ori)
            tested += en
    assert tested > 200
    # identical descriptors inside one node: order decides (first minimum with claims / last minimum)
    base = synth.random_descriptors(30, seed=8)
    q = np.repeat(base, 3, axis=0)
    t = np.repeat(base, 4, axis=0)
    fvq, fvt = synth.csr_buckets(len(q), 4, seed=1), synth.csr_buckets(len(t), 4, seed=2)
    ones_q, ones_t = np.ones(len(q), np.uint8), np.ones(len(t), np.uint8)
    zq, zt = np.zeros(len(q), np.float32), np.zeros(len(t), np.float32)
    em, en = oracle.search_by_bow_kfkf(q, zq, ones_q, t, zt, ones_t, fvq, fvt, 0.95, False)
    gm, gn = pkg.abi.SearchByBoWCrossCamKF(q, zq, ones_q, t, zt, ones_t, fvq, fvt, 0.95, False)
    assert np.array_equal(gm, em) and gn == en
    epi = dict(F12=np.zeros(9, np.float32) + np.float32([0, 0, 0, 0, 0, -1, 0, 1, 0]), ex=-1e4, ey=-1e4, kp1_x=np.full(len(q), 10, np.float32),
               kp1_y=np.full(len(q), 20, np.float32), kp2_x=np.full(len(t), 30, np.float32), kp2_y=np.full(len(t), 20, np.float32),
               kp2_octave=np.zeros(len(t), np.int32), level_sigma2=np.float32([1]), scale_factors=np.float32([1]))
    em, en = oracle.search_for_triangulation(q, zq, ones_q, t, zt, ones_t, fvq, fvt, epi, False)
    gm, gn = pkg.abi.SearchForTriangulation(q, zq, ones_q, t, zt, ones_t, fvq, fvt, epi, False)
    assert np.array_equal(gm, em) and gn == en and en > 20


@pytest.mark.parametrize("kw", [dict(n_per_cam=900, n_queries=700, seed=31, th=3.0),
                                dict(n_per_cam=2500, n_queries=2000, seed=32, th=6.0),
                                dict(n_per_cam=400, n_queries=300, seed=33, big_windows=25)])
def test_search_in_window_vs_oracle(pkg, oracle, synth, kw):
    """Candidate loops of Fuse x2, SearchBySim3CrossCam and SearchByProjection(KF, vpMapPoints, ...) (ORBmatcher.cc:1431-1556,
    1560-1706, 1713-1965, 693-799): independent queries, KeyFrame::GetFeaturesInArea's index quirk, the loop's octave gate,
    Fuse's chi-square gate -- every combination exact against the oracle, matches and best distances."""
    frame, q = _proj_problem(pkg, oracle, synth, **kw)
    inv_sigma2 = (1.0 / (np.float32(1.2) ** (2 * np.arange(8)))).astype(np.float32)
    m = pkg.ORBmatcher(0.8, True)
    for levels in (0, 1):                                   # pred - 1 .. pred (Fuse, Sim3) / pred - 1 .. pred + 1 (:757)
        q["max_level"] = (q["min_level"] + 1 + levels).astype(np.int32)
        for taken in (False, True):
            fr = dict(frame)
            if not taken:
                fr["taken"] = np.zeros_like(frame["taken"])
            for kf in (True, False):
                for chi in (None, inv_sigma2):
                    for th in (50, 100):
                        mq, bd, n = m.SearchInWindow(fr, q, th=th, kf_area=kf, chi2_inv_sigma2=chi)
                        emq, ebd, en = oracle.search_in_window(fr, q, th, kf, chi)
                        assert np.array_equal(mq, emq) and np.array_equal(bd, ebd) and n == en
    assert en > 0.2 * len(mq)
    fr = dict(frame); fr["taken"] = None                     # no "already matched" map at all
    mq, bd, n = m.SearchInWindow(fr, q, th=100, kf_area=True)
    fr["taken"] = np.zeros_like(frame["taken"])
    emq, ebd, en = oracle.search_in_window(fr, q, 100, True, None)
    assert np.array_equal(mq, emq) and n == en
    q0 = {k: v[:0] for k, v in q.items()}
    mq, bd, n = m.SearchInWindow(frame, q0)
    assert n == 0 and len(mq) == 0
    with pytest.raises(pkg.DcsError):                        # an octave beyond the sigma table must be refused, not read out of bounds
        m.SearchInWindow(frame, q, chi2_inv_sigma2=inv_sigma2[:3])


def test_search_by_sim3_agreement(pkg, oracle, synth):
    """SearchBySim3CrossCam (ORBmatcher.cc:1713-1965): two window searches + the agreement check, against the oracle's two
    directions combined the same way."""
    f1, q21 = _proj_problem(pkg, oracle, synth, n_per_cam=600, n_queries=600, seed=41, th=4.0)      # queries aimed at f1's features
    f2, q12 = _proj_problem(pkg, oracle, synth, n_per_cam=600, n_queries=600, seed=42, th=4.0)      # ... at f2's
    # queries are indexed by the camera-local feature of their own key frame, all of camera 0 here
    for q, fr_other, fr_own in ((q12, f2, f1), (q21, f1, f2)):
        q["cam"][:] = 0; q["max_level"] = (q["min_level"] + 1).astype(np.int32)
    # make direction 2 -> 1 the mirror image of 1 -> 2 for half of the queries so that agreements exist
    m1, _, _ = oracle.search_in_window(f2, q12, 100, True, None)
    for i1 in np.nonzero(m1 >= 0)[0][::2]:
        i2 = int(m1[i1])
        if i2 < len(q21["cam"]):
            q21["u"][i2], q21["v"][i2] = f1["kp_x"][i1], f1["kp_y"][i1]
            q21["desc"][i2] = f1["desc"][i1]; q21["min_level"][i2] = f1["kp_octave"][i1] - 1; q21["max_level"][i2] = f1["kp_octave"][i1]
            q21["radius"][i2] = 12.0; q21["valid"][i2] = 1
    m = pkg.ORBmatcher(0.8, True)
    match12, found = m.SearchBySim3(f1, q12, f2, q21, th=100)
    e1, _, _ = oracle.search_in_window(f2, q12, 100, True, None)
    e2, _, _ = oracle.search_in_window(f1, q21, 100, True, None)
    exp = np.full(len(e1), -1, np.int32)
    for i1 in range(len(e1)):
        if e1[i1] >= 0 and e1[i1] < len(e2) and e2[e1[i1]] == i1:
            exp[i1] = e1[i1]
    assert np.array_equal(match12, exp) and found == int((exp >= 0).sum()) and found > 20


@pytest.mark.parametrize("kw", [dict(n_per_cam=500, seed=21), dict(n_per_cam=1500, seed=23, crowd=0.5),
                                dict(n_per_cam=300, seed=24, window=400.0)])
def test_search_for_initialization_vs_oracle(pkg, oracle, synth, kw):
    """ORBmatcher::SearchForInitialization (ORBmatcher.cc:1117-1251) with its in-loop state -- vMatchedDistance gate, match
    stealing, the stale rotation histogram -- exact against the sequential oracle; window = 400 px overflows the candidate lists."""
    f2, q = synth.initialization_problem(**kw)
    off, idx = pkg.frame_grid(f2["cam_off"], f2["kp_x"], f2["kp_y"], f2["min_x"], f2["min_y"], f2["grid_w_inv"], f2["grid_h_inv"])
    f2["grid_off"], f2["grid_idx"] = off, idx
    for ratio in (0.9, 0.7):
        m = pkg.ORBmatcher(ratio, True)
        for ori in (True, False):
            m12, n = m.SearchForInitialization(f2, q, check_orientation=ori)
            e12, en = oracle.search_for_initialization(f2, q, ratio, ori)
            assert np.array_equal(m12, e12) and n == en
            assert n == int((m12 >= 0).sum()) and n > 0.15 * int(q["valid"].sum())
    f2t = dict(f2); f2t["taken"] = np.ones_like(f2["taken"])          # the frame's "taken" map is not part of this search
    m12b, nb = m.SearchForInitialization(f2t, q, check_orientation=False)
    assert np.array_equal(m12b, m12) and nb == n
    q0 = {k: v[:0] for k, v in q.items()}
    m12, n = m.SearchForInitialization(f2, q0)
    assert n == 0 and len(m12) == 0
