"""Known-answer tests for the matcher oracle (ORBmatcher.cc:57-59, 162-294, 1969-2031)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_hamming_kats(oracle, synth):
    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0 and oracle.descriptor_distance(z, o) == 256
    for bit in (0, 7, 8, 100, 255):
        a = z.copy()
        a[bit >> 3] |= 1 << (bit & 7)
        assert oracle.descriptor_distance(a, z) == 1
    d = synth.random_descriptors(64, seed=7)
    for i in range(0, 64, 2):
        x = int.from_bytes(d[i].tobytes(), "little") ^ int.from_bytes(d[i + 1].tobytes(), "little")
        assert oracle.descriptor_distance(d[i], d[i + 1]) == bin(x).count("1")


def test_knn2_semantics(oracle, synth):
    q = synth.random_descriptors(50, seed=1)
    t = synth.random_descriptors(70, seed=2)
    t[10] = q[3]; t[40] = q[3]                 # duplicate best: first index wins (strict <), second = same dist
    bi, bd, sd = oracle.knn2(q, t)
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    assert np.array_equal(bd, D.min(1)) and np.array_equal(bi, D.argmin(1))
    srt = np.sort(D, axis=1)
    assert np.array_equal(sd, srt[:, 1])
    assert bi[3] == 10 and bd[3] == 0 and sd[3] == 0
    # masked candidates are skipped; empty candidate set -> idx -1, dist 256/256 (ORBmatcher.cc:208-210)
    mask = np.zeros(70, np.uint8); mask[10] = 1
    assert oracle.knn2(q, t, mask)[0][3] == 40
    bi0, bd0, sd0 = oracle.knn2(q, t[:0])
    assert np.all(bi0 == -1) and np.all(bd0 == 256) and np.all(sd0 == 256)
    bi1, bd1, sd1 = oracle.knn2(q, t[:1])
    assert np.all(bi1 == 0) and np.all(sd1 == 256)
    assert len(oracle.knn2(q[:0], t)[0]) == 0


def test_knn2_grouped(oracle, synth):
    q = synth.random_descriptors(200, seed=3)
    t = synth.random_descriptors(300, seed=4)
    qn, qo, qi = synth.csr_buckets(200, 20, seed=5)
    tn, to, ti = synth.csr_buckets(300, 20, seed=6)
    assert np.array_equal(qn, tn)              # all 20 nodes populated on both sides
    bi, bd, sd = oracle.knn2_grouped(q, t, qo, qi, to, ti)
    for g in range(len(qn)):
        cand = ti[to[g]:to[g + 1]]
        for i in qi[qo[g]:qo[g + 1]]:
            b, d, s = oracle.knn2(q[i:i + 1], t[cand])
            assert bi[i] == cand[b[0]] and bd[i] == d[0] and sd[i] == s[0]


def test_ratio_rot_filter(oracle):
    # accept: best <= 50 and best < ratio*second (ORBmatcher.cc:233-236); strict variant best < 50 (:366)
    bi = np.array([0, 1, 2, 3, -1], np.int32)
    bd = np.array([50, 51, 30, 30, 256], np.int32)
    sd = np.array([100, 100, 40, 41, 256], np.int32)
    m, n = oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, False)
    assert m.tolist() == [0, -1, -1, 3, -1] and n == 2       # 30 < 0.75*40 is false (30 < 30)
    m, n = oracle.ratio_rot_filter(bi, bd, sd, 50, True, 0.75, False)
    assert m.tolist() == [-1, -1, -1, 3, -1] and n == 1
    # rotation histogram: bins of 12 deg... factor 1/30 on degrees -> bin = round(rot/30), 30 -> 0
    nq = 40
    bi = np.arange(nq, dtype=np.int32); bd = np.full(nq, 10, np.int32); sd = np.full(nq, 100, np.int32)
    qa = np.zeros(nq, np.float32); ta = np.zeros(nq, np.float32)
    qa[:20] = 100.0                     # rot 100 -> bin round(3.33) = 3   (20 votes)
    qa[20:30] = 200.0                   # bin 7                            (10 votes)
    qa[30:38] = 20.0; ta[30:38] = 30.0  # rot -10+360=350 -> round(11.67) = 12 (8 votes)
    qa[38] = 355.0                      # round(11.83) = 12 ... 355/30 = 11.83 -> 12 (1 more)
    qa[39] = 359.0                      # 359/30 = 11.97 -> 12
    m, n = oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, qa, ta)
    assert n == 40 and np.all(m >= 0)   # exactly three populated bins: all kept
    qa[39] = 50.0                       # a 4th bin (round(1.67)=2) with 1 vote: dropped
    m, n = oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, qa, ta)
    assert n == 39 and m[39] == -1
    # second/third maxima dropped when < 0.1 * max (ComputeThreeMaxima :2001-2009)
    qa[:] = 100.0; qa[0] = 200.0
    m, n = oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, qa, ta)
    assert n == 39 and m[0] == -1
    # bin 30 wraps to 0: rot = 359.9 -> round(11.997)=12; rot 900? not reachable; test rot=345 -> 11.5 -> 12 (half away)
    qa[:] = 345.0
    assert oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, qa, ta)[1] == 40


def test_search_by_bow_greedy(oracle, synth):
    """the faithful SearchByBoWCrossCam: already-claimed F features are skipped (:216)."""
    base = synth.random_descriptors(3, seed=8)
    kf = np.stack([base[0], base[0], base[1]])          # two identical KF queries, same node
    f = np.stack([base[0], synth.noisy_copy(base[:1], 30, seed=1)[0], base[2]])
    ang = np.zeros(3, np.float32)
    fv_kf = (np.array([5], np.int32), np.array([0, 3], np.int32), np.array([0, 1, 2], np.int32))
    fv_f = (np.array([5], np.int32), np.array([0, 3], np.int32), np.array([0, 1, 2], np.int32))
    m, n = oracle.search_by_bow_crosscam(kf, ang, np.ones(3, np.uint8), f, ang, fv_kf, fv_f, ratio=0.9, check_ori=False)
    # query 0 claims F0 (dist 0); query 1 cannot see F0 any more and takes F1 (dist 30) if it passes ratio
    assert m[0] == 0 and m[1] == 1 and n == 2
    # invalid KF features (no MapPoint) are skipped; disjoint nodes produce nothing
    m, n = oracle.search_by_bow_crosscam(kf, ang, np.array([0, 1, 1], np.uint8), f, ang, fv_kf, fv_f, ratio=0.9, check_ori=False)
    assert m[0] == 1 and n == 1
    fv_f2 = (np.array([6], np.int32), fv_f[1], fv_f[2])
    assert oracle.search_by_bow_crosscam(kf, ang, np.ones(3, np.uint8), f, ang, fv_kf, fv_f2, 0.9, False)[1] == 0
    # merge-join over several nodes equals the stateless grouped knn2 + filter when no F is contested
    q = synth.random_descriptors(120, seed=20)
    t = synth.noisy_copy(q, 10, seed=21)
    nodes, off, idx = synth.csr_buckets(120, 15, seed=22)
    qa = np.zeros(120, np.float32)
    m, n = oracle.search_by_bow_crosscam(q, qa, np.ones(120, np.uint8), t, qa, (nodes, off, idx), (nodes, off, idx), 0.75, True)
    assert n == 120 and np.array_equal(m, np.arange(120))


def test_match_golden(oracle):
    g = np.load(os.path.join(GOLDEN, "match_small.npz"))
    bi, bd, sd = oracle.knn2(g["q"], g["t"])
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_d"]) and np.array_equal(sd, g["second_d"])
    m, n = oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, g["q_angle"], g["t_angle"])
    assert np.array_equal(m, g["match"]) and n == int(g["n_matches"])


def test_distinctive_descriptors_by_definition(oracle, synth):
    """MapPoint.cc:270-340 restated with numpy: N x N Hamming table, sorted rows, median index (int)(0.5 (N-1))."""
    rng = np.random.default_rng(4)
    pool = synth.random_descriptors(200, seed=9)
    sizes = [0, 1, 2, 3, 6, 7, 20, 33]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    idx = np.concatenate([rng.choice(200, n, replace=False) for n in sizes if n]).astype(np.int32)
    best = oracle.distinctive_descriptors(pool, off, idx)
    bits = np.unpackbits(pool, axis=1).astype(np.int32)
    for p, n in enumerate(sizes):
        if n == 0:
            assert best[p] == -1
            continue
        b = bits[idx[off[p]:off[p + 1]]]
        d = (b[:, None, :] != b[None, :, :]).sum(-1)
        med = np.sort(d, axis=1)[:, int(0.5 * (n - 1))]
        assert best[p] == int(np.argmin(med))


def _grid_np(frame):
    """Frame::PosInGrid restated with numpy (cvRound = round half to even)."""
    cells = {}
    co = frame["cam_off"]
    for c in range(len(co) - 1):
        for i in range(co[c], co[c + 1]):
            px = int(np.rint(np.float32(np.float32(frame["kp_x"][i] - frame["min_x"][c]) * frame["grid_w_inv"][c])))
            py = int(np.rint(np.float32(np.float32(frame["kp_y"][i] - frame["min_y"][c]) * frame["grid_h_inv"][c])))
            if 0 <= px < 64 and 0 <= py < 48:
                cells.setdefault((c, px, py), []).append(i - co[c])
    return cells


def test_projection_search_oracle(oracle, synth):
    """GetFeaturesInArea + SearchByProjection restatement vs an independent numpy version of the same definitions."""
    frame, q = synth.projection_problem(n_per_cam=300, n_queries=220, seed=4)
    off, idx = oracle.frame_grid(frame["cam_off"], frame["kp_x"], frame["kp_y"], frame["min_x"], frame["min_y"],
                                 frame["grid_w_inv"], frame["grid_h_inv"])
    frame["grid_off"], frame["grid_idx"] = off, idx
    cells = _grid_np(frame)
    for (c, px, py), lst in cells.items():
        k = (c * 64 + px) * 48 + py
        assert list(idx[off[k]:off[k + 1]]) == lst
    assert off[-1] == sum(len(v) for v in cells.values()) <= frame["cam_off"][-1]

    def area(c, x, y, r, lo, hi):                          # Frame.cc:316-376 in float32
        x, y, r = np.float32(x), np.float32(y), np.float32(r)
        f32 = np.float32
        x0 = max(0, int(np.floor(f32(f32(f32(x - frame["min_x"][c]) - r) * frame["grid_w_inv"][c]))))
        x1 = min(63, int(np.ceil(f32(f32(f32(x - frame["min_x"][c]) + r) * frame["grid_w_inv"][c]))))
        y0 = max(0, int(np.floor(f32(f32(f32(y - frame["min_y"][c]) - r) * frame["grid_h_inv"][c]))))
        y1 = min(47, int(np.ceil(f32(f32(f32(y - frame["min_y"][c]) + r) * frame["grid_h_inv"][c]))))
        if x0 >= 64 or x1 < 0 or y0 >= 48 or y1 < 0:
            return []
        out = []
        for ix in range(x0, x1 + 1):
            for iy in range(y0, y1 + 1):
                for loc in cells.get((c, ix, iy), []):
                    g = frame["cam_off"][c] + loc
                    o = frame["kp_octave"][g]
                    if (lo > 0 or hi >= 0) and (o < lo or (hi >= 0 and o > hi)):
                        continue
                    if abs(f32(frame["kp_x"][g] - x)) < r and abs(f32(frame["kp_y"][g] - y)) < r:
                        out.append(loc)
        return out
    bits = np.unpackbits(frame["desc"], axis=1).astype(np.int32)
    qbits = np.unpackbits(q["desc"], axis=1).astype(np.int32)
    taken = frame["taken"].copy()
    exp_mq = np.full(len(q["cam"]), -1)
    for i in range(len(q["cam"])):
        if not q["valid"][i]:
            continue
        c = q["cam"][i]
        cand = area(c, q["u"][i], q["v"][i], q["radius"][i], q["min_level"][i], q["max_level"][i])
        assert cand == list(oracle.features_in_area(frame, c, q["u"][i], q["v"][i], q["radius"][i], q["min_level"][i], q["max_level"][i]))
        best = [256, -1, -1]; second = [256, -1]
        for loc in cand:
            g = frame["cam_off"][c] + loc
            if taken[g]:
                continue
            d = int((bits[g] != qbits[i]).sum())
            if d < best[0]:
                second = [best[0], best[1]]; best = [d, frame["kp_octave"][g], g]
            elif d < second[0]:
                second = [d, frame["kp_octave"][g]]
        if best[0] <= 100 and not (best[1] == second[1] and np.float32(best[0]) > np.float32(0.8) * np.float32(second[0])):
            taken[best[2]] = 1; exp_mq[i] = best[2]
    mq, qf, nm = oracle.search_by_projection(frame, q, 100, 0.8, False)
    assert np.array_equal(mq, exp_mq) and nm == int((exp_mq >= 0).sum()) and nm > 60
    assert all(qf[g] == i for i, g in enumerate(mq) if g >= 0)
    # the OnCam variant: no ratio test, rotation histogram keeps the three dominant bins
    mq2, qf2, nm2 = oracle.search_by_projection(frame, q, 100, 0.0, True)
    mq3, _, nm3 = oracle.search_by_projection(frame, q, 100, 0.0, False)
    assert nm2 < nm3 and set(np.nonzero(mq2 >= 0)[0]) <= set(np.nonzero(mq3 >= 0)[0])


def _np_frustum(fr, pts, cos_limit, th):
    """Frame::isInFrustum restated independently with numpy float32 / float64 element-wise arithmetic (vectorised over points)."""
    f32, f64 = np.float32, np.float64
    P, Pn = pts["pos"].astype(f32), pts["normal"].astype(f32)
    n = len(P)
    out = dict(in_view=np.zeros(n, np.uint8), cam=np.full(n, -1, np.int32), u=np.zeros(n, f32), v=np.zeros(n, f32), view_cos=np.zeros(n, f32),
               level=np.zeros(n, np.int32), radius=np.zeros(n, f32))
    todo = pts["candidate"].astype(bool).copy() if pts.get("candidate") is not None else np.ones(n, bool)
    for c in range(len(fr["fx"])):
        R, t, O = fr["Rsw"][c].reshape(3, 3), fr["tsw"][c], fr["Ow"][c]
        pic = []
        for r in range(3):
            t0 = (R[r, 0] * P[:, 0] + R[r, 1] * P[:, 1]) + R[r, 2] * P[:, 2]          # float32, left to right
            pic.append((t0.astype(f64) + f64(t[r])).astype(f32))
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            invz = f32(1.0) / pic[2]
            u = fr["fx"][c] * pic[0] * invz + fr["cx"][c]
            v = fr["fy"][c] * pic[1] * invz + fr["cy"][c]
            po = P - O
            dist = np.sqrt(po[:, 0].astype(f64) ** 2 + po[:, 1].astype(f64) ** 2 + po[:, 2].astype(f64) ** 2).astype(f32)
            dot = po[:, 0].astype(f64) * Pn[:, 0] + po[:, 1].astype(f64) * Pn[:, 1] + po[:, 2].astype(f64) * Pn[:, 2]
            vc = (dot / dist.astype(f64)).astype(f32)
            ratio = pts["max_dist"].astype(f32) / dist
            lvl = np.ceil(np.log(ratio.astype(f64)).astype(f32) / f32(fr["log_scale_factor"]))
        ok = todo & ~(pic[2] < 0) & ~((u < fr["min_x"][c]) | (u > fr["max_x"][c])) & ~((v < fr["min_y"][c]) | (v > fr["max_y"][c]))
        ok &= ~((dist < f32(0.8) * pts["min_dist"]) | (dist > f32(1.2) * pts["max_dist"])) & ~(vc < f32(cos_limit))
        lv = np.clip(np.nan_to_num(lvl, nan=0, posinf=1e6, neginf=-1e6), 0, len(fr["scale_factors"]) - 1).astype(np.int32)
        r = np.where(vc.astype(f64) > 0.998, f32(2.5), f32(4.0)).astype(f32)
        if f32(th) != f32(1.0):
            r = r * f32(th)
        for k, val in (("cam", c), ("u", u), ("v", v), ("view_cos", vc), ("level", lv), ("radius", r * fr["scale_factors"][lv])):
            out[k][ok] = val[ok] if isinstance(val, np.ndarray) else val
        out["in_view"][ok] = 1
        todo &= ~ok
    return out


def test_is_in_frustum_by_hand_and_vs_numpy(oracle, synth):
    f32 = np.float32
    scale = np.array([1, 1.2, 1.44, 1.728], f32)
    eye = np.eye(3, dtype=f32).reshape(9)
    side = np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0]], f32).reshape(9)          # second camera looks along +x
    fr = dict(Rsw=np.stack([eye, side]), tsw=np.zeros((2, 3), f32), Ow=np.zeros((2, 3), f32), fx=[500, 500], fy=[500, 500], cx=[320, 320],
              cy=[240, 240], min_x=[0, 0], max_x=[640, 640], min_y=[0, 0], max_y=[480, 480], log_scale_factor=f32(np.log(f32(1.2))), scale_factors=scale)
    pos = np.array([[0, 0, 5], [0.5, -0.25, 5], [0, 0, -5], [4, 0, 5], [0, 0, 5], [0, 0, 5], [0, 0, 5], [5, 0, 0.01], [0, 0, 5]], f32)
    nrm = np.array([[0, 0, 1], [0, 0, 1], [0, 0, -1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [1, 0, 0], [1, 0, 0], [0, 0, 1]], f32)
    max_d = np.array([7.5, 7.5, 7.5, 7.5, 4.0, 100.0, 7.5, 7.5, 7.5], f32)
    min_d = np.array([1, 1, 1, 1, 1, 80.0, 1, 1, 1], f32)
    cand = np.array([1, 1, 1, 1, 1, 1, 1, 1, 0], np.uint8)
    r = oracle.is_in_frustum(fr, dict(pos=pos, normal=nrm, min_dist=min_d, max_dist=max_d, candidate=cand), 0.5, 1.0)
    # 0: straight ahead; 1: off-centre; 2: behind both cameras' ... (cam1 sees x > 0 only); 3: outside cam0's image; 4: too far (5 > 1.2 * 4);
    # 5: too close (5 < 0.8 * 80); 6: normal perpendicular to the ray; 7: only the side camera sees it; 8: not a candidate
    assert r["in_view"].tolist() == [1, 1, 0, 0, 0, 0, 0, 1, 0]
    assert r["cam"].tolist() == [0, 0, -1, -1, -1, -1, -1, 1, -1]
    assert r["u"][0] == 320 and r["v"][0] == 240 and r["view_cos"][0] == 1
    assert r["u"][1] == f32(500) * f32(0.5) * (f32(1) / f32(5)) + f32(320) and r["v"][1] == f32(500) * f32(-0.25) * (f32(1) / f32(5)) + f32(240)
    assert r["level"][0] == 3 and r["radius"][0] == f32(2.5) * scale[3]           # log(1.5) / log(1.2) = 2.22 -> 3; head-on -> 2.5
    assert r["view_cos"][1] < 0.998 and r["radius"][1] == f32(4.0) * scale[r["level"][1]]
    assert r["level"][7] == 3 and abs(r["u"][7] - (500 * -0.01 / 5 + 320)) < 1e-3 and r["v"][7] == 240      # side camera: x_cam = -z_world
    r3 = oracle.is_in_frustum(fr, dict(pos=pos, normal=nrm, min_dist=min_d, max_dist=max_d, candidate=cand), 0.5, 3.0)
    assert r3["radius"][0] == f32(2.5) * f32(3.0) * scale[3]
    r1 = oracle.is_in_frustum({k: (v[:1] if isinstance(v, (list, np.ndarray)) and k != "scale_factors" else v) for k, v in fr.items()},
                              dict(pos=pos, normal=nrm, min_dist=min_d, max_dist=max_d, candidate=cand), 0.5, 1.0)
    assert r1["in_view"].tolist() == [1, 1, 0, 0, 0, 0, 0, 0, 0]                 # bForAllCam = false: only camera 0
    # the synthetic rig problem against the independent numpy statement
    frame, pts = synth.frustum_problem(n_points=4000, seed=9)
    for cos_limit, th in ((0.5, 1.0), (0.5, 3.0), (0.9, 5.0)):
        a, b = oracle.is_in_frustum(frame, pts, cos_limit, th), _np_frustum(frame, pts, cos_limit, th)
        for k in ("in_view", "cam", "u", "v", "view_cos", "radius"):
            assert np.array_equal(a[k], b[k]), k
        assert np.sum(a["level"] != b["level"]) <= 1                            # logf vs rounded log: an ulp can tip a ceil
    assert a["in_view"].sum() > 50 and (a["cam"] == 1).sum() > 20


def _one_node(n):
    return (np.int32([7]), np.int32([0, n]), np.arange(n, dtype=np.int32))


def test_kfkf_bow_strict_threshold_and_sticky_claims(oracle):
    """SearchByBoWCrossCam(KF1, c1, KF2, c2) (ORBmatcher.cc:297-414): best < TH_LOW is strict (:364), candidates need a MapPoint,
    and a KF2 feature stays claimed (vbMatched2) for later queries."""
    z = np.zeros((1, 32), np.uint8)
    def with_bits(k):
        d = z.copy(); bits = np.zeros(256, np.uint8); bits[:k] = 1; d[0] = np.packbits(bits); return d
    q = np.concatenate([z, z])                                   # two identical queries
    t = np.concatenate([with_bits(50), with_bits(49), with_bits(10)])
    ang = np.zeros(3, np.float32)
    m, n = oracle.search_by_bow_kfkf(q, ang[:2], [1, 1], t, ang, [1, 1, 0], _one_node(2), _one_node(3), 0.99, False)
    # candidate 2 (distance 10) has no MapPoint; query 0 takes candidate 1 (49 < 50, 49 < .99 * 50); query 1 is left with distance 50: not < TH_LOW
    assert m.tolist() == [1, -1] and n == 1
    m, n = oracle.search_by_bow_kfkf(q, ang[:2], [1, 1], t, ang, [1, 1, 1], _one_node(2), _one_node(3), 0.6, False)
    assert m.tolist() == [2, -1] and n == 1                      # 10 < .6 * 49; then best 49, second 50: ratio fails


def test_triangulation_last_tie_wins_and_epipolar_gates(oracle):
    """SearchForTriangulation (ORBmatcher.cc:1253-1427): `dist > bestDist` skips, so of equal distances the LAST candidate of the node
    list that passes the gates wins; candidates near the epipole (:1333) or off the epipolar line (:74-91) never lower bestDist."""
    z = np.zeros((1, 32), np.uint8)
    def with_bits(k):
        d = z.copy(); bits = np.zeros(256, np.uint8); bits[:k] = 1; d[0] = np.packbits(bits); return d
    t = np.concatenate([with_bits(20), with_bits(20), with_bits(5), with_bits(20), with_bits(51)])
    F = np.float32([0, 0, 0, 0, 0, -1, 0, 1, 0])                   # pure x-translation: epipolar lines are the rows y2 = y1
    epi = dict(F12=F, ex=-1000.0, ey=-1000.0, kp1_x=np.float32([100]), kp1_y=np.float32([50]),
               kp2_x=np.float32([90, 80, 70, 300, 60]), kp2_y=np.float32([50, 50.5, 58, 50, 50]), kp2_octave=np.int32([0, 0, 0, 0, 0]),
               level_sigma2=np.float32([1.0]), scale_factors=np.float32([1.0]))
    ang = np.zeros(5, np.float32)
    m, n = oracle.search_for_triangulation(z, ang[:1], [1], t, ang, [1, 1, 1, 1, 1], _one_node(1), _one_node(5), epi, False)
    assert m.tolist() == [3] and n == 1                          # 0, 1, 3 tie at 20 (1 is 0.5 px off the line: 0.25 < 3.84); 2 is 8 px off; 4 is > TH_LOW
    epi2 = dict(epi, ex=300.0, ey=50.0)                          # candidate 3 sits on the epipole
    m, n = oracle.search_for_triangulation(z, ang[:1], [1], t, ang, [1, 1, 1, 1, 1], _one_node(1), _one_node(5), epi2, False)
    assert m.tolist() == [1]
    m, n = oracle.search_for_triangulation(z, ang[:1], [1], t, ang, [1, 0, 1, 0, 1], _one_node(1), _one_node(5), epi, False)
    assert m.tolist() == [0]                                     # features that already hold a MapPoint are not candidates
    m, n = oracle.search_for_triangulation(z, ang[:1], [0], t, ang, [1, 1, 1, 1, 1], _one_node(1), _one_node(5), epi, False)
    assert m.tolist() == [-1] and n == 0


# ---------------------------------------------------------------------------------------------------------------------------
# row a10, the remaining variants: SearchForInitialization (in-loop vMatchedDistance / stealing) and the window searches with
# independent queries (Fuse x2, SearchBySim3CrossCam, SearchByProjection(KF, ...)) incl. KeyFrame::GetFeaturesInArea's index quirk
def _area_np(frame, cells, c, x, y, r, lo=None, hi=None, kf=False):
    """Frame::GetFeaturesInArea (Frame.cc:316-376) / KeyFrame::GetFeaturesInArea (KeyFrame.cc:728-765, kf=True) in float32"""
    f32 = np.float32
    x, y, r = f32(x), f32(y), f32(r)
    x0 = max(0, int(np.floor(f32(f32(f32(x - frame["min_x"][c]) - r) * frame["grid_w_inv"][c]))))
    x1 = min(63, int(np.ceil(f32(f32(f32(x - frame["min_x"][c]) + r) * frame["grid_w_inv"][c]))))
    y0 = max(0, int(np.floor(f32(f32(f32(y - frame["min_y"][c]) - r) * frame["grid_h_inv"][c]))))
    y1 = min(47, int(np.ceil(f32(f32(f32(y - frame["min_y"][c]) + r) * frame["grid_h_inv"][c]))))
    if x0 >= 64 or x1 < 0 or y0 >= 48 or y1 < 0:
        return []
    out = []
    for ix in range(x0, x1 + 1):
        for iy in range(y0, y1 + 1):
            for loc in cells.get((c, ix, iy), []):
                g = frame["cam_off"][c] + loc
                if not kf:
                    o = frame["kp_octave"][g]
                    if (lo > 0 or hi >= 0) and (o < lo or (hi >= 0 and o > hi)):
                        continue
                p = loc if kf else g                                   # KeyFrame.cc:756: mvTotalKeysUn[vCell[j]]
                if abs(f32(frame["kp_x"][p] - x)) < r and abs(f32(frame["kp_y"][p] - y)) < r:
                    out.append(loc)
    return out


def _with_grid(oracle, frame):
    off, idx = oracle.frame_grid(frame["cam_off"], frame["kp_x"], frame["kp_y"], frame["min_x"], frame["min_y"],
                                 frame["grid_w_inv"], frame["grid_h_inv"])
    frame["grid_off"], frame["grid_idx"] = off, idx
    return frame


def _init_np(frame2, cells, q, ratio, check_ori):
    """SearchForInitialization (ORBmatcher.cc:1117-1251) written independently of the oracle: plain Python over numpy bits"""
    f32 = np.float32
    N2, n1 = int(frame2["cam_off"][-1]), len(q["cam"])
    bits2 = np.unpackbits(frame2["desc"], axis=1).astype(np.int32)
    bits1 = np.unpackbits(q["desc"], axis=1).astype(np.int32)
    INT_MAX = 2 ** 31 - 1
    m12 = np.full(n1, -1); m21 = np.full(N2, -1); md = np.full(N2, INT_MAX, np.int64)
    hist = [[] for _ in range(30)]
    nm = 0
    for i in range(n1):
        if not q["valid"][i]:
            continue
        c = q["cam"][i]
        best, best2, bi = INT_MAX, INT_MAX, -1
        for loc in _area_np(frame2, cells, c, q["u"][i], q["v"][i], q["radius"][i], q["min_level"][i], q["max_level"][i]):
            g = frame2["cam_off"][c] + loc
            d = int((bits1[i] != bits2[g]).sum())
            if md[g] <= d:
                continue
            if d < best:
                best2, best, bi = best, d, g
            elif d < best2:
                best2 = d
        if best <= 50 and f32(best) < f32(f32(best2) * f32(ratio)):
            if m21[bi] >= 0:
                m12[m21[bi]] = -1; nm -= 1
            m12[i] = bi; m21[bi] = i; md[bi] = best; nm += 1
            if check_ori:
                rot = f32(q["angle"][i] - frame2["kp_angle"][bi])
                if rot < 0:
                    rot = f32(rot + f32(360))
                t = float(f32(rot * f32(1.0 / 30)))
                b = int(np.floor(t + 0.5))                             # round(): half away from zero, t >= 0
                hist[0 if b == 30 else b].append(i)
    if check_ori:
        sizes = [len(h) for h in hist]
        order = sorted(range(30), key=lambda b: (-sizes[b], b))         # ComputeThreeMaxima: strict >, first bin wins ties
        i1, i2, i3 = order[0], order[1], order[2]
        if sizes[i1] == 0:
            i1 = i2 = i3 = -1
        else:
            if sizes[i2] == 0 or f32(sizes[i2]) < f32(0.1) * f32(sizes[i1]):
                i2 = i3 = -1
            elif sizes[i3] == 0 or f32(sizes[i3]) < f32(0.1) * f32(sizes[i1]):
                i3 = -1
        for b in range(30):
            if b in (i1, i2, i3):
                continue
            for i in hist[b]:
                if m12[i] >= 0:
                    m12[i] = -1; nm -= 1
    return m12, nm


def test_search_for_initialization_oracle(oracle, synth):
    # --- hand-built: one F2 feature, rivals with distances 10, 10, 6, 8 in this order
    f2 = dict(cam_off=np.array([0, 2, 2], np.int32), kp_x=np.array([100, 400], np.float32), kp_y=np.array([100, 300], np.float32),
              kp_octave=np.zeros(2, np.int32), kp_angle=np.zeros(2, np.float32), desc=np.zeros((2, 32), np.uint8), taken=np.zeros(2, np.uint8),
              min_x=np.zeros(2, np.float32), min_y=np.zeros(2, np.float32), grid_w_inv=np.full(2, 0.1, np.float32), grid_h_inv=np.full(2, 0.1, np.float32))
    f2["desc"][1] = 255
    _with_grid(oracle, f2)

    def qdesc(nbits):
        d = np.zeros(32, np.uint8)
        for b in range(nbits):
            d[b >> 3] |= 1 << (b & 7)
        return d
    dists = [10, 10, 6, 8, 60]
    n = len(dists)
    q = dict(valid=np.ones(n, np.uint8), cam=np.zeros(n, np.int32), u=np.full(n, 110, np.float32), v=np.full(n, 95, np.float32),
             radius=np.full(n, 100, np.float32), min_level=np.zeros(n, np.int32), max_level=np.zeros(n, np.int32),
             desc=np.stack([qdesc(d) for d in dists]), angle=np.zeros(n, np.float32))
    m, nm = oracle.search_for_initialization(f2, q, 0.9, False)
    # q0 takes feature 0 at 10; q1 (10 <= 10: candidate skipped, :1176) gets nothing; q2 steals at 6; q3 (8 > 6) skipped; q4 too far
    assert list(m) == [-1, -1, 0, -1, -1] and nm == 1
    # a non-CAP / non-level-0 key point never searches (:1142-1149)
    q["valid"][2] = 0
    m, nm = oracle.search_for_initialization(f2, q, 0.9, False)
    assert list(m) == [-1, -1, -1, 0, -1] and nm == 1            # q0 took it at 10, q3 steals at 8
    # the ratio test against the second-best candidate (:1192): features at distance 9 and 10 -> 9 < (float)10 * 0.9f = 9.0f fails
    f2b = dict(f2); f2b["kp_x"] = np.array([100, 120], np.float32); f2b["kp_y"] = np.array([100, 100], np.float32)
    f2b["desc"] = np.stack([np.zeros(32, np.uint8), qdesc(19)])
    _with_grid(oracle, f2b)
    q1 = {k: v[2:3].copy() for k, v in q.items()}; q1["valid"][:] = 1; q1["desc"] = qdesc(9)[None]
    m, nm = oracle.search_for_initialization(f2b, q1, 0.9, False)   # dist to f0 = 9, to f1 = 10
    assert list(m) == [-1] and nm == 0
    m, nm = oracle.search_for_initialization(f2b, q1, 1.0, False)
    assert list(m) == [0] and nm == 1
    # --- against the independent restatement, with and without the rotation histogram
    for seed in (21, 22):
        f2, q = synth.initialization_problem(n_per_cam=260, seed=seed)
        _with_grid(oracle, f2)
        cells = _grid_np(f2)
        for ori in (False, True):
            m, nm = oracle.search_for_initialization(f2, q, 0.9, ori)
            em, enm = _init_np(f2, cells, q, 0.9, ori)
            assert np.array_equal(m, em) and nm == enm and nm == int((m >= 0).sum()) and nm > 20
    # the stealing rule really fired: some valid queries found an acceptable feature and lost it again
    m_no, _ = oracle.search_for_initialization(f2, q, 0.9, False)
    assert int((m_no >= 0).sum()) < int(q["valid"].sum())


def test_search_in_window_oracle(oracle, synth):
    frame, q = synth.projection_problem(n_per_cam=300, n_queries=260, seed=8, th=3.0)
    frame["taken"][:] = 0
    _with_grid(oracle, frame)
    cells = _grid_np(frame)
    q["max_level"] = (q["min_level"] + 1).astype(np.int32)                   # nPredictedLevel - 1 .. nPredictedLevel
    inv_sigma2 = (1.0 / (np.float32(1.2) ** (2 * np.arange(8)))).astype(np.float32)
    bits = np.unpackbits(frame["desc"], axis=1).astype(np.int32)
    qbits = np.unpackbits(q["desc"], axis=1).astype(np.int32)
    f32 = np.float32

    def expect(kf, chi, th):
        out = np.full(len(q["cam"]), -1); bd = np.full(len(q["cam"]), 256)
        for i in range(len(q["cam"])):
            if not q["valid"][i]:
                continue
            c = q["cam"][i]
            best, bi = 256, -1
            for loc in _area_np(frame, cells, c, q["u"][i], q["v"][i], q["radius"][i], -1, -1, kf=kf):
                g = frame["cam_off"][c] + loc
                o = frame["kp_octave"][g]
                if o < q["min_level"][i] or o > q["max_level"][i]:
                    continue
                if chi is not None:
                    ex, ey = f32(q["u"][i] - frame["kp_x"][g]), f32(q["v"][i] - frame["kp_y"][g])
                    e2 = f32(f32(ex * ex) + f32(ey * ey))
                    if float(f32(e2 * chi[o])) > 5.99:
                        continue
                d = int((bits[g] != qbits[i]).sum())
                if d < best:
                    best, bi = d, g
            bd[i] = best
            if best <= th:
                out[i] = bi
        return out, bd
    n_kf = {}
    for kf in (False, True):
        for chi in (None, inv_sigma2):
            for th in (50, 100):
                m, bd, acc = oracle.search_in_window(frame, q, th, kf, chi)
                em, ebd = expect(kf, chi, th)
                assert np.array_equal(m, em) and np.array_equal(bd, ebd) and acc == int((em >= 0).sum())
                n_kf[(kf, chi is None, th)] = acc
    assert n_kf[(False, True, 100)] > 100                                   # the searches find their targets ...
    assert n_kf[(False, False, 100)] < n_kf[(False, True, 100)]             # ... the chi2 gate removes some ...
    # ... and the KeyFrame quirk matters for camera 1 only (camera 0: local == global)
    m0, _, _ = oracle.search_in_window(frame, q, 100, False, None)
    m1, _, _ = oracle.search_in_window(frame, q, 100, True, None)
    cam = q["cam"]
    assert np.array_equal(m0[cam == 0], m1[cam == 0]) and not np.array_equal(m0[cam == 1], m1[cam == 1])
    # features flagged in `taken` are skipped (vpMatched[idx] of ORBmatcher.cc:508-509 as a snapshot)
    hit = m0[m0 >= 0][:20]
    frame["taken"][hit] = 1
    m2, _, _ = oracle.search_in_window(frame, q, 100, False, None)
    assert not np.isin(m2[m2 >= 0], hit).any()


def test_search_by_projection_kf_oracle(oracle, synth):
    """SearchByProjection(KF, query, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:416-536): the oracle against an independent
    restatement of the loop :494-529 -- KeyFrame window with the local-as-global index, vpMatched skip, octave gate, strict
    dist < bestDist, bestDist <= th, the match taken for the queries that follow."""
    frame, q = synth.projection_problem(n_per_cam=350, n_queries=420, seed=17, th=3.0)
    _with_grid(oracle, frame)
    cells = _grid_np(frame)
    q["max_level"] = (q["min_level"] + 1).astype(np.int32)                   # nPredictedLevel - 1 .. nPredictedLevel
    bits = np.unpackbits(frame["desc"], axis=1).astype(np.int32)
    qbits = np.unpackbits(q["desc"], axis=1).astype(np.int32)
    for th in (50, 100):
        matched = frame["taken"].copy()
        out = np.full(len(q["cam"]), -1)
        who = np.full(len(matched), -1)
        for i in range(len(q["cam"])):
            if not q["valid"][i]:
                continue
            c = q["cam"][i]
            best, bi = 256, -1
            for loc in _area_np(frame, cells, c, q["u"][i], q["v"][i], q["radius"][i], kf=True):
                g = frame["cam_off"][c] + loc
                if matched[g]:
                    continue
                o = frame["kp_octave"][g]
                if o < q["min_level"][i] or o > q["max_level"][i]:
                    continue
                d = int((bits[g] != qbits[i]).sum())
                if d < best:
                    best, bi = d, g
            if best <= th:
                matched[bi] = 1; out[i] = bi; who[bi] = i
        mq, qf, n = oracle.search_by_projection_kf(frame, q, th)
        assert np.array_equal(mq, out) and np.array_equal(qf, who) and n == int((out >= 0).sum())
    assert n > 40 and len(set(out[out >= 0])) == n                           # every feature matched at most once


def test_undistort_points_host_helper_equals_the_oracle(pkg, oracle):
    """dcs_undistort_points (pure host; the arithmetic the device chain runs per key point) against the oracle's restatement of cv::undistortPoints,
    bit for bit: the shipped rig's coefficients (Dual-LenaCV.yaml:12-35), one with tangential terms and k3, and k1 == 0 (copied through, Frame.cc:414)."""
    rng = np.random.default_rng(5)
    xy = np.concatenate([rng.uniform(-20, 680, (5000, 2)), [[0, 0], [640, 0], [0, 480], [640, 480], [326.7993, 262.9017]]]).astype(np.float32)
    for K4, dist in (((558.4684, 560.0944, 326.7993, 262.9017), (-0.3689, 0.1627, 0.0, 0.0)),
                     ((546.598, 546.254, 332.759, 247.385), (-0.361851421593862, 0.140443638558527, 0.0, 0.0)),
                     ((520.9, 521.0, 325.1, 249.7), (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)),
                     ((500.0, 500.0, 320.0, 240.0), (0.0, 0.1, 0.0, 0.0))):
        got = pkg.abi.undistort_points(xy, K4, dist)
        exp = oracle.undistort_points(xy, K4, list(dist) + [0.0] * (5 - len(dist))) if dist[0] != 0.0 else xy
        assert np.array_equal(got.view(np.uint32), np.asarray(exp, np.float32).view(np.uint32)), (K4, dist)
    # the undistorted corners are what Frame::ComputeImageBounds takes its bounds from (Frame.cc:454-476): outside the image for k1 < 0
    c = pkg.abi.undistort_points(np.array([[0, 0], [640, 0], [0, 480], [640, 480]], np.float32), (558.4684, 560.0944, 326.7993, 262.9017), (-0.3689, 0.1627, 0, 0))
    assert c[0, 0] < -50 and c[3, 0] > 690 and c[0, 1] < -40 and c[3, 1] > 500
