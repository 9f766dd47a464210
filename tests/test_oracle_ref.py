"""The oracle against the REAL reference where the reference compiles without OpenCV / Eigen (oracle/_ref, see oracle/Makefile
target `ref` and oracle/ref_shim.cpp): DBoW2's BowVector.cpp, FeatureVector.cpp, ScoringObject.cpp and the Hamming loops of
ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2015-2031) / FORB::distance (FORB.cpp:82-102).
Two layers: (1) golden vectors produced by that library (tests/golden/ref_dbow2.npz, make_golden_ref.py) -- they travel, so the
check also runs where the reference tree does not exist; (2) live comparison on fresh random inputs where it does."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_dbow2.npz")


def _bits(x):
    return np.asarray(x, np.float64).view(np.uint64)


def test_hamming_golden(oracle):
    g = np.load(GOLD)
    d = oracle.descriptor_distances(g["ham_a"], g["ham_b"])
    assert np.array_equal(d, g["ham_orbmatcher"]) and np.array_equal(d, g["ham_forb"])
    pop = np.unpackbits(g["ham_a"] ^ g["ham_b"], axis=1).sum(1)
    assert np.array_equal(d, pop) and d.max() == 256 and d.min() == 0
    bi, bd, sd = oracle.knn2(g["ham_a"][:64], g["ham_b"][:300])                   # the knn2 the matchers are built on uses the same distance
    full = np.unpackbits(g["ham_a"][:64, None, :] ^ g["ham_b"][None, :300, :], axis=2).sum(2)
    assert np.array_equal(bd, full.min(1)) and np.array_equal(bi, full.argmin(1))


def test_bow_vector_and_feature_vector_golden(oracle):
    g = np.load(GOLD)
    for case in range(4):
        word, weight = g["bow%d_in_word" % case], g["bow%d_in_weight" % case]
        for ine in (0, 1):
            for norm in (0, 1, 2):
                w, v = oracle.bow_vector(word, weight, ine, norm)
                assert np.array_equal(w, g["bow%d_%d_%d_word" % (case, ine, norm)])
                assert np.array_equal(_bits(v), _bits(g["bow%d_%d_%d_val" % (case, ine, norm)])), (case, ine, norm)   # bit for bit: same order of additions


def test_l1_score_golden(oracle):
    g = np.load(GOLD)
    n = g["scores"].shape[1]
    vecs = [(g["sv%d_word" % i], g["sv%d_val" % i]) for i in range(n)]
    off = np.cumsum([0] + [len(w) for w, _ in vecs]).astype(np.int32)
    dbw = np.concatenate([w for w, _ in vecs]).astype(np.int32)
    dbv = np.concatenate([v for _, v in vecs])
    for i, (w, v) in enumerate(vecs):
        s = oracle.bow_score_l1(w.astype(np.int32), v, off, dbw, dbv)
        assert np.array_equal(_bits(s), _bits(g["scores"][0, i])), i
    assert abs(g["scores"][0, 0, 0] - 1.0) < 1e-12 and g["scores"][0, 5, 0] == 0.0


def test_oracle_vs_reference_live(oracle):
    """fresh random inputs through oracle/_ref/libref.so itself (only where /root/reference exists: the build container)"""
    if oracle.ref() is None:
        pytest.skip("no reference tree on this machine: the committed golden vectors carry the pin")
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    a = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    b = a ^ (rng.integers(0, 256, (2000, 32), dtype=np.uint8) & rng.integers(0, 256, (2000, 32), dtype=np.uint8))
    d1, d2 = oracle.ref_distances(a, b)
    assert np.array_equal(oracle.descriptor_distances(a, b), d1) and np.array_equal(d1, d2)
    for _ in range(20):
        n, nw = int(rng.integers(1, 3000)), int(rng.integers(2, 5000))
        word = rng.integers(0, nw, n).astype(np.uint32)
        weight = rng.uniform(0.0, 9.0, nw)[word] * (rng.random(n) > 0.05)            # some stopped (zero-weight) words
        for ine in (0, 1):
            for norm in (0, 1, 2):
                w0, v0 = oracle.bow_vector(word, weight, ine, norm)
                w1, v1 = oracle.ref_bow_vector(word, weight, ine, norm)
                assert np.array_equal(w0, w1) and np.array_equal(_bits(v0), _bits(v1))
        w1, v1 = oracle.ref_bow_vector(word, weight, 0, 1)
        word2 = rng.integers(0, nw, n).astype(np.uint32)
        w2, v2 = oracle.ref_bow_vector(word2, rng.uniform(0.1, 9.0, nw)[word2], 0, 1)
        s = oracle.bow_score_l1(w1, v1, np.array([0, len(w2)], np.int32), w2, v2)[0]
        assert _bits(s) == _bits(oracle.ref_score(0, w1, v1, w2, v2))


def test_transform_against_reference_containers(oracle, synth):
    """The oracle's full transform (descent + accumulation) re-assembled with the reference's own BowVector / FeatureVector from
    the per-feature (word, weight, node) it reports: values bit for bit, CSR identical."""
    if oracle.ref() is None:
        pytest.skip("no reference tree on this machine")
    voc = synth.vocabulary(k=6, L=4, seed=5)
    V = oracle.Vocabulary(voc["k"], voc["L"], voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    desc = synth.random_descriptors(1500, seed=9)
    t = V.transform(desc, levelsup=2)
    leaf_w = np.asarray(voc["weight"])[np.asarray(voc["is_leaf"]) != 0]              # word id = rank among the leaves (loadFromTextFile :1432-1437)
    assert (t["word"] >= 0).all()
    w, v = oracle.ref_bow_vector(t["word"].astype(np.uint32), leaf_w[t["word"]], 0, 1)
    assert np.array_equal(w, t["bow_word"]) and np.array_equal(_bits(v), _bits(t["bow_val"]))
    fn, fo, fi = oracle.ref_feature_vector(t["node"].astype(np.uint32))
    assert np.array_equal(fn, t["fv_node"]) and np.array_equal(fo, t["fv_off"]) and np.array_equal(fi, t["fv_idx"])


def test_three_maxima_golden_and_live(oracle):
    """ORBmatcher::ComputeThreeMaxima: the oracle's restatement against the reference's own statements (golden vectors made by
    oracle/_ref, and live where the reference tree exists), including ties and the 0.1f * max1 boundaries."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_three_maxima.npz"))
    for h, ind in zip(g["hist"], g["ind"]):
        assert oracle.three_maxima(h) == tuple(int(v) for v in ind), h
    assert tuple(g["ind"][2]) == (-1, -1, -1) and tuple(g["ind"][3]) == (0, 1, 2) and tuple(g["ind"][4]) == (4, -1, -1)   # empty / all equal (strict ">" keeps the first three bins) / single bin
    if oracle.ref() is None:
        return
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    for _ in range(2000):
        h = rng.integers(0, rng.integers(2, 60), 30).astype(np.int32)
        assert oracle.three_maxima(h) == oracle.ref_three_maxima(h), h


def test_huber_kernel_golden_and_live(oracle):
    """row a15: the Huber kernel every solver of the oracle uses (orc_robust_huber; through it the rho_1 of the GPU's k_linearize /
    k_pose_opt, which the BA parity tests hold against the oracle) against RobustKernelHuber::setDelta + ::robustify of the reference
    (Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-91, its own statements): golden vectors everywhere, live where the reference exists."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_huber.npz"))
    for d, es, rhos in zip(g["delta"], g["e"], g["rho"]):
        for e, rho in zip(es, rhos):
            assert np.array_equal(_bits(oracle.robust_huber(d, e)), _bits(rho)), (d, e)
    if oracle.ref() is not None:
        rng = np.random.default_rng(77)
        for _ in range(2000):
            d = float(rng.uniform(0.05, 20.0))
            e = float(d * d * np.exp(rng.uniform(-10, 10)))
            assert np.array_equal(_bits(oracle.robust_huber(d, e)), _bits(oracle.ref_huber(d, e))), (d, e)
