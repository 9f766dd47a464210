#!/usr/bin/env python3
"""Golden vectors produced by the REAL reference code (oracle/_ref/libref.so = DBoW2's BowVector.cpp / FeatureVector.cpp /
ScoringObject.cpp and the Hamming loops of ORBmatcher.cc / FORB.cpp compiled from /root/reference, see oracle/Makefile `ref`).
Run in the build container (the reference tree does not travel); the .npz holds inputs and the reference's outputs only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

assert O.ref() is not None, "needs /root/reference"
rng = np.random.default_rng(2024)
out = {}
# Hamming distances: random pairs, near pairs, extremes
a = rng.integers(0, 256, (512, 32), dtype=np.uint8)
b = a.copy()
flip = rng.integers(0, 256, (512, 32), dtype=np.uint8) & rng.integers(0, 256, (512, 32), dtype=np.uint8) & rng.integers(0, 256, (512, 32), dtype=np.uint8)
b[:256] ^= flip[:256]
b[256:500] = rng.integers(0, 256, (244, 32), dtype=np.uint8)
a[500:] = 0; b[500:506] = 255; b[506:] = 0
out["ham_a"], out["ham_b"] = a, b
out["ham_orbmatcher"], out["ham_forb"] = O.ref_distances(a, b)
# BowVector accumulation + normalisation (transform's addWeight / addIfNotExist in feature order) and FeatureVector
for case, (n, n_words) in enumerate([(1000, 400), (37, 1000000), (2000, 50), (1, 5)]):
    word = rng.integers(0, n_words, n).astype(np.uint32)
    wtab = rng.uniform(0.01, 12.0, n_words if n_words < 5000 else 1)
    weight = wtab[word % len(wtab)]
    for ine in (0, 1):
        for norm in (0, 1, 2):
            w, v = O.ref_bow_vector(word, weight, ine, norm)
            out["bow%d_%d_%d_word" % (case, ine, norm)], out["bow%d_%d_%d_val" % (case, ine, norm)] = w, v
    out["bow%d_in_word" % case], out["bow%d_in_weight" % case] = word, weight
    node = rng.integers(0, max(n_words // 7, 2), n).astype(np.uint32)
    fn, fo, fi = O.ref_feature_vector(node)
    out["fv%d_in" % case], out["fv%d_node" % case], out["fv%d_off" % case], out["fv%d_idx" % case] = node, fn, fo, fi
# scores of every ScoringType on L1-normalised sparse vectors (the state DBoW2 scores in)
vecs = []
for n in (300, 300, 40, 1200, 1, 0):
    w = np.unique(rng.integers(0, 2000, n)).astype(np.uint32)
    v = rng.uniform(0.0, 1.0, len(w))
    if len(w):
        v /= v.sum()
    vecs.append((w, v))
for i, (w, v) in enumerate(vecs):
    out["sv%d_word" % i], out["sv%d_val" % i] = w, v
sc = np.zeros((6, len(vecs), len(vecs)))
for k in range(6):
    for i, (w1, v1) in enumerate(vecs):
        for j, (w2, v2) in enumerate(vecs):
            sc[k, i, j] = O.ref_score(k, w1, v1, w2, v2)
out["scores"] = sc
np.savez_compressed(os.path.join(HERE, "ref_dbow2.npz"), **out)
print("ref_dbow2.npz:", len(out), "arrays")

# ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1969-2010, the reference's own statements) on rotation histograms:
# random ones, ties, empty ones and the 0.1f * max1 boundaries
tm = np.random.default_rng(2025)
hist = [tm.integers(0, 40, 30), tm.integers(0, 3, 30), np.zeros(30, np.int64), np.full(30, 7), np.eye(1, 30, 4, dtype=np.int64)[0] * 9]
for m1, m2, m3 in [(10, 1, 1), (10, 2, 1), (20, 2, 1), (20, 2, 2), (30, 3, 2), (100, 10, 9), (100, 11, 10), (50, 5, 5), (50, 6, 4)]:
    h = np.zeros(30, np.int64); h[[3, 17, 29]] = (m1, m2, m3); hist.append(h)
    h = np.zeros(30, np.int64); h[[29, 0, 1]] = (m1, m2, m3); hist.append(h)
hist += [tm.integers(0, 12, 30) for _ in range(200)]
hist = np.stack(hist).astype(np.int32)
ind = np.array([O.ref_three_maxima(h) for h in hist], np.int32)
np.savez_compressed(os.path.join(HERE, "ref_three_maxima.npz"), hist=hist, ind=ind)
print("ref_three_maxima.npz:", hist.shape, "histograms")

# g2o's RobustKernelHuber (Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-91, the reference's own statements): the deltas the reference
# sets (sqrt(5.991) in LocalBundleAdjustment / PoseOptimization, sqrt(7.815) stereo, the float sqrt(3.99) of BundleAdjustment) and chi2
# values around delta^2, tiny, huge and exactly on the boundary
hk = np.random.default_rng(2026)
deltas = np.array([np.sqrt(5.991), np.sqrt(7.815), float(np.float32(np.sqrt(3.99))), 1.0, 0.25, 12.5])
es = []
for d in deltas:
    es.append(np.concatenate([[0.0, 1e-300, 1e-12, d * d, np.nextafter(d * d, 0), np.nextafter(d * d, 1e9), 1e12, 1e300],
                              hk.uniform(0, 3 * d * d, 40), d * d * np.exp(hk.uniform(-8, 8, 40))]))
es = np.stack(es)
rho = np.array([[O.ref_huber(d, e) for e in row] for d, row in zip(deltas, es)])
np.savez_compressed(os.path.join(HERE, "ref_huber.npz"), delta=deltas, e=es, rho=rho)
print("ref_huber.npz:", rho.shape)
