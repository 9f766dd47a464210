"""GPU parity of the BoW front half through the C ABI vs the oracle: per-feature words / nodes, BowVector
(word ids and the f64 values bit for bit), FeatureVector (CSR), L1 scores bit for bit; ragged trees, stopped words, ties,
every weighting / normalisation, the batched device API, a full-size 10^6-word vocabulary fed by real ORB descriptors, and the
chain transform -> FeatureVector -> SearchByBoW."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(pkg, oracle, v, scoring=0, weighting=0):
    args = (v["k"], v["L"], v["parent"], v["is_leaf"], v["desc"], v["weight"], scoring, weighting)
    return pkg.ORBVocabulary(*args), oracle.Vocabulary(*args)


def _same(got, exp):
    for key in ("word", "node", "bow_word", "fv_node", "fv_off", "fv_idx"):
        assert np.array_equal(got[key], exp[key]), key
    assert got["bow_val"].tobytes() == exp["bow_val"].tobytes()          # doubles, bit for bit


@pytest.mark.parametrize("seed,levelsup,scoring,weighting", [(1, 2, 0, 0), (2, 1, 0, 0), (3, 4, 0, 1), (4, 0, 1, 0), (5, 3, 5, 0), (6, 2, 0, 2),
                                                            (7, 2, 5, 3), (8, 9, 2, 0)])
def test_transform_ragged_trees(pkg, oracle, synth, seed, levelsup, scoring, weighting):
    v = synth.vocabulary(k=6, L=5, seed=seed, ragged=0.3, early_leaf=0.1, stop_frac=0.1, dup_frac=0.1)
    G, O = _both(pkg, oracle, v, scoring, weighting)
    info = G.info()
    assert info["n_nodes"] == len(v["parent"]) + 1 and info["n_words"] == O.n_words() == int(v["is_leaf"].sum())
    for n in (1, 2, 63, 700, 2049, 4096):
        feats = np.concatenate([synth.descriptors_near_words(v, n - n // 4, seed=seed + n), synth.random_descriptors(max(n // 4, 1), seed=n)])[:n]
        got, exp = G.transform(feats, levelsup), O.transform(feats, levelsup)
        _same(got, exp)
    assert (exp["word"] < 0).any() and len(exp["bow_word"]) < 4096
    assert len(G.transform(feats[:0])["bow_word"]) == 0
    with pytest.raises(pkg.DcsError):
        G.transform(np.zeros((4097, 32), np.uint8))
    G.close()


def test_vocabulary_validation(pkg, synth):
    v = synth.vocabulary(k=3, L=3, seed=1)
    bad = dict(v); bad["parent"] = v["parent"].copy(); bad["parent"][5] = 30            # parent after child
    with pytest.raises(pkg.DcsError):
        pkg.ORBVocabulary(3, 3, bad["parent"], v["is_leaf"], v["desc"], v["weight"])
    leaf = v["is_leaf"].copy(); leaf[0] = 1                                                # flagged leaf but has children
    with pytest.raises(pkg.DcsError):
        pkg.ORBVocabulary(3, 3, v["parent"], leaf, v["desc"], v["weight"])
    with pytest.raises(pkg.DcsError):
        pkg.ORBVocabulary(30, 3, v["parent"], v["is_leaf"], v["desc"], v["weight"])      # loadFromTextFile's header check


def test_full_size_vocabulary_batch_device_and_bow_matching(pkg, oracle, synth, tmp_path):
    """k = 10, L = 6 (1.1 M nodes, 10^6 words, the shape of ORBvoc.txt) on real ORB descriptors of a dual frame + the previous
    one, levelsup = 4 as Frame::ComputeBoW; then SearchByBoW on the FeatureVectors the GPU produced."""
    import torch
    v = synth.vocabulary_fast(10, 6, seed=3)
    G, O = _both(pkg, oracle, v)
    assert G.info() == dict(k=10, L=6, n_nodes=1111111, n_words=1000000)
    imgs = list(synth.frame_pair(640, 480, 0, 0)) + list(synth.frame_pair(640, 480, 0, 1))
    B = len(imgs) + 1                                                                      # + an image without features
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=4)
    cap = e.default_cap()
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    e.extract_batch_device(torch.from_numpy(np.stack(imgs)).cuda(), d_kp, d_desc, d_n, cap, stream=st)
    out = pkg.ORBVocabulary.bow_buffers(B, cap)
    G.transform_device(d_desc, d_n, cap, out, levelsup=4, stream=st)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    assert n[4] == 0 and int(out["bow_n"][4]) == 0 and int(out["fv_n"][4]) == 0 and int(out["fv_off"][4, 0]) == 0
    res = []
    for i in range(4):
        desc = d_desc[i, :n[i]].cpu().numpy()
        exp = O.transform(desc, 4)
        nw, nn = int(out["bow_n"][i]), int(out["fv_n"][i])
        got = dict(word=out["word"][i, :n[i]].cpu().numpy(), node=out["node"][i, :n[i]].cpu().numpy(), bow_word=out["bow_word"][i, :nw].cpu().numpy(),
                   bow_val=out["bow_val"][i, :nw].cpu().numpy(), fv_node=out["fv_node"][i, :nn].cpu().numpy(), fv_off=out["fv_off"][i, :nn + 1].cpu().numpy())
        got["fv_idx"] = out["fv_idx"][i, :got["fv_off"][-1]].cpu().numpy()
        _same(got, exp)
        assert 50 <= nn <= 100 and nw > 0.9 * n[i]                                       # level-2 nodes; nearly every feature its own word
        res.append((desc, d_kp[i, :n[i], 3].cpu().numpy(), got))
    # scores: frame 0 against a database of the 4 BowVectors (CSR), bit for bit, self-score = 1
    db_off = np.concatenate([[0], np.cumsum([len(r[2]["bow_word"]) for r in res])]).astype(np.int32)
    db_w, db_v = np.concatenate([r[2]["bow_word"] for r in res]), np.concatenate([r[2]["bow_val"] for r in res])
    s = pkg.ORBVocabulary.score(res[0][2]["bow_word"], res[0][2]["bow_val"], db_off, db_w, db_v)
    assert s.tobytes() == oracle.bow_score_l1(res[0][2]["bow_word"], res[0][2]["bow_val"], db_off, db_w, db_v).tobytes()
    assert abs(s[0] - 1.0) < 1e-12 and (s[1:] < 0.5).all()
    # SearchByBoW between cam0 at t0 (as the keyframe) and cam0 at t1 (as the frame), fed with the GPU's FeatureVectors
    (dk, ak, fk), (df, af, ff) = res[0], res[2]
    fv_k, fv_f = (fk["fv_node"], fk["fv_off"], fk["fv_idx"]), (ff["fv_node"], ff["fv_off"], ff["fv_idx"])
    valid = np.ones(len(dk), np.uint8)
    exp_m, exp_n = oracle.search_by_bow_crosscam(dk, ak, valid, df, af, fv_k, fv_f, 0.75, True)
    got_m, got_n = pkg.ORBmatcher(0.75, True).SearchByBoWCrossCam(dk, ak, valid, df, af, fv_k, fv_f)
    assert np.array_equal(got_m, exp_m) and got_n == exp_n and exp_n > 50
    e.close(); G.close()


def test_load_from_text_file(pkg, oracle, synth, tmp_path):
    v = synth.vocabulary(k=4, L=3, seed=9, ragged=0.2, stop_frac=0.1)
    path = os.path.join(tmp_path, "voc.txt")
    synth.vocabulary_to_text(v, path)
    G = pkg.ORBVocabulary.loadFromTextFile(path)
    feats = synth.descriptors_near_words(v, 300, seed=2)
    _same(G.transform(feats, 1), oracle.Vocabulary(v["k"], v["L"], v["parent"], v["is_leaf"], v["desc"], v["weight"]).transform(feats, 1))
    G.close()


def test_bow_golden(pkg):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_small.npz"))
    V = pkg.ORBVocabulary(int(g["k"]), int(g["L"]), g["parent"], g["is_leaf"], g["voc_desc"], g["weight"])
    r = V.transform(g["feats"], int(g["levelsup"]))
    _same(r, {k: g[k] for k in ("word", "node", "bow_word", "bow_val", "fv_node", "fv_off", "fv_idx")})
    r2 = V.transform(g["feats"][::2], int(g["levelsup"]))
    off = np.array([0, len(r["bow_word"]), len(r["bow_word"]) + len(r2["bow_word"])], np.int32)
    s = pkg.ORBVocabulary.score(r2["bow_word"], r2["bow_val"], off, np.concatenate([r["bow_word"], r2["bow_word"]]), np.concatenate([r["bow_val"], r2["bow_val"]]))
    assert s.tobytes() == g["score_half_vs_full_and_self"].tobytes()
    V.close()


def test_gpu_l1_score_equals_reference_scoring_object(pkg):
    """dcs_bow_score_l1 against golden scores produced by the REAL DBoW2 L1Scoring::score (oracle/_ref, compiled from
    /root/reference/Thirdparty/DBoW2/DBoW2/ScoringObject.cpp; tests/golden/make_golden_ref.py): f64 values bit for bit."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_dbow2.npz"))
    n = g["scores"].shape[1]
    vecs = [(g["sv%d_word" % i].astype(np.int32), g["sv%d_val" % i]) for i in range(n)]
    off = np.cumsum([0] + [len(w) for w, _ in vecs]).astype(np.int32)
    dbw, dbv = np.concatenate([w for w, _ in vecs]), np.concatenate([v for _, v in vecs])
    for i, (w, v) in enumerate(vecs):
        s = pkg.ORBVocabulary.score(w, v, off, dbw, dbv)
        assert np.array_equal(np.asarray(s).view(np.uint64), g["scores"][0, i].view(np.uint64)), i
