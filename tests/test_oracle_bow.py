"""Oracle known-answer tests for the BoW front half: greedy vocabulary-tree descent, BowVector /
FeatureVector assembly (std::map order, TF-IDF accumulation, L1 normalisation) and the L1 score -- against hand-computed
values and an independent pure-Python statement of DBoW2's definitions (TemplatedVocabulary.h:1151-1283, BowVector.cpp:34-88,
ScoringObject.cpp:23-67). The reference ships no vocabulary and no vectors for this path: parity unpinned."""
import os

import numpy as np


def _tiny():
    """k = 2, L = 2.  rows: 1 = A (all 0x00), 2 = B (all 0xFF), 3 = A0 (0x00..), 4 = A1 (first 4 bytes 0xFF), 5 = B0 (0xFF..),
    6 = B1 (last 4 bytes 0x00). Words in row order: A0 = 0, A1 = 1, B0 = 2, B1 = 3."""
    z, f = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    a1 = z.copy(); a1[:4] = 255
    b1 = f.copy(); b1[-4:] = 0
    parent = np.array([0, 0, 1, 1, 2, 2], np.int32)
    is_leaf = np.array([0, 0, 1, 1, 1, 1], np.uint8)
    desc = np.stack([z, f, z, a1, f, b1])
    weight = np.array([0, 0, 1.0, 2.0, 0.5, 4.0])
    return dict(k=2, L=2, parent=parent, is_leaf=is_leaf, desc=desc, weight=weight), z, f, a1, b1


def test_tiny_tree_by_hand(oracle):
    v, z, f, a1, b1 = _tiny()
    V = oracle.Vocabulary(v["k"], v["L"], v["parent"], v["is_leaf"], v["desc"], v["weight"])
    assert V.n_words() == 4
    half = z.copy(); half[:16] = 255                       # 128 bits from A and from B: tie at level 1 -> first child A
    near_a1 = a1.copy(); near_a1[10] = 1                   # 1 bit from A1
    feats = np.stack([z, near_a1, z, b1, half, a1])
    r = V.transform(feats, levelsup=1)
    # half: under A, distance to A0 = 128, to A1 = 96 -> A1
    assert r["word"].tolist() == [0, 1, 0, 3, 1, 1]
    assert r["node"].tolist() == [1, 1, 1, 2, 1, 1]         # node at level L - levelsup = 1: A = 1, B = 2
    assert r["bow_word"].tolist() == [0, 1, 3]
    raw = np.array([1.0 + 1.0, 2.0 + 2.0 + 2.0, 4.0])       # TF-IDF: the idf weight once per feature
    assert np.array_equal(r["bow_val"], raw / raw.sum())     # L1-normalised
    assert r["fv_node"].tolist() == [1, 2] and r["fv_off"].tolist() == [0, 5, 6] and r["fv_idx"].tolist() == [0, 1, 2, 4, 5, 3]
    # levelsup = 0: the node is the leaf itself; levelsup >= L: the root (:1250-1251)
    assert V.transform(feats, levelsup=0)["node"].tolist() == [3, 4, 3, 6, 4, 4]
    assert V.transform(feats, levelsup=2)["node"].tolist() == [0] * 6
    # a stopped word (weight 0) vanishes from both vectors (:1181)
    w2 = v["weight"].copy(); w2[3] = 0.0
    r2 = oracle.Vocabulary(2, 2, v["parent"], v["is_leaf"], v["desc"], w2).transform(feats, levelsup=1)
    assert r2["word"].tolist() == [0, -1, 0, 3, -1, -1] and r2["bow_word"].tolist() == [0, 3]
    assert np.array_equal(r2["bow_val"], np.array([2.0, 4.0]) / 6.0) and r2["fv_idx"].tolist() == [0, 2, 3]
    # IDF weighting: a word counts once (addIfNotExist); DOT_PRODUCT scoring: no normalisation, TF divides by the number of words
    r3 = oracle.Vocabulary(2, 2, v["parent"], v["is_leaf"], v["desc"], v["weight"], scoring=0, weighting=2).transform(feats, 1)
    assert np.array_equal(r3["bow_val"], np.array([1.0, 2.0, 4.0]) / 7.0)
    r4 = oracle.Vocabulary(2, 2, v["parent"], v["is_leaf"], v["desc"], v["weight"], scoring=5, weighting=0).transform(feats, 1)
    assert np.array_equal(r4["bow_val"], raw / 3.0)
    r5 = oracle.Vocabulary(2, 2, v["parent"], v["is_leaf"], v["desc"], v["weight"], scoring=1, weighting=0).transform(feats, 1)
    assert np.array_equal(r5["bow_val"], raw / np.sqrt((raw * raw).sum()))
    assert len(V.transform(feats[:0])["bow_word"]) == 0


def _py_transform(v, feats, levelsup):
    """DBoW2's definitions, independently in Python (dicts, no shared code with the oracle)."""
    n = len(v["parent"]) + 1
    kids = [[] for _ in range(n)]
    for i, p in enumerate(v["parent"]):
        kids[p].append(i + 1)
    word_of = {}
    for i, leaf in enumerate(v["is_leaf"]):
        if leaf:
            word_of[i + 1] = len(word_of)
    bits = np.unpackbits(v["desc"], axis=1)
    bow, fv, words, nodes = {}, {}, [], []
    for j, f in enumerate(np.unpackbits(feats, axis=1)):
        cur, level, nid = 0, 0, (0 if v["L"] - levelsup <= 0 else None)
        while kids[cur]:
            level += 1
            d = [int(np.sum(bits[c - 1] != f)) for c in kids[cur]]
            cur = kids[cur][int(np.argmin(d))]                 # argmin = first minimum
            if level == v["L"] - levelsup:
                nid = cur
        if nid is None:
            nid = cur
        w = v["weight"][cur - 1]
        if w > 0:
            bow[word_of[cur]] = bow.get(word_of[cur], 0.0) + w
            fv.setdefault(nid, []).append(j)
            words.append(word_of[cur]); nodes.append(nid)
        else:
            words.append(-1); nodes.append(-1)
    keys = sorted(bow)
    vals = [bow[k] for k in keys]
    norm = 0.0
    for x in vals:
        norm += abs(x)
    vals = [x / norm for x in vals] if norm > 0 else vals
    return words, nodes, keys, vals, sorted(fv), [fv[k] for k in sorted(fv)]


def test_random_ragged_tree_vs_definition(oracle, synth):
    for seed, levelsup in [(1, 2), (2, 1), (3, 3), (4, 0)]:
        v = synth.vocabulary(k=4, L=4, seed=seed, ragged=0.3, early_leaf=0.15, stop_frac=0.1, dup_frac=0.15)
        feats = np.concatenate([synth.descriptors_near_words(v, 150, seed=seed), synth.random_descriptors(50, seed=seed)])
        r = oracle.Vocabulary(v["k"], v["L"], v["parent"], v["is_leaf"], v["desc"], v["weight"]).transform(feats, levelsup)
        words, nodes, keys, vals, fnodes, flists = _py_transform(v, feats, levelsup)
        assert r["word"].tolist() == words and r["node"].tolist() == nodes
        assert r["bow_word"].tolist() == keys and np.array_equal(r["bow_val"], np.array(vals))
        assert r["fv_node"].tolist() == fnodes
        assert [r["fv_idx"][a:b].tolist() for a, b in zip(r["fv_off"][:-1], r["fv_off"][1:])] == flists
        assert (np.array(words) < 0).any() and len(keys) < len(feats)            # stopped words and repeated words both occur
        assert abs(r["bow_val"].sum() - 1.0) < 1e-12


def test_l1_score(oracle):
    a_w, a_v = np.array([1, 4, 9, 20], np.int32), np.array([0.1, 0.2, 0.3, 0.4])
    b_w, b_v = np.array([4, 5, 20], np.int32), np.array([0.5, 0.25, 0.25])
    c_w, c_v = np.array([2, 3], np.int32), np.array([0.5, 0.5])
    off = np.array([0, 4, 7, 9, 9], np.int32)
    s = oracle.bow_score_l1(a_w, a_v, off, np.concatenate([a_w, b_w, c_w]), np.concatenate([a_v, b_v, c_v]))
    common = (abs(0.2 - 0.5) - 0.2 - 0.5) + (abs(0.4 - 0.25) - 0.4 - 0.25)     # words 4 and 20, ascending
    assert s[0] == -(((-0.2) + (-0.4)) + (-0.6) + (-0.8)) / 2.0 and abs(s[0] - 1.0) < 1e-15      # identical vectors -> 1
    assert s[1] == -common / 2.0 and s[2] == 0.0 and s[3] == 0.0                                  # disjoint / empty -> 0


def test_text_format_round_trip(oracle, synth, tmp_path):
    v = synth.vocabulary(k=3, L=3, seed=5, ragged=0.2, stop_frac=0.1)
    path = os.path.join(tmp_path, "voc.txt")
    synth.vocabulary_to_text(v, path)
    with open(path) as f:
        assert f.readline().split() == ["3", "3", "0", "0"]
        first = f.readline().split()
    assert len(first) == 2 + 32 + 1 and first[0] == "0"                       # parent, leaf flag, 32 bytes, weight (:1411-1431)
    u = synth.vocabulary_from_text(path)
    for key in ("parent", "is_leaf", "desc", "weight"):
        assert np.array_equal(u[key], v[key]), key
    assert (u["k"], u["L"], u["scoring"], u["weighting"]) == (3, 3, 0, 0)



def test_bow_golden(oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_small.npz"))
    V = oracle.Vocabulary(int(g["k"]), int(g["L"]), g["parent"], g["is_leaf"], g["voc_desc"], g["weight"])
    r = V.transform(g["feats"], int(g["levelsup"]))
    for key in ("word", "node", "bow_word", "fv_node", "fv_off", "fv_idx"):
        assert np.array_equal(r[key], g[key]), key
    assert r["bow_val"].tobytes() == g["bow_val"].tobytes()
    r2 = V.transform(g["feats"][::2], int(g["levelsup"]))
    off = np.array([0, len(r["bow_word"]), len(r["bow_word"]) + len(r2["bow_word"])], np.int32)
    s = oracle.bow_score_l1(r2["bow_word"], r2["bow_val"], off, np.concatenate([r["bow_word"], r2["bow_word"]]), np.concatenate([r["bow_val"], r2["bow_val"]]))
    assert s.tobytes() == g["score_half_vs_full_and_self"].tobytes() and abs(s[1] - 1.0) < 1e-12


# ---------------------------------------------------------------------------------------------------------------------------
# KeyFrameDatabase::DetectLoopCandidatesForCam / DetectRelocalizationCandidates (src/KeyFrameDatabase.cc:111-372)
def _detect_np(loop, qid, qw, qv, db, dead, covis, st, connected, min_score):
    """written from the reference text independently of the oracle: dict inverted files, python lists, numpy float32 scalars"""
    f32 = np.float32
    inv = {}
    for k, (w, _) in enumerate(db):
        if not dead[k]:
            for x in w:
                inv.setdefault(int(x), []).append(k)
    sharing = []
    for x in qw:
        for k in inv.get(int(x), []):
            if st["query"][k] != qid:
                st["words"][k] = 0
                if not (loop and connected[k]):
                    st["query"][k] = qid
                    sharing.append(k)
            st["words"][k] += 1
    if not sharing:
        return []
    mx = max(int(st["words"][k]) for k in sharing)
    mn = int(f32(mx) * f32(0.8))

    def l1(k):
        w, v = db[k]
        dq, dk = dict(zip(map(int, qw), qv)), dict(zip(map(int, w), v))
        s = 0.0
        for x in sorted(set(dq) & set(dk)):
            s += abs(dq[x] - dk[x]) - abs(dq[x]) - abs(dk[x])
        return f32(-s / 2.0)
    sm = []
    for k in sharing:
        if st["words"][k] > mn:
            si = l1(k)
            st["score"][k] = si
            if not loop or si >= f32(min_score):
                sm.append((si, k))
    if not sm:
        return []
    am, best_acc = [], (f32(min_score) if loop else f32(0))
    for si, k in sm:
        best, acc, bk = si, si, k
        for k2 in covis[k]:
            if loop:
                if not (st["query"][k2] == qid and st["words"][k2] > mn):
                    continue
            elif st["query"][k2] != qid:
                continue
            acc = f32(acc + st["score"][k2])
            if st["score"][k2] > best:
                bk, best = k2, st["score"][k2]
        am.append((acc, bk))
        if acc > best_acc:
            best_acc = acc
    keep = f32(f32(0.75) * best_acc)
    out = []
    for acc, k in am:
        if acc > keep and k not in out:
            out.append(k)
    return out


def _fresh_state(n):
    return dict(query=np.full(n, -1, np.int32), words=np.zeros(n, np.int32), score=np.zeros(n, np.float32))


def test_detect_candidates_oracle(oracle, synth):
    kd = synth.keyframe_database(n_db=160, n_words=2500, words_per_kf=150, n_places=12, seed=3)
    db, covis, n = kd["db"], kd["covis"], len(kd["db"])
    dead = np.zeros(n, np.uint8); dead[[5, 40, 41, 150]] = 1
    for loop in (0, 1):
        st_o, st_n = _fresh_state(n), _fresh_state(n)
        total = 0
        for qi, (qw, qv, pl) in enumerate(kd["queries"]):
            connected = np.zeros(n, np.uint8)
            connected[(kd["place"] == pl) & (np.arange(n) >= n // 2)] = 1            # the second lap's key frames of this place are "local"
            qid = 1000 + qi // 2                                                  # pairs of queries share an id: the members are not reset (other camera pair)
            exp = _detect_np(loop, qid, qw, qv, db, dead, covis, st_n, connected, 0.05)
            got = oracle.detect_candidates(loop, qid, qw, qv, db, dead, covis, st_o, connected, 0.05)
            assert got == exp
            assert all(np.array_equal(st_o[k], st_n[k]) for k in st_o)
            assert not any(dead[k] for k in got) and (not loop or not any(connected[k] for k in got))
            if got and qi % 2 == 0:                                               # (a fresh id; the second query of a pair inherits marks)
                assert (kd["place"][got] == pl).mean() > 0.5                      # the candidates are revisits of the query's place
            total += len(got)
        assert total > 10
    # a query without any shared word, an empty database
    assert oracle.detect_candidates(0, 1, np.array([2499], np.int32) + 1, np.array([1.0]), db, dead, covis, _fresh_state(n)) == []
    assert oracle.detect_candidates(0, 1, kd["queries"][0][0], kd["queries"][0][1], [], np.zeros(0, np.uint8), [], _fresh_state(0)) == []
