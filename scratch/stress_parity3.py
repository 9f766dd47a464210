"""Randomised GPU-vs-oracle parity sweep, part 3 (round 2): window searches with independent queries, SearchForInitialization, the
key-frame database query + candidate selection, the BA batch against single solves. usage: stress_parity3.py [seconds] [seed]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle(); synth = pkg.synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cnt = dict(window=0, init=0, kfdb=0); bad = 0
def report(what, *a):
    global bad; bad += 1; print("MISMATCH", what, *a, flush=True)
inv_sigma2 = (1.0 / (np.float32(1.2) ** (2 * np.arange(8)))).astype(np.float32)
while time.time() < t_end:
    seed = int(rng.integers(1 << 30))
    # ---- window searches
    npc, nq = int(rng.integers(30, 2500)), int(rng.integers(1, 2000))
    fr, q = synth.projection_problem(n_per_cam=npc, n_queries=nq, seed=seed, th=float(rng.choice([1.0, 3.0, 8.0])), big_windows=int(rng.integers(0, min(nq, 20))))
    fr["grid_off"], fr["grid_idx"] = pkg.frame_grid(fr["cam_off"], fr["kp_x"], fr["kp_y"], fr["min_x"], fr["min_y"], fr["grid_w_inv"], fr["grid_h_inv"])
    q["max_level"] = (q["min_level"] + int(rng.integers(0, 3))).astype(np.int32)
    if rng.random() < 0.5: fr["taken"] = np.zeros_like(fr["taken"])
    kf, chi, th = bool(rng.integers(0, 2)), (inv_sigma2 if rng.random() < 0.5 else None), int(rng.integers(20, 120))
    m = pkg.ORBmatcher(0.8, True)
    mq, bd, n = m.SearchInWindow(fr, q, th=th, kf_area=kf, chi2_inv_sigma2=chi)
    emq, ebd, en = oracle.search_in_window(fr, q, th, kf, chi)
    if not (np.array_equal(mq, emq) and np.array_equal(bd, ebd) and n == en): report("window", seed, npc, nq, kf, chi is not None, th)
    cnt["window"] += 1
    # ---- SearchForInitialization
    npc = int(rng.integers(30, 2000))
    f2, qi = synth.initialization_problem(n_per_cam=npc, seed=seed, window=float(rng.choice([30.0, 100.0, 100.0, 300.0])), crowd=float(rng.uniform(0, 0.7)))
    f2["grid_off"], f2["grid_idx"] = pkg.frame_grid(f2["cam_off"], f2["kp_x"], f2["kp_y"], f2["min_x"], f2["min_y"], f2["grid_w_inv"], f2["grid_h_inv"])
    ratio, ori = float(rng.choice([0.6, 0.75, 0.9, 1.0])), bool(rng.integers(0, 2))
    m12, n = pkg.ORBmatcher(ratio, ori).SearchForInitialization(f2, qi)
    e12, en = oracle.search_for_initialization(f2, qi, ratio, ori)
    if not (np.array_equal(m12, e12) and n == en): report("init", seed, npc, ratio, ori)
    cnt["init"] += 1
    # ---- key-frame database
    n_db = int(rng.integers(1, 400)); nw = int(rng.integers(50, 3000)); wpk = int(rng.integers(5, min(nw, 300)))
    kd = synth.keyframe_database(n_db=n_db, n_words=nw, words_per_kf=wpk, n_places=int(rng.integers(1, 30)), seed=seed)
    db, covis = kd["db"], kd["covis"]
    dead = (rng.random(n_db) < 0.05).astype(np.uint8)
    loop = int(rng.integers(0, 2))
    kfdb = pkg.KeyFrameDatabase()
    for w, v in db: kfdb.add(w, v)
    for k in np.nonzero(dead)[0]: kfdb.erase(int(k))
    st = dict(query=np.full(n_db, -1, np.int32), words=np.zeros(n_db, np.int32), score=np.zeros(n_db, np.float32))
    for j, (qw, qv, pl) in enumerate(kd["queries"][:6]):
        conn = ((kd["place"] == pl) & (rng.random(n_db) < 0.5)).astype(np.uint8)
        qid = 7 + j // 2
        ms = float(rng.choice([0.0, 0.02, 0.1]))
        got = kfdb.DetectLoopCandidates(qid, qw, qv, covis, conn, ms) if loop else kfdb.DetectRelocalizationCandidates(qid, qw, qv, covis)
        exp = oracle.detect_candidates(loop, qid, qw, qv, db, dead, covis, st, conn, ms)
        if got != exp: report("kfdb", seed, n_db, nw, wpk, loop, j)
    kfdb.close()
    cnt["kfdb"] += 1
print("configs", cnt, "mismatches", bad)
