"""Known-answer tests for the local-BA oracle (Optimizer.cc:407-696 + vendored g2o arithmetic)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _cams(oracle, synth, exact=False):
    T0, T1 = synth.rig_extrinsics_f32()
    out = []
    for c, T in enumerate((T0, T1)):
        k = synth.RIG["cam%d" % c]
        adj, ext = synth.rig_adjoint_f32(T, exact)
        out.append(oracle.make_camera(np.float32(k["fx"]), np.float32(k["fy"]), np.float32(k["cx"]), np.float32(k["cy"]), ext, adj))
    return out


def test_rig_adjoint_helper(oracle, synth):
    T0, T1 = synth.rig_extrinsics_f32()
    for exact in (False, True):
        adj_py, ext_py = synth.rig_adjoint_f32(T1, exact)
        adj_c, ext_c = oracle.rig_adjoint(T1, exact)
        assert np.allclose(adj_py, adj_c, atol=1e-12) and np.allclose(ext_py, ext_c, atol=1e-12)
    adj, ext = oracle.rig_adjoint(T0, False)
    assert np.array_equal(adj, np.eye(6)) and ext.tolist() == [0, 0, 0, 0, 0, 0, 1]
    adj, _ = oracle.rig_adjoint(T1, False)
    assert np.all(adj[3:, :3] == 0)            # Q1: lower-left block (uninitialised in the reference) = 0
    assert np.abs(adj[:3, 3:]).max() > 0.05    # UR = R * t^


def test_edge_error_and_jacobian(oracle, synth):
    cams = _cams(oracle, synth, exact=True)
    cams_ref = _cams(oracle, synth, exact=False)
    rng = np.random.default_rng(0)
    pose = np.array([0.1, -0.2, 0.3, 0.05, -0.02, 0.03, 0.0])
    pose[6] = np.sqrt(1 - np.sum(pose[3:6] ** 2))
    for c in (0, 1):
        X = np.array([0.4, -0.3, 5.0]) if c == 0 else np.array([-4.0, 0.2, 3.5])
        e0, z = oracle.ba_edge_error(pose, X, cams[c], np.zeros(2))
        assert z > 0.5
        # error = obs - projection: a perfect observation gives 0
        e, _ = oracle.ba_edge_error(pose, X, cams[c], -e0)
        assert np.allclose(e, 0, atol=1e-12)
        Jp, Jx = oracle.ba_edge_jacobian(pose, X, cams[c])
        h = 1e-6
        Jp_fd, Jx_fd = np.zeros((2, 6)), np.zeros((2, 3))
        for k in range(6):
            d = np.zeros(6); d[k] = h
            ep, _ = oracle.ba_edge_error(oracle.se3_oplus(pose, d), X, cams[c], np.zeros(2))
            em, _ = oracle.ba_edge_error(oracle.se3_oplus(pose, -d), X, cams[c], np.zeros(2))
            Jp_fd[:, k] = (ep - em) / (2 * h)
        for k in range(3):
            d = np.zeros(3); d[k] = h
            ep, _ = oracle.ba_edge_error(pose, X + d, cams[c], np.zeros(2))
            em, _ = oracle.ba_edge_error(pose, X - d, cams[c], np.zeros(2))
            Jx_fd[:, k] = (ep - em) / (2 * h)
        assert np.allclose(Jx, Jx_fd, rtol=0, atol=2e-5 * np.abs(Jx_fd).max())
        # with g2o's true adjoint the pose Jacobian IS the derivative (checked to ~1e-7)
        assert np.allclose(Jp, Jp_fd, rtol=0, atol=2e-5 * np.abs(Jp_fd).max())
        # with the reference's matrix (UR = R t^, LL = 0) it is exact for cam0 and an approximation for cam1 (Q2)
        Jp_ref, Jx_ref = oracle.ba_edge_jacobian(pose, X, cams_ref[c])
        assert np.allclose(Jx_ref, Jx)
        dev = np.abs(Jp_ref - Jp_fd).max() / np.abs(Jp_fd).max()
        assert dev < 1e-5 if c == 0 else 1e-3 < dev < 0.2


def test_se3_oplus(oracle):
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    assert np.allclose(oracle.se3_oplus(ident, np.zeros(6)), ident)
    out = oracle.se3_oplus(ident, np.array([0, 0, 0, 1.0, -2.0, 3.0]))       # pure translation (theta < 1e-5 branch)
    assert np.allclose(out, [1, -2, 3, 0, 0, 0, 1])
    th = 0.3
    out = oracle.se3_oplus(ident, np.array([0, 0, th, 0, 0, 0]))              # rotation about z
    assert np.allclose(out[3:], [0, 0, np.sin(th / 2), np.cos(th / 2)], atol=1e-14)
    # left-multiplicative: exp(d) * T
    T = np.array([1.0, 2.0, 3.0, 0, 0, 0, 1.0])
    out = oracle.se3_oplus(T, np.array([0, 0, np.pi / 2, 0, 0, 0]))
    assert np.allclose(out[:3], [-2, 1, 3], atol=1e-12)
    # small-angle branch uses R = I + W + W^2 (se3quat.h:238-244), still ~ a rotation
    out = oracle.se3_oplus(ident, np.array([1e-6, 0, 0, 0, 0, 0]))
    assert abs(out[3] - 5e-7) < 1e-12 and abs(np.linalg.norm(out[3:]) - 1) < 1e-15


def _run(oracle, pb, **kw):
    prob = dict(pb); prob.update(kw)
    prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
    return oracle.ba_local(prob)


def test_ba_noise_free_converges_to_ground_truth(oracle, synth):
    """noise-free problem must converge to ground truth within 1e-4."""
    pb = synth.ba_problem(n_poses=10, n_fixed=3, n_points=120, obs_per_point=5, seed=3, noise=False)
    rng = np.random.default_rng(1)
    start = dict(pb)
    start["points"] = pb["points"] + rng.normal(0, 0.03, pb["points"].shape)
    poses = pb["poses"].copy()
    for i in range(len(poses)):
        if not pb["pose_fixed"][i]:
            poses[i, :3] += rng.normal(0, 0.02, 3)
    start["poses"] = poses
    r = _run(oracle, start)
    free = pb["pose_fixed"] == 0
    assert np.abs(r["poses"][free, :3] - pb["gt_poses"][free, :3]).max() < 1e-4
    assert np.abs(r["points"] - pb["gt_points"]).max() < 2e-3
    assert r["edge_outlier"].sum() == 0 and r["edge_level1"].sum() == 0
    assert np.array_equal(r["poses"][~free], pb["poses"][~free])           # fixed poses untouched
    tr = r["chi2_trace"][:sum(r["n_iters"])]
    assert tr[-1] < 1e-3 * tr[0] + 1e-3


def test_ba_small_golden_and_outliers(oracle, synth):
    g = np.load(os.path.join(GOLDEN, "ba_small.npz"))
    pb = synth.ba_problem(n_poses=12, n_fixed=3, n_points=150, obs_per_point=6, seed=7)
    assert np.array_equal(pb["poses"], g["in_poses"]) and np.array_equal(pb["obs"], g["obs"])
    r = _run(oracle, pb)
    assert np.allclose(r["poses"], g["poses"], atol=1e-9) and np.allclose(r["points"], g["points"], atol=1e-9)
    assert np.array_equal(r["edge_outlier"], g["edge_outlier"]) and np.array_equal(r["edge_level1"], g["edge_level1"])
    assert r["n_iters"] == g["n_iters"].tolist() and r["n_trials"] == g["n_trials"].tolist()
    # 5 % gross outliers (+-50 px) are excluded from round 2, plus the ~5 % of inliers above the
    # chi2 95 % quantile (5.991)
    n_bad = int(round(0.05 * len(pb["obs"])))
    assert n_bad * 0.9 <= r["edge_level1"].sum() <= n_bad * 2.4
    free = pb["pose_fixed"] == 0
    err0 = np.abs(pb["poses"][free, :3] - pb["gt_poses"][free, :3]).max()
    err1 = np.abs(r["poses"][free, :3] - pb["gt_poses"][free, :3]).max()
    assert err1 < 0.25 * err0
    # robust chi2 decreases monotonically inside each round (accepted steps only)
    k = r["n_iters"][0]
    assert np.all(np.diff(r["chi2_trace"][:k]) <= 1e-9) and np.all(np.diff(r["chi2_trace"][k:k + r["n_iters"][1]]) <= 1e-9)


def test_ba_stop_flag_and_degenerate(oracle, synth):
    pb = synth.ba_problem(n_poses=8, n_fixed=2, n_points=60, obs_per_point=4, seed=5)
    stop = np.ones(1, np.uint8)
    prob = dict(pb)
    prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
    r = oracle.ba_local(prob, stop_flag=stop)                      # Optimizer.cc:582-584: early return
    assert r["n_iters"] == [0, 0] and np.array_equal(r["poses"], pb["poses"]) and np.array_equal(r["points"], pb["points"])
    # all poses fixed: only points move (structure-only), still fine
    allfix = dict(pb); allfix["pose_fixed"] = np.ones(8, np.uint8)
    r = _run(oracle, allfix)
    assert np.array_equal(r["poses"], pb["poses"]) and r["n_iters"][0] >= 1
    assert not np.array_equal(r["points"], pb["points"])


def test_pose_optimization_oracle(oracle, synth):
    """Optimizer::PoseOptimization restatement: converges to the ground-truth pose from the motion-model guess, flags the
    gross outliers, handles the degenerate frames (< 3 edges: untouched, returns 0; < 10 edges: one round)."""
    pb = synth.pose_problem(n_frames=6, obs_per_frame=300, seed=3)
    prob = dict(pb)
    prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
    r = oracle.pose_optimization(prob)
    off = pb["edge_off"]
    # frame 0: 2 correspondences -> untouched
    assert r["n_inliers"][0] == 0 and np.array_equal(r["poses"][0], pb["poses"][0]) and not r["outlier"][off[0]:off[1]].any()
    # frame 1: 9 edges -> exactly one round of optimisation
    assert r["n_iters"][1][0] > 0 and (r["n_iters"][1][1:] == 0).all()
    for f in range(2, 6):
        err0 = np.abs(pb["poses"][f, :3] - pb["gt_poses"][f, :3]).max()
        err1 = np.abs(r["poses"][f, :3] - pb["gt_poses"][f, :3]).max()
        assert err1 < 0.02 and err1 < 0.5 * err0, (f, err0, err1)
        n = off[f + 1] - off[f]
        n_out = int(r["outlier"][off[f]:off[f + 1]].sum())
        assert r["n_inliers"][f] == n - n_out
        assert 0.05 * n < n_out < 0.35 * n                     # ~10 % gross outliers + the chi2 tail
        assert (r["n_iters"][f] > 0).all()
    # classification is the chi2 gate on the reported chi2 (float compare, Optimizer.cc:375-377)
    e = slice(off[2], off[6])
    assert np.array_equal(r["outlier"][e] != 0, r["edge_chi2"][e].astype(np.float32) > np.float32(5.991))


def test_linearize_blocks_are_the_sums_of_the_edge_jacobians(oracle, synth):
    """orc_ba_linearize (the checker of the GPU's first-linearisation tap, tests/test_gpu_ba.py) against an independent numpy
    recomposition: every block of H and b = sum over the vertex's edges of J^T (rho' Omega) J and -J^T (rho' Omega) e with the Jacobians
    of orc_ba_edge_jacobian (themselves held against central differences above) and g2o's Huber rho' (pinned to its own statements in
    test_oracle_ref.py); fixed poses own no block, their edges no H_pl."""
    for exact in (False, True):
        pb = synth.ba_problem(n_poses=8, n_fixed=2, n_points=50, obs_per_point=4, seed=31, exact_adjoint=exact)
        prob = dict(pb)
        prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
        lin = oracle.ba_linearize(prob)
        P, L, E = len(pb["poses"]), len(pb["points"]), len(pb["obs"])
        free = np.nonzero(pb["pose_fixed"] == 0)[0]
        assert lin["n_free"] == len(free) and np.array_equal(np.nonzero(lin["pose_idx"] >= 0)[0], free)
        Hpp, bp, Hll, bl, Hpl = np.zeros((P, 6, 6)), np.zeros((P, 6)), np.zeros((L, 3, 3)), np.zeros((L, 3)), np.zeros((E, 6, 3))
        delta = float(pb["huber_delta"])                           # the float-rounded sqrt(5.991) of Optimizer.cc:515
        n_sat = 0
        for e in range(E):
            p, l, cam = int(pb["edge_pose"][e]), int(pb["edge_point"][e]), prob["cams"][int(pb["edge_cam"][e])]
            Jp, Jx = oracle.ba_edge_jacobian(pb["poses"][p], pb["points"][l], cam)
            err, _z = oracle.ba_edge_error(pb["poses"][p], pb["points"][l], cam, pb["obs"][e])
            w = float(pb["inv_sigma2"][e])
            chi2 = w * float(err @ err)
            rho1 = 1.0 if chi2 <= delta * delta else delta / np.sqrt(chi2)
            n_sat += chi2 > delta * delta
            Hll[l] += rho1 * w * (Jx.T @ Jx); bl[l] += -rho1 * w * (Jx.T @ err)
            if not pb["pose_fixed"][p]:
                Hpp[p] += rho1 * w * (Jp.T @ Jp); bp[p] += -rho1 * w * (Jp.T @ err); Hpl[e] = rho1 * w * (Jp.T @ Jx)
        assert 0 < n_sat < E                                      # both Huber branches
        tol = dict(rtol=1e-10, atol=1e-8)
        assert np.allclose(lin["Hpp"], Hpp[free], **tol) and np.allclose(lin["bp"], bp[free], **tol)
        assert np.allclose(lin["Hll"], Hll, **tol) and np.allclose(lin["bl"], bl, **tol) and np.allclose(lin["Hpl"], Hpl, **tol)
        assert not np.any(lin["Hpl"][pb["pose_fixed"][pb["edge_pose"]] != 0])


def test_weakly_constrained_draw_is_chaotic_in_the_oracle_itself(oracle, synth):
    """The one hard mismatch of round 6's random sweeps (profiles/r06_parity_sweeps.txt): 29 poses, every point seen 4 times, 20 % gross outliers --
    GPU and oracle run the same LM iterations and trials and end 9.5e-4 apart in a pose translation, where north_star's bar is 1e-4. The bar has no
    meaning on this draw: the oracle ALONE moves by more than 1e-4 when its observations change in the sixteenth digit (below the rounding of one
    projection). What such problems are held to instead: tests/test_gpu_ba.py::test_ba_ill_conditioned_documented_bound."""
    pb = synth.ba_problem(n_poses=29, n_fixed=5, n_points=384, obs_per_point=4, seed=9455, outlier_frac=0.2, exact_adjoint=True)

    def run(q):
        prob = dict(q)
        prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in q["cams"]]
        return oracle.ba_local(prob)
    base = run(pb)
    rng = np.random.default_rng(1)
    nudged = dict(pb)
    nudged["obs"] = pb["obs"] * (1 + 1e-15 * rng.standard_normal(pb["obs"].shape))
    other = run(nudged)
    assert other["n_iters"] == base["n_iters"] and other["n_trials"] == base["n_trials"]
    assert np.abs(other["poses"][:, :3] - base["poses"][:, :3]).max() > 1e-4
