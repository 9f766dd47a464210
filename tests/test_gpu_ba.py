"""GPU parity: dual-camera local BA through the C ABI vs the oracle. Tolerance (north_star): pose translations
within 1e-4 (m); the GPU sums in a different order and factors the reduced camera system with a blocked dense
LDL^T on the matrix cores, so equality is to rounding, not bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL_T = 1e-4          # BASELINE.json north_star: "within 1e-4 on BA pose translations"


def _oracle_run(oracle, pb, **kw):
    prob = dict(pb); prob.update(kw)
    prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
    return oracle.ba_local(prob)


def _compare(got, exp, pb):
    assert np.abs(got["poses"][:, :3] - exp["poses"][:, :3]).max() < TOL_T
    assert np.abs(got["poses"][:, 3:] - exp["poses"][:, 3:]).max() < 1e-5
    assert np.abs(got["points"] - exp["points"]).max() < 1e-3
    assert got["n_iters"] == exp["n_iters"] and got["n_trials"] == exp["n_trials"]
    k = sum(exp["n_iters"])
    assert np.allclose(got["chi2_trace"][:k], exp["chi2_trace"][:k], rtol=1e-6)
    # borderline chi2 == 5.991 edges could flip with rounding; allow a handful
    assert np.sum(got["edge_level1"] != exp["edge_level1"]) <= max(2, len(exp["edge_level1"]) // 2000)
    assert np.sum(got["edge_outlier"] != exp["edge_outlier"]) <= max(2, len(exp["edge_outlier"]) // 2000)
    fixed = pb["pose_fixed"] != 0
    assert np.array_equal(got["poses"][fixed], pb["poses"][fixed])


@pytest.mark.parametrize("kw", [dict(n_poses=12, n_fixed=3, n_points=150, obs_per_point=6, seed=7),
                                dict(n_poses=7, n_fixed=2, n_points=60, obs_per_point=4, seed=2),
                                dict(n_poses=20, n_fixed=4, n_points=500, obs_per_point=8, seed=9, exact_adjoint=True)])
def test_ba_small_vs_oracle(pkg, oracle, synth, kw):
    pb = synth.ba_problem(**kw)
    _compare(pkg.Optimizer.LocalBundleAdjustment(pb), _oracle_run(oracle, pb), pb)


def _restrict(pb, keep):
    """the problem with only the edges keep[e] (same poses and points)"""
    q = dict(pb)
    for k in ("edge_pose", "edge_point", "edge_cam", "obs", "inv_sigma2"):
        q[k] = np.ascontiguousarray(np.asarray(pb[k])[keep])
    return q


def test_ba_sparse_covisibility_vs_oracle(pkg, oracle, synth):
    """Pose pairs WITHOUT a common point (their block of the reduced camera system stays zero: the pair lists are built on the
    device from per-pose bit rows, and an empty list must leave the block alone) and points seen by fixed poses only: two clusters
    of key frames that share no map point, then a chain in which a key frame only shares points with its neighbours."""
    pb = synth.ba_problem(n_poses=24, n_fixed=4, n_points=700, obs_per_point=8, seed=12)
    ep, el = np.asarray(pb["edge_pose"]), np.asarray(pb["edge_point"])
    P = len(pb["poses"])
    clusters = _restrict(pb, (ep < P // 2) == (el % 2 == 0))
    chain = _restrict(pb, np.abs((ep * 7) % P - (el % P)) <= 2)
    for q in (clusters, chain):
        assert len(q["edge_pose"]) > 200
        _compare(pkg.Optimizer.LocalBundleAdjustment(q), _oracle_run(oracle, q), q)


def test_ba_c4_vs_oracle_and_golden(pkg, oracle, synth):
    pb = synth.ba_problem()                     # 50 KF / 2000 MP / 20000 edges
    got = pkg.Optimizer.LocalBundleAdjustment(pb)
    exp = _oracle_run(oracle, pb)
    _compare(got, exp, pb)
    g = np.load(os.path.join(GOLDEN, "ba_c4.npz"))
    assert np.abs(got["poses"][:, :3] - g["poses"][:, :3]).max() < TOL_T
    assert got["n_iters"] == g["n_iters"].tolist()
    free = pb["pose_fixed"] == 0
    assert np.abs(got["poses"][free, :3] - pb["gt_poses"][free, :3]).max() < 0.02   # recovers from 5 cm / 0.02 rad noise


def test_ba_batch_c5_eight_streams(pkg, oracle, synth):
    """BASELINE config C5: 8 dual-camera streams, one LocalBundleAdjustment each (src/LocalMapping.cc:97-104), solved by ONE
    dcs_ba_local_batch call with the LM control flow on the device. Every problem must equal its own dcs_ba_local solve
    bit for bit (same kernels, same order of operations) and the oracle within the north star's 1e-4."""
    pbs = [synth.ba_problem(seed=42 + s) for s in range(8)]             # 8 x (50 KF / 2000 MP / 20000 edges), different maps
    got = pkg.Optimizer.LocalBundleAdjustmentBatch(pbs)
    assert len(got) == 8
    for s, pb in enumerate(pbs):
        one = pkg.Optimizer.LocalBundleAdjustment(pb)
        for k in ("poses", "points", "edge_chi2", "edge_outlier", "edge_level1", "chi2_trace"):
            assert np.array_equal(got[s][k], one[k]), (s, k)
        assert got[s]["n_iters"] == one["n_iters"] and got[s]["n_trials"] == one["n_trials"] and got[s]["lambda_"] == one["lambda_"]
        if s < 3:                                                       # the oracle needs ~2 s per C4 problem
            _compare(got[s], _oracle_run(oracle, pb), pb)


def test_ba_batch_ragged_and_stop_flags(pkg, oracle, synth):
    """Problems of different sizes and LDL^T paths in one batch (n <= 256 on one workgroup, n > 256 blocked), a problem
    stopped at entry (Optimizer.cc:582-585: untouched) and an empty batch."""
    kws = [dict(n_poses=12, n_fixed=3, n_points=150, obs_per_point=6, seed=7),
           dict(n_poses=60, n_fixed=4, n_points=800, obs_per_point=8, seed=13),      # n = 330: blocked fallback
           dict(n_poses=7, n_fixed=2, n_points=60, obs_per_point=4, seed=2),
           dict(n_poses=20, n_fixed=4, n_points=500, obs_per_point=8, seed=9, exact_adjoint=True),
           dict(n_poses=8, n_fixed=2, n_points=60, obs_per_point=4, seed=5)]
    pbs = [synth.ba_problem(**kw) for kw in kws]
    pbs[4] = dict(pbs[4]); pbs[4]["pose_fixed"] = np.ones(8, np.uint8)              # every pose fixed: structure-only
    stop = [None, None, np.ones(1, np.uint8), np.zeros(1, np.uint8), None]
    got = pkg.Optimizer.LocalBundleAdjustmentBatch(pbs, stop)
    for i, pb in enumerate(pbs):
        if i == 2:
            assert got[i]["n_iters"] == [0, 0] and np.array_equal(got[i]["poses"], pb["poses"]) and np.array_equal(got[i]["points"], pb["points"])
            assert not got[i]["edge_outlier"].any()
            continue
        _compare(got[i], _oracle_run(oracle, pb), pb)
    assert pkg.Optimizer.LocalBundleAdjustmentBatch([]) == []
    dup = dict(pbs[0]); dup["edge_pose"] = pbs[0]["edge_pose"].copy(); dup["edge_point"] = pbs[0]["edge_point"].copy()
    dup["edge_pose"][1], dup["edge_point"][1] = dup["edge_pose"][0], dup["edge_point"][0]
    with pytest.raises(pkg.DcsError):
        pkg.Optimizer.LocalBundleAdjustment(dup)


@pytest.mark.parametrize("ba_cus", [0, 32], ids=["time-sliced", "solver on 32 CUs, front end on 224"])
def test_ba_batch_concurrent_with_extraction(pkg, oracle, synth, ba_cus):
    """The reference's threading: Tracking extracts while LocalMapping runs LocalBundleAdjustment (src/LocalMapping.cc:97-104).
    A host thread solves a BA batch on the solver's own HIP stream while this thread runs dcs_orb_extract_batch_device on
    another stream; both results must equal their solo runs bit for bit (and the oracle). Second variant: the chip partitioned with CU
    masks (dcs_ba_set_cu_range / dcs_stream_create_cu_range) -- where the kernels run must not change a bit of what they compute."""
    import threading
    import torch
    pbs = [synth.ba_problem(n_poses=30, n_fixed=5, n_points=900, obs_per_point=8, seed=60 + s) for s in range(4)]
    solo = pkg.Optimizer.LocalBundleAdjustmentBatch(pbs)
    pkg.abi.ba_release_thread()
    pkg.abi.ba_set_cu_range(0, ba_cus)
    raw_stream = pkg.abi.cu_range_stream(ba_cus, 256 - ba_cus) if ba_cus else None
    imgs = [im for f in range(4) for im in synth.frame_pair(640, 480, 1, f)]
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=len(imgs))
    cap, B = e.default_cap(), len(imgs)
    d_img = torch.from_numpy(np.stack(imgs)).cuda()
    bufs = [(torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"), torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"),
             torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(2)]
    st = torch.cuda.ExternalStream(raw_stream) if ba_cus else torch.cuda.Stream()
    e.extract_batch_device(d_img, *bufs[0], cap, stream=st.cuda_stream)
    torch.cuda.synchronize()
    got, err = [], []

    def ba_thread():
        try:
            for _ in range(6):
                got.append(pkg.Optimizer.LocalBundleAdjustmentBatch(pbs))
        except Exception as ex:            # noqa: BLE001
            err.append(ex)
        finally:
            pkg.abi.ba_release_thread()
    th = threading.Thread(target=ba_thread)
    th.start()
    n_ext = 0
    while th.is_alive() or n_ext < 3:
        e.extract_batch_device(d_img, *bufs[1], cap, stream=st.cuda_stream)
        st.synchronize()
        n_ext += 1
        assert torch.equal(bufs[1][2], bufs[0][2]) and torch.equal(bufs[1][1], bufs[0][1])
        assert torch.equal(bufs[1][0].view(torch.int32), bufs[0][0].view(torch.int32))
    th.join()
    assert not err, err
    assert len(got) == 6 and n_ext >= 3
    for r in got:
        for s in range(4):
            for k in ("poses", "points", "edge_outlier", "chi2_trace"):
                assert np.array_equal(r[s][k], solo[s][k]), (s, k)
    _compare(solo[0], _oracle_run(oracle, pbs[0]), pbs[0])
    okp, odesc = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(imgs[3])
    n3 = int(bufs[1][2][3])
    assert n3 == len(okp) and np.array_equal(bufs[1][1][3, :n3].cpu().numpy(), odesc)
    e.close()
    pkg.abi.ba_set_cu_range(0, 0)
    if raw_stream:
        torch.cuda.synchronize()
        pkg.abi.lib().dcs_stream_destroy(raw_stream)
    with pytest.raises(pkg.DcsError):
        pkg.abi.cu_range_stream(250, 16)                      # beyond the chip's 256 CUs


def test_ba_ill_conditioned_documented_bound(pkg, oracle, synth):
    """A weakly constrained problem far from the C4 shape (31 poses, every point seen by only 3 of them, 20 % gross outliers; the
    worst case of the round-1 random sweep, gpurun_out/stress2.log:23808). The chi2 traces of two IEEE-correct solvers with
    different summation orders agree to 1e-15 at the start and separate ~10x per LM iteration; on the convergence plateau
    rho ~ 0 changes sign with rounding, so one side may run one LM trial more. Documented bound for such problems:
    translations within 2e-2, the first round's chi2 trace to 1e-6, trial counts within one. C4-shaped problems stay at 1e-9."""
    pb = synth.ba_problem(n_poses=31, n_fixed=1, n_points=326, obs_per_point=3, seed=8141, outlier_frac=0.2, exact_adjoint=True)
    got, exp = pkg.Optimizer.LocalBundleAdjustment(pb), _oracle_run(oracle, pb)
    assert got["n_iters"][0] == exp["n_iters"][0] and abs(got["n_iters"][1] - exp["n_iters"][1]) <= 1
    assert abs(sum(got["n_trials"]) - sum(exp["n_trials"])) <= 2
    k = exp["n_iters"][0]
    assert np.allclose(got["chi2_trace"][:k], exp["chi2_trace"][:k], rtol=1e-6)
    assert np.abs(got["poses"][:, :3] - exp["poses"][:, :3]).max() < 2e-2
    assert np.sum(got["edge_level1"] != exp["edge_level1"]) <= 2
    assert np.isfinite(got["poses"]).all() and np.isfinite(got["points"]).all()


def test_ba_noise_free_ground_truth(pkg, synth):
    pb = synth.ba_problem(n_poses=10, n_fixed=3, n_points=120, obs_per_point=5, seed=3, noise=False)
    rng = np.random.default_rng(1)
    start = dict(pb)
    start["points"] = pb["points"] + rng.normal(0, 0.03, pb["points"].shape)
    poses = pb["poses"].copy()
    free = pb["pose_fixed"] == 0
    poses[free, :3] += rng.normal(0, 0.02, (free.sum(), 3))
    start["poses"] = poses
    r = pkg.Optimizer.LocalBundleAdjustment(start)
    assert np.abs(r["poses"][free, :3] - pb["gt_poses"][free, :3]).max() < TOL_T
    assert r["edge_outlier"].sum() == 0


def test_ba_stop_flag_all_fixed_and_bad_input(pkg, oracle, synth):
    pb = synth.ba_problem(n_poses=8, n_fixed=2, n_points=60, obs_per_point=4, seed=5)
    r = pkg.Optimizer.LocalBundleAdjustment(pb, stop_flag=np.ones(1, np.uint8))
    assert r["n_iters"] == [0, 0] and np.array_equal(r["poses"], pb["poses"]) and np.array_equal(r["points"], pb["points"])
    allfix = dict(pb); allfix["pose_fixed"] = np.ones(8, np.uint8)
    got, exp = pkg.Optimizer.LocalBundleAdjustment(allfix), _oracle_run(oracle, allfix)
    assert np.array_equal(got["poses"], pb["poses"]) and np.abs(got["points"] - exp["points"]).max() < 1e-6
    # an edge out of range (pose, point, camera; first and last edge): refused by the list build's first pass, before anything is indexed with it
    for key, at, val in (("edge_pose", 0, 99), ("edge_pose", -1, -1), ("edge_point", 3, 60), ("edge_point", -1, -7), ("edge_cam", 5, 2), ("edge_cam", 0, -1)):
        bad = dict(pb); bad[key] = pb[key].copy(); bad[key][at] = val
        with pytest.raises(pkg.DcsError, match="out of range"):
            pkg.Optimizer.LocalBundleAdjustment(bad)
    with pytest.raises(pkg.DcsError, match="out of range"):                      # inside a batch: the error names the problem
        bad = dict(pb); bad["edge_point"] = pb["edge_point"].copy(); bad["edge_point"][7] = 1 << 20
        pkg.Optimizer.LocalBundleAdjustmentBatch([pb, bad, pb])


def test_search_by_bow_greedy_vs_oracle(pkg, oracle, synth):
    img0, img1 = synth.frame_pair(640, 480, 0, 0)
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    kp0, d0 = o.extract(img0)
    kp1, d1 = o.extract(img1)
    rng = np.random.default_rng(3)
    for n_buckets, ratio, ori in [(100, 0.75, True), (12, 0.9, True), (300, 0.6, False), (1, 0.75, True)]:
        fv_kf = synth.csr_buckets(len(d0), n_buckets, seed=n_buckets)
        fv_f = synth.csr_buckets(len(d1), n_buckets, seed=n_buckets + 1)
        valid = (rng.random(len(d0)) < 0.8).astype(np.uint8)
        exp_m, exp_n = oracle.search_by_bow_crosscam(d0, kp0["angle"], valid, d1, kp1["angle"], fv_kf, fv_f, ratio, ori)
        got_m, got_n = pkg.ORBmatcher(ratio, ori).SearchByBoWCrossCam(d0, kp0["angle"], valid, d1, kp1["angle"], fv_kf, fv_f)
        assert np.array_equal(got_m, exp_m) and got_n == exp_n, (n_buckets, ratio, ori)
    # contested candidates: identical queries in one node must resolve in order (ORBmatcher.cc:216)
    base = synth.random_descriptors(40, seed=8)
    kf = np.repeat(base, 3, axis=0)
    f = np.concatenate([base, synth.noisy_copy(base, 8, seed=1), synth.noisy_copy(base, 16, seed=2)])
    ang = np.zeros(len(kf), np.float32)
    fv = synth.csr_buckets(len(kf), 5, seed=4)
    exp_m, exp_n = oracle.search_by_bow_crosscam(kf, ang, np.ones(len(kf), np.uint8), f, ang, fv, fv, 0.95, False)
    got_m, got_n = pkg.ORBmatcher(0.95, False).SearchByBoWCrossCam(kf, ang, np.ones(len(kf), np.uint8), f, ang, fv, fv)
    assert np.array_equal(got_m, exp_m) and got_n == exp_n and exp_n > 40


def test_ba_blocked_mfma_ldlt_fallback(pkg, oracle, synth, opts):
    """n > 256 (or the test hook) uses the multi-launch blocked LDL^T whose trailing update runs on
    v_mfma_f64_16x16x4_f64; both solvers must agree with the oracle."""
    pb = synth.ba_problem(n_poses=12, n_fixed=3, n_points=150, obs_per_point=6, seed=7)
    opts("DCS_BA_FORCE_BLOCKED_LDLT", 1)
    _compare(pkg.Optimizer.LocalBundleAdjustment(pb), _oracle_run(oracle, pb), pb)
    opts("DCS_BA_FORCE_BLOCKED_LDLT", 0)
    big = synth.ba_problem(n_poses=60, n_fixed=4, n_points=800, obs_per_point=8, seed=13)      # 55 free poses -> n = 330 > 256
    _compare(pkg.Optimizer.LocalBundleAdjustment(big), _oracle_run(oracle, big), big)


@pytest.mark.parametrize("robust,iters", [(True, 10), (False, 5)])
def test_global_bundle_adjustment_vs_oracle(pkg, oracle, synth, robust, iters):
    """Optimizer::BundleAdjustment (Optimizer.cc:70-248): one round of nIterations, Huber sqrt(3.99) or no kernel,
    only fixId fixed, no outlier re-classification between rounds."""
    pb = synth.ba_problem(n_poses=14, n_fixed=1, n_points=220, obs_per_point=6, seed=11)
    got = pkg.Optimizer.BundleAdjustment(pb, nIterations=iters, bRobust=robust)
    delta = float(np.float32(np.sqrt(3.99))) if robust else 0.0
    exp = _oracle_run(oracle, pb, iters1=iters, iters2=0, huber_delta=delta)
    assert exp["n_iters"][1] == 0 and got["n_iters"][1] == 0 and exp["n_iters"][0] >= 2
    assert not got["edge_level1"].any() or got["n_iters"][1] == 0
    _compare(got, exp, pb)
    if not robust:                              # without the kernel the trace is the plain chi2: strictly larger on outliers
        rob = _oracle_run(oracle, pb, iters1=iters, iters2=0, huber_delta=float(np.float32(np.sqrt(3.99))))
        assert exp["chi2_trace"][0] > rob["chi2_trace"][0]


@pytest.fixture(params=[0, 1], ids=["composed transform", "oracle's per-edge arithmetic"])
def exact_edge(request, pkg):
    """both builds of k_pose_opt2 (option DCS_POSE_EXACT_EDGE: 0 = the default, an edge's point through the composed world -> camera matrix;
    1 = through the oracle's own operations, NOTES R6.3)"""
    with pkg.abi.options(DCS_POSE_EXACT_EDGE=request.param):
        yield request.param


@pytest.mark.parametrize("seed,obs", [(8, 350), (21, 350), (22, 120), (23, 700), (24, 1000), (25, 1900), (26, 2048), (27, 2300), (28, 2500)])
def test_pose_optimization_vs_oracle(pkg, oracle, synth, seed, obs, exact_edge):
    """Optimizer::PoseOptimization batched on the GPU (one workgroup per frame, LM loop on the device) vs the oracle:
    same outlier flags, iteration counts and inlier counts; poses to rounding. Nine seeds; 120 .. 2 300 observations per frame = 1 .. 9
    register slots per lane of k_pose_opt2, 2 500 = k_pose_opt (the general kernel) beside it in one call."""
    pb = synth.pose_problem(n_frames=24 if obs <= 1000 else 8, obs_per_frame=obs, seed=seed)
    prob = dict(pb)
    prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
    exp = oracle.pose_optimization(prob)
    got = pkg.Optimizer.PoseOptimization(pb)
    # on the convergence plateau an LM iteration more or less is a rounding matter (rho ~ 0): same counts up to +-1 in a few rounds
    dn = np.abs(got["n_iters"] - exp["n_iters"])
    assert dn.max() <= 1 and np.count_nonzero(dn) <= max(2, dn.size // 10), (int(dn.max()), int(np.count_nonzero(dn)), dn.size)
    assert np.abs(got["poses"][:, :3] - exp["poses"][:, :3]).max() < 1e-7
    assert np.abs(got["poses"][:, 3:] - exp["poses"][:, 3:]).max() < 1e-8
    assert np.array_equal(got["poses"][0], pb["poses"][0]) and got["n_inliers"][0] == 0      # < 3 correspondences: untouched
    flips = int(np.sum(got["outlier"] != exp["outlier"]))                                  # chi2 exactly on the gate may flip
    assert flips <= 2
    assert np.abs(got["n_inliers"] - exp["n_inliers"]).max() <= flips
    assert np.allclose(got["edge_chi2"], exp["edge_chi2"], rtol=1e-6, atol=1e-9)
    # the batch is per-frame independent: a sub-batch gives the same frames
    sub = dict(pb)
    f0, f1 = 5, min(9, len(pb["edge_off"]) - 1)
    e0, e1 = pb["edge_off"][f0], pb["edge_off"][f1]
    sub.update(poses=pb["poses"][f0:f1], edge_off=pb["edge_off"][f0:f1 + 1] - e0, xw=pb["xw"][e0:e1], obs=pb["obs"][e0:e1],
               inv_sigma2=pb["inv_sigma2"][e0:e1], edge_cam=pb["edge_cam"][e0:e1])
    got2 = pkg.Optimizer.PoseOptimization(sub)
    assert np.array_equal(got2["poses"], got["poses"][f0:f1]) and np.array_equal(got2["outlier"], got["outlier"][e0:e1])


def test_ba_window_of_sixty_free_poses_vs_oracle(pkg, oracle, synth):
    """Optimizer::LocalBundleAdjustment takes EVERY covisible key frame (Optimizer.cc:415-422: the window is unbounded). 60 free poses:
    reduced camera system n = 360 > 256, beyond the one-workgroup factorisation -- both rounds, against the oracle."""
    pb = synth.ba_problem(n_poses=67, n_fixed=6, n_points=2400, obs_per_point=8, seed=61)       # + fixId, the oldest free pose
    assert int((pb["pose_fixed"] == 0).sum()) == 60
    _compare(pkg.Optimizer.LocalBundleAdjustment(pb), _oracle_run(oracle, pb), pb)


def test_global_bundle_adjustment_at_map_size_vs_oracle(pkg, oracle, synth):
    """Optimizer::BundleAdjustment over a whole map (Optimizer.cc:70-248): 200 key frames, 20 000 map points, 160 000 dual-camera edges,
    only the first pose fixed -> reduced camera system n = 1 194; one round of 4 iterations with the Huber kernel, against the oracle."""
    pb = synth.ba_problem(n_poses=200, n_fixed=1, n_points=20000, obs_per_point=8, seed=5)
    assert len(pb["edge_pose"]) >= 150000
    got = pkg.Optimizer.BundleAdjustment(pb, nIterations=4, bRobust=True)
    exp = _oracle_run(oracle, pb, iters1=4, iters2=0, huber_delta=float(np.float32(np.sqrt(3.99))))
    assert exp["n_iters"][0] >= 2
    _compare(got, exp, pb)


def test_pose_optimization_four_camera_rig_with_one_crowded_camera(pkg, oracle, synth, exact_edge):
    """k_pose_opt2 keeps a camera's edges in the registers of the waves the camera gets: a four-camera rig with more than 768 edges on ONE
    camera does not fit (each camera has one wave of 64 lanes x 12 slots), the kernel declines the frame and k_pose_opt takes it in the same call. Frames of the
    same call that do fit stay with k_pose_opt2; both against the oracle. Cameras 2, 3 are copies of camera 1 (same model, other id)."""
    pb = synth.pose_problem(n_frames=6, obs_per_frame=5000, seed=41)
    rng = np.random.default_rng(7)
    keep, cam_new = [], []
    off = [0]
    for f in range(6):
        e0, e1 = int(pb["edge_off"][f]), int(pb["edge_off"][f + 1])
        idx = np.arange(e0, e1)
        c = pb["edge_cam"][e0:e1]
        i0, i1 = idx[c == 0], idx[c == 1]
        if f in (2, 3):   i0, i1 = i0[:1100], i1[:600]          # camera 0: 1 100 edges > 64 x 12 -> declined
        elif f >= 4:      i0, i1 = i0[:500], i1[:900]           # fits: 500 / 300 / 300 / 300
        sel = np.sort(np.concatenate([i0, i1]))
        cn = pb["edge_cam"][sel].copy()
        ones = np.nonzero(cn == 1)[0]
        cn[ones] = 1 + rng.integers(0, 3, len(ones))
        keep.append(sel); cam_new.append(cn); off.append(off[-1] + len(sel))
    keep = np.concatenate(keep)
    pb4 = dict(pb, edge_off=np.asarray(off, np.int32), xw=pb["xw"][keep], obs=pb["obs"][keep], inv_sigma2=pb["inv_sigma2"][keep],
               edge_cam=np.concatenate(cam_new).astype(np.int32), cams=[pb["cams"][0], pb["cams"][1], pb["cams"][1], pb["cams"][1]])
    assert np.bincount(pb4["edge_cam"][off[2]:off[3]], minlength=4)[0] == 1100 and off[3] - off[2] <= 2048
    prob = dict(pb4)
    prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb4["cams"]]
    exp = oracle.pose_optimization(prob)
    got = pkg.Optimizer.PoseOptimization(pb4)
    assert np.abs(got["n_iters"] - exp["n_iters"]).max() <= 1
    assert np.abs(got["poses"][:, :3] - exp["poses"][:, :3]).max() < 1e-7 and np.abs(got["poses"][:, 3:] - exp["poses"][:, 3:]).max() < 1e-8
    flips = int(np.sum(got["outlier"] != exp["outlier"]))
    assert flips <= 2 and np.abs(got["n_inliers"] - exp["n_inliers"]).max() <= flips
    assert (got["n_inliers"][2:] > 100).all()


def test_pose_optimization_dual_rig_at_every_uneven_split_of_the_largest_frame(pkg, oracle, synth, exact_edge):
    """k_pose_opt2 gives the rig's two cameras 2 + 2 or 3 + 1 waves; the dual rig must fit its 12 register slots per lane at ANY split of a
    frame of up to 2 304 edges (worst cases: one third / two thirds, where the split changes, and a frame seen by one camera only). No frame may fall
    back to k_pose_opt (DCS_POSE_FAST stays on; n_iters etc. are compared with the oracle like everywhere)."""
    splits = [(1365, 683), (1366, 682), (2048, 0), (1536, 768), (0, 2304), (1152, 1152), (767, 1537)]      # (in an order the scene's frames can supply)
    pb = synth.pose_problem(n_frames=2 + len(splits), obs_per_frame=8000, seed=43)
    keep, off = [], [0]
    for f in range(2 + len(splits)):
        e0, e1 = int(pb["edge_off"][f]), int(pb["edge_off"][f + 1])
        idx = np.arange(e0, e1)
        if f >= 2:
            c = pb["edge_cam"][e0:e1]
            k0, k1 = splits[f - 2]
            i0, i1 = idx[c == 0], idx[c == 1]
            assert len(i0) >= k0 and len(i1) >= k1
            idx = np.sort(np.concatenate([i0[:k0], i1[:k1]]))
        keep.append(idx); off.append(off[-1] + len(idx))
    keep = np.concatenate(keep)
    pb2 = dict(pb, edge_off=np.asarray(off, np.int32), xw=pb["xw"][keep], obs=pb["obs"][keep], inv_sigma2=pb["inv_sigma2"][keep], edge_cam=pb["edge_cam"][keep])
    prob = dict(pb2)
    prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb2["cams"]]
    exp = oracle.pose_optimization(prob)
    got = pkg.Optimizer.PoseOptimization(pb2)
    assert np.abs(got["n_iters"] - exp["n_iters"]).max() <= 1
    assert np.abs(got["poses"][:, :3] - exp["poses"][:, :3]).max() < 1e-7 and np.abs(got["poses"][:, 3:] - exp["poses"][:, 3:]).max() < 1e-8
    flips = int(np.sum(got["outlier"] != exp["outlier"]))
    assert flips <= 2 and np.abs(got["n_inliers"] - exp["n_inliers"]).max() <= flips
    assert (got["n_inliers"][2:] > 100).all()


def test_ba_large_map_fits_the_arena(pkg, oracle, synth):
    """Global-BA-sized map (> 65 000 points: the arena estimate of round 1 ran out there): the device arena is sized by a
    dry run of the layout, so the call must succeed and agree with the oracle."""
    pb = synth.ba_problem(n_poses=10, n_fixed=2, n_points=70000, obs_per_point=2, seed=3)
    pb["iters1"], pb["iters2"] = 2, 0
    _compare(pkg.Optimizer.LocalBundleAdjustment(pb), _oracle_run(oracle, pb), pb)


@pytest.mark.parametrize("n_free", [37, 39, 40, 41, 42])
def test_ba_reduced_system_sizes_around_the_slot_builds(pkg, oracle, synth, n_free):
    """n = 6 * free poses = 222 ... 252: k_ldlt_mfma's 18-slot build (n <= 240, 15 block rows) and its 20-slot build (16 block rows)."""
    pb = synth.ba_problem(n_poses=n_free + 3, n_fixed=3, n_points=400, obs_per_point=8, seed=100 + n_free)
    _compare(pkg.Optimizer.LocalBundleAdjustment(pb), _oracle_run(oracle, pb), pb)


def test_ba_back_to_back_calls_do_not_inherit_the_previous_calls_progress(pkg, synth):
    """A finished call returns while the steps it had queued ahead are still in the solver's streams (its results come down on a stream
    of their own); those steps' k_post launches still store to the pinned progress words. The next call on the thread must drain them
    before it resets the words -- otherwise it reads 'all problems done' from its predecessor and downloads before running a step.
    Two different problems of the same size, alternated 40 times (single and as batches of 2): every result equals the first, isolated
    solve of that problem bit for bit. The race is timing-dependent: against a side build without the drain (-DDCS_BA_NO_DRAIN) this
    test failed in 2 of 3 runs on the MI355X box (through the library's own 'lost step' error both times, which nothing guarantees),
    so a pass is evidence, not proof; the ordering argument is in BaContext::drain()."""
    pa = pkg.Optimizer.prepare(synth.ba_problem(n_poses=14, n_fixed=3, n_points=220, obs_per_point=6, seed=301))
    pb = pkg.Optimizer.prepare(synth.ba_problem(n_poses=14, n_fixed=3, n_points=220, obs_per_point=6, seed=302))
    keys = ("poses", "points", "edge_chi2", "edge_outlier", "chi2_trace")
    ref = {}
    for name, p in (("a", pa), ("b", pb)):
        r = p.solve()
        ref[name] = ({k: np.array(r[k], copy=True) for k in keys}, list(r["n_iters"]), list(r["n_trials"]))
    assert not np.array_equal(ref["a"][0]["poses"], ref["b"][0]["poses"])
    for it in range(40):
        for name, p in (("a", pa), ("b", pb)):
            r = p.solve()
            for k in keys:
                assert np.array_equal(r[k], ref[name][0][k]), (it, name, k)
            assert list(r["n_iters"]) == ref[name][1] and list(r["n_trials"]) == ref[name][2]
    for it in range(10):
        out = pkg.Optimizer.LocalBundleAdjustmentBatch([pa, pb] if it % 2 == 0 else [pb, pa])
        names = ("a", "b") if it % 2 == 0 else ("b", "a")
        for r, name in zip(out, names):
            for k in keys:
                assert np.array_equal(r[k], ref[name][0][k]), ("batch", it, name, k)


@pytest.mark.parametrize("kw", [dict(n_poses=9, n_fixed=3, n_points=80, obs_per_point=5, seed=21),
                                dict(n_poses=9, n_fixed=3, n_points=80, obs_per_point=5, seed=21, exact_adjoint=True),
                                dict(n_poses=14, n_fixed=0, n_points=200, obs_per_point=7, seed=23),
                                dict(n_poses=50, n_fixed=10, n_points=2000, obs_per_point=10, seed=42)])
def test_first_linearisation_blocks_vs_oracle(pkg, oracle, synth, kw):
    """Rows a14 / a15 tapped directly (dcs_ba_debug_linearize): H_pp, H_ll, H_pl, b_p, b_l after ONE linearisation -- computeError,
    linearizeOplus (types_six_dof_expmap.cpp:123-161: J_pose through the rig's 'adjoint', exact or the reference's with the zero block;
    J_point through R(T_ext T_mcs)) and constructQuadraticForm with Huber's rho' (base_binary_edge.hpp:55-120) -- block by block against
    the oracle's buildSystem at rtol 1e-12. A Jacobian sign or a swapped row that Levenberg-Marquardt would still converge through cannot
    hide here; fixed poses own no block and their edges no H_pl. 5 % of the edges are gross outliers, so both Huber branches are hit."""
    pb = synth.ba_problem(**kw)
    if kw.get("n_fixed") == 0:
        pb = dict(pb); pb["pose_fixed"] = pb["pose_fixed"].copy(); pb["pose_fixed"][4] = 1      # one fixed pose in the middle of the index range
    got = pkg.Optimizer.prepare(pb).linearize()
    prob = dict(pb)
    prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
    exp = oracle.ba_linearize(prob)
    assert got["n_free"] == exp["n_free"] == int((pb["pose_fixed"] == 0).sum()) and np.array_equal(got["pose_idx"], exp["pose_idx"])
    for k in ("Hpp", "bp", "Hll", "bl", "Hpl"):
        g, e = got[k], exp[k]
        assert g.shape == e.shape
        scale = np.abs(e).reshape(len(e), -1).max(axis=1).reshape((-1,) + (1,) * (e.ndim - 1))      # per block: entries relative to the block's largest
        assert np.all(np.abs(g - e) <= 1e-12 * np.maximum(scale, 1e-300)), (k, float(np.abs(g - e).max()))
    fixed_edges = pb["pose_fixed"][pb["edge_pose"]] != 0
    assert fixed_edges.any() and not np.any(got["Hpl"][fixed_edges]) and np.any(got["Hpl"][~fixed_edges])
    assert np.array_equal(got["Hpp"], np.transpose(got["Hpp"], (0, 2, 1)))                         # symmetric by construction (upper entries mirrored)
    # and the oracle's blocks are what the per-edge Jacobians say (numpy recomposition of one pose block and one point block)
    i = 0
    p_of = int(np.nonzero(exp["pose_idx"] == i)[0][0])
    H, b = np.zeros((6, 6)), np.zeros(6)
    delta = float(pb["huber_delta"])                           # the float-rounded sqrt(5.991) of Optimizer.cc:515
    for e_ in np.nonzero(pb["edge_pose"] == p_of)[0]:
        cam = prob["cams"][int(pb["edge_cam"][e_])]
        Jp, Jx = oracle.ba_edge_jacobian(pb["poses"][p_of], pb["points"][pb["edge_point"][e_]], cam)
        err, _z = oracle.ba_edge_error(pb["poses"][p_of], pb["points"][pb["edge_point"][e_]], cam, pb["obs"][e_])
        w = float(pb["inv_sigma2"][e_])
        chi2 = w * float(err @ err)
        rho1 = 1.0 if chi2 <= delta * delta else delta / np.sqrt(chi2)
        H += rho1 * w * (Jp.T @ Jp); b += -rho1 * w * (Jp.T @ err)
    assert np.allclose(exp["Hpp"][i], H, rtol=1e-10, atol=1e-9) and np.allclose(exp["bp"][i], b, rtol=1e-10, atol=1e-9)
