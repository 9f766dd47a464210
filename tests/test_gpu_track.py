"""The device-resident tracking chain dcs_track_local_map = Tracking::SearchLocalPoints (src/Tracking.cc:1617-1680: Frame::isInFrustum for
every local map point, then ORBmatcher(0.8).SearchByProjection) + Optimizer::PoseOptimization (Tracking.cc:1321), batched over frames,
against the ORACLE's three stages composed on the host exactly as the reference strings them together: frustum outputs -> queries of the
search (level -+ 1, radius), the search's assignment -> the optimiser's edges in ascending feature order. Matching is exact; poses to
1e-9 (the optimiser's own parity bar, tests/test_gpu_ba.py::test_pose_optimization_vs_oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _with_grid(pkg, frames):
    for fr in frames:
        ft = fr["features"]
        ft["grid_off"], ft["grid_idx"] = pkg.frame_grid(ft["cam_off"], ft["kp_x"], ft["kp_y"], ft["min_x"], ft["min_y"], ft["grid_w_inv"], ft["grid_h_inv"])
    return frames


def _oracle_chain(oracle, fr, prm):
    ft, pts = fr["features"], fr["points"]
    fru = oracle.is_in_frustum(fr["view"], pts, prm["viewing_cos_limit"], prm["th"])
    q = dict(valid=fru["in_view"], cam=np.maximum(fru["cam"], 0), u=fru["u"], v=fru["v"], radius=fru["radius"], min_level=fru["level"] - 1,
             max_level=fru["level"] + 1, desc=fr["desc"], angle=np.zeros(len(fru["u"]), np.float32))
    mq, qf, nm = oracle.search_by_projection(ft, q, prm["th_high"], prm["nn_ratio"], False)
    N = int(ft["cam_off"][-1])
    src = np.where(qf >= 0, qf, np.where(fr["has_point"] != 0, -2, -1))
    feat = np.nonzero(src != -1)[0]
    xw = np.where((src[feat] >= 0)[:, None], pts["pos"][np.maximum(src[feat], 0)], fr["point_xw"][feat]).astype(np.float64)
    cam = (np.searchsorted(ft["cam_off"], feat, side="right") - 1).astype(np.int32)
    prob = dict(poses=fr["pose"][None, :], edge_off=np.array([0, len(feat)], np.int32), xw=xw,
                obs=np.stack([ft["kp_x"][feat], ft["kp_y"][feat]], 1).astype(np.float64),
                inv_sigma2=prm["inv_level_sigma2"][ft["kp_octave"][feat]].astype(np.float64), edge_cam=cam,
                cams=[oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in prm["cams"]],
                huber_delta=prm["huber_delta"], chi2_th=prm["chi2_th"], its=prm["its"])
    po = oracle.pose_optimization(prob)
    outl = np.zeros(N, np.uint8)
    outl[feat] = po["outlier"][:len(feat)]
    return dict(match_of_point=mq, point_of_feature=src.astype(np.int32), n_matches=nm, pose=po["poses"][0], n_inliers=int(po["n_inliers"][0]), outlier=outl,
                n_edges=len(feat))


@pytest.mark.parametrize("kw", [dict(n_frames=5, n_points=1200, n_features=900, seed=17),
                                dict(n_frames=3, n_points=2500, n_features=1600, seed=29, th=3.0),
                                dict(n_frames=2, n_points=400, n_features=300, seed=5, pre_matched=0.6),
                                dict(n_frames=2, n_points=9000, n_features=2000, seed=61),                 # a local map of the reference's size: more points in
                                dict(n_frames=1, n_points=14000, n_features=2000, seed=62, th=3.0)])      # view than the resolver keeps on chip (4 096)
def test_track_local_map_equals_the_three_stages(pkg, oracle, synth, kw):
    frames, prm = synth.tracking_problem(**kw)
    _with_grid(pkg, frames)
    got = pkg.abi.PreparedTracking(frames, prm).track()
    total_new = 0
    for k, fr in enumerate(frames):
        exp = _oracle_chain(oracle, fr, prm)
        g = got[k]
        assert np.array_equal(g["match_of_point"], exp["match_of_point"]), k
        assert np.array_equal(g["point_of_feature"], exp["point_of_feature"]), k
        assert g["n_matches"] == exp["n_matches"] and g["n_inliers"] == exp["n_inliers"], (k, g["n_matches"], exp["n_matches"], g["n_inliers"], exp["n_inliers"])
        assert np.array_equal(g["outlier"], exp["outlier"]), k
        assert np.abs(g["pose"] - exp["pose"]).max() < 1e-9, (k, float(np.abs(g["pose"] - exp["pose"]).max()))
        total_new += int((exp["point_of_feature"] >= 0).sum())
        # the synthetic scene is meaningful: the search assigns many points, mostly to the features that stem from them, and the pose moves towards the truth
        new = exp["point_of_feature"] >= 0
        assert new.sum() > 0.3 * (fr["feature_source"] >= 0).sum() * (1 - kw.get("pre_matched", 0.25))
        assert (fr["feature_source"][new] == exp["point_of_feature"][new]).mean() > 0.9
    assert total_new > 100


def test_track_local_map_equals_the_separate_entry_points(pkg, synth):
    """... and the three C-ABI calls issued one after the other through host buffers (what a caller would do without the chain)"""
    frames, prm = synth.tracking_problem(n_frames=2, n_points=900, n_features=700, seed=41)
    _with_grid(pkg, frames)
    got = pkg.abi.PreparedTracking(frames, prm).track()
    m = pkg.ORBmatcher(prm["nn_ratio"], False)
    for k, fr in enumerate(frames):
        ft, pts = fr["features"], fr["points"]
        fru = pkg.isInFrustum(fr["view"], pts, prm["viewing_cos_limit"], prm["th"])
        q = dict(valid=fru["in_view"], cam=np.maximum(fru["cam"], 0), u=fru["u"], v=fru["v"], radius=fru["radius"], min_level=fru["level"] - 1,
                 max_level=fru["level"] + 1, desc=fr["desc"], angle=np.zeros(len(fru["u"]), np.float32))
        mq, qf, nm = m.SearchByProjection(ft, q, prm["th_high"], use_ratio=True, check_orientation=False)
        assert np.array_equal(got[k]["match_of_point"], mq) and got[k]["n_matches"] == nm
        src = np.where(qf >= 0, qf, np.where(fr["has_point"] != 0, -2, -1))
        feat = np.nonzero(src != -1)[0]
        xw = np.where((src[feat] >= 0)[:, None], pts["pos"][np.maximum(src[feat], 0)], fr["point_xw"][feat]).astype(np.float64)
        prob = dict(poses=fr["pose"][None, :], edge_off=np.array([0, len(feat)], np.int32), xw=xw,
                    obs=np.stack([ft["kp_x"][feat], ft["kp_y"][feat]], 1).astype(np.float64),
                    inv_sigma2=prm["inv_level_sigma2"][ft["kp_octave"][feat]].astype(np.float64),
                    edge_cam=(np.searchsorted(ft["cam_off"], feat, side="right") - 1).astype(np.int32), cams=prm["cams"],
                    huber_delta=prm["huber_delta"], chi2_th=prm["chi2_th"], its=prm["its"])
        po = pkg.Optimizer.PoseOptimization(prob)
        assert np.array_equal(got[k]["pose"], po["poses"][0])                         # the same kernel on the same edges: bit for bit
        assert got[k]["n_inliers"] == int(po["n_inliers"][0])


def test_track_local_map_edge_cases(pkg, synth):
    frames, prm = synth.tracking_problem(n_frames=2, n_points=300, n_features=200, seed=3)
    _with_grid(pkg, frames)
    # a frame whose local map is empty: nothing is assigned, the pose is optimised over the points it already holds
    frames[1]["points"] = {k: v[:0] for k, v in frames[1]["points"].items()}
    frames[1]["desc"] = frames[1]["desc"][:0]
    got = pkg.abi.PreparedTracking(frames, prm).track()
    assert got[1]["n_matches"] == 0 and not np.any(got[1]["point_of_feature"] >= 0)
    assert np.array_equal(got[1]["point_of_feature"] == -2, frames[1]["has_point"] != 0)
    assert pkg.abi.PreparedTracking([], prm).track() == []


def test_track_local_map_null_arrays_for_an_empty_local_map(pkg, synth):
    """a C caller passes NULL for the arrays of an empty local map (n_points = 0): the uploads copy the real counts, nothing reads the pointers"""
    frames, prm = synth.tracking_problem(n_frames=2, n_points=300, n_features=200, seed=3)
    _with_grid(pkg, frames)
    frames[0]["points"] = {k: v[:0] for k, v in frames[0]["points"].items()}
    frames[0]["desc"] = frames[0]["desc"][:0]
    ref = pkg.abi.PreparedTracking(frames, prm).track()
    got = pkg.abi.PreparedTracking(frames, prm, null_empty=True).track()
    for a, b in zip(ref, got):
        assert a["n_matches"] == b["n_matches"] and a["n_inliers"] == b["n_inliers"]
        assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["point_of_feature"], b["point_of_feature"]) and np.array_equal(a["outlier"], b["outlier"])
    assert got[0]["n_matches"] == 0


def test_track_local_map_rejects_more_cameras_than_intrinsics(pkg, synth):
    """features.n_cams > prm->n_cams would index the optimiser's by-value camera table out of range: refused before anything is enqueued"""
    frames, prm = synth.tracking_problem(n_frames=1, n_points=200, n_features=150, seed=9)
    _with_grid(pkg, frames)
    prm = dict(prm)
    prm["cams"] = prm["cams"][:1]
    assert len(frames[0]["features"]["cam_off"]) - 1 == 2
    with pytest.raises(Exception) as ei:
        pkg.abi.PreparedTracking(frames, prm).track()
    assert "cameras" in str(ei.value)


# ---------------------------------------------------------------------------------------------------------------------------------
# dcs_track_frame_device: the chain fed from the extractor's slots in HBM (frame assembly + the 64 x 48 grid on the device), both modes

def _to_slots(ft, cap, first_slot=1, n_slots=None):
    """a frame's features laid out like dcs_orb_extract_batch_device's outputs: dcs_keypoint[slot][cap], u8[slot][cap][32], int32[slot]"""
    C_ = len(ft["cam_off"]) - 1
    n_slots = n_slots or first_slot + C_ + 1
    kp = np.zeros((n_slots, cap, 7), np.float32)
    kp[..., 2] = 31.0
    desc = np.zeros((n_slots, cap, 32), np.uint8)
    n = np.zeros(n_slots, np.int32)
    kpi = kp.view(np.int32)
    for c in range(C_):
        a, b = int(ft["cam_off"][c]), int(ft["cam_off"][c + 1])
        m = b - a
        assert m <= cap
        kp[first_slot + c, :m, 0] = ft["kp_x"][a:b]; kp[first_slot + c, :m, 1] = ft["kp_y"][a:b]; kp[first_slot + c, :m, 3] = ft["kp_angle"][a:b]
        kpi[first_slot + c, :m, 5] = ft["kp_octave"][a:b]; kpi[first_slot + c, :m, 6] = -1
        kp[first_slot + c, m:, 0] = 1e6                      # stale slot contents behind the count must never be looked at
        desc[first_slot + c, :m] = ft["desc"][a:b]; desc[first_slot + c, m:] = 0xA5
        n[first_slot + c] = m
    return kp, desc, n


def _device_frames(frames, cap, mode):
    import torch
    out, keep = [], []
    for fr in frames:
        ft = fr["features"]
        kp, desc, n = _to_slots(ft, cap)
        d_kp, d_desc, d_n = torch.from_numpy(kp).cuda(), torch.from_numpy(desc).cuda(), torch.from_numpy(n).cuda()
        keep += [d_kp, d_desc, d_n]
        dev = dict(d_kp=d_kp.data_ptr(), d_desc=d_desc.data_ptr(), d_n=d_n.data_ptr(), cap=cap, first_slot=1, n_cams=len(ft["cam_off"]) - 1,
                   min_x=ft["min_x"], min_y=ft["min_y"], grid_w_inv=ft["grid_w_inv"], grid_h_inv=ft["grid_h_inv"])
        d = dict(dev=dev, view=fr["view"], pose=fr["pose"])
        if mode == 0:
            d.update(held=dict(taken=ft["taken"], has_point=fr["has_point"], point_xw=fr["point_xw"]), points=fr["points"], desc=fr["desc"])
        else:
            mm = fr["mm"]
            d.update(held=None, points=dict(pos=mm["pos"]), desc=mm["desc"], q_cam=mm["q_cam"], q_octave=mm["q_octave"], q_angle=mm["q_angle"])
        out.append(d)
    torch.cuda.synchronize()
    return out, keep


@pytest.mark.parametrize("kw,cap", [(dict(n_frames=3, n_points=1300, n_features=1000, seed=57), 700),
                                    (dict(n_frames=3, n_points=1677, n_features=90, seed=976754755 % 100000, pre_matched=0.6), 118)])
def test_track_frame_device_mode0_equals_the_host_buffer_chain(pkg, synth, kw, cap):
    """features taken from the extractor's slot layout in HBM, assembled and gridded on the device: bit for bit dcs_track_local_map.
    (Second case: frames of fewer than 256 features -- the randomised sweep found stale staging bytes behind their `has_point` arrays.)"""
    frames, prm = synth.tracking_problem(**kw)
    _with_grid(pkg, frames)
    ref = pkg.abi.PreparedTracking(frames, prm).track()
    dfr, keep = _device_frames(frames, cap=cap, mode=0)
    got = pkg.abi.PreparedTrackingDevice(dfr, prm, mode=0).track()
    for k, (a, b) in enumerate(zip(ref, got)):
        ft = frames[k]["features"]
        assert np.array_equal(b["n_features"], np.diff(ft["cam_off"])), k
        assert np.array_equal(a["match_of_point"], b["match_of_point"]) and np.array_equal(a["point_of_feature"], b["point_of_feature"]), k
        assert a["n_matches"] == b["n_matches"] and a["n_inliers"] == b["n_inliers"] and np.array_equal(a["outlier"], b["outlier"]), k
        assert np.array_equal(a["pose"], b["pose"]), k


@pytest.mark.parametrize("mode", [0, 1])
def test_track_frame_device_first_call_of_a_cold_thread(pkg, synth, mode):
    """The arena of a host thread starts with one 1-MB block and opens a new hipMalloc block whenever a call outgrows the current one; round 5's
    dcs_track_frame_device zero-filled [first key-point array, end of the last one) as ONE range, which on a thread's first call (or a call four
    times larger than the one before) spanned two unrelated allocations. Every case here runs on a FRESH thread whose first call is
    dcs_track_frame_device, with F x cap chosen on both sides of the 1-MB boundary; then a second call on the same thread that is several times
    larger. All equal to the same frames through a warm thread's call (bit for bit: the chain is deterministic)."""
    import threading
    make = synth.tracking_problem if mode == 0 else synth.motion_model_problem
    frames, prm = make(n_frames=40, n_points=500, n_features=600, seed=91)
    _with_grid(pkg, frames)
    keys = ("match_of_point", "point_of_feature", "outlier", "pose", "n_features")
    ref = {}
    for cap in (700, 1096, 2096):                               # warm reference on this thread (its arena has seen larger calls already)
        dfr, keep = _device_frames(frames, cap=cap, mode=mode)
        pt = pkg.abi.PreparedTrackingDevice(dfr, prm, mode=mode)
        pt.track()
        ref[cap] = (dfr, keep, pt.track())
    errors = []

    def same(a, b):
        return a["n_matches"] == b["n_matches"] and a["n_inliers"] == b["n_inliers"] and all(np.array_equal(a[k], b[k]) for k in keys)

    def cold(F, cap, grow_to):
        try:
            dfr = ref[cap][0]
            got = pkg.abi.PreparedTrackingDevice(dfr[:F], prm, mode=mode).track()             # the thread's FIRST call
            for k in range(F):
                if not same(got[k], ref[cap][2][k]): errors.append(("first call", F, cap, k))
            if grow_to:
                got = pkg.abi.PreparedTrackingDevice(dfr[:grow_to], prm, mode=mode).track()    # outgrows the block the first call left
                for k in range(grow_to):
                    if not same(got[k], ref[cap][2][k]): errors.append(("grown call", F, grow_to, cap, k))
        except Exception as ex:                                 # noqa: BLE001 -- reported with the case
            errors.append((F, cap, grow_to, repr(ex)))

    for cap in (700, 1096, 2096):
        for F, grow_to in ((1, 0), (5, 0), (6, 24), (16, 0), (40, 0), (2, 9)):
            t = threading.Thread(target=cold, args=(F, cap, grow_to))
            t.start(); t.join()
    assert not errors, errors[:5]


def _oracle_motion_model(oracle, fr, prm, check_ori):
    ft, mm = fr["features"], fr["mm"]
    q = oracle.motion_model_queries(fr["view"], mm["pos"], mm["q_cam"], mm["q_octave"], prm["th"])
    nq, N = len(mm["pos"]), int(ft["cam_off"][-1])
    mq, qf, nm = np.full(nq, -1, np.int32), np.full(N, -1, np.int32), 0
    for c in range(len(ft["cam_off"]) - 1):                 # SearchByProjection(Fcur, Flast, th) = one SearchByProjectionOnCam per camera, a histogram each
        qc = dict(valid=(q["valid"] & (mm["q_cam"] == c)).astype(np.uint8), cam=mm["q_cam"], u=q["u"], v=q["v"], radius=q["radius"], min_level=q["min_level"],
                  max_level=q["max_level"], desc=mm["desc"], angle=mm["q_angle"])
        mq_c, qf_c, nm_c = oracle.search_by_projection(ft, qc, prm["th_high"], 0.0, check_ori)
        sel = mm["q_cam"] == c
        mq[sel] = mq_c[sel]; qf = np.maximum(qf, qf_c); nm += nm_c
    feat = np.nonzero(qf >= 0)[0]
    cam = (np.searchsorted(ft["cam_off"], feat, side="right") - 1).astype(np.int32)
    prob = dict(poses=fr["pose"][None, :], edge_off=np.array([0, len(feat)], np.int32), xw=mm["pos"][qf[feat]].astype(np.float64),
                obs=np.stack([ft["kp_x"][feat], ft["kp_y"][feat]], 1).astype(np.float64),
                inv_sigma2=prm["inv_level_sigma2"][ft["kp_octave"][feat]].astype(np.float64), edge_cam=cam,
                cams=[oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in prm["cams"]],
                huber_delta=prm["huber_delta"], chi2_th=prm["chi2_th"], its=prm["its"])
    po = oracle.pose_optimization(prob)
    outl = np.zeros(N, np.uint8)
    outl[feat] = po["outlier"][:len(feat)]
    return dict(match_of_point=mq, point_of_feature=qf, n_matches=nm, pose=po["poses"][0], n_inliers=int(po["n_inliers"][0]), outlier=outl)


@pytest.mark.parametrize("kw", [dict(n_frames=3, n_points=1200, n_features=900, seed=31, th=7.0, check_ori=True),
                                dict(n_frames=2, n_points=2000, n_features=1500, seed=77, th=14.0, check_ori=True),
                                dict(n_frames=2, n_points=600, n_features=500, seed=5, th=7.0, check_ori=False)])
def test_track_frame_device_mode1_motion_model_vs_oracle(pkg, oracle, synth, kw):
    """TrackWithMotionModel's search + optimisation (SearchByProjectionOnCam per camera with its rotation histogram, then PoseOptimization)
    on device-resident features, against the oracle's stages composed: geometry -> per-camera ordered search -> optimiser"""
    kw = dict(kw)
    check = kw.pop("check_ori")
    frames, prm = synth.motion_model_problem(**kw)
    _with_grid(pkg, frames)
    dfr, keep = _device_frames(frames, cap=1200, mode=1)
    got = pkg.abi.PreparedTrackingDevice(dfr, prm, mode=1, check_orientation=check).track()
    n_new = 0
    for k, fr in enumerate(frames):
        exp = _oracle_motion_model(oracle, fr, prm, check)
        g = got[k]
        assert np.array_equal(g["n_features"], np.diff(fr["features"]["cam_off"])), k
        assert np.array_equal(g["match_of_point"], exp["match_of_point"]), k
        assert np.array_equal(g["point_of_feature"], exp["point_of_feature"]), k
        assert g["n_matches"] == exp["n_matches"] and g["n_inliers"] == exp["n_inliers"], (k, g["n_matches"], exp["n_matches"], g["n_inliers"], exp["n_inliers"])
        assert np.array_equal(g["outlier"], exp["outlier"]), k
        assert np.abs(g["pose"] - exp["pose"]).max() < 1e-9, (k, float(np.abs(g["pose"] - exp["pose"]).max()))
        # the scene means something: most queries the last frame shared with this one find their feature
        truth = fr["mm"]["truth_feature"]
        hit = (exp["match_of_point"] == truth) & (truth >= 0)
        assert hit.sum() > 0.5 * (truth >= 0).sum(), (k, int(hit.sum()), int((truth >= 0).sum()))
        n_new += int(exp["n_matches"])
    assert n_new > 100


def test_track_frame_device_edge_cases(pkg, oracle, synth):
    """slots whose count exceeds the capacity are clamped to it, an error code in place of a count (the extractor's DCS_ERR_CAPACITY) means
    no features, a frame without queries only assembles, and several frames of different sizes share one call"""
    import torch
    frames, prm = synth.motion_model_problem(n_frames=3, n_points=500, n_features=400, seed=12)
    _with_grid(pkg, frames)
    dfr, keep = _device_frames(frames, cap=600, mode=1)
    # frame 1: no queries at all; frame 2: camera 1 reports an error code instead of a count
    dfr[1]["points"] = dict(pos=np.zeros((0, 3), np.float32)); dfr[1]["desc"] = np.zeros((0, 32), np.uint8)
    for k in ("q_cam", "q_octave"):
        dfr[1][k] = np.zeros(0, np.int32)
    dfr[1]["q_angle"] = np.zeros(0, np.float32)
    d_n2 = keep[3 * 2 + 2]
    n_before = d_n2.cpu().numpy().copy()
    d_n2[2] = -2
    torch.cuda.synchronize()
    got = pkg.abi.PreparedTrackingDevice(dfr, prm, mode=1, check_orientation=True).track()
    exp0 = _oracle_motion_model(oracle, frames[0], prm, True)
    assert np.array_equal(got[0]["match_of_point"], exp0["match_of_point"]) and np.abs(got[0]["pose"] - exp0["pose"]).max() < 1e-9
    assert got[1]["n_matches"] == 0 and got[1]["n_inliers"] == 0 and np.array_equal(got[1]["pose"], frames[1]["pose"])       # fewer than 3 edges: pose untouched
    assert np.array_equal(got[1]["n_features"], np.diff(frames[1]["features"]["cam_off"]))
    assert list(got[2]["n_features"]) == [int(n_before[1]), 0]
    cam1 = frames[2]["mm"]["q_cam"] == 1
    assert np.all(got[2]["match_of_point"][cam1] == -1)                                                                      # nothing to match in the empty camera
    assert np.all(got[2]["match_of_point"][~cam1] < int(n_before[1]))
    # a capacity below the count: the frame is what fits
    ft = frames[0]["features"]
    small = int(min(np.diff(ft["cam_off"]))) - 7
    kp, desc, n = _to_slots(dict(ft, cam_off=np.array([0, small, 2 * small], np.int32), kp_x=np.concatenate([ft["kp_x"][ft["cam_off"][c]:ft["cam_off"][c] + small] for c in (0, 1)]),
                                 kp_y=np.concatenate([ft["kp_y"][ft["cam_off"][c]:ft["cam_off"][c] + small] for c in (0, 1)]),
                                 kp_angle=np.concatenate([ft["kp_angle"][ft["cam_off"][c]:ft["cam_off"][c] + small] for c in (0, 1)]),
                                 kp_octave=np.concatenate([ft["kp_octave"][ft["cam_off"][c]:ft["cam_off"][c] + small] for c in (0, 1)]),
                                 desc=np.concatenate([ft["desc"][ft["cam_off"][c]:ft["cam_off"][c] + small] for c in (0, 1)])), small)
    n[1:3] = small + 50                                                                                                      # the extractor found more than the slot holds
    t = [torch.from_numpy(a).cuda() for a in (kp, desc, n)]
    d0 = dict(dfr[0]); d0["dev"] = dict(dfr[0]["dev"], d_kp=t[0].data_ptr(), d_desc=t[1].data_ptr(), d_n=t[2].data_ptr(), cap=small)
    torch.cuda.synchronize()
    g = pkg.abi.PreparedTrackingDevice([d0], prm, mode=1, check_orientation=True).track()[0]
    assert list(g["n_features"]) == [small, small] and len(g["point_of_feature"]) == 2 * small


def test_track_frame_device_undistorts_key_points_like_the_frame_constructor(pkg, oracle, synth):
    """the shipped rig has k1 = -0.37 (Dual-LenaCV.yaml:17, 30): Frame::UndistortKeyPoints (Frame.cc:410-441) runs on the device when K / dist are given.
    The chain fed with DISTORTED key points in the slots + the coefficients equals the host-buffer chain fed with the oracle's cv::undistortPoints of them."""
    frames, prm = synth.tracking_problem(n_frames=2, n_points=900, n_features=700, seed=71)
    K = np.array([[558.4684, 560.0944, 326.7993, 262.9017], [546.598, 546.254, 332.759, 247.385]], np.float32)
    dist = np.array([[-0.3689, 0.1627, 0.0, 0.0, 0.0], [-0.361851421593862, 0.140443638558527, 0.0, 0.0, 0.0]], np.float32)
    und = []
    for fr in frames:                                           # the synthetic key points play the distorted ones; the host-buffer chain gets their undistorted positions
        ft = dict(fr["features"])
        x, y = ft["kp_x"].copy(), ft["kp_y"].copy()
        for c in (0, 1):
            a, b = int(ft["cam_off"][c]), int(ft["cam_off"][c + 1])
            u = oracle.undistort_points(np.stack([x[a:b], y[a:b]], 1), K[c], dist[c])
            x[a:b], y[a:b] = u[:, 0], u[:, 1]
        ft["kp_x"], ft["kp_y"] = x, y
        f2 = dict(fr); f2["features"] = ft
        und.append(f2)
        assert np.abs(x - fr["features"]["kp_x"]).max() > 3.0    # the distortion moves key points by pixels
    _with_grid(pkg, und)
    ref = pkg.abi.PreparedTracking(und, prm).track()
    dfr, keep = _device_frames(frames, cap=500, mode=0)
    for d in dfr:
        d["dev"]["K"], d["dev"]["dist"] = K, dist
    got = pkg.abi.PreparedTrackingDevice(dfr, prm, mode=0).track()
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a["match_of_point"], b["match_of_point"]) and np.array_equal(a["point_of_feature"], b["point_of_feature"]), k
        assert a["n_matches"] == b["n_matches"] and a["n_inliers"] == b["n_inliers"] and np.array_equal(a["outlier"], b["outlier"]), k
        assert np.array_equal(a["pose"], b["pose"]), k
    # dist == NULL / k1 == 0: taken as they are (the other tests of this file)
