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
                                dict(n_frames=2, n_points=400, n_features=300, seed=5, pre_matched=0.6)])
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
