"""ctypes binding of the CPU parity oracle (TEST INFRASTRUCTURE -- see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: the reference has no golden vectors and cannot be built here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CANDIDATE = np.dtype([("x", "<i2"), ("y", "<i2"), ("score", "<i4")])
assert KEYPOINT.itemsize == 28 and CANDIDATE.itemsize == 8


def build(force=False):
    """Compile oracle/*.cpp into oracle/_build/liboracle.so (g++ -O3, no -march=native)."""
    srcs = [os.path.join(_HERE, f) for f in ("orb_oracle.cpp", "match_oracle.cpp", "ba_oracle.cpp", "bow_oracle.cpp",
                                             "oracle.h", "brief_pattern.inc", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


REFERENCE_DIR = os.environ.get("DCS_REFERENCE_DIR", "/root/reference")
_REF_PATH = os.path.join(_HERE, "_ref", "libref.so")
_ref = None


def build_ref():
    """oracle/_ref/libref.so: the dependency-free pieces of the REAL reference (DBoW2 BowVector / FeatureVector / ScoringObject,
    the Hamming loops of ORBmatcher::DescriptorDistance and FORB::distance) compiled from where they lie (Makefile target
    `ref`, ref_shim.cpp). Returns the path, or None where the reference tree does not exist (the GPU box)."""
    if os.path.isdir(REFERENCE_DIR):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref", "REF=" + REFERENCE_DIR])
    return _REF_PATH if os.path.exists(_REF_PATH) else None


def ref():
    """ctypes handle of oracle/_ref/libref.so or None"""
    global _ref
    if _ref is None:
        path = build_ref()
        if path is None:
            return None
        L = C.CDLL(path)
        vp, ci = C.c_void_p, C.c_int
        L.ref_descriptor_distance.argtypes = [vp, vp]
        L.ref_forb_distance.argtypes = [vp, vp]
        L.ref_bow_vector.argtypes = [ci, vp, vp, ci, ci, vp, vp]
        L.ref_score.argtypes = [ci, ci, vp, vp, ci, vp, vp]
        L.ref_score.restype = C.c_double
        L.ref_feature_vector.argtypes = [ci, vp, vp, vp, vp]
        L.ref_huber.argtypes = [C.c_double, C.c_double, vp]
        L.ref_huber.restype = None
        _ref = L
    return _ref


def ref_distances(a, b):
    """(ORBmatcher::DescriptorDistance, FORB::distance) of descriptor rows a[i], b[i] through the reference's own loops"""
    a, b = _c(a, np.uint8).reshape(-1, 32), _c(b, np.uint8).reshape(-1, 32)
    L = ref()
    d1 = np.array([L.ref_descriptor_distance(a[i].ctypes.data, b[i].ctypes.data) for i in range(len(a))], np.int32)
    d2 = np.array([L.ref_forb_distance(a[i].ctypes.data, b[i].ctypes.data) for i in range(len(a))], np.int32)
    return d1, d2


def ref_huber(delta, e):
    """g2o's RobustKernelHuber (the reference's own statements): setDelta(delta), robustify(e) -> (rho, rho', rho'')"""
    rho = np.zeros(3)
    ref().ref_huber(float(delta), float(e), _p(rho))
    return rho


def robust_huber(delta, e):
    """the oracle's Huber kernel (orc_robust_huber)"""
    rho = np.zeros(3)
    fn = lib().orc_robust_huber
    fn.argtypes = [C.c_double, C.c_double, C.c_void_p]
    fn.restype = None
    fn(float(delta), float(e), _p(rho))
    return rho


def ref_three_maxima(sizes):
    """ORBmatcher::ComputeThreeMaxima (the reference's own statements) on a histogram given as bin sizes -> (ind1, ind2, ind3)"""
    sizes = _c(sizes, np.int32)
    ind = np.zeros(3, np.int32)
    ref().ref_three_maxima(_p(sizes), len(sizes), _p(ind))
    return tuple(int(v) for v in ind)


def three_maxima(sizes):
    sizes = _c(sizes, np.int32)
    ind = np.zeros(3, np.int32)
    lib().orc_three_maxima(_p(sizes), len(sizes), _p(ind))
    return tuple(int(v) for v in ind)


def ref_bow_vector(word, weight, if_not_exist=False, norm=1):
    word, weight = _c(word, np.uint32), _c(weight, np.float64)
    ow, ov = np.zeros(max(len(word), 1), np.uint32), np.zeros(max(len(word), 1))
    n = ref().ref_bow_vector(len(word), _p(word), _p(weight), int(if_not_exist), int(norm), _p(ow), _p(ov))
    return ow[:n].astype(np.int32), ov[:n].copy()


def ref_score(kind, w1, v1, w2, v2):
    w1, w2, v1, v2 = _c(w1, np.uint32), _c(w2, np.uint32), _c(v1, np.float64), _c(v2, np.float64)
    return float(ref().ref_score(int(kind), len(w1), _p(w1), _p(v1), len(w2), _p(w2), _p(v2)))


def ref_feature_vector(node):
    node = _c(node, np.uint32)
    n = len(node)
    on, oo, oi = np.zeros(max(n, 1), np.uint32), np.zeros(n + 1, np.int32), np.zeros(max(n, 1), np.uint32)
    k = ref().ref_feature_vector(n, _p(node), _p(on), _p(oo), _p(oi))
    return on[:k].astype(np.int32), oo[:k + 1].copy(), oi[:oo[k]].astype(np.int32)


class BaCamera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("ext", C.c_double * 7), ("adj", C.c_double * 36)]


class BaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32), ("n_cams", C.c_int32),
                ("poses", C.c_void_p), ("pose_fixed", C.c_void_p), ("points", C.c_void_p),
                ("edge_pose", C.c_void_p), ("edge_point", C.c_void_p), ("edge_cam", C.c_void_p),
                ("obs", C.c_void_p), ("inv_sigma2", C.c_void_p), ("cams", C.c_void_p),
                ("huber_delta", C.c_double), ("chi2_th", C.c_double),
                ("iters1", C.c_int32), ("iters2", C.c_int32)]


class BaResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("points", C.c_void_p), ("edge_chi2", C.c_void_p),
                ("edge_outlier", C.c_void_p), ("edge_level1", C.c_void_p),
                ("n_iters", C.c_int32 * 2), ("n_trials", C.c_int32 * 2), ("lambda_", C.c_double * 2),
                ("chi2_trace", C.c_double * 32)]


class ProjFrame(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("cam_off", C.c_void_p), ("kp_x", C.c_void_p), ("kp_y", C.c_void_p),
                ("kp_octave", C.c_void_p), ("kp_angle", C.c_void_p), ("desc", C.c_void_p), ("taken", C.c_void_p),
                ("min_x", C.c_void_p), ("min_y", C.c_void_p), ("grid_w_inv", C.c_void_p), ("grid_h_inv", C.c_void_p),
                ("grid_off", C.c_void_p), ("grid_idx", C.c_void_p)]


class ProjQueries(C.Structure):
    _fields_ = [("n", C.c_int32), ("valid", C.c_void_p), ("cam", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p),
                ("radius", C.c_void_p), ("min_level", C.c_void_p), ("max_level", C.c_void_p), ("desc", C.c_void_p),
                ("angle", C.c_void_p)]


def _proj_structs(frame, queries):
    """dicts (see synth.projection_problem) -> (ProjFrame, ProjQueries, keep-alive list)"""
    keep = []

    def a(x, dt):
        arr = _c(x, dt)
        keep.append(arr)
        return _p(arr).value
    f = ProjFrame(len(frame["cam_off"]) - 1, a(frame["cam_off"], np.int32), a(frame["kp_x"], np.float32), a(frame["kp_y"], np.float32),
                  a(frame["kp_octave"], np.int32), a(frame["kp_angle"], np.float32), a(frame["desc"], np.uint8),
                  a(frame["taken"], np.uint8), a(frame["min_x"], np.float32), a(frame["min_y"], np.float32),
                  a(frame["grid_w_inv"], np.float32), a(frame["grid_h_inv"], np.float32), a(frame["grid_off"], np.int32),
                  a(frame["grid_idx"], np.int32))
    q = ProjQueries(len(queries["cam"]), a(queries["valid"], np.uint8), a(queries["cam"], np.int32), a(queries["u"], np.float32),
                    a(queries["v"], np.float32), a(queries["radius"], np.float32), a(queries["min_level"], np.int32),
                    a(queries["max_level"], np.int32), a(queries["desc"], np.uint8), a(queries["angle"], np.float32))
    return f, q, keep


def frame_grid(cam_off, kp_x, kp_y, min_x, min_y, w_inv, h_inv):
    cam_off = _c(cam_off, np.int32)
    n_cams, N = len(cam_off) - 1, int(cam_off[-1])
    kx, ky = _c(kp_x, np.float32), _c(kp_y, np.float32)
    mx, my, wi, hi = (_c(v, np.float32) for v in (min_x, min_y, w_inv, h_inv))
    off = np.zeros(n_cams * 64 * 48 + 1, np.int32)
    idx = np.zeros(max(N, 1), np.int32)
    n = lib().orc_frame_grid(n_cams, _p(cam_off), _p(kx), _p(ky), _p(mx), _p(my), _p(wi), _p(hi), _p(off), _p(idx))
    return off, idx[:n]


def features_in_area(frame, c, x, y, r, min_level, max_level):
    f, _, keep = _proj_structs(frame, dict(valid=[], cam=[], u=[], v=[], radius=[], min_level=[], max_level=[], desc=np.zeros((0, 32), np.uint8), angle=[]))
    out = np.zeros(max(int(frame["cam_off"][-1]), 1), np.int32)
    L = lib()
    L.orc_features_in_area.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = L.orc_features_in_area(C.byref(f), int(c), float(x), float(y), float(r), int(min_level), int(max_level), _p(out), len(out))
    return out[:n]


def search_by_projection(frame, queries, th_high=100, nn_ratio=0.8, check_orientation=False):
    f, q, keep = _proj_structs(frame, queries)
    N, n = int(frame["cam_off"][-1]), len(queries["cam"])
    mq, qf, nm = np.full(max(n, 1), -1, np.int32), np.full(max(N, 1), -1, np.int32), C.c_int32()
    L = lib()
    L.orc_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_search_by_projection(C.byref(f), C.byref(q), int(th_high), float(nn_ratio), int(check_orientation), _p(mq), _p(qf), C.byref(nm))
    return mq[:n], qf[:N], nm.value


def search_by_projection_kf(frame, queries, th=50):
    """SearchByProjection(KF, query, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:416-536): (match_of_query, query_of_feature, n)"""
    f, q, keep = _proj_structs(frame, queries)
    N, n = int(frame["cam_off"][-1]), len(queries["cam"])
    mq, qf, nm = np.full(max(n, 1), -1, np.int32), np.full(max(N, 1), -1, np.int32), C.c_int32()
    L = lib()
    L.orc_search_by_projection_kf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_search_by_projection_kf(C.byref(f), C.byref(q), int(th), _p(mq), _p(qf), C.byref(nm))
    return mq[:n], qf[:N], nm.value


def search_in_window(frame, queries, th=50, kf_area=True, chi2_inv_sigma2=None):
    """Fuse x2 / SearchBySim3CrossCam / SearchByProjection(KF, ...) candidate loops (independent queries):
    (match_of_query [global feature or -1], best_dist, accepted)"""
    f, q, keep = _proj_structs(frame, queries)
    n = len(queries["cam"])
    mq, bd = np.full(max(n, 1), -1, np.int32), np.full(max(n, 1), 256, np.int32)
    chi = None if chi2_inv_sigma2 is None else _c(chi2_inv_sigma2, np.float32)
    L = lib()
    L.orc_search_in_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    acc = L.orc_search_in_window(C.byref(f), C.byref(q), int(th), int(bool(kf_area)), None if chi is None else _p(chi), _p(mq), _p(bd))
    return mq[:n], bd[:n], acc


def search_for_initialization(frame2, queries, nn_ratio=0.9, check_orientation=True):
    """SearchForInitialization (ORBmatcher.cc:1117-1251): (match12 [global F2 feature or -1], nmatches)"""
    f, q, keep = _proj_structs(frame2, queries)
    n = len(queries["cam"])
    m12 = np.full(max(n, 1), -1, np.int32)
    L = lib()
    L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    nm = L.orc_search_for_initialization(C.byref(f), C.byref(q), float(nn_ratio), int(check_orientation), _p(m12))
    return m12[:n], nm


class PoseProblem(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_cams", C.c_int32), ("poses", C.c_void_p), ("edge_off", C.c_void_p),
                ("xw", C.c_void_p), ("obs", C.c_void_p), ("inv_sigma2", C.c_void_p), ("edge_cam", C.c_void_p),
                ("cams", C.c_void_p), ("huber_delta", C.c_double), ("chi2_th", C.c_float * 4), ("its", C.c_int32 * 4)]


class PoseResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("outlier", C.c_void_p), ("n_inliers", C.c_void_p), ("edge_chi2", C.c_void_p),
                ("n_iters", C.c_void_p)]


class FrustumFrame(C.Structure):
    _fields_ = [("n_cams", C.c_int32)] + [(k, C.c_void_p) for k in ("Rsw", "tsw", "Ow", "fx", "fy", "cx", "cy", "min_x", "max_x", "min_y", "max_y")] + \
               [("log_scale_factor", C.c_float), ("n_scale_levels", C.c_int32), ("scale_factors", C.c_void_p)]


def _frustum_frame(cls, fr, keep):
    a = {k: _c(fr[k], np.float32) for k in ("Rsw", "tsw", "Ow", "fx", "fy", "cx", "cy", "min_x", "max_x", "min_y", "max_y", "scale_factors")}
    keep.append(a)
    n_cams = len(a["fx"])
    return cls(n_cams, *[a[k].ctypes.data for k in ("Rsw", "tsw", "Ow", "fx", "fy", "cx", "cy", "min_x", "max_x", "min_y", "max_y")],
               float(np.float32(fr["log_scale_factor"])), len(a["scale_factors"]), a["scale_factors"].ctypes.data)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_orb_create.restype = C.c_void_p
        L.orc_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_orb_destroy.argtypes = [C.c_void_p]
        L.orc_orb_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_orb_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_orb_level_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_orb_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_orb_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_gauss7_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_roi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_score.argtypes = [C.c_void_p, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_brief.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orc_distribute_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_knn2_grouped.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
        L.orc_ratio_rot_filter.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_search_by_bow_crosscam.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_float, C.c_int, C.c_void_p]
        L.orc_ba_local.argtypes = [C.POINTER(BaProblem), C.c_void_p, C.POINTER(BaResult)]
        L.orc_ba_edge_error.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BaCamera), C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.orc_ba_edge_jacobian.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BaCamera), C.c_void_p, C.c_void_p]
        L.orc_se3_oplus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rig_adjoint.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_is_in_frustum.argtypes = [C.POINTER(FrustumFrame), C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_float] + [C.c_void_p] * 7
        L.orc_vocab_create.restype = C.c_void_p
        L.orc_vocab_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_vocab_destroy.argtypes = [C.c_void_p]
        L.orc_vocab_words.argtypes = [C.c_void_p]
        L.orc_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.POINTER(C.c_int)] + [C.c_void_p] * 3 + [C.POINTER(C.c_int)]
        L.orc_bow_score_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class OrbOracle:
    """Mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-113) on the CPU restatement."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self._h = lib().orc_orb_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self._h:
            raise ValueError("bad extractor parameters")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_orb_destroy(self._h)
            self._h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        npl = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        lib().orc_orb_tables(self._h, _p(sc), _p(isc), _p(s2), _p(is2), _p(npl), _p(umax))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, n_per_level=npl, umax=umax)

    def extract(self, img, cap=None):
        """operator()(image) -> (keypoints[N] structured, descriptors[N,32] u8)."""
        img = _c(img, np.uint8)
        rows, cols = img.shape
        cap = cap or max(self.nfeatures * 2 + 64, 64)
        kp = np.zeros(cap, KEYPOINT)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        rc = lib().orc_orb_extract(self._h, _p(img), rows, cols, cols, _p(kp), _p(desc), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError("orc_orb_extract rc=%d (n=%d, cap=%d)" % (rc, n.value, cap))
        return kp[:n.value].copy(), desc[:n.value].copy()

    def level_dims(self, level):
        w, h = C.c_int(), C.c_int()
        lib().orc_orb_level_dims(self._h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_image(self, level, blurred=False):
        w, h = self.level_dims(level)
        out = np.zeros((h, w), np.uint8)
        n = lib().orc_orb_level_copy(self._h, level, int(blurred), _p(out))
        return out if n else None

    def level_candidates(self, level):
        n = lib().orc_orb_level_candidates(self._h, level, None, 0)
        out = np.zeros(max(n, 1), CANDIDATE)
        lib().orc_orb_level_candidates(self._h, level, _p(out), n)
        return out[:n]

    def level_keypoints(self, level):
        n = lib().orc_orb_level_keypoints(self._h, level, None, 0)
        out = np.zeros(max(n, 1), KEYPOINT)
        lib().orc_orb_level_keypoints(self._h, level, _p(out), n)
        return out[:n]


def resize_linear_u8(src, dw, dh):
    src = _c(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
    return dst


def gauss7_u8(src):
    src = _c(src, np.uint8)
    dst = np.zeros_like(src)
    lib().orc_gauss7_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), src.shape[1])
    return dst


def fast_roi(roi, threshold):
    roi = _c(roi, np.uint8)
    cap = roi.size
    out = np.zeros(max(cap, 1), CANDIDATE)
    n = lib().orc_fast_roi(_p(roi), roi.shape[1], roi.shape[0], roi.shape[1], threshold, _p(out), cap)
    return out[:n]


def fast_score(img, x, y):
    img = _c(img, np.uint8)
    ptr = img.ctypes.data + y * img.shape[1] + x
    return lib().orc_fast_score(C.c_void_p(ptr), img.shape[1])


def fast_atan2(y, x):
    return lib().orc_fast_atan2(float(y), float(x))


def ic_angle(img, x, y):
    img = _c(img, np.uint8)
    return lib().orc_ic_angle(_p(img), img.shape[1], x, y)


def brief(blurred, x, y, angle_deg):
    blurred = _c(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orc_brief(_p(blurred), blurred.shape[1], x, y, float(angle_deg), _p(d))
    return d


def distribute_octree(cand, min_x, max_x, min_y, max_y, n_target):
    cand = _c(cand, CANDIDATE)
    out = np.zeros(max(len(cand), 1), CANDIDATE)
    n = lib().orc_distribute_octree(_p(cand), len(cand), min_x, max_x, min_y, max_y, n_target, _p(out), len(out))
    return out[:n]


def descriptor_distance(a, b):
    a, b = _c(a, np.uint8), _c(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def sincosf(x, restated=False):
    """cosf / sinf of float32 radians: the host libm's (what the reference calls, ORBextractor.cc:112-113) or the oracle's
    restatement of glibc's algorithm (the expressions the GPU kernel evaluates)."""
    x = _c(x, np.float32)
    c, s = np.zeros(len(x), np.float32), np.zeros(len(x), np.float32)
    fn = lib().orc_sincosf_restated if restated else lib().orc_libm_sincosf
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    fn.restype = None
    fn(_p(x), len(x), _p(c), _p(s))
    return c, s


def knn2(q, t, t_mask=None):
    q, t = _c(q, np.uint8).reshape(-1, 32), _c(t, np.uint8).reshape(-1, 32)
    nq, nt = len(q), len(t)
    bi, bd, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(3))
    m = _c(t_mask, np.uint8) if t_mask is not None else None
    lib().orc_knn2(_p(q), nq, _p(t), nt, _p(m), _p(bi), _p(bd), _p(sd))
    return bi[:nq], bd[:nq], sd[:nq]


def distinctive_descriptors(pool, off, idx):
    pool = _c(pool, np.uint8).reshape(-1, 32)
    off, idx = _c(off, np.int32), _c(idx, np.int32)
    best = np.zeros(max(len(off) - 1, 1), np.int32)
    lib().orc_distinctive_descriptors(_p(pool), _p(off), _p(idx), len(off) - 1, _p(best))
    return best[:len(off) - 1]


def knn2_grouped(q, t, q_off, q_idx, t_off, t_idx):
    q, t = _c(q, np.uint8).reshape(-1, 32), _c(t, np.uint8).reshape(-1, 32)
    q_off, q_idx, t_off, t_idx = (_c(a, np.int32) for a in (q_off, q_idx, t_off, t_idx))
    nq = len(q)
    bi, bd, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(3))
    lib().orc_knn2_grouped(_p(q), nq, _p(t), len(t), len(q_off) - 1, _p(q_off), _p(q_idx), _p(t_off), _p(t_idx),
                           _p(bi), _p(bd), _p(sd))
    return bi[:nq], bd[:nq], sd[:nq]


def ratio_rot_filter(best_idx, best_d, second_d, th=50, th_strict=False, ratio=0.75, check_ori=True,
                     q_angle=None, t_angle=None):
    best_idx, best_d, second_d = (_c(a, np.int32) for a in (best_idx, best_d, second_d))
    nq = len(best_idx)
    qa = _c(q_angle, np.float32) if q_angle is not None else np.zeros(max(nq, 1), np.float32)
    ta = _c(t_angle, np.float32) if t_angle is not None else np.zeros(1, np.float32)
    match = np.full(max(nq, 1), -1, np.int32)
    n = lib().orc_ratio_rot_filter(nq, _p(best_idx), _p(best_d), _p(second_d), th, int(th_strict), float(ratio),
                                   int(check_ori), _p(qa), _p(ta), _p(match))
    return match[:nq], n


def search_by_bow_crosscam(desc_kf, ang_kf, kf_valid, desc_f, ang_f, kf_fv, f_fv, ratio=0.75, check_ori=True):
    """kf_fv / f_fv: (nodes, off, idx) CSR feature vectors with ascending node ids."""
    desc_kf, desc_f = _c(desc_kf, np.uint8).reshape(-1, 32), _c(desc_f, np.uint8).reshape(-1, 32)
    ang_kf, ang_f = _c(ang_kf, np.float32), _c(ang_f, np.float32)
    kf_valid = _c(kf_valid, np.uint8)
    kn, ko, ki = (_c(a, np.int32) for a in kf_fv)
    fn, fo, fi = (_c(a, np.int32) for a in f_fv)
    match = np.full(max(len(desc_f), 1), -1, np.int32)
    n = lib().orc_search_by_bow_crosscam(_p(desc_kf), _p(ang_kf), _p(kf_valid), len(desc_kf),
                                         _p(desc_f), _p(ang_f), len(desc_f),
                                         _p(kn), _p(ko), _p(ki), len(kn), _p(fn), _p(fo), _p(fi), len(fn),
                                         float(ratio), int(check_ori), _p(match))
    return match[:len(desc_f)], n


def search_by_bow_kfkf(desc1, ang1, valid1, desc2, ang2, valid2, fv1, fv2, ratio=0.75, check_ori=True):
    """SearchByBoWCrossCam(KF1, c1, KF2, c2) (ORBmatcher.cc:297-414) -> (match12[n1], nmatches)"""
    desc1, desc2 = _c(desc1, np.uint8).reshape(-1, 32), _c(desc2, np.uint8).reshape(-1, 32)
    ang1, ang2, valid1, valid2 = _c(ang1, np.float32), _c(ang2, np.float32), _c(valid1, np.uint8), _c(valid2, np.uint8)
    n1_, o1, i1 = (_c(a, np.int32) for a in fv1)
    n2_, o2, i2 = (_c(a, np.int32) for a in fv2)
    match = np.full(max(len(desc1), 1), -1, np.int32)
    fn = lib().orc_search_by_bow_kfkf
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] * 2 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] * 2 + [C.c_float, C.c_int, C.c_void_p]
    n = fn(_p(desc1), _p(ang1), _p(valid1), len(desc1), _p(desc2), _p(ang2), _p(valid2), len(desc2),
           _p(n1_), _p(o1), _p(i1), len(n1_), _p(n2_), _p(o2), _p(i2), len(n2_), float(ratio), int(check_ori), _p(match))
    return match[:len(desc1)], n


def search_for_triangulation(desc1, ang1, free1, desc2, ang2, free2, fv1, fv2, epi, check_ori=True):
    """SearchForTriangulation (ORBmatcher.cc:1253-1427) for one camera. epi = dict(F12[9], ex, ey, kp1_x, kp1_y, kp2_x, kp2_y,
    kp2_octave, level_sigma2, scale_factors) -> (match12[n1], nmatches)"""
    desc1, desc2 = _c(desc1, np.uint8).reshape(-1, 32), _c(desc2, np.uint8).reshape(-1, 32)
    ang1, ang2, free1, free2 = _c(ang1, np.float32), _c(ang2, np.float32), _c(free1, np.uint8), _c(free2, np.uint8)
    n1_, o1, i1 = (_c(a, np.int32) for a in fv1)
    n2_, o2, i2 = (_c(a, np.int32) for a in fv2)
    F = _c(epi["F12"], np.float32).reshape(9)
    x1, y1, x2, y2 = (_c(epi[k], np.float32) for k in ("kp1_x", "kp1_y", "kp2_x", "kp2_y"))
    oc, sg, sc = _c(epi["kp2_octave"], np.int32), _c(epi["level_sigma2"], np.float32), _c(epi["scale_factors"], np.float32)
    match = np.full(max(len(desc1), 1), -1, np.int32)
    fn = lib().orc_search_for_triangulation
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    fn.argtypes = [vp, vp, vp, ci] * 2 + [vp, vp, vp, ci] * 2 + [vp, cf, cf, vp, vp, vp, vp, vp, vp, vp, ci, vp]
    n = fn(_p(desc1), _p(ang1), _p(free1), len(desc1), _p(desc2), _p(ang2), _p(free2), len(desc2),
           _p(n1_), _p(o1), _p(i1), len(n1_), _p(n2_), _p(o2), _p(i2), len(n2_), _p(F), float(np.float32(epi["ex"])), float(np.float32(epi["ey"])),
           _p(x1), _p(y1), _p(x2), _p(y2), _p(oc), _p(sg), _p(sc), int(check_ori), _p(match))
    return match[:len(desc1)], n


def make_camera(fx, fy, cx, cy, ext7, adj36):
    c = BaCamera()
    c.fx, c.fy, c.cx, c.cy = fx, fy, cx, cy
    for i in range(7):
        c.ext[i] = float(ext7[i])
    for i in range(36):
        c.adj[i] = float(np.asarray(adj36).reshape(-1)[i])
    return c


def rig_adjoint(T44_f32, exact=False):
    T = _c(T44_f32, np.float32).reshape(16)
    adj = np.zeros(36, np.float64)
    ext = np.zeros(7, np.float64)
    lib().orc_rig_adjoint(_p(T), int(exact), _p(adj), _p(ext))
    return adj.reshape(6, 6), ext


def ba_edge_error(pose7, point3, cam, obs2):
    pose7, point3, obs2 = (_c(a, np.float64) for a in (pose7, point3, obs2))
    e = np.zeros(2)
    z = C.c_double()
    lib().orc_ba_edge_error(_p(pose7), _p(point3), C.byref(cam), _p(obs2), _p(e), C.byref(z))
    return e, z.value


def ba_edge_jacobian(pose7, point3, cam):
    pose7, point3 = _c(pose7, np.float64), _c(point3, np.float64)
    jp, jx = np.zeros(12), np.zeros(6)
    lib().orc_ba_edge_jacobian(_p(pose7), _p(point3), C.byref(cam), _p(jp), _p(jx))
    return jp.reshape(2, 6), jx.reshape(2, 3)


def se3_oplus(pose7, update6):
    pose7, update6 = _c(pose7, np.float64), _c(update6, np.float64)
    out = np.zeros(7)
    lib().orc_se3_oplus(_p(pose7), _p(update6), _p(out))
    return out


def pose_optimization(prob):
    """prob: dict as made by synth.pose_problem (cams = list of BaCamera)."""
    poses = _c(prob["poses"], np.float64)
    off, cam = _c(prob["edge_off"], np.int32), _c(prob["edge_cam"], np.int32)
    xw, obs, w = (_c(prob[k], np.float64) for k in ("xw", "obs", "inv_sigma2"))
    cams = (BaCamera * len(prob["cams"]))(*prob["cams"])
    F, E = len(poses), len(cam)
    pb = PoseProblem(F, len(prob["cams"]), _p(poses).value, _p(off).value, _p(xw).value, _p(obs).value, _p(w).value,
                     _p(cam).value, C.cast(cams, C.c_void_p).value, float(prob["huber_delta"]),
                     (C.c_float * 4)(*prob["chi2_th"]), (C.c_int32 * 4)(*prob["its"]))
    out_poses, outl, ninl = np.zeros((F, 7)), np.zeros(max(E, 1), np.uint8), np.zeros(F, np.int32)
    chi2, nit = np.zeros(max(E, 1)), np.zeros((F, 4), np.int32)
    res = PoseResult(_p(out_poses).value, _p(outl).value, _p(ninl).value, _p(chi2).value, _p(nit).value)
    rc = lib().orc_pose_optimization(C.byref(pb), C.byref(res))
    if rc != 0:
        raise RuntimeError("orc_pose_optimization rc=%d" % rc)
    return dict(poses=out_poses, outlier=outl[:E], n_inliers=ninl, edge_chi2=chi2[:E], n_iters=nit)


def ba_local(prob, stop_flag=None):
    """prob: dict with poses[P,7], pose_fixed[P], points[L,3], edge_pose/point/cam[E], obs[E,2],
    inv_sigma2[E], cams (list of BaCamera), huber_delta, chi2_th, iters1, iters2."""
    poses = _c(prob["poses"], np.float64)
    fixed = _c(prob["pose_fixed"], np.uint8)
    points = _c(prob["points"], np.float64)
    ep, el, ec = (_c(prob[k], np.int32) for k in ("edge_pose", "edge_point", "edge_cam"))
    obs = _c(prob["obs"], np.float64)
    w = _c(prob["inv_sigma2"], np.float64)
    cams = (BaCamera * len(prob["cams"]))(*prob["cams"])
    P, L, E = len(poses), len(points), len(ep)
    pb = BaProblem(P, L, E, len(prob["cams"]), _p(poses).value, _p(fixed).value, _p(points).value,
                   _p(ep).value, _p(el).value, _p(ec).value, _p(obs).value, _p(w).value,
                   C.cast(cams, C.c_void_p).value,
                   float(prob.get("huber_delta", np.sqrt(5.991))), float(prob.get("chi2_th", 5.991)),
                   int(prob.get("iters1", 5)), int(prob.get("iters2", 10)))
    out_poses, out_points = np.zeros((P, 7)), np.zeros((L, 3))
    chi2, outl, lvl1 = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8)
    res = BaResult(_p(out_poses).value, _p(out_points).value, _p(chi2).value, _p(outl).value, _p(lvl1).value)
    sf = _p(stop_flag) if stop_flag is not None else None
    rc = lib().orc_ba_local(C.byref(pb), sf, C.byref(res))
    if rc != 0:
        raise RuntimeError("orc_ba_local rc=%d" % rc)
    return dict(poses=out_poses, points=out_points, edge_chi2=chi2, edge_outlier=outl, edge_level1=lvl1,
                n_iters=list(res.n_iters), n_trials=list(res.n_trials), lambda_=list(res.lambda_),
                chi2_trace=np.array(res.chi2_trace))


def ba_linearize(prob):
    """orc_ba_linearize: H / b blocks after the first linearisation of ba_local's problem (no lambda, no solve)"""
    poses = _c(prob["poses"], np.float64)
    fixed = _c(prob["pose_fixed"], np.uint8)
    points = _c(prob["points"], np.float64)
    ep, el, ec = (_c(prob[k], np.int32) for k in ("edge_pose", "edge_point", "edge_cam"))
    obs = _c(prob["obs"], np.float64)
    w = _c(prob["inv_sigma2"], np.float64)
    cams = (BaCamera * len(prob["cams"]))(*prob["cams"])
    P, L, E = len(poses), len(points), len(ep)
    pb = BaProblem(P, L, E, len(prob["cams"]), _p(poses).value, _p(fixed).value, _p(points).value,
                   _p(ep).value, _p(el).value, _p(ec).value, _p(obs).value, _p(w).value,
                   C.cast(cams, C.c_void_p).value,
                   float(prob.get("huber_delta", np.sqrt(5.991))), float(prob.get("chi2_th", 5.991)),
                   int(prob.get("iters1", 5)), int(prob.get("iters2", 10)))
    Hpp, bp, Hll, bl, Hpl = np.zeros((P, 6, 6)), np.zeros((P, 6)), np.zeros((L, 3, 3)), np.zeros((L, 3)), np.zeros((E, 6, 3))
    pose_idx, n_free = np.zeros(P, np.int32), C.c_int()
    f = lib().orc_ba_linearize
    f.restype = C.c_int
    rc = f(C.byref(pb), _p(Hpp), _p(bp), _p(Hll), _p(bl), _p(Hpl), _p(pose_idx), C.byref(n_free))
    if rc != 0:
        raise RuntimeError("orc_ba_linearize rc=%d" % rc)
    n = n_free.value
    return dict(Hpp=Hpp[:n], bp=bp[:n], Hll=Hll, bl=bl, Hpl=Hpl, pose_idx=pose_idx, n_free=n)


class Vocabulary:
    """DBoW2 vocabulary tree from the columns of the reference's text format (row i = node i + 1)."""

    def __init__(self, k, L, parent, is_leaf, desc, weight, scoring=0, weighting=0):
        parent, is_leaf = _c(parent, np.int32), _c(is_leaf, np.uint8)
        desc, weight = _c(desc, np.uint8).reshape(-1, 32), _c(weight, np.float64)
        self._h = lib().orc_vocab_create(k, L, scoring, weighting, len(parent), _p(parent), _p(is_leaf), _p(desc), _p(weight))
        if not self._h:
            raise ValueError("bad vocabulary")
        self.k, self.L = k, L

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_vocab_destroy(self._h)
            self._h = None

    def n_words(self):
        return lib().orc_vocab_words(self._h)

    def transform(self, desc, levelsup=4):
        """-> dict(word[n], node[n], bow_word, bow_val, fv_node, fv_off, fv_idx)"""
        desc = _c(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        m = max(n, 1)
        word, node, bw, fn, fi = (np.zeros(m, np.int32) for _ in range(5))
        fo = np.zeros(m + 1, np.int32)
        bv = np.zeros(m)
        nw, nn = C.c_int(), C.c_int()
        rc = lib().orc_bow_transform(self._h, _p(desc), n, levelsup, _p(word), _p(node), _p(bw), _p(bv), C.byref(nw), _p(fn), _p(fo), _p(fi), C.byref(nn))
        if rc != 0:
            raise RuntimeError("orc_bow_transform rc=%d" % rc)
        return dict(word=word[:n], node=node[:n], bow_word=bw[:nw.value].copy(), bow_val=bv[:nw.value].copy(), fv_node=fn[:nn.value].copy(),
                    fv_off=fo[:nn.value + 1].copy(), fv_idx=fi[:fo[nn.value]].copy())


def bow_vector(word, weight, if_not_exist=False, norm=1):
    """the oracle's BowVector accumulation (addWeight / addIfNotExist in feature order) + normalize, as transform() runs it"""
    word, weight = _c(word, np.uint32), _c(weight, np.float64)
    ow, ov = np.zeros(max(len(word), 1), np.int32), np.zeros(max(len(word), 1))
    fn = lib().orc_bow_vector
    fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    n = fn(len(word), _p(word), _p(weight), int(if_not_exist), int(norm), _p(ow), _p(ov))
    return ow[:n].copy(), ov[:n].copy()


def descriptor_distances(a, b):
    a, b = _c(a, np.uint8).reshape(-1, 32), _c(b, np.uint8).reshape(-1, 32)
    L = lib()
    return np.array([L.orc_descriptor_distance(a[i].ctypes.data, b[i].ctypes.data) for i in range(len(a))], np.int32)


def bow_score_l1(q_word, q_val, db_off, db_word, db_val):
    q_word, q_val = _c(q_word, np.int32), _c(q_val, np.float64)
    db_off, db_word, db_val = _c(db_off, np.int32), _c(db_word, np.int32), _c(db_val, np.float64)
    n_db = len(db_off) - 1
    score = np.zeros(max(n_db, 1))
    lib().orc_bow_score_l1(_p(q_word), _p(q_val), len(q_word), _p(db_off), _p(db_word), _p(db_val), n_db, _p(score))
    return score[:n_db]


def detect_candidates(loop, query_id, q_word, q_val, db, dead, covis, state, connected=None, min_score=0.0):
    """KeyFrameDatabase::DetectLoopCandidatesForCam / DetectRelocalizationCandidates on flat arrays. db = list of (word, val) BowVectors
    in insertion order; covis = list of neighbour entry lists; state = dict(query=int32[n], words=int32[n], score=float32[n]), updated
    in place. Returns the candidate entry ids in the reference's order."""
    n = len(db)
    off = np.cumsum([0] + [len(w) for w, _ in db]).astype(np.int32)
    dbw = _c(np.concatenate([np.asarray(w, np.int32) for w, _ in db]) if n else np.zeros(0, np.int32), np.int32)
    dbv = _c(np.concatenate([np.asarray(v, np.float64) for _, v in db]) if n else np.zeros(0), np.float64)
    coff = np.cumsum([0] + [len(c) for c in covis]).astype(np.int32)
    cidx = _c(np.concatenate([np.asarray(c, np.int32) for c in covis]) if sum(len(c) for c in covis) else np.zeros(1, np.int32), np.int32)
    q_word, q_val = _c(q_word, np.int32), _c(q_val, np.float64)
    dead = _c(dead, np.uint8)
    conn = _c(connected if connected is not None else np.zeros(max(n, 1), np.uint8), np.uint8)
    out = np.zeros(max(n, 1), np.int32)
    L = lib()
    L.orc_detect_candidates.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    assert state["query"].dtype == np.int32 and state["words"].dtype == np.int32 and state["score"].dtype == np.float32
    k = L.orc_detect_candidates(int(bool(loop)), int(query_id), _p(q_word), _p(q_val), len(q_word), n, _p(off), _p(dbw), _p(dbv), _p(dead), _p(conn),
                                float(min_score), _p(coff), _p(cidx), _p(state["query"]), _p(state["words"]), _p(state["score"]), _p(out), len(out))
    return [int(v) for v in out[:k]]


def undistort_points(xy, K4, dist):
    """cv::undistortPoints(xy, K, dist, noArray(), K) on float points [n, 2] (Frame::UndistortKeyPoints); K4 = (fx, fy, cx, cy), dist = (k1, k2, p1, p2[, k3])"""
    xy = _c(xy, np.float32).reshape(-1, 2)
    K4, dist = _c(K4, np.float32), _c(dist, np.float32)
    out = np.zeros((max(len(xy), 1), 2), np.float32)
    L = lib()
    L.orc_undistort_points.restype = None
    L.orc_undistort_points.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_undistort_points(len(xy), _p(xy), _p(K4), _p(dist), len(dist), _p(out))
    return out[:len(xy)]


def motion_model_queries(frame, pos, q_cam, q_octave, th):
    """the query columns (valid, u, v, radius, min_level, max_level) of SearchByProjectionOnCam for the last frame's features with a map point"""
    keep = []
    f = _frustum_frame(FrustumFrame, frame, keep)
    pos, q_cam, q_octave = _c(pos, np.float32).reshape(-1, 3), _c(q_cam, np.int32), _c(q_octave, np.int32)
    n = len(pos)
    m = max(n, 1)
    out = dict(valid=np.zeros(m, np.uint8), u=np.zeros(m, np.float32), v=np.zeros(m, np.float32), radius=np.zeros(m, np.float32),
               min_level=np.zeros(m, np.int32), max_level=np.zeros(m, np.int32))
    L = lib()
    L.orc_motion_model_queries.restype = None
    L.orc_motion_model_queries.argtypes = [C.POINTER(FrustumFrame), C.c_int] + [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 6
    L.orc_motion_model_queries(C.byref(f), n, _p(pos), _p(q_cam), _p(q_octave), float(th), *[_p(out[k]) for k in ("valid", "u", "v", "radius", "min_level", "max_level")])
    return {k: a[:n] for k, a in out.items()}


def is_in_frustum(frame, pts, viewing_cos_limit=0.5, th=1.0):
    """Frame::isInFrustum + PredictScale + search window for pts = dict(pos[n,3], normal[n,3], min_dist[n], max_dist[n], candidate[n] or None)."""
    keep = []
    f = _frustum_frame(FrustumFrame, frame, keep)
    pos, nrm = _c(pts["pos"], np.float32).reshape(-1, 3), _c(pts["normal"], np.float32).reshape(-1, 3)
    mind, maxd = _c(pts["min_dist"], np.float32), _c(pts["max_dist"], np.float32)
    cand = _c(pts["candidate"], np.uint8) if pts.get("candidate") is not None else None
    n = len(pos)
    m = max(n, 1)
    out = dict(in_view=np.zeros(m, np.uint8), cam=np.zeros(m, np.int32), u=np.zeros(m, np.float32), v=np.zeros(m, np.float32),
               view_cos=np.zeros(m, np.float32), level=np.zeros(m, np.int32), radius=np.zeros(m, np.float32))
    lib().orc_is_in_frustum(C.byref(f), n, _p(pos), _p(nrm), _p(mind), _p(maxd), _p(cand), float(viewing_cos_limit), float(th),
                            *[_p(out[k]) for k in ("in_view", "cam", "u", "v", "view_cos", "level", "radius")])
    return {k: a[:n] for k, a in out.items()}
