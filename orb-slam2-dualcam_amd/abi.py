"""ctypes binding of include/dcs_abi.h (libdcs_hip.so) -- the host-side Python mirror used by
tests/ and bench.py. The reference is C++ (no Python), so this layer only exists to drive the C ABI:
class / method names follow the reference's seams (ORBextractor, ORBmatcher, Optimizer).

No CPU fallback: if the library is missing or no GPU is visible, calls raise DcsError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCS_LIB_PATH") or os.path.join(_HERE, "lib", "libdcs_hip.so")      # DCS_LIB_PATH: A/B timing of two builds (scratch/)

KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CANDIDATE = np.dtype([("x", "<i2"), ("y", "<i2"), ("score", "<i4")])

DCS_OK, DCS_ERR_INVALID, DCS_ERR_CAPACITY, DCS_ERR_HIP, DCS_ERR_NO_DEVICE, DCS_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5

# every symbol include/dcs_abi.h declares (checked by tests/test_abi_symbols.py)
SYMBOLS = [
    "dcs_last_error", "dcs_version", "dcs_device_count", "dcs_option_count", "dcs_option_name", "dcs_option_get", "dcs_option_set",
    "dcs_orb_create", "dcs_orb_destroy", "dcs_orb_tables", "dcs_orb_extract", "dcs_orb_extract_batch",
    "dcs_orb_extract_batch_device", "dcs_orb_debug_level_dims", "dcs_orb_debug_level",
    "dcs_orb_debug_candidates", "dcs_orb_debug_quadtree_fallbacks", "dcs_orb_debug_host_path", "dcs_orb_debug_fast_hw", "dcs_orb_debug_emit_levels", "dcs_debug_sincosf", "dcs_orb_required_cap", "dcs_orb_last_timing", "dcs_orb_timing_totals", "dcs_orb_set_timing", "dcs_distribute_octree",
    "dcs_hamming_knn2", "dcs_hamming_knn2_grouped", "dcs_match_filter", "dcs_match_bf",
    "dcs_match_bf_batch_device", "dcs_search_by_bow", "dcs_search_by_bow_kf", "dcs_search_for_triangulation", "dcs_distinctive_descriptors", "dcs_pose_optimization", "dcs_frame_grid", "dcs_search_by_projection", "dcs_search_by_projection_kf", "dcs_search_in_window", "dcs_search_for_initialization",
    "dcs_ba_local", "dcs_ba_local_batch", "dcs_ba_debug_linearize", "dcs_ba_timing", "dcs_track_local_map", "dcs_track_frame_device", "dcs_undistort_points", "dcs_rig_adjoint", "dcs_pose_from_matrix", "dcs_pose_to_matrix",
    "dcs_comm_unique_id", "dcs_comm_create", "dcs_comm_destroy", "dcs_comm_info", "dcs_features_allgather",
    "dcs_stream_create_cu_range", "dcs_stream_destroy", "dcs_host_alloc", "dcs_host_free", "dcs_streams_share_queue", "dcs_stream_create_apart", "dcs_ba_avoid_streams", "dcs_ba_set_cu_range", "dcs_ba_release_thread",
    "dcs_is_in_frustum", "dcs_vocab_create", "dcs_vocab_destroy", "dcs_vocab_info", "dcs_bow_transform_device", "dcs_bow_transform", "dcs_bow_score_l1",
    "dcs_kfdb_create", "dcs_kfdb_destroy", "dcs_kfdb_add", "dcs_kfdb_erase", "dcs_kfdb_clear", "dcs_kfdb_size", "dcs_kfdb_query",
]


class DcsError(RuntimeError):
    def __init__(self, rc, where):
        self.rc = rc
        msg = lib().dcs_last_error().decode() if _lib is not None else ""
        super().__init__("%s failed: rc=%d %s" % (where, rc, msg))


class FrustumFrame(C.Structure):
    _fields_ = [("n_cams", C.c_int32)] + [(k, C.c_void_p) for k in ("Rsw", "tsw", "Ow", "fx", "fy", "cx", "cy", "min_x", "max_x", "min_y", "max_y")] + \
               [("log_scale_factor", C.c_float), ("n_scale_levels", C.c_int32), ("scale_factors", C.c_void_p)]


class Epipolar(C.Structure):
    _fields_ = [("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float)] + \
               [(k, C.c_void_p) for k in ("kp1_x", "kp1_y", "kp2_x", "kp2_y", "kp2_octave", "level_sigma2", "scale_factors")] + [("n_levels", C.c_int32)]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("device", C.c_int32),
                ("max_images", C.c_int32), ("host_threads", C.c_int32)]


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
                ("chi2_trace", C.c_double * 32), ("gpu_ms", C.c_float)]


_lib = None


def lib():
    """Load libdcs_hip.so (raises if it was not built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libdcs_hip.so not built (run __graft_entry__.build() / ./build.sh); no CPU fallback exists")
        try:
            # torch bundles its own libamdhip64.so.7 (ROCm 7.0); ours is linked against the same SONAME. Two HIP
            # runtimes in one process cannot both own the GPU, so let torch's load first and share it.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.dcs_last_error.restype = C.c_char_p
        L.dcs_version.restype = C.c_char_p
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        pci = C.POINTER(ci)
        sigs = {
            "dcs_orb_create": [C.POINTER(OrbParams), C.POINTER(vp)],
            "dcs_orb_destroy": [vp],
            "dcs_orb_tables": [vp] * 6,
            "dcs_orb_extract": [vp, vp, ci, ci, ci, vp, vp, ci, pci],
            "dcs_orb_extract_batch": [vp, vp, ci, ci, ci, ci, vp, vp, ci, vp],
            "dcs_orb_extract_batch_device": [vp, vp, ci, ci, ci, ci, vp, vp, ci, vp, vp],
            "dcs_orb_debug_level_dims": [vp, ci, pci, pci],
            "dcs_orb_debug_level": [vp, ci, ci, ci, vp],
            "dcs_orb_debug_candidates": [vp, ci, ci, vp, ci, pci],
            "dcs_orb_debug_quadtree_fallbacks": [vp, pci],
            "dcs_orb_debug_host_path": [vp, pci, pci],
            "dcs_orb_debug_fast_hw": [vp, pci],
            "dcs_orb_debug_emit_levels": [vp, pci],
            "dcs_debug_sincosf": [vp, ci, vp, vp],
            "dcs_orb_required_cap": [vp, ci, ci, pci],
            "dcs_orb_last_timing": [vp, vp],
            "dcs_orb_timing_totals": [vp, vp, vp, ci],
            "dcs_orb_set_timing": [vp, ci],
            "dcs_distribute_octree": [vp, ci, ci, ci, ci, ci, ci, vp, ci, pci],
            "dcs_hamming_knn2": [vp, ci, vp, ci, vp, vp, vp, vp],
            "dcs_hamming_knn2_grouped": [vp, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp],
            "dcs_match_filter": [ci, vp, vp, vp, ci, ci, cf, ci, vp, vp, vp, pci],
            "dcs_match_bf": [vp, vp, ci, vp, vp, ci, ci, cf, ci, vp, pci],
            "dcs_match_bf_batch_device": [vp, vp, vp, ci, vp, ci, ci, cf, ci, vp, vp, vp, vp, vp],
            "dcs_search_by_bow": [vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, cf, ci, vp, pci],
            "dcs_search_by_bow_kf": [vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, cf, ci, vp, pci],
            "dcs_search_for_triangulation": [vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, C.POINTER(Epipolar), ci, vp, pci],
            "dcs_distinctive_descriptors": [vp, ci, vp, vp, ci, vp],
            "dcs_ba_local": [C.POINTER(BaProblem), vp, C.POINTER(BaResult)],
            "dcs_ba_local_batch": [ci, vp, vp, vp],
            "dcs_ba_debug_linearize": [C.POINTER(BaProblem), vp, vp, vp, vp, vp, vp, pci],
            "dcs_ba_timing": [ci, vp],
            "dcs_track_local_map": [ci, vp, vp, vp],
            "dcs_track_frame_device": [ci, vp, vp, ci, ci, vp, vp],
            "dcs_stream_create_cu_range": [ci, ci, C.POINTER(vp)],
            "dcs_streams_share_queue": [vp, vp, pci],
            "dcs_stream_create_apart": [vp, ci, C.POINTER(vp), pci],
            "dcs_ba_avoid_streams": [vp, ci],
            "dcs_host_alloc": [C.POINTER(vp), C.c_size_t],
            "dcs_host_free": [vp],
            "dcs_ba_set_cu_range": [ci, ci],
            "dcs_ba_release_thread": [],
            "dcs_comm_unique_id": [vp],
            "dcs_comm_create": [vp, ci, ci, C.POINTER(vp)],
            "dcs_comm_destroy": [vp],
            "dcs_comm_info": [vp, pci, pci],
            "dcs_features_allgather": [vp, vp, vp, vp, ci, ci, vp, vp, vp, vp],
            "dcs_pose_optimization": [C.POINTER(PoseProblem), C.POINTER(PoseResult)],
            "dcs_frame_grid": [ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, pci],
            "dcs_search_by_projection": [C.POINTER(ProjFrame), C.POINTER(ProjQueries), ci, cf, ci, vp, vp, pci],
            "dcs_search_by_projection_kf": [C.POINTER(ProjFrame), C.POINTER(ProjQueries), ci, vp, vp, pci],
            "dcs_kfdb_create": [C.POINTER(vp)], "dcs_kfdb_destroy": [vp], "dcs_kfdb_add": [vp, vp, vp, ci, pci], "dcs_kfdb_erase": [vp, ci],
            "dcs_kfdb_clear": [vp], "dcs_kfdb_size": [vp, pci], "dcs_kfdb_query": [vp, vp, vp, ci, vp, vp, vp],
            "dcs_search_in_window": [C.POINTER(ProjFrame), C.POINTER(ProjQueries), ci, ci, vp, ci, vp, vp, pci],
            "dcs_search_for_initialization": [C.POINTER(ProjFrame), C.POINTER(ProjQueries), cf, ci, vp, pci],
            "dcs_rig_adjoint": [vp, ci, vp, vp],
            "dcs_pose_from_matrix": [vp, vp],
            "dcs_pose_to_matrix": [vp, vp],
            "dcs_is_in_frustum": [C.POINTER(FrustumFrame), ci, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp, vp, vp],
            "dcs_vocab_create": [ci, ci, ci, ci, ci, vp, vp, vp, vp, C.POINTER(vp)],
            "dcs_vocab_destroy": [vp],
            "dcs_vocab_info": [vp, pci, pci, pci, pci],
            "dcs_bow_transform_device": [vp, vp, vp, ci, ci, ci] + [vp] * 11,
            "dcs_bow_transform": [vp, vp, ci, ci, vp, vp, vp, vp, pci, vp, vp, vp, pci],
            "dcs_bow_score_l1": [vp, vp, ci, vp, vp, vp, ci, vp],
        }
        for name, argtypes in sigs.items():
            fn = getattr(L, name, None)       # a missing symbol is reported by tests/test_abi_symbols.py
            if fn is not None:
                fn.argtypes = argtypes
        if hasattr(L, "dcs_orb_destroy"):
            L.dcs_orb_destroy.restype = None
        if hasattr(L, "dcs_vocab_destroy"):
            L.dcs_vocab_destroy.restype = None
            L.dcs_kfdb_destroy.restype = None
        if hasattr(L, "dcs_comm_destroy"):
            L.dcs_comm_destroy.restype = None
        if hasattr(L, "dcs_stream_destroy"):
            L.dcs_stream_destroy.restype = None
            L.dcs_stream_destroy.argtypes = [vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _check(rc, where):
    if rc != DCS_OK:
        raise DcsError(rc, where)


def set_option(name, value):
    """process-wide value of a library option (include/dcs_abi.h "options"): read per call by the handle-less entry points, copied by an
    extractor handle when it is created"""
    L = lib()
    L.dcs_option_set.argtypes = [C.c_char_p, C.c_int64]
    _check(L.dcs_option_set(name.encode(), int(value)), "dcs_option_set(%s)" % name)


def get_option(name):
    L = lib()
    L.dcs_option_get.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    v = C.c_int64()
    _check(L.dcs_option_get(name.encode(), C.byref(v)), "dcs_option_get(%s)" % name)
    return v.value


class options:
    """with options(DCS_ORB_FUSED_BLUR=0, ...): the values inside the block, the previous ones restored afterwards"""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def device_count():
    return lib().dcs_device_count()


class ORBextractor:
    """ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-113) on the HIP library."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7,
                 device=-1, max_images=2, host_threads=0):
        self.nfeatures, self.nlevels, self.max_images = nfeatures, nlevels, max_images
        self._h = C.c_void_p()
        prm = OrbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, max_images, host_threads)
        _check(lib().dcs_orb_create(C.byref(prm), C.byref(self._h)), "dcs_orb_create")

    def close(self):
        if getattr(self, "_h", None):
            lib().dcs_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def default_cap(self, rows=None, cols=None):
        if rows and cols:
            return self.required_cap(rows, cols)
        return self.nfeatures + 4 * self.nlevels + 64

    def required_cap(self, rows, cols):
        n = C.c_int()
        _check(lib().dcs_orb_required_cap(self._h, rows, cols, C.byref(n)), "dcs_orb_required_cap")
        return n.value

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        npl = np.zeros(n, np.int32)
        _check(lib().dcs_orb_tables(self._h, _p(sc), _p(isc), _p(s2), _p(is2), _p(npl)), "dcs_orb_tables")
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, n_per_level=npl)

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactors(self):
        return self.tables()["scale"]

    def __call__(self, image, cap=None):
        """operator()(image, mask, keypoints, descriptors) -> (keypoints[N], descriptors[N,32])."""
        kps, descs = self.extract_batch([image], cap)
        return kps[0], descs[0]

    def extract_batch(self, images, cap=None, stride=None):
        """images: [rows, cols] uint8 arrays; with `stride` they are passed as they lie (rows `stride` bytes apart, e.g. HostFrames.frames)"""
        images = [_c(im, np.uint8) for im in images] if stride is None else list(images)
        n = len(images)
        rows, cols = images[0].shape if images[0].ndim == 2 else (0, 0)
        cap = cap or max(self.default_cap(), self.required_cap(rows, cols) if rows and cols else 0)
        kp = np.zeros((n, cap), KEYPOINT)
        desc = np.zeros((n, cap, 32), np.uint8)
        n_out = np.zeros(n, np.int32)
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in images])
        rc = lib().dcs_orb_extract_batch(self._h, C.cast(ptrs, C.c_void_p), n, rows, cols, stride or cols, _p(kp), _p(desc), cap, _p(n_out))
        _check(rc, "dcs_orb_extract_batch")
        return [kp[i, :n_out[i]].copy() for i in range(n)], [desc[i, :n_out[i]].copy() for i in range(n)]

    def extract_batch_device(self, d_images, d_kp, d_desc, d_n, cap, stream=None, cols=None):
        """d_images: torch uint8 [n, rows, stride] on the GPU (cols <= stride valid bytes per row); outputs torch
        buffers (slotted)."""
        n, rows, stride = d_images.shape
        rc = lib().dcs_orb_extract_batch_device(self._h, d_images.data_ptr(), n, rows, cols or stride, stride, d_kp.data_ptr(),
                                                d_desc.data_ptr(), cap, d_n.data_ptr(), stream)
        _check(rc, "dcs_orb_extract_batch_device")

    def emit_levels(self):
        """pyramid levels the FAST cells of the last call produced (0: the k_resize chain did)"""
        v = C.c_int()
        _check(lib().dcs_orb_debug_emit_levels(self._h, C.byref(v)), "dcs_orb_debug_emit_levels")
        return v.value

    def fast_hw(self):
        """1: k_fast_cells runs its hardware-specific forms on this handle (the start-up probe passed), 0: the plain forms"""
        v = C.c_int()
        _check(lib().dcs_orb_debug_fast_hw(self._h, C.byref(v)), "dcs_orb_debug_fast_hw")
        return v.value

    def host_path(self):
        """(direct, graph_replayed) of the last host-buffer call"""
        d, g = C.c_int(), C.c_int()
        _check(lib().dcs_orb_debug_host_path(self._h, C.byref(d), C.byref(g)), "dcs_orb_debug_host_path")
        return d.value, g.value

    def level_dims(self, level):
        w, h = C.c_int(), C.c_int()
        _check(lib().dcs_orb_debug_level_dims(self._h, level, C.byref(w), C.byref(h)), "dcs_orb_debug_level_dims")
        return w.value, h.value

    def level_image(self, image, level, blurred=False):
        w, h = self.level_dims(level)
        out = np.zeros((h, w), np.uint8)
        _check(lib().dcs_orb_debug_level(self._h, image, level, int(blurred), _p(out)), "dcs_orb_debug_level")
        return out

    def level_candidates(self, image, level):
        n = C.c_int()
        _check(lib().dcs_orb_debug_candidates(self._h, image, level, None, 0, C.byref(n)), "dcs_orb_debug_candidates")
        out = np.zeros(max(n.value, 1), CANDIDATE)
        _check(lib().dcs_orb_debug_candidates(self._h, image, level, _p(out), n.value, C.byref(n)), "dcs_orb_debug_candidates")
        return out[:n.value]

    def quadtree_fallbacks(self):
        n = C.c_int()
        _check(lib().dcs_orb_debug_quadtree_fallbacks(self._h, C.byref(n)), "dcs_orb_debug_quadtree_fallbacks")
        return n.value

    STAGES = ("pyramid_us", "fast_us", "compact_us", "blur_us", "quadtree_us", "describe_us", "total_us")

    def set_timing(self, mode):
        """stage markers of later calls: 0 none, 1 the FAST stage only, 2 every stage (the default); dcs_orb_set_timing"""
        _check(lib().dcs_orb_set_timing(self._h, int(mode)), "dcs_orb_set_timing")

    def timing_totals(self, reset=False):
        """(dict of summed stage microseconds, number of calls) since the last reset; does not stall async callers."""
        sums = np.zeros(7, np.float64)
        n = C.c_int64()
        _check(lib().dcs_orb_timing_totals(self._h, _p(sums), C.byref(n), int(reset)), "dcs_orb_timing_totals")
        return dict(zip(self.STAGES, sums.tolist())), n.value

    def last_timing(self):
        t = np.zeros(7, np.float32)
        _check(lib().dcs_orb_last_timing(self._h, _p(t)), "dcs_orb_last_timing")
        return dict(zip(("pyramid_us", "fast_us", "compact_us", "blur_us", "quadtree_us", "describe_us", "total_us"),
                        t.tolist()))


def debug_sincosf(x):
    """the describe kernel's cosf / sinf (glibc's algorithm on the GPU) for float32 angles in radians"""
    x = _c(x, np.float32)
    c, s = np.zeros(len(x), np.float32), np.zeros(len(x), np.float32)
    _check(lib().dcs_debug_sincosf(_p(x), len(x), _p(c), _p(s)), "dcs_debug_sincosf")
    return c, s


def distribute_octree(cand, min_x, max_x, min_y, max_y, n_target):
    cand = _c(cand, CANDIDATE)
    out = np.zeros(max(len(cand), 1), CANDIDATE)
    n = C.c_int()
    _check(lib().dcs_distribute_octree(_p(cand), len(cand), min_x, max_x, min_y, max_y, n_target, _p(out), len(out), C.byref(n)),
           "dcs_distribute_octree")
    return out[:n.value]


class ORBmatcher:
    """The inner kernels of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:45-309)."""
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)

    @staticmethod
    def knn2(q, t, t_mask=None):
        q, t = _c(q, np.uint8).reshape(-1, 32), _c(t, np.uint8).reshape(-1, 32)
        nq = len(q)
        bi, bd, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(3))
        m = _c(t_mask, np.uint8) if t_mask is not None else None
        _check(lib().dcs_hamming_knn2(_p(q), nq, _p(t), len(t), _p(m), _p(bi), _p(bd), _p(sd)), "dcs_hamming_knn2")
        return bi[:nq], bd[:nq], sd[:nq]

    @staticmethod
    def knn2_grouped(q, t, q_off, q_idx, t_off, t_idx):
        q, t = _c(q, np.uint8).reshape(-1, 32), _c(t, np.uint8).reshape(-1, 32)
        q_off, q_idx, t_off, t_idx = (_c(a, np.int32) for a in (q_off, q_idx, t_off, t_idx))
        if len(q_off) != len(t_off) or len(q_off) < 1:
            raise ValueError("q_off and t_off describe the same groups: equal lengths (n_groups + 1) expected")
        nq = len(q)
        bi, bd, sd = (np.zeros(max(nq, 1), np.int32) for _ in range(3))
        _check(lib().dcs_hamming_knn2_grouped(_p(q), nq, _p(t), len(t), len(q_off) - 1, _p(q_off), _p(q_idx), _p(t_off),
                                              _p(t_idx), _p(bi), _p(bd), _p(sd)), "dcs_hamming_knn2_grouped")
        return bi[:nq], bd[:nq], sd[:nq]

    def filter(self, best_idx, best_d, second_d, th=50, th_strict=False, q_angle=None, t_angle=None):
        best_idx, best_d, second_d = (_c(a, np.int32) for a in (best_idx, best_d, second_d))
        nq = len(best_idx)
        qa = _c(q_angle, np.float32) if q_angle is not None else None
        ta = _c(t_angle, np.float32) if t_angle is not None else None
        match = np.full(max(nq, 1), -1, np.int32)
        n = C.c_int()
        _check(lib().dcs_match_filter(nq, _p(best_idx), _p(best_d), _p(second_d), th, int(th_strict), self.mfNNratio,
                                      int(self.mbCheckOrientation and qa is not None), _p(qa), _p(ta), _p(match), C.byref(n)),
               "dcs_match_filter")
        return match[:nq], n.value

    def match_bf(self, q, q_kp, t, t_kp, th=50):
        q, t = _c(q, np.uint8).reshape(-1, 32), _c(t, np.uint8).reshape(-1, 32)
        q_kp, t_kp = _c(q_kp, KEYPOINT), _c(t_kp, KEYPOINT)
        match = np.full(max(len(q), 1), -1, np.int32)
        n = C.c_int()
        _check(lib().dcs_match_bf(_p(q), _p(q_kp), len(q), _p(t), _p(t_kp), len(t), th, self.mfNNratio,
                                  int(self.mbCheckOrientation), _p(match), C.byref(n)), "dcs_match_bf")
        return match[:len(q)], n.value

    def match_bf_batch_device(self, d_desc, d_kp, d_n, cap, d_pairs, n_pairs, d_match, d_n_matches, d_best, d_second,
                              th=50, stream=None):
        _check(lib().dcs_match_bf_batch_device(d_desc.data_ptr(), d_kp.data_ptr(), d_n.data_ptr(), cap, d_pairs.data_ptr(),
                                               n_pairs, th, self.mfNNratio, int(self.mbCheckOrientation), d_match.data_ptr(),
                                               d_n_matches.data_ptr(), d_best.data_ptr(), d_second.data_ptr(), stream),
               "dcs_match_bf_batch_device")

    @staticmethod
    def _proj_structs(frame, queries):
        keep = []

        def a(x, dt):
            if x is None:
                return None
            arr = _c(x, dt)
            keep.append(arr)
            return _p(arr).value
        f = ProjFrame(len(frame["cam_off"]) - 1, a(frame["cam_off"], np.int32), a(frame["kp_x"], np.float32), a(frame["kp_y"], np.float32),
                      a(frame["kp_octave"], np.int32), a(frame["kp_angle"], np.float32), a(frame["desc"], np.uint8),
                      a(frame.get("taken"), np.uint8), a(frame["min_x"], np.float32), a(frame["min_y"], np.float32),
                      a(frame["grid_w_inv"], np.float32), a(frame["grid_h_inv"], np.float32), a(frame["grid_off"], np.int32),
                      a(frame["grid_idx"], np.int32))
        q = ProjQueries(len(queries["cam"]), a(queries["valid"], np.uint8), a(queries["cam"], np.int32), a(queries["u"], np.float32),
                        a(queries["v"], np.float32), a(queries["radius"], np.float32), a(queries["min_level"], np.int32),
                        a(queries["max_level"], np.int32), a(queries["desc"], np.uint8), a(queries["angle"], np.float32))
        return f, q, keep

    def SearchByProjection(self, frame, queries, th_high=100, use_ratio=True, check_orientation=False):
        """ORBmatcher::SearchByProjection(F, map points, th) (ORBmatcher.cc:539-624; use_ratio) or
        SearchByProjectionOnCam (:954-1113 and :812-951; use_ratio=False, check_orientation=mbCheckOrientation) on flat inputs
        (dicts laid out like dcs_proj_frame / dcs_proj_queries). Returns (match_of_query, query_of_feature, n)."""
        f, q, keep = self._proj_structs(frame, queries)
        N, n = int(frame["cam_off"][-1]), len(queries["cam"])
        mq, qf, nm = np.full(max(n, 1), -1, np.int32), np.full(max(N, 1), -1, np.int32), C.c_int()
        _check(lib().dcs_search_by_projection(C.byref(f), C.byref(q), int(th_high), float(self.mfNNratio) if use_ratio else 0.0,
                                              int(check_orientation), _p(mq), _p(qf), C.byref(nm)), "dcs_search_by_projection")
        return mq[:n], qf[:N], nm.value

    def SearchByProjectionKF(self, frame, queries, th=50):
        """ORBmatcher::SearchByProjection(KF, query, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:416-536, loop closing): the key
        frame's own window (KeyFrame::GetFeaturesInArea incl. its index quirk), octave gate in the loop, best only, sequential
        `taken` chain. Returns (match_of_query, query_of_feature, n)."""
        f, q, keep = self._proj_structs(frame, queries)
        N, n = int(frame["cam_off"][-1]), len(queries["cam"])
        mq, qf, nm = np.full(max(n, 1), -1, np.int32), np.full(max(N, 1), -1, np.int32), C.c_int()
        _check(lib().dcs_search_by_projection_kf(C.byref(f), C.byref(q), int(th), _p(mq), _p(qf), C.byref(nm)), "dcs_search_by_projection_kf")
        return mq[:n], qf[:N], nm.value

    def SearchInWindow(self, frame, queries, th=50, kf_area=True, chi2_inv_sigma2=None):
        """Candidate loops of Fuse x2 (ORBmatcher.cc:1431-1556, 1560-1706), SearchBySim3CrossCam (:1713-1965, one direction per call)
        and SearchByProjection(KF, vpMapPoints, sAlreadyFound, th, ORBdist) (:693-799): independent queries, best only.
        Returns (match_of_query [global feature or -1], best_dist, accepted)."""
        f, q, keep = self._proj_structs(frame, queries)
        n = len(queries["cam"])
        mq, bd, nm = np.full(max(n, 1), -1, np.int32), np.full(max(n, 1), 256, np.int32), C.c_int()
        chi = None if chi2_inv_sigma2 is None else _c(chi2_inv_sigma2, np.float32)
        _check(lib().dcs_search_in_window(C.byref(f), C.byref(q), int(th), int(bool(kf_area)), None if chi is None else _p(chi),
                                          0 if chi is None else len(chi), _p(mq), _p(bd), C.byref(nm)), "dcs_search_in_window")
        return mq[:n], bd[:n], nm.value

    def SearchBySim3(self, frame1, queries_1in2, frame2, queries_2in1, th=100):
        """SearchBySim3CrossCam (ORBmatcher.cc:1713-1965) for one camera pair: queries_1in2 = KF1's map points projected into KF2
        (searched in frame2), queries_2in1 the other way; both indexed by the camera-local feature of their own key frame.
        Returns (match12 [local KF2 feature or -1], nFound) after the agreement check (:1950-1964)."""
        m1, _, _ = self.SearchInWindow(frame2, queries_1in2, th=th, kf_area=True)
        m2, _, _ = self.SearchInWindow(frame1, queries_2in1, th=th, kf_area=True)
        c2 = np.asarray(queries_1in2["cam"], np.int32)
        c1 = np.asarray(queries_2in1["cam"], np.int32)
        off2, off1 = np.asarray(frame2["cam_off"]), np.asarray(frame1["cam_off"])
        match12 = np.full(len(m1), -1, np.int32)
        found = 0
        for i1 in range(len(m1)):
            if m1[i1] < 0:
                continue
            i2 = int(m1[i1] - off2[c2[i1]])                 # vnMatch1[i1] = bestIdx2local
            if i2 < len(m2) and m2[i2] >= 0 and int(m2[i2] - off1[c1[i2]]) == i1:
                match12[i1] = i2
                found += 1
        return match12, found

    def SearchForInitialization(self, frame2, queries, check_orientation=None):
        """ORBmatcher::SearchForInitialization (ORBmatcher.cc:1117-1251) on flat inputs (see dcs_search_for_initialization).
        Returns (vnMatches12 [global F2 feature or -1], nmatches)."""
        f, q, keep = self._proj_structs(frame2, queries)
        n = len(queries["cam"])
        m12, nm = np.full(max(n, 1), -1, np.int32), C.c_int()
        co = self.mbCheckOrientation if check_orientation is None else check_orientation
        _check(lib().dcs_search_for_initialization(C.byref(f), C.byref(q), float(self.mfNNratio), int(co), _p(m12), C.byref(nm)),
               "dcs_search_for_initialization")
        return m12[:n], nm.value

    def SearchByBoWCrossCam(self, desc_kf, ang_kf, kf_valid, desc_f, ang_f, kf_fv, f_fv):
        """SearchByBoWCrossCam(F,cF,KF,cKF) (ORBmatcher.cc:162-294) on flat inputs; returns (match_f, n)."""
        desc_kf, desc_f = _c(desc_kf, np.uint8).reshape(-1, 32), _c(desc_f, np.uint8).reshape(-1, 32)
        ang_kf, ang_f, kf_valid = _c(ang_kf, np.float32), _c(ang_f, np.float32), _c(kf_valid, np.uint8)
        kn, ko, ki = (_c(a, np.int32) for a in kf_fv)
        fn, fo, fi = (_c(a, np.int32) for a in f_fv)
        match = np.full(max(len(desc_f), 1), -1, np.int32)
        n = C.c_int()
        _check(lib().dcs_search_by_bow(_p(desc_kf), _p(ang_kf), _p(kf_valid), len(desc_kf), _p(desc_f), _p(ang_f), len(desc_f),
                                       _p(kn), _p(ko), _p(ki), len(kn), _p(fn), _p(fo), _p(fi), len(fn),
                                       self.mfNNratio, int(self.mbCheckOrientation), _p(match), C.byref(n)), "dcs_search_by_bow")
        return match[:len(desc_f)], n.value


class FeatureComm:
    """dcs_comm: the RCCL communicator of the cross-GPU feature exchange (one per process / GPU)."""

    @staticmethod
    def unique_id():
        buf = np.zeros(128, np.uint8)
        _check(lib().dcs_comm_unique_id(_p(buf)), "dcs_comm_unique_id")
        return buf

    def __init__(self, uid, rank, world):
        uid = _c(uid, np.uint8)
        self._h = C.c_void_p()
        _check(lib().dcs_comm_create(_p(uid), int(rank), int(world), C.byref(self._h)), "dcs_comm_create")
        self.rank, self.world = int(rank), int(world)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().dcs_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allgather_features(self, d_kp, d_desc, d_n, cap, g_kp, g_desc, g_n, stream=None):
        """torch CUDA tensors: kp [S, cap, 7] f32, desc [S, cap, 32] u8, n [S] i32 -> g_* [world * S, ...] (rank-major)."""
        S = int(d_kp.shape[0])
        _check(lib().dcs_features_allgather(self._h, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), S, int(cap),
                                            g_kp.data_ptr(), g_desc.data_ptr(), g_n.data_ptr(), stream), "dcs_features_allgather")


def SearchByBoWCrossCamKF(desc1, ang1, valid1, desc2, ang2, valid2, fv1, fv2, ratio=0.75, check_ori=True):
    """ORBmatcher::SearchByBoWCrossCam(KF1, c1, KF2, c2, vpMatches12) (ORBmatcher.cc:297-414) -> (match12[n1], nmatches)"""
    desc1, desc2 = _c(desc1, np.uint8).reshape(-1, 32), _c(desc2, np.uint8).reshape(-1, 32)
    ang1, ang2, valid1, valid2 = _c(ang1, np.float32), _c(ang2, np.float32), _c(valid1, np.uint8), _c(valid2, np.uint8)
    a1, a2 = [_c(a, np.int32) for a in fv1], [_c(a, np.int32) for a in fv2]
    match, n = np.full(max(len(desc1), 1), -1, np.int32), C.c_int(0)
    _check(lib().dcs_search_by_bow_kf(_p(desc1), _p(ang1), _p(valid1), len(desc1), _p(desc2), _p(ang2), _p(valid2), len(desc2),
                                      _p(a1[0]), _p(a1[1]), _p(a1[2]), len(a1[0]), _p(a2[0]), _p(a2[1]), _p(a2[2]), len(a2[0]),
                                      float(ratio), int(check_ori), _p(match), C.byref(n)), "dcs_search_by_bow_kf")
    return match[:len(desc1)], n.value


def SearchForTriangulation(desc1, ang1, free1, desc2, ang2, free2, fv1, fv2, epi, check_ori=True):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:1253-1427) for one camera; epi as in oracle.search_for_triangulation"""
    desc1, desc2 = _c(desc1, np.uint8).reshape(-1, 32), _c(desc2, np.uint8).reshape(-1, 32)
    ang1, ang2, free1, free2 = _c(ang1, np.float32), _c(ang2, np.float32), _c(free1, np.uint8), _c(free2, np.uint8)
    a1, a2 = [_c(a, np.int32) for a in fv1], [_c(a, np.int32) for a in fv2]
    keep = {k: _c(epi[k], np.float32) for k in ("kp1_x", "kp1_y", "kp2_x", "kp2_y", "level_sigma2", "scale_factors")}
    keep["kp2_octave"] = _c(epi["kp2_octave"], np.int32)
    e = Epipolar()
    for i, v in enumerate(np.asarray(epi["F12"], np.float32).reshape(9)):
        e.F12[i] = float(v)
    e.ex, e.ey = float(np.float32(epi["ex"])), float(np.float32(epi["ey"]))
    for k in ("kp1_x", "kp1_y", "kp2_x", "kp2_y", "kp2_octave", "level_sigma2", "scale_factors"):
        setattr(e, k, keep[k].ctypes.data)
    e.n_levels = len(keep["scale_factors"])
    match, n = np.full(max(len(desc1), 1), -1, np.int32), C.c_int(0)
    _check(lib().dcs_search_for_triangulation(_p(desc1), _p(ang1), _p(free1), len(desc1), _p(desc2), _p(ang2), _p(free2), len(desc2),
                                              _p(a1[0]), _p(a1[1]), _p(a1[2]), len(a1[0]), _p(a2[0]), _p(a2[1]), _p(a2[2]), len(a2[0]),
                                              C.byref(e), int(check_ori), _p(match), C.byref(n)), "dcs_search_for_triangulation")
    return match[:len(desc1)], n.value


def frame_grid(cam_off, kp_x, kp_y, min_x, min_y, w_inv, h_inv):
    """Frame::PosInGrid + grid fill (Frame.cc:180-196, 380-390) as CSR (grid_off, grid_idx)."""
    cam_off = _c(cam_off, np.int32)
    n_cams, N = len(cam_off) - 1, int(cam_off[-1])
    kx, ky = _c(kp_x, np.float32), _c(kp_y, np.float32)
    mx, my, wi, hi = (_c(v, np.float32) for v in (min_x, min_y, w_inv, h_inv))
    off, idx, n = np.zeros(n_cams * 64 * 48 + 1, np.int32), np.zeros(max(N, 1), np.int32), C.c_int()
    _check(lib().dcs_frame_grid(n_cams, _p(cam_off), _p(kx), _p(ky), _p(mx), _p(my), _p(wi), _p(hi), _p(off), _p(idx), C.byref(n)),
           "dcs_frame_grid")
    return off, idx[:n.value]


def ComputeDistinctiveDescriptors(pool, off, idx):
    """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:270-340) for a batch of map points (CSR lists into a
    descriptor pool); returns the position of the chosen descriptor inside each list (-1 for empty lists)."""
    pool = _c(pool, np.uint8).reshape(-1, 32)
    off, idx = _c(off, np.int32), _c(idx, np.int32)
    best = np.zeros(max(len(off) - 1, 1), np.int32)
    _check(lib().dcs_distinctive_descriptors(_p(pool), len(pool), _p(off), _p(idx), len(off) - 1, _p(best)), "dcs_distinctive_descriptors")
    return best[:len(off) - 1]


def rig_adjoint(T44_f32, exact=False):
    T = _c(T44_f32, np.float32).reshape(16)
    ext, adj = np.zeros(7), np.zeros(36)
    _check(lib().dcs_rig_adjoint(_p(T), int(exact), _p(ext), _p(adj)), "dcs_rig_adjoint")
    return adj.reshape(6, 6), ext


def pose_from_matrix(T44_f32):
    T = _c(T44_f32, np.float32).reshape(16)
    out = np.zeros(7)
    _check(lib().dcs_pose_from_matrix(_p(T), _p(out)), "dcs_pose_from_matrix")
    return out


def pose_to_matrix(pose7):
    p = _c(pose7, np.float64)
    out = np.zeros(16, np.float32)
    _check(lib().dcs_pose_to_matrix(_p(p), _p(out)), "dcs_pose_to_matrix")
    return out.reshape(4, 4)


def make_camera(fx, fy, cx, cy, ext7, adj36):
    c = BaCamera()
    c.fx, c.fy, c.cx, c.cy = float(fx), float(fy), float(cx), float(cy)
    for i in range(7):
        c.ext[i] = float(ext7[i])
    a = np.asarray(adj36, np.float64).reshape(-1)
    for i in range(36):
        c.adj[i] = float(a[i])
    return c


class ProjFrame(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("cam_off", C.c_void_p), ("kp_x", C.c_void_p), ("kp_y", C.c_void_p),
                ("kp_octave", C.c_void_p), ("kp_angle", C.c_void_p), ("desc", C.c_void_p), ("taken", C.c_void_p),
                ("min_x", C.c_void_p), ("min_y", C.c_void_p), ("grid_w_inv", C.c_void_p), ("grid_h_inv", C.c_void_p),
                ("grid_off", C.c_void_p), ("grid_idx", C.c_void_p)]


class ProjQueries(C.Structure):
    _fields_ = [("n", C.c_int32), ("valid", C.c_void_p), ("cam", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p),
                ("radius", C.c_void_p), ("min_level", C.c_void_p), ("max_level", C.c_void_p), ("desc", C.c_void_p),
                ("angle", C.c_void_p)]


class PoseProblem(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_cams", C.c_int32), ("poses", C.c_void_p), ("edge_off", C.c_void_p),
                ("xw", C.c_void_p), ("obs", C.c_void_p), ("inv_sigma2", C.c_void_p), ("edge_cam", C.c_void_p),
                ("cams", C.c_void_p), ("huber_delta", C.c_double), ("chi2_th", C.c_float * 4), ("its", C.c_int32 * 4)]


class PoseResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("outlier", C.c_void_p), ("n_inliers", C.c_void_p), ("edge_chi2", C.c_void_p),
                ("n_iters", C.c_void_p)]


class PreparedBA:
    """The flat problem marshalled once (contiguous arrays + dcs_ba_problem / dcs_ba_result structs); solve() is then a
    bare dcs_ba_local call, like the C++ caller in INTEGRATION.md section 3."""

    def __init__(self, prob):
        self.poses = _c(prob["poses"], np.float64)
        self.fixed = _c(prob["pose_fixed"], np.uint8)
        self.points = _c(prob["points"], np.float64)
        self.ep, self.el, self.ec = (_c(prob[k], np.int32) for k in ("edge_pose", "edge_point", "edge_cam"))
        self.obs = _c(prob["obs"], np.float64)
        self.w = _c(prob["inv_sigma2"], np.float64)
        cam_list = [c if isinstance(c, BaCamera) else make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"])
                    for c in prob["cams"]]
        self.cams = (BaCamera * len(cam_list))(*cam_list)
        P, L, E = len(self.poses), len(self.points), len(self.ep)
        self.pb = BaProblem(P, L, E, len(cam_list), _p(self.poses).value, _p(self.fixed).value, _p(self.points).value,
                            _p(self.ep).value, _p(self.el).value, _p(self.ec).value, _p(self.obs).value, _p(self.w).value,
                            C.cast(self.cams, C.c_void_p).value,
                            float(prob.get("huber_delta", np.sqrt(5.991))), float(prob.get("chi2_th", 5.991)),
                            int(prob.get("iters1", 5)), int(prob.get("iters2", 10)))
        self.out_poses, self.out_points = np.zeros((P, 7)), np.zeros((L, 3))
        self.chi2, self.outl, self.lvl1 = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8)
        self.res = BaResult(_p(self.out_poses).value, _p(self.out_points).value, _p(self.chi2).value, _p(self.outl).value,
                            _p(self.lvl1).value)

    def solve(self, stop_flag=None):
        sf = _p(stop_flag) if stop_flag is not None else None
        _check(lib().dcs_ba_local(C.byref(self.pb), sf, C.byref(self.res)), "dcs_ba_local")
        return self.result()

    def linearize(self):
        """dcs_ba_debug_linearize: the H / b blocks after the first linearisation (no lambda, no solve)"""
        P, L, E = len(self.poses), len(self.points), len(self.ep)
        Hpp, bp, Hll, bl, Hpl = np.zeros((P, 6, 6)), np.zeros((P, 6)), np.zeros((L, 3, 3)), np.zeros((L, 3)), np.zeros((E, 6, 3))
        pose_idx, n_free = np.zeros(P, np.int32), C.c_int()
        _check(lib().dcs_ba_debug_linearize(C.byref(self.pb), _p(Hpp), _p(bp), _p(Hll), _p(bl), _p(Hpl), _p(pose_idx), C.byref(n_free)), "dcs_ba_debug_linearize")
        n = n_free.value
        return dict(Hpp=Hpp[:n], bp=bp[:n], Hll=Hll, bl=bl, Hpl=Hpl, pose_idx=pose_idx, n_free=n)

    def result(self):
        res = self.res
        return dict(poses=self.out_poses, points=self.out_points, edge_chi2=self.chi2, edge_outlier=self.outl, edge_level1=self.lvl1,
                    n_iters=list(res.n_iters), n_trials=list(res.n_trials), lambda_=list(res.lambda_),
                    chi2_trace=np.array(res.chi2_trace), gpu_ms=float(res.gpu_ms))


class Optimizer:
    """Optimizer::LocalBundleAdjustment (reference include/Optimizer.h:55) on a flat problem."""

    @staticmethod
    def prepare(prob):
        return PreparedBA(prob)

    @staticmethod
    def PoseOptimization(prob):
        """Optimizer::PoseOptimization (Optimizer.cc:250-405) for a batch of frames (flat problem, see dcs_pose_problem)."""
        poses = _c(prob["poses"], np.float64)
        off, cam = _c(prob["edge_off"], np.int32), _c(prob["edge_cam"], np.int32)
        xw, obs, w = (_c(prob[k], np.float64) for k in ("xw", "obs", "inv_sigma2"))
        cam_list = [c if isinstance(c, BaCamera) else make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"])
                    for c in prob["cams"]]
        cams = (BaCamera * len(cam_list))(*cam_list)
        F, E = len(poses), len(cam)
        pb = PoseProblem(F, len(cam_list), _p(poses).value, _p(off).value, _p(xw).value, _p(obs).value, _p(w).value,
                         _p(cam).value, C.cast(cams, C.c_void_p).value, float(prob["huber_delta"]),
                         (C.c_float * 4)(*prob["chi2_th"]), (C.c_int32 * 4)(*prob["its"]))
        out_poses, outl, ninl = np.zeros((F, 7)), np.zeros(max(E, 1), np.uint8), np.zeros(F, np.int32)
        chi2, nit = np.zeros(max(E, 1)), np.zeros((F, 4), np.int32)
        res = PoseResult(_p(out_poses).value, _p(outl).value, _p(ninl).value, _p(chi2).value, _p(nit).value)
        _check(lib().dcs_pose_optimization(C.byref(pb), C.byref(res)), "dcs_pose_optimization")
        return dict(poses=out_poses, outlier=outl[:E], n_inliers=ninl, edge_chi2=chi2[:E], n_iters=nit)

    @staticmethod
    def BundleAdjustment(prob, nIterations=5, bRobust=True, stop_flag=None):
        """Optimizer::BundleAdjustment (Optimizer.cc:70-248): one LM round of nIterations, Huber delta sqrt(3.99) when
        bRobust; `prob` is the flat problem with only fixId marked fixed."""
        p = dict(prob)
        p["iters1"], p["iters2"] = int(nIterations), 0
        p["huber_delta"] = float(np.float32(np.sqrt(3.99))) if bRobust else 0.0     # const float thHuber2D (:107)
        return PreparedBA(p).solve(stop_flag)

    @staticmethod
    def timing(on):
        """dcs_ba_timing: returns {ldlt_us, ldlt_launches, step_us, steps} accumulated since the last call and
        switches the event timing of this thread's BA calls on / off."""
        out = np.zeros(4)
        _check(lib().dcs_ba_timing(int(bool(on)), _p(out)), "dcs_ba_timing")
        return dict(ldlt_us=out[0], ldlt_launches=out[1], step_us=out[2], steps=out[3])

    @staticmethod
    def LocalBundleAdjustmentBatch(probs, stop_flags=None):
        """dcs_ba_local_batch: one Optimizer::LocalBundleAdjustment per dual-camera stream (BASELINE config C5,
        src/LocalMapping.cc:97-104), all problems in one call. `probs`: flat problems or PreparedBA objects;
        `stop_flags`: None or a list of (uint8 array | None). Returns one result dict per problem."""
        preps = [p if isinstance(p, PreparedBA) else PreparedBA(p) for p in probs]
        n = len(preps)
        pbs = (C.c_void_p * max(n, 1))(*[C.addressof(p.pb) for p in preps])
        ress = (C.c_void_p * max(n, 1))(*[C.addressof(p.res) for p in preps])
        sfs = None
        if stop_flags is not None:
            sfs = (C.c_void_p * max(n, 1))(*[(_p(f).value if f is not None else None) for f in stop_flags])
        _check(lib().dcs_ba_local_batch(n, C.cast(pbs, C.c_void_p), C.cast(sfs, C.c_void_p) if sfs is not None else None,
                                        C.cast(ress, C.c_void_p)), "dcs_ba_local_batch")
        outs = []
        for p in preps:
            out = p.result()
            outs.append({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in out.items()})
        return outs

    @staticmethod
    def LocalBundleAdjustment(prob, stop_flag=None):
        prep = prob if isinstance(prob, PreparedBA) else PreparedBA(prob)
        out = prep.solve(stop_flag)
        if prep is not prob:                               # one-shot call: the caller owns the outputs
            return out
        return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in out.items()}


class ORBVocabulary:
    """ORB_SLAM2::ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (include/ORBVocabulary.h:31-32) with the
    tree resident in HBM: transform (Frame::ComputeBoW, src/Frame.cc:393-406) and score (KeyFrameDatabase.cc:250-372)."""
    L1_NORM, TF_IDF = 0, 0

    def __init__(self, k, L, parent, is_leaf, desc, weight, scoring=0, weighting=0):
        parent, is_leaf = _c(parent, np.int32), _c(is_leaf, np.uint8)
        desc, weight = _c(desc, np.uint8).reshape(-1, 32), _c(weight, np.float64)
        if not (len(parent) == len(is_leaf) == len(desc) == len(weight)):
            raise ValueError("vocabulary columns differ in length")
        self._h = C.c_void_p()
        _check(lib().dcs_vocab_create(k, L, scoring, weighting, len(parent), _p(parent), _p(is_leaf), _p(desc), _p(weight), C.byref(self._h)),
               "dcs_vocab_create")

    @classmethod
    def loadFromTextFile(cls, path):
        from . import synth
        v = synth.vocabulary_from_text(path)
        return cls(v["k"], v["L"], v["parent"], v["is_leaf"], v["desc"], v["weight"], v["scoring"], v["weighting"])

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.dcs_vocab_destroy(self._h)
            self._h = None

    __del__ = close

    def info(self):
        k, L, nn, nw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(lib().dcs_vocab_info(self._h, C.byref(k), C.byref(L), C.byref(nn), C.byref(nw)), "dcs_vocab_info")
        return dict(k=k.value, L=L.value, n_nodes=nn.value, n_words=nw.value)

    def transform(self, desc, levelsup=4):
        """one image -> dict(word[n], node[n], bow_word, bow_val, fv_node, fv_off, fv_idx)"""
        desc = _c(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        m = max(n, 1)
        word, node, bw, fn, fi = (np.zeros(m, np.int32) for _ in range(5))
        fo = np.zeros(m + 1, np.int32)
        bv = np.zeros(m)
        nw, nn = C.c_int(), C.c_int()
        _check(lib().dcs_bow_transform(self._h, _p(desc), n, levelsup, _p(word), _p(node), _p(bw), _p(bv), C.byref(nw), _p(fn), _p(fo), _p(fi),
                                       C.byref(nn)), "dcs_bow_transform")
        return dict(word=word[:n], node=node[:n], bow_word=bw[:nw.value].copy(), bow_val=bv[:nw.value].copy(), fv_node=fn[:nn.value].copy(),
                    fv_off=fo[:nn.value + 1].copy(), fv_idx=fi[:fo[nn.value]].copy())

    def transform_device(self, d_desc, d_n, cap, out, levelsup=4, stream=None):
        """d_desc torch uint8 [n_images, cap, 32], d_n int32 [n_images]; `out` = dict of preallocated torch buffers
        (see bow_buffers)."""
        n_images = d_desc.shape[0]
        _check(lib().dcs_bow_transform_device(self._h, d_desc.data_ptr(), d_n.data_ptr(), n_images, cap, levelsup, out["word"].data_ptr(),
                                              out["node"].data_ptr(), out["weight"].data_ptr(), out["bow_word"].data_ptr(), out["bow_val"].data_ptr(),
                                              out["bow_n"].data_ptr(), out["fv_node"].data_ptr(), out["fv_off"].data_ptr(), out["fv_idx"].data_ptr(),
                                              out["fv_n"].data_ptr(), stream), "dcs_bow_transform_device")

    @staticmethod
    def bow_buffers(n_images, cap, device="cuda"):
        import torch
        i32 = dict(dtype=torch.int32, device=device)
        f64 = dict(dtype=torch.float64, device=device)
        return dict(word=torch.zeros((n_images, cap), **i32), node=torch.zeros((n_images, cap), **i32), weight=torch.zeros((n_images, cap), **f64),
                    bow_word=torch.zeros((n_images, cap), **i32), bow_val=torch.zeros((n_images, cap), **f64), bow_n=torch.zeros(n_images, **i32),
                    fv_node=torch.zeros((n_images, cap), **i32), fv_off=torch.zeros((n_images, cap + 1), **i32),
                    fv_idx=torch.zeros((n_images, cap), **i32), fv_n=torch.zeros(n_images, **i32))

    @staticmethod
    def score(q_word, q_val, db_off, db_word, db_val):
        """L1 score of one BowVector against a CSR database of BowVectors -> float64 [n_db]"""
        q_word, q_val = _c(q_word, np.int32), _c(q_val, np.float64)
        db_off, db_word, db_val = _c(db_off, np.int32), _c(db_word, np.int32), _c(db_val, np.float64)
        n_db = len(db_off) - 1
        score = np.zeros(max(n_db, 1))
        _check(lib().dcs_bow_score_l1(_p(q_word), _p(q_val), len(q_word), _p(db_off), _p(db_word), _p(db_val), n_db, _p(score)), "dcs_bow_score_l1")
        return score[:n_db]


class KeyFrameDatabase:
    """KeyFrameDatabase of ONE camera (src/KeyFrameDatabase.cc) over dcs_kfdb_*: add / erase / clear and the two candidate
    searches on flat inputs. Per-entry state the reference keeps in KeyFrame members (mnLoopQuery, mnLoopWords, mLoopScore,
    mnRelocQuery, mnRelocWords, mRelocScore) lives in arrays here and persists across queries like the members do."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(lib().dcs_kfdb_create(C.byref(self._h)), "dcs_kfdb_create")
        # q_id / words / score = mnLoopQuery|mnRelocQuery, mnLoopWords|mnRelocWords, mLoopScore|mRelocScore of the entries (one search
        # kind per instance; databases of several cameras may share these arrays: assign the same objects)
        self.q_id = np.zeros(0, np.int64); self.words = np.zeros(0, np.int32); self.score = np.zeros(0, np.float32)

    def close(self):
        if self._h:
            lib().dcs_kfdb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = C.c_int()
        _check(lib().dcs_kfdb_size(self._h, C.byref(n)), "dcs_kfdb_size")
        return n.value

    def add(self, word, val):
        word, val = _c(word, np.int32), _c(val, np.float64)
        e = C.c_int()
        _check(lib().dcs_kfdb_add(self._h, _p(word), _p(val), len(word), C.byref(e)), "dcs_kfdb_add")
        self.q_id = np.append(self.q_id, -1); self.words = np.append(self.words, 0).astype(np.int32); self.score = np.append(self.score, 0).astype(np.float32)
        return e.value

    def erase(self, entry):
        _check(lib().dcs_kfdb_erase(self._h, int(entry)), "dcs_kfdb_erase")

    def clear(self):
        _check(lib().dcs_kfdb_clear(self._h), "dcs_kfdb_clear")
        self.q_id = np.zeros(0, np.int64); self.words = np.zeros(0, np.int32); self.score = np.zeros(0, np.float32)

    def query(self, q_word, q_val):
        q_word, q_val = _c(q_word, np.int32), _c(q_val, np.float64)
        n = len(self)
        common, first, score = np.zeros(max(n, 1), np.int32), np.full(max(n, 1), -1, np.int32), np.zeros(max(n, 1), np.float32)
        _check(lib().dcs_kfdb_query(self._h, _p(q_word), _p(q_val), len(q_word), _p(common), _p(first), _p(score)), "dcs_kfdb_query")
        return common[:n], first[:n], score[:n]

    def _detect(self, query_id, q_word, q_val, covis, connected, min_score, loop):
        common, first, score = self.query(q_word, q_val)
        n = len(common)
        # the walk over the inverted files (:128-149, :257-272) leaves, per entry that shares a word: its place in lKFsSharingWords
        # = (first shared word, entry id) order, and the members. An entry that already carries this query's id (the same key frame
        # queried again for another camera pair: the members are per key frame, not per camera) is NOT listed again and its word
        # count keeps growing; a connected key frame (loop search) never gets the id, so its count restarts at every word: 1.
        sharing = []
        for k in np.lexsort((np.arange(n), first)):
            if common[k] <= 0:
                continue
            if self.q_id[k] != query_id:
                if loop and connected[k]:
                    self.words[k] = 1
                else:
                    self.q_id[k] = query_id; self.words[k] = common[k]; sharing.append(int(k))
            else:
                self.words[k] += common[k]
        if not sharing:
            return []
        max_common = max(int(self.words[k]) for k in sharing)
        min_common = int(np.float32(max_common) * np.float32(0.8))
        score_and_match = []
        for k in sharing:
            if self.words[k] > min_common:
                self.score[k] = score[k]
                if not loop or score[k] >= np.float32(min_score):
                    score_and_match.append((np.float32(score[k]), k))
        if not score_and_match:
            return []
        acc_and_match, best_acc = [], (np.float32(min_score) if loop else np.float32(0))
        for si, k in score_and_match:
            best, acc, best_k = si, si, k
            for k2 in covis[k]:
                if self.q_id[k2] != query_id or (loop and not self.words[k2] > min_common):
                    continue
                acc = np.float32(acc + self.score[k2])
                if self.score[k2] > best:
                    best_k, best = k2, self.score[k2]
            acc_and_match.append((acc, best_k))
            if acc > best_acc:
                best_acc = acc
        keep = np.float32(np.float32(0.75) * best_acc)
        out = []
        for acc, k in acc_and_match:
            if acc > keep and k not in out:
                out.append(int(k))
        return out

    def DetectRelocalizationCandidates(self, frame_id, q_word, q_val, covis):
        """src/KeyFrameDatabase.cc:237-372; covis[k] = GetBestCovisibilityKeyFrames(10) of entry k as entry ids"""
        return self._detect(frame_id, q_word, q_val, covis, None, 0.0, False)

    def DetectLoopCandidates(self, kf_id, q_word, q_val, covis, connected, min_score):
        """DetectLoopCandidatesForCam (:111-235); connected[k] = entry k is in pKF->GetConnectedKeyFrames()"""
        return self._detect(kf_id, q_word, q_val, covis, connected, min_score, True)


def isInFrustum(frame, pts, viewing_cos_limit=0.5, th=1.0):
    """Frame::isInFrustum (Frame.cc:244-312) for a batch of map points + PredictScale + the SearchByProjection window.
    frame: dict(Rsw[c,9], tsw[c,3], Ow[c,3], fx, fy, cx, cy, min_x, max_x, min_y, max_y [c], log_scale_factor, scale_factors[L]);
    pts: dict(pos[n,3], normal[n,3], min_dist[n], max_dist[n], candidate[n] or None). Returns the per-point outputs; the columns
    valid / cam / u / v / radius / min_level / max_level of the projection queries follow directly (see projection_queries)."""
    a = {k: _c(frame[k], np.float32) for k in ("Rsw", "tsw", "Ow", "fx", "fy", "cx", "cy", "min_x", "max_x", "min_y", "max_y", "scale_factors")}
    f = FrustumFrame(len(a["fx"]), *[a[k].ctypes.data for k in ("Rsw", "tsw", "Ow", "fx", "fy", "cx", "cy", "min_x", "max_x", "min_y", "max_y")],
                     float(np.float32(frame["log_scale_factor"])), len(a["scale_factors"]), a["scale_factors"].ctypes.data)
    pos, nrm = _c(pts["pos"], np.float32).reshape(-1, 3), _c(pts["normal"], np.float32).reshape(-1, 3)
    mind, maxd = _c(pts["min_dist"], np.float32), _c(pts["max_dist"], np.float32)
    cand = _c(pts["candidate"], np.uint8) if pts.get("candidate") is not None else None
    n = len(pos)
    m = max(n, 1)
    out = dict(in_view=np.zeros(m, np.uint8), cam=np.zeros(m, np.int32), u=np.zeros(m, np.float32), v=np.zeros(m, np.float32),
               view_cos=np.zeros(m, np.float32), level=np.zeros(m, np.int32), radius=np.zeros(m, np.float32))
    _check(lib().dcs_is_in_frustum(C.byref(f), n, _p(pos), _p(nrm), _p(mind), _p(maxd), _p(cand), float(viewing_cos_limit), float(th),
                                   *[_p(out[k]) for k in ("in_view", "cam", "u", "v", "view_cos", "level", "radius")]), "dcs_is_in_frustum")
    return {k: v[:n] for k, v in out.items()}


def projection_queries(fr, desc, angle=None):
    """isInFrustum outputs + the map points' descriptors -> the query dict of ORBmatcher.SearchByProjection (ORBmatcher.cc:557-565)."""
    n = len(fr["in_view"])
    return dict(valid=fr["in_view"], cam=np.maximum(fr["cam"], 0).astype(np.int32), u=fr["u"], v=fr["v"], radius=fr["radius"],
                min_level=(fr["level"] - 1).astype(np.int32), max_level=(fr["level"] + 1).astype(np.int32), desc=_c(desc, np.uint8).reshape(-1, 32),
                angle=_c(angle, np.float32) if angle is not None else np.zeros(n, np.float32))


def cu_range_stream(first_cu, n_cus):
    """dcs_stream_create_cu_range: raw hipStream_t (int) restricted to CUs [first_cu, first_cu + n_cus) of the CU-mask order"""
    h = C.c_void_p()
    _check(lib().dcs_stream_create_cu_range(int(first_cu), int(n_cus), C.byref(h)), "dcs_stream_create_cu_range")
    return h.value


def streams_share_queue(a, b):
    """dcs_streams_share_queue: does work on raw stream b wait for work on raw stream a (one hardware queue)? a / b: hipStream_t as int (0 = legacy default stream)"""
    sh = C.c_int(0)
    _check(lib().dcs_streams_share_queue(C.c_void_p(a or None), C.c_void_p(b or None), C.byref(sh)), "dcs_streams_share_queue")
    return bool(sh.value)


def stream_apart(avoid):
    """dcs_stream_create_apart: (raw non-blocking hipStream_t as int, apart?) on another hardware queue than the raw streams in `avoid`"""
    arr = (C.c_void_p * max(1, len(avoid)))(*[C.c_void_p(a or None) for a in avoid])
    h, ok = C.c_void_p(), C.c_int(0)
    _check(lib().dcs_stream_create_apart(C.cast(arr, C.c_void_p), len(avoid), C.byref(h), C.byref(ok)), "dcs_stream_create_apart")
    return h.value, bool(ok.value)


def ba_avoid_streams(avoid):
    """dcs_ba_avoid_streams: solver streams created from now on keep off the hardware queues of these raw streams ([] clears)"""
    arr = (C.c_void_p * max(1, len(avoid)))(*[C.c_void_p(a or None) for a in avoid])
    _check(lib().dcs_ba_avoid_streams(C.cast(arr, C.c_void_p), len(avoid)), "dcs_ba_avoid_streams")


class HostFrames:
    """dcs_host_alloc: n page-locked frames of rows x cols bytes at a 4-byte aligned stride, as numpy views (frames[i] is [rows, cols]);
    dcs_orb_extract_batch reads such frames in place (no staging copy). close() frees the block."""

    def __init__(self, n, rows, cols):
        self.n, self.rows, self.cols, self.stride = n, rows, cols, (cols + 3) & ~3
        h = C.c_void_p()
        _check(lib().dcs_host_alloc(C.byref(h), n * rows * self.stride), "dcs_host_alloc")
        self._p = h.value
        buf = (C.c_uint8 * (n * rows * self.stride)).from_address(self._p)
        self._block = np.frombuffer(buf, np.uint8).reshape(n, rows, self.stride)
        self.frames = [self._block[i, :, :cols] for i in range(n)]

    def close(self):
        if self._p:
            self.frames, self._block = None, None
            _check(lib().dcs_host_free(self._p), "dcs_host_free")
            self._p = None


def ba_set_cu_range(first_cu, n_cus):
    _check(lib().dcs_ba_set_cu_range(int(first_cu), int(n_cus)), "dcs_ba_set_cu_range")


def ba_release_thread():
    _check(lib().dcs_ba_release_thread(), "dcs_ba_release_thread")


class TrackFrame(C.Structure):
    _fields_ = [("features", ProjFrame), ("has_point", C.c_void_p), ("point_xw", C.c_void_p), ("view", FrustumFrame), ("pose", C.c_void_p),
                ("n_points", C.c_int32), ("pos", C.c_void_p), ("normal", C.c_void_p), ("min_dist", C.c_void_p), ("max_dist", C.c_void_p),
                ("candidate", C.c_void_p), ("desc", C.c_void_p)]


class TrackParams(C.Structure):
    _fields_ = [("viewing_cos_limit", C.c_float), ("th", C.c_float), ("th_high", C.c_int32), ("nn_ratio", C.c_float), ("n_levels", C.c_int32),
                ("inv_level_sigma2", C.c_void_p), ("n_cams", C.c_int32), ("cams", C.c_void_p), ("huber_delta", C.c_double),
                ("chi2_th", C.c_float * 4), ("its", C.c_int32 * 4)]


class TrackResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("n_inliers", C.c_void_p), ("n_matches", C.c_void_p), ("match_of_point", C.c_void_p),
                ("point_of_feature", C.c_void_p), ("outlier", C.c_void_p)]


class PreparedTracking:
    """dcs_track_local_map on a batch of frames laid out like synth.tracking_problem (every frame's features carry their grid): the flat
    problem marshalled once, track() is then the bare C call."""

    def __init__(self, frames, params, null_empty=False):
        """null_empty: empty arrays go over as NULL pointers (what a C caller with an empty local map passes)"""
        self.keep = []

        def a(x, dt):
            if x is None or (null_empty and np.size(x) == 0):
                return None
            arr = _c(x, dt)
            self.keep.append(arr)
            return _p(arr).value
        F = len(frames)
        self.F = F
        arr = (TrackFrame * max(F, 1))()
        self.N, self.np_ = [], []
        for k, fr in enumerate(frames):
            ft, vw, pt = fr["features"], fr["view"], fr["points"]
            t = arr[k]
            t.features = ProjFrame(len(ft["cam_off"]) - 1, a(ft["cam_off"], np.int32), a(ft["kp_x"], np.float32), a(ft["kp_y"], np.float32),
                                   a(ft["kp_octave"], np.int32), a(ft.get("kp_angle"), np.float32), a(ft["desc"], np.uint8), a(ft["taken"], np.uint8),
                                   a(ft["min_x"], np.float32), a(ft["min_y"], np.float32), a(ft["grid_w_inv"], np.float32), a(ft["grid_h_inv"], np.float32),
                                   a(ft["grid_off"], np.int32), a(ft["grid_idx"], np.int32))
            t.has_point = a(fr.get("has_point"), np.uint8)
            t.point_xw = a(fr["point_xw"], np.float32)
            v = FrustumFrame()
            v.n_cams = len(vw["fx"])
            for key in ("Rsw", "tsw", "Ow", "fx", "fy", "cx", "cy", "min_x", "max_x", "min_y", "max_y"):
                setattr(v, key, a(vw[key], np.float32))
            v.log_scale_factor = float(vw["log_scale_factor"]); v.n_scale_levels = len(vw["scale_factors"]); v.scale_factors = a(vw["scale_factors"], np.float32)
            t.view = v
            t.pose = a(fr["pose"], np.float64)
            t.n_points = len(pt["pos"])
            t.pos, t.normal = a(pt["pos"], np.float32), a(pt["normal"], np.float32)
            t.min_dist, t.max_dist = a(pt["min_dist"], np.float32), a(pt["max_dist"], np.float32)
            t.candidate = a(pt.get("candidate"), np.uint8)
            t.desc = a(fr["desc"], np.uint8)
            self.N.append(int(ft["cam_off"][-1])); self.np_.append(len(pt["pos"]))
        self.frames = arr
        cam_list = [c if isinstance(c, BaCamera) else make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in params["cams"]]
        self.cams = (BaCamera * len(cam_list))(*cam_list)
        sig = a(params["inv_level_sigma2"], np.float32)
        self.prm = TrackParams(float(params["viewing_cos_limit"]), float(params["th"]), int(params["th_high"]), float(params["nn_ratio"]),
                               len(params["inv_level_sigma2"]), sig, len(cam_list), C.cast(self.cams, C.c_void_p).value, float(params["huber_delta"]),
                               (C.c_float * 4)(*params["chi2_th"]), (C.c_int32 * 4)(*params["its"]))
        self.poses = np.zeros((max(F, 1), 7)); self.n_inl = np.zeros(max(F, 1), np.int32); self.n_match = np.zeros(max(F, 1), np.int32)
        self.mop = [np.full(max(n, 1), -1, np.int32) for n in self.np_]
        self.pof = [np.full(max(n, 1), -1, np.int32) for n in self.N]
        self.outl = [np.zeros(max(n, 1), np.uint8) for n in self.N]
        self.p_mop = (C.c_void_p * max(F, 1))(*[_p(x).value for x in self.mop])
        self.p_pof = (C.c_void_p * max(F, 1))(*[_p(x).value for x in self.pof])
        self.p_outl = (C.c_void_p * max(F, 1))(*[_p(x).value for x in self.outl])
        self.res = TrackResult(_p(self.poses).value, _p(self.n_inl).value, _p(self.n_match).value, C.cast(self.p_mop, C.c_void_p).value,
                               C.cast(self.p_pof, C.c_void_p).value, C.cast(self.p_outl, C.c_void_p).value)

    def track(self):
        _check(lib().dcs_track_local_map(self.F, C.cast(self.frames, C.c_void_p), C.byref(self.prm), C.byref(self.res)), "dcs_track_local_map")
        return [dict(pose=self.poses[k].copy(), n_inliers=int(self.n_inl[k]), n_matches=int(self.n_match[k]), match_of_point=self.mop[k][:self.np_[k]].copy(),
                     point_of_feature=self.pof[k][:self.N[k]].copy(), outlier=self.outl[k][:self.N[k]].copy()) for k in range(self.F)]


class DevFrame(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("cap", C.c_int32), ("first_slot", C.c_int32), ("d_kp", C.c_void_p), ("d_desc", C.c_void_p), ("d_n", C.c_void_p),
                ("min_x", C.c_void_p), ("min_y", C.c_void_p), ("grid_w_inv", C.c_void_p), ("grid_h_inv", C.c_void_p), ("K", C.c_void_p), ("dist", C.c_void_p)]


class TrackDevFrame(C.Structure):
    _fields_ = [("features", DevFrame), ("view", FrustumFrame), ("pose", C.c_void_p), ("n_held", C.c_int32), ("taken", C.c_void_p), ("has_point", C.c_void_p),
                ("point_xw", C.c_void_p), ("n_points", C.c_int32), ("pos", C.c_void_p), ("normal", C.c_void_p), ("min_dist", C.c_void_p), ("max_dist", C.c_void_p),
                ("candidate", C.c_void_p), ("desc", C.c_void_p), ("q_cam", C.c_void_p), ("q_octave", C.c_void_p), ("q_angle", C.c_void_p)]


class TrackDevResult(C.Structure):
    _fields_ = [("r", TrackResult), ("n_features", C.c_void_p)]


def undistort_points(xy, K4, dist5):
    """dcs_undistort_points: cv::undistortPoints(xy, K, dist, noArray(), K) as Frame::UndistortKeyPoints calls it"""
    xy = _c(xy, np.float32).reshape(-1, 2)
    K4, d5 = _c(K4, np.float32), np.zeros(5, np.float32)
    d5[:len(dist5)] = dist5
    out = np.zeros((max(len(xy), 1), 2), np.float32)
    L = lib()
    L.dcs_undistort_points.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(L.dcs_undistort_points(len(xy), _p(xy), _p(K4), _p(d5), _p(out)), "dcs_undistort_points")
    return out[:len(xy)]


class PreparedTrackingDevice:
    """dcs_track_frame_device: the tracking chain on frames whose features are where dcs_orb_extract_batch_device left them. frames[k] = dict(
    dev = dict(d_kp, d_desc, d_n: device pointers (ints); cap, first_slot, n_cams; min_x, min_y, grid_w_inv, grid_h_inv: per camera), view, pose,
    held = None or dict(taken, has_point, point_xw) in the frame's compact feature order, points = dict(pos[, normal, min_dist, max_dist, candidate]),
    desc, and for mode 1 q_cam, q_octave, q_angle). track() -> per frame dict incl. n_features[n_cams]; arrays cut to the real feature count."""

    def __init__(self, frames, params, mode, check_orientation=False, stream=None):
        self.keep = []

        def a(x, dt):
            if x is None:
                return None
            arr = _c(x, dt)
            self.keep.append(arr)
            return _p(arr).value
        F = len(frames)
        self.F, self.mode, self.check, self.stream = F, int(mode), int(bool(check_orientation)), stream
        arr = (TrackDevFrame * max(F, 1))()
        self.Ncap, self.np_, self.C = [], [], []
        for k, fr in enumerate(frames):
            dv, vw, pt = fr["dev"], fr["view"], fr["points"]
            t = arr[k]
            C_ = int(dv["n_cams"])
            t.features = DevFrame(C_, int(dv["cap"]), int(dv["first_slot"]), int(dv["d_kp"]), int(dv["d_desc"]), int(dv["d_n"]), a(dv["min_x"], np.float32),
                                  a(dv["min_y"], np.float32), a(dv["grid_w_inv"], np.float32), a(dv["grid_h_inv"], np.float32), a(dv.get("K"), np.float32), a(dv.get("dist"), np.float32))
            v = FrustumFrame()
            v.n_cams = len(vw["fx"])
            for key in ("Rsw", "tsw", "Ow", "fx", "fy", "cx", "cy", "min_x", "max_x", "min_y", "max_y"):
                setattr(v, key, a(vw[key], np.float32))
            v.log_scale_factor = float(vw["log_scale_factor"]); v.n_scale_levels = len(vw["scale_factors"]); v.scale_factors = a(vw["scale_factors"], np.float32)
            t.view = v
            t.pose = a(fr["pose"], np.float64)
            held = fr.get("held")
            t.n_held = 0 if held is None else len(held["taken"])
            if held is not None:
                t.taken, t.has_point, t.point_xw = a(held["taken"], np.uint8), a(held.get("has_point"), np.uint8), a(held["point_xw"], np.float32)
            t.n_points = len(pt["pos"])
            t.pos = a(pt["pos"], np.float32)
            t.normal, t.min_dist, t.max_dist = a(pt.get("normal"), np.float32), a(pt.get("min_dist"), np.float32), a(pt.get("max_dist"), np.float32)
            t.candidate = a(pt.get("candidate"), np.uint8)
            t.desc = a(fr["desc"], np.uint8)
            t.q_cam, t.q_octave, t.q_angle = a(fr.get("q_cam"), np.int32), a(fr.get("q_octave"), np.int32), a(fr.get("q_angle"), np.float32)
            self.Ncap.append(C_ * int(dv["cap"])); self.np_.append(len(pt["pos"])); self.C.append(C_)
        self.frames = arr
        cam_list = [c if isinstance(c, BaCamera) else make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in params["cams"]]
        self.cams = (BaCamera * len(cam_list))(*cam_list)
        sig = a(params["inv_level_sigma2"], np.float32)
        self.prm = TrackParams(float(params["viewing_cos_limit"]), float(params["th"]), int(params["th_high"]), float(params["nn_ratio"]),
                               len(params["inv_level_sigma2"]), sig, len(cam_list), C.cast(self.cams, C.c_void_p).value, float(params["huber_delta"]),
                               (C.c_float * 4)(*params["chi2_th"]), (C.c_int32 * 4)(*params["its"]))
        self.poses = np.zeros((max(F, 1), 7)); self.n_inl = np.zeros(max(F, 1), np.int32); self.n_match = np.zeros(max(F, 1), np.int32)
        self.nfeat = np.zeros(max(sum(self.C), 1), np.int32)
        self.mop = [np.full(max(n, 1), -1, np.int32) for n in self.np_]
        self.pof = [np.full(max(n, 1), -1, np.int32) for n in self.Ncap]
        self.outl = [np.zeros(max(n, 1), np.uint8) for n in self.Ncap]
        self.p_mop = (C.c_void_p * max(F, 1))(*[_p(x).value for x in self.mop])
        self.p_pof = (C.c_void_p * max(F, 1))(*[_p(x).value for x in self.pof])
        self.p_outl = (C.c_void_p * max(F, 1))(*[_p(x).value for x in self.outl])
        self.res = TrackDevResult(TrackResult(_p(self.poses).value, _p(self.n_inl).value, _p(self.n_match).value, C.cast(self.p_mop, C.c_void_p).value,
                                              C.cast(self.p_pof, C.c_void_p).value, C.cast(self.p_outl, C.c_void_p).value), _p(self.nfeat).value)

    def track(self):
        _check(lib().dcs_track_frame_device(self.F, C.cast(self.frames, C.c_void_p), C.byref(self.prm), self.mode, self.check, C.byref(self.res),
                                            C.c_void_p(self.stream or 0)), "dcs_track_frame_device")
        out, at = [], 0
        for k in range(self.F):
            nf = self.nfeat[at:at + self.C[k]].copy(); at += self.C[k]
            N = int(nf.sum())
            out.append(dict(pose=self.poses[k].copy(), n_inliers=int(self.n_inl[k]), n_matches=int(self.n_match[k]), match_of_point=self.mop[k][:self.np_[k]].copy(),
                            point_of_feature=self.pof[k][:N].copy(), outlier=self.outl[k][:N].copy(), n_features=nf))
        return out
