"""The C-ABI library loads without a GPU and exports every symbol include/dcs_abi.h declares; compute entry
points fail loudly (DCS_ERR_NO_DEVICE) instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import gpu_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "dcs_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dcs_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(pkg):
    lib = pkg.abi.lib()
    declared = _declared()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(pkg.abi.SYMBOLS) == declared
    assert lib.dcs_version().decode().startswith("dcs-hip")


def test_header_is_plain_c(tmp_path):
    """include/dcs_abi.h is what a cgo / JNI / ctypes binding compiles against: it must stand alone as C99 (no C++, every type it
    names declared by its own includes)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "dcs_abi.h"\nint main(void) { dcs_keypoint k; (void)k; return 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_struct_layouts_match_header(pkg):
    assert pkg.abi.KEYPOINT.itemsize == 28 and pkg.abi.CANDIDATE.itemsize == 8
    assert C.sizeof(pkg.abi.OrbParams) == 32
    assert C.sizeof(pkg.abi.BaCamera) == 8 * (4 + 7 + 36)
    assert C.sizeof(pkg.abi.BaProblem) == 16 + 9 * 8 + 16 + 8
    assert pkg.abi.BaResult.chi2_trace.offset == 5 * 8 + 8 + 8 + 16


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback(pkg):
    assert pkg.abi.device_count() == 0
    with pytest.raises(pkg.DcsError) as ei:
        pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    assert ei.value.rc == pkg.abi.DCS_ERR_NO_DEVICE
    q = np.zeros((4, 32), np.uint8)
    with pytest.raises(pkg.DcsError) as ei:
        pkg.ORBmatcher.knn2(q, q)
    assert ei.value.rc == pkg.abi.DCS_ERR_NO_DEVICE


def test_host_helpers_without_gpu(pkg, oracle, synth):
    T0, T1 = synth.rig_extrinsics_f32()
    for exact in (False, True):
        adj, ext = pkg.abi.rig_adjoint(T1, exact)
        oadj, oext = oracle.rig_adjoint(T1, exact)
        assert np.array_equal(adj, oadj) and np.array_equal(ext, oext)
    p = pkg.abi.pose_from_matrix(T1)
    assert np.allclose(p[:3], T1[:3, 3]) and abs(np.linalg.norm(p[3:]) - 1) < 1e-15 and p[6] > 0
    assert np.allclose(pkg.abi.pose_to_matrix(p), T1, atol=1e-6)
    # bad arguments are reported, not crashed on
    bad = pkg.abi.OrbParams(0, 1.2, 8, 20, 7, -1, 1, 0)
    h = C.c_void_p()
    assert pkg.abi.lib().dcs_orb_create(C.byref(bad), C.byref(h)) == pkg.abi.DCS_ERR_INVALID
    assert b"bad ORB parameters" in pkg.abi.lib().dcs_last_error()
    # malformed CSR lists (feature vectors) are rejected before anything is sent to the GPU
    q = np.zeros((4, 32), np.uint8)
    out = [np.zeros(4, np.int32) for _ in range(3)]
    good_off, idx = np.array([0, 2, 4], np.int32), np.array([0, 1, 2, 3], np.int32)
    for off, ix in [(np.array([0, 3, 2], np.int32), idx), (np.array([1, 2, 4], np.int32), idx), (good_off, np.array([0, 1, 2, 4], np.int32)),
                    (good_off, np.array([0, -1, 2, 3], np.int32))]:
        rc = pkg.abi.lib().dcs_hamming_knn2_grouped(q.ctypes.data, 4, q.ctypes.data, 4, 2, good_off.ctypes.data, idx.ctypes.data, off.ctypes.data,
                                                    ix.ctypes.data, *[o.ctypes.data for o in out])
        assert rc == pkg.abi.DCS_ERR_INVALID
    with pytest.raises(ValueError):
        pkg.ORBmatcher.knn2_grouped(q, q, good_off, idx, good_off[:2], idx[:2])
