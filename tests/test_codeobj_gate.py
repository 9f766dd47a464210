"""The build gate (tools/check_codeobj.py, run by build.sh) on the library that is in the tree: every gfx950 kernel without spilled vector registers,
without scratch memory and below 48 KB of code; spilled SCALAR registers only for the kernels tools/codeobj_allow.txt names -- and a name there covers
that kernel only (a substring match once let k_pose_opt's entry cover k_pose_opt2). CPU only: the code objects' metadata is read with llvm-readelf."""
import os, subprocess, sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_codeobj as cc  # noqa: E402

LIB = os.path.join(ROOT, "orb-slam2-dualcam_amd", "lib", "libdcs_hip.so")


def test_allow_list_names_a_kernel_exactly():
    assert cc.names("k_pose_opt", "k_pose_opt") and not cc.names("k_pose_opt", "k_pose_opt2") and not cc.names("k_pose_opt", "k_pose_opt2<true>")
    assert cc.names("k_octree_hist", "k_octree_hist<1>") and not cc.names("k_octree", "k_octree_hist<1>")
    assert cc.names("k_ldlt_mfma<13>", "k_ldlt_mfma<13>") and not cc.names("k_ldlt_mfma<13>", "k_ldlt_mfma<11>")
    assert not cc.names("k_track_resolve", "k_track_resolve_cam")


@pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists(os.path.join(cc.LLVM, "llvm-readelf")), reason="needs the built library and the ROCm LLVM tools")
def test_the_library_in_the_tree_passes_the_gate():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_codeobj.py"), LIB, "--sgpr-only-allow-file", os.path.join(ROOT, "tools", "codeobj_allow.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "failing kernels: 0" in r.stdout
    # and without the allow list the kernels it names are the ONLY ones that fail (scalar spills): the gate is not vacuous
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_codeobj.py"), LIB, "--verbose"], capture_output=True, text=True)
    failing = [ln for ln in r2.stdout.splitlines() if "<-- FAIL" in ln]
    assert r2.returncode != 0 and 0 < len(failing) <= 12
    assert all("spills" in ln and "scratch" not in ln.split("<-- FAIL")[1] and "code size" not in ln for ln in failing), failing
