"""The header-only C++ mirrors of the reference classes (orb-slam2-dualcam_amd/host/*.h) compile against the C ABI
with plain g++ and give the oracle's results when driven like the reference's own call sites."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "orb-slam2-dualcam_amd")


def _fnv(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _build(tmp_path):
    exe = str(tmp_path / "mirror_test")
    torch_lib = None
    try:
        import torch
        torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    except ImportError:
        pass
    cmd = ["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "host"),
           os.path.join(ROOT, "tests", "cpp", "mirror_test.cpp"), "-L", os.path.join(PKG, "lib"), "-ldcs_hip",
           "-Wl,-rpath," + os.path.join(PKG, "lib"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_mirror_headers_compile(tmp_path, pkg):
    pkg.abi.lib()
    _build(tmp_path)


@pytest.mark.gpu
def test_mirror_matches_oracle(tmp_path, pkg, oracle, synth):
    exe = _build(tmp_path)
    img0, img1 = synth.frame_pair(640, 480, 0, 0)
    raw = tmp_path / "pair.raw"
    raw.write_bytes(img0.tobytes() + img1.tobytes())
    out = subprocess.run([exe, str(raw), "480", "640", "1000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = dict((l.split()[0], l.split()[1:]) for l in out.stdout.strip().splitlines())
    feats = [oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(im) for im in (img0, img1)]
    for i in range(2):
        kp, desc = feats[i]
        n, hk, hd = lines["kp%d" % i]
        assert int(n) == len(kp) and int(hk, 16) == _fnv(kp.tobytes()) and int(hd, 16) == _fnv(desc.tobytes())
    bi, bd, sd = oracle.knn2(feats[0][1], feats[1][1])
    m, n = oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, feats[0][0]["angle"], feats[1][0]["angle"])
    assert int(lines["match"][0]) == n and int(lines["match"][1], 16) == _fnv(m.tobytes())
    assert lines["empty"] == ["0"]
