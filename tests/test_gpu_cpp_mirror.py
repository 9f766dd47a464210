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
    voc = synth.vocabulary(k=6, L=5, seed=4, ragged=0.2, stop_frac=0.05)
    voc_path = str(tmp_path / "voc.txt")
    synth.vocabulary_to_text(voc, voc_path)
    out = subprocess.run([exe, str(raw), "480", "640", "1000", voc_path], capture_output=True, text=True, timeout=120)
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
    # ORBVocabulary mirror: loadFromTextFile + transform(levelsup 4 -> level-1 nodes of this L = 5 tree) + score
    import struct
    V = oracle.Vocabulary(voc["k"], voc["L"], voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    r0, r1 = V.transform(feats[0][1], 4), V.transform(feats[1][1], 4)
    assert int(lines["bow"][0]) == V.n_words() and int(lines["bow"][1]) == len(r0["bow_word"])
    assert int(lines["bow"][2], 16) == _fnv(r0["bow_word"].tobytes()) and int(lines["bow"][3], 16) == _fnv(r0["bow_val"].tobytes())
    assert int(lines["fv"][0]) == len(r0["fv_node"]) and int(lines["fv"][1], 16) == _fnv(r0["fv_node"].tobytes())
    assert int(lines["fv"][2], 16) == _fnv(r0["fv_off"].tobytes()) and int(lines["fv"][3], 16) == _fnv(r0["fv_idx"].tobytes())
    s01 = oracle.bow_score_l1(r0["bow_word"], r0["bow_val"], [0, len(r1["bow_word"])], r1["bow_word"], r1["bow_val"])[0]
    s00 = oracle.bow_score_l1(r0["bow_word"], r0["bow_val"], [0, len(r0["bow_word"])], r0["bow_word"], r0["bow_val"])[0]
    assert int(lines["score"][0], 16) == _fnv(struct.pack("<d", s01)) and int(lines["score"][1], 16) == _fnv(struct.pack("<d", s00))
