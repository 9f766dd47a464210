"""The DCS_WITH_OPENCV half of the C++ boundary (orb-slam2-dualcam_amd/host/ORBextractor.h: the reference's exact
`operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray)`, include/ORBextractor.h:59-61) goes through a
compiler: `g++ -fsyntax-only -DDCS_WITH_OPENCV` against tests/cpp/cv_syntax_stub.h -- a declaration-only SYNTAX STAND-IN for the few
OpenCV names the block touches, which pins nothing, links nothing and computes nothing (the image has no OpenCV). The translation unit
calls the operator the way Frame::ExtractORB does (src/Frame.cc:210-213) and static_asserts cv::KeyPoint's size and field offsets
against dcs_keypoint."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "orb-slam2-dualcam_amd", "host")
FLAGS = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-DDCS_WITH_OPENCV",
         "-I", os.path.join(ROOT, "include"), "-I", HOST, "-I", os.path.join(ROOT, "tests", "cpp", "cv_stub_include")]


def _syntax(args, src=None):
    return subprocess.run(FLAGS + args + (["-x", "c++", "-"] if src is not None else []), input=src, capture_output=True, text=True)


def test_reference_call_site_parses_against_the_mirror():
    p = _syntax([os.path.join(ROOT, "tests", "cpp", "cv_boundary_syntax.cpp")])
    assert p.returncode == 0, p.stderr[-3000:]


@pytest.mark.parametrize("header", ["ORBextractor.h", "ORBmatcher.h", "Optimizer.h", "ORBVocabulary.h", "KeyFrameDatabase.h"])
def test_every_mirror_header_stands_alone_with_opencv_defined(header):
    p = _syntax([], '#include "%s"\n' % header)
    assert p.returncode == 0, p.stderr[-3000:]


def test_the_check_has_teeth():
    """a call with the wrong argument list (the mask left out) and a wrong KeyPoint layout must NOT parse"""
    bad_call = '#include "ORBextractor.h"\nvoid f(ORB_SLAM2::ORBextractor& e, cv::Mat& im, std::vector<cv::KeyPoint>& k, cv::Mat& d) { e(im, k, d); }\n'
    assert _syntax([], bad_call).returncode != 0
    p = _syntax(["-DDCS_STUB_BREAK_LAYOUT"], '#include "ORBextractor.h"\n')
    assert p.returncode != 0 and "cv::KeyPoint layout" in p.stderr
