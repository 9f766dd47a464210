"""The reference-typed half of the matcher / optimiser boundary (orb-slam2-dualcam_amd/host/ReferenceAdapters.h, -DDCS_WITH_REFERENCE_MODEL) goes
through a compiler: the reference's own signatures -- Optimizer::LocalBundleAdjustment(KeyFramePtr, bool*, MapPtr, size_t), BundleAdjustment,
GlobalBundleAdjustemnt, PoseOptimization(FramePtr), ORBmatcher::SearchByProjection (two overloads), SearchByProjectionOnCam, SearchByBoWCrossCam
(include/Optimizer.h:49-56, include/ORBmatcher.h:65-200) -- with bodies written against the accessors of the reference's data model, parsed and
type-checked by `g++ -fsyntax-only -Wall -Wextra -Werror` against tests/cpp/slam_model_stub.h: a declaration-only SYNTAX STAND-IN of Frame /
KeyFrame / MapPoint / Map / Cameras that pins nothing, links nothing and computes nothing. The translation unit holds the reference's own call
lines (LocalMapping.cc:103; Tracking.cc:822, 1321, 1406, 1427, 1680)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "orb-slam2-dualcam_amd", "host")
FLAGS = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-DDCS_WITH_OPENCV", "-DDCS_WITH_REFERENCE_MODEL",
         "-I", os.path.join(ROOT, "include"), "-I", HOST, "-I", os.path.join(ROOT, "tests", "cpp"), "-I", os.path.join(ROOT, "tests", "cpp", "cv_stub_include")]


def _syntax(args, src=None):
    return subprocess.run(FLAGS + args + (["-x", "c++", "-"] if src is not None else []), input=src, capture_output=True, text=True)


def test_reference_call_lines_resolve_to_the_reference_typed_members():
    p = _syntax([os.path.join(ROOT, "tests", "cpp", "reference_model_syntax.cpp")])
    assert p.returncode == 0, p.stderr[-4000:]


def test_the_adapters_are_absent_without_the_switch():
    """without -DDCS_WITH_REFERENCE_MODEL the mirrors know nothing of FramePtr: the flat-array API stands alone (as the other tests use it)"""
    flags = [f for f in FLAGS if f != "-DDCS_WITH_REFERENCE_MODEL"]
    p = subprocess.run(flags + ["-x", "c++", "-"], input='#include "ORBmatcher.h"\n#include "Optimizer.h"\n#include "ReferenceAdapters.h"\n', capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]


def test_the_check_has_teeth():
    """a call with the reference's argument order broken, and an accessor the data model does not have, must NOT parse"""
    head = '#include "slam_model_stub.h"\n#include "ORBextractor.h"\n#include "ReferenceAdapters.h"\nusing namespace ORB_SLAM2;\n'
    bad_call = head + "void f(KeyFramePtr kf, MapPtr m, bool* stop) { Optimizer::LocalBundleAdjustment(kf, m, stop, 0); }\n"
    assert _syntax([], bad_call).returncode != 0
    bad_member = head + "int g(FramePtr f) { return f->mvKeysThatDoNotExist.size(); }\n"
    assert _syntax([], bad_member).returncode != 0
    good = head + "void h(KeyFramePtr kf, MapPtr m, bool* stop) { Optimizer::LocalBundleAdjustment(kf, stop, m, 0); }\n"
    assert _syntax([], good).returncode == 0, _syntax([], good).stderr[-2000:]
