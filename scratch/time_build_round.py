"""Host time of the BA structure build (CSR lists + pose-pair lists) of the C4 problem; runs without a GPU.
usage: time_build_round.py [reps]   (DCS_LIB_PATH selects the build)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_pkg
pkg = load_pkg()
import importlib
synth = importlib.import_module("orb_slam2_dualcam_amd.synth")
abi = importlib.import_module("orb_slam2_dualcam_amd.abi")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prep = pkg.Optimizer.prepare(synth.ba_problem(seed=42))
fn = abi.lib().dcs_debug_ba_build_ms
fn.restype = C.c_double
fn.argtypes = [C.c_void_p, C.c_int]
for _ in range(3):
    print("build_round: %.4f ms" % fn(C.addressof(prep.pb), reps))
