#!/usr/bin/env python3
"""In-kernel timeline of k_pose_opt2 (GPU box). Needs a side build with -DDCS_POSE_PROF=<pass number>:

    DCS_OUT_DIR=scratch/ab/pose_prof DCS_OBJ_DIR=scratch/ab/pose_prof_obj DCS_EXTRA_FLAGS=-DDCS_POSE_PROF=20 ./build.sh
    DCS_LIB_PATH=$PWD/scratch/ab/pose_prof/libdcs_hip.so python tools/pose_timeline.py [n_points] [n_features]

Prints, for frame 0 of a one-frame dcs_track_local_map call, every wave's clock at the stage boundaries of the recorded pass (relative to the
earliest wave's start of that pass) and the kernel's span on the shader clock and on the 100 MHz real-time clock (= the shader clock's rate)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nfe = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
fr, prm = synth.tracking_problem(n_frames=1, n_points=npts, n_features=nfe, seed=23)
for f in fr:
    ft = f["features"]
    ft["grid_off"], ft["grid_idx"] = pkg.frame_grid(ft["cam_off"], ft["kp_x"], ft["kp_y"], ft["min_x"], ft["min_y"], ft["grid_w_inv"], ft["grid_h_inv"])
pt = pkg.abi.PreparedTracking(fr, prm)
for _ in range(5): r = pt.track()
lib = pkg.abi.lib() if callable(getattr(pkg.abi, "lib", None)) else ctypes.CDLL(pkg.abi.LIB_PATH)
buf = (ctypes.c_ulonglong * (4 * 16))()
fn = lib.dcs_debug_pose_prof
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert fn(buf) == 0
t = np.array(list(buf), dtype=np.uint64).reshape(4, 16).astype(np.int64)
names = ["(compose +) sweep", "reduce", "adjoint map + partials out", "barrier", "totals in", "LM rule", "solve", "exp map"]
span_clk = (t[:, 10] - t[:, 9]).max(); span_rt = (t[:, 12] - t[:, 11]).max()
print("frame 0: %d edges, %d slots per lane (wave 0), %d passes; kernel span %d shader ticks = %.1f us on the 100 MHz clock -> %.0f MHz; %.0f ticks per pass"
      % (t[0, 14], t[0, 15], t[0, 13], span_clk, span_rt / 100.0, span_clk / (span_rt / 100.0), span_clk / max(1, t[0, 13])))
t0 = t[:, 8].min()
print("pass recorded: stage ends per wave, ticks after the earliest wave entered the pass")
print("%-28s" % "wave (slots)" + "".join("%8d" % w for w in range(4)))
print("%-28s" % "  slots per lane" + "".join("%8d" % t[w, 15] for w in range(4)))
print("%-28s" % "  enters the pass" + "".join("%8d" % (t[w, 8] - t0) for w in range(4)))
for k, nm in enumerate(names):
    print("%-28s" % ("  " + nm) + "".join("%8d" % (t[w, k] - t0) for w in range(4)))
