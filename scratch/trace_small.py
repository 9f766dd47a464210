# kernel timeline of ONE dual-frame host call (run under rocprofv3 --kernel-trace --memory-copy-trace)
import os, sys, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
a, b = synth.frame_pair(640, 480, 0, 0)
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
L = pkg.abi.lib(); cap = ext.default_cap()
kp_o = np.zeros((2, cap), pkg.abi.KEYPOINT); d_o = np.zeros((2, cap, 32), np.uint8); n_o = np.zeros(2, np.int32)
ptrs = (C.c_void_p * 2)(a.ctypes.data, b.ctypes.data)
P = lambda x: x.ctypes.data_as(C.c_void_p)
for _ in range(30): L.dcs_orb_extract_batch(ext._h, C.cast(ptrs, C.c_void_p), 2, 480, 640, 640, P(kp_o), P(d_o), cap, P(n_o))
