"""512 host images through dcs_orb_extract_batch with the per-stage host trace (DCS_ORB_HOST_TRACE=1); usage: host_pipe_trace.py [images] [chunk]"""
import os, sys, time, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DCS_ORB_HOST_TRACE"] = "1"
if len(sys.argv) > 2: os.environ["DCS_ORB_HOST_CHUNK"] = sys.argv[2]
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
imgs = []
for f in range(B // 2): imgs.extend(synth.frame_pair(640, 480, 0, f % 8))
imgs = [np.ascontiguousarray(im.copy()) for im in imgs]
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=B)
cap = ext.default_cap()
kp = np.zeros((B, cap), pkg.abi.KEYPOINT); desc = np.zeros((B, cap, 32), np.uint8); n_out = np.zeros(B, np.int32)
ptrs = (C.c_void_p * B)(*[im.ctypes.data for im in imgs])
call = lambda: pkg.abi.lib().dcs_orb_extract_batch(ext._h, C.cast(ptrs, C.c_void_p), B, 480, 640, 640, kp.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, n_out.ctypes.data_as(C.c_void_p))
for _ in range(3): assert call() == 0
ts = []
for _ in range(10):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
print("median %.3f ms, min %.3f ms per call of %d images, %d features; host cores %d" % (np.median(ts) * 1e3, min(ts) * 1e3, B, int(n_out.sum()), os.cpu_count()))
