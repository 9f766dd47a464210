# host-image batch through dcs_orb_extract_batch: per-stage trace (DCS_ORB_HOST_TRACE=1), chunk sizes and staging-thread counts
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
import ctypes as C
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
imgs = []
for f in range(B // 2): imgs.extend(synth.frame_pair(640, 480, 0, f % 8))
imgs = [np.ascontiguousarray(im).copy() for im in imgs]
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=B)
cap = ext.default_cap()
kp = np.zeros((B, cap), pkg.abi.KEYPOINT); desc = np.zeros((B, cap, 32), np.uint8); n_out = np.zeros(B, np.int32)
ptrs = (C.c_void_p * B)(*[im.ctypes.data for im in imgs])
call = lambda: pkg.abi.lib().dcs_orb_extract_batch(ext._h, C.cast(ptrs, C.c_void_p), B, 480, 640, 640, kp.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, n_out.ctypes.data_as(C.c_void_p))
for _ in range(3): assert call() == 0
t0 = time.perf_counter(); R = 5
for _ in range(R): call()
dt = (time.perf_counter() - t0) / R
print("B=%d chunk=%s threads=%s: %.3f ms/call, %.1f kfeatures/s, %.1f GB/s of images" % (B, os.environ.get("DCS_ORB_HOST_CHUNK", "64"), os.environ.get("DCS_ORB_STAGING_THREADS", "default"), dt * 1e3, int(n_out.sum()) / dt / 1e3, B * 640 * 480 / dt / 1e9))
ts = []
for _ in range(12):
    t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
print("per call ms:", " ".join("%.2f" % t for t in ts))
import torch
x = torch.empty(B * 640 * 480, dtype=torch.uint8).pin_memory(); y = torch.empty_like(x, device="cuda")
torch.cuda.synchronize()
for _ in range(2): y.copy_(x, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): y.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("raw pinned H2D of %d MB: %.3f ms = %.1f GB/s" % (x.numel() >> 20, dt * 1e3, x.numel() / dt / 1e9))
t0 = time.perf_counter()
for _ in range(5): x.copy_(y, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("raw pinned D2H: %.3f ms = %.1f GB/s" % (dt * 1e3, x.numel() / dt / 1e9))
