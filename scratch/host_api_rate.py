import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
import torch
for B in (2, 16, 64, 256):
    imgs = []
    for f in range(B // 2): imgs.extend(synth.frame_pair(640, 480, 0, f % 8))
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=B)
    import ctypes as C
    cap = ext.default_cap()
    kp = np.zeros((B, cap), pkg.abi.KEYPOINT); desc = np.zeros((B, cap, 32), np.uint8); n_out = np.zeros(B, np.int32)
    ptrs = (C.c_void_p * B)(*[im.ctypes.data for im in imgs])
    call = lambda: pkg.abi.lib().dcs_orb_extract_batch(ext._h, C.cast(ptrs, C.c_void_p), B, 480, 640, 640, kp.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, n_out.ctypes.data_as(C.c_void_p))
    for _ in range(3): assert call() == 0
    t0 = time.perf_counter(); R = max(3, 200 // B)
    for _ in range(R): call()
    dt = (time.perf_counter() - t0) / R
    nf = int(n_out.sum())
    print("host API extract_batch B=%d images: %.3f ms/call, %.1f kfeatures/s (H2D images + D2H keypoints/descriptors included)" % (B, dt * 1e3, nf / dt / 1e3))
    ext.close()
# device-resident latency, one dual frame
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
imgs = list(synth.frame_pair(640, 480, 0, 0))
d_img = torch.from_numpy(np.stack(imgs)).cuda(); cap = ext.default_cap()
d_kp = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda"); d_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda"); d_n = torch.zeros(2, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(5): ext.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    ext.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=s); torch.cuda.synchronize()
print("device-resident single dual frame latency: %.3f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
