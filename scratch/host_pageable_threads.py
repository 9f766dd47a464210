"""512 pageable frames per dcs_orb_extract_batch call: packing threads of the staging pool (DCS_ORB_STAGING_THREADS is read when the handle is created)"""
import os, sys, time, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
B, H, W = 512, 480, 640
pairs = [synth.frame_pair(W, H, 0, f) for f in range(8)]
imgs = [np.ascontiguousarray(pairs[(i // 2) % 8][i % 2]).copy() for i in range(B)]
for rnd in range(2):
    for nt in (8, 16, 32, 12, 4):
        pkg.abi.set_option("DCS_ORB_STAGING_THREADS", nt)
        ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=B); ext.set_timing(0)
        cap = ext.default_cap()
        kp = np.zeros((B, cap), pkg.abi.KEYPOINT); desc = np.zeros((B, cap, 32), np.uint8); n_out = np.zeros(B, np.int32)
        ptrs = (C.c_void_p * B)(*[f.ctypes.data for f in imgs])
        call = lambda: pkg.abi.lib().dcs_orb_extract_batch(ext._h, C.cast(ptrs, C.c_void_p), B, H, W, W, kp.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, n_out.ctypes.data_as(C.c_void_p))
        for _ in range(3): assert call() == 0
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[7]
        print("round %d threads %2d: %.2f ms per call (mean %.2f), %.1f kfeatures/s, %.1f GB/s" % (rnd, nt, t * 1e3, sum(ts) / 15 * 1e3, int(n_out.sum()) / t / 1e3, B * W * H / t / 1e9))
        ext.close()
