# per-call latency of the host-buffer seams with 1000 x 1000 features (what one reference call costs here)
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
a, b = synth.frame_pair(640, 480, 0, 0)
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
m = pkg.ORBmatcher(0.75, True)
kps, descs = ext.extract_batch([a, b])
def per_call(fn, reps=200):
    for _ in range(10): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return "median %.4f ms, p10 %.4f, p90 %.4f" % (ts[len(ts) // 2] * 1e3, ts[len(ts) // 10] * 1e3, ts[9 * len(ts) // 10] * 1e3)
print("extract_batch(2 images)      ", per_call(lambda: ext.extract_batch([a, b])))
print("dcs_match_bf                 ", per_call(lambda: m.match_bf(descs[0], kps[0], descs[1], kps[1], 50)))
print("dcs_hamming_knn2             ", per_call(lambda: pkg.ORBmatcher.knn2(descs[0], descs[1])))
import ctypes as C
L = pkg.abi.lib()
q, t = np.ascontiguousarray(descs[0]), np.ascontiguousarray(descs[1])
bi, bd, sd = (np.zeros(len(q), np.int32) for _ in range(3))
P = lambda x: x.ctypes.data_as(C.c_void_p)
print("dcs_hamming_knn2 raw ctypes  ", per_call(lambda: L.dcs_hamming_knn2(P(q), len(q), P(t), len(t), None, P(bi), P(bd), P(sd))))
cap = ext.default_cap()
kp_o = np.zeros((2, cap), pkg.abi.KEYPOINT); d_o = np.zeros((2, cap, 32), np.uint8); n_o = np.zeros(2, np.int32)
ptrs = (C.c_void_p * 2)(a.ctypes.data, b.ctypes.data)
print("dcs_orb_extract_batch raw    ", per_call(lambda: L.dcs_orb_extract_batch(ext._h, C.cast(ptrs, C.c_void_p), 2, 480, 640, 640, P(kp_o), P(d_o), cap, P(n_o))))
