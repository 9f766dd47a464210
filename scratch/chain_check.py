"""chained FAST: does it engage, is it bit-equal to the per-level launches, how long does a batch take. usage: chain_check.py [n_images]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
imgs = []
for i in range(n // 2):
    a, b = synth.frame_pair(640, 480, i, 0); imgs += [a, b]
res = {}
for chain in (0, 5, 4, 3, 2, 1):
    with pkg.abi.options(DCS_ORB_EMIT=15, DCS_ORB_CHAIN=chain):
        ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=n)
    k, ds = ext.extract_batch(list(imgs))
    t0 = time.perf_counter()
    for _ in range(5): k, ds = ext.extract_batch(list(imgs))
    dt = (time.perf_counter() - t0) / 5
    dig = hash(b"".join(x.tobytes() for x in k) + b"".join(np.ascontiguousarray(x).tobytes() for x in ds))
    res[chain] = dig
    print("chain option %d: engaged from level %d, emit levels %d, %.3f ms per call (host API), digest %x" % (chain, ext.chain_first(), ext.emit_levels(), dt * 1e3, dig & 0xffffffff))
    ext.close()
print("all equal:", len(set(res.values())) == 1)
