"""device-side time of one extraction call at several batch sizes: resize chain, emitting FAST, chained FAST. usage: chain_sizes.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
import torch
def run(n, w, h, nf, **opts):
    imgs = []
    for i in range(max(n // 2, 1)):
        a, b = synth.frame_pair(w, h, i % 8, 0); imgs += [a, b]
    imgs = imgs[:n]
    d = torch.from_numpy(np.stack(imgs)).cuda()
    with pkg.abi.options(**opts):
        ext = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_images=n)
    cap = ext.required_cap(h, w)
    kp = torch.empty((n, cap, 20), dtype=torch.uint8, device="cuda"); ds = torch.empty((n, cap, 32), dtype=torch.uint8, device="cuda"); cnt = torch.empty(n, dtype=torch.int32, device="cuda")
    ext.set_timing(0)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): ext.extract_batch_device(d, kp, ds, cnt, cap, stream=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): ext.extract_batch_device(d, kp, ds, cnt, cap, stream=st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    cf = ext.chain_first(); el = ext.emit_levels()
    dig = hash(kp.cpu().numpy().tobytes() + cnt.cpu().numpy().tobytes())
    ext.close()
    return ms, cf, el, dig
for (w, h, nf) in ((640, 480, 1000), (1280, 720, 2000)):
    for n in (8, 16, 32, 64, 128, 256):
        if w == 1280 and n > 128: continue
        row = []
        for name, o in (("resize", dict(DCS_ORB_EMIT=0)), ("emit", dict(DCS_ORB_EMIT=15)), ("chain5", dict(DCS_ORB_EMIT=15, DCS_ORB_CHAIN=5)), ("chain4", dict(DCS_ORB_EMIT=15, DCS_ORB_CHAIN=4)), ("chain3", dict(DCS_ORB_EMIT=15, DCS_ORB_CHAIN=3)), ("chain2", dict(DCS_ORB_EMIT=15, DCS_ORB_CHAIN=2))):
            ms, cf, el, dig = run(n, w, h, nf, **o)
            row.append("%s %.3f" % (name, ms))
        print("%dx%d n=%3d: " % (w, h, n) + "  ".join(row), flush=True)
