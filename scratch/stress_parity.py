"""Randomised GPU-vs-oracle parity sweep (extraction, matching); prints every mismatch. usage: stress_parity.py [seconds]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle(); synth = pkg.synth
import torch
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n_ext = n_match = bad = 0
big = [synth.frame_pair(1280, 720, s, f) for s in range(2) for f in range(2)]
while time.time() < t_end:
    # ---- extraction
    w, h = (int(rng.integers(120, 900)), int(rng.integers(100, 700))) if rng.random() < 0.9 else (int(rng.integers(900, 1281)), int(rng.integers(500, 721)))
    nf = int(rng.integers(50, 2600)); nl = int(rng.integers(1, 13)); sf = float(np.float32(rng.uniform(1.03, 1.9)))
    ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1))
    src = big[int(rng.integers(0, 4))][int(rng.integers(0, 2))]
    y0, x0 = int(rng.integers(0, 720 - h + 1)), int(rng.integers(0, 1280 - w + 1))
    img = np.ascontiguousarray(src[y0:y0 + h, x0:x0 + w])
    mode = rng.integers(0, 4)
    if mode == 1: img = (img // 64 * 64).astype(np.uint8)              # posterised: many ties
    if mode == 2: img = rng.integers(0, 256, img.shape, dtype=np.uint8)  # noise
    try:
        nb = int(rng.integers(1, 6))
        ext = pkg.ORBextractor(nf, sf, nl, ini, mn, max_images=nb)
    except pkg.DcsError:
        continue
    try:
        cap = ext.required_cap(h, w)
        imgs = [img, img[::-1].copy(), img[:, ::-1].copy(), (255 - img), np.roll(img, 7, axis=1)][:nb]
        kps, descs = ext.extract_batch(imgs, cap=cap)
    except pkg.DcsError as ex:
        ext.close(); continue                                             # image too small for the pyramid etc.
    for i, im in enumerate(imgs):
        okp, od = oracle.OrbOracle(nf, sf, nl, ini, mn).extract(im, cap=cap)
        if kps[i].tobytes() != okp.tobytes() or not np.array_equal(descs[i], od):
            bad += 1; print("EXTRACT MISMATCH", w, h, nf, nl, sf, ini, mn, mode, i, len(kps[i]), len(okp), flush=True)
    n_ext += 1
    ext.close()
    # ---- matching (device batch path = matrix-core kernel)
    nq, nt = int(rng.choice([0, 1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 1000, 1023, 1024, 1025, 2047])), int(rng.choice([0, 1, 2, 15, 16, 17, 100, 127, 128, 129, 255, 256, 257, 1000, 1500]))
    cap = max(nq, nt, 1) + int(rng.integers(0, 50))
    dq = synth.random_descriptors(max(nq, 1), seed=int(rng.integers(1 << 30)))[:nq]
    dt = synth.random_descriptors(max(nt, 1), seed=int(rng.integers(1 << 30)))[:nt]
    if nq and nt and rng.random() < 0.7:
        k = min(nq, nt); dq[:k] = synth.noisy_copy(dt[:k], flip_bits=int(rng.integers(0, 60)), seed=3)
        if rng.random() < 0.3: dt[rng.integers(0, nt, 5)] = dt[0]        # duplicates: ties
    d_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda"); d_kp = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda")
    d_desc[0, :nq] = torch.from_numpy(dq).cuda(); d_desc[1, :nt] = torch.from_numpy(dt).cuda()
    d_n = torch.tensor([nq, nt], dtype=torch.int32, device="cuda"); d_pairs = torch.tensor([[0, 1]], dtype=torch.int32, device="cuda")
    d_m = torch.zeros((1, cap), dtype=torch.int32, device="cuda"); d_nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_b = torch.zeros((1, cap), dtype=torch.int32, device="cuda"); d_s = torch.zeros((1, cap), dtype=torch.int32, device="cuda")
    pkg.ORBmatcher(0.75, False).match_bf_batch_device(d_desc, d_kp, d_n, cap, d_pairs, 1, d_m, d_nm, d_b, d_s, 256, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    if nq:
        bi, bd, sd = oracle.knn2(dq, dt) if nt else (np.full(nq, -1), np.full(nq, 256), np.full(nq, 256))
        gb, gs = d_b[0, :nq].cpu().numpy(), d_s[0, :nq].cpu().numpy()
        if not (np.array_equal(gb, bd) and np.array_equal(gs, sd)):
            bad += 1; print("KNN2 MISMATCH", nq, nt, cap, np.nonzero(gb != bd)[0][:5], np.nonzero(gs != sd)[0][:5], flush=True)
    n_match += 1
print("extraction configs %d, matching configs %d, mismatches %d" % (n_ext, n_match, bad))
