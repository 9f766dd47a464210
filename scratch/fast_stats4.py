"""Selectivity of cheap necessary tests on the compass survivors (benchmark scene, CPU, oracle pyramid), th = 20 / 7:
A any-polarity diagonal quad (what k_fast_cells step 2b does), B same-polarity compass + diagonal quads, C B + the two odd quads
(1,5,9,13) and (3,7,11,15), D four consecutive of the eight even ring pixels, against the true 9-arc share."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); O = e.load_oracle()
img, _ = pkg.synth.frame_pair(640, 480, 0, 0)
orc = O.OrbOracle(1000, 1.2, 8, 20, 7); orc.extract(img)
ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
for th in (20, 7):
    tot = np.zeros(6)
    for l in range(8):
        v = orc.level_image(l).astype(np.int32); H, W = v.shape
        c = v[3:-3, 3:-3]
        R = [v[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] for dx, dy in ring]
        def quad(i):
            a, b, cc, d = R[i], R[i + 4], R[i + 8], R[i + 12]
            return np.minimum(np.maximum(a, cc), np.maximum(b, d)) > c + th, np.maximum(np.minimum(a, cc), np.minimum(b, d)) < c - th
        (b0, d0), (b1, d1), (b2, d2), (b3, d3) = quad(0), quad(1), quad(2), quad(3)
        B = np.stack([r > c + th for r in R]); D = np.stack([r < c - th for r in R])
        def arcs(M, idx, n):
            MM = np.concatenate([M[idx], M[idx][:n]]); out = np.zeros(M.shape[1:], bool)
            for k in range(len(idx)): out |= MM[k:k + n].all(0)
            return out
        corner = arcs(B, list(range(16)), 9) | arcs(D, list(range(16)), 9)
        s1 = b0 | d0
        A = s1 & (b2 | d2); Bq = (b0 & b2) | (d0 & d2); C = (b0 & b1 & b2 & b3) | (d0 & d1 & d2 & d3)
        ev = list(range(0, 16, 2))
        Dd = arcs(B, ev, 4) | arcs(D, ev, 4)
        for t in (A, Bq, C, Dd): assert not (corner & ~t).any()
        tot += [s1.sum(), A.sum(), Bq.sum(), C.sum(), (s1 & Dd).sum(), corner.sum()]
    print("th %2d: of the compass survivors: A %.0f%%  B %.0f%%  C %.0f%%  D %.0f%%  true %.0f%%" % ((th,) + tuple(100 * tot[k] / tot[0] for k in range(1, 6))))
