"""How selective would a second necessary test be? Among the compass survivors (ring 0/4/8/12) of the benchmark scene: the share that
also passes the same test on the diagonal ring pixels (2/6/10/14), and the share that really holds a 9-arc (CPU, oracle pyramid)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); O = e.load_oracle()
img, _ = pkg.synth.frame_pair(640, 480, 0, 0)
orc = O.OrbOracle(1000, 1.2, 8, 20, 7); orc.extract(img)
ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]   # (dx, dy), OpenCV order
for th in (20, 7):
    tot = np.zeros(4)
    for l in range(8):
        v = orc.level_image(l).astype(np.int32); H, W = v.shape
        c = v[3:-3, 3:-3]
        R = [v[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx] for dx, dy in ring]
        def quad(a, b, cc, d):
            br = np.minimum(np.maximum(a, cc), np.maximum(b, d)) > c + th
            dk = np.maximum(np.minimum(a, cc), np.minimum(b, d)) < c - th
            return br, dk
        b1, d1 = quad(R[0], R[4], R[8], R[12]); b2, d2 = quad(R[2], R[6], R[10], R[14])
        B = np.stack([r > c + th for r in R]); D = np.stack([r < c - th for r in R])
        def arc9(M):
            MM = np.concatenate([M, M[:8]]); out = np.zeros(M.shape[1:], bool)
            for k in range(16): out |= MM[k:k + 9].all(0)
            return out
        corner = arc9(B) | arc9(D)
        s1 = b1 | d1; s2 = (b1 & b2) | (d1 & d2)
        assert not (corner & ~s2).any()
        tot += [s1.size, s1.sum(), s2.sum(), corner.sum()]
        print("th %2d level %d: compass %.1f%% of pixels; of those: diagonal test keeps %.0f%%, true 9-arc %.0f%%" % (th, l, 100 * s1.mean(), 100 * s2.sum() / max(s1.sum(), 1), 100 * corner.sum() / max(s1.sum(), 1)))
    print("th %2d all levels: compass %.1f%%, diagonal keeps %.0f%% of them, true corners %.0f%%" % (th, 100 * tot[1] / tot[0], 100 * tot[2] / tot[1], 100 * tot[3] / tot[1]))
