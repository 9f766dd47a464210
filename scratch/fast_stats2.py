"""Compass survivors per FAST cell over the whole pyramid of the benchmark scene (CPU, oracle pyramid)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); O = e.load_oracle()
img, _ = pkg.synth.frame_pair(640, 480, 0, 0)
orc = O.OrbOracle(1000, 1.2, 8, 20, 7); orc.extract(img)
tot_px = tot_s20 = tot_s7 = 0; iters20 = []; cells_all = 0
for l in range(8):
    v = orc.level_image(l).astype(np.int32); H, W = v.shape
    def surv(th):
        c = v[3:-3, 3:-3]; r0 = v[6:, 3:-3]; r8 = v[:-6, 3:-3]; r4 = v[3:-3, 6:]; r12 = v[3:-3, :-6]
        br = np.minimum(np.maximum(r0, r8), np.maximum(r4, r12)) > c + th
        dk = np.maximum(np.minimum(r0, r8), np.minimum(r4, r12)) < c - th
        m = np.zeros((H, W), bool); m[3:-3, 3:-3] = br | dk; return m
    m20, m7 = surv(20), surv(7)
    x0, x1, y0, y1 = 16 - 3, W - 16 + 3, 16 - 3, H - 16 + 3        # ORBextractor.cc:773-776
    wd, hd = x1 - x0, y1 - y0; nc, nr = wd // 30, hd // 30
    wc, hc = -(-wd // nc), -(-hd // nr)
    n_l = []
    for i in range(nr):
        iy = y0 + i * hc; my = min(iy + hc + 6, y1)
        if iy >= y1 - 3: continue
        for j in range(nc):
            ix = x0 + j * wc; mx = min(ix + wc + 6, x1)
            if ix >= x1 - 6: continue
            n_l.append(int(m20[iy + 3:my - 3, ix + 3:mx - 3].sum()))
    n_l = np.array(n_l); cells_all += len(n_l)
    iters20 += list(np.ceil(n_l / 64))
    print("level %d %dx%d: cells %d, survivors@20 %.1f%% (mean %.0f / cell, mean score rounds %.2f), @7 %.1f%%" % (l, W, H, len(n_l), 100 * m20[16:-16, 16:-16].mean(), n_l.mean(), np.ceil(n_l / 64).mean(), 100 * m7[16:-16, 16:-16].mean()))
print("cells", cells_all, "mean score rounds per cell", np.mean(iters20))
