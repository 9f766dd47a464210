import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); O = e.load_oracle(); synth = pkg.synth
img0, img1 = synth.frame_pair(640, 480, 0, 0)
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
ext.extract_batch([img0, img1])
o = O.OrbOracle(1000, 1.2, 8, 20, 7); o.extract(img0)
for l in range(1, 4):
    a = ext.level_image(0, l).astype(int); b = o.level_image(l).astype(int)
    d = np.argwhere(a != b)
    print("level", l, a.shape, "mismatches", len(d))
    if len(d):
        ys, xs = d[:, 0], d[:, 1]
        print("  x range", xs.min(), xs.max(), "y range", ys.min(), ys.max(), "unique x mod 4", np.unique(xs % 4), "unique y mod 8", np.unique(ys % 8))
        for (y, x) in d[:6]: print("   ", y, x, a[y, x], b[y, x])
