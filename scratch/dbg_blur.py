"""where the blurred debug levels differ from the oracle's (DCS_ORB_FUSED_BLUR=0 python scratch/dbg_blur.py [w h])"""
import os, sys
os.environ.setdefault("DCS_ORB_FUSED_BLUR", "0")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from conftest import load_pkg
import oracle
oracle.build(); oracle.lib()
pkg = load_pkg()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
rng = np.random.default_rng(3)
imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(3)]
e = pkg.ORBextractor(500, 1.2, 8, 20, 7, max_images=3)
e.extract_batch(imgs)
for i in range(3):
    o = oracle.OrbOracle(500, 1.2, 8, 20, 7); o.extract(imgs[i])
    for l in range(8):
        a, b = e.level_image(i, l, blurred=True), oracle.gauss7_u8(o.level_image(l))
        d = np.argwhere(a != b)
        if len(d):
            y, x = d[0]; print("first", y, x, "gpu", a[y, x], "oracle", b[y, x], "diffs", sorted(set((a.astype(int) - b)[a != b].tolist()))[:12])
            print("img", i, "level", l, a.shape, "mismatches", len(d), "cols", sorted(set(d[:, 1]))[:20], "rows", sorted(set(d[:, 0]))[:10], "..", d[:, 0].max())
print("done")
