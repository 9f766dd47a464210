import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import importlib.util
pkg = load_pkg()
spec = importlib.util.spec_from_file_location('oracle', 'oracle/oracle.py'); o = importlib.util.module_from_spec(spec); spec.loader.exec_module(o)
def deep_image(seed, npatch=9, size=10):
    rng = np.random.default_rng(seed)
    img = np.full((480, 640), 120, np.uint8)
    for k in range(1, npatch + 1):
        x = int(30 + 560 * (1 - 2.0 ** -k)); y = int(30 + 400 * (1 - 2.0 ** -k))
        s = max(4, size - k // 2)
        img[y:y + s, x:x + s] = rng.integers(0, 256, (s, s), dtype=np.uint8)
    return img
for seed in range(4):
    for variant in range(3):
        img = deep_image(seed, 9 + variant, 12)
        if variant == 2:
            img[::, :] = np.where(np.random.default_rng(seed).random((480, 640)) < 0.002, 255, img)
        e = pkg.ORBextractor(1000, 1.2, 8, 12, 5, max_images=1)
        kp, desc = e(img)
        fb = e.quadtree_fallbacks()
        okp, od = o.OrbOracle(1000, 1.2, 8, 12, 5).extract(img)
        print(seed, variant, len(kp), len(okp), "fallbacks", fb, "same", kp.tobytes() == okp.tobytes() and np.array_equal(desc, od))
        e.close()
