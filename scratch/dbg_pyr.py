import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle(); synth = pkg.synth
for (w, h, B) in [(640, 480, 2), (640, 480, 4), (1280, 720, 2), (333, 245, 2)]:
    imgs = list(synth.frame_pair(w, h, 0, 0))
    ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=B)
    for rep in range(2):
        ext.extract_batch(imgs)
        for i in range(2):
            o = oracle.OrbOracle(1000, 1.2, 8, 20, 7); o.extract(imgs[i])
            for l in range(8):
                a, b = ext.level_image(i, l), o.level_image(l)
                bad = np.argwhere(a != b)
                if len(bad):
                    print(w, h, B, "rep", rep, "img", i, "lvl", l, a.shape, len(bad), "rows", np.unique(bad[:, 0])[:10], "cols", np.unique(bad[:, 1])[:10], "n_cols", len(np.unique(bad[:, 1])))
    ext.close()
print("done")
