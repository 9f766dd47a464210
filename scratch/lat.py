import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_pkg
pkg = load_pkg(); synth = pkg.synth
a, b = synth.frame_pair(640, 480, 0, 0)
e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
m = pkg.ORBmatcher(0.75, True)
def t(fn, reps=100):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
kps, descs = e.extract_batch([a, b])
print("extract only      %.3f ms" % t(lambda: e.extract_batch([a, b])))
print("match only        %.3f ms" % t(lambda: m.match_bf(descs[0], kps[0], descs[1], kps[1], 50)))
def both():
    k, d = e.extract_batch([a, b]); m.match_bf(d[0], k[0], d[1], k[1], 50)
print("extract + match   %.3f ms" % t(both))
print("extract only      %.3f ms" % t(lambda: e.extract_batch([a, b])))
