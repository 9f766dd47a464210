"""C3 shape (dual 1280x720, 2000 feat/cam, 64 dual frames per step) over 1 / 2 extraction lanes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, bench
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
for lanes in (1, 2, 1, 2):
    p3 = bench.Pipeline(pkg, torch, dev, 0, 1280, 720, 2000, 1, 64, lanes, 0, 16)
    for e_ in p3.exts:
        e_.set_timing(0)
    dt = p3.run(30, 3)
    print("C3 lanes", lanes, round(p3.features_per_step() * 30 / dt / 1e3, 1), "kfeatures/s", round(dt / 30 * 1e3, 3), "ms/step")
    p3.close()
