import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import oracle
pkg = load_pkg(); oracle.build(); oracle.lib(); synth = pkg.synth
frame, q = synth.projection_problem(n_per_cam=2000, n_queries=1500, seed=2)
off, idx = pkg.frame_grid(frame["cam_off"], frame["kp_x"], frame["kp_y"], frame["min_x"], frame["min_y"], frame["grid_w_inv"], frame["grid_h_inv"])
frame["grid_off"], frame["grid_idx"] = off, idx
m = pkg.ORBmatcher(0.8, True)
m.SearchByProjection(frame, q)
t0 = time.perf_counter()
for _ in range(20): r = m.SearchByProjection(frame, q)
print("gpu host-API ms", (time.perf_counter() - t0) / 20 * 1e3, "matches", r[2])
t0 = time.perf_counter()
for _ in range(20): r = oracle.search_by_projection(frame, q, 100, 0.8, False)
print("oracle ms", (time.perf_counter() - t0) / 20 * 1e3)
