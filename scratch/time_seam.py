"""dcs_search_by_projection / dcs_search_by_bow / dcs_match_bf through host buffers: median time per call (the seams a drop-in integration calls per search)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
fr_, q_ = synth.projection_problem(n_per_cam=1000, n_queries=800, seed=13)
fr_["grid_off"], fr_["grid_idx"] = pkg.frame_grid(fr_["cam_off"], fr_["kp_x"], fr_["kp_y"], fr_["min_x"], fr_["min_y"], fr_["grid_w_inv"], fr_["grid_h_inv"])
m1 = pkg.ORBmatcher(0.8, False)
def per_call(fn, reps=60):
    for _ in range(5): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[reps // 2] * 1e3, 4)
print("dcs_search_by_projection_ms", per_call(lambda: m1.SearchByProjection(fr_, q_, 100, use_ratio=True, check_orientation=False)))
