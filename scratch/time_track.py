"""the per-frame tracking chain (dcs_track_local_map) at batch 1 / 16: wall time per frame; DCS_POSE_FAST=0 selects the round-4 k_pose_opt
usage: time_track.py [n_points] [n_features] [reps]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nfe = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
fr, prm = synth.tracking_problem(n_frames=16, n_points=npts, n_features=nfe, seed=23)
for f in fr:
    ft = f["features"]
    ft["grid_off"], ft["grid_idx"] = pkg.frame_grid(ft["cam_off"], ft["kp_x"], ft["kp_y"], ft["min_x"], ft["min_y"], ft["grid_w_inv"], ft["grid_h_inv"])
def med(fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[reps // 2]
out = {}
for nf in (1, 16):
    pt = pkg.abi.PreparedTracking(fr[:nf], prm)
    r = pt.track()
    out[nf] = round(med(pt.track) / nf * 1e3, 4)
    if nf == 1: edges = int((r[0]["point_of_feature"] != -1).sum())
print("DCS_POSE_FAST=%s DCS_POSE_EXACT_EDGE=%s edges(frame 0)=%d ms per frame: batch 1 %.4f, batch 16 %.4f" % (os.environ.get("DCS_POSE_FAST", "1"), os.environ.get("DCS_POSE_EXACT_EDGE", "0"), edges, out[1], out[16]))
