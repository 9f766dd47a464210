"""phase timestamps of one k_octree_hist task (side build -DDCS_OCT_PROF=<task>, DCS_LIB_PATH): runs a 512-image extraction and prints the deltas"""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
import torch
B = 512
imgs = []
for f in range(B // 2): imgs.extend(synth.frame_pair(640, 480, 0, f % 8))
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=B)
d_img = torch.from_numpy(np.stack(imgs)).cuda(); cap = ext.default_cap()
d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): ext.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=s)
torch.cuda.synchronize()
out = (C.c_longlong * 64)()
assert pkg.abi.lib().dcs_debug_oct_prof(out) == 0
t = list(out)
names = ["zero bins", "histogram", "pyramid", "depth 1/0", "decide D", "list L_D", "tail passes", "final sort", "emit"]
print("task n=%d F=%d D=%d tail=%d; total %d ticks" % (t[10], t[11], t[12], t[13], t[9] - t[0]))
for k, nm in enumerate(names): print("  %-12s %7d" % (nm, t[k + 1] - t[k]))
