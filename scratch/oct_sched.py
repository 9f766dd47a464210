"""schedule of every k_octree_hist task (side build -DDCS_OCT_PROF_ALL, DCS_LIB_PATH): start / end per workgroup on the 100 MHz wall clock"""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
import torch
B = 512; L = 8
imgs = []
for f in range(B // 2): imgs.extend(synth.frame_pair(640, 480, 0, f % 8))
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=B)
d_img = torch.from_numpy(np.stack(imgs)).cuda(); cap = ext.default_cap()
d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): ext.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=s)
torch.cuda.synchronize()
out = (C.c_longlong * (4 * 8192))()
assert pkg.abi.lib().dcs_debug_oct_all(out) == 0
a = np.array(out[:4 * B * L], dtype=np.int64).reshape(B * L, 4)
t0 = a[:, 0].min()
st = (a[:, 0] - t0) / 100.0; en = (a[:, 1] - t0) / 100.0          # us
print("kernel span %.1f us; sum of task durations %.0f us = %.1f us per CU slot (1024 slots)" % (en.max(), (en - st).sum(), (en - st).sum() / 1024))
for l in range(L):
    m = slice(l * B, (l + 1) * B)
    print("level %d: duration median %.1f max %.1f us; starts %.1f .. %.1f; ends %.1f .. %.1f" % (l, np.median(en[m] - st[m]), (en[m] - st[m]).max(), st[m].min(), st[m].max(), en[m].min(), en[m].max()))
xcc = a[:, 3] & 0xF
for x in range(8):
    m = xcc == x
    print("xcc %d: %d tasks, busy until %.1f us, levels %s" % (x, m.sum(), en[m].max() if m.any() else 0, np.bincount(np.arange(B * L)[m] // B, minlength=L).tolist()))
# resident tasks over time
ts = np.arange(0, en.max(), 5.0)
print("resident:", [int(((st <= t) & (en > t)).sum()) for t in ts])
