import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
import torch
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
imgs = []
for f in range(P): imgs.extend(synth.frame_pair(640, 480, 0, f % 8))
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2 * P); cap = ext.default_cap()
d_img = torch.from_numpy(np.stack(imgs)).cuda(); S = 2 * P
d_kp = torch.zeros((S, cap, 7), dtype=torch.float32, device="cuda"); d_desc = torch.zeros((S, cap, 32), dtype=torch.uint8, device="cuda"); d_n = torch.zeros(S, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): ext.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ext.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e6
tot, n = ext.timing_totals()
print("extract alone: %.1f us per %d images; per call:" % (dt, S), {k: round(v / max(n, 1), 1) for k, v in tot.items()})
