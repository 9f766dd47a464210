import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
import torch
P = 128
imgs = []
for f in range(P): imgs.extend(synth.frame_pair(640, 480, 0, f % 8))
ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2 * P); cap = ext.default_cap()
d_img = torch.from_numpy(np.stack(imgs)).cuda(); S = 2 * P
d_kp = torch.zeros((S, cap, 7), dtype=torch.float32, device="cuda"); d_desc = torch.zeros((S, cap, 32), dtype=torch.uint8, device="cuda"); d_n = torch.zeros(S, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ext.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=st); torch.cuda.synchronize()
pairs = []
for f in range(P): pairs += [(2 * f, 2 * f + 1), (2 * f, (2 * f + 2) % S), (2 * f + 1, (2 * f + 3) % S)]
n_pairs = len(pairs); d_pairs = torch.tensor(pairs, dtype=torch.int32, device="cuda")
d_match = torch.zeros((n_pairs, cap), dtype=torch.int32, device="cuda"); d_nm = torch.zeros(n_pairs, dtype=torch.int32, device="cuda")
d_b = torch.zeros((n_pairs, cap), dtype=torch.int32, device="cuda"); d_s = torch.zeros((n_pairs, cap), dtype=torch.int32, device="cuda")
m = pkg.ORBmatcher(0.75, True)
for _ in range(3): m.match_bf_batch_device(d_desc, d_kp, d_n, cap, d_pairs, n_pairs, d_match, d_nm, d_b, d_s, 50, stream=st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m.match_bf_batch_device(d_desc, d_kp, d_n, cap, d_pairs, n_pairs, d_match, d_nm, d_b, d_s, 50, stream=st)
torch.cuda.synchronize(); print("match (knn2 + filter) alone: %.1f us for %d pairs, matches %d" % ((time.perf_counter() - t0) / 20 * 1e6, n_pairs, int(d_nm.sum())))
