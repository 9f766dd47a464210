"""Times the BoW transform on the GPU (full-size synthetic vocabulary, k = 10, L = 6) vs the oracle. usage: time_bow.py [n_images] [per_image]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle(); synth = pkg.synth
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cap = 1096
v = synth.vocabulary_fast(10, 6, seed=3)
G = pkg.ORBVocabulary(v["k"], v["L"], v["parent"], v["is_leaf"], v["desc"], v["weight"])
O = oracle.Vocabulary(v["k"], v["L"], v["parent"], v["is_leaf"], v["desc"], v["weight"])
rng = np.random.default_rng(0)
base = np.concatenate([synth.descriptors_near_words(v, N * 8 - N * 2, seed=1, flip=20), synth.random_descriptors(N * 2, seed=2)])
d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
for i in range(B):
    d_desc[i, :N] = torch.from_numpy(base[rng.permutation(len(base))[:N]]).cuda()
d_n = torch.full((B,), N, dtype=torch.int32, device="cuda")
out = pkg.ORBVocabulary.bow_buffers(B, cap)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    G.transform_device(d_desc, d_n, cap, out, 4, st)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
ev0.record()
for _ in range(reps):
    G.transform_device(d_desc, d_n, cap, out, 4, st)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / reps
print("GPU transform: %d images x %d descriptors: %.3f ms -> %.1f M descriptors/s, %.0f images/s" % (B, N, ms, B * N / ms / 1e3, B / ms * 1e3))
d0 = d_desc[0, :N].cpu().numpy()
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 3.0:
    O.transform(d0, 4); k += 1
cpu = (time.perf_counter() - t0) / k
print("CPU oracle: %.3f ms per image of %d -> %.2f M descriptors/s (1 thread); GPU/CPU = %.0fx" % (cpu * 1e3, N, N / cpu / 1e6, (B * N / ms / 1e3) / (N / cpu / 1e6)))
