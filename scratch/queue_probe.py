"""which streams share a hardware queue (dcs_streams_share_queue), pooled torch streams vs dcs_stream_create_apart"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
A = pkg.abi
torch.cuda.init(); torch.zeros(1, device="cuda")
ss = [torch.cuda.Stream() for _ in range(6)]
raws = [0] + [s.cuda_stream for s in ss]
print("pooled torch streams (+ the legacy default stream first): share matrix")
for a in raws:
    print(" ".join("X" if A.streams_share_queue(a, b) else "." for b in raws))
got = []
for i in range(5):
    raw, ok = A.stream_apart([0] + got)
    got.append(raw)
    print("apart stream", i, "apart from default +", i, "earlier:", ok)
al = [0] + got
for a in al:
    print(" ".join("X" if A.streams_share_queue(a, b) else "." for b in al))
import time
t0 = time.perf_counter(); A.streams_share_queue(got[0], got[1]); print("one probe: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
