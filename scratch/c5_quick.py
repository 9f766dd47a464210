# C5 on one GPU, time-sliced only: front end alone / BA alone / both (environment switches are read by the library)
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
import __graft_entry__ as e
pkg = e.load_package()
dev = torch.device("cuda", 0)
class A: pass
orig = bench.run_c5
import types
src = bench.run_c5
# run only the time-sliced configuration
def sliced_only():
    g = bench.run_c5.__globals__
    return None
out = bench.run_c5(pkg, torch, dev, 0, A())
print(json.dumps({k: out[k] for k in ("alone", "concurrent", "time_sliced_vs_alone")}))
