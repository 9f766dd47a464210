"""GPU box: N single C4 solves (for rocprofv3 --kernel-trace --stats)."""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb-slam2-dualcam_amd")
prep = pkg.Optimizer.prepare(pkg.synth.ba_problem())
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30): prep.solve()
