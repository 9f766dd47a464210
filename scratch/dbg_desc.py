import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_pkg
import importlib.util
pkg = load_pkg()
spec = importlib.util.spec_from_file_location('oracle', 'oracle/oracle.py'); o = importlib.util.module_from_spec(spec); spec.loader.exec_module(o)
import importlib
synth = importlib.import_module('orb-slam2-dualcam_amd.synth')
img0, img1 = synth.frame_pair(640, 480, 0, 0)
e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=4)
kps, descs = e.extract_batch([img0, img1])
okp, od = o.OrbOracle(1000, 1.2, 8, 20, 7).extract(img0)
kp = kps[0]; d = descs[0]
print(len(kp), len(okp), kp.dtype)
n = min(len(kp), len(okp))
for f in kp.dtype.names:
    bad = np.nonzero(kp[f][:n] != okp[f][:n])[0]
    print(f, len(bad), bad[:10])
bad = np.nonzero((d[:n] != od[:n]).any(axis=1))[0]
print("desc rows differing", len(bad), bad[:20])
if len(bad):
    i = bad[0]; print(kp[i], okp[i]); print(d[i]); print(od[i])
