"""one synth.ba_problem(**kwargs) on the GPU against the oracle: usage ba_case.py "dict(n_poses=21, ...)" """
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle(); synth = pkg.synth
args = eval(sys.argv[1])
pb = synth.ba_problem(**args)
prob = dict(pb); prob["cams"] = [oracle.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
got, exp = pkg.Optimizer.LocalBundleAdjustment(pb), oracle.ba_local(prob)
k = sum(exp["n_iters"])
print("fused=%s dt %.3e iters %s/%s trials %s/%s chi2 rel %s" % (os.environ.get("DCS_BA_FUSED_UPDATE", "1"), np.abs(got["poses"][:, :3] - exp["poses"][:, :3]).max(), got["n_iters"], exp["n_iters"], got["n_trials"], exp["n_trials"],
      np.array2string(np.abs(np.array(got["chi2_trace"][:k]) / np.array(exp["chi2_trace"][:k]) - 1), precision=1)))
