"""The opt-in paths of the BA solver (read from the environment once per process) still give the default path's results:
the LM step as one hipGraph, one step of look-ahead, the ordered download, one stream group, the pose-pair lists built on the main
stream instead of the side stream (bit for bit), and -- other summation orders, to 1e-9 -- the column-by-column VALU factorisation and
the two-launch update + error evaluation (k_solve_update + k_error<1>, the path of problems with more than 512 poses)."""
import json, os, subprocess, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, json, hashlib, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as e
pkg = e.load_package(); synth = pkg.synth
probs = [synth.ba_problem(n_poses=12, n_fixed=3, n_points=150, obs_per_point=5, seed=3),
         synth.ba_problem(n_poses=30, n_fixed=6, n_points=600, obs_per_point=8, seed=4),
         synth.ba_problem(seed=42)]
out = []
h = hashlib.sha1()
for pb in probs:
    r = pkg.Optimizer.LocalBundleAdjustment(pb)
    for k in ("poses", "points", "edge_outlier", "edge_level1", "edge_chi2", "chi2_trace"): h.update(np.ascontiguousarray(r[k]).tobytes())
    out.append(dict(t=r["poses"][:, :3].tolist(), iters=list(r["n_iters"]), trials=list(r["n_trials"])))
rb = pkg.Optimizer.LocalBundleAdjustmentBatch([pkg.Optimizer.prepare(q) for q in probs] * 2)
for x in rb:
    for k in ("poses", "points", "edge_outlier", "edge_level1", "edge_chi2", "chi2_trace"): h.update(np.ascontiguousarray(x[k]).tobytes())
g = pkg.Optimizer.BundleAdjustment(probs[0], 5, True)
h.update(np.ascontiguousarray(g["poses"]).tobytes())
print("RESULT " + json.dumps(dict(digest=h.hexdigest(), solves=out)))
""" % ROOT


def _run(extra):
    env = dict(os.environ); env.update(extra)
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_ba_opt_in_paths_equal_default():
    base = _run({})
    for extra in ({"DCS_BA_GRAPH": "1"}, {"DCS_BA_LOOKAHEAD": "1"}, {"DCS_BA_DL_STREAM": "0"}, {"DCS_BA_GROUPS": "1"}, {"DCS_BA_GROUPS": "4"},
                  {"DCS_BA_PAIRS_SIDE": "0"}):
        got = _run(extra)
        assert got["digest"] == base["digest"], extra
    for extra in ({"DCS_BA_LDLT_VALU": "1"}, {"DCS_BA_FUSED_UPDATE": "0"}, {"DCS_BA_FUSED_UPDATE": "0", "DCS_BA_GRAPH": "1"}):
        other = _run(extra)
        for a, b in zip(other["solves"], base["solves"]):
            assert a["iters"] == b["iters"] and a["trials"] == b["trials"], extra
            assert np.abs(np.array(a["t"]) - np.array(b["t"])).max() < 1e-9, extra
