"""The opt-in paths of the BA solver (library options, read per call) still give the default path's results: the LM step as one
hipGraph, one step of look-ahead, the ordered download, one stream group, the pose-pair lists built on the main stream instead of the side
stream (bit for bit), and -- other summation orders, to 1e-9 -- the two-launch update + error
evaluation (k_solve_update + k_error<1>, the path of problems with more than 512 poses). Round 5: options instead of environment switches,
so every variant runs in THIS process."""
import hashlib
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(pkg, synth):
    probs = [synth.ba_problem(n_poses=12, n_fixed=3, n_points=150, obs_per_point=5, seed=3),
             synth.ba_problem(n_poses=30, n_fixed=6, n_points=600, obs_per_point=8, seed=4),
             synth.ba_problem(seed=42)]
    out = []
    h = hashlib.sha1()
    for pb in probs:
        r = pkg.Optimizer.LocalBundleAdjustment(pb)
        for k in ("poses", "points", "edge_outlier", "edge_level1", "edge_chi2", "chi2_trace"):
            h.update(np.ascontiguousarray(r[k]).tobytes())
        out.append(dict(t=r["poses"][:, :3].copy(), iters=list(r["n_iters"]), trials=list(r["n_trials"])))
    rb = pkg.Optimizer.LocalBundleAdjustmentBatch([pkg.Optimizer.prepare(q) for q in probs] * 2)
    for x in rb:
        for k in ("poses", "points", "edge_outlier", "edge_level1", "edge_chi2", "chi2_trace"):
            h.update(np.ascontiguousarray(x[k]).tobytes())
    g = pkg.Optimizer.BundleAdjustment(probs[0], 5, True)
    h.update(np.ascontiguousarray(g["poses"]).tobytes())
    return dict(digest=h.hexdigest(), solves=out)


def test_ba_opt_in_paths_equal_default(pkg, synth):
    base = _run(pkg, synth)
    # (round 6: the four-launch step DCS_BA_FRONT -- k_front + the linearising trial kernel, bit-identical and slower -- was deleted with its option)
    for extra in ({"DCS_BA_GRAPH": 1}, {"DCS_BA_LOOKAHEAD": 1}, {"DCS_BA_LOOKAHEAD": 2}, {"DCS_BA_LOOKAHEAD": 3}, {"DCS_BA_DL_STREAM": 0}, {"DCS_BA_GROUPS": 1}, {"DCS_BA_GROUPS": 4}, {"DCS_BA_PAIRS_SIDE": 0},
                  {"DCS_BA_SCHUR_WAVE": 0}, {"DCS_BA_SCHUR_WAVE": 2}, {"DCS_BA_SCHUR_WAVE": 2, "DCS_BA_GRAPH": 1}):
        with pkg.abi.options(**extra):
            got = _run(pkg, synth)
        assert got["digest"] == base["digest"], extra
    for extra in ({"DCS_BA_FUSED_UPDATE": 0}, {"DCS_BA_FUSED_UPDATE": 0, "DCS_BA_GRAPH": 1}):
        with pkg.abi.options(**extra):
            other = _run(pkg, synth)
        for a, b in zip(other["solves"], base["solves"]):
            assert a["iters"] == b["iters"] and a["trials"] == b["trials"], extra
            assert np.abs(a["t"] - b["t"]).max() < 1e-9, extra
    assert _run(pkg, synth)["digest"] == base["digest"]           # and the defaults are back
