"""Hardware queues: HIP maps a process's streams onto a few hardware queues and two streams on one queue run one after the other.
dcs_streams_share_queue measures it, dcs_stream_create_apart makes streams that provably run side by side (csrc/common.cpp); the
library's own side streams and the solver's streams are created that way (config C5: src/LocalMapping.cc:97-104 beside
src/Tracking.cc:236-269)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_probe_and_apart_streams(pkg):
    import torch
    A = pkg.abi
    torch.zeros(1, device="cuda")
    pooled = [torch.cuda.Stream() for _ in range(8)]                 # more pooled streams than hardware queues: some of them must share one
    raws = [s.cuda_stream for s in pooled]
    assert A.streams_share_queue(raws[0], raws[0])
    share = [[A.streams_share_queue(a, b) for b in raws] for a in raws]
    assert any(share[i][j] for i in range(8) for j in range(8) if i != j)
    for i in range(8):
        for j in range(8):
            assert share[i][j] == share[j][i], (i, j)                # the relation is symmetric (and the probe repeatable)
    # two streams apart from the legacy default stream and from each other: what the matcher / an extraction lane / the solver need
    s1, ok1 = A.stream_apart([0])
    s2, ok2 = A.stream_apart([0, s1])
    assert ok1 and ok2
    assert not A.streams_share_queue(0, s1) and not A.streams_share_queue(0, s2) and not A.streams_share_queue(s1, s2)
    # kernels launched on both really overlap: results of work enqueued on the two streams are both correct afterwards
    x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    with torch.cuda.stream(torch.cuda.ExternalStream(s1)):
        a = (x * 2).sum()
    with torch.cuda.stream(torch.cuda.ExternalStream(s2)):
        b = (x * 3).sum()
    torch.cuda.synchronize()
    assert float(b) == pytest.approx(1.5 * float(a), rel=1e-6)
    # asking for more mutually apart streams than the process has queues is reported, not an error
    got, flags = [s1, s2], []
    for _ in range(6):
        s, ok = A.stream_apart([0] + got)
        got.append(s); flags.append(ok)
    assert not all(flags)
    torch.cuda.synchronize()
    for s in got:
        A.lib().dcs_stream_destroy(s)


def test_solver_streams_keep_off_the_front_end(pkg, synth):
    """dcs_ba_avoid_streams: the solver context built after the call runs on other hardware queues than the named streams; results
    are the same bits as without it."""
    import torch
    A = pkg.abi
    probs = [synth.ba_problem(n_poses=12, n_fixed=3, n_points=300, obs_per_point=4, seed=7 + i) for i in range(3)]     # three problems: two stream groups
    A.ba_release_thread()
    A.ba_avoid_streams([])
    ref = pkg.Optimizer.LocalBundleAdjustmentBatch(probs)
    front, ok = A.stream_apart([0])
    assert ok
    A.ba_release_thread()
    A.ba_avoid_streams([0, front])
    out = pkg.Optimizer.LocalBundleAdjustmentBatch(probs)
    for r, o in zip(ref, out):
        for k in r:
            if isinstance(r[k], np.ndarray):
                assert np.array_equal(r[k], o[k]), k
            elif k != "gpu_ms":                                  # (a wall-clock figure)
                assert r[k] == o[k], k
    A.ba_avoid_streams([])
    A.ba_release_thread()
    torch.cuda.synchronize()
    A.lib().dcs_stream_destroy(front)
