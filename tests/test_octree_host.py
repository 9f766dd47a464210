"""The product's sort/scan quadtree (csrc/octree.cpp) against the oracle's literal std::list restatement
of DistributeOctTree (ORBextractor.cc:539-763). Pure host code: runs without a GPU through the C ABI."""
import numpy as np
import pytest


def _rand_cands(rng, n, w, h, clustered):
    pts = set()
    while len(pts) < n:
        if clustered and rng.random() < 0.7:
            cx, cy = rng.integers(3, w - 3), rng.integers(3, h - 3)
            x, y = int(np.clip(cx + rng.normal(0, 6), 3, w - 4)), int(np.clip(cy + rng.normal(0, 6), 3, h - 4))
        else:
            x, y = int(rng.integers(3, w - 3)), int(rng.integers(3, h - 3))
        pts.add((x, y))
    order = list(pts)
    rng.shuffle(order)
    return [(x, y, int(rng.integers(7, 60))) for x, y in order]


@pytest.mark.parametrize("w,h", [(608, 448), (1248, 688), (147, 102), (448, 608), (300, 300)])
def test_octree_matches_oracle_random(pkg, oracle, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    for trial in range(30):
        n = int(rng.integers(1, 1500))
        N = int(rng.integers(1, 500))
        c = np.array(_rand_cands(rng, n, w, h, trial % 2 == 0), oracle.CANDIDATE)
        exp = oracle.distribute_octree(c, 16, 16 + w, 16, 16 + h, N)
        got = pkg.abi.distribute_octree(c, 16, 16 + w, 16, 16 + h, N)
        assert got.tobytes() == exp.tobytes(), (w, h, n, N, trial)


def test_octree_matches_oracle_on_real_candidates(pkg, oracle, synth):
    img, _ = synth.frame_pair(640, 480, 0, 0)
    for nfeat in (200, 1000, 3000):
        e = oracle.OrbOracle(nfeat, 1.2, 8, 20, 7)
        e.extract(img, cap=8000)
        t = e.tables()
        for l in range(8):
            w, h = e.level_dims(l)
            c = e.level_candidates(l)
            exp = oracle.distribute_octree(c, 16, w - 16, 16, h - 16, int(t["n_per_level"][l]))
            got = pkg.abi.distribute_octree(c, 16, w - 16, 16, h - 16, int(t["n_per_level"][l]))
            assert got.tobytes() == exp.tobytes(), (nfeat, l)


def test_octree_edge_cases(pkg, oracle):
    C = oracle.CANDIDATE
    assert len(pkg.abi.distribute_octree(np.zeros(0, C), 16, 624, 16, 464, 100)) == 0
    one = np.array([(5, 5, 9)], C)
    assert pkg.abi.distribute_octree(one, 16, 624, 16, 464, 100).tobytes() == one.tobytes()
    # equal scores everywhere: first max (original order) wins, identical to the oracle
    rng = np.random.default_rng(4)
    c = np.array([(x, y, 20) for x, y, _ in _rand_cands(rng, 400, 608, 448, True)], C)
    for N in (1, 7, 50, 399, 400, 1000):
        assert pkg.abi.distribute_octree(c, 16, 624, 16, 464, N).tobytes() == oracle.distribute_octree(c, 16, 624, 16, 464, N).tobytes()
