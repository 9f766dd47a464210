"""N > 1 path on CPU: world_size-2 gloo processes shard dual frames, all-gather their newest features with the
same pack / all-gather / unpack code bench.py runs over RCCL, and every rank reproduces the cross-rank
relocalisation match (checked with the oracle, because the product matcher has no CPU path)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_pkg
    import oracle as O
    pkg = load_pkg()
    from orb_slam2_dualcam_amd import sharding
    synth = pkg.synth
    cap = 400
    # unit sharding: 5 dual frames over 2 ranks
    mine = sharding.shard_units(5, rank, world)
    assert mine == list(range(rank, 5, world))
    # each rank extracts (with the oracle) the newest dual frame of its own stream
    img0, img1 = synth.frame_pair(640, 480, rank, 0)
    crop = (slice(100, 340), slice(200, 520))
    feats = [O.OrbOracle(300, 1.2, 8, 20, 7).extract(np.ascontiguousarray(im[crop])) for im in (img0, img1)]
    kp = torch.zeros((2, cap, 7), dtype=torch.float32)
    desc = torch.zeros((2, cap, 32), dtype=torch.uint8)
    n = torch.zeros(2, dtype=torch.int32)
    for c, (k, d) in enumerate(feats):
        n[c] = len(k)
        kp[c, :len(k)] = torch.from_numpy(k.view(np.float32).reshape(-1, 7).copy())
        desc[c, :len(k)] = torch.from_numpy(d)
    g_kp, g_desc, g_n = sharding.allgather_features(kp, desc, n, cap)
    assert g_kp.shape == (2 * world, cap, 7) and g_desc.shape == (2 * world, cap, 32)
    # own slots come back bit-identical; pack/unpack round trip
    assert torch.equal(g_kp[2 * rank:2 * rank + 2].view(torch.int32), kp.view(torch.int32))
    assert torch.equal(g_desc[2 * rank:2 * rank + 2], desc) and torch.equal(g_n[2 * rank:2 * rank + 2], n)
    k2, d2, n2 = sharding.unpack_features(sharding.pack_features(kp, desc, n, cap), cap)
    assert torch.equal(k2.view(torch.int32), kp.view(torch.int32)) and torch.equal(d2, desc) and torch.equal(n2, n)
    # cross-rank relocalisation match: my cam0 vs the other rank's cam1
    res = {}
    for (qs, ts) in sharding.reloc_pairs(rank, world):
        nq, nt = int(g_n[qs]), int(g_n[ts])
        dq, dt = g_desc[qs, :nq].numpy(), g_desc[ts, :nt].numpy()
        aq = g_kp[qs, :nq, 3].numpy().copy(); at = g_kp[ts, :nt, 3].numpy().copy()
        bi, bd, sd = O.knn2(dq, dt)
        m, cnt = O.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, aq, at)
        res[(qs, ts)] = (m.copy(), cnt)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), n=g_n.numpy(), desc_sum=g_desc.to(torch.int64).sum().item(),
             pairs=np.array(list(res.keys())), counts=np.array([v[1] for v in res.values()]))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_and_cross_rank_match_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["n"], r1["n"]) and r0["desc_sum"] == r1["desc_sum"]      # every rank holds the same gathered set
    assert r0["n"].min() > 250
    assert r0["pairs"].tolist() == [[0, 3]] and r1["pairs"].tolist() == [[2, 1]]


def test_sharding_helpers():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    load_pkg()
    from orb_slam2_dualcam_amd import sharding
    assert sharding.shard_units(8, 3, 8) == [3] and sharding.shard_units(10, 1, 4) == [1, 5, 9]
    units = sorted(u for r in range(8) for u in sharding.shard_units(19, r, 8))
    assert units == list(range(19))                      # a partition: every unit exactly once
    assert sharding.record_bytes(1096) == 1096 * 60 + 64
    assert sharding.reloc_pairs(2, 4) == [(4, 1), (4, 3), (4, 7)]
