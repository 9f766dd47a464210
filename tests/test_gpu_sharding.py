"""GPU side of the multi-GPU path on the single GPU of the test box: the feature wire format packed /
unpacked on the device, the cross-rank relocalisation match over the gathered slot array (two ranks emulated on one
GPU), and the all-gather itself through RCCL with world size 1. The world-2 exchange runs on CPU/gloo in
tests/test_sharding_gloo.py; N = 2, 4, 8 on real GPUs is the driver's scaling bench."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _extract_rank(pkg, synth, rank, cap, torch):
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
    imgs = synth.frame_pair(640, 480, rank, 0)
    d_img = torch.from_numpy(np.stack(imgs)).cuda()
    d_kp = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(2, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    e.close()
    return imgs, d_kp, d_desc, d_n


def test_two_ranks_emulated_on_one_gpu(pkg, oracle, synth):
    import torch
    from orb_slam2_dualcam_amd import sharding
    world, cap = 2, 1096
    ranks = [_extract_rank(pkg, synth, r, cap, torch) for r in range(world)]
    sends = [sharding.pack_features(kp, desc, n, cap) for (_, kp, desc, n) in ranks]
    assert sends[0].shape == (2, sharding.record_bytes(cap)) and sends[0].is_cuda
    g_recv = torch.cat(sends, 0)                              # what all_gather_into_tensor delivers: rank-major concatenation
    g_kp = torch.zeros((2 * world, cap, 7), dtype=torch.float32, device="cuda")
    g_desc = torch.zeros((2 * world, cap, 32), dtype=torch.uint8, device="cuda")
    g_n = torch.zeros(2 * world, dtype=torch.int32, device="cuda")
    sharding.unpack_features(g_recv, cap, g_kp, g_desc, g_n)
    for r, (_, kp, desc, n) in enumerate(ranks):              # bit-exact round trip, NaN-safe (compare the bytes)
        assert torch.equal(g_kp[2 * r:2 * r + 2].view(torch.int32), kp.view(torch.int32))
        assert torch.equal(g_desc[2 * r:2 * r + 2], desc) and torch.equal(g_n[2 * r:2 * r + 2], n)
    feats = [[oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(im) for im in imgs] for (imgs, _, _, _) in ranks]
    for rank in range(world):
        pairs = sharding.reloc_pairs(rank, world)
        assert pairs == [(2 * rank, 2 * (1 - rank) + 1)]
        x_pairs = torch.tensor(pairs, dtype=torch.int32, device="cuda")
        x_match = torch.zeros((world - 1, cap), dtype=torch.int32, device="cuda")
        x_nm = torch.zeros(world - 1, dtype=torch.int32, device="cuda")
        x_b = torch.zeros((world - 1, cap), dtype=torch.int32, device="cuda")
        x_s = torch.zeros((world - 1, cap), dtype=torch.int32, device="cuda")
        pkg.ORBmatcher(0.75, True).match_bf_batch_device(g_desc, g_kp, g_n, cap, x_pairs, world - 1, x_match, x_nm, x_b, x_s, 50,
                                                         stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        (kq, dq), (kt, dt) = feats[rank][0], feats[1 - rank][1]
        bi, bd, sd = oracle.knn2(dq, dt)
        m, n = oracle.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, kq["angle"], kt["angle"])
        assert int(x_nm[0]) == n and np.array_equal(x_match[0, :len(dq)].cpu().numpy(), m)


def test_allgather_through_rccl_world_1(pkg, synth):
    import torch
    import torch.distributed as dist
    from orb_slam2_dualcam_amd import sharding
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cap = 1096
        _, kp, desc, n = _extract_rank(pkg, synth, 0, cap, torch)
        g_kp, g_desc, g_n = sharding.allgather_features(kp, desc, n, cap)
        torch.cuda.synchronize()
        assert torch.equal(g_kp.view(torch.int32), kp.view(torch.int32)) and torch.equal(g_desc, desc) and torch.equal(g_n, n)
    finally:
        dist.destroy_process_group()


def test_cabi_allgather_world_1(pkg, synth):
    """dcs_comm_* + dcs_features_allgather (the exchange a C++ host owns, include/dcs_abi.h): RCCL bound by the library, world
    size 1 on the test box -- the slot arrays come back unchanged, in place, no packing."""
    import torch
    cap = 1096
    _, kp, desc, n = _extract_rank(pkg, synth, 0, cap, torch)
    comm = pkg.abi.FeatureComm(pkg.abi.FeatureComm.unique_id(), 0, 1)
    g_kp, g_desc, g_n = torch.zeros_like(kp), torch.zeros_like(desc), torch.zeros_like(n)
    for _ in range(2):
        comm.allgather_features(kp, desc, n, cap, g_kp, g_desc, g_n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(g_kp.view(torch.int32), kp.view(torch.int32)) and torch.equal(g_desc, desc) and torch.equal(g_n, n)
    assert int(g_n.min()) > 900
    comm.close()
    with pytest.raises(pkg.DcsError):
        pkg.abi.FeatureComm(pkg.abi.FeatureComm.unique_id(), 3, 2)
