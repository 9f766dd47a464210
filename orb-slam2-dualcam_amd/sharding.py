"""Frame-pair sharding across GPUs and the descriptor all-gather.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the MI355X node, "gloo" in the CPU
tests). Units of work are independent: dual frame / stream i runs on rank i mod world with no data-path
collective. The only exchange is the all-gather of the newest dual frame's features so that every rank can run
cross-camera relocalisation matching against features held on other GPUs (the analogue of
SearchByBoWCrossCam(curFrame, camS, KF, CAP), reference Tracking.cc:822).

Wire format: fixed-capacity slots so a plain equal-count all-gather works -- per camera slot
    [cap x 28 B keypoints | cap x 32 B descriptors | int32 count | pad to 64 B]
(2 cameras x (1096 x 60 + 64) B = 131.6 KB per rank at the default capacity: latency-bound on xGMI, which is why one
single-step all-gather is used instead of ring-pipelined chunks.)
"""
import torch
import torch.distributed as dist

KP_BYTES, DESC_BYTES = 28, 32


def shard_units(n_units, rank, world):
    """Indices of the dual frames / streams owned by `rank` (unit i -> rank i mod world)."""
    return list(range(rank, n_units, world))


def record_bytes(cap):
    return cap * (KP_BYTES + DESC_BYTES) + 64


def pack_features(kp, desc, n, cap, out=None):
    """kp [S, cap, 7] float32, desc [S, cap, 32] uint8, n [S] int32 -> uint8 [S, record_bytes(cap)]."""
    S = kp.shape[0]
    rec = record_bytes(cap)
    if out is None:
        out = torch.zeros((S, rec), dtype=torch.uint8, device=kp.device)
    out[:, :cap * KP_BYTES] = kp.reshape(S, -1).view(torch.uint8)
    out[:, cap * KP_BYTES:cap * (KP_BYTES + DESC_BYTES)] = desc.reshape(S, -1)
    out[:, cap * (KP_BYTES + DESC_BYTES):cap * (KP_BYTES + DESC_BYTES) + 4] = n.contiguous().view(torch.uint8).reshape(S, 4)
    return out


def unpack_features(buf, cap, kp=None, desc=None, n=None):
    """inverse of pack_features for a [T, record_bytes(cap)] buffer; optional preallocated outputs."""
    T = buf.shape[0]
    k = buf[:, :cap * KP_BYTES].contiguous().view(torch.float32).reshape(T, cap, 7)
    d = buf[:, cap * KP_BYTES:cap * (KP_BYTES + DESC_BYTES)].reshape(T, cap, 32)
    c = buf[:, cap * (KP_BYTES + DESC_BYTES):cap * (KP_BYTES + DESC_BYTES) + 4].contiguous().view(torch.int32).reshape(T)
    if kp is not None:
        kp.copy_(k); desc.copy_(d); n.copy_(c)
        return kp, desc, n
    return k, d.contiguous(), c


def allgather_features(kp, desc, n, cap, send=None, recv=None, group=None):
    """All ranks contribute S feature slots; returns (kp [W*S, cap, 7], desc [W*S, cap, 32], n [W*S]) ordered by rank.
    `send` / `recv` are optional preallocated uint8 staging buffers ([S, rec] and [W*S, rec])."""
    world = dist.get_world_size(group)
    S = kp.shape[0]
    send = pack_features(kp, desc, n, cap, send)
    if recv is None:
        recv = torch.empty((world * S, send.shape[1]), dtype=torch.uint8, device=kp.device)
    dist.all_gather_into_tensor(recv.view(world * S, -1), send, group=group)     # concatenation along dim 0, rank-major
    return unpack_features(recv.view(world * S, -1), cap)


def reloc_pairs(rank, world, slots_per_rank=2, query_cam=0, train_cam=1):
    """(query slot, train slot) pairs of the cross-GPU relocalisation match in the gathered slot array:
    this rank's `query_cam` against every OTHER rank's `train_cam`."""
    return [(rank * slots_per_rank + query_cam, r * slots_per_rank + train_cam) for r in range(world) if r != rank]
