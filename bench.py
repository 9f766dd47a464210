#!/usr/bin/env python3
"""bench.py -- dual-frame ORB extract + brute-force match throughput (kfeatures/s) on N MI355X, plus the
local-BA iteration rate, with roofline and CPU-baseline legs (see DESIGN.md "Measurement").

A "step" = one pass of the hot path over one HBM-resident batch of `--pairs` consecutive dual 640x480
frames (BASELINE.json configs[1]: 1000 features/camera, 8 levels, scale 1.2, FAST 20/7):
  extract 2*pairs images -> 3 matches per dual frame (cam0<->cam1 at t, cam_c(t)<->cam_c(t-1), knn2 +
  TH_LOW + ratio 0.75 + rotation histogram). kfeatures/s = keypoints returned / wall time / 1000.
N > 1: one process per GPU (torch.distributed, RCCL), each rank owns its own stream of frame pairs
(weak scaling, no data-path collective inside extraction/matching); once per step the ranks all-gather
the newest dual frame's features (the cross-camera relocalisation exchange of the north star) and
match their cam0 against every other rank's cam1.

`python bench.py --gpus N` launches its own N ranks (re-exec under torch.distributed.run) when it is not
already running as a rank (RANK / WORLD_SIZE unset); under an external torchrun it just joins.

Extra legs on rank 0 at N = 1 (each one can be switched off): cpu_baseline (oracle, 1 thread) and
cpu_all_cores, latency (one dual frame through the host-buffer API, PCIe included), with_transfers
(the default batch through the host-buffer API), matcher (solo, i8-MFMA TOPS), c3 (dual 1280x720, 2000
features), local_ba (C4 + its MFMA / HBM roofline + the 8-problem batch), c5_one_gpu (8 streams of
extraction + matching next to 8 concurrent local BAs, time-sliced on ONE GPU), bow, per_frame_chain (SearchLocalPoints +
PoseOptimization per frame: dcs_track_local_map against the three host-buffer calls it replaces).

Under a rank environment (any N, every rank takes part, rank 0 reports node sums) two legs follow the headline:
c3_scaled = BASELINE configs[2] (64 dual 1280x720 / 2000-feature frames per step per rank + the per-step feature
all-gather + cross-rank match) and c5_node = configs[4] (per rank ONE dual-camera sequence on the main thread next
to ONE local-BA loop on a second thread). `--no-rank-legs` skips them.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
I8_MFMA_PEAK_TOPS = 3944.0       # MI355X_MICROARCH.md: v_mfma_i32_16x16x64_i8 dense, measured ceiling
FP4_MFMA_PEAK_TOPS = 7228.0      # cdna_hip_programming.md: v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 operands, measured floor of the 16x16 shape (spec ~10 PF dense)
F64_MFMA_PEAK_GFLOPS = 78600.0   # AMD's public MI355X figure for FP64 matrix (= FP64 vector); the in-container guide has no f64 row
N_CU = 256
F64_VECTOR_PEAK_GFLOPS = 78600.0   # FP64 vector = FP64 matrix on MI355X (public spec): 16 FMA lanes per SIMD per cycle at 2.4 GHz


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=256, help="dual frames per step per GPU")
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip every CPU leg)")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-bow", action="store_true")
    ap.add_argument("--no-c3", action="store_true")
    ap.add_argument("--no-c5", action="store_true")
    ap.add_argument("--no-two-lanes", action="store_true", help="skip the two-extraction-lanes leg")
    ap.add_argument("--no-host-api", action="store_true", help="skip the latency / with_transfers legs")
    ap.add_argument("--stage-timing", choices=("sampled", "all"), default="sampled",
                    help="sampled: only the FAST stage is bracketed by events inside the timed region, the others in extra steps after it; all: every stage inside the timed region")
    ap.add_argument("--lanes", type=int, default=1, help="independent extractor handles/streams the batch is split over (overlaps latency-bound kernels)")
    ap.add_argument("--serial", action="store_true", help="profiling aid: synchronise after the extraction and after the matching of every step (no kernel of one overlaps the other); with DCS_ORB_NO_OVERLAP=1 every kernel runs alone")
    ap.add_argument("--exchange", choices=("cabi", "torch"), default="cabi",
                    help="N > 1 (or any run under a rank environment): the per-step feature all-gather goes through the library's own RCCL communicator "
                         "(dcs_features_allgather, default) or through torch.distributed + pack / unpack")
    ap.add_argument("--input-sets", type=int, default=3, help="distinct HBM-resident input batches rotated across steps (3 x 157 MB > the 256 MB Infinity Cache)")
    ap.add_argument("--no-rank-legs", action="store_true", help="runs under a rank environment: skip the c3_scaled / c5_node legs that follow the headline")
    ap.add_argument("--rank-leg-seconds", type=float, default=1.5, help="length of each window of the c5_node leg (front end alone, then front end next to the solver)")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="CPU check of the N > 1 entry: launcher -> ranks -> gloo process group -> the feature all-gather, then one JSON line")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """--gpus N without a rank environment: become the launcher (one process per GPU under torch.distributed.run)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def selftest_rank(args):
    """The N > 1 path on CPU (gloo): the sharding code bench runs over RCCL up to and including its first collective, then the two
    rank-path legs (c3_scaled, c5_node) with the GPU work replaced by stand-ins of the same SHAPES -- the collectives, the two-thread
    structure of c5_node, the node sums and the JSON keys are the code the GPU run executes."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    entry.load_package()
    from orb_slam2_dualcam_amd import sharding
    cap = 64
    kp = torch.full((2, cap, 7), float(rank), dtype=torch.float32)
    desc = torch.full((2, cap, 32), rank + 1, dtype=torch.uint8)
    n = torch.tensor([10 + rank, 20 + rank], dtype=torch.int32)
    g_kp, g_desc, g_n = sharding.allgather_features(kp, desc, n, cap)
    ok = g_n.tolist() == [v for r in range(world) for v in (10 + r, 20 + r)] and \
        all(int(g_desc[2 * r, 0, 0]) == r + 1 for r in range(world)) and all(q == 2 * rank for q, _ in sharding.reloc_pairs(rank, world))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    env = RankEnv(torch, dist, torch.device("cpu"), rank, world)
    legs = {}
    if not args.no_rank_legs:
        legs["c3_scaled"] = leg_c3_scaled(env, StandInC3(torch, sharding, rank, world), steps=3, warmup=1)
        legs["c5_node"] = leg_c5_node(env, *stand_in_c5(rank), seconds=0.2)
    if rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": world, "backend": "gloo", "allgather_ok": bool(flag.item()),
                          "units_of_rank0": sharding.shard_units(2 * world + 1, 0, world), **legs}))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if flag.item() else 1


# ---------------------------------------------------------------------------------------------------------------- rank-path legs
# What an N-GPU run measures beyond the headline: BASELINE configs[2] ("dual 1280x720, 2000 feat/cam, 1 -> 8 GPUs frame-sharded with RCCL
# descriptor all-gather") and configs[4] ("8 concurrent dual-camera sequences, one per GPU, + per-stream local BA, whole-node throughput").
# Every rank runs the same leg on its own GPU; rank 0 reports per-rank figures and node sums. The GPU work is behind small objects so that
# tests/test_bench_launcher.py can drive the very same collectives / threads / report code at world size 2 on gloo.
C3_FRAMES_PER_STEP = 64
C3_CAP_STAND_IN = 2096                              # slot capacity of a 2000-feature, 8-level handle: nfeatures + 4 x levels + 64 (abi.py default_cap)


class RankEnv:
    def __init__(self, torch, dist, dev, rank, world, distributed=True):
        self.torch, self.dist, self.dev, self.rank, self.world, self.distributed = torch, dist, dev, rank, world, distributed

    def barrier(self):
        if self.distributed:
            self.dist.barrier()

    def gather(self, values):
        """every rank's list of floats -> [world][len] on every rank"""
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.dev)
        if not self.distributed:
            return [t.tolist()]
        out = self.torch.zeros(self.world * len(values), dtype=self.torch.float64, device=self.dev)
        self.dist.all_gather_into_tensor(out, t)
        return out.reshape(self.world, len(values)).tolist()


def leg_c3_scaled(env, work, steps, warmup):
    """configs[2] on every rank: `C3_FRAMES_PER_STEP` dual 1280x720 / 2000-feature frames per step per rank (extract + 3 matches per dual
    frame) + the per-step all-gather of the newest dual frame's features + the cross-rank relocalisation match. Weak scaling: the node's
    rate = features of all ranks / the slowest rank's time."""
    dt = work.run(steps, warmup, env.barrier)
    rows = env.gather([dt, work.features_timed(), work.allgather_us(), work.features_per_step()])
    ag_bytes, x_name = int(work.allgather_bytes()), work.exchange_name()
    work.close()
    dt_max = max(r[0] for r in rows)
    per_rank = [round(r[1] / r[0] / 1e3, 1) for r in rows]
    return {"workload": "configs[2] on every rank: dual 1280x720 stream, 2000 feat/cam, %d dual frames per step per rank (extract + 3 BF matches each) "
                        "+ per-step all-gather of the newest dual frame's features + cross-rank match" % work.frames_per_step,
            "n_gpus": env.world, "steps": steps, "warmup": warmup,
            "kfeatures_s": round(sum(r[1] for r in rows) / dt_max / 1e3, 1),
            "dual_frames_s": round(env.world * work.frames_per_step * steps / dt_max, 1),
            "ms_per_step": round(dt_max / steps * 1e3, 3),
            "per_rank_kfeatures_s": per_rank, "per_rank_features_per_step": [int(r[3]) for r in rows],
            "allgather_us": round(max(r[2] for r in rows), 2), "allgather_bytes_per_rank": ag_bytes,
            "exchange": x_name}


def leg_c5_node(env, front_step, front_sync, front_features, ba_solve, ba_release, seconds):
    """configs[4] as specified: per rank ONE dual-camera sequence -- the main thread extracts + matches one new dual 1280x720 frame per
    step, the per-frame pattern of the Tracking thread (src/Tracking.cc:236-269) -- next to ONE local BA loop on a second host thread, the
    LocalMapping thread's Optimizer::LocalBundleAdjustment (src/LocalMapping.cc:97-104). Two windows of `seconds`: the front end alone,
    then both; the solver alone is timed over a few solves in between."""
    def front_window(stop_after):
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < stop_after:
            for _ in range(8):
                front_step()
            n += 8
        front_sync()
        return n, time.perf_counter() - t0
    for _ in range(8):
        front_step()
    front_sync()
    env.barrier()
    n_alone, t_alone = front_window(seconds)
    ba_solve()
    it_alone, tb0 = 0, time.perf_counter()
    while time.perf_counter() - tb0 < min(seconds, 0.5) or it_alone == 0:
        it_alone += ba_solve()
    tb_alone = time.perf_counter() - tb0
    stop, ready = threading.Event(), threading.Event()
    stat = {"its": 0, "solves": 0, "t": 0.0, "err": None}

    def ba_worker():                                  # the LocalMapping thread: its own solver context (arena, streams) is built by its first solve
        try:
            ba_solve()
            ready.set()
            t0 = time.perf_counter()
            while not stop.is_set():
                stat["its"] += ba_solve()
                stat["solves"] += 1
            stat["t"] = time.perf_counter() - t0
            ba_release()
        except Exception as e:                        # noqa: BLE001
            stat["err"] = str(e)
            ready.set()
    th = threading.Thread(target=ba_worker)
    th.start()
    ready.wait()
    env.barrier()
    n_both, t_both = front_window(seconds)
    stop.set()
    th.join()
    if stat["err"]:
        raise RuntimeError("c5_node solver thread: %s" % stat["err"])
    f = front_features()
    rows = env.gather([n_both / t_both, f * n_both / t_both / 1e3, stat["its"] / max(stat["t"], 1e-9), n_alone / t_alone, it_alone / tb_alone, stat["solves"]])
    return {"workload": "configs[4]: per rank ONE dual 1280x720 sequence (2000 feat/cam, one new dual frame per step: extract + 3 matches, main thread, at most two frames in flight) "
                        "next to ONE local BA loop (50 KF / 2000 MP / 20k edges, dcs_ba_local on a second host thread)",
            "n_gpus": env.world, "window_s": seconds,
            "concurrent": {"dual_frames_s": round(sum(r[0] for r in rows), 1), "kfeatures_s": round(sum(r[1] for r in rows), 1), "ba_iters_s": round(sum(r[2] for r in rows), 1),
                           "per_rank_dual_frames_s": [round(r[0], 1) for r in rows], "per_rank_ba_iters_s": [round(r[2], 1) for r in rows],
                           "ba_solves_per_rank": [int(r[5]) for r in rows]},
            "alone": {"dual_frames_s": round(sum(r[3] for r in rows), 1), "ba_iters_s": round(sum(r[4] for r in rows), 1),
                      "per_rank_dual_frames_s": [round(r[3], 1) for r in rows], "per_rank_ba_iters_s": [round(r[4], 1) for r in rows]},
            "concurrent_vs_alone": {"front_end": round(sum(r[0] for r in rows) / max(sum(r[3] for r in rows), 1e-9), 3),
                                    "ba": round(sum(r[2] for r in rows) / max(sum(r[4] for r in rows), 1e-9), 3)}}


class StandInC3:
    """CPU stand-in of the c3_scaled work for the gloo self-test: slots of the C3 shape, the all-gather of sharding.py, no extraction."""
    frames_per_step = C3_FRAMES_PER_STEP

    def __init__(self, torch, sharding, rank, world):
        self.torch, self.sharding, self.rank, self.world, self.cap = torch, sharding, rank, world, C3_CAP_STAND_IN
        self.kp = torch.full((2, self.cap, 7), float(rank), dtype=torch.float32)
        self.desc = torch.full((2, self.cap, 32), rank + 1, dtype=torch.uint8)
        self.n = torch.tensor([1900 + rank, 1950 + rank], dtype=torch.int32)
        self.ag, self.timed = [], 0

    def run(self, steps, warmup, barrier):
        for i in range(warmup + steps):
            if i == warmup:
                barrier()
                t0 = time.perf_counter()
            ta = time.perf_counter()
            g_kp, g_desc, g_n = self.sharding.allgather_features(self.kp, self.desc, self.n, self.cap)
            assert g_n.tolist() == [v for r in range(self.world) for v in (1900 + r, 1950 + r)] and tuple(g_desc.shape) == (2 * self.world, self.cap, 32)
            if i >= warmup:
                self.ag.append((time.perf_counter() - ta) * 1e6)
        barrier()
        self.timed = steps
        return time.perf_counter() - t0

    def features_per_step(self):
        return 2 * self.frames_per_step * 1950

    def features_timed(self):
        return self.features_per_step() * self.timed

    def allgather_us(self):
        return sum(self.ag) / max(len(self.ag), 1)

    def allgather_bytes(self):
        return 2 * self.sharding.record_bytes(self.cap)

    def exchange_name(self):
        return "stand-in: torch.distributed all_gather_into_tensor on gloo"

    def close(self):
        pass


def stand_in_c5(rank):
    def front_step():
        time.sleep(0.0005)

    def ba_solve():
        time.sleep(0.002)
        return 15
    return front_step, (lambda: None), (lambda: 3900 + rank), ba_solve, (lambda: None)


class Pipeline:
    """extraction + matching of `streams` x `frames` dual frames per step, HBM-resident, double-buffered feature slots:
    matching of step i (main stream) overlaps the extraction of step i + 1 (extractor streams)."""

    def __init__(self, pkg, torch, dev, local_rank, W, H, NF, streams, frames, n_lanes, stream_base, n_events, n_sets=1, stream_factory=None):
        synth = pkg.synth
        self.pkg, self.torch, self.dev = pkg, torch, dev
        self.W, self.H, self.NF, self.streams, self.frames = W, H, NF, streams, frames
        P = self.P = streams * frames
        n_unique = min(frames, 8) if streams == 1 else min(frames, 2)
        self.n_unique = n_unique
        # `n_sets` distinct input batches, rotated step by step: set r holds the synthetic frames r * n_unique .. (r + 1) * n_unique - 1 of
        # every stream, so no step re-reads the images of the step before it out of the Infinity Cache
        self.n_sets = n_sets = max(1, n_sets)
        self.host_frames = {}
        self.d_imgs = []
        for r in range(n_sets):
            imgs = []
            for s in range(streams):
                for f in range(n_unique):
                    self.host_frames[(s, r * n_unique + f)] = synth.frame_pair(W, H, stream_base + s, r * n_unique + f)
            for s in range(streams):
                for f in range(frames):
                    imgs.extend(self.host_frames[(s, r * n_unique + f % n_unique)])
            host = np.stack(imgs)
            if r == 0:
                self.host_imgs = host
            self.d_imgs.append(torch.from_numpy(host).to(dev))
        self.d_img = self.d_imgs[0]
        self.set_features = [None] * n_sets           # keypoints one step returns for input set r (filled by run())
        n_lanes = self.n_lanes = max(1, min(n_lanes, P))
        self.lane_pairs = [P // n_lanes + (1 if i < P % n_lanes else 0) for i in range(n_lanes)]
        self.exts = [pkg.ORBextractor(NF, 1.2, 8, 20, 7, device=local_rank, max_images=2 * lp) for lp in self.lane_pairs]
        self.ext = self.exts[0]
        # the extraction lanes run next to the matcher (the caller's current stream): every lane stream is created on a hardware queue of
        # its own (dcs_stream_create_apart; torch.cuda.Stream() hands out pooled streams whose queue is an accident of the pool's state)
        self.own_streams = []
        self.main_raw = torch.cuda.current_stream().cuda_stream

        def apart_stream():
            raw, _ = pkg.abi.stream_apart([self.main_raw] + self.own_streams)
            self.own_streams.append(raw)
            return torch.cuda.ExternalStream(raw, device=dev)
        if stream_factory is None and not os.environ.get("DCS_BENCH_POOLED_STREAMS"):
            stream_factory = apart_stream
        self.lane_streams = [(stream_factory() if stream_factory else torch.cuda.Stream(device=dev)) for _ in range(n_lanes)]
        self.lane_done = [torch.cuda.Event() for _ in range(n_lanes)]
        self.matcher = pkg.ORBmatcher(0.75, True)
        cap = self.cap = self.ext.default_cap()
        NP = self.NP = 2 * streams                     # "t-1" slots: the newest dual frame of every stream from the previous step
        S = self.S = NP + 2 * P
        self.NB = 2
        self.d_kp_b = [torch.zeros((S, cap, 7), dtype=torch.float32, device=dev) for _ in range(self.NB)]
        self.d_desc_b = [torch.zeros((S, cap, 32), dtype=torch.uint8, device=dev) for _ in range(self.NB)]
        self.d_n_b = [torch.zeros(S, dtype=torch.int32, device=dev) for _ in range(self.NB)]
        self.match_done = [torch.cuda.Event() for _ in range(self.NB)]
        pairs = []
        for s in range(streams):
            for f in range(frames):
                c0 = NP + 2 * (s * frames + f)
                c1 = c0 + 1
                p0 = (c0 - 2) if f > 0 else 2 * s
                pairs += [(c0, c1), (c0, p0), (c1, p0 + 1)]
        self.n_pairs = len(pairs)
        self.pairs = pairs
        self.d_pairs = torch.tensor(pairs, dtype=torch.int32, device=dev)
        self.d_match = torch.zeros((self.n_pairs, cap), dtype=torch.int32, device=dev)
        self.d_nm = torch.zeros(self.n_pairs, dtype=torch.int32, device=dev)
        self.d_b = torch.zeros((self.n_pairs, cap), dtype=torch.int32, device=dev)
        self.d_s = torch.zeros((self.n_pairs, cap), dtype=torch.int32, device=dev)
        # newest dual frame of every stream inside the batch (source of the next step's "t-1" slots)
        self.newest = torch.tensor([NP + 2 * (s * frames + frames - 1) + c for s in range(streams) for c in (0, 1)], dtype=torch.int64, device=dev)
        self.ev_all = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_events)]
        self.step_no = 0
        self.serial = False
        self.post_match = None                          # hook: called inside step() after the matcher launch (all-gather leg)

    def step(self):
        torch = self.torch
        it = self.step_no
        ev = self.ev_all[it % len(self.ev_all)]
        self.step_no += 1
        cur, prv = it % self.NB, (it - 1) % self.NB
        d_kp, d_desc, d_n = self.d_kp_b[cur], self.d_desc_b[cur], self.d_n_b[cur]
        NP = self.NP
        first = 0
        for li in range(self.n_lanes):
            a, b = 2 * first, 2 * (first + self.lane_pairs[li])
            if it >= self.NB:
                self.lane_streams[li].wait_event(self.match_done[cur])
            self.exts[li].extract_batch_device(self.d_imgs[it % self.n_sets][a:b], d_kp[NP + a:NP + b], d_desc[NP + a:NP + b], d_n[NP + a:NP + b], self.cap,
                                               stream=self.lane_streams[li].cuda_stream)
            self.lane_done[li].record(self.lane_streams[li])
            first += self.lane_pairs[li]
        for li in range(self.n_lanes):
            torch.cuda.current_stream().wait_event(self.lane_done[li])
        if self.serial:
            torch.cuda.synchronize()
        if self.streams == 1:
            d_kp[0:2].copy_(self.d_kp_b[prv][self.S - 2:]); d_desc[0:2].copy_(self.d_desc_b[prv][self.S - 2:]); d_n[0:2].copy_(self.d_n_b[prv][self.S - 2:])
        else:
            d_kp[0:NP].copy_(self.d_kp_b[prv][self.newest]); d_desc[0:NP].copy_(self.d_desc_b[prv][self.newest]); d_n[0:NP].copy_(self.d_n_b[prv][self.newest])
        ev[0].record()
        self.matcher.match_bf_batch_device(d_desc, d_kp, d_n, self.cap, self.d_pairs, self.n_pairs, self.d_match, self.d_nm, self.d_b, self.d_s, 50,
                                           stream=torch.cuda.current_stream().cuda_stream)
        ev[1].record()
        if self.post_match is not None:
            self.post_match(d_kp, d_desc, d_n, ev)
        self.match_done[cur].record()
        if self.serial:
            torch.cuda.synchronize()

    def last_slots(self):
        c = (self.step_no - 1) % self.NB
        return self.d_kp_b[c], self.d_desc_b[c], self.d_n_b[c]

    def features_per_step(self):
        return int(self.last_slots()[2][self.NP:].sum().item())

    def features_in_steps(self, first_step, n_steps):
        """keypoints returned by steps [first_step, first_step + n_steps) (input set = step number mod n_sets)"""
        return sum(self.set_features[(first_step + i) % self.n_sets] for i in range(n_steps))

    def run(self, steps, warmup, barrier=None):
        torch = self.torch
        # set-up, outside warm-up and timing: one step per input set to learn how many keypoints each set yields (the step counter is
        # realigned to a multiple of n_sets afterwards, so "set = step mod n_sets" holds for every later step)
        if self.set_features[0] is None:
            while self.step_no % self.n_sets:
                self.step()
            for r in range(self.n_sets):
                self.step()
                torch.cuda.synchronize()
                self.set_features[r] = self.features_per_step()
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        for e_ in self.exts:
            e_.timing_totals(reset=True)
        self.step_no_timed0 = self.step_no
        if barrier:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        if barrier:
            barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def close(self):
        for e_ in self.exts:
            e_.close()
        self.torch.cuda.synchronize()
        self.lane_streams = []
        for raw in self.own_streams:
            self.pkg.abi.lib().dcs_stream_destroy(raw)
        self.own_streams = []


class FeatureExchange:
    """The per-step exchange of the north star, bound to one Pipeline: all-gather of the newest dual frame's features (2 camera slots per
    rank) through the library's own RCCL communicator (dcs_features_allgather) or through torch.distributed + pack / unpack, then the
    cross-GPU relocalisation match: this rank's cam0 against every other rank's cam1. Installed as Pipeline.post_match."""

    def __init__(self, pkg, torch, dist, sharding, dev, rank, world, comm, pipe, stream):
        self.torch, self.dist, self.sharding, self.comm, self.pipe, self.stream, self.world = torch, dist, sharding, comm, pipe, stream, world
        cap = self.cap = pipe.cap
        self.rec = sharding.record_bytes(cap)        # per camera: kp 28 B + desc 32 B per slot + count
        self.g_kp = torch.zeros((2 * world, cap, 7), dtype=torch.float32, device=dev)
        self.g_desc = torch.zeros((2 * world, cap, 32), dtype=torch.uint8, device=dev)
        self.g_n = torch.zeros(2 * world, dtype=torch.int32, device=dev)
        nx = max(world - 1, 1)
        self.x_pairs = torch.tensor(sharding.reloc_pairs(rank, world) or [(0, 1)], dtype=torch.int32, device=dev)
        self.x_match = torch.zeros((nx, cap), dtype=torch.int32, device=dev)
        self.x_nm = torch.zeros(nx, dtype=torch.int32, device=dev)
        self.x_b = torch.zeros((nx, cap), dtype=torch.int32, device=dev)
        self.x_s = torch.zeros((nx, cap), dtype=torch.int32, device=dev)
        self.g_send = torch.zeros((2, self.rec), dtype=torch.uint8, device=dev)
        self.g_recv = torch.zeros((2 * world, self.rec), dtype=torch.uint8, device=dev)

    def __call__(self, d_kp, d_desc, d_n, ev):
        S, cap = self.pipe.S, self.cap
        if self.comm is not None:
            ev[2].record()
            self.comm.allgather_features(d_kp[S - 2:], d_desc[S - 2:], d_n[S - 2:], cap, self.g_kp, self.g_desc, self.g_n, self.stream)
            ev[3].record()
        else:
            self.sharding.pack_features(d_kp[S - 2:], d_desc[S - 2:], d_n[S - 2:], cap, self.g_send)
            ev[2].record()
            self.dist.all_gather_into_tensor(self.g_recv, self.g_send)
            ev[3].record()
            self.sharding.unpack_features(self.g_recv, cap, self.g_kp, self.g_desc, self.g_n)
        if self.world > 1:
            self.pipe.matcher.match_bf_batch_device(self.g_desc, self.g_kp, self.g_n, cap, self.x_pairs, self.world - 1, self.x_match, self.x_nm, self.x_b, self.x_s, 50,
                                                    stream=self.stream)


class GpuC3:
    """the c3_scaled work of one rank: a Pipeline of 64 dual 1280x720 / 2000-feature frames per step + the FeatureExchange"""
    frames_per_step = C3_FRAMES_PER_STEP

    def __init__(self, pkg, torch, dist, sharding, dev, local_rank, rank, world, comm, stream, exchange_name, n_events):
        self.torch = torch
        self.pipe = Pipeline(pkg, torch, dev, local_rank, 1280, 720, 2000, 1, self.frames_per_step, 1, rank, n_events, n_sets=2)
        for e_ in self.pipe.exts:
            e_.set_timing(0)                          # no stage figures on this leg: no markers on its stream
        self.x = FeatureExchange(pkg, torch, dist, sharding, dev, rank, world, comm, self.pipe, stream)
        self.pipe.post_match = self.x
        self._name, self.steps = exchange_name, 0

    def run(self, steps, warmup, barrier):
        self.steps = steps
        return self.pipe.run(steps, warmup, barrier)

    def features_per_step(self):
        return self.pipe.features_per_step()

    def features_timed(self):
        return self.pipe.features_in_steps(self.pipe.step_no_timed0, self.steps)

    def allgather_us(self):
        p = self.pipe
        evs = [p.ev_all[(p.step_no_timed0 + i) % len(p.ev_all)] for i in range(self.steps)]
        return sum(e[2].elapsed_time(e[3]) for e in evs) * 1e3 / max(self.steps, 1)

    def allgather_bytes(self):
        return 2 * self.x.rec

    def exchange_name(self):
        return self._name

    def close(self):
        self.pipe.close()
        del self.x, self.pipe
        self.torch.cuda.empty_cache()


def gpu_c5(pkg, torch, dev, local_rank, rank, n_events=64):
    """the leaves of leg_c5_node on one GPU: one dual 1280x720 / 2000-feature frame per step (extract + 3 matches) and one C4 local BA"""
    pipe = Pipeline(pkg, torch, dev, local_rank, 1280, 720, 2000, 1, 1, 1, rank, n_events, n_sets=1)
    for e_ in pipe.exts:
        e_.set_timing(0)
    pkg.abi.ba_avoid_streams([pipe.main_raw] + [s_.cuda_stream for s_ in pipe.lane_streams])      # the solver's streams: other hardware queues
    prep = pkg.Optimizer.prepare(pkg.synth.ba_problem(seed=42 + rank))
    pipe.step()
    torch.cuda.synchronize()

    def front_step():                                 # at most two frames in flight (the two feature slot sets): frame i is enqueued once frame i - 2 is done
        it = pipe.step_no
        if it >= pipe.NB:
            pipe.match_done[it % pipe.NB].synchronize()
        pipe.step()

    def ba_solve():
        return int(sum(prep.solve()["n_iters"]))

    def release():
        pkg.abi.ba_release_thread()

    def close():
        pipe.close()
        pkg.abi.ba_avoid_streams([])
    return front_step, torch.cuda.synchronize, pipe.features_per_step, ba_solve, release, close


def run_c5(pkg, torch, dev, local_rank, args, n_ba=8):
    """BASELINE config C5 on ONE GPU: 8 dual-camera streams (1280x720, 2000 features / camera), one new dual frame per stream per
    step, next to one local BA per stream (dcs_ba_local_batch from a second host thread, its own HIP streams). The reference's
    pattern: Tracking thread extracts / matches while the LocalMapping thread runs LocalBundleAdjustment (src/LocalMapping.cc:97-104).
    Measured time-sliced (both sides on every CU) and PARTITIONED: the solver's streams on `ba_cus` compute units, the front end's on
    the others (CU masks; sweep over the split)."""
    synth = pkg.synth
    preps = [pkg.Optimizer.prepare(synth.ba_problem(seed=42 + s)) for s in range(n_ba)]
    steps = 150                                   # the concurrent window is 3 x this: ~150 ms, 30+ BA rounds

    def one_config(ba_cus, mask_front=True):
        ext_streams = []

        def masked_stream():
            ext_streams.append(pkg.abi.cu_range_stream(ba_cus, N_CU - ba_cus))
            return torch.cuda.ExternalStream(ext_streams[-1], device=dev)
        pkg.abi.ba_release_thread()
        pkg.abi.ba_set_cu_range(0, ba_cus)          # 0 CUs = no restriction
        front_masked = bool(ba_cus) and mask_front
        main = masked_stream() if front_masked else torch.cuda.current_stream()
        with torch.cuda.stream(main):
            pipe = Pipeline(pkg, torch, dev, local_rank, 1280, 720, 2000, 8, 1, 1, 0, 64, stream_factory=masked_stream if front_masked else None)
            for e_ in pipe.exts:
                e_.set_timing(0)
            if not ba_cus:
                pkg.abi.ba_avoid_streams([pipe.main_raw] + [s_.cuda_stream for s_ in pipe.lane_streams])     # the solver's streams: other hardware queues
            for _ in range(2):
                pkg.Optimizer.LocalBundleAdjustmentBatch(preps)

            def ba_calls(n):
                its, t0 = 0, time.perf_counter()
                for _ in range(n):
                    _check_batch(pkg, preps)
                    its += sum(sum(p.res.n_iters) for p in preps)
                return its, time.perf_counter() - t0
            dt_alone = pipe.run(steps, 3)
            feats = pipe.features_per_step()
            its_alone, t_ba_alone = ba_calls(10)
            stop = threading.Event()
            ba_stat = {"its": 0, "calls": 0, "t": 0.0}

            ready = threading.Event()

            def ba_worker():
                _check_batch(pkg, preps)          # this thread's solver context (arena, streams on hardware queues of their own) is built here, once
                ready.set()
                t0 = time.perf_counter()
                while not stop.is_set():
                    _check_batch(pkg, preps)
                    ba_stat["its"] += sum(sum(p.res.n_iters) for p in preps)
                    ba_stat["calls"] += 1
                ba_stat["t"] = time.perf_counter() - t0
                pkg.abi.ba_release_thread()

            th = threading.Thread(target=ba_worker)
            th.start()
            ready.wait()
            time.sleep(0.02)
            dt_both = pipe.run(3 * steps, 2)
            stop.set()
            th.join()
            pipe.close()
        torch.cuda.synchronize()
        for s_ in ext_streams:
            pkg.abi.lib().dcs_stream_destroy(s_)
        return {"ba_cus": ba_cus, "front_end_cus": N_CU - ba_cus if front_masked else N_CU, "features_per_step": feats,
                "concurrent": {"kfeatures_s": round(feats * 3 * steps / dt_both / 1e3, 1), "dual_frames_s": round(8 * 3 * steps / dt_both, 1),
                               "ba_iters_s": round(ba_stat["its"] / max(ba_stat["t"], 1e-9), 1), "ba_rounds": ba_stat["calls"]},
                "alone": {"kfeatures_s": round(feats * steps / dt_alone / 1e3, 1), "dual_frames_s": round(8 * steps / dt_alone, 1),
                          "ba_iters_s": round(its_alone / t_ba_alone, 1)}}

    sliced = one_config(0)
    sweep = [one_config(c) for c in ((16, 32, 48, 64) if not os.environ.get("DCS_BENCH_C5_NO_SWEEP") else (32,))]
    # only the solver restricted, the front end on every CU (its streams unmasked)
    sweep += [one_config(c, mask_front=False) for c in ((64, 128) if not os.environ.get("DCS_BENCH_C5_NO_SWEEP") else ())]
    again = [one_config(0) for _ in range(int(os.environ.get("DCS_BENCH_C5_REPEAT", "1")))]       # the first configuration once more, now warm
    pkg.abi.ba_avoid_streams([])
    pkg.abi.ba_release_thread()
    pkg.abi.ba_set_cu_range(0, 0)
    # the split that keeps the most of BOTH sides relative to the unrestricted stand-alone rates
    ref_f, ref_b = sliced["alone"]["kfeatures_s"], sliced["alone"]["ba_iters_s"]
    for r in sweep:
        r["vs_unrestricted_alone"] = {"front_end": round(r["concurrent"]["kfeatures_s"] / max(ref_f, 1e-9), 3), "ba": round(r["concurrent"]["ba_iters_s"] / max(ref_b, 1e-9), 3)}
    best = max(sweep, key=lambda r: min(r["vs_unrestricted_alone"]["front_end"] / 0.9, r["vs_unrestricted_alone"]["ba"] / 0.7))
    out = {"workload": "configs[4] on ONE GPU (C5 proper is one stream per GPU): 8 dual 1280x720 streams, 2000 feat/cam, 1 new dual frame per stream per step "
                       "(extract + 3 matches) next to 8 concurrent local BAs (50 KF / 2000 MP each, one dcs_ba_local_batch per round)",
           "features_per_step": sliced["features_per_step"],
           "concurrent": sliced["concurrent"], "alone": sliced["alone"],
           "time_sliced_vs_alone": {"front_end": round(sliced["concurrent"]["kfeatures_s"] / max(ref_f, 1e-9), 3), "ba": round(sliced["concurrent"]["ba_iters_s"] / max(ref_b, 1e-9), 3)},
           "partitioned": best, "partition_sweep": sweep, "time_sliced_repeat": again}
    return out


def _check_batch(pkg, preps):
    import ctypes as C
    n = len(preps)
    pbs = (C.c_void_p * n)(*[C.addressof(p.pb) for p in preps])
    ress = (C.c_void_p * n)(*[C.addressof(p.res) for p in preps])
    rc = pkg.abi.lib().dcs_ba_local_batch(n, C.cast(pbs, C.c_void_p), None, C.cast(ress, C.c_void_p))
    if rc:
        raise RuntimeError("dcs_ba_local_batch -> %d: %s" % (rc, pkg.abi.lib().dcs_last_error().decode()))


def cpu_frame_pair(O, o, a, b, prev):
    """what one step does for one dual frame, on the oracle: 2 extractions + 3 knn2 / ratio / rot-hist matches"""
    ka, da = o.extract(a)
    kb, db = o.extract(b)
    jobs = [(da, ka, db, kb)]
    if prev is not None:
        jobs += [(da, ka, prev[0], prev[1]), (db, kb, prev[2], prev[3])]
    else:
        jobs += [(da, ka, da, ka), (db, kb, db, kb)]
    for (q, kq, t, kt) in jobs:
        bi, bd, sd = O.knn2(q, t)
        O.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, kq["angle"], kt["angle"])
    return (da, ka, db, kb), len(ka) + len(kb)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    in_rank_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_rank_env:
        sys.exit(launch_ranks(args, argv))
    if args.selftest_launch:
        if not in_rank_env:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        sys.exit(selftest_rank(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # Under a rank environment (the driver's torchrun, bench.py's own launcher, or the GPU test that sets RANK / WORLD_SIZE for one rank)
    # the run is "distributed" even at world size 1: RCCL is initialised and the per-step feature exchange runs inside the timed region.
    distributed = in_rank_env
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    pkg = entry.load_package()
    synth = pkg.synth
    dev = torch.device("cuda", local_rank)

    P, W, H, NF = args.pairs, args.width, args.height, args.nfeatures
    n_ev = args.steps + args.warmup + 2 * args.input_sets
    pipe = Pipeline(pkg, torch, dev, local_rank, W, H, NF, 1, P, args.lanes, rank, n_ev, n_sets=args.input_sets)
    pipe.serial = args.serial
    ext, cap, S, n_pairs, n_lanes = pipe.ext, pipe.cap, pipe.S, pipe.n_pairs, pipe.n_lanes
    matcher = pipe.matcher
    stream = torch.cuda.current_stream().cuda_stream
    exchange_impl = None
    if distributed:
        from orb_slam2_dualcam_amd import sharding
        # the exchange of the north star through the library's own communicator (dcs_comm_*, csrc/comm.cpp): brought up once, here,
        # with the unique id broadcast over the process group; every rank must agree on the path, so a failure anywhere (or a bring-up
        # that does not return within 60 s) sends ALL ranks to the torch.distributed path and the line says so
        comm, comm_err = None, None
        if args.exchange == "cabi":
            box = {}

            def bring_up():
                try:
                    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
                    if rank == 0:
                        uid.copy_(torch.from_numpy(pkg.abi.FeatureComm.unique_id()).to(dev))
                    dist.broadcast(uid, 0)
                    box["comm"] = pkg.abi.FeatureComm(uid.cpu().numpy(), rank, world)
                except Exception as e:               # noqa: BLE001
                    box["err"] = str(e)
            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(60.0)
            comm, comm_err = box.get("comm"), box.get("err", "bring-up timed out" if th.is_alive() else None)
            ok = torch.tensor([1 if comm is not None else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not int(ok.item()):
                comm = None
        exchange_impl = "dcs_features_allgather (library-owned RCCL communicator, slot arrays in place)" if comm is not None else \
            "torch.distributed all_gather_into_tensor + pack / unpack" + (" (C-ABI communicator unavailable: %s)" % comm_err if args.exchange == "cabi" else "")
        xchg = FeatureExchange(pkg, torch, dist, sharding, dev, rank, world, comm, pipe, stream)
        rec = xchg.rec
        pipe.post_match = xchg
    stage_keys = ("pyramid_us", "fast_us", "compact_us", "blur_us", "quadtree_us", "describe_us", "total_us")
    acc = {k: 0.0 for k in stage_keys}
    acc["match_us"] = 0.0
    acc["allgather_us"] = 0.0

    def barrier():
        if distributed:
            dist.barrier()

    # Stage markers: every hipEventRecord is a packet between two kernels on the extraction stream; with all seven stages bracketed they cost
    # 2.7 % of a step (1.405 vs 1.443 ms). Inside the timed region only the FAST stage -- the roofline kernel -- is bracketed
    # (dcs_orb_set_timing mode 1: two markers per step); the other stages are sampled in extra steps after it with every marker on.
    all_markers = args.stage_timing == "all"
    for e_ in pipe.exts:
        e_.set_timing(2 if all_markers else 1)
    dt = pipe.run(args.steps, args.warmup, barrier)
    # What did the timed region produce? The first four images of the LAST timed step and the three matches among them are copied out here
    # (the clock has stopped) and held against the oracle in the CPU leg below: "parity_sample" on the line.
    parity_snap = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and pipe.frames >= 2 and pipe.streams == 1:
        kp_l, de_l, n_l = pipe.last_slots()
        a0 = pipe.NP
        parity_snap = dict(kp=kp_l[a0:a0 + 4].cpu().numpy(), desc=de_l[a0:a0 + 4].cpu().numpy(), n=n_l[a0:a0 + 4].cpu().numpy(),
                           match=pipe.d_match[3:6].cpu().numpy(), nm=pipe.d_nm[3:6].cpu().numpy(), set=(pipe.step_no - 1) % pipe.n_sets)
    for e_ in pipe.exts:                  # kernel times: summed over the lanes (each lane launches its own kernels)
        sums, n_timed = e_.timing_totals()
        for k in stage_keys:
            acc[k] += sums[k]
    for i in range(args.steps):
        ev = pipe.ev_all[(pipe.step_no_timed0 + i) % len(pipe.ev_all)]
        acc["match_us"] += ev[0].elapsed_time(ev[1]) * 1000.0
        if distributed:
            acc["allgather_us"] += ev[2].elapsed_time(ev[3]) * 1000.0
    n_sample = 0
    if not all_markers:                   # the other stages: extra steps with every marker on (same pipeline, all ranks; not part of `value`)
        n_sample = min(args.steps, 10)
        for e_ in pipe.exts:
            e_.set_timing(2); e_.timing_totals(reset=True)
        for _ in range(n_sample):
            pipe.step()
        torch.cuda.synchronize()
        for e_ in pipe.exts:
            sums, _n = e_.timing_totals()
            for k in stage_keys:
                if k != "fast_us":
                    acc[k] += sums[k] * (args.steps / n_sample)      # scaled to the K timed steps the report divides by
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    feats_timed = pipe.features_in_steps(pipe.step_no_timed0, args.steps)     # keypoints this rank returned inside the timed region
    n_feat_step = int(round(feats_timed / args.steps))
    n_match_step = int(pipe.d_nm.sum().item())
    tot = torch.tensor([feats_timed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(tot)
    feats_all = float(tot.item())
    value = feats_all / dt / 1000.0

    out = None
    if rank == 0:
        px = [w * h for w, h in (ext.level_dims(l) for l in range(ext.nlevels))]
        sum_px, px7, sum_17 = sum(px), px[-1], sum(px[1:])
        n_img = 2 * P
        n_avg = n_feat_step / n_img
        K = args.steps
        LN = n_lanes                                 # every extraction kernel is launched once per lane per step
        n_img_l = n_img / LN
        # Round 5: with the emitting FAST (k_fast_cells<EMIT>, the default for batches like this one) the cells of level l write level l + 1 --
        # SURVEY.md 8(d)'s pyramid row (levels read + levels written) and its FAST row (every level read) are ONE stage, bracketed by the FAST
        # markers; its algorithmic bytes are the sum of the two rows restricted to the levels that were emitted. The resize chain's entry then
        # covers the levels it still produced (none by default).
        E_lv = ext.emit_levels()
        pyr_fused = (sum(px[:E_lv]) + sum(px[1:E_lv + 1])) if E_lv else 0
        pyr_chain = (sum(px[E_lv:-1]) + sum(px[E_lv + 1:])) if E_lv < len(px) - 1 else 0
        fast_key = "k_fast_cells+pyramid" if E_lv else "k_fast_cells"
        algo = {                                     # algorithmic bytes per LAUNCH x images per launch
            "k_resize(x%d)" % (len(px) - 1 - E_lv): pyr_chain * n_img_l,
            fast_key: (sum_px + pyr_fused) * n_img_l,
            "k_blur": 2 * sum_px * n_img_l,
            "k_describe": int((749 + 512 + 60) * n_avg * n_img_l),
        }
        dur = {"k_resize(x%d)" % (len(px) - 1 - E_lv): acc["pyramid_us"] / (K * LN), fast_key: acc["fast_us"] / (K * LN),
               "k_blur": acc["blur_us"] / (K * LN), "k_describe": acc["describe_us"] / (K * LN)}
        if pyr_chain == 0:
            del algo["k_resize(x0)"], dur["k_resize(x0)"]
        kernels = {k: dict(us=round(dur[k], 2), algo_bytes=int(algo[k]),
                           gbps=round(algo[k] / max(dur[k], 1e-3) / 1e3, 2)) for k in algo}
        # roofline-style entry of the second-largest stage, so that it has a number to beat: SURVEY 8(d)'s orientation + BRIEF + output rows per key
        # point, PLUS its blur row (2 x pyramid pixels) when the fused form runs -- k_describe<true> blurs the 512 tested pixels of each patch itself and
        # never reads or writes a blurred pyramid, which is what those bytes would have been
        kd = kernels["k_describe"]
        d_bytes = algo["k_describe"] + (2 * sum_px * n_img_l if acc["blur_us"] == 0 else 0)
        kd["roofline"] = {"bound": "hbm", "achieved": round(d_bytes / max(dur["k_describe"], 1e-3) / 1e3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(d_bytes / max(dur["k_describe"], 1e-3) / 1e3 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(d_bytes),
                          "note": "vector + matrix + LDS issue bound (8 workgroups per CU), not byte bound: DESIGN.md section 4"}
        dom = max(dur, key=lambda k: dur[k])
        # Hardware counters cannot be read inside this process: the HBM traffic and the instruction counts come from the committed
        # rocprofv3 --pmc passes of this same command (profiles/, produced by tools/collect_profiles.sh) and are labelled "imported";
        # they are used only when the workload matches. A "launch" of the roofline kernel = its launches of ONE step (FAST runs one
        # launch per LDS size class), timed together by the hipEvents around the stage.
        traffic, traffic_raw, counters_src, pmc_k = None, None, None, {}
        for prof in ("r06_pmc_counters.json", "r05_pmc_counters.json", "r04_pmc_counters.json", "r03_pmc_counters.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", prof)))
                if (P, W, H, NF, LN) == tuple(pmc.get("bench_args", ())):
                    pmc_k = pmc["kernels"]
                    kk = pmc_k[dom.split("(")[0].split("+")[0]]
                    # FETCH_SIZE counts the L2's line requests at 64 B each; the lines are 128 B: a kernel that reads a known number of bytes with
                    # 4- / 8- / 12- / 16-byte loads per lane is reported at exactly half (profiles/r04_fetch_calibration.json,
                    # scratch/probe/fetch_calib.hip), so the corrected figure is 2 x FETCH_SIZE. WRITE_SIZE is taken as reported (not calibrated;
                    # a tenth of the total here). Below the algorithmic bytes = part of the pyramid (written by k_resize just before) is still
                    # in the 256 MB Infinity Cache / the L2s when it is read.
                    traffic_raw = int((kk["FETCH_SIZE_per_step"] + kk["WRITE_SIZE_per_step"]) * 1024 / LN)
                    traffic = int((2 * kk["FETCH_SIZE_per_step"] + kk["WRITE_SIZE_per_step"]) * 1024 / LN)
                    counters_src = "imported: profiles/%s (rocprofv3 --pmc passes of this command, collected in a separate run); traffic = 2 x FETCH_SIZE + WRITE_SIZE " \
                                   "(128-byte lines counted at 64 B: profiles/r04_fetch_calibration.json), traffic_raw = as reported" % prof
                    break
            except Exception:
                traffic = None
        # Issue pressure per unit while a kernel runs, each as [lo, hi] (profiles/r04_valu_rate_probe.txt, r03_lds_rate_probe.txt; 1 tick = 1 / 720 MHz):
        #   valu  SQ_INSTS_VALU x cost / (1024 SIMDs x duration); hi prices every instruction at the slow class (1.33 ticks = 4.3 shader cycles), lo
        #         at the fast / slow mix of the kernel's loop code (tools/isa_class_mix.py -> profiles/r04_isa_class_mix.json; fast = 0.8 ticks)
        #   lds   SQ_INSTS_LDS x cost / (256 CUs x duration x 2.39 GHz); lo: every instruction a plain read (2.5 CU-cycles), hi: the loop code's static mix of
        #         reads (2.5) and writes / wide reads / permutes (4.2)
        #   salu  SQ_INSTS_SALU x cost / (1024 x duration): lo prices the loop code's static mix of two-operand (2.4 ticks), one-operand (1.3) and
        #         wait / branch / compare (0.4) scalar instructions, hi every instruction at 1.3 ticks -- the scalar unit is shared by the CU's four SIMDs
        try:
            mix = json.load(open(os.path.join(ROOT, "profiles", "r05_isa_class_mix.json" if os.path.exists(os.path.join(ROOT, "profiles", "r05_isa_class_mix.json")) else "r04_isa_class_mix.json")))["kernels"]
        except Exception:
            mix = {}
        mix_key = {"k_resize": "k_resize<true>", "k_fast_cells": "k_fast_cells<48, true, true>" if E_lv else "k_fast_cells<48, true, false>", "k_describe": "k_describe<true>", "k_blur": "k_blur_fold"}
        TICK_NS = 1.0 / 0.72
        for k in kernels:
            base = k.split("(")[0].split("+")[0]
            kk = pmc_k.get(base)
            if not kk or dur[k] <= 0:
                continue
            simd_ns = 4 * N_CU * dur[k] * 1e3
            m = mix.get(mix_key.get(base, base), {})
            loops = m.get("loops") or {}
            scope = loops if loops.get("valu_fast", 0) + loops.get("valu_slow", 0) >= 60 else (m.get("all") or {})
            if "SQ_INSTS_VALU_per_step" in kk:
                nv = kk["SQ_INSTS_VALU_per_step"] / LN
                lo_t = scope.get("ticks_per_valu_instruction", 1.33)
                kernels[k]["valu_issue_frac"] = [round(nv * lo_t * TICK_NS / simd_ns, 4), round(min(1.0, nv * 1.33 * TICK_NS / simd_ns), 4)]      # (hi is a bound: capped at the time there is)
                kernels[k]["valu_fast_class_share"] = scope.get("valu_fast_share")
            if "SQ_INSTS_LDS_per_step" in kk:
                nl = kk["SQ_INSTS_LDS_per_step"] / LN
                n_r, n_w = scope.get("lds_read", 0), scope.get("lds_write", 0)
                c_mix = (2.5 * n_r + 4.2 * n_w) / (n_r + n_w) if n_r + n_w else 4.2          # static read / write mix of the kernel's loop code
                kernels[k]["lds_issue_frac"] = [round(nl * c_ / (N_CU * dur[k] * 1e-6 * 2.39e9), 4) for c_ in (2.5, c_mix)]
            if "SQ_INSTS_SALU_per_step" in kk:
                ns_ = kk["SQ_INSTS_SALU_per_step"] / LN
                n_sc = scope.get("sop2", 0) + scope.get("sop1", 0) + scope.get("s_other", 0)
                lo_c = (2.4 * scope.get("sop2", 0) + 1.3 * scope.get("sop1", 0) + 0.4 * scope.get("s_other", 0)) / n_sc if n_sc else 1.3
                kernels[k]["salu_issue_frac"] = sorted([round(ns_ * lo_c * TICK_NS / simd_ns, 4), round(ns_ * 1.3 * TICK_NS / simd_ns, 4)])
        roofline = dict(kernel=dom, bound="hbm", achieved=kernels[dom]["gbps"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(kernels[dom]["gbps"] / HBM_PEAK_GBS, 5), traffic=traffic, traffic_raw=traffic_raw, traffic_source=counters_src,
                        algorithmic_bytes_per_launch=int(algo[dom]), avg_launch_us=round(dur[dom], 2),
                        valu_issue_frac=kernels[dom].get("valu_issue_frac"), lds_issue_frac=kernels[dom].get("lds_issue_frac"),
                        salu_issue_frac=kernels[dom].get("salu_issue_frac"),
                        stage=("FAST + ComputePyramid in one stage: the cells of level l write level l + 1 from LDS (levels 1..%d); algorithmic bytes = SURVEY 8(d)'s FAST row "
                               "(%d B / image) + its pyramid row for those levels (%d B / image)" % (E_lv, sum_px, pyr_fused)) if E_lv else "FAST alone (the k_resize chain made the pyramid)",
                        frac_fast_row_only=round(sum_px * n_img_l / max(dur[dom], 1e-3) / 1e3 / HBM_PEAK_GBS, 5) if E_lv and dom == fast_key else None,
                        note="an integer stencil (round 4: 601 vector + 289 scalar + 171 LDS wave-instructions per 30-px cell; the emitting form adds ~200 vector instructions per cell "
                             "for the next level's pixels): the byte roof is not its wall, vector issue is (>= 0.92 of the SIMDs' issue time priced by instruction class), with the "
                             "LDS pipe and the CU's scalar unit beside it (DESIGN.md section 4)")
        out = {
            "metric": "dual-frame ORB extract+match kfeatures/s; local-BA iters/s (50 KF / 2k MP)",
            "value": round(value, 2), "unit": "kfeatures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: dual %dx%d stream, %d feat/cam, 8 levels, extract + BF match (3 matches/dual frame)"
                                   % ({(640, 480, 1000): "configs[1]", (1280, 720, 2000): "configs[2] shape"}.get((W, H, NF), "custom shape (not a BASELINE config)"), W, H, NF),
                       "dual_frames_per_step_per_gpu": P, "extractor_lanes": n_lanes, "features_per_step_per_gpu": n_feat_step,
                       "matches_per_step_per_gpu": n_match_step, "parallelism": "frame-pair shard x%d" % world},
            "roofline": roofline,
            "stage_us_per_step": {k: round(v / K, 2) for k, v in acc.items()},
            "stage_timing": ("every stage bracketed by hipEvents inside the timed region (--stage-timing all; the markers cost ~2.7 % of a step)" if all_markers else
                             "fast_us (the roofline kernel) and match_us: hipEvents inside the timed region, every step; the other stages: %d extra steps after it with "
                             "every stage marker on (seven markers per step cost ~2.7 %% of a step, so the timed region carries only the two around FAST)" % n_sample),
            "kernels": kernels,
        }
        out["config"]["input_sets_rotated"] = pipe.n_sets
        if distributed:
            out["allgather_us"] = round(acc["allgather_us"] / K, 2)
            out["allgather_bytes_per_rank"] = int(2 * rec)
            out["exchange"] = exchange_impl

    solo = rank == 0 and world == 1
    # ---- matcher alone (nothing else on the GPU): the i8 matrix-core Hamming kernel against its own peak
    if solo:
        d_kp, d_desc, d_n = pipe.last_slots()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            matcher.match_bf_batch_device(d_desc, d_kp, d_n, cap, pipe.d_pairs, n_pairs, pipe.d_match, pipe.d_nm, pipe.d_b, pipe.d_s, 50, stream=stream)
        torch.cuda.synchronize()
        reps = 20
        e0.record()
        for _ in range(reps):
            matcher.match_bf_batch_device(d_desc, d_kp, d_n, cap, pipe.d_pairs, n_pairs, pipe.d_match, pipe.d_nm, pipe.d_b, pipe.d_s, 50, stream=stream)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        nn = pipe.last_slots()[2].cpu().numpy().astype(np.int64)
        dist_count = int(sum(nn[a] * nn[b] for a, b in pipe.pairs))
        tops = dist_count * 512 / us / 1e6            # 256 multiply-accumulates per Hamming distance on the matrix cores
        # which form the library launches (match_kernels.hip, launch_knn2_pairs_mfma): FP4 block-scaled when the slot fits its 14-bit index field
        fp4 = cap <= 16383 and pkg.abi.get_option("DCS_KNN2_I8") == 0
        kname, peak, unit = ("k_knn2_pairs_fp4", FP4_MFMA_PEAK_TOPS, "TOPS (FP4 block-scaled MFMA 16x16x128)") if fp4 else ("k_knn2_pairs_mfma", I8_MFMA_PEAK_TOPS, "TOPS (i8 MFMA)")
        out["matcher"] = {"kernel": kname + " + k_filter_pairs", "solo_us_per_step": round(us, 2), "pairs": n_pairs,
                          "distances": dist_count, "achieved": round(tops, 1), "peak": peak, "unit": unit,
                          "frac": round(tops / peak, 4), "gdistances_s": round(dist_count / us / 1e3, 1),
                          "algo_bytes": int((32 * 2 * n_avg + 12 * n_avg) * n_pairs),
                          "note": "not bound by the matrix pipe: ranking the keys, re-expanding the train tiles and the matrix instructions each cost about a third (DESIGN.md section 4)"}
        out["kernels"][kname + "+k_filter_pairs"] = dict(us=round(us, 2), note="solo; inside a step it runs underneath the next extraction (stage_us_per_step.match_us is that window)")

    # ---- host-buffer API (what the reference's seams hand over): one dual frame per call (latency), the whole batch (throughput)
    if solo and not args.no_host_api:
        import ctypes as C
        a, b = pipe.host_frames[(0, 0)]
        ext1 = pkg.ORBextractor(NF, 1.2, 8, 20, 7, device=local_rank, max_images=2)
        m1 = pkg.ORBmatcher(0.75, True)
        # the seam as a C++ caller drives it: caller-owned output buffers allocated once, the two C-ABI calls per dual frame
        L_ = pkg.abi.lib()
        kp1 = np.zeros((2, cap), pkg.abi.KEYPOINT); de1 = np.zeros((2, cap, 32), np.uint8); n1 = np.zeros(2, np.int32)
        mt1 = np.zeros(cap, np.int32); nm1 = C.c_int()
        ptr2 = (C.c_void_p * 2)(a.ctypes.data, b.ctypes.data)
        vp_ = lambda x: x.ctypes.data_as(C.c_void_p)      # noqa: E731

        def one_frame():
            rc_ = L_.dcs_orb_extract_batch(ext1._h, C.cast(ptr2, C.c_void_p), 2, H, W, W, vp_(kp1), vp_(de1), cap, vp_(n1))
            t_mid = time.perf_counter()
            rc_ = rc_ or L_.dcs_match_bf(vp_(de1[0]), vp_(kp1[0]), int(n1[0]), vp_(de1[1]), vp_(kp1[1]), int(n1[1]), 50, C.c_float(0.75), 1, vp_(mt1), C.byref(nm1))
            if rc_:
                raise RuntimeError("latency leg rc=%d: %s" % (rc_, L_.dcs_last_error().decode()))
            return t_mid
        for _ in range(10):
            one_frame()
        reps, tls, tes = 100, [], []
        for _ in range(reps):
            t1 = time.perf_counter()
            t_mid = one_frame()
            t2 = time.perf_counter()
            tls.append(t2 - t1); tes.append(t_mid - t1)
        tl, te = sorted(tls)[reps // 2], sorted(tes)[reps // 2]
        kps, descs = ext1.extract_batch([a, b])
        out["latency"] = {"workload": "1 dual frame per call: dcs_orb_extract_batch (2 host images in, keypoints + descriptors out) + dcs_match_bf (host buffers), "
                                      "synchronous, PCIe included -- what Frame::ExtractORB issues per frame; caller-owned buffers, median of 100",
                          "ms_per_dual_frame": round(tl * 1e3, 3), "ms_extract": round(te * 1e3, 3), "ms_match": round((tl - te) * 1e3, 3),
                          "kfeatures_s": round(int(n1.sum()) / tl / 1e3, 2)}
        ext1.close()
        # per-call latency of the other host-buffer seams (each = what one ORBmatcher member call of the reference costs here)
        seam = {}
        O_ = entry.load_oracle() if args.cpu_seconds > 0 else None

        def per_call(fn, reps=60):
            for _ in range(5):
                fn()
            ts_ = []
            for _ in range(reps):
                t0_ = time.perf_counter()
                fn()
                ts_.append(time.perf_counter() - t0_)
            return round(sorted(ts_)[reps // 2] * 1e3, 4)      # median: one arena re-growth (a 5 ms hipMalloc) in 30 calls tripled the mean
        seam["dcs_match_bf_ms"] = per_call(lambda: m1.match_bf(descs[0], kps[0], descs[1], kps[1], 50))
        seam["dcs_hamming_knn2_ms"] = per_call(lambda: pkg.ORBmatcher.knn2(descs[0], descs[1]))
        fr_, q_ = synth.projection_problem(n_per_cam=1000, n_queries=800, seed=13)
        fr_["grid_off"], fr_["grid_idx"] = pkg.frame_grid(fr_["cam_off"], fr_["kp_x"], fr_["kp_y"], fr_["min_x"], fr_["min_y"], fr_["grid_w_inv"], fr_["grid_h_inv"])
        seam["dcs_search_by_projection_ms"] = per_call(lambda: m1.SearchByProjection(fr_, q_, 100, use_ratio=True, check_orientation=False))
        fvk, fvf = synth.csr_buckets(len(descs[0]), 100, seed=100), synth.csr_buckets(len(descs[1]), 100, seed=101)
        ones = np.ones(len(descs[0]), np.uint8)
        seam["dcs_search_by_bow_ms"] = per_call(lambda: m1.SearchByBoWCrossCam(descs[0], kps[0]["angle"], ones, descs[1], kps[1]["angle"], fvk, fvf))
        seam["note"] = "median of 60 calls; host buffers in, host buffers out, synchronous; ~1000 x 1000 features; ctypes call overhead included"
        out["seam_latency"] = seam
        # the C ABI as a C++ host drives it: caller-owned output buffers allocated once, one call per batch (the Python convenience
        # wrapper would add 18 MB of numpy allocation and 1024 slice copies per call -- that is not the library's time)
        import ctypes as C
        host_imgs = [np.ascontiguousarray(pipe.host_imgs[i]) for i in range(2 * P)]
        nb_ = len(host_imgs)
        kp_h = np.zeros((nb_, cap), pkg.abi.KEYPOINT); desc_h = np.zeros((nb_, cap, 32), np.uint8); n_h = np.zeros(nb_, np.int32)
        ptrs = (C.c_void_p * nb_)(*[im.ctypes.data for im in host_imgs])

        def host_call():
            rc_ = pkg.abi.lib().dcs_orb_extract_batch(ext._h, C.cast(ptrs, C.c_void_p), nb_, H, W, W, kp_h.ctypes.data_as(C.c_void_p),
                                                      desc_h.ctypes.data_as(C.c_void_p), cap, n_h.ctypes.data_as(C.c_void_p))
            if rc_:
                raise RuntimeError("dcs_orb_extract_batch rc=%d" % rc_)
        ext.set_timing(0)                 # no stage figures are reported for the host-buffer legs: no markers on their streams
        for _ in range(3):
            host_call()
        tts = []
        for _ in range(15):
            t0 = time.perf_counter()
            host_call()
            tts.append(time.perf_counter() - t0)
        tt, tt_mean = sorted(tts)[len(tts) // 2], sum(tts) / len(tts)
        nf = int(n_h.sum())
        out["with_transfers"] = {"workload": "the default batch (%d images) through dcs_orb_extract_batch: pageable host images in, keypoints + descriptors out (caller-owned buffers), "
                                             "extraction only; pipeline of image chunks (pack || DMA up || kernels || DMA down || scatter); median of 15 calls" % nb_,
                                 "ms_per_call": round(tt * 1e3, 2), "ms_per_call_mean": round(tt_mean * 1e3, 2), "kfeatures_s": round(nf / tt / 1e3, 1),
                                 "image_GBps": round(nb_ * W * H / tt / 1e9, 2)}

        # the same batch out of a page-locked frame ring (dcs_host_alloc): the DMA reads the caller's frames in place, no staging copy
        hf = pkg.abi.HostFrames(nb_, H, W)
        for i_ in range(nb_):
            hf.frames[i_][:] = host_imgs[i_]
        ptrs_l = (C.c_void_p * nb_)(*[f_.ctypes.data for f_ in hf.frames])
        kp_l = np.zeros_like(kp_h); desc_l = np.zeros_like(desc_h); n_l = np.zeros_like(n_h)

        def locked_call(n_=nb_):
            rc_ = pkg.abi.lib().dcs_orb_extract_batch(ext._h, C.cast(ptrs_l, C.c_void_p), n_, H, W, hf.stride, kp_l.ctypes.data_as(C.c_void_p),
                                                      desc_l.ctypes.data_as(C.c_void_p), cap, n_l.ctypes.data_as(C.c_void_p))
            if rc_:
                raise RuntimeError("dcs_orb_extract_batch (page-locked frames) rc=%d" % rc_)
        for _ in range(3):
            locked_call()
        same = bool(np.array_equal(n_l, n_h) and kp_l.tobytes() == kp_h.tobytes() and np.array_equal(desc_l, desc_h))
        tls = []
        for _ in range(15):
            t0 = time.perf_counter()
            locked_call()
            tls.append(time.perf_counter() - t0)
        tl_, tl_mean = sorted(tls)[len(tls) // 2], sum(tls) / len(tls)
        nf_l = int(n_l.sum())
        ext2 = pkg.ORBextractor(NF, 1.2, 8, 20, 7, device=local_rank, max_images=2)       # like the latency leg's handle: one dual frame per call

        def locked_pair():
            rc_ = pkg.abi.lib().dcs_orb_extract_batch(ext2._h, C.cast(ptrs_l, C.c_void_p), 2, H, W, hf.stride, kp_l.ctypes.data_as(C.c_void_p),
                                                      desc_l.ctypes.data_as(C.c_void_p), cap, n_l.ctypes.data_as(C.c_void_p))
            if rc_:
                raise RuntimeError("dcs_orb_extract_batch (page-locked dual frame) rc=%d" % rc_)
        t2s = []
        for it_ in range(110):
            t0 = time.perf_counter()
            locked_pair()
            if it_ >= 10:
                t2s.append(time.perf_counter() - t0)
        ext2.close()
        out["with_transfers_page_locked"] = {"workload": "the same %d images in a page-locked frame ring (dcs_host_alloc): read in place by the DMA, no staging copy on the host; median of 15 calls" % nb_,
                                             "ms_per_call": round(tl_ * 1e3, 2), "ms_per_call_mean": round(tl_mean * 1e3, 2), "kfeatures_s": round(nf_l / tl_ / 1e3, 1),
                                             "image_GBps": round(nb_ * W * H / tl_ / 1e9, 2), "same_bytes_as_pageable": same,
                                             "ms_extract_one_dual_frame": round(sorted(t2s)[50] * 1e3, 3)}
        hf.close()

    # ---- CPU baseline (rank 0, N = 1 only): the oracle ("port"), single thread, bounded sample; then every host core
    if solo and args.cpu_seconds > 0:
        O = entry.load_oracle()
        o = O.OrbOracle(NF, 1.2, 8, 20, 7)
        prev, feats, n_done = None, 0, 0
        aff0 = os.sched_getaffinity(0)
        core = sorted(aff0)[len(aff0) // 2]          # the single-thread baseline stays on ONE core (taskset -c)
        os.sched_setaffinity(0, {core})
        tc0 = time.perf_counter()
        while time.perf_counter() - tc0 < args.cpu_seconds and n_done < 400:
            a, b = pipe.host_frames[(0, n_done % pipe.n_unique)]
            prev, nf = cpu_frame_pair(O, o, a, b, prev)
            feats += nf
            n_done += 1
        tc = time.perf_counter() - tc0
        os.sched_setaffinity(0, aff0)
        out["cpu_baseline"] = {"value": round(feats / tc / 1000.0, 3), "unit": "kfeatures/s", "cores": 1, "kind": "port", "pinned_core": core,
                               "sample": "%d dual %dx%d frames (extract x2 + 3 knn2/filter each) in %.1f s, oracle -O3 single thread pinned to core %d; host has %d cores"
                                         % (n_done, W, H, tc, core, os.cpu_count())}
        out["speedup_vs_cpu_1thread"] = round(out["value"] / max(out["cpu_baseline"]["value"], 1e-9), 1)
        if parity_snap is not None:
            # images 0..3 of the last timed step = the two dual frames f = 0, 1 of its input set; matches 3..5 of the step = (cam0(1), cam1(1)),
            # (cam0(1), cam0(0)), (cam1(1), cam1(0)) -- every operand among the four images
            bad = []
            r_ = parity_snap["set"]
            imgs4 = list(pipe.host_frames[(0, r_ * pipe.n_unique)]) + list(pipe.host_frames[(0, r_ * pipe.n_unique + 1 % pipe.n_unique)])
            ok_, od_ = [], []
            for i4, im in enumerate(imgs4):
                k4, d4 = o.extract(im, cap=cap)
                ok_.append(k4); od_.append(d4)
                n4 = int(parity_snap["n"][i4])
                g_k = parity_snap["kp"][i4][:n4].copy().view(pkg.abi.KEYPOINT).reshape(-1)
                if n4 != len(k4) or g_k.tobytes() != k4.tobytes() or not np.array_equal(parity_snap["desc"][i4][:n4], d4):
                    bad.append("image %d" % i4)
            for j4, (q4, t4) in enumerate(((2, 3), (2, 0), (3, 1))):
                bi, bd, sd = O.knn2(od_[q4], od_[t4])
                m4, nm4 = O.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, ok_[q4]["angle"], ok_[t4]["angle"])
                if nm4 != int(parity_snap["nm"][j4]) or not np.array_equal(parity_snap["match"][j4][:len(m4)], m4):
                    bad.append("match %d" % j4)
            out["parity_sample"] = "ok" if not bad else "MISMATCH: " + ", ".join(bad)
            out["parity_sample_what"] = "4 images (key points + descriptors, byte for byte) and the 3 matches among them of the last timed step vs the oracle, after the clock stopped"
        # all host cores: one stream of dual frames per worker PROCESS (tools/cpu_workers.py; a separate process tree, no GPU runtime in it)
        n_proc = max(1, os.cpu_count() or 1)
        try:
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_workers.py"), "--width", str(W), "--height", str(H), "--nfeatures", str(NF),
                                 "--seconds", str(min(args.cpu_seconds, 6.0)), "--procs", str(n_proc)], capture_output=True, text=True, timeout=120)
            res = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
            out["cpu_all_cores"] = {"value": round(res["kfeatures_s"], 2), "unit": "kfeatures/s", "cores": res["procs"], "kind": "port",
                                    "sample": "%d processes x one dual-frame stream each, %.1f s wall incl. start-up (oracle -O3, one process per host core)"
                                              % (res["procs"], res["seconds"])}
            out["speedup_vs_cpu_all_cores"] = round(out["value"] / max(out["cpu_all_cores"]["value"], 1e-9), 2)
        except Exception as e:                       # noqa: BLE001
            out["cpu_all_cores"] = "unavailable: %s" % e

    # ---- the headline workload with the batch split over TWO extraction lanes (two handles, two streams on hardware queues of their own,
    # the same 256 dual frames per step): kernels of one lane fill the ramps, tails and latency-bound stretches of the other's. Reported
    # beside `value`, which stays on one lane so that a launch of the roofline kernel has the chip to itself.
    if solo and not args.no_two_lanes and not args.no_c3 and args.lanes == 1:
        p2 = Pipeline(pkg, torch, dev, local_rank, W, H, NF, 1, P, 2, rank, 8, n_sets=args.input_sets)
        for e_ in p2.exts:
            e_.set_timing(0)
        dt2 = p2.run(args.steps, args.warmup)
        f2 = p2.features_in_steps(p2.step_no_timed0, args.steps)
        out["two_lanes"] = {"workload": "the headline batch split over 2 extraction lanes (NOTES.md section 7 item 11)", "kfeatures_s": round(f2 / dt2 / 1e3, 2),
                            "ms_per_step": round(dt2 / args.steps * 1e3, 3), "vs_one_lane": round(f2 / dt2 / 1e3 / max(out["value"], 1e-9), 4)}
        p2.close()
        del p2
        torch.cuda.empty_cache()

    # ---- C3 leg: dual 1280x720, 2000 features / camera (BASELINE configs[2]) on this GPU
    if solo and not args.no_c3 and (W, H, NF) != (1280, 720, 2000):
        p3 = Pipeline(pkg, torch, dev, local_rank, 1280, 720, 2000, 1, 64, 1, 0, 16)
        for e_ in p3.exts:
            e_.set_timing(0)                                  # no stage figures are reported for this leg: no markers on its stream
        dt3 = p3.run(8, 2)
        f3 = p3.features_per_step()
        out["c3"] = {"workload": "configs[2] at 1 GPU: dual 1280x720 stream, 2000 feat/cam, extract + BF match, 64 dual frames per step",
                     "kfeatures_s": round(f3 * 8 / dt3 / 1e3, 1), "ms_per_step": round(dt3 / 8 * 1e3, 3), "features_per_step": f3,
                     "dual_frames_s": round(64 * 8 / dt3, 1)}
        p3.close()

    # ---- local BA leg (C4: 50 KF / 2000 MP / 20k dual-camera edges), rank 0, N = 1
    if solo and not args.no_ba and hasattr(pkg.abi.lib(), "dcs_ba_local"):
        pb = synth.ba_problem()
        prep = pkg.Optimizer.prepare(pb)                             # flat problem marshalled once, as a C++ caller holds it
        prep.solve()                                                 # warm-up (allocations, code objects)
        reps, iters, tb0 = 30, 0, time.perf_counter()
        gpu_ms = 0.0
        for _ in range(reps):
            r = prep.solve()
            iters += sum(r["n_iters"])
            gpu_ms += r["gpu_ms"]
        tb = time.perf_counter() - tb0
        n_free = int((np.asarray(pb["pose_fixed"]) == 0).sum())
        E, L = len(pb["obs"]), len(pb["points"])
        ba = {"metric": "local-BA iters/s (50 KF / 2k MP / %d edges)" % E, "value": round(iters / tb, 1),
              "unit": "LM iterations/s", "solves": reps, "iters_per_solve": iters / reps, "trials_per_solve": sum(r["n_trials"]),
              "ms_per_solve_wall": round(tb / reps * 1e3, 3), "ms_per_solve_optimise_phase": round(gpu_ms / reps, 3), "dtype": "f64"}
        # roofline of the reduced-camera-system factorisation, timed live with hipEvents on the solver's own stream
        pkg.Optimizer.timing(True)
        for _ in range(10):
            prep.solve()
        tm = pkg.Optimizer.timing(False)
        if tm["ldlt_launches"] > 0:
            n_sys = 6 * n_free
            flops = n_sys ** 3 / 3.0
            us_l = tm["ldlt_us"] / tm["ldlt_launches"]
            us_s = tm["step_us"] / max(tm["steps"], 1)
            gf = flops / us_l / 1e3
            bytes_trial = E * 144 + L * 96 + (n_sys ** 2) * 8 + E * (8 + 16 + 8 + 1 + 56 + 24)      # Schur reads + S write + residual pass
            ba["roofline"] = {"kernel": "k_ldlt_mfma", "bound": "mfma", "achieved": round(gf / 1e3, 5), "peak": F64_MFMA_PEAK_GFLOPS / 1e3, "unit": "TFLOP/s",
                              "frac": round(gf / F64_MFMA_PEAK_GFLOPS, 6), "traffic": None,
                              "flops_per_launch": int(flops), "n": n_sys, "avg_launch_us": round(us_l, 2),
                              "note": "one workgroup factors one reduced camera system: 1 of %d CUs is busy (frac_of_one_cu %.3f); f64 MFMA peak = public spec"
                                      % (N_CU, gf / (F64_MFMA_PEAK_GFLOPS / N_CU)),
                              "frac_of_one_cu": round(gf / (F64_MFMA_PEAK_GFLOPS / N_CU), 4),
                              "hbm_per_trial": {"bound": "hbm", "algorithmic_bytes": int(bytes_trial), "avg_step_us": round(us_s, 2),
                                                "achieved": round(bytes_trial / us_s / 1e3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                "frac": round(bytes_trial / us_s / 1e3 / HBM_PEAK_GBS, 5)}}
        # the 8-problem batch (config C5's BA half alone): one dcs_ba_local_batch per round
        preps8 = [prep] + [pkg.Optimizer.prepare(synth.ba_problem(seed=43 + s)) for s in range(7)]
        _check_batch(pkg, preps8)
        reps8, it8, t80 = 10, 0, time.perf_counter()
        for _ in range(reps8):
            _check_batch(pkg, preps8)
            it8 += sum(sum(p.res.n_iters) for p in preps8)
        t8 = time.perf_counter() - t80
        ba["batch8"] = {"value": round(it8 / t8, 1), "unit": "LM iterations/s (8 problems per dcs_ba_local_batch call)", "ms_per_call": round(t8 / reps8 * 1e3, 3),
                        "vs_single": round(it8 / t8 / max(ba["value"], 1e-9), 2)}
        if args.cpu_seconds > 0:
            O = entry.load_oracle()
            prob = dict(pb)
            prob["cams"] = [O.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
            aff0 = os.sched_getaffinity(0)
            core = sorted(aff0)[len(aff0) // 2]
            os.sched_setaffinity(0, {core})
            tc0, it_c, n_c = time.perf_counter(), 0, 0
            while n_c < 1 or (time.perf_counter() - tc0 < min(args.cpu_seconds, 8.0) and n_c < 20):
                rc = O.ba_local(prob)
                it_c += sum(rc["n_iters"]); n_c += 1
            tcb = time.perf_counter() - tc0
            os.sched_setaffinity(0, aff0)
            ba["cpu_baseline"] = {"value": round(it_c / tcb, 2), "unit": "LM iterations/s", "cores": 1, "kind": "port", "pinned_core": core,
                                  "sample": "%d solves of the C4 problem in %.1f s, oracle -O3 single thread pinned to core %d (dense LDLT)" % (n_c, tcb, core)}
            ba["speedup_vs_cpu_1thread"] = round(ba["value"] / max(ba["cpu_baseline"]["value"], 1e-9), 1)
        # ---- windows beyond the one-workgroup factorisation (the reference's window is unbounded: Optimizer.cc:415-422 takes every covisible key frame,
        # Optimizer::BundleAdjustment :70-248 the whole map): 60 free poses (n = 360) and a global-BA map (200 KF / 20 000 MP, one round), each with
        # its own single-thread oracle baseline on a bounded sample
        def large_leg(name, kw, gba_iters, reps_l, cpu_cap):
            pbl = synth.ba_problem(**kw)
            if gba_iters:
                pbl = dict(pbl); pbl["iters1"], pbl["iters2"] = gba_iters, 0; pbl["huber_delta"] = float(np.float32(np.sqrt(3.99)))
            prl = pkg.Optimizer.prepare(pbl)
            prl.solve()
            itl, tl0 = 0, time.perf_counter()
            for _ in range(reps_l):
                itl += sum(prl.solve()["n_iters"])
            tl = time.perf_counter() - tl0
            pkg.Optimizer.timing(True)
            rl = prl.solve()
            tml = pkg.Optimizer.timing(False)
            nl = 6 * int((np.asarray(pbl["pose_fixed"]) == 0).sum())
            leg = {"workload": name, "n_reduced_system": nl, "edges": len(pbl["obs"]), "value": round(itl / tl, 1), "unit": "LM iterations/s",
                   "ms_per_solve": round(tl / reps_l * 1e3, 3), "iters_per_solve": itl / reps_l,
                   "factorisation_us_per_trial": round(tml["ldlt_us"] / max(sum(rl["n_trials"]), 1), 1),
                   "factorisation": "k_ldlt_step x %d launches + k_ldlt_back per trial (blocked LDL^T over many workgroups)" % ((nl + 15) // 16)}
            if args.cpu_seconds > 0:
                O2 = entry.load_oracle()
                probl = dict(pbl)
                probl["cams"] = [O2.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pbl["cams"]]
                aff = os.sched_getaffinity(0)
                core2 = sorted(aff)[len(aff) // 2]
                os.sched_setaffinity(0, {core2})
                tq0, it_q, n_q = time.perf_counter(), 0, 0
                while n_q < 1 or (time.perf_counter() - tq0 < min(args.cpu_seconds, cpu_cap) and n_q < 10):
                    it_q += sum(O2.ba_local(probl)["n_iters"]); n_q += 1
                tq = time.perf_counter() - tq0
                os.sched_setaffinity(0, aff)
                leg["cpu_baseline"] = {"value": round(it_q / tq, 2), "unit": "LM iterations/s", "cores": 1, "kind": "port", "pinned_core": core2,
                                       "sample": "%d solves of the same problem in %.1f s, oracle -O3 single thread (dense LDLT)" % (n_q, tq)}
                leg["speedup_vs_cpu_1thread"] = round(leg["value"] / max(leg["cpu_baseline"]["value"], 1e-9), 1)
            return leg
        ba["p60"] = large_leg("LocalBundleAdjustment, 60 free + 7 fixed key frames / 2 400 MP (reduced camera system n = 360)",
                              dict(n_poses=67, n_fixed=6, n_points=2400, obs_per_point=8, seed=61), 0, 10, 4.0)
        out["local_ba"] = ba
        out["global_ba"] = large_leg("Optimizer::BundleAdjustment shape: 200 KF / 20 000 MP / 160 000 edges, one round of 5 iterations (n = 1 194)",
                                     dict(n_poses=200, n_fixed=1, n_points=20000, obs_per_point=8, seed=5), 5, 3, 6.0)

    # ---- C5 on one GPU
    if solo and not args.no_c5 and not args.no_ba:
        out["c5_one_gpu"] = run_c5(pkg, torch, dev, local_rank, args)

    # ---- BoW front-half leg: transform of the step's descriptors with a full-size synthetic vocabulary
    if solo and not args.no_bow and hasattr(pkg.abi.lib(), "dcs_bow_transform_device"):
        voc = synth.vocabulary_fast(10, 6, seed=3)                   # k = 10, L = 6: 1.1 M nodes, 10^6 words (the shape of ORBvoc.txt)
        V = pkg.ORBVocabulary(voc["k"], voc["L"], voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
        d_kp, d_desc, d_n = pipe.last_slots()
        bufs = pkg.ORBVocabulary.bow_buffers(S, cap)
        for _ in range(3):
            V.transform_device(d_desc, d_n, cap, bufs, 4, stream)
        torch.cuda.synchronize()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        b0.record()
        for _ in range(reps):
            V.transform_device(d_desc, d_n, cap, bufs, 4, stream)
        b1.record(); torch.cuda.synchronize()
        bms = b0.elapsed_time(b1) / reps
        n_desc = int(d_n.sum().item())
        bow = {"metric": "BoW transform (10^6-word vocabulary, levelsup 4) Mdescriptors/s", "value": round(n_desc / bms / 1e3, 1), "unit": "Mdescriptors/s",
               "images_per_call": S, "ms_per_call": round(bms, 4), "dtype": "u8 (Hamming) + f64 (word values)", "data": "synthetic vocabulary"}
        if args.cpu_seconds > 0:
            O = entry.load_oracle()
            OV = O.Vocabulary(voc["k"], voc["L"], voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
            n0 = int(d_n[2].item()); h0 = d_desc[2, :n0].cpu().numpy()
            tc0, n_c = time.perf_counter(), 0
            while n_c < 1 or time.perf_counter() - tc0 < min(args.cpu_seconds, 2.0):
                OV.transform(h0, 4); n_c += 1
            tcb = time.perf_counter() - tc0
            bow["cpu_baseline"] = {"value": round(n0 * n_c / tcb / 1e6, 3), "unit": "Mdescriptors/s", "cores": 1, "kind": "port",
                                   "sample": "%d transforms of one image (%d descriptors) in %.1f s, oracle -O3 single thread" % (n_c, n0, tcb)}
            bow["speedup_vs_cpu_1thread"] = round(bow["value"] / max(bow["cpu_baseline"]["value"], 1e-9), 1)
        # the key-frame database query behind loop detection / relocalisation (dcs_kfdb_query): 2000 key frames x ~330 words
        if hasattr(pkg.abi.lib(), "dcs_kfdb_query"):
            kd = synth.keyframe_database(n_db=2000, n_words=100000, words_per_kf=300, n_places=100, seed=5)
            kf = pkg.KeyFrameDatabase()
            for w_, v_ in kd["db"]:
                kf.add(w_, v_)
            qw_, qv_, _ = kd["queries"][0]
            for _ in range(3):
                kf.query(qw_, qv_)
            tq0, nq_ = time.perf_counter(), 50
            for _ in range(nq_):
                kf.query(qw_, qv_)
            tq = (time.perf_counter() - tq0) / nq_
            bow["kfdb_query"] = {"workload": "shared-word counts + first shared word + L1 score of one BowVector (%d words) against %d resident key frames (host buffers out)" % (len(qw_), len(kf)),
                                 "ms_per_query": round(tq * 1e3, 4)}
            if args.cpu_seconds > 0:
                O = entry.load_oracle()
                st_ = dict(query=np.full(len(kf), -1, np.int32), words=np.zeros(len(kf), np.int32), score=np.zeros(len(kf), np.float32))
                tq0 = time.perf_counter()
                O.detect_candidates(0, 1, qw_, qv_, kd["db"], np.zeros(len(kf), np.uint8), kd["covis"], st_)
                bow["kfdb_query"]["cpu_ms_per_detect_call"] = round((time.perf_counter() - tq0) * 1e3, 3)
                bow["kfdb_query"]["cpu_note"] = "oracle DetectRelocalizationCandidates incl. building the inverted files from the flat database (kind: port, 1 core)"
            kf.close()
        out["bow"] = bow
        V.close()

    # ---- the Tracking thread's steady-state chain per frame: SearchLocalPoints + PoseOptimization, device-resident and
    # batched (dcs_track_local_map) against the three host-buffer calls per frame it replaces, and the oracle on one host core
    if solo and not args.no_host_api and hasattr(pkg.abi.lib(), "dcs_track_local_map"):
        fr_t, prm_t = synth.tracking_problem(n_frames=16, n_points=2000, n_features=2000, seed=23)
        for f_ in fr_t:
            ft_ = f_["features"]
            ft_["grid_off"], ft_["grid_idx"] = pkg.frame_grid(ft_["cam_off"], ft_["kp_x"], ft_["kp_y"], ft_["min_x"], ft_["min_y"], ft_["grid_w_inv"], ft_["grid_h_inv"])

        def med(fn, reps=30):
            for _ in range(3):
                fn()
            ts_ = []
            for _ in range(reps):
                t0_ = time.perf_counter(); fn(); ts_.append(time.perf_counter() - t0_)
            return sorted(ts_)[reps // 2]
        chain = {"workload": "per frame: isInFrustum of 2000 local map points -> SearchByProjection into ~2000 features (dual camera) -> PoseOptimization over the "
                             "features that hold a map point (4 x 10 LM iterations); host buffers in, pose + assignment + outlier flags out"}
        for nf_ in (1, 16):
            pt_ = pkg.abi.PreparedTracking(fr_t[:nf_], prm_t)
            chain["chained_ms_per_frame_batch_of_%d" % nf_] = round(med(pt_.track) / nf_ * 1e3, 4)
        with pkg.abi.options(DCS_POSE_EXACT_EDGE=1):            # the other build of k_pose_opt2 (every edge's point and residual through the oracle's own operations)
            pt_ = pkg.abi.PreparedTracking(fr_t[:1], prm_t)
            chain["exact_edge_build"] = {"chained_ms_per_frame_batch_of_1": round(med(pt_.track) * 1e3, 4),
                                         "what": "option DCS_POSE_EXACT_EDGE=1: rounds of PoseOptimization one LM iteration from the oracle in 9.2 % of random batches "
                                                 "instead of 16.8 % (profiles/r06_pose_flip_stats.txt)"}
        m_t = pkg.ORBmatcher(prm_t["nn_ratio"], False)

        def three_calls():
            f_ = fr_t[0]
            ft_, pts_ = f_["features"], f_["points"]
            fru_ = pkg.isInFrustum(f_["view"], pts_, prm_t["viewing_cos_limit"], prm_t["th"])
            q_ = pkg.projection_queries(fru_, f_["desc"])
            mq_, qf_, _nm = m_t.SearchByProjection(ft_, q_, prm_t["th_high"], use_ratio=True, check_orientation=False)
            src_ = np.where(qf_ >= 0, qf_, np.where(f_["has_point"] != 0, -2, -1))
            feat_ = np.nonzero(src_ != -1)[0]
            xw_ = np.where((src_[feat_] >= 0)[:, None], pts_["pos"][np.maximum(src_[feat_], 0)], f_["point_xw"][feat_]).astype(np.float64)
            pkg.Optimizer.PoseOptimization(dict(poses=f_["pose"][None, :], edge_off=np.array([0, len(feat_)], np.int32), xw=xw_,
                                                obs=np.stack([ft_["kp_x"][feat_], ft_["kp_y"][feat_]], 1).astype(np.float64),
                                                inv_sigma2=prm_t["inv_level_sigma2"][ft_["kp_octave"][feat_]].astype(np.float64),
                                                edge_cam=(np.searchsorted(ft_["cam_off"], feat_, side="right") - 1).astype(np.int32), cams=prm_t["cams"],
                                                huber_delta=prm_t["huber_delta"], chi2_th=prm_t["chi2_th"], its=prm_t["its"]))
        chain["three_calls_ms_per_frame"] = round(med(three_calls) * 1e3, 4)
        chain["note"] = "median of 30 calls incl. ctypes / numpy marshalling of the Python harness on both sides"
        if args.cpu_seconds > 0:
            O_ = entry.load_oracle()
            cams_o = [O_.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in prm_t["cams"]]

            def oracle_chain():
                f_ = fr_t[0]
                ft_, pts_ = f_["features"], f_["points"]
                fru_ = O_.is_in_frustum(f_["view"], pts_, prm_t["viewing_cos_limit"], prm_t["th"])
                q_ = pkg.projection_queries(fru_, f_["desc"])
                mq_, qf_, _nm = O_.search_by_projection(ft_, q_, prm_t["th_high"], prm_t["nn_ratio"], False)
                src_ = np.where(qf_ >= 0, qf_, np.where(f_["has_point"] != 0, -2, -1))
                feat_ = np.nonzero(src_ != -1)[0]
                xw_ = np.where((src_[feat_] >= 0)[:, None], pts_["pos"][np.maximum(src_[feat_], 0)], f_["point_xw"][feat_]).astype(np.float64)
                O_.pose_optimization(dict(poses=f_["pose"][None, :], edge_off=np.array([0, len(feat_)], np.int32), xw=xw_,
                                          obs=np.stack([ft_["kp_x"][feat_], ft_["kp_y"][feat_]], 1).astype(np.float64),
                                          inv_sigma2=prm_t["inv_level_sigma2"][ft_["kp_octave"][feat_]].astype(np.float64),
                                          edge_cam=(np.searchsorted(ft_["cam_off"], feat_, side="right") - 1).astype(np.int32), cams=cams_o,
                                          huber_delta=prm_t["huber_delta"], chi2_th=prm_t["chi2_th"], its=prm_t["its"]))
            chain["cpu_baseline"] = {"value": round(med(oracle_chain, 10) * 1e3, 4), "unit": "ms per frame", "cores": 1, "kind": "port",
                                     "sample": "median of 10 frames, oracle -O3 single thread (same marshalling)"}
        # roofline-style entry of the chain's dominant kernel (k_pose_opt2: one workgroup per frame, 60 dependent passes): f64 operations of the sweeps
        # + the scalar part per pass against the f64 vector peak of ONE CU; its duration is imported from the committed rocprofv3 trace of
        # scratch/time_track.py (profiles/r06_track_kernel_stats.csv), the same frames as this leg
        try:
            import csv as _csv
            with open(os.path.join(ROOT, "profiles", "r06_track_kernel_stats.csv")) as fh:
                row = next(r for r in _csv.DictReader(fh) if "k_pose_opt2" in r["Name"])
            us_k = float(row["MinNs"]) / 1e3                   # batch-of-1 launches (one workgroup) are the minimum of the trace
            edges = int((pkg.abi.PreparedTracking(fr_t[:1], prm_t).track()[0]["point_of_feature"] != -1).sum())
            passes, f64_per_edge, f64_scalar = 66, 2 * 107, 2 * 400      # measured passes of this frame (62 trials + 4 classifying passes); FMA = 2 flops
            flops = passes * (edges * f64_per_edge + 4 * f64_scalar)
            peak_cu = F64_VECTOR_PEAK_GFLOPS / N_CU
            chain["pose_kernel"] = {"kernel": "k_pose_opt2", "edges": edges, "us_batch_of_1": round(us_k, 1), "bound": "latency (dependent f64 chain on one CU)",
                                    "achieved": round(flops / us_k / 1e3, 2), "peak": round(peak_cu, 1), "unit": "GFLOP/s of ONE CU (f64 vector)",
                                    "frac": round(flops / us_k / 1e3 / peak_cu, 4), "source": "imported: profiles/r06_track_kernel_stats.csv"}
        except Exception:
            pass
        out["per_frame_chain"] = chain

    # ---- one dual frame from host images to the pose, the reference's steady state per frame (Frame ctor -> TrackWithMotionModel -> TrackLocalMap,
    # src/Tracking.cc:1384-1427, 1617-1680): images up, extraction into slots in HBM, then dcs_track_frame_device twice -- the features never
    # leave the device, the map side comes from the host both times. The map is fitted to what the extractor found (synth.scene_from_features).
    if solo and not args.no_host_api and hasattr(pkg.abi.lib(), "dcs_track_frame_device"):
        a_, b_ = synth.frame_pair(W, H, 0, 3)
        ext2 = pkg.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, max_images=2)
        cap2 = ext2.default_cap(H, W)
        h_img = torch.from_numpy(np.stack([a_, b_])).pin_memory()
        d_img = torch.empty(h_img.shape, dtype=torch.uint8, device="cuda")
        d_kp2 = torch.zeros((2, cap2, 7), dtype=torch.float32, device="cuda"); d_desc2 = torch.zeros((2, cap2, 32), dtype=torch.uint8, device="cuda")
        d_n2 = torch.zeros(2, dtype=torch.int32, device="cuda")
        st2 = torch.cuda.current_stream().cuda_stream

        def extract2():
            d_img.copy_(h_img, non_blocking=True)
            ext2.extract_batch_device(d_img, d_kp2, d_desc2, d_n2, cap2, stream=st2)
        extract2(); torch.cuda.synchronize()
        n2 = d_n2.cpu().numpy()
        kp2 = d_kp2.cpu().numpy().reshape(2, cap2, 7).copy().view(pkg.abi.KEYPOINT).reshape(2, cap2)
        de2 = d_desc2.cpu().numpy()
        # the shipped rig's intrinsics and distortion (Dual-LenaCV.yaml:12-35, k1 = -0.37): Frame::UndistortKeyPoints runs inside dcs_track_frame_device; the map
        # is fitted to the UNDISTORTED key points and the image bounds are the undistorted corners (Frame::ComputeImageBounds, Frame.cc:454-476)
        K_rig = np.array([[synth.RIG["cam%d" % c][k] for k in ("fx", "fy", "cx", "cy")] for c in (0, 1)], np.float32)
        dist_rig = np.array([[-0.3689, 0.1627, 0.0, 0.0, 0.0], [-0.361851421593862, 0.140443638558527, 0.0, 0.0, 0.0]], np.float32)
        kp_u, bnd = [], [[], [], [], []]
        for c in (0, 1):
            kc = kp2[c][:n2[c]].copy()
            u_ = pkg.abi.undistort_points(np.stack([kc["x"], kc["y"]], 1), K_rig[c], dist_rig[c])
            kc["x"], kc["y"] = u_[:, 0], u_[:, 1]
            kp_u.append(kc)
            cr = pkg.abi.undistort_points(np.array([[0, 0], [W, 0], [0, H], [W, H]], np.float32), K_rig[c], dist_rig[c])
            for b_, v_ in zip(bnd, (min(cr[0, 0], cr[2, 0]), max(cr[1, 0], cr[3, 0]), min(cr[0, 1], cr[1, 1]), max(cr[2, 1], cr[3, 1]))):
                b_.append(v_)
        fs_, prm_s = synth.scene_from_features(kp_u, [de2[c][:n2[c]] for c in (0, 1)], bounds=bnd)
        dev_ = dict(d_kp=d_kp2.data_ptr(), d_desc=d_desc2.data_ptr(), d_n=d_n2.data_ptr(), cap=cap2, first_slot=0, n_cams=2, K=K_rig, dist=dist_rig, **fs_["grid"])
        mm_ = fs_["mm"]
        prm_mm = dict(prm_s); prm_mm["th"] = 7.0; prm_mm["nn_ratio"] = 0.0
        N2 = int(n2.sum())
        held_ = dict(taken=np.zeros(N2, np.uint8), has_point=np.zeros(N2, np.uint8), point_xw=np.zeros((N2, 3), np.float32))
        pts_lm = dict(fs_["points"]); pts_lm["candidate"] = np.ones(len(pts_lm["pos"]), np.uint8)
        pt_mm = pkg.abi.PreparedTrackingDevice([dict(dev=dev_, view=fs_["view"], pose=fs_["pose"], held=None, points=dict(pos=mm_["pos"]), desc=mm_["desc"],
                                                      q_cam=mm_["q_cam"], q_octave=mm_["q_octave"], q_angle=mm_["q_angle"])], prm_mm, mode=1, check_orientation=True, stream=st2)
        pt_lm = pkg.abi.PreparedTrackingDevice([dict(dev=dev_, view=fs_["view"], pose=fs_["pose"], held=held_, points=pts_lm, desc=fs_["desc"])], prm_s, mode=0, stream=st2)
        # the arrays the prepared call points at (updated in place between the two stages)
        arrs = {id(x): x for x in pt_lm.keep}
        tk_, hp_, px_, cd_ = held_["taken"], held_["has_point"], held_["point_xw"], pts_lm["candidate"]
        assert all(id(x) in arrs for x in (tk_, hp_, px_, cd_)), "the prepared call copied an array the harness updates in place"

        def glue(r1):                                        # TrackWithMotionModel's bookkeeping (Tracking.cc:1429-1448) + the candidate flags of SearchLocalPoints (:1625-1640)
            pof = r1["point_of_feature"]
            good = (pof >= 0) & (r1["outlier"] == 0)
            tk_[:] = good; hp_[:] = good
            px_[good] = mm_["pos"][pof[good]]
            cd_[:] = 1; cd_[mm_["point"][pof[good]]] = 0

        def whole():
            extract2()
            r1 = pt_mm.track()[0]
            glue(r1)
            return r1, pt_lm.track()[0]
        r1_, r2_ = whole()
        tot_ = {}

        def stage(fn, reps=40, sync=False):
            for _ in range(3):
                fn()
            ts_ = []
            for _ in range(reps):
                t0_ = time.perf_counter(); fn()
                if sync:
                    torch.cuda.synchronize()
                ts_.append(time.perf_counter() - t0_)
            return round(sorted(ts_)[reps // 2] * 1e3, 4)
        tot_["ms_total"] = stage(whole)
        tot_["ms_upload_extract_alone"] = stage(extract2, sync=True)
        tot_["ms_motion_model_stage_alone"] = stage(lambda: pt_mm.track())
        tot_["ms_local_map_stage_alone"] = stage(lambda: pt_lm.track())
        out["per_frame_total"] = dict(workload="1 dual 640x480 frame: 2 page-locked host images up -> dcs_orb_extract_batch_device (slots in HBM) -> dcs_track_frame_device mode 1 "
                                               "(Frame::UndistortKeyPoints with the rig file's k1 = -0.37, TrackWithMotionModel: SearchByProjectionOnCam x 2 with rotation histograms + PoseOptimization) -> host bookkeeping -> "
                                               "dcs_track_frame_device mode 0 (SearchLocalPoints + PoseOptimization); features stay on the device, map arrays from the host; "
                                               "the local-map stage starts from the same pose guess (the harness does not rebuild the view matrices); median of 40",
                                      features=int(N2), mm_queries=int(len(mm_["pos"])), mm_matches=int(r1_["n_matches"]), mm_inliers=int(r1_["n_inliers"]),
                                      local_map_points=int(len(pts_lm["pos"])), lm_new_matches=int(r2_["n_matches"]), lm_inliers=int(r2_["n_inliers"]), **tot_)
        ext2.close()

    # ---- distributed runs: the rank-path legs (configs[2] and configs[4] as an N-GPU run measures them), then the OTHER exchange path once
    # as a cross-check of the gathered slot arrays. RCCL has never run with more than one rank on the builder's side, so a watchdog prints
    # the line with whatever is complete if a collective does not come back.
    if distributed:
        printed = threading.Event()

        def emit():
            if rank == 0 and not printed.is_set():
                printed.set()
                print(json.dumps(out), flush=True)

        done = threading.Event()
        phase = {"name": "rank legs"}

        def watchdog():
            if not done.wait(240.0):
                if rank == 0:
                    out["watchdog"] = "timed out in: %s" % phase["name"]
                emit()
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        env = RankEnv(torch, dist, dev, rank, world)
        if not args.no_rank_legs:
            res = {}
            try:
                phase["name"] = "c3_scaled"
                res["c3_scaled"] = leg_c3_scaled(env, GpuC3(pkg, torch, dist, sharding, dev, local_rank, rank, world, comm, stream, exchange_impl, 16), steps=8, warmup=2)
            except Exception as e:                   # noqa: BLE001
                res["c3_scaled"] = "unavailable: %s" % e
            if not args.no_ba:
                try:
                    phase["name"] = "c5_node"
                    leaves = gpu_c5(pkg, torch, dev, local_rank, rank)
                    res["c5_node"] = leg_c5_node(env, *leaves[:5], seconds=args.rank_leg_seconds)
                    leaves[5]()
                except Exception as e:               # noqa: BLE001
                    res["c5_node"] = "unavailable: %s" % e
            if rank == 0:
                out.update(res)
        phase["name"] = "exchange cross-check"
        try:
            d_kp, d_desc, d_n = pipe.last_slots()
            if comm is not None:                     # timed region used the C ABI: compare with torch.distributed
                sharding.pack_features(d_kp[S - 2:], d_desc[S - 2:], d_n[S - 2:], cap, xchg.g_send)
                dist.all_gather_into_tensor(xchg.g_recv, xchg.g_send)
                t_kp, t_desc, t_n = sharding.unpack_features(xchg.g_recv, cap)
                comm.allgather_features(d_kp[S - 2:], d_desc[S - 2:], d_n[S - 2:], cap, xchg.g_kp, xchg.g_desc, xchg.g_n, stream)
                torch.cuda.synchronize()
                same = bool(torch.equal(t_n, xchg.g_n) and torch.equal(t_desc, xchg.g_desc) and torch.equal(t_kp.view(torch.int32), xchg.g_kp.view(torch.int32)))
                if rank == 0:
                    out["exchange_crosscheck"] = {"other_path": "torch.distributed all_gather_into_tensor + pack / unpack", "slot_arrays_equal": same}
                comm.close()
            elif rank == 0:
                out["exchange_crosscheck"] = "skipped (the C-ABI communicator is not up)"
        except Exception as e:                       # noqa: BLE001
            if rank == 0:
                out["exchange_crosscheck"] = "unavailable: %s" % e
        done.set()
        emit()
    elif rank == 0:
        print(json.dumps(out))
    pipe.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
