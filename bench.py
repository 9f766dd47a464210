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
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def pyramid_px(pkg, ext):
    dims = [ext.level_dims(l) for l in range(ext.nlevels)]
    px = [w * h for w, h in dims]
    return px


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=256, help="dual frames per step per GPU")
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-bow", action="store_true")
    ap.add_argument("--lanes", type=int, default=1, help="independent extractor handles/streams the batch is split over (overlaps latency-bound kernels)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    pkg = entry.load_package()
    synth = pkg.synth
    dev = torch.device("cuda", local_rank)

    P, W, H, NF = args.pairs, args.width, args.height, args.nfeatures
    # ---- HBM-resident synthetic input: stream = rank, n_unique distinct dual frames tiled to P
    n_unique = min(P, 8)
    frames = []
    for f in range(n_unique):
        frames.extend(synth.frame_pair(W, H, rank, f))
    imgs = np.stack([frames[(2 * (p % n_unique)) + c] for p in range(P) for c in (0, 1)])
    d_img = torch.from_numpy(imgs).to(dev)

    n_lanes = max(1, min(args.lanes, P))
    lane_pairs = [P // n_lanes + (1 if i < P % n_lanes else 0) for i in range(n_lanes)]
    exts = [pkg.ORBextractor(NF, 1.2, 8, 20, 7, device=local_rank, max_images=2 * lp) for lp in lane_pairs]
    ext = exts[0]
    lane_streams = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)]
    lane_done = [torch.cuda.Event() for _ in range(n_lanes)]
    lane_go = torch.cuda.Event()
    matcher = pkg.ORBmatcher(0.75, True)
    cap = ext.default_cap()
    S = 2 * P + 2                                   # feature slots: [prev cam0, prev cam1, batch ...]
    # two slot sets: matching of step i (main stream) overlaps the extraction of step i + 1 (lane streams)
    NB = 2
    d_kp_b = [torch.zeros((S, cap, 7), dtype=torch.float32, device=dev) for _ in range(NB)]
    d_desc_b = [torch.zeros((S, cap, 32), dtype=torch.uint8, device=dev) for _ in range(NB)]
    d_n_b = [torch.zeros(S, dtype=torch.int32, device=dev) for _ in range(NB)]
    match_done = [torch.cuda.Event() for _ in range(NB)]
    pairs = []
    for f in range(P):
        c0, c1 = 2 + 2 * f, 3 + 2 * f
        pairs += [(c0, c1), (c0, c0 - 2), (c1, c1 - 2)]
    n_pairs = len(pairs)
    d_pairs = torch.tensor(pairs, dtype=torch.int32, device=dev)
    d_match = torch.zeros((n_pairs, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
    d_b = torch.zeros((n_pairs, cap), dtype=torch.int32, device=dev)
    d_s = torch.zeros((n_pairs, cap), dtype=torch.int32, device=dev)
    if world > 1:
        from orb_slam2_dualcam_amd import sharding
        rec = sharding.record_bytes(cap)             # per camera: kp 28 B + desc 32 B per slot + count
        g_send = torch.zeros((2, rec), dtype=torch.uint8, device=dev)
        g_recv = torch.zeros((2 * world, rec), dtype=torch.uint8, device=dev)
        g_kp = torch.zeros((2 * world, cap, 7), dtype=torch.float32, device=dev)
        g_desc = torch.zeros((2 * world, cap, 32), dtype=torch.uint8, device=dev)
        g_n = torch.zeros(2 * world, dtype=torch.int32, device=dev)
        x_pairs = torch.tensor(sharding.reloc_pairs(rank, world), dtype=torch.int32, device=dev)
        x_match = torch.zeros((world - 1, cap), dtype=torch.int32, device=dev)
        x_nm = torch.zeros(world - 1, dtype=torch.int32, device=dev)
        x_b = torch.zeros((world - 1, cap), dtype=torch.int32, device=dev)
        x_s = torch.zeros((world - 1, cap), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    stage_keys = ("pyramid_us", "fast_us", "compact_us", "blur_us", "quadtree_us", "describe_us", "total_us")
    acc = {k: 0.0 for k in stage_keys}
    acc["match_us"] = 0.0
    acc["allgather_us"] = 0.0
    # per-step torch events (match, all-gather) are read AFTER the timed region so that nothing stalls the stream
    ev_all = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps + args.warmup)]
    step_no = [0]

    def step(timed):
        it = step_no[0]
        ev = ev_all[it]
        step_no[0] += 1
        cur, prv = it % NB, (it - 1) % NB
        d_kp, d_desc, d_n = d_kp_b[cur], d_desc_b[cur], d_n_b[cur]
        # the batch is split over `n_lanes` extractor handles on their own streams; they only wait for the matcher to
        # be done with this slot set (step it - 2), so extraction of step it overlaps matching of step it - 1
        first = 0
        for li in range(n_lanes):
            a, b = 2 * first, 2 * (first + lane_pairs[li])
            if it >= NB:
                lane_streams[li].wait_event(match_done[cur])
            exts[li].extract_batch_device(d_img[a:b], d_kp[2 + a:2 + b], d_desc[2 + a:2 + b], d_n[2 + a:2 + b], cap,
                                          stream=lane_streams[li].cuda_stream)
            lane_done[li].record(lane_streams[li])
            first += lane_pairs[li]
        for li in range(n_lanes):
            torch.cuda.current_stream().wait_event(lane_done[li])
        # newest dual frame of the previous step is "t-1" of this one
        d_kp[0:2].copy_(d_kp_b[prv][S - 2:]); d_desc[0:2].copy_(d_desc_b[prv][S - 2:]); d_n[0:2].copy_(d_n_b[prv][S - 2:])
        ev[0].record()
        matcher.match_bf_batch_device(d_desc, d_kp, d_n, cap, d_pairs, n_pairs, d_match, d_nm, d_b, d_s, 50, stream=stream)
        ev[1].record()
        if world > 1:
            # exchange the newest dual frame: pack (kp | desc | n) per camera, one all-gather, cross-GPU reloc match
            sharding.pack_features(d_kp[S - 2:], d_desc[S - 2:], d_n[S - 2:], cap, g_send)
            ev[2].record()
            dist.all_gather_into_tensor(g_recv, g_send)
            ev[3].record()
            sharding.unpack_features(g_recv, cap, g_kp, g_desc, g_n)
            matcher.match_bf_batch_device(g_desc, g_kp, g_n, cap, x_pairs, world - 1, x_match, x_nm, x_b, x_s, 50, stream=stream)
        match_done[cur].record()
        return None

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    for e_ in exts:
        e_.timing_totals(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e_ in exts:                       # kernel times: summed over the lanes (each lane launches its own kernels)
        sums, n_timed = e_.timing_totals()
        for k in stage_keys:
            acc[k] += sums[k]
    for ev in ev_all[args.warmup:]:
        acc["match_us"] += ev[0].elapsed_time(ev[1]) * 1000.0
        if world > 1:
            acc["allgather_us"] += ev[2].elapsed_time(ev[3]) * 1000.0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    n_feat_step = int(d_n_b[(step_no[0] - 1) % NB][2:].sum().item())
    n_match_step = int(d_nm.sum().item())
    tot = torch.tensor([n_feat_step], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tot)
    feats_all = float(tot.item())
    value = feats_all * args.steps / dt / 1000.0

    out = None
    if rank == 0:
        px = pyramid_px(pkg, ext)
        sum_px, px7, sum_17 = sum(px), px[-1], sum(px[1:])
        n_img = 2 * P
        n_avg = n_feat_step / n_img
        K = args.steps
        LN = n_lanes                                 # every extraction kernel is launched once per lane per step
        n_img_l = n_img / LN
        algo = {                                     # algorithmic bytes per LAUNCH x images per launch
            "k_resize(x7)": ((sum_px - px7) + sum_17) * n_img_l,
            "k_fast_cells": sum_px * n_img_l,
            "k_blur": 2 * sum_px * n_img_l,
            "k_describe": int((749 + 512 + 60) * n_avg * n_img_l),
            "k_knn2_pairs_mfma+k_filter_pairs": int((32 * 2 * n_avg + 12 * n_avg) * n_pairs),
        }
        dur = {"k_resize(x7)": acc["pyramid_us"] / (K * LN), "k_fast_cells": acc["fast_us"] / (K * LN),
               "k_blur": acc["blur_us"] / (K * LN), "k_describe": acc["describe_us"] / (K * LN),
               "k_knn2_pairs_mfma+k_filter_pairs": acc["match_us"] / K}
        kernels = {k: dict(us=round(dur[k], 2), algo_bytes=int(algo[k]),
                           gbps=round(algo[k] / max(dur[k], 1e-3) / 1e3, 2)) for k in algo}
        dom = max((k for k in dur if k != "k_knn2_pairs_mfma+k_filter_pairs"), key=lambda k: dur[k])
        traffic = None                               # HBM bytes per launch from the committed PMC passes (profiles/), same workload only
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
            if (P, W, H, NF, LN) == tuple(pmc.get("bench_args", ())):
                kk = pmc["kernels"][dom.split("(")[0].split("+")[0]]
                traffic = int((kk["FETCH_SIZE_KB_avg_per_launch"] + kk["WRITE_SIZE_KB_avg_per_launch"]) * 1024)
        except Exception:
            traffic = None
        roofline = dict(kernel=dom, bound="hbm", achieved=kernels[dom]["gbps"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(kernels[dom]["gbps"] / HBM_PEAK_GBS, 5), traffic=traffic,
                        algorithmic_bytes_per_launch=int(algo[dom]), avg_launch_us=round(dur[dom], 2))
        out = {
            "metric": "dual-frame ORB extract+match kfeatures/s; local-BA iters/s (50 KF / 2k MP)",
            "value": round(value, 2), "unit": "kfeatures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: dual %dx%d stream, %d feat/cam, 8 levels, extract + BF match (3 matches/dual frame)" % (W, H, NF),
                       "dual_frames_per_step_per_gpu": P, "extractor_lanes": n_lanes, "features_per_step_per_gpu": n_feat_step,
                       "matches_per_step_per_gpu": n_match_step, "parallelism": "frame-pair shard x%d" % world},
            "roofline": roofline,
            "stage_us_per_step": {k: round(v / K, 2) for k, v in acc.items()},
            "kernels": kernels,
        }

    # ---- CPU baseline (rank 0, N = 1 only): the oracle ("port"), single thread, bounded sample
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        O = entry.load_oracle()
        o = O.OrbOracle(NF, 1.2, 8, 20, 7)
        prev = None
        feats = 0
        n_done = 0
        tc0 = time.perf_counter()
        while time.perf_counter() - tc0 < args.cpu_seconds and n_done < 400:
            f = n_done % n_unique
            a, b = frames[2 * f], frames[2 * f + 1]
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
            prev = (da, ka, db, kb)
            feats += len(ka) + len(kb)
            n_done += 1
        tc = time.perf_counter() - tc0
        out["cpu_baseline"] = {"value": round(feats / tc / 1000.0, 3), "unit": "kfeatures/s", "cores": 1, "kind": "port",
                               "sample": "%d dual %dx%d frames (extract x2 + 3 knn2/filter each) in %.1f s, oracle -O3 single thread; host has %d cores"
                                         % (n_done, W, H, tc, os.cpu_count())}
        out["speedup_vs_cpu_1thread"] = round(out["value"] / max(out["cpu_baseline"]["value"], 1e-9), 1)

    # ---- local BA leg (C4: 50 KF / 2000 MP / 20k dual-camera edges), rank 0, N = 1
    if rank == 0 and world == 1 and not args.no_ba and hasattr(pkg.abi.lib(), "dcs_ba_local"):
        pb = synth.ba_problem()
        prep = pkg.Optimizer.prepare(pb)                             # flat problem marshalled once, as a C++ caller holds it
        prep.solve()                                                 # warm-up (allocations, code objects)
        reps, iters, tb0 = 10, 0, time.perf_counter()
        gpu_ms = 0.0
        for _ in range(reps):
            r = prep.solve()
            iters += sum(r["n_iters"])
            gpu_ms += r["gpu_ms"]
        tb = time.perf_counter() - tb0
        ba = {"metric": "local-BA iters/s (50 KF / 2k MP / %d edges)" % len(pb["obs"]), "value": round(iters / tb, 1),
              "unit": "LM iterations/s", "iters_per_solve": iters / reps, "ms_per_solve_wall": round(tb / reps * 1e3, 2),
              "ms_per_solve_optimise_phase": round(gpu_ms / reps, 2), "dtype": "f64"}
        if args.cpu_seconds > 0:
            O = entry.load_oracle()
            prob = dict(pb)
            prob["cams"] = [O.make_camera(c["fx"], c["fy"], c["cx"], c["cy"], c["ext7"], c["adj"]) for c in pb["cams"]]
            tc0, it_c, n_c = time.perf_counter(), 0, 0
            while n_c < 1 or (time.perf_counter() - tc0 < min(args.cpu_seconds, 8.0) and n_c < 20):
                rc = O.ba_local(prob)
                it_c += sum(rc["n_iters"]); n_c += 1
            tcb = time.perf_counter() - tc0
            ba["cpu_baseline"] = {"value": round(it_c / tcb, 2), "unit": "LM iterations/s", "cores": 1, "kind": "port",
                                  "sample": "%d solves of the C4 problem in %.1f s, oracle -O3 single thread (dense LDLT)" % (n_c, tcb)}
            ba["speedup_vs_cpu_1thread"] = round(ba["value"] / max(ba["cpu_baseline"]["value"], 1e-9), 1)
        out["local_ba"] = ba

    # ---- BoW front-half leg: transform of the step's descriptors with a full-size synthetic vocabulary
    if rank == 0 and world == 1 and not args.no_bow and hasattr(pkg.abi.lib(), "dcs_bow_transform_device"):
        voc = synth.vocabulary_fast(10, 6, seed=3)                   # k = 10, L = 6: 1.1 M nodes, 10^6 words (the shape of ORBvoc.txt)
        V = pkg.ORBVocabulary(voc["k"], voc["L"], voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
        d_kp, d_desc, d_n = d_kp_b[(step_no[0] - 1) % NB], d_desc_b[(step_no[0] - 1) % NB], d_n_b[(step_no[0] - 1) % NB]
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
        out["bow"] = bow
        V.close()

    if rank == 0:
        print(json.dumps(out))
    for e_ in exts:
        e_.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
