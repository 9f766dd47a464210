#!/usr/bin/env python3
"""All-host-cores CPU baseline for bench.py: N worker PROCESSES (fork; no GPU runtime in this process tree), each running
the oracle on its own stream of dual frames for a fixed time: 2 extractions + 3 knn2 / ratio / rotation-histogram matches
per dual frame, exactly what one bench step does per dual frame. Prints one JSON line."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FRAMES = None
ARGS = None


def worker(t):
    import oracle as O
    o = O.OrbOracle(ARGS.nfeatures, 1.2, 8, 20, 7)
    prev, feats, k = None, 0, 0
    t_end = time.perf_counter() + ARGS.seconds
    while time.perf_counter() < t_end:
        a, b = FRAMES[(t + k) % len(FRAMES)]
        ka, da = o.extract(a)
        kb, db = o.extract(b)
        jobs = [(da, ka, db, kb)]
        jobs += [(da, ka, prev[0], prev[1]), (db, kb, prev[2], prev[3])] if prev is not None else [(da, ka, da, ka), (db, kb, db, kb)]
        for (q, kq, tt, kt) in jobs:
            bi, bd, sd = O.knn2(q, tt)
            O.ratio_rot_filter(bi, bd, sd, 50, False, 0.75, True, kq["angle"], kt["angle"])
        prev = (da, ka, db, kb)
        feats += len(ka) + len(kb)
        k += 1
    return feats


def main():
    global FRAMES, ARGS
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ARGS = ap.parse_args()
    import oracle as O
    O.build(); O.lib()
    from conftest import load_pkg
    synth = load_pkg().synth
    FRAMES = [synth.frame_pair(ARGS.width, ARGS.height, 0, f) for f in range(4)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(ARGS.procs) as pool:
        feats = pool.map(worker, range(ARGS.procs), chunksize=1)
    dt = time.perf_counter() - t0
    print(json.dumps({"features": int(sum(feats)), "seconds": dt, "procs": ARGS.procs, "kfeatures_s": sum(feats) / dt / 1e3}))


if __name__ == "__main__":
    main()
