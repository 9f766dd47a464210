"""C5-shaped front end alone (8 dual 1280x720 streams, one dual frame each per step): is it bound by the host's launch rate?
   rocprofv3 --kernel-trace --output-format csv -d /tmp/c5b -- python scratch/c5_front_busy.py run ; python scratch/c5_front_busy.py show /tmp/c5b"""
import sys, os, glob, csv, time
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import torch, bench
    import __graft_entry__ as entry
    pkg = entry.load_package()
    dev = torch.device("cuda:0")
    p = bench.Pipeline(pkg, torch, dev, 0, 1280, 720, 2000, 8, 1, 1, 0, 64)
    for e_ in p.exts:
        e_.set_timing(0)
    dt = p.run(200, 5)
    print("front end alone: %.1f kfeatures/s, %.3f ms per step" % (p.features_per_step() * 200 / dt / 1e3, dt / 200 * 1e3))
else:
    ev = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-30:], r.get("Queue_Id", "")))
    ev.sort()
    ev = ev[len(ev) // 2:]                                    # the second half: steady state
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("span %.2f ms, some kernel running %.1f %% of it, %d kernels (%.1f us each on average incl. overlap)" % ((t1 - t0) / 1e6, 100.0 * busy / (t1 - t0), len(ev), sum(e - s for s, e, _, _ in ev) / len(ev) / 1e3))
    import collections
    q = collections.Counter(x[3] for x in ev); print("kernels per queue:", dict(q))
