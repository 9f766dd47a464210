"""one dual frame per call (dcs_orb_extract_batch on 2 host images + dcs_match_bf): the kernel / copy timeline of the median call from a
rocprofv3 trace. Usage on the GPU box:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lt -- python scratch/latency_trace.py run
   python scratch/latency_trace.py show /tmp/lt"""
import sys, os, glob, csv, time
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ctypes as C, numpy as np
    import __graft_entry__ as e
    pkg = e.load_package(); synth = pkg.synth
    W, H, NF = 640, 480, 1000
    a, b = synth.frame_pair(W, H, 0, 0)
    ext1 = pkg.ORBextractor(NF, 1.2, 8, 20, 7, max_images=2)
    cap = ext1.default_cap()
    L_ = pkg.abi.lib()
    kp1 = np.zeros((2, cap), pkg.abi.KEYPOINT); de1 = np.zeros((2, cap, 32), np.uint8); n1 = np.zeros(2, np.int32)
    mt1 = np.zeros(cap, np.int32); nm1 = C.c_int()
    ptr2 = (C.c_void_p * 2)(a.ctypes.data, b.ctypes.data)
    vp_ = lambda x: x.ctypes.data_as(C.c_void_p)
    ts = []
    for i in range(60):
        t0 = time.perf_counter()
        assert L_.dcs_orb_extract_batch(ext1._h, C.cast(ptr2, C.c_void_p), 2, H, W, W, vp_(kp1), vp_(de1), cap, vp_(n1)) == 0
        t1 = time.perf_counter()
        assert L_.dcs_match_bf(vp_(de1[0]), vp_(kp1[0]), int(n1[0]), vp_(de1[1]), vp_(kp1[1]), int(n1[1]), 50, C.c_float(0.75), 1, vp_(mt1), C.byref(nm1)) == 0
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    ts = ts[10:]
    print("host: extract median %.1f us, match median %.1f us" % (sorted(x[0] for x in ts)[len(ts) // 2] * 1e6, sorted(x[1] for x in ts)[len(ts) // 2] * 1e6))
else:
    ev = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
    for f in glob.glob(sys.argv[2] + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
    ev.sort()
    # calls are separated by the resize chain's first kernel: take the events between the 40th and 41st "first k_resize after a k_filter"
    starts = [i for i, x in enumerate(ev) if "k_resize" in x[2] and (i == 0 or "k_resize" not in ev[i - 1][2]) and not any("k_resize" in y[2] for y in ev[max(0, i - 3):i])]
    i0, i1 = starts[40], starts[41]
    # include the copy that precedes the resize chain
    while i0 > 0 and ev[i0 - 1][2].startswith("COPY") and ev[i0][0] - ev[i0 - 1][1] < 50000: i0 -= 1
    seg = ev[i0:i1]
    t0 = seg[0][0]
    prev_end = t0
    busy = 0
    for s, e_, n in seg:
        print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e_ - s) / 1e3, n))
        busy += e_ - s; prev_end = max(prev_end, e_)
    print("span %.1f us, busy %.1f us, %d events" % ((prev_end - t0) / 1e3, busy / 1e3, len(seg)))
