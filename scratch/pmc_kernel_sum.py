"""Per-dispatch averages of one kernel's counters from a rocprofv3 --kernel-trace --pmc run, split by grid size (batch of 1 vs batch of 16):
usage pmc_kernel_sum.py <dir> <kernel name substring>"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
dur = collections.defaultdict(list)
name = sys.argv[2]
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if name not in r["Kernel_Name"]: continue
        g = r.get("Grid_Size", "?")
        acc[g][r["Counter_Name"]] += float(r["Counter_Value"]); calls[g].add(r["Dispatch_Id"])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if name not in r["Kernel_Name"]: continue
        dur[r.get("Grid_Size_X", r.get("Grid_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# %s: per-dispatch averages by grid size (threads); durations from the kernel trace of the same run (counter collection serialises and slows dispatches)" % name)
for g, d in sorted(acc.items(), key=lambda kv: int(kv[0]) if kv[0].isdigit() else 0):
    n = len(calls[g])
    print("grid %-8s dispatches %5d  " % (g, n) + "  ".join("%s=%.4g" % (c, v / n) for c, v in sorted(d.items())))
for g, us in sorted(dur.items(), key=lambda kv: int(kv[0]) if str(kv[0]).isdigit() else 0):
    us = sorted(us)
    print("grid_x %-8s dispatches %5d  median_us %.1f  min_us %.1f" % (g, len(us), us[len(us) // 2], us[0]))
