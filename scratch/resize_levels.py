"""per-level durations of k_resize from a rocprofv3 kernel trace: usage resize_levels.py <dir>"""
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_resize" in r["Kernel_Name"]:
            d[int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g in sorted(d, reverse=True):
    v = sorted(d[g]); print("grid %9d threads: n %4d median %7.1f us  min %7.1f" % (g, len(v), v[len(v) // 2], v[0]))
print("sum of medians %.1f us" % sum(sorted(v)[len(v) // 2] for v in d.values()))
