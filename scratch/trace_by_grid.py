"""per (kernel, grid) durations from a rocprofv3 kernel trace: usage trace_by_grid.py <dir> [name filter ...]"""
import csv, glob, sys, collections
d = collections.defaultdict(list)
flt = sys.argv[2:]
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if flt and not any(x in n for x in flt): continue
        short = n.split("(")[0].replace("dcs::", "").replace("void ", "")[:48]
        key = (short, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k in sorted(d, key=lambda k: -sum(d[k])):
    v = sorted(d[k]); med = v[len(v) // 2]
    print("%-48s grid %7d x %5d x %3d  n %4d  median %8.1f us  min %8.1f" % (k[0], k[1], k[2], k[3], len(v), med, v[0]))
