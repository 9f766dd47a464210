"""aggregate k_resize / k_fast_cells durations by grid size from a rocprofv3 kernel trace (csv) under the given directory"""
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not fs: sys.exit("no kernel_trace.csv under " + sys.argv[1])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"]
    if "k_resize" in n or "k_pyramid" in n or "k_fast_cells" in n or "k_describe" in n or "k_knn2" in n:
        key = (n.split("(")[0][-28:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v = v[len(v) // 3:]          # skip warm-up launches
    print("%-30s grid %8s x %-6s n=%4d  avg %8.1f us  min %8.1f" % (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v)))
