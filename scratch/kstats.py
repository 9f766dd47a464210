"""print Name / Calls / average / total / share of a rocprofv3 *_kernel_stats.csv: the file itself, or the first one found under a directory"""
import csv, glob, os, sys
src = sys.argv[1]
if os.path.isdir(src):
    fs = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)
    if not fs: sys.exit("no kernel_stats.csv under " + src)
    src = fs[0]
rows = list(csv.DictReader(open(src)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print("%-44s calls %6s avg %9.1f us  total %9.1f us  %5.1f%%" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, float(r["Percentage"])))
