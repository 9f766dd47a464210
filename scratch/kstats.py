"""print Name / Calls / AverageNs of a rocprofv3 *_kernel_stats.csv found under the given directory"""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs: sys.exit("no kernel_stats.csv under " + sys.argv[1])
for r in list(csv.DictReader(open(fs[0])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("%-60s %6s %10.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
