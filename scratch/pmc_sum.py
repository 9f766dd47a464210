"""Sum rocprofv3 --pmc counter_collection csv per kernel: usage pmc_sum.py <dir>"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-28:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = len(calls[k])
    print("%-28s n=%5d " % (k, n) + " ".join("%s=%.3g" % (c.replace("SQ_", ""), v / n) for c, v in sorted(d.items())))
