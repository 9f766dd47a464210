"""Sum rocprofv3 --pmc counter_collection csv per kernel (+ average dispatch duration from the kernel trace of the same run):
usage pmc_sum.py <dir>"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-28:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][-28:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = len(calls[k])
    us = sorted(dur.get(k, [0.0]))
    print("%-28s n=%5d med_us=%.1f " % (k, n, us[len(us) // 2]) + " ".join("%s=%.3g" % (c.replace("SQ_", ""), v / n) for c, v in sorted(d.items())))
