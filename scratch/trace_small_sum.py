import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(f))]
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))) for r in csv.DictReader(open(f))]
rows.sort()
# last call = the last run of events ending with a D2H copy
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
tail = rows[-n:]
t0 = tail[0][0]
prev = None
for s, e, k in tail:
    print("%8.1f us  +%6.1f dur  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3, k))
    prev = e
print("span %.1f us" % ((tail[-1][1] - t0) / 1e3))
