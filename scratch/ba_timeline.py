"""Timeline of the last BA call in a rocprofv3 kernel trace: usage ba_timeline.py <dir> [n_kernels]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "dcs::k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %6.1f  dur %7.1f  %-22s grid %sx%sx%s wg %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0].replace("dcs::", "").replace("void ", "")[:22],
          r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"]))
    prev_end = e
