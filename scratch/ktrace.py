"""per-kernel duration list from rocprofv3 kernel_trace csv: usage ktrace.py <dir> <substr>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    g = [r["Grid_Size_X"] + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"] for r in rows]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    print(" ".join("%.1f(%s)" % (a, b) for a, b in list(zip(d, g))[-n:]))
