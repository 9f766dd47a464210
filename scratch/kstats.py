import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print("%-30s calls %6s avg %9.1f us  total %9.1f us  %5.1f%%" % (r["Name"].split("(")[0][-30:], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3, float(r["Percentage"])))
