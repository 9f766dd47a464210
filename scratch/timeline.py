"""GPU timeline occupancy from a rocprofv3 kernel trace: usage timeline.py <dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-24:]) for r in rows)
# steady state: from the 5th FAST launch to the last filter kernel
fast = [e for e in ev if "k_fast_cells" in e[2]]
filt = [e for e in ev if "k_filter_pairs" in e[2]]
t_lo, t_hi = fast[4][0], filt[-1][1]
ev = [e for e in ev if e[0] >= t_lo and e[1] <= t_hi]
print("steps in window:", len([e for e in ev if "k_fast_cells" in e[2]]))
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
gaps = []
for s, e, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.1f us, union busy %.1f us (%.1f %%), sum of kernel durations %.1f us" % (span / 1e3, busy / 1e3, 100 * busy / span, sum(e - s for s, e, _ in ev) / 1e3))
import collections
g = collections.defaultdict(lambda: [0, 0])
for d, n in gaps: g[n][0] += d; g[n][1] += 1
for n, (d, c) in sorted(g.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  idle before %-26s total %8.1f us in %4d gaps (avg %.1f us)" % (n, d / 1e3, c, d / 1e3 / c))
