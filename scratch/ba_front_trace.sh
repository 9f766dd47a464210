#!/bin/bash
# durations of every kernel of ONE C4 solve in launch order (the last solve of the run): usage ba_front_trace.sh [env assignments]
cd /tmp; export TMPDIR=/tmp
env "$@" timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ftrace -- python $GRAFT_REPO_ROOT/scratch/time_ba_batch.py 1 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ftrace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "dcs::" in r["Kernel_Name"]]
# last solve: from the last k_ctl_init
last = max(i for i, r in enumerate(rows) if "k_ctl_init" in r["Kernel_Name"])
prev_end = None
for r in rows[last:last + 80]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dcs::", "")
    print("%-28s %7.1f us  gap %6.1f" % (name[:28], (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
PY
rm -rf gpurun_out/ftrace
