#!/bin/bash
# GPU box: FETCH_SIZE reported for kernels that read a known number of bytes (scratch/probe/fetch_calib.hip) -> gpurun_out/<tag>/fetch_calibration.json
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/fcal; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fcal -- $R/scratch/probe/fetch_calib > $O/fetch_calib.log 2>&1
python - "$O" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
true = {"unsigned char": (1 << 30) // 4, "unsigned int>": 1 << 30, "unsigned int, 2": 1 << 30, "u3": (1 << 30) // 12 * 12, "unsigned int, 4": 1 << 30, "k_read_roi": (1 << 20) * 60 * 4 * 8}
rows = collections.defaultdict(list)
for f in glob.glob("/tmp/fcal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in rows.items():
    key = "k_read_roi" if "k_read_roi" in k else next((t for t in true if t != "k_read_roi" and ("<" + t) in k.replace("k_read<HIP_vector_type<", "<").replace("k_read<", "<")), None)
    if key is None: print("unmatched kernel:", k); continue
    label = {"unsigned char": "1 B per lane", "unsigned int>": "4 B per lane", "unsigned int, 2": "8 B per lane", "u3": "12 B per lane", "unsigned int, 4": "16 B per lane", "k_read_roi": "8 B per lane, 48-byte row pieces (ROI shape)"}[key]
    res[label] = {"true_bytes": true[key], "FETCH_SIZE_KB_per_launch": [round(x, 1) for x in v], "reported_over_true": [round(x * 1024 / true[key], 4) for x in v]}
json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- scratch/probe/fetch_calib (1 GiB buffer, each kernel reads it once, two repetitions)",
           "note": "reported_over_true = FETCH_SIZE (KB x 1024) / bytes the kernel reads; its inverse is the correction for a kernel whose loads have that width",
           "widths": res}, open(out + "/fetch_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
