"""FETCH_SIZE / WRITE_SIZE passes of rocprofv3 -> profiles/r01_pmc_hbm_traffic.json
usage: pmc_to_json.py <fetch_dir> <write_dir> <out.json> P W H NF LANES"""
import csv, glob, json, sys, collections
def collect(d, counter):
    acc = collections.defaultdict(float); calls = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("dcs::", "")
            acc[k] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    return {k: (acc[k] / len(calls[k]), len(calls[k])) for k in acc}
fe, wr = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) -- python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-ba",
       "bench_args": [int(a) for a in sys.argv[4:9]],
       "note": "counter units are KB as reported by rocprofv3; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, "
               "so wide streaming reads are under-reported by up to 2x (x2 column); uncalibrated for narrow / scattered access",
       "kernels": {}}
for k in sorted(fe):
    if not k.startswith("k_"): continue
    out["kernels"][k] = {"FETCH_SIZE_KB_avg_per_launch": round(fe[k][0], 1), "FETCH_SIZE_x2_KB": round(2 * fe[k][0], 1),
                         "WRITE_SIZE_KB_avg_per_launch": round(wr.get(k, (0, 0))[0], 1), "launches": fe[k][1]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:1500])
