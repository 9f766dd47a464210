"""rocprofv3 --pmc passes (one directory per pass: FETCH_SIZE, WRITE_SIZE, SQ_* ...) -> profiles/rNN_pmc_counters.json
Every counter is reported PER STEP of the bench (a step launches some kernels more than once: 7 x k_resize, one k_fast_cells per LDS
size class): total over the run / number of steps, where the number of steps = launches of k_describe (one per step and lane).
usage: pmc_to_json.py <out.json> P W H NF LANES <pass_dir> [<pass_dir> ...]"""
import csv, glob, json, sys, collections
out_path, bench_args, dirs = sys.argv[1], [int(a) for a in sys.argv[2:7]], sys.argv[7:]
def kname(n):
    return n.split("(")[0].split("<")[0].replace("void ", "").replace("dcs::", "")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
n_steps = {c: len(v) for c, v in launches.get("k_describe", {}).items()}
out = {"command": "rocprofv3 --kernel-trace --pmc <counter(s)> (one pass per counter group, --kernel-trace only) -- python bench.py --steps 5 --warmup 1 --cpu-seconds 0 "
                  "--no-ba --no-bow --no-c3 --no-c5 --no-host-api",
       "bench_args": bench_args,
       "note": "per STEP of the bench = total over the run / launches of k_describe. FETCH_SIZE / WRITE_SIZE are KB as reported by rocprofv3; "
               "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so wide streaming reads are under-reported by up to 2x "
               "(the x2 figure is listed beside the raw one); SQ_* are summed over the chip; SQ_ACTIVE_INST_* are in units of 4 cycles",
       "kernels": {}}
for k in sorted(tot):
    if not k.startswith("k_"): continue
    e = {}
    for c, v in sorted(tot[k].items()):
        ns = max(n_steps.get(c, 0), 1)
        e[c + "_per_step"] = round(v / ns, 1)
        e.setdefault("launches_per_step", round(len(launches[k][c]) / ns, 2))
        if c == "FETCH_SIZE": e["FETCH_SIZE_x2_per_step"] = round(2 * v / ns, 1)
    out["kernels"][k] = e
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out["kernels"].items() if k in ("k_fast_cells", "k_describe", "k_resize")}, indent=1)[:2500])
