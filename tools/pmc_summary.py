"""Summarise rocprofv3 --pmc passes (one directory per pass) into per-kernel means.  usage: pmc_summary.py <dir> [out.json]
PMC_LAST=n: means over the LAST n dispatches of each kernel only (a hipGraph run's replays: the warm-up and capture passes in
front of them are eager launches with a different zero-fill / conversion pattern)."""
import csv, glob, json, collections, os, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    if not any(s in k for s in ("decoder_bwd", "render_fwd", "grid_bwd", "convert_kernel", "step_kernel", "composite", "sample_kernel", "sample_prepare", "tracker_")):
        continue
    last = int(os.environ.get("PMC_LAST", "0"))
    if last > 0:
        cs = {c: v[-last:] for c, v in cs.items()}
    d = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
    d["dispatches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:      # KB; gfx950: FETCH_SIZE counts half of wide coalesced fetches
        d["bytes_per_launch"] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
    out[k] = d
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
