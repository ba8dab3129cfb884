"""Summarise scratch/prof_evi.sh: per-kernel average durations of the three EvI kernels (16 applies per launch) and the
HBM traffic of the column sweep (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, calibrated on
torch's elementwise add of the same run, which reads one 64-field batch; WRITE_SIZE as reported)."""
import csv, glob, json, os, re, sys
out = sys.argv[1]
short = lambda n: re.sub(r"\(.*", "", n)
res = {}
for mode in ("rowblock", "rowdual", "colsweep"):
    fs = glob.glob(os.path.join(out, "kt_" + mode, "**", "*kernel_stats.csv"), recursive=True)
    if not fs: continue
    print("== %s (16 applies per launch)" % mode)
    tot = 0.0
    for r in list(csv.DictReader(open(fs[0])))[:8]:
        n = short(r["Name"])
        if any(k in n for k in ("spmm_", "sweep_", "dual_combine")):
            print("  %-60s calls %4s avg %10.1f us  -> %8.2f us per apply" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["AverageNs"]) / 16e3))
            tot += float(r["AverageNs"]) / 16e3
    res[mode] = {"us_per_apply_sum_of_kernels": tot}
    print("  sum of the apply kernels: %.2f us per 64-field apply" % tot)
log = open(os.path.join(out, "pmc_fetch.log")).read()
m = re.search(r"algorithmic bytes per apply (\d+), calibration read bytes (\d+)", log)
alg, calb = int(m.group(1)), int(m.group(2))
pm = {}
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    fs = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == ctr: agg.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "sweep" in k or "elementwise" in k:
            pm.setdefault(k, {})[ctr] = {"n": len(v), "mean_bytes": 1024 * sum(v) / len(v)}
print("== PMC of the column sweep, one apply per launch (bytes per launch)")
for k, d in pm.items():
    print("  %-70s %s" % (k[:70], "  ".join("%s n=%d mean %.4e" % (c, v["n"], v["mean_bytes"]) for c, v in d.items())))
sw = [k for k in pm if "spmm_sweep" in k][0]
cb = [k for k in pm if "sweep_combine" in k][0]
cal = [k for k in pm if "elementwise" in k and "FETCH_SIZE" in pm[k]]
calf = max(pm[k]["FETCH_SIZE"]["mean_bytes"] for k in cal)
fetch = 2 * (pm[sw]["FETCH_SIZE"]["mean_bytes"] + pm[cb]["FETCH_SIZE"]["mean_bytes"])
write = pm[sw]["WRITE_SIZE"]["mean_bytes"] + pm[cb]["WRITE_SIZE"]["mean_bytes"]
print("calibration: elementwise add raw FETCH %.4e for %d bytes read -> raw/known %.4f (x2 correction applies)" % (calf, calb, calf / calb))
print("column sweep + combine: fetched (x2) %.4e + written %.4e = %.4e bytes per apply; algorithmic %d -> %.3f x" % (fetch, write, fetch + write, alg, (fetch + write) / alg))
res["pmc"] = {"kernels": pm, "algorithmic_bytes_per_apply": alg, "traffic_bytes_per_apply": fetch + write, "traffic_over_algorithmic": (fetch + write) / alg,
              "calibration_raw_over_known": calf / calb}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
