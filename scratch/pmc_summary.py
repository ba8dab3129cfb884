"""Summarise the rocprofv3 outputs of scratch/prof_r02.sh: kernel stats (avg duration of the SpMM
kernel per launch and per 64-field apply) and the PMC passes (FETCH_SIZE doubled per the gfx950
correction of MI355X_MICROARCH.md, calibrated in the same run on torch's elementwise add, which reads
and writes one 39.2 MB field batch)."""
import csv, glob, json, os, re, sys
out = sys.argv[1]
def short(n):
    return re.sub(r"\(.*", "", n)
res = {}
for tag in ("kt_default", "kt_driver20"):
    fs = glob.glob(os.path.join(out, tag, "**", "*kernel_stats.csv"), recursive=True)
    if not fs: continue
    rows = list(csv.DictReader(open(fs[0])))
    log = open(os.path.join(out, tag + ".log")).read()
    line = [l for l in log.splitlines() if l.startswith("{")]
    bench = json.loads(line[-1]) if line else {}
    print("== %s" % tag)
    for r in rows[:6]:
        print("  %-70s calls %5s avg %10.1f us" % (short(r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
    # launches of the timed region only: trace file, last `launches` calls of the spmm kernel
    tr = glob.glob(os.path.join(out, tag, "**", "*kernel_trace.csv"), recursive=True)
    if tr and bench:
        ks = [r for r in csv.DictReader(open(tr[0])) if "spmm_rowblock" in r["Kernel_Name"]]
        n = bench["roofline"]["launches"]
        last = ks[-n:]
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
        steps = bench["steps"]
        per_apply = sum(durs) / steps
        B = bench["roofline"]["algorithmic_bytes_per_launch"] * n
        print("  timed region: %d launches, mean %.1f us per launch, %.3f us per 64-field apply -> %.1f GB/s = %.3f of 8 TB/s (in-bench HIP events: %.3f us, frac %.3f)"
              % (n, sum(durs) / n, per_apply, B / sum(durs) / 1e3, B / sum(durs) / 1e3 / 8000, bench["roofline"]["kernel_us"], bench["roofline"]["frac"]))
        res[tag] = {"launches": n, "mean_launch_us": sum(durs) / n, "us_per_apply": per_apply, "frac_of_8TBps": B / sum(durs) / 1e3 / 8000,
                    "bench_kernel_us": bench["roofline"]["kernel_us"], "bench_frac": bench["roofline"]["frac"]}
pm = {}
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    fs = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    if not fs: continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != ctr: continue
        agg.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "spmm_rowblock" in k or "elementwise" in k:
            pm.setdefault(k, {})[ctr] = {"launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v), "values_KiB": v if len(v) <= 8 else None}
print("== PMC (KiB per launch)")
for k, d in pm.items():
    for c, v in d.items():
        print("  %-70s %-10s n=%3d mean %12.1f min %12.1f max %12.1f" % (k[:70], c, v["launches"], v["mean_KiB"], v["min_KiB"], v["max_KiB"]))
res["pmc"] = pm
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
