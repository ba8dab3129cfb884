"""Summarise rocprofv3 csv output found under a directory: kernel trace -> per (kernel, grid) count / median / mean / min / max us;
counter collection (FETCH_SIZE, WRITE_SIZE passes) -> mean per dispatch in MB (FETCH_SIZE doubled: on gfx950 it reports half the
bytes of wide coalesced reads, MI355X_MICROARCH.md "HBM"; rocprofv3 prints these two in KiB).  usage: rocsum.py DIR [name filter] [--json out.json]"""
import csv, glob, json, os, re, statistics as st, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "").replace("ibh::", "")
res = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    by = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if flt and flt not in k: continue
        grid = "%sx%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", "1"), r.get("Grid_Size_Z", "1"))
        by.setdefault((k, grid), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# kernel trace %s" % os.path.relpath(f, d))
    print("%-78s %-16s %6s %9s %9s %9s %9s" % ("kernel", "grid (threads)", "calls", "median", "mean", "min", "max"))
    for (k, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print("%-78s %-16s %6d %9.2f %9.2f %9.2f %9.2f" % (k[:78], g, len(v), st.median(v), sum(v) / len(v), min(v), max(v)))
        res.setdefault("trace", {})["%s|%s" % (k, g)] = {"calls": len(v), "median_us": st.median(v), "mean_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    by = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if flt and flt not in k and "elementwise" not in k: continue
        by.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    print("# counters %s" % os.path.relpath(f, d))
    for (k, c), v in sorted(by.items()):
        mb = sum(v) / len(v) * 1024 / 1e6 * (2.0 if c == "FETCH_SIZE" else 1.0)
        print("%-78s %-11s dispatches %5d  mean %10.3f MB per dispatch%s" % (k[:78], c, len(v), mb, "  (raw x 2)" if c == "FETCH_SIZE" else ""))
        res.setdefault("pmc", {}).setdefault(k, {})[c] = {"dispatches": len(v), "mean_MB": mb}
if jout:
    json.dump(res, open(jout, "w"), indent=1)
