"""File the round-5 measurement set (scratch/collect_r05.sh -> gpurun_out/r05/) under profiles/r05_*: summaries, bench JSON lines,
rocprofv3 kernel-stats csv of the three headline runs (one run per mode: summary and csv of the SAME run), PMC summaries, and
profiles/r05_pmc_traffic.json (what bench.py's roofline.traffic reads)."""
import glob, json, os, shutil
src, dst = "gpurun_out/r05", "profiles"
def put(path, name):
    if os.path.exists(path) and os.path.getsize(path) > 0:
        body = [l for l in open(path, errors="replace").read().splitlines() if not l.startswith("/opt/amdgpu")]
        open(os.path.join(dst, "r05_" + name), "w").write("\n".join(body) + "\n")
        print("profiles/r05_" + name)
for f in sorted(glob.glob(os.path.join(src, "*.txt")) + glob.glob(os.path.join(src, "*.json.log")) + glob.glob(os.path.join(src, "*.bench.json"))):
    put(f, os.path.basename(f))
for tag in ("kt_default", "kt_driver20", "kt_depth1", "kt_depth1_plain"):
    for f in glob.glob(os.path.join(src, tag, "**", "*kernel_stats.csv"), recursive=True):
        put(f, "bench_g5_AvI_64f_%s_kernel_stats.csv" % tag[3:])
    if os.path.exists(os.path.join(src, tag + ".summary.json")):
        shutil.copy(os.path.join(src, tag + ".summary.json"), os.path.join(dst, "r05_%s.summary.json" % tag)); print("profiles/r05_%s.summary.json" % tag)
# PMC traffic per launch of the headline kernels (FETCH_SIZE doubled per the gfx950 correction; calibration in the same passes:
# torch's elementwise add over one 39.2 MB field batch reads 39.25 MB and writes 39.23 MB by these counters)
out = {}
for tag, key, per in (("pmc_default", "spmm_rowblock_g5_AvI_64f_d32", 32), ("pmc_depth1", "spmm_rowblock_g5_AvI_64f_d1", 1)):
    try:
        fe = json.load(open(os.path.join(src, tag + "_FETCH_SIZE.json")))["pmc"]
        wr = json.load(open(os.path.join(src, tag + "_WRITE_SIZE.json")))["pmc"]
    except (OSError, KeyError):
        continue
    k = [n for n in fe if n.startswith("spmm_")][0]
    cal = [n for n in fe if "CUDAFunctorOnSelf_add" in n]
    out[key] = {"kernel": k, "applies_per_launch": per, "fetch_MB": fe[k]["FETCH_SIZE"]["mean_MB"], "write_MB": wr[k]["WRITE_SIZE"]["mean_MB"],
                "traffic_bytes": (fe[k]["FETCH_SIZE"]["mean_MB"] + wr[k]["WRITE_SIZE"]["mean_MB"]) * 1e6, "dispatches": fe[k]["FETCH_SIZE"]["dispatches"],
                "calibration_elementwise_add_39.2MB": {"fetch_MB": fe[cal[0]]["FETCH_SIZE"]["mean_MB"], "write_MB": wr[cal[0]]["WRITE_SIZE"]["mean_MB"]} if cal else None}
    B = 40282228
    csr = 12 * 82870 + 4 * 123
    alg = per * (B - csr) + csr
    out[key]["algorithmic_bytes_per_launch"] = alg
    out[key]["traffic_over_algorithmic"] = out[key]["traffic_bytes"] / alg
if out:
    json.dump(out, open(os.path.join(dst, "r05_pmc_traffic.json"), "w"), indent=1); print("profiles/r05_pmc_traffic.json", {k: round(v["traffic_over_algorithmic"], 3) for k, v in out.items()})
