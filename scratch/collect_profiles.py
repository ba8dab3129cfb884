"""Copy the judged summaries of scratch/prof_r02.sh from gpurun_out/prof_r02 into profiles/ (round 2)."""
import glob, json, os, shutil
src = "gpurun_out/prof_r02"
os.makedirs("profiles", exist_ok=True)
for tag in ("kt_default", "kt_driver20"):
    f = glob.glob(os.path.join(src, tag, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], "profiles/r02_bench_g5_AvI_64f_%s_kernel_stats.csv" % tag[3:])
    log = [l for l in open(os.path.join(src, tag + ".log")).read().splitlines() if l.startswith("{")]
    if log:
        open("profiles/r02_bench_g5_AvI_64f_%s.json.log" % tag[3:], "w").write(log[-1] + "\n")
shutil.copy(os.path.join(src, "summary.txt"), "profiles/r02_bench_g5_AvI_64f_summary.txt")
s = json.load(open(os.path.join(src, "summary.json")))
pm = s["pmc"]
k = [x for x in pm if "spmm_rowblock" in x][0]
cal = [x for x in pm if "CUDAFunctorOnSelf_add" in x or "elementwise_kernel<4, at::native::CUDAFunc" in x][0]
fetch, write = pm[k]["FETCH_SIZE"]["mean_KiB"] * 1024, pm[k]["WRITE_SIZE"]["mean_KiB"] * 1024
cal_fetch = pm[cal]["FETCH_SIZE"]["mean_KiB"] * 1024
nf, ncol, nrow, nnz, depth = 64, 76611, 122, 82870, 32
alg = 12 * nnz + 4 * (nrow + 1) + depth * 8 * nf * (ncol + nrow)
out = {
    "spmm_rowblock_g5_AvI_64f_d32": {
        "kernel": k, "launches_sampled": pm[k]["FETCH_SIZE"]["launches"], "applies_per_launch": depth,
        "FETCH_SIZE_bytes_raw": fetch, "FETCH_SIZE_bytes_corrected_x2": 2 * fetch, "WRITE_SIZE_bytes": write,
        "traffic_bytes": 2 * fetch + write, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (2 * fetch + write) / alg,
        "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --steps 64 --warmup 32 --no-cpu-baseline",
    },
    "calibration": {
        "kernel": cal, "note": "torch's x0 + c elementwise kernel of the same run reads one 64 x 76611 f64 batch (39 224 832 B) and writes one",
        "FETCH_SIZE_bytes_raw": cal_fetch, "known_read_bytes": 8 * nf * ncol, "raw_over_known": cal_fetch / (8.0 * nf * ncol),
        "WRITE_SIZE_bytes": pm[cal]["WRITE_SIZE"]["mean_KiB"] * 1024,
    },
    "kernel_trace": {t: s[t] for t in ("kt_default", "kt_driver20") if t in s},
}
json.dump(out, open("profiles/r02_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["spmm_rowblock_g5_AvI_64f_d32"], indent=1))
