// chain_bench.hip -- what bounds ONE 64-field apply per launch at the 5 km headline shape (122 rows x 76 611 columns,
// 82 870 entries, 39.2 MB of X)?  Kernel DURATIONS (hipExtLaunchKernel start/stop events: what rocprofv3 reports), cold
// (14 rotating X buffers, 550 MB) and warm (one buffer: Infinity-Cache resident), of stripped-down variants of the row kernel:
//   EMPTY      the grid alone
//   STREAM     every wave reads its (row, field) piece with arithmetic addresses: no matrix at all
//   RUN        STREAM + the weights: vals staged through LDS (coalesced), X loads issued BEFORE the staging barrier
//   FULL       row pointer -> staged (colind, vals) -> gathered X (the product kernel's chain); RP = where the row bounds come from
// build: hipcc -O3 --offload-arch=gfx950 scratch/chain_bench.hip -o scratch/chain_bench ; run from the repo root
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <vector>

constexpr int NROW = 122, NF = 64;
constexpr long LDX = 76611;
struct RowArg { int beg[NROW + 1]; int c0[NROW]; };

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dppm(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dppm<0xB1, 0xf>(v); v += dppm<0x4E, 0xf>(v); v += dppm<0x141, 0xf>(v); v += dppm<0x140, 0xf>(v);
    v += dppm<0x142, 0xa>(v); v += dppm<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double xload(__amdgpu_buffer_rsrc_t rs, int off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
}

// MODE 0 EMPTY, 1 STREAM, 2 RUN, 3 FULL(rowptr global), 4 FULL(rowptr kernarg), 5 FULL (row bounds + staging, X arithmetic: no gather dependency on LDS)
#define STAMP(i) do { if (ts && lane == 0) ts[((long)blockIdx.x * NW + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
template <int MODE, int NW, int U>
__global__ __launch_bounds__(NW * 64, 8) void k(const RowArg ra, const int *__restrict__ rowptr, const int *__restrict__ colind,
                                               const double *__restrict__ vals, const double *__restrict__ X, double *__restrict__ Y,
                                               unsigned long long *__restrict__ ts = nullptr) {
    if (MODE == 0) return;
    constexpr int T = NW * 64, SEG = 1024, ST = SEG / T;
    __shared__ int s_col[SEG];
    __shared__ double s_val[SEG];
    const int nfc = NF / NW;
    // XCD owns field chunks (nfc multiple of 8) as the product kernel does
    const int xb = blockIdx.x & 7, slot = blockIdx.x >> 3, per = nfc >> 3;
    const int r = slot / per, fc = xb + 8 * (slot - r * per);
    if (r >= NROW) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f = fc * NW + wave;
    STAMP(0);
    int beg, end, c0;
    if (MODE == 3 || MODE == 5) { beg = rowptr[r]; end = rowptr[r + 1]; c0 = ra.c0[r]; }
    else { beg = ra.beg[r]; end = ra.beg[r + 1]; c0 = ra.c0[r]; }
    const int n = end - beg;
    if (ts) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); STAMP(1); }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (long)f * LDX), 0, (int)LDX * 8, 0x00020000);
    double acc = 0.0;
    if (MODE == 1) {
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int k = lane + 64 * u; x[u] = xload(rs, (c0 + (k < n ? k : n - 1)) << 3); }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += x[u];
    } else if (MODE == 2 || MODE == 5) {
        double vv[ST];
#pragma unroll
        for (int i = 0; i < ST; ++i) vv[i] = vals[beg + min((int)threadIdx.x + i * T, n - 1)];
        __builtin_amdgcn_sched_barrier(0);
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int k = lane + 64 * u; x[u] = xload(rs, (c0 + (k < n ? k : n - 1)) << 3); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < ST; ++i) { const int k = threadIdx.x + i * T; if (k < n) s_val[k] = vv[i]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) { const int k = lane + 64 * u; const bool ok = k < n; const double v = s_val[ok ? k : n - 1]; acc = ok ? fma(v, x[u], acc) : acc; }
    } else {
        int cc[ST]; double vv[ST];
#pragma unroll
        for (int i = 0; i < ST; ++i) { const int k = min((int)threadIdx.x + i * T, n - 1); cc[i] = colind[beg + k]; vv[i] = vals[beg + k]; }
#pragma unroll
        for (int i = 0; i < ST; ++i) { const int k = threadIdx.x + i * T; if (k < n) { s_col[k] = cc[i]; s_val[k] = vv[i]; } }
        __syncthreads();
        STAMP(2);
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int k = lane + 64 * u; x[u] = xload(rs, s_col[k < n ? k : n - 1] << 3); }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
        if (ts) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(U - 1) : "memory"); STAMP(4); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(5); }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int k = lane + 64 * u; const bool ok = k < n; const double v = s_val[ok ? k : n - 1]; acc = ok ? fma(v, x[u], acc) : acc; }
    }
    acc = wave_sum(acc);
    STAMP(6);
    if (lane == 0) Y[(long)f * NROW + r] = acc;
    if (ts) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(7); }
}

int main(int argc, char **argv) {
    const long total = NF * LDX; const int nbuf = 14;
    double *x; hipMalloc(&x, sizeof(double) * total * nbuf); hipMemset(x, 0, sizeof(double) * total * nbuf);
    double *y; hipMalloc(&y, sizeof(double) * 65536);
    std::vector<int> rp(NROW + 1, 0), c0(NROW, 0);
    { FILE *f = fopen("scratch/rowlens_g5.txt", "r"); if (!f) return 1; for (int i = 0; i < NROW; ++i) { int v; if (fscanf(f, "%d", &v) != 1) return 1; rp[i + 1] = rp[i] + v; } fclose(f); }
    const long nnz = rp[NROW];
    std::vector<int> col(nnz); std::vector<double> val(nnz, 1.0);
    for (int r = 0; r < NROW; ++r) { c0[r] = (int)((long)rp[r] * (LDX - 900) / nnz); for (int k = rp[r]; k < rp[r + 1]; ++k) col[k] = c0[r] + (k - rp[r]); }
    if (argc > 1) {            // the real matrix: int32 rowptr[NROW+1] then colind[nnz] (scratch/dump_csr.py)
        FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
        std::vector<int> rr(NROW + 1);
        if (fread(rr.data(), 4, NROW + 1, f) != (size_t)NROW + 1 || rr[NROW] != nnz) { fprintf(stderr, "csr file does not match\n"); return 3; }
        if (fread(col.data(), 4, nnz, f) != (size_t)nnz) return 4;
        fclose(f); rp = rr; printf("REAL colind from %s\n", argv[1]);
    }
    int *d_rp, *d_col; double *d_val;
    hipMalloc(&d_rp, 4 * (NROW + 1)); hipMalloc(&d_col, 4 * nnz); hipMalloc(&d_val, 8 * nnz);
    hipMemcpy(d_rp, rp.data(), 4 * (NROW + 1), hipMemcpyHostToDevice); hipMemcpy(d_col, col.data(), 4 * nnz, hipMemcpyHostToDevice);
    hipMemcpy(d_val, val.data(), 8 * nnz, hipMemcpyHostToDevice);
    RowArg ra; for (int i = 0; i <= NROW; ++i) ra.beg[i] = rp[i]; for (int i = 0; i < NROW; ++i) ra.c0[i] = c0[i];
    const int N = 200;
    const double gap_us = getenv("CHAIN_GAP_US") ? atof(getenv("CHAIN_GAP_US")) : 0.0;      // host delay between launches (a slow submitter)
    std::vector<hipEvent_t> e0(N), e1(N);
    for (int i = 0; i < N; ++i) { hipEventCreate(&e0[i]); hipEventCreate(&e1[i]); }
    auto run = [&](auto launch, const char *name) {
        for (int cold = 1; cold >= 0; --cold) {
            for (int i = 0; i < 20; ++i) launch(i, cold, nullptr, nullptr);
            hipDeviceSynchronize();
            for (int i = 0; i < N; ++i) {
                launch(i, cold, e0[i], e1[i]);
                if (gap_us > 0) { auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < gap_us) {} }
            }
            hipDeviceSynchronize();
            std::vector<float> us(N);
            for (int i = 0; i < N; ++i) { float ms; hipEventElapsedTime(&ms, e0[i], e1[i]); us[i] = ms * 1e3f; }
            std::sort(us.begin(), us.end());
            printf("%-34s %s  med %6.2f  min %6.2f  p90 %6.2f us\n", name, cold ? "cold" : "warm", us[N / 2], us[0], us[N * 9 / 10]);
        }
    };
#define RUN(MODE, NW, U, NAME) run([&](int i, int cold, hipEvent_t a, hipEvent_t b) { \
        hipExtLaunchKernelGGL((k<MODE, NW, U>), dim3(NROW * (NF / NW)), dim3(NW * 64), 0, 0, a, b, 0, ra, d_rp, d_col, d_val, x + (long)(cold ? i % nbuf : 0) * total, y, (unsigned long long *)nullptr); }, NAME " NW=" #NW " U=" #U)
    if (getenv("CHAIN_TIMELINE")) {       // in-kernel timeline of FULL (global rowptr), NW = 8, U = 14: s_memrealtime stamps (100 MHz) per wave
        constexpr int NW = 8; const int nb = NROW * (NF / NW); const long nwave = (long)nb * NW;
        unsigned long long *ts; hipMalloc(&ts, 8 * 8 * nwave);
        std::vector<unsigned long long> h(8 * nwave);
        const char *names[8] = {"start", "rowptr", "staged+barrier", "X issued", "first X", "last X", "reduced", "stored"};
        for (int cold = 1; cold >= 0; --cold) {
            std::vector<std::vector<double>> rel(8);
            for (int rep = 0; rep < 40; ++rep) {
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<3, NW, 14>), dim3(nb), dim3(NW * 64), 0, 0, ra, d_rp, d_col, d_val, x + (long)(cold ? (4 * rep + i) % nbuf : 0) * total, y, (unsigned long long *)nullptr);
                hipMemset(ts, 0, 8 * 8 * nwave);
                hipDeviceSynchronize();
                hipLaunchKernelGGL((k<3, NW, 14>), dim3(nb), dim3(NW * 64), 0, 0, ra, d_rp, d_col, d_val, x + (long)(cold ? (4 * rep + 3) % nbuf : 0) * total, y, ts);
                hipMemcpy(h.data(), ts, 8 * 8 * nwave, hipMemcpyDeviceToHost);
                unsigned long long t0 = ~0ull;
                for (long w = 0; w < nwave; ++w) if (h[8 * w]) t0 = std::min(t0, h[8 * w]);
                for (long w = 0; w < nwave; ++w) if (h[8 * w]) for (int q = 0; q < 8; ++q) rel[q].push_back((double)(h[8 * w + q] - t0) * 0.01);
            }
            printf("timeline %s (us after the first wave's start; all waves of 40 launches)\n", cold ? "cold" : "warm");
            for (int q = 0; q < 8; ++q) { auto &v = rel[q]; std::sort(v.begin(), v.end()); size_t m = v.size();
                printf("  %-16s first %5.2f  p10 %5.2f  median %5.2f  p90 %5.2f  last %5.2f\n", names[q], v[0], v[m / 10], v[m / 2], v[m * 9 / 10], v[m - 1]); }
        }
        return 0;
    }
    RUN(0, 4, 8, "EMPTY"); RUN(0, 8, 8, "EMPTY");
    for (int nb : {64, 256, 512, 1024, 2048, 4096}) {
        char nm[64]; snprintf(nm, sizeof nm, "EMPTY grid %d x 256", nb);
        run([&](int i, int cold, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k<0, 4, 8>), dim3(nb), dim3(256), 0, 0, a, b, 0, ra, d_rp, d_col, d_val, x, y, (unsigned long long *)nullptr); }, nm);
    }
    RUN(1, 4, 14, "STREAM (no matrix)"); RUN(1, 8, 14, "STREAM (no matrix)");
    RUN(2, 4, 14, "RUN (X before staging)"); RUN(2, 8, 14, "RUN (X before staging)");
    RUN(5, 8, 14, "RUN + global rowptr");
    RUN(4, 8, 14, "FULL kernarg rowptr"); RUN(3, 8, 14, "FULL global rowptr"); RUN(3, 4, 14, "FULL global rowptr");
    return 0;
}
