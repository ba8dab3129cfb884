// Microbenchmark 2: same 39 MB, but read the way the SpMM reads it: block = (row r, field chunk fc),
// wave w reads FPW fields x [c0_r, c0_r + len) (pieces of ~6 KB at stride ldx*8 = 613 KB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int FPW, int UNROLL, bool XCD>
__global__ __launch_bounds__(256) void piece_sum(const double* __restrict__ X, long ldx, int nrow, int nfc, int len_, double* __restrict__ out, const int* __restrict__ rowptr) {
    int b = blockIdx.x, nb = nrow * nfc;
    if (XCD) { int q = nb >> 3, rem = nb & 7, x = b & 7, slot = b >> 3; b = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + slot; }
    const int r = b / nfc, fc = b - r * nfc;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f0 = fc * 4 * FPW + wave * FPW;
    const int len = rowptr ? rowptr[r + 1] - rowptr[r] : len_;
    const long c0 = rowptr ? (long)rowptr[r] * 76611 / 82870 : (long)r * len_;
    double acc[FPW];
    for (int j = 0; j < FPW; ++j) acc[j] = 0;
    for (int k0 = lane; k0 - lane < len; k0 += 64 * UNROLL) {
        double x[FPW][UNROLL];
#pragma unroll
        for (int j = 0; j < FPW; ++j)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { int k = k0 + u * 64; x[j][u] = X[(long)(f0 + j) * ldx + c0 + (k < len ? k : len - 1)]; }
#pragma unroll
        for (int j = 0; j < FPW; ++j)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc[j] += x[j][u];
    }
    for (int j = 0; j < FPW; ++j) { double a = acc[j]; for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64); if (lane == 0) out[(long)(f0 + j) * nrow + r] = a; }
}
int main() {
    const int nrow = 122, len = 628, nf = 64; const long ldx = 76611; const long total = nf * ldx; const int nbuf = 14;
    double* x; hipMalloc(&x, sizeof(double) * total * nbuf); hipMemset(x, 0, sizeof(double) * total * nbuf);
    double* out; hipMalloc(&out, sizeof(double) * 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char* name) {
        for (int i = 0; i < 30; ++i) launch(i);
        hipDeviceSynchronize(); hipEventRecord(e0);
        const int steps = 300; for (int i = 0; i < steps; ++i) launch(i);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1e3 / steps; printf("%-44s %7.2f us  %7.0f GB/s\n", name, us, (double)nrow * len * nf * 8 / us / 1e3);
    };
#define RUN(FPW, U, XCD) time([&](int i) { hipLaunchKernelGGL((piece_sum<FPW, U, XCD>), dim3(nrow * (nf / (4 * FPW))), dim3(256), 0, 0, x + (long)(i % nbuf) * total, ldx, nrow, nf / (4 * FPW), len, out, RP); }, RP ? "REAL lens FPW=" #FPW " U=" #U " xcd=" #XCD : "uniform   FPW=" #FPW " U=" #U " xcd=" #XCD)
    std::vector<int> rp(nrow + 1, 0);
    { FILE* f = fopen("scratch/rowlens_g5.txt", "r"); for (int i = 0; i < nrow; ++i) { int v; if (fscanf(f, "%d", &v) != 1) return 1; rp[i + 1] = rp[i] + v; } fclose(f); }
    int* d_rp; hipMalloc(&d_rp, sizeof(int) * (nrow + 1)); hipMemcpy(d_rp, rp.data(), sizeof(int) * (nrow + 1), hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
    const int* RP = pass ? d_rp : nullptr;
    RUN(4, 4, true); RUN(4, 4, false); RUN(2, 8, true); RUN(2, 8, false); RUN(2, 4, true); RUN(1, 8, true); RUN(1, 8, false); RUN(4, 2, true); RUN(1, 4, true); RUN(1, 10, true); RUN(2, 10, true); RUN(1, 14, true); RUN(2, 14, true);
    }
    return 0;
}
