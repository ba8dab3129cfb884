// write floor with the shortrow kernel's store pattern: workgroup (row block, field chunk) stores
// 256 rows x FPER fields of a field-major Y (planes ldy apart), nothing read.  Variants: chunk
// fastest vs row-block fastest, with/without a dependent dummy load chain in front.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int FPER, bool CHUNK_FAST, int CHAIN>
__global__ __launch_bounds__(256) void k(double* __restrict__ y, long ldy, int nrow, int nf, const int* __restrict__ dummy) {
    const int nfy = nf / FPER;
    const int nrb = (nrow + 255) / 256;
    int rb, fy;
    if (CHUNK_FAST) { rb = blockIdx.x / nfy; fy = blockIdx.x % nfy; } else { fy = blockIdx.x / nrb; rb = blockIdx.x % nrb; }
    const int r = rb * 256 + threadIdx.x;
    if (r >= nrow) return;
    double v = 1.0;
    if (CHAIN) { int a = dummy[r]; if (CHAIN > 1) a = dummy[(a + r) % nrow]; if (CHAIN > 2) a = dummy[(a + r + 1) % nrow]; v = (double)a; }
#pragma unroll
    for (int j = 0; j < FPER; ++j) __builtin_nontemporal_store(v, &y[(long)(fy * FPER + j) * ldy + r]);
}
template <int FPER, bool CF, int CHAIN>
void run(const char* name, double* y, long ldy, int nrow, int nf, int* dummy) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = ((nrow + 255) / 256) * (nf / FPER);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<FPER, CF, CHAIN>), dim3(grid), dim3(256), 0, 0, y, ldy, nrow, nf, dummy);
    hipDeviceSynchronize(); hipEventRecord(e0);
    const int steps = 40;
    for (int i = 0; i < steps; ++i) hipLaunchKernelGGL((k<FPER, CF, CHAIN>), dim3(grid), dim3(256), 0, 0, y + (i & 1) * ldy * nf, ldy, nrow, nf, dummy);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / steps, B = 8.0 * nrow * nf;
    printf("%-28s fper=%2d %9.2f us  %7.0f GB/s\n", name, FPER, us, B / us / 1e3);
}
int main() {
    const int nrow = 1907329, nf = 64; const long ldy = nrow;
    double* y; hipMalloc(&y, (size_t)ldy * nf * 8 * 2);
    int* dummy; hipMalloc(&dummy, (size_t)nrow * 4); hipMemset(dummy, 0, (size_t)nrow * 4);
    run<4, true, 0>("chunk-fast no-load", y, ldy, nrow, nf, dummy);
    run<8, true, 0>("chunk-fast no-load", y, ldy, nrow, nf, dummy);
    run<64, true, 0>("chunk-fast no-load", y, ldy, nrow, nf, dummy);
    run<4, false, 0>("rowblock-fast no-load", y, ldy, nrow, nf, dummy);
    run<8, false, 0>("rowblock-fast no-load", y, ldy, nrow, nf, dummy);
    run<4, true, 1>("chunk-fast chain1", y, ldy, nrow, nf, dummy);
    run<4, true, 3>("chunk-fast chain3", y, ldy, nrow, nf, dummy);
    run<8, true, 3>("chunk-fast chain3", y, ldy, nrow, nf, dummy);
    run<4, false, 3>("rowblock-fast chain3", y, ldy, nrow, nf, dummy);
    run<64, true, 3>("chunk-fast chain3", y, ldy, nrow, nf, dummy);
    return 0;
}
