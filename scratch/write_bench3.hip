// store-pattern floor, continued: rows per thread (bigger contiguous piece per plane and workgroup),
// leading-dimension padding, plain vs nontemporal, 16-byte stores (2 rows per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int FPER, int RPT, bool NT, bool V2>
__global__ __launch_bounds__(256) void k(double* __restrict__ y, long ldy, int nrow, int nf) {
    const int nfy = nf / FPER;
    const int rb = blockIdx.x / nfy, fy = blockIdx.x % nfy;
    if (V2) {
        const long r = ((long)rb * RPT * 256 + threadIdx.x) * 2;      // RPT pieces of 512 rows, lane owns 2 consecutive rows
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const long rr = r + (long)q * 512;
            if (rr + 1 < nrow)
#pragma unroll
                for (int j = 0; j < FPER; ++j) {
                    typedef double v2d __attribute__((ext_vector_type(2)));
                    v2d v = {1.0, 2.0};
                    v2d* p = reinterpret_cast<v2d*>(&y[(long)(fy * FPER + j) * ldy + rr]);
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
        }
    } else {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const long r = (long)rb * RPT * 256 + q * 256 + threadIdx.x;
            if (r < nrow)
#pragma unroll
                for (int j = 0; j < FPER; ++j) {
                    if (NT) __builtin_nontemporal_store(1.0, &y[(long)(fy * FPER + j) * ldy + r]); else y[(long)(fy * FPER + j) * ldy + r] = 1.0;
                }
        }
    }
}
template <int FPER, int RPT, bool NT, bool V2>
void run(const char* name, double* y, long ldy, int nrow, int nf) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rows_per_wg = 256 * RPT * (V2 ? 2 : 1);
    const int grid = ((nrow + rows_per_wg - 1) / rows_per_wg) * (nf / FPER);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<FPER, RPT, NT, V2>), dim3(grid), dim3(256), 0, 0, y, ldy, nrow, nf);
    hipDeviceSynchronize(); hipEventRecord(e0);
    const int steps = 40;
    for (int i = 0; i < steps; ++i) hipLaunchKernelGGL((k<FPER, RPT, NT, V2>), dim3(grid), dim3(256), 0, 0, y + (i & 1) * ldy * nf, ldy, nrow, nf);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / steps, B = 8.0 * nrow * nf;
    printf("%-22s ldy=%8ld fper=%2d rpt=%d nt=%d v2=%d %9.2f us  %7.0f GB/s\n", name, ldy, FPER, RPT, (int)NT, (int)V2, us, B / us / 1e3);
}
int main() {
    const int nrow = 1907329, nf = 64;
    double* y; hipMalloc(&y, (size_t)(nrow + 4096) * nf * 8 * 2);
    for (long ldy : {(long)nrow, (long)nrow + 7, (long)((nrow + 511) / 512 * 512), (long)((nrow + 511) / 512 * 512 + 32)}) {
        run<4, 1, true, false>("base", y, ldy, nrow, nf);
        run<4, 4, true, false>("4 pieces/thread", y, ldy, nrow, nf);
        run<4, 16, true, false>("16 pieces/thread", y, ldy, nrow, nf);
    }
    const long ldy2 = (nrow + 511) / 512 * 512;
    run<4, 1, true, true>("16B stores", y, ldy2, nrow, nf);
    run<4, 4, true, true>("16B stores", y, ldy2, nrow, nf);
    run<4, 4, false, true>("16B stores", y, ldy2, nrow, nf);
    run<8, 4, true, true>("16B stores", y, ldy2, nrow, nf);
    run<1, 16, true, true>("16B stores", y, ldy2, nrow, nf);
    run<1, 16, true, false>("one plane per wg", y, ldy2, nrow, nf);
    return 0;
}
