// Store-side ceiling of the I-row applies: NF planes of NROW doubles written the way spmm_shortrow_kernel writes them (a thread = R
// consecutive rows, G planes between load phases), against one flat stream.  usage: store_bench [nrow] [nf]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int R, int G, bool NT>
__global__ __launch_bounds__(256) void k_planes(const double *__restrict__ v, double *__restrict__ Y, long ldy, long nrow, int nf) {
    const long r0 = ((long)blockIdx.x * 256 + threadIdx.x) * R;
    if (r0 >= nrow) return;
    double a[R];
#pragma unroll
    for (int i = 0; i < R; ++i) a[i] = v[r0 + i];                 // (the row's value: 8 bytes read per row, as the matrix entry is)
    for (int f0 = 0; f0 < nf; f0 += G) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            double *p = Y + (long)(f0 + g) * ldy + r0;
            if (R == 1) { if (NT) __builtin_nontemporal_store(a[0] * (f0 + g + 1), p); else p[0] = a[0] * (f0 + g + 1); }
            else {
#pragma unroll
                for (int i = 0; i < R; i += 2) {
                    d2 x; x.x = a[i] * (f0 + g + 1); x.y = a[i + 1] * (f0 + g + 1);
                    if (NT) __builtin_nontemporal_store(x, reinterpret_cast<d2 *>(p + i)); else *reinterpret_cast<d2 *>(p + i) = x;
                }
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_flat(double *__restrict__ Y, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) { d2 x; x.x = 1.0; x.y = 2.0; __builtin_nontemporal_store(x, reinterpret_cast<d2 *>(Y + i)); }
}
int main(int argc, char **argv) {
    const long nrow = argc > 1 ? atol(argv[1]) : 17598064; const int nf = argc > 2 ? atoi(argv[2]) : 16;
    const long ldy = (nrow + 63) / 64 * 64; const int nbuf = 3;
    double *v, *Y; hipMalloc(&v, 8 * nrow); hipMemset(v, 0, 8 * nrow); hipMalloc(&Y, 8 * ldy * nf * nbuf);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char *name, double bytes) {
        for (int i = 0; i < 3; ++i) launch(i);
        hipDeviceSynchronize(); hipEventRecord(e0);
        const int steps = 12; for (int i = 0; i < steps; ++i) launch(i);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / steps; printf("%-40s %9.1f us  %7.0f GB/s\n", name, us, bytes / us / 1e3); fflush(stdout);
    };
    const double B = 8.0 * nrow * nf + 8.0 * nrow;
#define RUN(R, G, NT) time([&](int i) { hipLaunchKernelGGL((k_planes<R, G, NT>), dim3((unsigned)((nrow / R + 255) / 256)), dim3(256), 0, 0, v, Y + (long)(i % nbuf) * ldy * nf, ldy, nrow, nf); }, "rows/thread " #R " planes/phase " #G " nt " #NT, B)
    RUN(1, 4, true); RUN(1, 8, true); RUN(1, 16, true); RUN(1, 16, false); RUN(2, 4, true); RUN(2, 8, true); RUN(2, 16, true); RUN(4, 4, true); RUN(4, 8, true); RUN(4, 16, true); RUN(4, 16, false);
    time([&](int i) { hipLaunchKernelGGL(k_flat, dim3((unsigned)((ldy * nf / 2 + 255) / 256)), dim3(256), 0, 0, Y + (long)(i % nbuf) * ldy * nf, ldy * nf); }, "flat stream, 16 B per thread", 8.0 * ldy * nf);
    time([&](int i) { hipMemsetAsync(Y + (long)(i % nbuf) * ldy * nf, 0, 8 * ldy * nf, 0); }, "hipMemsetAsync", 8.0 * ldy * nf);
    return 0;
}
