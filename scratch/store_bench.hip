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
// the apply's read chain in front of the stores: MODE 0 rowptr -> (colind, val) -> XT gather (three dependent rounds, as the kernel
// has them); MODE 1 a per-row copy of the first entry beside rowptr: (rowptr, col0, val0) -> XT gather (two rounds)
template <int G, int MODE>
__global__ __launch_bounds__(256) void k_chain(const int *__restrict__ rowptr, const int *__restrict__ colind, const double *__restrict__ vals,
                                               const int *__restrict__ col0, const double *__restrict__ val0, const double *__restrict__ XT, const double *__restrict__ XT2, int ldt,
                                               double *__restrict__ Y, long ldy, long nrow, int nf) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrow) return;
    int c; double v;
    const int beg = rowptr[r], end = rowptr[r + 1];
    if (MODE != 1) { c = colind[beg]; v = vals[beg]; } else { c = col0[r]; v = val0[r]; }
    if (MODE == 3) {                                                 // everything the real kernel reads of a two-entry row: 36 bytes
        const int cb = col0[r]; const double vb = val0[r], w = XT2[r];
        if (w == 0.0) v = 0.0;
        v += 1e-30 * vb * (double)cb;
    }
    if (end - beg != 1) v = 0.0;
    for (int f0 = 0; f0 < nf; f0 += G) {
        double a[G];
        const d2 *p = reinterpret_cast<const d2 *>(XT + (long)c * ldt + f0);
#pragma unroll
        for (int g = 0; g < G; g += 2) { const d2 x = p[g >> 1]; a[g] = v * x.x; a[g + 1] = v * x.y; }
        if (MODE >= 2) {                                             // a second entry per row (IvE: the two classes of an ice cell)
            const d2 *q = reinterpret_cast<const d2 *>(XT + (long)(c + 7 + (r & 1)) * ldt + f0);
#pragma unroll
            for (int g = 0; g < G; g += 2) { const d2 x = q[g >> 1]; a[g] = fma(v, x.x, a[g]); a[g + 1] = fma(v, x.y, a[g + 1]); }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) __builtin_nontemporal_store(a[g], Y + (long)(f0 + g) * ldy + r);
    }
}
__global__ void k_fill_chain(int *rowptr, int *colind, double *vals, int *col0, double *val0, long nrow, int ncol) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r > nrow) return;
    rowptr[r] = (int)r;
    if (r < nrow) { colind[r] = col0[r] = (int)((r / 55) % ncol); vals[r] = val0[r] = 1.0 + 1e-9 * (double)(r & 1023); }
}
__global__ __launch_bounds__(256) void k_flat(double *__restrict__ Y, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) { d2 x; x.x = 1.0; x.y = 2.0; __builtin_nontemporal_store(x, reinterpret_cast<d2 *>(Y + i)); }
}
int main(int argc, char **argv) {
    const long nrow = argc > 1 ? atol(argv[1]) : 17598064; const int nf = argc > 2 ? atoi(argv[2]) : 16;
    const long ldy = (nrow + 63) / 64 * 64; const int nbuf = 3;
    double *v, *Y; hipMalloc(&v, 8 * nrow); hipMemset(v, 0x3f, 8 * nrow); hipMalloc(&Y, 8 * ldy * nf * nbuf);
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
    {
        const int ncol = 60000, ldt = (nf + 15) & ~15;
        int *rowptr, *colind, *col0; double *vals, *val0, *XT;
        hipMalloc(&rowptr, 4 * (nrow + 1)); hipMalloc(&colind, 4 * nrow); hipMalloc(&col0, 4 * nrow); hipMalloc(&vals, 8 * nrow); hipMalloc(&val0, 8 * nrow);
        hipMalloc(&XT, 8l * ncol * ldt); hipMemset(XT, 0, 8l * ncol * ldt);
        hipLaunchKernelGGL(k_fill_chain, dim3((unsigned)((nrow + 256) / 256)), dim3(256), 0, 0, rowptr, colind, vals, col0, val0, nrow, ncol);
        const double Bc = 8.0 * nrow * nf + 16.0 * nrow;
#define RUNC(G, MODE) time([&](int i) { hipLaunchKernelGGL((k_chain<G, MODE>), dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, 0, rowptr, colind, vals, col0, val0, XT, v, ldt, Y + (long)(i % nbuf) * ldy * nf, ldy, nrow, nf); }, "chain mode " #MODE " planes/phase " #G, Bc)
        RUNC(4, 0); RUNC(8, 0); RUNC(16, 0); RUNC(4, 1); RUNC(8, 1); RUNC(16, 1); RUNC(4, 2); RUNC(8, 2); RUNC(16, 2); RUNC(4, 3); RUNC(8, 3);
    }
    time([&](int i) { hipLaunchKernelGGL(k_flat, dim3((unsigned)((ldy * nf / 2 + 255) / 256)), dim3(256), 0, 0, Y + (long)(i % nbuf) * ldy * nf, ldy * nf); }, "flat stream, 16 B per thread", 8.0 * ldy * nf);
    time([&](int i) { hipMemsetAsync(Y + (long)(i % nbuf) * ldy * nf, 0, 8 * ldy * nf, 0); }, "hipMemsetAsync", 8.0 * ldy * nf);
    return 0;
}
