// Microbenchmark: how fast can ONE launch stream ~40 MB from HBM (cold) on MI355X?
// Each block sums a contiguous chunk of doubles; variants: bytes/lane (8|16), blocks, loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void stream_sum(const double* __restrict__ x, long n_per_block, double* __restrict__ out) {
    const double* p = x + (long)blockIdx.x * n_per_block;
    double acc = 0;
    if (VEC == 1) {
        for (long i = threadIdx.x; i < n_per_block; i += 256 * UNROLL) {
            double v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { long k = i + u * 256; v[u] = k < n_per_block ? p[k] : 0.0; }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    } else {
        const double2* q = reinterpret_cast<const double2*>(p);
        long n2 = n_per_block / 2;
        for (long i = threadIdx.x; i < n2; i += 256 * UNROLL) {
            double2 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { long k = i + u * 256; v[u] = k < n2 ? q[k] : double2{0, 0}; }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y;
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}
__global__ void empty_kernel(double* out) { if (threadIdx.x == 1234567) out[0] = 1; }

int main() {
    const long total = 64L * 76611;          // doubles per batch (39.2 MB)
    const int nbuf = 14;
    double* x; CK(hipMalloc(&x, sizeof(double) * total * nbuf));
    CK(hipMemset(x, 0, sizeof(double) * total * nbuf));
    double* out; CK(hipMalloc(&out, sizeof(double) * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char* name) {
        for (int i = 0; i < 30; ++i) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int steps = 300;
        for (int i = 0; i < steps; ++i) launch(i);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1e3 / steps;
        printf("%-40s %7.2f us  %7.0f GB/s\n", name, us, total * 8 / us / 1e3);
    };
    time([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(488), dim3(256), 0, 0, out); }, "empty 488 blocks");
    for (int blocks : {256, 488, 512, 976, 1024, 1952, 2048, 4096}) {
        long npb = (total / blocks) & ~1L;
        char name[128];
        snprintf(name, sizeof name, "8B/lane  U4  blocks=%d", blocks);
        time([&](int i) { hipLaunchKernelGGL((stream_sum<1, 4>), dim3(blocks), dim3(256), 0, 0, x + (long)(i % nbuf) * total, npb, out); }, name);
        snprintf(name, sizeof name, "8B/lane  U8  blocks=%d", blocks);
        time([&](int i) { hipLaunchKernelGGL((stream_sum<1, 8>), dim3(blocks), dim3(256), 0, 0, x + (long)(i % nbuf) * total, npb, out); }, name);
        snprintf(name, sizeof name, "16B/lane U4  blocks=%d", blocks);
        time([&](int i) { hipLaunchKernelGGL((stream_sum<2, 4>), dim3(blocks), dim3(256), 0, 0, x + (long)(i % nbuf) * total, npb, out); }, name);
        snprintf(name, sizeof name, "16B/lane U8  blocks=%d", blocks);
        time([&](int i) { hipLaunchKernelGGL((stream_sum<2, 8>), dim3(blocks), dim3(256), 0, 0, x + (long)(i % nbuf) * total, npb, out); }, name);
    }
    return 0;
}
