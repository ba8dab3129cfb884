// write-only floor: 1 GB and 39 MB, 8 B/lane stores (plain / nontemporal), field-major pieces like shortrow
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool NT>
__global__ void fill_k(double* __restrict__ y, long n, double v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, &y[i]); else y[i] = v; }
}
int main() {
    for (long n : {64L * 76611, 64L * 1907329}) {
        double* y; hipMalloc(&y, n * 8 * 2);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int nt = 0; nt < 2; ++nt) for (int blocks : {1024, 4096, 16384}) {
            auto launch = [&](int i) { if (nt) hipLaunchKernelGGL(fill_k<true>, dim3(blocks), dim3(256), 0, 0, y + (i & 1) * n, n, 1.0); else hipLaunchKernelGGL(fill_k<false>, dim3(blocks), dim3(256), 0, 0, y + (i & 1) * n, n, 1.0); };
            for (int i = 0; i < 10; ++i) launch(i);
            hipDeviceSynchronize(); hipEventRecord(e0);
            int steps = n > 10000000 ? 50 : 300;
            for (int i = 0; i < steps; ++i) launch(i);
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
            double us = ms * 1e3 / steps;
            printf("write %8.1f MB nt=%d blocks=%5d  %9.2f us  %7.0f GB/s\n", n * 8 / 1e6, nt, blocks, us, n * 8 / us / 1e3);
        }
        hipFree(y);
    }
    return 0;
}
