#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(double* out) { if (threadIdx.x == 1234567) out[0] = 1; }
__global__ __launch_bounds__(1024) void empty_kernel1k(double* out) { if (threadIdx.x == 1234567) out[0] = 1; }
int main() {
    double* out; hipMalloc(&out, 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {64, 256, 1024}) for (int blocks : {64, 256, 512, 1024, 2048, 4096, 8192}) {
        auto launch = [&]() { if (threads == 1024) hipLaunchKernelGGL(empty_kernel1k, dim3(blocks), dim3(threads), 0, 0, out); else hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(threads), 0, 0, out); };
        for (int i = 0; i < 50; ++i) launch();
        hipDeviceSynchronize(); hipEventRecord(e0);
        for (int i = 0; i < 1000; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty threads=%4d blocks=%5d  %6.2f us\n", threads, blocks, ms);
    }
    return 0;
}
