// empty_bench.hip -- the floor of ONE kernel launch on this stack: an empty kernel on several grids, 200 launches each, timed
// (a) by HIP events attached to the dispatch (hipExtLaunchKernelGGL start / stop: what the library's bench hooks use) and (b) by
// whatever wraps this program (run it directly after `rocprofv3 --kernel-trace --stats --`: the trace's durations of k_empty_<grid>
// are the cross-check the round-3 review asked for).   hipcc -O3 --offload-arch=gfx950 scratch/empty_bench.hip -o scratch/empty_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-result"
template <int TAG> __global__ __launch_bounds__(256) void k_empty(int *p) { if (p && threadIdx.x == 9999) *p = TAG; }
template <int TAG> static void run(int blocks) {
    const int N = 200;
    std::vector<hipEvent_t> e0(N), e1(N);
    for (int i = 0; i < N; ++i) { hipEventCreate(&e0[i]); hipEventCreate(&e1[i]); }
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty<TAG>, dim3(blocks), dim3(256), 0, 0, (int *)nullptr);
    hipDeviceSynchronize();
    for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(k_empty<TAG>, dim3(blocks), dim3(256), 0, 0, e0[i], e1[i], 0, (int *)nullptr);
    hipDeviceSynchronize();
    std::vector<float> us(N);
    for (int i = 0; i < N; ++i) { float ms; hipEventElapsedTime(&ms, e0[i], e1[i]); us[i] = ms * 1e3f; }
    std::sort(us.begin(), us.end());
    printf("EMPTY grid %5d x 256 (k_empty<%d>): HIP events med %6.2f  min %6.2f  p90 %6.2f us\n", blocks, TAG, us[N / 2], us[0], us[N * 9 / 10]);
}
int main() {
    run<64>(64); run<256>(256); run<976>(976); run<1952>(1952); run<4096>(4096);
    return 0;
}
