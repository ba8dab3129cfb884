// dependent fp64 add chain: cycles per add (s_memtime), registers vs LDS broadcast vs readlane operands
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_chain(const double *in, double *out, long long *t, int mode) {
    __shared__ double buf[64];
    const int lane = threadIdx.x;
    double v = in[lane];
    buf[lane] = v;
    __syncthreads();
    double s = in[64];
    long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
#pragma unroll 1
        for (int r = 0; r < 64; ++r) {
#pragma unroll
            for (int j = 0; j < 64; ++j) s = s + v;            // register operand
        }
    } else if (mode == 1) {
#pragma unroll 1
        for (int r = 0; r < 64; ++r) {
#pragma unroll
            for (int j = 0; j < 64; ++j) s = s + buf[j];       // LDS broadcast
        }
    } else if (mode == 5) {
        // LDS -> registers in blocks of 32 values, the next block's reads issued before this block's adds
        double a[32], b[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) a[j] = buf[j];
#pragma unroll 1
        for (int r = 0; r < 64; ++r) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 32; ++j) b[j] = buf[32 + j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 32; ++j) s = s + a[j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 32; ++j) a[j] = buf[j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 32; ++j) s = s + b[j];
        }
    } else if (mode == 3) {
        if (lane == 0) {
#pragma unroll 1
            for (int r = 0; r < 64; ++r) {
#pragma unroll
                for (int j = 0; j < 64; ++j) s = s + buf[j];       // LDS, one lane
            }
        }
    } else if (mode == 4) {
        if (lane < 16) {
#pragma unroll 1
            for (int r = 0; r < 64; ++r) {
#pragma unroll
                for (int j = 0; j < 64; ++j) s = s + buf[j];       // LDS, 16 lanes
            }
        }
    } else {
        const int lo = __double2loint(v), hi = __double2hiint(v);
#pragma unroll 1
        for (int r = 0; r < 64; ++r) {
#pragma unroll
            for (int j = 0; j < 64; ++j) s = s + __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = s;
    if (lane == 0) t[mode] = t1 - t0;
}
int main() {
    double *in, *out; long long *t;
    hipMalloc(&in, 65 * 8); hipMalloc(&out, 64 * 8); hipMalloc(&t, 6 * 8);
    double h[65]; for (int i = 0; i < 65; ++i) h[i] = 1.0 + i * 1e-3;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 6; ++m) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, in, out, t, m);
    long long ht[6]; hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
    const char *names[6] = {"register operand", "LDS broadcast", "readlane pair", "LDS one lane", "LDS 16 lanes", "LDS blocks of 32"};
    for (int m = 0; m < 6; ++m) printf("%-18s %lld ticks for 4096 dependent adds = %.2f per add\n", names[m], ht[m], ht[m] / 4096.0);
    return 0;
}
