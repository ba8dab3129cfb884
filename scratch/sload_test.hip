// sload_test.hip -- sequential (ordered) row sums, one wave per row: (A) the v_readlane chain over coalesced chunks, (B) wide SCALAR
// loads (s_load_dwordx16 from an 8-byte-aligned address) feeding v_add_f64 directly.  Checks B bitwise against the host at odd
// alignments and times both.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scratch/sload_test.hip -o scratch/sload_test
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
struct __attribute__((aligned(8))) D8 { double v[8]; };
__global__ __launch_bounds__(256) void k_scalar(const double *__restrict__ vs, const uint32_t *__restrict__ rp, double *__restrict__ out, int n) {
    const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= n) return;
    const uint32_t b = rp[r], e = rp[r + 1];
    double sum = 0.0;
    const double *p = vs + b, *pe = vs + e;
    for (; p + 32 <= pe; p += 32) {
        const D8 d0 = *reinterpret_cast<const D8 *>(p), d1 = *reinterpret_cast<const D8 *>(p + 8);
        const D8 d2 = *reinterpret_cast<const D8 *>(p + 16), d3 = *reinterpret_cast<const D8 *>(p + 24);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum = sum + d0.v[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum = sum + d1.v[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum = sum + d2.v[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum = sum + d3.v[j];
    }
    for (; p + 8 <= pe; p += 8) {
        const D8 d = *reinterpret_cast<const D8 *>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum = sum + d.v[j];
    }
    for (; p < pe; ++p) sum = sum + *p;
    if ((threadIdx.x & 63) == 0) out[r] = sum;
}
__global__ __launch_bounds__(256) void k_chain(const double *__restrict__ vs, const uint32_t *__restrict__ rp, double *__restrict__ out, int n) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    const uint32_t b = rp[r], e = rp[r + 1];
    double sum = 0.0;
    for (uint32_t base = b; base < e; base += 256) {
        double v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const uint32_t k = base + 64u * c + lane; v[c] = k < e ? vs[k] : 0.0; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t cb = base + 64u * c;
            if (cb >= e) break;
            const int cnt = (int)min(64u, e - cb);
            const int lo = __double2loint(v[c]), hi = __double2hiint(v[c]);
            for (int j = 0; j < cnt; ++j) sum = sum + __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));
        }
    }
    if (lane == 0) out[r] = sum;
}
// (C) a THREAD per row: every lane walks its own row (64 rows of a wave in flight; loads touch 64 lines per instruction)
__global__ __launch_bounds__(256) void k_thread(const double *__restrict__ vs, const uint32_t *__restrict__ rp, double *__restrict__ out, int n) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const uint32_t b = rp[r], e = rp[r + 1];
    double sum = 0.0;
    uint32_t k = b;
    for (; k + 8 <= e; k += 8) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = vs[k + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum = sum + t[j];
    }
    for (; k < e; ++k) sum = sum + vs[k];
    out[r] = sum;
}
// (D) FOUR rows per wave: lanes 16k .. 16k+15 hold 16 consecutive values of row k's chunk; step j broadcasts lane j of every
// 16-lane row to its row (DPP row_newbcast) -- 2 moves + 1 add serve four rows' chains at once.  Shorter rows are padded with +0.0
// (sum + 0.0 == sum bitwise: the sums here start from +0.0 and can never be -0.0).
template <int J> __device__ __forceinline__ double rowbcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + J, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + J, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(256) void k_rows4(const double *__restrict__ vs, const uint32_t *__restrict__ rp, double *__restrict__ out, int n) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, li = lane & 15;
    const int r = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;
    const uint32_t b = r < n ? rp[r] : 0u, e = r < n ? rp[r + 1] : 0u;
    uint32_t len = e - b;
    // the longest of the wave's four rows bounds the loop (wave-uniform)
    uint32_t mx = len;
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, 16, 64)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, 32, 64));
    mx = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);
    double sum = 0.0;
    for (uint32_t base = 0; base < mx; base += 128) {             // eight loads of 16 values per row in flight
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t k = base + 16u * c + li;
            v[c] = k < len ? vs[b + k] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (base + 16u * c >= mx) break;
            sum = sum + rowbcast<0>(v[c]); sum = sum + rowbcast<1>(v[c]); sum = sum + rowbcast<2>(v[c]); sum = sum + rowbcast<3>(v[c]);
            sum = sum + rowbcast<4>(v[c]); sum = sum + rowbcast<5>(v[c]); sum = sum + rowbcast<6>(v[c]); sum = sum + rowbcast<7>(v[c]);
            sum = sum + rowbcast<8>(v[c]); sum = sum + rowbcast<9>(v[c]); sum = sum + rowbcast<10>(v[c]); sum = sum + rowbcast<11>(v[c]);
            sum = sum + rowbcast<12>(v[c]); sum = sum + rowbcast<13>(v[c]); sum = sum + rowbcast<14>(v[c]); sum = sum + rowbcast<15>(v[c]);
        }
    }
    if (li == 0 && r < n) out[r] = sum;
}
int main() {
    const int n = 50000;
    std::mt19937_64 g(7);
    std::vector<uint32_t> rp(n + 1, 0);
    for (int i = 0; i < n; ++i) rp[i + 1] = rp[i] + 300 + (uint32_t)(g() % 101);      // ~350 per row, every alignment
    const size_t nnz = rp[n];
    std::vector<double> v(nnz);
    for (auto &x : v) x = (double)(g() % 1000003) * 1.37e-3 + 1e-9 * (double)(g() % 977);
    std::vector<double> ref(n);
    for (int i = 0; i < n; ++i) { double s = 0.0; for (uint32_t k = rp[i]; k < rp[i + 1]; ++k) s = s + v[k]; ref[i] = s; }
    double *dv, *dout; uint32_t *drp;
    hipMalloc(&dv, 8 * nnz + 4096); hipMalloc(&dout, 8 * n); hipMalloc(&drp, 4 * (n + 1));
    hipMemcpy(dv, v.data(), 8 * nnz, hipMemcpyHostToDevice); hipMemcpy(drp, rp.data(), 4 * (n + 1), hipMemcpyHostToDevice);
    std::vector<double> got(n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(dout, 0, 8 * n);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_chain, dim3((n + 3) / 4), dim3(256), 0, 0, dv, drp, dout, n);
            else if (mode == 3) hipLaunchKernelGGL(k_rows4, dim3((n + 15) / 16), dim3(256), 0, 0, dv, drp, dout, n);
            else if (mode == 2) hipLaunchKernelGGL(k_thread, dim3((n + 255) / 256), dim3(256), 0, 0, dv, drp, dout, n);
            else hipLaunchKernelGGL(k_scalar, dim3((n + 3) / 4), dim3(256), 0, 0, dv, drp, dout, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(got.data(), dout, 8 * n, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < n; ++i) bad += memcmp(&got[i], &ref[i], 8) != 0;
            printf("%s rep %d: %.1f us for %zu values (%.2f ns/value), mismatches %d\n", mode == 3 ? "rows4 " : mode == 2 ? "thread" : mode ? "scalar" : "chain ", rep, ms * 1e3, nnz, ms * 1e6 / nnz, bad);
        }
    }
    return 0;
}
