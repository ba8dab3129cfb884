// prims.hip -- exclusive scan and stable LSD radix sort for wave64 (gfx950).
//
// These are the ordering primitives behind the COO -> CSR assembly (K2): the
// reference obtains the same orderings from Eigen's setFromTriplets
// (RegridMatrices_Dynamic.cpp:75-81,86-90,178-183,187-190) and from
// spsparse::SparseSet's first-seen numbering.  Both are integer/byte work and
// HBM-bound: a scan moves 12 B/element, a sort pass 2 x 12 B/element + a
// histogram read.  Everything is deterministic (no float atomics, stable sort).
#include "prims.h"

namespace ibh {

Arena::~Arena() { for (auto &b : blocks) (void)hipFree(b.p); }
void *Arena::get_bytes(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    for (auto &b : blocks)
        if (b.cap - b.used >= bytes) { char *p = b.p + b.used; b.used += bytes; return p; }
    const size_t cap = bytes > (size_t(64) << 20) ? bytes : (size_t(64) << 20);
    Block nb{nullptr, cap, bytes};
    IBH_HIP(hipMalloc(reinterpret_cast<void **>(&nb.p), cap));
    blocks.push_back(nb);
    return nb.p;
}
Arena &arena() { static thread_local Arena a; return a; }
void release_workspace() {
    Arena &a = arena();
    for (auto &b : a.blocks) (void)hipFree(b.p);
    a.blocks.clear();
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread over the block; total = block sum.
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan_u32(v, lane);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wave) wbase += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return wbase + inc - v;
}

// ---- scan ----------------------------------------------------------------------------------
constexpr int SC_T = 256, SC_I = 8, SC_TILE = SC_T * SC_I;

__global__ __launch_bounds__(SC_T) void scan_tile_sums(const uint32_t *__restrict__ in, size_t n,
                                                       uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_wave[SC_T / 64];
    const size_t base = (size_t)blockIdx.x * SC_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SC_I; ++i) {
        const size_t idx = base + (size_t)i * SC_T + threadIdx.x;
        if (idx < n) s += in[idx];
    }
    uint32_t tot;
    (void)block_excl_scan<SC_T>(s, s_wave, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void scan_sums_inplace(uint32_t *__restrict__ sums, int nb,
                                                          uint32_t *__restrict__ total) {
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nb ? sums[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<1024>(v, s_wave, tot);
        if (i < nb) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ __launch_bounds__(SC_T) void scan_tile_apply(const uint32_t *in, uint32_t *out, size_t n,
                                                        const uint32_t *__restrict__ offs) {
    __shared__ uint32_t tile[SC_TILE + SC_T];   // +1 pad per 8: thread t owns tile[9t .. 9t+7]
    __shared__ uint32_t s_wave[SC_T / 64];
    const size_t base = (size_t)blockIdx.x * SC_TILE;
#pragma unroll
    for (int i = 0; i < SC_I; ++i) {
        const int e = i * SC_T + threadIdx.x;
        const size_t idx = base + e;
        tile[e + (e >> 3)] = idx < n ? in[idx] : 0u;
    }
    __syncthreads();
    uint32_t v[SC_I], sum = 0;
#pragma unroll
    for (int i = 0; i < SC_I; ++i) { v[i] = tile[threadIdx.x * 9 + i]; sum += v[i]; }
    uint32_t tot;
    uint32_t run = block_excl_scan<SC_T>(sum, s_wave, tot) + offs[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SC_I; ++i) { tile[threadIdx.x * 9 + i] = run; run += v[i]; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SC_I; ++i) {
        const int e = i * SC_T + threadIdx.x;
        const size_t idx = base + e;
        if (idx < n) out[idx] = tile[e + (e >> 3)];
    }
}

void exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total, hipStream_t stream) {
    if (n == 0) {
        if (total) IBH_HIP(hipMemsetAsync(total, 0, sizeof(uint32_t), stream));
        return;
    }
    const size_t nb = (n + SC_TILE - 1) / SC_TILE;
    IBH_CHECK(nb < (1ul << 31), "scan too large");
    uint32_t *sums = arena().get<uint32_t>(nb);
    hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)nb), dim3(SC_T), 0, stream, in, n, sums);
    hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(1024), 0, stream, sums, (int)nb, total);
    hipLaunchKernelGGL(scan_tile_apply, dim3((unsigned)nb), dim3(SC_T), 0, stream, in, out, n, sums);
    IBH_HIP(hipGetLastError());
}

// ---- radix sort ----------------------------------------------------------------------------
constexpr int RS_T = 256, RS_I = 16, RS_TILE = RS_T * RS_I;

__global__ __launch_bounds__(RS_T) void rs_hist(const uint64_t *__restrict__ keys, size_t n, int shift,
                                                uint32_t mask, uint32_t *__restrict__ table, int nblocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_I; ++i) {
        const size_t idx = base + (size_t)i * RS_T + threadIdx.x;
        if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= mask) table[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Wave w of the block owns the contiguous chunk [w*64*RS_I, (w+1)*64*RS_I) of the tile and walks
// it in rows of 64 keys, so (wave, row, lane) order == global order and ranks are stable.
__global__ __launch_bounds__(RS_T) void rs_scatter(const uint64_t *__restrict__ keys_in,
                                                   const uint32_t *__restrict__ vals_in,
                                                   uint64_t *__restrict__ keys_out,
                                                   uint32_t *__restrict__ vals_out, size_t n, int shift,
                                                   int nbits, const uint32_t *__restrict__ table, int nblocks) {
    __shared__ uint32_t cnt[RS_T / 64][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t mask = (1u << nbits) - 1u;
    for (int i = threadIdx.x; i < (RS_T / 64) * 256; i += RS_T) (&cnt[0][0])[i] = 0;
    __syncthreads();

    const size_t wbase = (size_t)blockIdx.x * RS_TILE + (size_t)wave * 64 * RS_I;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    volatile uint32_t *wc = cnt[wave];
    uint64_t key[RS_I];
    uint32_t val[RS_I], rank[RS_I];
#pragma unroll
    for (int i = 0; i < RS_I; ++i) {
        const size_t idx = wbase + (size_t)i * 64 + lane;
        const bool ok = idx < n;
        key[i] = ok ? keys_in[idx] : 0ull;
        val[i] = ok ? vals_in[idx] : 0u;
        const uint32_t d = (uint32_t)(key[i] >> shift) & mask;
        unsigned long long peers = __ballot(ok);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t old = 0;
        const int leader = ok ? __ffsll((long long)peers) - 1 : 0;
        if (ok && lane == leader) {
            old = wc[d];
            wc[d] = old + (uint32_t)__popcll(peers);
        }
        old = __shfl(old, leader, 64);
        rank[i] = old + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    // digit d: global start of this block's run, then running offsets over the 4 waves
    if (threadIdx.x <= mask) {
        uint32_t run = table[(size_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < RS_T / 64; ++w) {
            const uint32_t c = cnt[w][threadIdx.x];
            cnt[w][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_I; ++i) {
        const size_t idx = wbase + (size_t)i * 64 + lane;
        if (idx < n) {
            const uint32_t d = (uint32_t)(key[i] >> shift) & mask;
            const uint32_t pos = cnt[wave][d] + rank[i];
            keys_out[pos] = key[i];
            vals_out[pos] = val[i];
        }
    }
}

bool radix_sort_pairs(uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t n,
                      const KeyField *fields, int nfields, hipStream_t stream) {
    if (n <= 1) return false;
    IBH_CHECK(n < (1ul << 32), "sort too large");
    const size_t nblocks = (n + RS_TILE - 1) / RS_TILE;
    uint32_t *table = arena().get<uint32_t>(256 * nblocks);
    bool in_alt = false;
    for (int f = 0; f < nfields; ++f) {
        int done = 0;
        const int total = fields[f].nbits;
        const int passes = (total + 7) / 8;
        for (int p = 0; p < passes; ++p) {
            const int nbits = (total - done + (passes - p) - 1) / (passes - p);   // spread bits evenly
            const int shift = fields[f].shift + done;
            const uint32_t mask = (1u << nbits) - 1u;
            const uint64_t *kin = in_alt ? keys_alt : keys;
            const uint32_t *vin = in_alt ? vals_alt : vals;
            uint64_t *kout = in_alt ? keys : keys_alt;
            uint32_t *vout = in_alt ? vals : vals_alt;
            hipLaunchKernelGGL(rs_hist, dim3((unsigned)nblocks), dim3(RS_T), 0, stream, kin, n, shift, mask,
                               table, (int)nblocks);
            exclusive_scan_u32(table, table, (size_t)(mask + 1) * nblocks, nullptr, stream);
            hipLaunchKernelGGL(rs_scatter, dim3((unsigned)nblocks), dim3(RS_T), 0, stream, kin, vin, kout, vout, n,
                               shift, nbits, table, (int)nblocks);
            IBH_HIP(hipGetLastError());
            in_alt = !in_alt;
            done += nbits;
        }
    }
    return in_alt;
}

}  // namespace ibh
