// prims.hip -- exclusive scan and stable LSD radix sort for wave64 (gfx950).
//
// These are the ordering primitives behind the COO -> CSR assembly (K2): the
// reference obtains the same orderings from Eigen's setFromTriplets
// (RegridMatrices_Dynamic.cpp:75-81,86-90,178-183,187-190) and from
// spsparse::SparseSet's first-seen numbering.  Both are integer/byte work and
// HBM-bound: a scan moves 12 B/element, a sort pass 2 x 12 B/element + a
// histogram read.  Everything is deterministic (no float atomics, stable sort).
#include "prims.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

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
// One arena per (host thread, device): blocks hipMalloc'd on one GPU must never serve a build that
// runs on another (ibh_set_device switches devices inside one process).  Every thread's set is
// registered process-wide so that ibh_release_cached_memory() reaches the worker threads' arenas too
// (assemble_batch: three workers, each with blocks of >= 64 MB and GBs after an Antarctic batch).
namespace {
struct ArenaSet;
struct ArenaRegistry {
    std::mutex mu;
    std::vector<ArenaSet *> sets;
};
ArenaRegistry &arena_registry() { static ArenaRegistry *r = new ArenaRegistry; return *r; }      // leaked: outlives thread_local teardown
struct ArenaSet {
    std::map<int, Arena> by_device;
    ArenaSet() {
        std::lock_guard<std::mutex> lk(arena_registry().mu);
        arena_registry().sets.push_back(this);
    }
    ~ArenaSet() {
        std::lock_guard<std::mutex> lk(arena_registry().mu);
        auto &v = arena_registry().sets;
        v.erase(std::remove(v.begin(), v.end(), this), v.end());
    }
};
ArenaSet &arena_set() { static thread_local ArenaSet s; return s; }
}  // namespace
Arena &arena() {
    int dev = 0;
    IBH_HIP(hipGetDevice(&dev));
    ArenaSet &s = arena_set();
    std::lock_guard<std::mutex> lk(arena_registry().mu);      // the map may be walked by release_workspace()
    return s.by_device[dev];
}
// Frees the blocks of every thread's arenas.  The caller guarantees that no build is in flight (a build holds
// pointers into its thread's arena).
void release_workspace() {
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    std::lock_guard<std::mutex> lk(arena_registry().mu);
    for (ArenaSet *set : arena_registry().sets)
        for (auto &kv : set->by_device) {
            if (have) (void)hipSetDevice(kv.first);
            for (auto &b : kv.second.blocks) (void)hipFree(b.p);
            kv.second.blocks.clear();
        }
    if (have) (void)hipSetDevice(cur);
}

void readback_sync(void *dst, const void *dsrc, size_t bytes, hipStream_t stream) {
    struct Pinned {
        void *p = nullptr;
        ~Pinned() { if (p) (void)hipHostFree(p); }
    };
    static thread_local Pinned pin;
    IBH_CHECK(bytes <= 256, "internal: read-back of %zu bytes", bytes);
    if (!pin.p) IBH_HIP(hipHostMalloc(&pin.p, 256, hipHostMallocDefault));
    IBH_HIP(hipMemcpyAsync(pin.p, dsrc, bytes, hipMemcpyDeviceToHost, stream));
    IBH_HIP(hipStreamSynchronize(stream));
    memcpy(dst, pin.p, bytes);
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

template <class IN>
__global__ __launch_bounds__(SC_T) void scan_tile_sums(const IN *__restrict__ in, size_t n,
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

template <class IN>
__global__ __launch_bounds__(SC_T) void scan_tile_apply(const IN *in, uint32_t *out, size_t n,
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

// ---- the same in ONE launch: chained scan (decoupled look-back) ---------------------------------------------------------------
// Every tile's workgroup publishes its sum, wave 0 looks back over its predecessors 64 at a time -- a predecessor's word says
// {sum of that tile alone} or {sum of everything up to and including it}; the first of the second kind ends the walk -- and
// publishes the tile's own inclusive prefix.  Workgroups are dispatched in index order, so a tile only ever waits for tiles that are
// already running or done.  A status word is one 64-bit value {sum, epoch << 2 | kind}, stored and loaded whole at device scope
// (coherent across the XCDs' L2s); the epoch of the call makes every older word invalid, so the buffer is never cleared.  A 5 km
// build makes three to eight scans of <= 182 k elements: two launches fewer each (~5 us apiece at that size, round 5).
namespace {
struct ScanState {
    unsigned long long *status = nullptr;
    size_t cap = 0;
    unsigned epoch = 0;
    int device = -1;
};
ScanState &scan_state() { static thread_local ScanState s; return s; }
}  // namespace
template <class IN>
__global__ __launch_bounds__(SC_T) void scan_chained(const IN *__restrict__ in, uint32_t *__restrict__ out, size_t n, unsigned long long *status,
                                                     unsigned epoch, uint32_t *__restrict__ total) {
    __shared__ uint32_t tile[SC_TILE + SC_T];   // +1 pad per 8: thread t owns tile[9t .. 9t+7]
    __shared__ uint32_t s_wave[SC_T / 64];
    __shared__ uint32_t s_prefix;
    const size_t t = blockIdx.x, base = t * SC_TILE;
#pragma unroll
    for (int i = 0; i < SC_I; ++i) {
        const int e = i * SC_T + threadIdx.x;
        const size_t idx = base + e;
        tile[e + (e >> 3)] = idx < n ? (uint32_t)in[idx] : 0u;
    }
    __syncthreads();
    uint32_t v[SC_I], sum = 0;
#pragma unroll
    for (int i = 0; i < SC_I; ++i) { v[i] = tile[threadIdx.x * 9 + i]; sum += v[i]; }
    uint32_t tot;
    uint32_t run = block_excl_scan<SC_T>(sum, s_wave, tot);
    constexpr unsigned long long AGG = 1, PRE = 2;
    const unsigned long long tag = (unsigned long long)epoch << 2;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        uint32_t prefix = 0;
        if (t == 0) {
            if (lane == 0) __hip_atomic_store(status, ((unsigned long long)tot << 32) | tag | PRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(status + t, ((unsigned long long)tot << 32) | tag | AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long j = (long)t - 1;
            while (true) {
                const long idx = j - lane;
                unsigned long long w = ((unsigned long long)0 << 32) | tag | PRE;      // before the first tile: an empty prefix
                if (idx >= 0) {
                    do { w = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    while ((w & 0xfffffffcull) != (tag & 0xfffffffcull) || (w & 3ull) == 0);
                }
                const unsigned long long pre = __ballot((w & 3ull) == PRE);
                uint32_t val = (uint32_t)(w >> 32);
                if (pre) {
                    const int first = __builtin_ctzll(pre);      // the nearest predecessor that knows everything before it
                    if (lane > first) val = 0;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) val += __shfl_xor(val, off, 64);
                prefix += val;
                if (pre) break;
                j -= 64;
            }
            if (lane == 0) __hip_atomic_store(status + t, ((unsigned long long)(prefix + tot) << 32) | tag | PRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_prefix = prefix;
            if (total && t == gridDim.x - 1) *total = prefix + tot;
        }
    }
    __syncthreads();
    run += s_prefix;
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

template <class IN>
static void exclusive_scan_any(const IN *in, uint32_t *out, size_t n, uint32_t *total, hipStream_t stream) {
    if (n == 0) {
        if (total) IBH_HIP(hipMemsetAsync(total, 0, sizeof(uint32_t), stream));
        return;
    }
    const size_t nb = (n + SC_TILE - 1) / SC_TILE;
    IBH_CHECK(nb < (1ul << 31), "scan too large");
    if (get_tuning("scan_chained", 1)) {
        ScanState &ss = scan_state();
        int dev = 0;
        IBH_HIP(hipGetDevice(&dev));
        if (ss.device != dev || ss.cap < nb) {                  // (the words of an older, smaller buffer die with it: a fresh one starts at zero)
            if (ss.status) (void)hipFree(ss.status);
            ss.cap = std::max<size_t>(nb, 4096) * 2;
            IBH_HIP(hipMalloc(&ss.status, ss.cap * sizeof(unsigned long long)));
            IBH_HIP(hipMemsetAsync(ss.status, 0, ss.cap * sizeof(unsigned long long), stream));
            IBH_HIP(hipStreamSynchronize(stream));              // (other streams of this thread may scan next)
            ss.device = dev; ss.epoch = 0;
        }
        if (++ss.epoch >= (1u << 30)) {                         // (the epoch field is 30 bits: start over on a cleared buffer)
            IBH_HIP(hipDeviceSynchronize());
            IBH_HIP(hipMemsetAsync(ss.status, 0, ss.cap * sizeof(unsigned long long), stream));
            IBH_HIP(hipStreamSynchronize(stream));
            ss.epoch = 1;
        }
        hipLaunchKernelGGL(scan_chained<IN>, dim3((unsigned)nb), dim3(SC_T), 0, stream, in, out, n, ss.status, ss.epoch, total);
        IBH_HIP(hipGetLastError());
        return;
    }
    uint32_t *sums = arena().get<uint32_t>(nb);
    hipLaunchKernelGGL(scan_tile_sums<IN>, dim3((unsigned)nb), dim3(SC_T), 0, stream, in, n, sums);
    hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(1024), 0, stream, sums, (int)nb, total);
    hipLaunchKernelGGL(scan_tile_apply<IN>, dim3((unsigned)nb), dim3(SC_T), 0, stream, in, out, n, sums);
    IBH_HIP(hipGetLastError());
}
void exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total, hipStream_t stream) {
    exclusive_scan_any(in, out, n, total, stream);
}
void exclusive_scan_u8(const uint8_t *in, uint32_t *out, size_t n, uint32_t *total, hipStream_t stream) {
    exclusive_scan_any(in, out, n, total, stream);
}

// ---- three-channel scan -----------------------------------------------------------------------
// The generator passes of the assembly produce, per exchange cell, three tiny counts (new row keys,
// new column keys, contributions).  Scanning them as ONE packed stream (4 B in, 12 B out per cell,
// three kernels) replaces three separate scans.  pk bits: [0,2) row first-occurrence flags, [2,4)
// column flags, [4,6) number of contributions.  Inside a tile the three running sums share a u64
// (21 bits each: a tile sums to at most 2 * 2048).
__device__ __forceinline__ uint64_t s3_unpack(uint32_t pk) {
    return (uint64_t)__popc(pk & 3u) | ((uint64_t)__popc((pk >> 2) & 3u) << 21) | ((uint64_t)((pk >> 4) & 3u) << 42);
}
template <int THREADS>
__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t *s_wave, uint64_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint64_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
        const uint64_t t = s_wave[w];
        if (w < wave) wbase += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return wbase + inc - v;
}
__global__ __launch_bounds__(SC_T) void s3_tile_sums(const uint32_t *__restrict__ pk, size_t n, uint32_t *__restrict__ sums,
                                                     size_t nb) {
    __shared__ uint64_t s_wave[SC_T / 64];
    const size_t base = (size_t)blockIdx.x * SC_TILE;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SC_I; ++i) {
        const size_t idx = base + (size_t)i * SC_T + threadIdx.x;
        if (idx < n) s += s3_unpack(pk[idx]);
    }
    uint64_t tot;
    (void)block_excl_scan_u64<SC_T>(s, s_wave, tot);
    if (threadIdx.x == 0) {
        sums[blockIdx.x] = (uint32_t)(tot & 0x1fffffu);
        sums[nb + blockIdx.x] = (uint32_t)((tot >> 21) & 0x1fffffu);
        sums[2 * nb + blockIdx.x] = (uint32_t)(tot >> 42);
    }
}
__global__ __launch_bounds__(1024) void s3_sums_inplace(uint32_t *__restrict__ sums, int nb, uint32_t *__restrict__ totals) {
    __shared__ uint32_t s_wave[16];
    for (int ch = 0; ch < 3; ++ch) {
        uint32_t *p = sums + (size_t)ch * nb;
        uint32_t carry = 0;
        for (int base = 0; base < nb; base += 1024) {
            const int i = base + threadIdx.x;
            const uint32_t v = i < nb ? p[i] : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<1024>(v, s_wave, tot);
            if (i < nb) p[i] = carry + ex;
            carry += tot;
        }
        if (threadIdx.x == 0) totals[ch] = carry;
    }
}
__global__ __launch_bounds__(SC_T) void s3_tile_apply(const uint32_t *__restrict__ pk, size_t n, const uint32_t *__restrict__ sums,
                                                      size_t nb, uint32_t *__restrict__ o0, uint32_t *__restrict__ o1,
                                                      uint32_t *__restrict__ o2) {
    __shared__ uint32_t tile[SC_TILE + SC_T];
    __shared__ uint64_t s_wave[SC_T / 64];
    const size_t base = (size_t)blockIdx.x * SC_TILE;
#pragma unroll
    for (int i = 0; i < SC_I; ++i) {
        const int e = i * SC_T + threadIdx.x;
        const size_t idx = base + e;
        tile[e + (e >> 3)] = idx < n ? pk[idx] : 0u;
    }
    __syncthreads();
    uint64_t v[SC_I], sum = 0;
#pragma unroll
    for (int i = 0; i < SC_I; ++i) { v[i] = s3_unpack(tile[threadIdx.x * 9 + i]); sum += v[i]; }
    uint64_t tot;
    uint64_t run = block_excl_scan_u64<SC_T>(sum, s_wave, tot);
    const uint32_t b0 = sums[blockIdx.x], b1 = sums[nb + blockIdx.x], b2 = sums[2 * nb + blockIdx.x];
#pragma unroll
    for (int i = 0; i < SC_I; ++i) {
        const size_t idx = base + (size_t)threadIdx.x * SC_I + i;
        if (idx < n) {
            o0[idx] = b0 + (uint32_t)(run & 0x1fffffu);
            o1[idx] = b1 + (uint32_t)((run >> 21) & 0x1fffffu);
            o2[idx] = b2 + (uint32_t)(run >> 42);
        }
        run += v[i];
    }
}
void exclusive_scan3(const uint32_t *pk, size_t n, uint32_t *o0, uint32_t *o1, uint32_t *o2, uint32_t *totals3,
                     hipStream_t stream) {
    if (n == 0) {
        IBH_HIP(hipMemsetAsync(totals3, 0, 3 * sizeof(uint32_t), stream));
        return;
    }
    const size_t nb = (n + SC_TILE - 1) / SC_TILE;
    IBH_CHECK(nb < (1ul << 31), "scan too large");
    uint32_t *sums = arena().get<uint32_t>(3 * nb);
    hipLaunchKernelGGL(s3_tile_sums, dim3((unsigned)nb), dim3(SC_T), 0, stream, pk, n, sums, nb);
    hipLaunchKernelGGL(s3_sums_inplace, dim3(1), dim3(1024), 0, stream, sums, (int)nb, totals3);
    hipLaunchKernelGGL(s3_tile_apply, dim3((unsigned)nb), dim3(SC_T), 0, stream, pk, n, sums, nb, o0, o1, o2);
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
// it in rows of 64 keys, so (wave, row, lane) order == global order and ranks are stable.  The tile
// is first put in digit order in LDS and then written out: consecutive threads store consecutive
// addresses inside each digit's run (a 4096-key tile over 256 digits gives ~128-byte runs) instead
// of scattering single 12-byte pairs.
__global__ __launch_bounds__(RS_T) void rs_scatter(const uint64_t *__restrict__ keys_in,
                                                   const uint32_t *__restrict__ vals_in,
                                                   uint64_t *__restrict__ keys_out,
                                                   uint32_t *__restrict__ vals_out, size_t n, int shift,
                                                   int nbits, const uint32_t *__restrict__ table, int nblocks) {
    __shared__ uint64_t sk[RS_TILE];
    __shared__ uint32_t sv[RS_TILE];
    __shared__ uint32_t cnt[RS_T / 64][256];
    __shared__ uint32_t gbase[256];          // global position of the digit's run minus its start inside the tile
    __shared__ uint32_t s_wave[RS_T / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t mask = (1u << nbits) - 1u;
    for (int i = threadIdx.x; i < (RS_T / 64) * 256; i += RS_T) (&cnt[0][0])[i] = 0;
    __syncthreads();

    const size_t tbase = (size_t)blockIdx.x * RS_TILE;
    const size_t wbase = tbase + (size_t)wave * 64 * RS_I;
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
    // digit d (thread d): start of its run inside the tile, then running offsets over the 4 waves
    {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < RS_T / 64; ++w) tot += cnt[w][threadIdx.x];
        uint32_t all;
        uint32_t run = block_excl_scan<RS_T>(tot, s_wave, all);
        gbase[threadIdx.x] = threadIdx.x <= mask ? table[(size_t)threadIdx.x * nblocks + blockIdx.x] - run : 0u;
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
            const uint32_t lp = cnt[wave][d] + rank[i];
            sk[lp] = key[i];
            sv[lp] = val[i];
        }
    }
    __syncthreads();
    const uint32_t ntile = (uint32_t)(n - tbase < (size_t)RS_TILE ? n - tbase : (size_t)RS_TILE);
#pragma unroll
    for (int i = 0; i < RS_I; ++i) {
        const uint32_t lp = (uint32_t)i * RS_T + threadIdx.x;
        if (lp < ntile) {
            const uint64_t k = sk[lp];
            const uint32_t pos = gbase[(uint32_t)(k >> shift) & mask] + lp;
            keys_out[pos] = k;
            vals_out[pos] = sv[lp];
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


// ---- adaptive ordering ------------------------------------------------------------------------
// A sorted exchange grid (ExchangeGrid's constructor sorts the cells, AbbrGrid.cpp:10-21) emits
// contributions that are already ALMOST in (row, col) order: the first-seen dense numbering is
// monotone along the emission order except where an ice cell straddles two atmosphere cells.  A
// full radix sort moves every element ~10 times to fix that.  Instead:
//   1. one analysis pass finds every position k with max(keys[0,k)) <= min(keys[k,n)) ("cuts"):
//      the pieces between cuts can be sorted independently;
//   2. pieces of <= 8192 elements are sorted in LDS (a stable LSD radix sort of the piece: the same order the device-wide
//      radix sort would give);
//   3. if a piece is larger, nothing is touched and the caller falls back to the radix sort,
//      skipping the low field when it is already non-decreasing along the sequence.
// Everything is integer work; the result is identical to radix_sort_pairs by construction.
constexpr int OA_T = 256, OA_I = 8, OA_TILE = OA_T * OA_I;

__device__ __forceinline__ uint64_t u64max(uint64_t a, uint64_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint64_t u64min(uint64_t a, uint64_t b) { return a < b ? a : b; }

__global__ __launch_bounds__(OA_T) void oa_tiles(const uint64_t *__restrict__ keys, size_t n,
                                                 uint64_t *__restrict__ tmin, uint64_t *__restrict__ tmax,
                                                 uint32_t *__restrict__ tflags) {
    __shared__ uint64_t s_mn[OA_T / 64], s_mx[OA_T / 64];
    __shared__ uint32_t s_fl[OA_T / 64];
    const size_t base = (size_t)blockIdx.x * OA_TILE;
    uint64_t mn = ~0ull, mx = 0;
    uint32_t fl = 0;
#pragma unroll
    for (int i = 0; i < OA_I; ++i) {
        const size_t idx = base + (size_t)i * OA_T + threadIdx.x;
        if (idx < n) {
            const uint64_t k = keys[idx];
            mn = u64min(mn, k); mx = u64max(mx, k);
            if (idx > 0) {
                const uint64_t p = keys[idx - 1];
                if (k < p) fl |= ORD_FULL_DEC;
                if ((k >> 32) < (p >> 32)) fl |= ORD_HI_DEC;
                if ((uint32_t)k < (uint32_t)p) fl |= ORD_LO_DEC;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = u64min(mn, __shfl_xor(mn, off, 64));
        mx = u64max(mx, __shfl_xor(mx, off, 64));
        fl |= __shfl_xor(fl, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; s_fl[wave] = fl; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < OA_T / 64; ++w) { mn = u64min(mn, s_mn[w]); mx = u64max(mx, s_mx[w]); fl |= s_fl[w]; }
        tmin[blockIdx.x] = mn; tmax[blockIdx.x] = mx; tflags[blockIdx.x] = fl;
    }
}

// pm[t] = max over tiles before t, sm[t] = min over tiles after t (one workgroup; nt <= n/2048)
__global__ __launch_bounds__(1024) void oa_tile_prefix(const uint64_t *__restrict__ tmin,
                                                       const uint64_t *__restrict__ tmax,
                                                       const uint32_t *__restrict__ tflags, long nt,
                                                       uint64_t *__restrict__ pm, uint64_t *__restrict__ sm,
                                                       OrderInfo *__restrict__ info) {
    __shared__ uint64_t s[1024];
    const int t = threadIdx.x;
    uint32_t fl = 0;
    for (long i = t; i < nt; i += 1024) fl |= tflags[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) fl |= __shfl_xor(fl, off, 64);
    if ((t & 63) == 0 && fl) atomicOr(&info->flags, fl);
    uint64_t carry = 0;
    for (long base = 0; base < nt; base += 1024) {
        const long i = base + t;
        s[t] = i < nt ? tmax[i] : 0;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint64_t o = t >= off ? s[t - off] : 0;
            __syncthreads();
            s[t] = u64max(s[t], o);
            __syncthreads();
        }
        const uint64_t excl = t ? s[t - 1] : 0;
        if (i < nt) pm[i] = u64max(carry, excl);
        carry = u64max(carry, s[1023]);
        __syncthreads();
    }
    carry = ~0ull;
    for (long top = nt; top > 0; top -= 1024) {
        const long i = top - 1 - t;                   // thread 0 holds the LAST tile of this block of tiles
        s[t] = i >= 0 ? tmin[i] : ~0ull;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint64_t o = t >= off ? s[t - off] : ~0ull;
            __syncthreads();
            s[t] = u64min(s[t], o);
            __syncthreads();
        }
        const uint64_t excl = t ? s[t - 1] : ~0ull;
        if (i >= 0) sm[i] = u64min(carry, excl);
        carry = u64min(carry, s[1023]);
        __syncthreads();
    }
}

// cut[k] = 1 where max(keys[0,k)) <= min(keys[k,n)); thread t owns elements [8t, 8t+8) of the tile
__global__ __launch_bounds__(OA_T) void oa_cut_flags(const uint64_t *__restrict__ keys, size_t n,
                                                     const uint64_t *__restrict__ pm, const uint64_t *__restrict__ sm,
                                                     uint32_t *__restrict__ cut, uint32_t *__restrict__ head) {
    __shared__ uint64_t tile[OA_TILE + OA_T];
    __shared__ uint64_t s_mx[OA_T / 64], s_mn[OA_T / 64];
    const size_t base = (size_t)blockIdx.x * OA_TILE;
#pragma unroll
    for (int i = 0; i < OA_I; ++i) {
        const int e = i * OA_T + threadIdx.x;
        const size_t idx = base + e;
        tile[e + (e >> 3)] = idx < n ? keys[idx] : ~0ull;      // the tail never lowers a min; it follows every real key
    }
    __syncthreads();
    uint64_t v[OA_I], tmx = 0, tmn = ~0ull;
#pragma unroll
    for (int i = 0; i < OA_I; ++i) { v[i] = tile[threadIdx.x * 9 + i]; tmx = u64max(tmx, v[i]); tmn = u64min(tmn, v[i]); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t imx = tmx, imn = tmn;                  // inclusive prefix max / inclusive suffix min over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t a = __shfl_up(imx, off, 64), b = __shfl_down(imn, off, 64);
        if (lane >= off) imx = u64max(imx, a);
        if (lane + off < 64) imn = u64min(imn, b);
    }
    if (lane == 63) s_mx[wave] = imx;
    if (lane == 0) s_mn[wave] = imn;
    uint64_t emx = __shfl_up(imx, 1, 64), emn = __shfl_down(imn, 1, 64);
    if (lane == 0) emx = 0;
    if (lane == 63) emn = ~0ull;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < OA_T / 64; ++w) {
        if (w < wave) emx = u64max(emx, s_mx[w]);
        if (w > wave) emn = u64min(emn, s_mn[w]);
    }
    uint64_t before = u64max(pm[blockIdx.x], emx);
    uint64_t sfx[OA_I + 1];
    sfx[OA_I] = u64min(sm[blockIdx.x], emn);
#pragma unroll
    for (int i = OA_I - 1; i >= 0; --i) sfx[i] = u64min(v[i], sfx[i + 1]);
    // head[k] = cut[k] && !cut[k+1]: the first element of a piece of >= 2 elements.  cut[k+1] needs
    // max(keys[0..k]) and min(keys[k+1..n)), both at hand (sfx[i+1] starts at the next element).
#pragma unroll
    for (int i = 0; i < OA_I; ++i) {
        const size_t idx = base + (size_t)threadIdx.x * OA_I + i;
        const bool c = idx == 0 || before <= sfx[i];
        before = u64max(before, v[i]);
        const bool cnext = before <= sfx[i + 1];
        if (idx < n) {
            cut[idx] = c ? 1u : 0u;
            head[idx] = (c && idx + 1 < n && !cnext) ? 1u : 0u;
        }
    }
}

// Pieces of one element need no work; the others are listed as [wstart, wend).  A piece of >= 2
// elements has exactly one head (cut[k] && !cut[k+1]) and one end (the next cut, or n), in the same
// order, so the rank of an end is (#heads before it) - 1: one scan serves both.
__global__ void oa_work_list(const uint32_t *__restrict__ cut, const uint32_t *__restrict__ head,
                             const uint32_t *__restrict__ wpos, size_t n, const OrderInfo *__restrict__ info,
                             uint32_t *__restrict__ wstart, uint32_t *__restrict__ wend) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (head[k]) wstart[wpos[k]] = (uint32_t)k;
    if (k > 0 && cut[k] && !cut[k - 1]) wend[wpos[k] - 1] = (uint32_t)k;
    if (k == n - 1 && !cut[k]) wend[info->nchunks - 1] = (uint32_t)n;
}
// longest piece + the three size-class lists the sort kernels walk (order inside a list is irrelevant)
__global__ void oa_work_classify(const uint32_t *__restrict__ wstart, const uint32_t *__restrict__ wend,
                                 OrderInfo *__restrict__ info, uint32_t *__restrict__ small, uint32_t *__restrict__ mid,
                                 uint32_t *__restrict__ big) {
    const uint32_t nch = info->nchunks;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t m = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t c0 = blockIdx.x * blockDim.x; c0 < nch; c0 += stride) {      // whole waves iterate together
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t len = c < nch ? wend[c] - wstart[c] : 0u;
        m = max(m, len);
        const bool is_s = len > 1 && len <= (uint32_t)CS_SMALL, is_m = len > (uint32_t)CS_SMALL && len <= (uint32_t)CS_MID,
                   is_b = len > (uint32_t)CS_MID;
        const unsigned long long ms = __ballot(is_s), mm = __ballot(is_m), mb = __ballot(is_b);
        uint32_t bs = 0, bm = 0, bb = 0;
        if (lane == 0) {
            if (ms) bs = atomicAdd(&info->nsmall, (uint32_t)__popcll(ms));
            if (mm) bm = atomicAdd(&info->nmid, (uint32_t)__popcll(mm));
            if (mb) bb = atomicAdd(&info->nbig, (uint32_t)__popcll(mb));
        }
        bs = __shfl(bs, 0, 64); bm = __shfl(bm, 0, 64); bb = __shfl(bb, 0, 64);
        if (is_s) small[bs + (uint32_t)__popcll(ms & lt)] = c;
        if (is_m) mid[bm + (uint32_t)__popcll(mm & lt)] = c;
        if (is_b) big[bb + (uint32_t)__popcll(mb & lt)] = c;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor(m, off, 64));
    if (lane == 0 && m) atomicMax(&info->maxlen, m);
}

// One workgroup per piece with LO < len <= CAP: stable LSD radix sort in LDS of a permutation of the
// piece (keys stay put in LDS; only 16-bit positions move), 8-bit digits, skipping digits whose
// bits do not vary inside the piece (one atmosphere row => the whole high field).  Ranking is the
// wave-ballot scheme of rs_scatter.  Nothing happens at all when any piece exceeds CS_BIG.
template <int CAP, int LO, int THREADS>
__global__ __launch_bounds__(THREADS) void oa_chunk_sort(uint64_t *__restrict__ keys, uint32_t *__restrict__ idx,
                                                         const uint32_t *__restrict__ wstart,
                                                         const uint32_t *__restrict__ wend,
                                                         const uint32_t *__restrict__ list,
                                                         const OrderInfo *__restrict__ info) {
    constexpr int NW = THREADS / 64, ROWS = CAP / THREADS;
    extern __shared__ uint64_t cs_lds[];
    uint64_t *sk = cs_lds;                                                   // [CAP]
    uint16_t *perm0 = reinterpret_cast<uint16_t *>(cs_lds + CAP);            // [CAP]
    uint16_t *perm1 = perm0 + CAP;                                           // [CAP]
    uint32_t *cnt = reinterpret_cast<uint32_t *>(perm1 + CAP);               // [NW][256]
    uint32_t *s_wave = cnt + NW * 256;                                       // [NW]
    uint32_t *s_var = s_wave + NW;                                           // [2]
    if (info->maxlen > (uint32_t)CS_BIG) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const uint32_t nlist = LO >= CS_MID ? info->nbig : LO >= CS_SMALL ? info->nmid : info->nsmall;
    {
      for (uint32_t q = blockIdx.x; q < nlist; q += gridDim.x) {
        const uint32_t c = list[q];
        const uint32_t b = wstart[c], len = wend[c] - b;
        if (threadIdx.x < 2) s_var[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t k0 = keys[b];
        uint64_t var = 0;
        for (uint32_t i = threadIdx.x; i < len; i += THREADS) {
            const uint64_t k = keys[b + i];
            sk[i] = k; perm0[i] = (uint16_t)i;
            var |= k ^ k0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) var |= __shfl_xor(var, off, 64);
        if (lane == 0 && var) { atomicOr(&s_var[0], (uint32_t)var); atomicOr(&s_var[1], (uint32_t)(var >> 32)); }
        __syncthreads();
        var = ((uint64_t)s_var[1] << 32) | s_var[0];
        // wave w owns positions [w*per, (w+1)*per), walked in rows of 64: (wave, row, lane) order == sequence order
        const uint32_t per = ((len + NW * 64 - 1) / (NW * 64)) * 64;
        uint16_t *pin = perm0, *pout = perm1;
        for (int shift = 0; shift < 64; shift += 8) {
            const uint32_t m = (uint32_t)(var >> shift) & 0xffu;
            if (!m) continue;
            const int nbits = 32 - __clz(m);
            for (int i = threadIdx.x; i < NW * 256; i += THREADS) cnt[i] = 0;
            __syncthreads();
            volatile uint32_t *wc = cnt + wave * 256;
            uint32_t rank[ROWS], dig[ROWS];
            uint16_t pv[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const uint32_t pos = wave * per + r * 64 + lane;
                const bool ok = (uint32_t)(r * 64) < per && pos < len;
                const uint16_t p = ok ? pin[pos] : (uint16_t)0;
                const uint32_t d = ok ? (uint32_t)(sk[p] >> shift) & 0xffu : 0u;
                unsigned long long peers = __ballot(ok);
                for (int bb = 0; bb < nbits; ++bb) {
                    const bool bit = (d >> bb) & 1u;
                    const unsigned long long mm = __ballot(bit);
                    peers &= bit ? mm : ~mm;
                }
                uint32_t old = 0;
                const int leader = ok ? __ffsll((long long)peers) - 1 : 0;
                if (ok && lane == leader) { old = wc[d]; wc[d] = old + (uint32_t)__popcll(peers); }
                old = __shfl(old, leader, 64);
                rank[r] = old + (uint32_t)__popcll(peers & lt);
                dig[r] = d; pv[r] = p;
            }
            __syncthreads();
            uint32_t tot = 0;
            if (threadIdx.x < 256)
                for (int w = 0; w < NW; ++w) tot += cnt[w * 256 + threadIdx.x];
            uint32_t all;
            uint32_t run = block_excl_scan<THREADS>(threadIdx.x < 256 ? tot : 0u, s_wave, all);
            if (threadIdx.x < 256)
                for (int w = 0; w < NW; ++w) {
                    const uint32_t q = cnt[w * 256 + threadIdx.x];
                    cnt[w * 256 + threadIdx.x] = run;
                    run += q;
                }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const uint32_t pos = wave * per + r * 64 + lane;
                if ((uint32_t)(r * 64) < per && pos < len) pout[cnt[wave * 256 + dig[r]] + rank[r]] = pv[r];
            }
            __syncthreads();
            uint16_t *tmp = pin; pin = pout; pout = tmp;
        }
        uint32_t vi[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t i = r * THREADS + threadIdx.x;
            vi[r] = i < len ? idx[b + pin[i]] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t i = r * THREADS + threadIdx.x;
            if (i < len) { keys[b + i] = sk[pin[i]]; idx[b + i] = vi[r]; }
        }
        __syncthreads();
      }
    }
}
template <int CAP, int THREADS> constexpr size_t chunk_sort_lds() {
    return (size_t)CAP * 8 + (size_t)CAP * 4 + (size_t)(THREADS / 64) * 256 * 4 + (size_t)(THREADS / 64) * 4 + 8;
}

// Enqueues the analysis and the piece sorts.  Afterwards *d_info holds: flags (ORD_*), nchunks =
// number of pieces with >= 2 elements, maxlen = the longest of them.  The data is in (key, idx)
// order iff maxlen <= CS_BIG (8192); otherwise it is untouched.
void order_and_chunk_sort(uint64_t *keys, uint32_t *idx, size_t n, OrderInfo *d_info, hipStream_t st, bool try_pieces) {
    IBH_HIP(hipMemsetAsync(d_info, 0, sizeof(OrderInfo), st));
    if (n < 2) return;
    if (!try_pieces) {          // flags only (which fields ever decrease): the caller expects non-local disorder
        Arena &A = arena();
        const size_t nt = (n + OA_TILE - 1) / OA_TILE;
        uint64_t *tmin = A.get<uint64_t>(nt), *tmax = A.get<uint64_t>(nt), *pm = A.get<uint64_t>(nt), *sm = A.get<uint64_t>(nt);
        uint32_t *tflags = A.get<uint32_t>(nt);
        hipLaunchKernelGGL(oa_tiles, dim3((unsigned)nt), dim3(OA_T), 0, st, keys, n, tmin, tmax, tflags);
        hipLaunchKernelGGL(oa_tile_prefix, dim3(1), dim3(1024), 0, st, tmin, tmax, tflags, (long)nt, pm, sm, d_info);
        IBH_HIP(hipMemsetAsync(&d_info->maxlen, 0xFF, sizeof(uint32_t), st));     // "longer than any LDS piece"
        return;
    }
    IBH_CHECK(n < (1ul << 32), "sort too large");
    constexpr size_t lds_small = chunk_sort_lds<CS_SMALL, 256>(), lds_mid = chunk_sort_lds<CS_MID, 512>(),
                     lds_big = chunk_sort_lds<CS_BIG, 1024>();
    {   // once per device: the attribute belongs to the function's code object on that device
        static std::mutex mu;
        static std::map<int, bool> done;
        int dev = 0;
        IBH_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (!done[dev]) {
            IBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&oa_chunk_sort<CS_BIG, CS_MID, 1024>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
            done[dev] = true;
        }
    }
    Arena &A = arena();
    const size_t nt = (n + OA_TILE - 1) / OA_TILE;
    uint64_t *tmin = A.get<uint64_t>(nt), *tmax = A.get<uint64_t>(nt), *pm = A.get<uint64_t>(nt), *sm = A.get<uint64_t>(nt);
    uint32_t *cut = A.get<uint32_t>(n), *head = A.get<uint32_t>(n), *wpos = A.get<uint32_t>(n);
    uint32_t *wstart = A.get<uint32_t>(n / 2 + 1), *wend = A.get<uint32_t>(n / 2 + 1);
    const unsigned ge = (unsigned)((n + 255) / 256);
    uint32_t *tflags = A.get<uint32_t>(nt);
    hipLaunchKernelGGL(oa_tiles, dim3((unsigned)nt), dim3(OA_T), 0, st, keys, n, tmin, tmax, tflags);
    hipLaunchKernelGGL(oa_tile_prefix, dim3(1), dim3(1024), 0, st, tmin, tmax, tflags, (long)nt, pm, sm, d_info);
    hipLaunchKernelGGL(oa_cut_flags, dim3((unsigned)nt), dim3(OA_T), 0, st, keys, n, pm, sm, cut, head);
    exclusive_scan_u32(head, wpos, n, &d_info->nchunks, st);
    hipLaunchKernelGGL(oa_work_list, dim3(ge), dim3(256), 0, st, cut, head, wpos, n, d_info, wstart, wend);
    uint32_t *lsmall = A.get<uint32_t>(n / 2 + 1), *lmid = A.get<uint32_t>(n / CS_SMALL + 1), *lbig = A.get<uint32_t>(n / CS_MID + 1);
    const unsigned gmax = (unsigned)std::min<size_t>((n + 1023) / 1024, 1024);
    hipLaunchKernelGGL(oa_work_classify, dim3(gmax), dim3(256), 0, st, wstart, wend, d_info, lsmall, lmid, lbig);
    const unsigned gs = (unsigned)std::min<size_t>((n + 1) / 2, 256 * 12);
    hipLaunchKernelGGL((oa_chunk_sort<CS_SMALL, 1, 256>), dim3(gs), dim3(256), lds_small, st, keys, idx, wstart, wend, lsmall, d_info);
    if (n > (size_t)CS_SMALL) {
        const unsigned gm = (unsigned)std::min<size_t>(n / CS_SMALL, 1024);
        hipLaunchKernelGGL((oa_chunk_sort<CS_MID, CS_SMALL, 512>), dim3(gm), dim3(512), lds_mid, st, keys, idx, wstart, wend, lmid, d_info);
    }
    if (n > (size_t)CS_MID) {
        const unsigned gb = (unsigned)std::min<size_t>(n / CS_MID, 512);
        hipLaunchKernelGGL((oa_chunk_sort<CS_BIG, CS_MID, 1024>), dim3(gb), dim3(1024), lds_big, st, keys, idx, wstart, wend, lbig, d_info);
    }
    IBH_HIP(hipGetLastError());
}

bool order_is_final(const OrderInfo &h) { return !(h.flags & ORD_FULL_DEC) || h.maxlen <= (uint32_t)CS_BIG; }

bool radix_after_analysis(const OrderInfo &h, uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt,
                          size_t n, int lo_bits, int hi_bits, hipStream_t st) {
    KeyField f[2]; int nf = 0;
    if (lo_bits > 0 && (h.flags & ORD_LO_DEC)) f[nf++] = KeyField{0, lo_bits};        // else: stable sort by the high field suffices
    if (hi_bits > 0) f[nf++] = KeyField{32, hi_bits};
    return radix_sort_pairs(keys, keys_alt, vals, vals_alt, n, f, nf, st);
}

bool adaptive_sort_pairs(uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t n,
                         int lo_bits, int hi_bits, OrderInfo *d_info, hipStream_t st) {
    if (n < 2) return false;
    order_and_chunk_sort(keys, vals, n, d_info, st);
    OrderInfo h;
    readback_sync(&h, d_info, sizeof(h), st);
    order_debug(h, n, lo_bits, hi_bits);
    if (order_is_final(h)) return false;                                              // in order / pieces sorted in place
    return radix_after_analysis(h, keys, keys_alt, vals, vals_alt, n, lo_bits, hi_bits, st);
}

void order_debug(const OrderInfo &h, size_t n, int lo_bits, int hi_bits) {
    static const bool dbg = getenv("IBH_DEBUG_SORT") != nullptr;
    if (dbg) fprintf(stderr, "[ibh sort] n=%zu flags=%u pieces=%u maxlen=%u small=%u mid=%u big=%u lo_bits=%d hi_bits=%d\n", n, h.flags, h.nchunks, h.maxlen, h.nsmall, h.nmid, h.nbig, lo_bits, hi_bits);
}

}  // namespace ibh
