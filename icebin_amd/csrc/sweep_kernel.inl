// sweep_kernel.inl -- K1d: column-sweep apply for matrices with short columns (EvI / EvX: an ice cell is a column of
// the two elevation classes it lies between; AvI / AvX: of one GCM cell).
//
// Row by row an E-row matrix reads every X element once per class row it feeds (measured at 1 km: 2.0 GB fetched for
// 1.03 GB algorithmic); the band structure (rowdual) still wastes 64-byte lines (1.39 x).  Here ALL columns are swept
// ONCE, in ascending order.  The structure (assemble.hip build_sweep_from_csr): entries ordered by column and paired
// into ITEMS (a column + <= 2 of its entries), 64 items a block, tb blocks a task; a task's local row table = the distinct
// rows its columns touch ("slots").  Per block:
//   load     the four waves of a workgroup bring the block's X tile (64 planes x 64 columns) in with plain loads, lane =
//            column -- one instruction = one field plane x 512 contiguous bytes for consecutive columns, whatever the
//            alignment of the planes (odd leading dimensions) -- one block ahead, and park it in LDS;
//   consume  transposed: the LANES ARE THE FIELDS (of one batch, or -- fewer than 33 fields -- of 64/nfl batches of a batched
//            launch), wave w takes columns [16w, 16w+16) of the block.  Which rows a column feeds is then WAVE-UNIFORM
//            (scalar metadata, broadcast weights), so the accumulators need no cross-lane reduction at all: one live pair
//            of rows in two registers, spilled to a per-wave LDS table acc[slot][lane] when the sweep moves on to another
//            pair (neighbouring ice cells lie between the same two classes for long runs).
// A task leaves one partial sum per (row it touches, lane); sweep_combine_kernel (spmm.hip) adds the partial rows of a row
// in task order.  Fixed order throughout -> bitwise reproducible; entries are predicated by existence, never multiplied by
// a padded zero (0*NaN must not leak into a row).  Design notes and measurements: DESIGN.md K1d.
#pragma once
#include <hip/hip_runtime.h>

#ifndef IBH_MAX_BATCH
#define IBH_MAX_BATCH 32
#endif

namespace ibh {

constexpr int SWEEP_CB = 64;             // columns per block
constexpr int SWEEP_NW = 4;              // waves per workgroup
constexpr int SWEEP_TS = SWEEP_CB + 1;   // padded tile row
// One ITEM per column (a column with more than two entries: several): the column, and <= 2 of its entries:
// meta = slot0 | slot1 << 8 | has0 << 16 | has1 << 17 (0: empty slot), v0, v1 the two values.  Items are packed: block b =
// items [64 b, 64 b + 64) (the last block of the matrix may be short), task t = blocks [t tb, t tb + tb): no descriptors.
constexpr unsigned SWEEP_HAS0 = 1u << 16, SWEEP_HAS1 = 1u << 17;
struct SweepView {
    const int *task_p0;        // [ntask]   first partial-sum row of the task
    const int *task_ns;        // [ntask]   rows ("slots") the task's columns touch
    const int *it_col;         // [nblk*64] (not read when every column has exactly one item: column = item index)
    const unsigned *it_meta;   // [nblk*64]
    const double *it_v0, *it_v1;
    int tb, nblk, nitems;
};
struct SweepBatch {
    const double *x[IBH_MAX_BATCH];
    double *p[IBH_MAX_BATCH];  // partial sums [nprow][ldp]
};

inline size_t sweep_lds_bytes(int nslot) { return ((size_t)64 * SWEEP_TS + (size_t)SWEEP_NW * nslot * 64 + 2 * SWEEP_CB) * 8; }

__device__ __forceinline__ int sweep_xcd_contiguous(int b, int nb) {
    const int q = nb >> 3, rem = nb & 7;
    const int x = b & 7, slot = b >> 3;
    const int base = x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q;
    return base + slot;
}

typedef const unsigned __attribute__((address_space(4))) *sweep_cu32;
typedef double sweep_double2 __attribute__((ext_vector_type(2)));

// FULL: all 64 planes of the wave row exist (64 | nvar, or nvar == nfl and a whole number of batch groups).  MODE (diagnostics, scratch/sweep_bench.hip): 0 the kernel; 1 tile traffic only;
// 2 the kernel with s_memtime stamps around its phases (per wave sums to sb.p[1])
// IDENT: every column has exactly one item (AvI, AvX: <= 2 entries per column), so item index = column and the column list
// is never read -- the first tile loads of a task depend on nothing but the kernel arguments.
template <bool FULL, bool IDENT, int MODE = 0>
__global__ __launch_bounds__(SWEEP_NW * 64) void spmm_sweep_kernel(const SweepView sv, const SweepBatch sb, long ldx, int nf,
                                                                   int nslot_max, long ldp, int lg, int nbatch)
{
    constexpr int NW = SWEEP_NW, TS = SWEEP_TS, FPW = 64 / NW, CPW = SWEEP_CB / NW;   // planes loaded / columns consumed per wave
    extern __shared__ double s_mem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *tile = s_mem;                                           // [64 fields][TS]
    double *acc = s_mem + 64 * TS + wave * (nslot_max * 64);        // [slots][64 fields] per wave
    const int t = sweep_xcd_contiguous(blockIdx.x, gridDim.x);
    // The 64 lanes are (batch, field) pairs: nfl = 1 << lg fields of 64 / nfl field batches (lg = 6: the 64 fields of field
    // block blockIdx.y of ONE batch).  Fewer than 33 fields per batch would leave lanes idle; the batches of a batched launch
    // sweep the same columns and feed the same rows, so they share a wave: plane p = (batch z*G + (p >> lg), field p & (nfl-1)).
    const int z = blockIdx.z, nfl = 1 << lg, G = 64 >> lg;
    const int b0 = t * sv.tb, b1 = min(sv.nblk, b0 + sv.tb);
    const int p0 = sv.task_p0[t], ns = min(sv.task_ns[t], nslot_max);        // needed at the very end only
    for (int s = 0; s < nslot_max; ++s) acc[s * 64 + lane] = 0.0;
    auto plane_field = [&](int p) { return lg == 6 ? (int)blockIdx.y * 64 + p : (p & (nfl - 1)); };
    auto plane_batch = [&](int p) { return lg == 6 ? z : z * G + (p >> lg); };
    const bool lane_live = plane_field(lane) < nf && plane_batch(lane) < nbatch;     // this lane's (batch, field) exists
    // this wave loads planes [16 wave, 16 wave + 16): two halves of eight (a half never straddles two batches: nfl >= 8), one
    // buffer descriptor per half based at the half's first plane; planes that do not exist repeat the last one that does
    __amdgpu_buffer_rsrc_t rs[2];
    unsigned poff[FPW];                                             // byte offsets of the planes from their half's first
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pf = wave * FPW + 8 * h;
        const int bq = FULL ? plane_batch(pf) : min(plane_batch(pf), nbatch - 1);
        const int f0 = FULL ? plane_field(pf) : min(plane_field(pf), nf - 1);
        rs[h] = __builtin_amdgcn_make_buffer_rsrc((void *)(sb.x[bq] + (long)f0 * ldx), 0, -1 /* 4 GB */, 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = FULL ? plane_field(pf + i) : min(plane_field(pf + i), nf - 1);
            poff[8 * h + i] = (unsigned)((long)(f - f0) * ldx * 8);
        }
    }

    int cur = -1;                         // meta word of the live pair
    double a0 = 0.0, a1 = 0.0;
    auto spill = [&]() {
        if (cur >= 0) {
            acc[(cur & 255) * 64 + lane] += a0;
            if (cur & (int)SWEEP_HAS1) acc[((cur >> 8) & 255) * 64 + lane] += a1;
        }
    };
    // One register tile: the loads of block k+1 (X tile and the weights of the wave's 16 columns, lanes 0..15) are in flight
    // while block k is consumed (two tiles in flight -- every load unconditional so that the stores of block k wait for the
    // oldest tile only -- were measured SLOWER, 239 against 228 us at 1 km: the kernel is bound by the address/issue work of
    // its loads and the extra tail loads and the empty step of odd tasks cost more than the latency they hide).  What the consumer needs per column must not cost a VALU -> SGPR transfer (v_readlane:
    // measured, five of them per column made the consume phase as long as the memory latency it is supposed to hide) nor a
    // load inside the consume phase (a serial round trip per block):
    //   * the 16 metadata words come through ONE scalar load, issued after the previous block has been consumed -- it
    //     lands while the wave waits for the tile;
    //   * the weights are parked in LDS next to the tile and read back with wave-uniform addresses (a broadcast).
    // Block sizes travel 64 at a time, one per lane (a descriptor load inside the block loop would be a dependent round
    // trip in front of every tile load).
    unsigned long tA = 0, tB = 0, tC = 0, tD = 0, dA = 0, dB = 0, dC = 0, dD = 0, dn = 0;    // MODE 2: phase clocks
    double xr[FPW], v0 = 0.0, v1 = 0.0;
    sweep_double2 *wq = reinterpret_cast<sweep_double2 *>(s_mem + 64 * TS + NW * nslot_max * 64) + wave * CPW;    // [CPW] (w0, w1) per wave
    const int jb = wave * CPW;                              // this wave consumes columns [jb, jb + CPW) of a block
    {
        const int kc = b0, nbk = b1 - b0;
        unsigned mk[CPW];
        auto count = [&](int k) { return min(SWEEP_CB, sv.nitems - (kc + k) * SWEEP_CB); };      // items of block k: 64 but for the matrix' last block
        // the column list of a block is fetched one block ahead of its tile loads (two ahead of its consumption)
        auto columns = [&](int k) {
            const int kk = min(k, nbk - 1);
            const int it = (kc + kk) * SWEEP_CB + min(lane, count(kk) - 1);       // lanes past the last item repeat it
            return IDENT ? it : sv.it_col[it];
        };
        int cnx = columns(0);
        auto issue = [&](int k) {
            // Unconditional loads (exact vmcnt bookkeeping); planes past nf repeat the last plane.
            // buffer loads: the wave's first plane is the descriptor base, the plane is a scalar offset, the column a 32-bit
            // per-lane byte offset -- no 64-bit VALU address arithmetic per load (the launch checks 16 * ldx * 8 < 2^32)
            const int boff = cnx << 3;
            cnx = columns(k + 1);
#pragma unroll
            for (int i = 0; i < FPW; ++i)
                xr[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs[i >> 3], boff, (int)poff[i], 0));
            const long it = (long)(kc + k) * SWEEP_CB + jb + (lane & (CPW - 1));
            v0 = sv.it_v0[it]; v1 = sv.it_v1[it];
        };
        auto meta = [&](int k) {
            sweep_cu32 mp = (sweep_cu32)(sv.it_meta + (long)(kc + k) * SWEEP_CB + jb);
#pragma unroll
            for (int j = 0; j < CPW; ++j) mk[j] = mp[j];
        };
        issue(0);
        meta(0);
        for (int k = 0; k < nbk; ++k) {
            if (MODE == 2) tA = __builtin_amdgcn_s_memtime();
            const bool mine = jb < count(k);                                 // some of this wave's column slots hold items
#pragma unroll
            for (int i = 0; i < FPW; ++i) tile[(wave * FPW + i) * TS + lane] = xr[i];
            { sweep_double2 w; w.x = v0; w.y = v1; wq[lane & (CPW - 1)] = w; }
            __syncthreads();
            if (MODE == 2) { tB = __builtin_amdgcn_s_memtime(); dA += tB - tA; }
            if (k + 1 < nbk) issue(k + 1);
            if (MODE == 2) { tC = __builtin_amdgcn_s_memtime(); dB += tC - tB; }
            if (mine) {
                if (MODE == 1) {
                    double x[CPW];
#pragma unroll
                    for (int j = 0; j < CPW; ++j) x[j] = tile[lane * TS + jb + j];
#pragma unroll
                    for (int j = 0; j < CPW; ++j) a0 += x[j];
                    cur = (int)SWEEP_HAS0;
                } else {
                    // eight columns at a time: their tile values and weight pairs up front -- one LDS round trip per half, not
                    // one per column; sixteen at a time cost 48 more VGPRs (152 -> 104) and measured 3 % slower
#pragma unroll
                    for (int h = 0; h < CPW; h += 8) {
                        double x[8];
                        sweep_double2 w[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) { x[j] = tile[lane * TS + jb + h + j]; w[j] = wq[h + j]; }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (__builtin_expect((int)mk[h + j] != cur, 0)) {   // not the live pair of rows
                                if (mk[h + j] == 0) continue;                   // empty slot
                                spill();                                        // the sweep moves on to another pair
                                cur = (int)mk[h + j]; a0 = 0.0; a1 = 0.0;
                            }
                            // a column with one entry carries w1 = 0 and no HAS1: a1 then collects garbage (0*NaN) that spill()
                            // never adds
                            a0 = fma(w[j].x, x[j], a0);
                            a1 = fma(w[j].y, x[j], a1);
                        }
                    }
                }
            }
            if (MODE == 2) { tD = __builtin_amdgcn_s_memtime(); dC += tD - tC; }
            if (k + 1 < nbk) meta(k + 1);
            __syncthreads();
            if (MODE == 2) { dD += __builtin_amdgcn_s_memtime() - tD; ++dn; }
        }
    }
    spill();
    if (MODE == 2 && lane == 0 && sb.p[1]) {
        unsigned long *dbg = reinterpret_cast<unsigned long *>(sb.p[1]) + ((long)t * NW + wave) * 8;
        dbg[0] = dA; dbg[1] = dB; dbg[2] = dC; dbg[3] = dD; dbg[4] = dn;
    }
    double *__restrict__ P = sb.p[z];
    const int fld = (lg == 6 ? (int)blockIdx.y * 64 : 0) + lane;    // this lane's column of the partial-sum rows
    __syncthreads();
    // the task's partial sums: the waves' tables added in wave order, slots dealt round-robin to the waves
    const double *acc0 = s_mem + 64 * TS;
    for (int s = wave; s < ns; s += NW) {
        double tot = acc0[s * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) tot += acc0[w * (nslot_max * 64) + s * 64 + lane];
        if (lane_live) P[(long)(p0 + s) * ldp + fld] = tot;
    }
}

}  // namespace ibh
