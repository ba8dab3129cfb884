// spmm.hip -- K1: CSR x dense-field products for Weighted::apply on gfx950.
//
// Replaces the Eigen SparseMatrix<ColMajor> * Dense product inside
// ibmisc::linear::Weighted_Eigen::apply (call sites modele/merge_topo.cpp:65,
// modele/icebin22m.cpp:153; inline products IceCoupler.cpp:237,445).
//
// Fields are field-major (X[f*ldx + col], Y[f*ldy + row]): one field is one
// contiguous vector, as blitz::Array<double,2>(nvar, n) is (icebin22m.cpp:142).
// fp64, HBM-bound (<= 0.25 flop/byte): no MFMA.  Two shapes, two kernels:
//
//  * rowblock  -- few long rows (AvI/EvI/AvE direction: 10^2..10^5 rows of
//    10^1..10^4 nnz).  One 256-thread workgroup per (row, chunk of fields).
//    The row segment (colind, vals) is staged once in LDS by all four waves
//    with coalesced loads, then every wave streams its own fields: lane k reads
//    X[f, col[k]], so a wave-load is 64 consecutive-ish doubles of one field
//    (dense column ids of one GCM cell are near-contiguous by construction of
//    the first-seen numbering).  Per-lane accumulators, one butterfly reduction
//    per (row, field) at the end, no atomics -> bitwise reproducible.
//    blockIdx is remapped so that the field-chunks of one row land on the same
//    XCD and share the row segment through that XCD's L2.
//
//  * shortrow  -- millions of rows of 1-3 nnz (IvA/IvE direction).  One thread
//    per row, looping over a chunk of fields with the row's entries held in
//    registers; lanes own consecutive rows, so the dominant traffic -- the
//    Y stores -- is fully coalesced (512 B per wave-store, non-temporal); the
//    small X (nf x nA_d) is served from L2.
#include "common.h"
#include "sweep_kernel.inl"
#include <hip/hip_ext.h>
#include <mutex>

namespace ibh {

// ---- tuning knobs (ibh_set_tuning) ---------------------------------------------------------
// Process-wide and read on every apply while another host thread may be calling ibh_set_tuning:
// guarded by a mutex (an uncontended lock is ~20 ns against a >= 3 us launch).
namespace {
struct Tuning {
    std::mutex mu;
    std::unordered_map<std::string, int> map;
};
Tuning &tuning() { static Tuning *t = new Tuning; return *t; }
}  // namespace
// the matrix whose apply is being set up on this thread: its own options (ibh_weighted_set_option) win over the process-wide map
static thread_local const ibh_weighted *tl_handle = nullptr;
struct HandleScope {
    const ibh_weighted *prev;
    explicit HandleScope(const ibh_weighted *w) : prev(tl_handle) { tl_handle = w; }
    ~HandleScope() { tl_handle = prev; }
};
int get_tuning(const char *key, int dflt) {
    if (tl_handle && !tl_handle->opts.empty()) {
        auto it = tl_handle->opts.find(key);
        if (it != tl_handle->opts.end()) return it->second;
    }
    Tuning &t = tuning();
    std::lock_guard<std::mutex> lk(t.mu);
    if (t.map.empty()) return dflt;
    auto it = t.map.find(key);
    return it == t.map.end() ? dflt : it->second;
}
void set_tuning(const char *key, int value) {
    Tuning &t = tuning();
    std::lock_guard<std::mutex> lk(t.mu);
    if (value == INT32_MIN) t.map.erase(key);       // back to the built-in default
    else t.map[key] = value;
}

// ---- launch timing (ibh_set_launch_events) ---------------------------------------------------------
// A caller that wants the duration of the SpMM kernel ITSELF (bench.py's roofline figure: the kernel's
// launch duration, not the host's submission latency in front of it) hands in a pair of HIP events; the
// next rowblock launch of this thread attaches them to its own dispatch (hipExtLaunchKernel: start =
// kernel begins, stop = kernel ends -- the interval rocprofv3's kernel trace reports).  One-shot.
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
// chain (ibh_weighted_apply_chain_device): the next I-row launch of this thread waits on a device counter instead of the queue
static thread_local unsigned *g_chain_cnt = nullptr;
static thread_local unsigned g_chain_n = 0;
void set_launch_events(hipEvent_t start, hipEvent_t stop) { g_ev_start = start; g_ev_stop = stop; }

// ---- helpers -------------------------------------------------------------------------------
// Sum over the 64 lanes of a wave, result valid in every lane.  DPP moves instead of ds_bpermute:
// the shuffle form is six dependent trips through the LDS crossbar (0.6 us measured at the end of
// every workgroup of the headline kernel, on its critical path); this is ~25 VALU instructions.
// Fixed combination order -> bitwise reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]: pairs
    v += dpp_move<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]: quads
    v += dpp_move<0x141, 0xf>(v);    // row_half_mirror: 8 lanes
    v += dpp_move<0x140, 0xf>(v);    // row_mirror: 16 lanes (every lane of a row holds the row sum)
    v += dpp_move<0x142, 0xa>(v);    // row_bcast15 into rows 1,3: lanes 16-31 = rows 0+1, lanes 48-63 = rows 2+3
    v += dpp_move<0x143, 0xc>(v);    // row_bcast31 into rows 2,3: lanes 48-63 = total
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).  Give each
// XCD a contiguous range of logical ids, so neighbours in logical order share an L2.
__device__ __forceinline__ int xcd_contiguous(int b, int nb) {
    const int q = nb >> 3, rem = nb & 7;
    const int x = b & 7, slot = b >> 3;
    const int base = x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q;
    return base + slot;
}

// blockIdx -> (row, field chunk).  Workgroups are dealt round-robin over the 8 XCDs, each with its
// own L2.  Two rows that share X lines (an ice cell straddling two GCM cells is a column of both)
// read them for the SAME field chunk, so:
//   mode 1 (nfc a multiple of 8, or 1/2/4): an XCD owns whole field chunks and runs every row for
//           them -- no X line is fetched by two XCDs, all XCDs do identical work; the (small) CSR
//           is read by every XCD and served from the memory-side Infinity Cache after the first.
//   mode 0: contiguous ranges of (row, chunk) pairs per XCD (rows of one chunk share the staged
//           CSR segment through L2); used when the chunk count does not split over 8 XCDs.
__device__ __forceinline__ bool block_to_task(int b, int nrow, int nfc, int mode, int &r, int &fc) {
    if (mode == 1) {
        const int x = b & 7, slot = b >> 3;
        if (nfc >= 8) {                       // nfc % 8 == 0
            const int per = nfc >> 3;
            r = slot / per;
            fc = x + 8 * (slot - r * per);
        } else {                              // nfc in {1,2,4}: 8/nfc XCDs share one chunk, split by row range
            const int m = 8 / nfc;
            fc = x % nfc;
            const int part = x / nfc;
            const int r0 = (int)((long)nrow * part / m), r1 = (int)((long)nrow * (part + 1) / m);
            r = r0 + slot;
            if (r >= r1) return false;
        }
        return r < nrow;
    }
    const int logical = xcd_contiguous(b, (int)gridDim.x);     // gridDim.x = nrow*nfc rounded up to a multiple of 8
    if (logical >= nrow * nfc) return false;
    r = logical / nfc;
    fc = logical - r * nfc;
    return true;
}

constexpr int RB_SEG = 1024;    // nnz staged per pass: 12 KB of LDS

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// One field's X row through a buffer descriptor: 32-bit per-lane byte offset, SGPR base.  The
// same voffset (col << 3) serves every field of the wave, so a gathered element costs one
// buffer_load + one FMA instead of a 64-bit address computation per load; at the 40 MB headline
// size the kernel is as much instruction-issue- as bandwidth-limited.
__device__ __forceinline__ double xload(__amdgpu_buffer_rsrc_t rs, int byte_off) {
#ifndef IBH_X_AUX
#define IBH_X_AUX 0
#endif
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, IBH_X_AUX));
}

// Field batches of one launch (ibh_weighted_apply_many_device): batch q reads x[q] and writes y[q],
// all with the same nvar / leading dimensions.  Passed by value in the kernarg segment, so a launch
// needs no device-side pointer table and stays stream-ordered and graph-capturable.
struct BatchPtrs {
    const double *x[IBH_MAX_BATCH];
    double *y[IBH_MAX_BATCH];
};

// FPW = fields per wave, WK = waves that split the nnz range of the row, NW = waves per workgroup.
// The NW waves form WF = NW/WK groups over fields; a block covers FB = FPW*WF fields and stages
// the row segment once for all of them.
//
// Batched launches (nbatch > 1): blockIdx.y = batch group; a workgroup serves `qi` consecutive
// batches of its (row, field chunk) with the row segment staged ONCE (rows of <= RB_SEG entries).
// A 40 MB apply is a latency-sized problem on this chip (launch + rowptr -> segment -> X are
// ~4 us of an 11 us launch); in one launch over several batches the workgroups of batch q+1 start
// while those of batch q still stream, so the prologue and the drain are paid once per LAUNCH.
//
// DUAL (EvI, EvX; WK == 1): the "row" is a BAND -- the ice cells of one GCM cell whose elevation lies
// between two neighbouring classes.  Such a cell is a column of two rows of M (weights 1-r and r on
// the lower and the upper class); as a band entry it carries both values (vals, vals2; bits 30/31 of
// the column say which exist) and is read ONCE, feeding two accumulators.  Row by row every X element
// of these matrices is fetched twice (measured: 2.0 GB for 1.03 GB algorithmic at 1 km).  The kernel
// then stores the two partial sums of the band (Y = lower-class sums, Y2 = upper-class sums, no
// fill); dual_combine_kernel adds, for every row, its own lower sum and the upper sum of the band
// below it.  The band arrays are a filtered copy of the CSR (assemble.hip build_bands).
template <int FPW, int WK, int UNROLL, int NW, bool DUAL = false>
__global__ __launch_bounds__(NW * 64, (FPW == 1 && !DUAL) ? 8 : 1) void spmm_rowblock_kernel(
    const int *__restrict__ rowptr, const int *__restrict__ colind, const double *__restrict__ vals,
    const BatchPtrs bp, int nbatch, int qi, long ldx, int ncol, long ldy, int nrow, int nf, int nfc,
    int xcd_mode, const double *__restrict__ wM, double fill,
    const double *__restrict__ vals2 = nullptr, double *__restrict__ Y2 = nullptr,
    const int *__restrict__ rowperm = nullptr, long y2_stride = 0)
{
    static_assert(!DUAL || WK == 1, "bands are not split across waves");
    constexpr int RB_THREADS = NW * 64;
    constexpr int RB_STAGE = RB_SEG / RB_THREADS;
    constexpr int WF = NW / WK;
    constexpr int FB = FPW * WF;
    constexpr int STEP = WK * 64;
    constexpr int BATCH = UNROLL * STEP;
    __shared__ int s_col[RB_SEG];
    __shared__ double s_val[RB_SEG];
    __shared__ double s_val2[DUAL ? RB_SEG : 1];
    __shared__ double s_part[WK][FB];

    int r, fc;
    if (!block_to_task(blockIdx.x, nrow, nfc, xcd_mode, r, fc)) return;
    if (rowperm) r = rowperm[r];        // longest rows first: the launch ends on short workgroups
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // provably wave-uniform
    const int wf = wave % WF, wk = wave / WF;
    const int f0 = fc * FB + wf * FPW;
    const int q0 = blockIdx.y * qi, q1 = min(nbatch, q0 + qi);

    const int beg = rowptr[r], end = rowptr[r + 1];
    const bool one_seg = end - beg <= RB_SEG;
    const bool dead = DUAL ? false : wM[r] == 0.0;     // mask_result, IceCoupler.cpp:186-201
    // The row is walked in segments of RB_SEG entries staged in LDS.  The loads of segment s+1 are
    // issued before segment s is processed (they fly while X streams) and written to LDS after it.
    int cc[RB_STAGE];
    double vv[RB_STAGE], vv2[DUAL ? RB_STAGE : 1];
    auto stage_load = [&](int seg, int n) {       // all loads first (clamped, unconditional)
#pragma unroll
        for (int i = 0; i < RB_STAGE; ++i) {
            const int k = min((int)threadIdx.x + i * RB_THREADS, n - 1);
            cc[i] = colind[seg + k];
            vv[i] = vals[seg + k];
            if (DUAL) vv2[i] = vals2[seg + k];
        }
    };
    auto stage_store = [&](int n) {
#pragma unroll
        for (int i = 0; i < RB_STAGE; ++i) {
            const int k = threadIdx.x + i * RB_THREADS;
            if (k < n) { s_col[k] = cc[i]; s_val[k] = vv[i]; if (DUAL) s_val2[k] = vv2[i]; }
        }
    };

    for (int q = q0; q < q1; ++q) {
        const double *__restrict__ X = bp.x[q];
        double *__restrict__ Y = bp.y[q];
        __amdgpu_buffer_rsrc_t rs[FPW];
#pragma unroll
        for (int j = 0; j < FPW; ++j) {
            const int f = f0 + j < nf ? f0 + j : nf - 1;    // clamp: tail fields read valid memory, never stored
            rs[j] = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (long)f * ldx), 0, ncol * 8, 0x00020000);
        }
        double acc[FPW], acc2[DUAL ? FPW : 1];
#pragma unroll
        for (int j = 0; j < FPW; ++j) { acc[j] = 0.0; if (DUAL) acc2[j] = 0.0; }

        if (beg < end && (q == q0 || !one_seg)) {       // a row of one segment stays staged for every batch
            stage_load(beg, min(RB_SEG, end - beg));
            if (q != q0) __syncthreads();               // every wave is done with the previous batch's last segment
            stage_store(min(RB_SEG, end - beg));
            __syncthreads();
        }
        for (int seg = beg; seg < end; seg += RB_SEG) {
            const int n = min(RB_SEG, end - seg);
            const int nxt = seg + RB_SEG;
            const bool more = nxt < end;
            if (more) stage_load(nxt, min(RB_SEG, end - nxt));
            // Full batches: every lane issues UNROLL*FPW loads before the first FMA, no predication.
            // The weights are read from LDS when the gathered values are consumed, not before the loads are issued: a slot in
            // flight then holds only its x (2 VGPRs per field), so that 14-16 loads per lane fit the 64 registers at which all
            // workgroups of a single 64-field apply are resident at once (8 waves per SIMD).
            const int nfull = n - n % BATCH;
            int kb = 0;
            if constexpr (!DUAL) {
                for (; kb < nfull; kb += BATCH) {
                    double x[FPW][UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const int off = s_col[kb + wk * 64 + lane + u * STEP] << 3;
#pragma unroll
                        for (int j = 0; j < FPW; ++j) x[j][u] = xload(rs[j], off);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const double v = s_val[kb + wk * 64 + lane + u * STEP];
#pragma unroll
                        for (int j = 0; j < FPW; ++j) acc[j] = fma(v, x[j][u], acc[j]);
                    }
                }
                // Tail batch: lanes past the end re-read the last entry and are masked at the FMA (never
                // multiplied by 0: 0*NaN must not leak into a row).
                if (kb + wk * 64 < n) {
                    double x[FPW][UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const int k = kb + wk * 64 + lane + u * STEP;
                        const int off = s_col[k < n ? k : n - 1] << 3;
#pragma unroll
                        for (int j = 0; j < FPW; ++j) x[j][u] = xload(rs[j], off);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const int k = kb + wk * 64 + lane + u * STEP;
                        const bool ok = k < n;
                        const double v = s_val[ok ? k : n - 1];
#pragma unroll
                        for (int j = 0; j < FPW; ++j) acc[j] = ok ? fma(v, x[j][u], acc[j]) : acc[j];
                    }
                }
            } else {
            for (; kb < nfull; kb += BATCH) {
                int off[UNROLL];
                double v[UNROLL], v2[UNROLL];
                bool h0[UNROLL], h1[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int k = kb + wk * 64 + lane + u * STEP;
                    const int c = s_col[k];
                    off[u] = (c & 0x3fffffff) << 3;
                    v[u] = s_val[k];
                    v2[u] = s_val2[k]; h0[u] = (c >> 30) & 1; h1[u] = c < 0;
                }
                double x[FPW][UNROLL];
#pragma unroll
                for (int j = 0; j < FPW; ++j)
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) x[j][u] = xload(rs[j], off[u]);
#pragma unroll
                for (int j = 0; j < FPW; ++j)
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {     // a missing partner is never multiplied (0*NaN)
                        acc[j] = h0[u] ? fma(v[u], x[j][u], acc[j]) : acc[j];
                        acc2[j] = h1[u] ? fma(v2[u], x[j][u], acc2[j]) : acc2[j];
                    }
            }
            if (kb + wk * 64 < n) {
                int off[UNROLL];
                double v[UNROLL], v2[UNROLL];
                bool ok[UNROLL], ok2[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int k = kb + wk * 64 + lane + u * STEP;
                    ok[u] = k < n;
                    const int kk = ok[u] ? k : n - 1;
                    const int c = s_col[kk];
                    off[u] = (c & 0x3fffffff) << 3;
                    v[u] = s_val[kk];
                    v2[u] = s_val2[kk]; ok2[u] = ok[u] && c < 0; ok[u] = ok[u] && ((c >> 30) & 1);
                }
                double x[FPW][UNROLL];
#pragma unroll
                for (int j = 0; j < FPW; ++j)
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) x[j][u] = xload(rs[j], off[u]);
#pragma unroll
                for (int j = 0; j < FPW; ++j)
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        acc[j] = ok[u] ? fma(v[u], x[j][u], acc[j]) : acc[j];
                        acc2[j] = ok2[u] ? fma(v2[u], x[j][u], acc2[j]) : acc2[j];
                    }
            }
            }
            if (more) {
                __syncthreads();                      // every wave is done reading this segment
                stage_store(min(RB_SEG, end - nxt));
                __syncthreads();
            }
        }
#pragma unroll
        for (int j = 0; j < FPW; ++j) acc[j] = wave_sum(acc[j]);
        if (DUAL) {                          // partial sums of the band; fill and the pairing happen in dual_combine_kernel
#pragma unroll
            for (int j = 0; j < FPW; ++j) acc2[j] = wave_sum(acc2[j]);
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < FPW; ++j)
                    if (f0 + j < nf) { Y[(long)(f0 + j) * ldy + r] = acc[j]; Y2[(long)q * y2_stride + (long)(f0 + j) * ldy + r] = acc2[j]; }
            }
            continue;
        }
        if (WK == 1) {
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < FPW; ++j)
                    if (f0 + j < nf) Y[(long)(f0 + j) * ldy + r] = dead ? fill : acc[j];
            }
        } else {
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < FPW; ++j) s_part[wk][wf * FPW + j] = acc[j];
            }
            __syncthreads();
            if (threadIdx.x < FB) {
                const int f = fc * FB + threadIdx.x;
                double s = s_part[0][threadIdx.x];
#pragma unroll
                for (int qq = 1; qq < WK; ++qq) s += s_part[qq][threadIdx.x];
                if (f < nf) Y[(long)f * ldy + r] = dead ? fill : s;
            }
            if (q + 1 < q1) __syncthreads();           // s_part is reused by the next batch
        }
    }
}

// ---- rowone: ONE apply per launch, one field per wave -- the lean form of rowblock for the reference's own call shape ----------
// (Weighted_Eigen::apply, one call per matrix: merge_topo.cpp:65, icebin22m.cpp:153).  Same staging, same per-lane order,
// same wave_sum as rowblock with FPW = WK = 1 -> the same bits; what is gone is everything a single launch does not need: the
// batch loop and its pointer table in the kernel arguments, the cross-segment prefetch registers, the full / tail batch split.
// A row of <= U*64 entries (every row of the 5 km matrices) is ONE staged segment and ONE batch of U gathers per lane; the
// weights are read from LDS when the gathers land, so a slot in flight holds two registers and U = 14..16 fits the 64
// registers at which all workgroups of a 64-field apply are resident at once.  Measured against rowblock on the headline
// shape (scratch/chain_bench.hip has the decomposition: empty launch 4.1 us, matrix-free stream 7.9, this chain 9.8).
template <int NW, int U>
__global__ __launch_bounds__(NW * 64, 8) void spmm_rowone_kernel(
    const int *__restrict__ rowptr, const int *__restrict__ colind, const double *__restrict__ vals,
    const double *__restrict__ X, double *__restrict__ Y, long ldx, int ncol, long ldy, int nrow, int nf, int nfc,
    int xcd_mode, const double *__restrict__ wM, double fill)
{
    constexpr int T = NW * 64, SEG = U * 64, ST = (SEG + T - 1) / T;
    __shared__ int s_col[SEG];
    __shared__ double s_val[SEG];
    int r, fc;
    if (!block_to_task(blockIdx.x, nrow, nfc, xcd_mode, r, fc)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fw = fc * NW + wave;
    const int f = fw < nf ? fw : nf - 1;            // clamp: tail fields read valid memory, never stored
    const int beg = rowptr[r], end = rowptr[r + 1];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (long)f * ldx), 0, ncol * 8, 0x00020000);
    double acc = 0.0;
    for (int seg = beg; seg < end; seg += SEG) {
        const int n = min(SEG, end - seg);
        int cc[ST];
        double vv[ST];
#pragma unroll
        for (int i = 0; i < ST; ++i) {              // all loads first (clamped, unconditional)
            const int k = min((int)threadIdx.x + i * T, n - 1);
            cc[i] = colind[seg + k];
            vv[i] = vals[seg + k];
        }
        if (seg != beg) __syncthreads();            // every wave is done with the previous segment
#pragma unroll
        for (int i = 0; i < ST; ++i) {
            const int k = threadIdx.x + i * T;
            if (k < n) { s_col[k] = cc[i]; s_val[k] = vv[i]; }
        }
        __syncthreads();
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = lane + 64 * u;
            x[u] = xload(rs, s_col[k < n ? k : n - 1] << 3);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {               // lanes past the end are masked, never multiplied by 0 (0*NaN)
            const int k = lane + 64 * u;
            const bool ok = k < n;
            const double v = s_val[ok ? k : n - 1];
            acc = ok ? fma(v, x[u], acc) : acc;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0 && fw < nf) Y[(long)fw * ldy + r] = wM[r] == 0.0 ? fill : acc;      // mask_result, IceCoupler.cpp:186-201
}
// Tried on top of this and dropped (measured on MI355X, 5 km AvI, 64 fields): RUN HINTS -- the first-seen numbering makes a
// row's columns a contiguous run behind ~18 scattered ones, so the gathers of the run were issued with arithmetic addresses
// together with the staging loads -- bought nothing (10.44 against 10.48 us): the FMA chain starts with chunk 0, whose gather
// still waits for the staged columns and then queues behind every other load of the CU.  In-kernel stamps
// (scratch/chain_bench.hip, CHAIN_TIMELINE) show why: the X phase is throughput-bound -- a wave needs a median 3.5 us just to
// ISSUE its 14 gathers because the memory pipeline is full -- so what a single launch can still gain is its 1.2 us prologue.
// A staging-free variant ("rowdirect": a wave owns (row, 4 fields) and reads colind / vals itself, all chunks of the row in
// flight at once, row pointers in the kernel arguments, no LDS, no barrier) was SLOWER at every size (5 km 12.6 against
// 11.7 us, 1 km 235 against 186): 1.5 x the vector-memory instructions per gathered element, and row pointers in the kernarg
// segment are a memory round trip like any other (12.64 against 12.60 us with / without).

// ---- rowgroup: E-row matrices (EvI, EvX), one workgroup per (GCM cell, chunk of fields) ------------------------------------------
// An ice cell between two elevation classes is a column of TWO rows of M -- (GCM cell a, class k) and (a, k+1) -- so row by row
// every X element is gathered twice, and each of the ~10 class rows of a GCM cell touches ~15 % of the cell's footprint: most
// of every line it fetches belongs to other classes (5 km EvI: ~100 MB effective for a 42 MB apply).  Here the rows of one GCM
// cell form a GROUP whose entries are listed once per distinct column, ascending, as items {col, slot0, slot1, v0, v1}
// (assemble.hip build_groups_from_csr): the column set of a group is the AvI row of its GCM cell -- a contiguous run behind a
// few cells shared with earlier GCM cells -- so the gathers are those of the A-row kernel: every X element once, whole lines.
// Lane = column, wave = field (as rowone): lane L multiplies its column's value into the sums of the column's two classes,
// kept in a per-wave LDS table acc[slot][lane] (bank = lane: conflict-free whatever the slots); at the end one wave_sum per
// class.  A group holds ALL entries of its rows, so the result is final: no partial sums, no second kernel.  Fixed order
// throughout (per lane: items lane, lane + 64, ... ascending; then the wave sum) -> bitwise reproducible; entries are
// predicated by existence, never multiplied by a padded zero (0*NaN).
struct GroupView {
    const int *ptr, *ns, *slotrow, *col;
    const unsigned *meta;
    const double *v0, *v1;
    int ngrp, nslot;
};
constexpr unsigned RG_HAS0 = 1u << 16, RG_HAS1 = 1u << 17;
// PAIR: a second matrix applied to this one's result in the epilogue (ibh_weighted_apply_pair_device; see ibh_weighted::pair_*)
struct PairView {
    const double *w;
    const unsigned *mask;
    const int *row;
    const double *wM2;
    double *Y2;
    long ldy2;
    unsigned *done;     // chain (ibh_weighted_apply_chain_device): every workgroup counts itself here when its results are visible
};
// a workgroup of the first kernel of a chain is finished: its stores are released to the device, then it counts itself
// Chain counters {c, flag}: every workgroup of the first kernel clears `flag` when it starts (the kernel before it in the queue --
// the previous chain's last kernel -- has completed: a barrier packet) and counts itself at `c` when its results are out; the
// one that counts last re-arms c and raises the flag.  The last kernel's workgroups only READ the flag: same-address
// read-modify-writes go through one L2 channel one after the other (1 200 waiting workgroups counting themselves cost 20 us).
// (the results the next kernel reads were stored with chain_store -- written through to the device-coherent level -- so no
// release fence is needed, which at agent scope writes back the whole L2 of the XCD: every wave waits for its own stores, the
// barrier collects the waves)
__device__ __forceinline__ void chain_begin(unsigned *cnt) {
    if (threadIdx.x == 0) __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_signal(unsigned *cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ void chain_store(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a value the kernel before this one wrote (and released at agent scope): a load that is coherent across the XCDs' L2s
__device__ __forceinline__ double chain_load(const double *p) {
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
#ifndef CHAIN_SLEEP
#define CHAIN_SLEEP 32
#endif
// the first thing a workgroup of the chain's last kernel does: wait until the kernel before it has raised the flag (that kernel
// was dispatched first, so its workgroups are placed first and always finish)
__device__ __forceinline__ void chain_wait(unsigned *cnt, unsigned) {
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(CHAIN_SLEEP);
    }
    __syncthreads();
}
template <int NW, int U, int TW, bool PAIR = false>
__global__ __launch_bounds__(NW * 64) void spmm_rowgroup_kernel(const GroupView gv, const BatchPtrs bp, long ldx, int ncol, long ldy, int nf,
                                                               int nfc, int xcd_mode, const double *__restrict__ wM, double fill,
                                                               const PairView pv = PairView{})
{
    constexpr int T = NW * 64, SEG = U * 64, ST = SEG / T;
    static_assert(SEG % T == 0, "segment is a multiple of the workgroup");
    extern __shared__ double rg_lds[];
    double *s_v0 = rg_lds, *s_v1 = rg_lds + SEG;
    int *s_col = reinterpret_cast<int *>(rg_lds + 2 * SEG);
    unsigned *s_meta = reinterpret_cast<unsigned *>(s_col + SEG);
    double *s_tab = rg_lds + 3 * SEG;                       // [NW][nslot][64]
    int g, fc;
    if (PAIR && pv.done) chain_begin(pv.done);
    if (!block_to_task(blockIdx.x, gv.ngrp, nfc, xcd_mode, g, fc)) {
        if (PAIR && pv.done) chain_signal(pv.done);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fw = fc * NW + wave;
    const int f = fw < nf ? fw : nf - 1;            // clamp: tail fields read valid memory, never stored
    const double *__restrict__ X = bp.x[blockIdx.y];
    double *__restrict__ Y = bp.y[blockIdx.y];
    const int beg = gv.ptr[g], end = gv.ptr[g + 1], ns = gv.ns[g];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (long)f * ldx), 0, ncol * 8, 0x00020000);
    // TW = 32: lanes L and L + 32 share a table entry (half the LDS: all workgroups of a 64-field apply at 5 km are resident
    // at once instead of in two rounds); their adds are issued as two instructions, lower half first -- LDS operations of a wave
    // execute in issue order, so the sum of an entry is still formed in one fixed order
    double *tab = s_tab + (long)wave * gv.nslot * TW + (lane & (TW - 1));
    for (int s = 0; s < ns; ++s) tab[s * TW] = 0.0;
    // The group is walked in segments of SEG items staged in LDS; the loads of segment s+1 are issued before segment s is
    // processed (they fly while X streams) and written to LDS after it.
    int cc[ST];
    unsigned mm[ST];
    double a0[ST], a1[ST];
    auto stage_load = [&](int seg, int n) {         // all loads first (clamped, unconditional)
#pragma unroll
        for (int i = 0; i < ST; ++i) {
            const int k = seg + min((int)threadIdx.x + i * T, n - 1);
            cc[i] = gv.col[k]; mm[i] = gv.meta[k]; a0[i] = gv.v0[k]; a1[i] = gv.v1[k];
        }
    };
    auto stage_store = [&](int n) {
#pragma unroll
        for (int i = 0; i < ST; ++i) {
            const int k = threadIdx.x + i * T;
            if (k < n) { s_col[k] = cc[i]; s_meta[k] = mm[i]; s_v0[k] = a0[i]; s_v1[k] = a1[i]; }
        }
    };
    if (beg < end) {
        stage_load(beg, min(SEG, end - beg));
        stage_store(min(SEG, end - beg));
        __syncthreads();
    }
    for (int seg = beg; seg < end; seg += SEG) {
        const int n = min(SEG, end - seg);
        const int nxt = seg + SEG;
        const bool more = nxt < end;
        if (more) stage_load(nxt, min(SEG, end - nxt));
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = lane + 64 * u;
            x[u] = xload(rs, s_col[k < n ? k : n - 1] << 3);
        }
        // the items' slots and weights come out of LDS while the gathers fly ...
        unsigned m[U];
        double w0[U], w1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = lane + 64 * u;
            const int kk = k < n ? k : n - 1;
            m[u] = k < n ? s_meta[kk] : 0u;             // lanes past the end: no has-bit
            w0[u] = s_v0[kk]; w1[u] = s_v1[kk];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ... and the products go to the LDS pipe as ds_add_f64 WITHOUT return, back to back: the sums of a lane are added in
        // program order (deterministic) and the wave never waits for a read-modify-write round trip
        if (TW == 64) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (m[u] & RG_HAS0) __hip_atomic_fetch_add(tab + (m[u] & 0xffu) * TW, w0[u] * x[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (m[u] & RG_HAS1) __hip_atomic_fetch_add(tab + ((m[u] >> 8) & 0xffu) * TW, w1[u] * x[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double p0 = w0[u] * x[u], p1 = w1[u] * x[u];
                const bool lo = lane < 32;
                if ((m[u] & RG_HAS0) && lo) __hip_atomic_fetch_add(tab + (m[u] & 0xffu) * TW, p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_sched_barrier(0);
                if ((m[u] & RG_HAS0) && !lo) __hip_atomic_fetch_add(tab + (m[u] & 0xffu) * TW, p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_sched_barrier(0);
                if ((m[u] & RG_HAS1) && lo) __hip_atomic_fetch_add(tab + ((m[u] >> 8) & 0xffu) * TW, p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_sched_barrier(0);
                if ((m[u] & RG_HAS1) && !lo) __hip_atomic_fetch_add(tab + ((m[u] >> 8) & 0xffu) * TW, p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            __syncthreads();                        // every wave is done reading this segment
            stage_store(min(SEG, end - nxt));
            __syncthreads();
        }
    }
    // class sums: lane L adds one half (32 lanes' worth) of slot L % 32, starting at a lane-dependent offset (bank = 2 x index:
    // two lanes per bank instead of all 64), then the two halves meet; fixed order per lane -> reproducible
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const int sl = lane & 31, half = lane >> 5;
    double tot = 0.0;
    if (sl < ns) {
        constexpr int HW = TW / 2;
        const double *row = s_tab + ((long)wave * gv.nslot + sl) * TW + half * HW;
#pragma unroll 8
        for (int j = 0; j < HW; ++j) tot += row[(j + lane) & (HW - 1)];
    }
    tot += __shfl_xor(tot, 32, 64);
    double y1 = 0.0;
    if (half == 0 && sl < ns) {
        const int r = gv.slotrow[g * IBH_GSLOTS + sl];
        y1 = wM[r] == 0.0 ? fill : tot;                         // mask_result, IceCoupler.cpp:186-201
        if (fw < nf) Y[(long)fw * ldy + r] = y1;
    }
    if (PAIR) {
        // the second matrix's row of this group: lane s holds weight x (the value just stored for slot s), summed over the wave
        // in the fixed order of wave_sum -- entry k of the row in lane k, as the row kernel holds it when the second matrix's
        // columns are this group's rows in slot order
        const int a2 = pv.row[g];
        if (a2 >= 0) {
            double v = 0.0;
            if (half == 0 && sl < ns && ((pv.mask[g] >> sl) & 1u)) v = pv.w[g * IBH_GSLOTS + sl] * y1;
            const double s2 = wave_sum(v);
            if (lane == 0 && fw < nf) {
                const double o2 = pv.wM2[a2] == 0.0 ? fill : s2;
                if (pv.done) chain_store(&pv.Y2[(long)fw * pv.ldy2 + a2], o2); else pv.Y2[(long)fw * pv.ldy2 + a2] = o2;
            }
        }
        if (pv.done) chain_signal(pv.done);
    }
}

// Measured on MI355X (us per apply, one launch / 16 per launch; rowblock in brackets): 5 km EvI 64 fields 17.1 / 14.8 (23.0 / 16.5),
// 16 fields 7.5 / 3.9 (8.8 / 4.9); 1 km EvI 64 fields 223 / 204 (colsweep 231 / 203).  The read-modify-write of the class table
// was the first bottleneck (32.5 us at 5 km with plain LDS loads and stores: a dependent round trip per item); a variant without
// staging (every wave reads its items from global memory, LDS = tables only, no barrier) lost at every size (21.4 / 16.3 us at
// 5 km, 322 at 1 km): as for the A-row kernel, extra vector-memory instructions cost more than barriers and LDS footprint.
// Two fields per wave (the items read once per 16 fields instead of 8: 1.45 -> 1.22 x traffic at 1 km) doubles the class tables:
// 250 against 217 us at 1 km with 8 waves, 216 with 4 (no gain), 20.3 against 17.1 us at 5 km.
// A lean form for groups that fit one staged segment (U = 12..16 gathers per lane, slots and weights read from LDS when the
// gathers land, no cross-segment prefetch -- what made rowone fast) does NOT carry over: 17.5 us at best (U = 14, half-width
// tables), 24-29 with full tables (two workgroups per CU).  What bounds this kernel is the LDS pipe, not the gathers: per item
// and field it moves ~56 bytes through LDS (col, meta, two weights, two read-modify-write adds) where the A-row kernel moves 12.

// ---- grouptile: the row groups with X staged through LDS and register sums (round 5) ---------------------------------------------
// What bounded the row-group kernel above was the LDS pipe: wave = field, lane = column, so every wave re-read each item's
// {col, meta, v0, v1} and issued two ds_add_f64 per item (~56 B through LDS per item and field).  Here the roles are turned:
// the columns of a group are cut into TILES of SEG = 256 or 128 items (assemble.hip build_group_tiles); a workgroup serves (group, F
// fields, F = 16 or 32) and for every tile
//   1. gathers X[f, col[k]] for the tile's items -- lane = item, one field per load, exactly the gathers of the kernel above:
//      every X element once per GCM cell, whole lines -- into an LDS tile s_x[f][k];
//   2. walks the tile's ENTRIES, sorted by (slot, item) = the rows of the CSR restricted to the tile, in QUADS of four entries
//      of one slot (the lists are padded to fours with {zero column, weight 0}): lane = (sublane j, field f), J = 64 / F sublanes;
//      a sublane reads its entry of the quad -- {8 k, v}, two LDS reads broadcast within the sublane -- then x = s_x[f][k] (one
//      conflict-free ds_read_b64) and does one FMA into a REGISTER: the slot of a quad is wave-uniform.  ~20 B through LDS per
//      entry and field, no atomics, no read-modify-write, no predicate.
// Which wave sums what is static, so the sums stay in registers across tiles: every slot's list of a tile is cut into SP parts
// (SP = the largest power of two with SP * ns <= 32, by the group's number of rows ns -- a GCM cell of two classes keeps all
// waves busy as one of sixteen does); the units u = s * SP + part, in this order, are dealt to the waves in contiguous, balanced
// ranges -- consecutive units are consecutive quads of the tile's list, so a wave walks ONE run of quads per tile and only
// switches its accumulator on the way.
// Summation order of (row, field): entry e of a part's run of the tile goes to partial sum e mod 4, tiles ascending; per part
// (p0 + p1) + (p2 + p3); parts ascending -- fixed by the matrix alone, so one apply, a batch, every field count and both F
// agree bitwise.
struct TileView {
    const int *gptr, *tptr, *tcol;
    const unsigned short *ek, *eptr;
    const double *ev;
    const int *ns, *slotrow;
    int ngrp;
    long long *dbg;     // scratch/r05 timeline builds only (GT_TIMELINE): 16 stamps per workgroup
};
#ifdef GT_TIMELINE
#define GT_STAMP(i) do { if (tv.dbg && threadIdx.x == 0 && blockIdx.y == 0) tv.dbg[(long)blockIdx.x * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#define GT_STAMP_W(i) do { if (tv.dbg && lane == 0 && blockIdx.y == 0) tv.dbg[(long)blockIdx.x * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GT_STAMP(i) do {} while (0)
#define GT_STAMP_W(i) do {} while (0)
#endif
#ifndef GT_WPS
#define GT_WPS 4
#endif
#ifndef GT_ULIMIT
#define GT_ULIMIT 16
#endif
#ifndef GT_BATCH
#define GT_BATCH 4
#endif
constexpr int GT_TABP = 65;          // row stride of the epilogue's table of partial sums [unit][lane]
template <int F, int SEG>
constexpr size_t grouptile_lds() { return (size_t)F * (SEG + 2) * 8 + ibh_gt_ecap(SEG) * 10 + 256; }     // (+ slack: a batch reads a few steps past its run)
// sums of the two 32-lane halves of a wave: lo = lanes 0..31, hi = lanes 32..63, each in the order wave_sum adds a wave whose
// other half is zero
__device__ __forceinline__ void wave_half_sums(double v, double &lo, double &hi) {
    v += dpp_move<0xB1, 0xf>(v);
    v += dpp_move<0x4E, 0xf>(v);
    v += dpp_move<0x141, 0xf>(v);
    v += dpp_move<0x140, 0xf>(v);
    v += dpp_move<0x142, 0xa>(v);    // row_bcast15 into rows 1,3: lane 31 = rows 0+1, lane 63 = rows 2+3
    lo = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 31), __builtin_amdgcn_readlane(__double2loint(v), 31));
    hi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <int F, int NS, int SEG, int NW, bool PAIR = false>
__global__ __launch_bounds__(NW * 64, GT_WPS) void spmm_grouptile_kernel(const TileView tv, const BatchPtrs bp, long ldx, int ncol, long ldy, int nf,
                                                                            int nfc, int xcd_mode, const double *__restrict__ wM, double fill,
                                                                            const PairView pv = PairView{})
{
    constexpr int NB = SEG / 64, T = NW * 64, SEGP = SEG + 2, ECAP = ibh_gt_ecap(SEG);
    constexpr int FW = F / (NW / NB);                   // fields a wave gathers for its 64 items
    constexpr int EPT = (ECAP + T - 1) / T;      // entries staged per thread
    constexpr int J = 64 / F, R = 4 / J;                // sublanes; entries of a quad per sublane = partial sums per lane and unit
    constexpr int OS = ((NS > GT_ULIMIT ? NS : GT_ULIMIT) + NW - 1) / NW;     // most units (slot, part) of a wave: groups of <= NS rows
    static_assert(F == 16 || F == 32, "fields per workgroup");
    static_assert(NW % NB == 0 && F % (NW / NB) == 0, "shape");
    extern __shared__ double gt_lds[];
    double *s_x = gt_lds;                                                // [F][SEGP]; columns SEG, SEG + 1 stay zero
    double *s_ev = s_x + F * SEGP;                                       // [ECAP]
    unsigned short *s_ek = reinterpret_cast<unsigned short *>(s_ev + ECAP);      // [ECAP]
    double *s_tab = gt_lds;                                              // epilogue: [R][32 units][GT_TABP], over s_x
    int g, fc;
    if (PAIR && pv.done) chain_begin(pv.done);
    if (!block_to_task(blockIdx.x, tv.ngrp, nfc, xcd_mode, g, fc)) {
        if (PAIR && pv.done) chain_signal(pv.done);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    GT_STAMP(0);
    const double *__restrict__ X = bp.x[blockIdx.y];
    double *__restrict__ Y = bp.y[blockIdx.y];
    const int t0 = tv.tptr[g], t1 = tv.tptr[g + 1], ns = tv.ns[g];
    const int nitems = tv.gptr[g + 1] - tv.gptr[g];
    GT_STAMP(1 + 0 * (t0 + t1 + ns + nitems));
    const int ib = wave % NB, fg = wave / NB;           // this wave gathers items [64 ib, 64 ib + 64) for fields [FW fg, FW fg + FW)
    const int j = lane / F, f = lane % F;
    // parts per slot: the largest power of two with sp * ns <= 32; this wave's units [ub, ue)
    constexpr int UL = GT_ULIMIT;                       // most units of a group
    static_assert(UL == 32 || UL == 16, "units");
    const int lsp = ns * 16 <= UL ? 4 : ns * 8 <= UL ? 3 : ns * 4 <= UL ? 2 : ns * 2 <= UL ? 1 : 0;
    const int nunit = ns << lsp;
    const int ub = (wave * nunit) / NW, ue = ((wave + 1) * nunit) / NW;
    if (threadIdx.x < 2 * F) s_x[(threadIdx.x >> 1) * SEGP + SEG + (threadIdx.x & 1)] = 0.0;

    double x[FW];
    double evv[EPT];
    unsigned short ekk[EPT];
    // The HEAD of a tile -- the columns of this wave's 64 items and the tile's row of entry offsets (one per slot, lane i of every
    // wave holds offset i) -- is fetched one tile BEFORE its X values and entries are: inside the loop a tile costs one memory
    // round trip (its gathers, issued with addresses already in registers), not two, and the unit boundaries come out of
    // registers (v_readlane) instead of LDS.
    int c_nxt = 0, ep_nxt = 0;
    auto tile_n = [&](int t) { return min(SEG, nitems - (t - t0) * SEG); };
    auto load_head = [&](int t) {
        if (ib * 64 < tile_n(t)) c_nxt = tv.tcol[(long)t * SEG + ib * 64 + lane];
        ep_nxt = tv.eptr[(long)t * IBH_GT_EP + min(lane, IBH_GT_EP - 1)];
    };
    auto load_tile = [&](int t, int c, int ne) {        // into registers; flies while the previous tile is summed
        if (ib * 64 < tile_n(t)) {
            const int off = c << 3;
#pragma unroll
            for (int i = 0; i < FW; ++i) {
                const int fw = fc * F + fg * FW + i;
                const __amdgpu_buffer_rsrc_t rs =
                    __builtin_amdgcn_make_buffer_rsrc((void *)(X + (long)(fw < nf ? fw : nf - 1) * ldx), 0, ncol * 8, 0x00020000);
                x[i] = xload(rs, off);
            }
        }
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = threadIdx.x + i * T;
            if (e < ne) { ekk[i] = tv.ek[(long)t * ECAP + e]; evv[i] = tv.ev[(long)t * ECAP + e]; }
        }
    };
    auto store_tile = [&](int t) {
        if (ib * 64 < tile_n(t)) {
#pragma unroll
            for (int i = 0; i < FW; ++i) s_x[(fg * FW + i) * SEGP + ib * 64 + lane] = x[i];
        }
#pragma unroll
        for (int i = 0; i < EPT; ++i) {                 // (beyond the tile's entries: stale values nobody uses)
            const int e = threadIdx.x + i * T;
            if (e < ECAP) { s_ek[e] = ekk[i]; s_ev[e] = evv[i]; }
        }
    };

    double acc[OS][R];
#pragma unroll
    for (int o = 0; o < OS; ++o)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[o][r] = 0.0;
    // per lane: field f's row of the X tile; this sublane's entry of step 0 (weights); the entry whose item a lane fetches for a batch
    const char *a_x = reinterpret_cast<const char *>(s_x) + f * SEGP * 8;
    const char *a_ev = reinterpret_cast<const char *>(s_ev) + j * 8;
    const char *a_ek = reinterpret_cast<const char *>(s_ek) + (J * (lane & 15) + j) * 2;

    int ep_cur = 0;
    if (t0 < t1) {
        load_head(t0);
        const int c0 = c_nxt;
        ep_cur = ep_nxt;
        if (t0 + 1 < t1) load_head(t0 + 1);
        load_tile(t0, c0, __builtin_amdgcn_readlane(ep_cur, IBH_GSLOTS));
    }
    for (int t = t0; t < t1; ++t) {
        if (t != t0) __syncthreads();                   // every wave is done with the previous tile
        if (t == t0) GT_STAMP(2); else if (t == t0 + 1) GT_STAMP(6);
        store_tile(t);
        if (t == t0) GT_STAMP(3); else if (t == t0 + 1) GT_STAMP(7);
        __syncthreads();
        if (t == t0) GT_STAMP(4); else if (t == t0 + 1) GT_STAMP(8);
        const int ep = ep_cur;
        if (t + 1 < t1) {
            const int c = c_nxt;
            ep_cur = ep_nxt;
            if (t + 2 < t1) load_head(t + 2);
            load_tile(t + 1, c, __builtin_amdgcn_readlane(ep_cur, IBH_GSLOTS));
        }
        __builtin_amdgcn_sched_barrier(0);              // (the loads above fly while the tile is summed: nothing of them sinks below)
        if (t == t0) GT_STAMP(13);
        // Unit o of this wave = steps [B[o], B[o + 1]) of the tile's list (a step = J entries x F fields), walked in batches of
        // 8 steps: lane 16 r + i fetches the item index 8 k of the entry that step i of the batch hands to sublane r / (F / 16) --
        // one LDS read per lane and batch; a DPP row broadcast folded into the address add then gives every step its X
        // address, the weights come as broadcast reads at fixed offsets, and the 16 reads of a batch are in flight together.
        int B[OS + 1];
#pragma unroll
        for (int o = 0; o <= OS; ++o) {
            const int u = min(ub + o, ue);              // (o past this wave's units: the end of its run)
            const int uu = min(u, nunit - 1);
            const int sl = uu >> lsp, part = u >= nunit ? (1 << lsp) : (u & ((1 << lsp) - 1));
            const int lb = __builtin_amdgcn_readlane(ep, sl) >> 2, nq = (__builtin_amdgcn_readlane(ep, sl + 1) >> 2) - lb;
            B[o] = (lb + ((nq * part) >> lsp)) * R;
        }
        if (t == t0) { GT_STAMP(14); if (tv.dbg && threadIdx.x == 0 && blockIdx.y == 0) tv.dbg[(long)blockIdx.x * 16 + 15] = ((long long)(B[OS] - B[0]) << 32) | (unsigned)(B[1] - B[0]); }
#pragma unroll
        for (int o = 0; o < OS; ++o) {
            for (int p0 = B[o]; p0 < B[o + 1]; p0 += GT_BATCH) {
                const int nb = B[o + 1] - p0;
                const int mk = *reinterpret_cast<const unsigned short *>(a_ek + p0 * J * 2);
                const char *pv = a_ev + p0 * J * 8;
                double xx[GT_BATCH], vv[GT_BATCH];
#define GT_XR(S)                                                                                                                          \
                xx[S] = *reinterpret_cast<const double *>(a_x + __builtin_amdgcn_update_dpp(0, mk, 0x150 + (S), 0xf, 0xf, false));        \
                vv[S] = *reinterpret_cast<const double *>(pv + (S) * J * 8)
#define GT_FM(S) if ((S) < nb) acc[o][(S) % R] = fma(vv[S], xx[S], acc[o][(S) % R])
                GT_XR(0); GT_XR(1); GT_XR(2); GT_XR(3);
#if GT_BATCH == 8
                if (nb > 4) { GT_XR(4); GT_XR(5); GT_XR(6); GT_XR(7); }
#endif
                GT_FM(0); GT_FM(1); GT_FM(2); GT_FM(3);
#if GT_BATCH == 8
                if (nb > 4) { GT_FM(4); GT_FM(5); GT_FM(6); GT_FM(7); }
#endif
#undef GT_XR
#undef GT_FM
            }
        }
        if (t == t0) GT_STAMP(5); else if (t == t0 + 1) GT_STAMP(9);
    }
    GT_STAMP(10);
    __syncthreads();                                    // s_x is done: the partial sums take its place
    GT_STAMP(11);
#pragma unroll
    for (int o = 0; o < OS; ++o)
        if (ub + o < ue) {
#pragma unroll
            for (int r = 0; r < R; ++r) s_tab[(r * 32 + ub + o) * GT_TABP + lane] = acc[o][r];
        }
    __syncthreads();
    // thread -> (field, slot): lanes 0..31 of a wave are the slots of one field, lanes 32..63 those of the next
    const int sl = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < F / (2 * NW); ++i) {
        const int fl = 2 * (wave * (F / (2 * NW)) + i) + half, fw = fc * F + fl;
        double y1 = 0.0;
        if (sl < ns) {
            double tot = 0.0;
            for (int p = 0; p < (1 << lsp); ++p) {
                const double *row = s_tab + ((sl << lsp) + p) * GT_TABP + fl;
                double pp;
                if constexpr (J == 4) pp = (row[0] + row[16]) + (row[32] + row[48]);
                else pp = (row[0] + row[32]) + (row[32 * GT_TABP] + row[32 * GT_TABP + 32]);
                tot = p == 0 ? pp : tot + pp;
            }
            const int r = tv.slotrow[g * IBH_GSLOTS + sl];
            y1 = wM[r] == 0.0 ? fill : tot;             // mask_result, IceCoupler.cpp:186-201
            if (fw < nf) Y[(long)fw * ldy + r] = y1;
        }
        if (PAIR) {                                     // the second matrix's row of this group, as in spmm_rowgroup_kernel
            const int a2 = pv.row[g];
            if (a2 >= 0) {
                double v2 = 0.0;
                if (sl < ns && ((pv.mask[g] >> sl) & 1u)) v2 = pv.w[g * IBH_GSLOTS + sl] * y1;
                double s_lo, s_hi;
                wave_half_sums(v2, s_lo, s_hi);
                const double s2 = half ? s_hi : s_lo;
                if (sl == 0 && fw < nf) {
                    const double o2 = pv.wM2[a2] == 0.0 ? fill : s2;
                    if (pv.done) chain_store(&pv.Y2[(long)fw * pv.ldy2 + a2], o2); else pv.Y2[(long)fw * pv.ldy2 + a2] = o2;
                }
            }
        }
    }
    GT_STAMP(12);
    if (PAIR && pv.done) chain_signal(pv.done);
}

constexpr int SR_THREADS = 256;

// XT[c*ldt + f] = X[f*ldx + c]; pad columns f in [nf, ldt) are zeroed (read for tail fields, never stored)
__global__ void transpose_fields_kernel(const BatchPtrs bp, long ldx, int nf, int ncol,
                                        double *__restrict__ XT0, int ldt, long xt_stride) {
    __shared__ double tile[16][65];
    const double *__restrict__ X = bp.x[blockIdx.z];
    double *__restrict__ XT = XT0 + (long)blockIdx.z * xt_stride;
    const int cb = blockIdx.x * 64, fb = blockIdx.y * 16;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 256 threads: 64 columns x 4
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f = fb + ty * 4 + q, c = cb + tx;
        tile[ty * 4 + q][tx] = (f < nf && c < ncol) ? X[(long)f * ldx + c] : 0.0;
    }
    __syncthreads();
    const int fo = threadIdx.x & 15, co = threadIdx.x >> 4;          // 16 fields x 16 columns per pass
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cb + co + q * 16;
        if (c < ncol && fb + fo < ldt) XT[(long)c * ldt + fb + fo] = tile[fo][co + q * 16];
    }
}

// REALIGN: the result planes Y[f, :] do not start on 64-byte lines (odd leading dimension: the
// reference's A_b(nvar, n) is contiguous, so ldy = nrow_d).  A wave's 512-byte store would then
// straddle nine lines and leave two of them partially written, to be completed by the neighbouring
// wave: measured 4.0 instead of 6.0 TB/s of store bandwidth on this chip.  The workgroup therefore
// computes 256 rows but OWNS, per plane, the line-aligned run of <= 255 rows that starts at the first
// 64-byte boundary at or after its first row (consecutive workgroups advance by 248 rows, so the
// runs tile the plane exactly); values cross lanes through LDS and every wave stores whole lines.
constexpr int SR_STEP = SR_THREADS - 8;
// XT: X is read from its transpose XT[c, f] (row stride ldx = fields padded to 16): the G fields of
// one entry are 8*G contiguous bytes per lane, fetched with 16-byte loads.  With the field-major X a
// wave gather touches up to 64 lines PER FIELD when the lanes' columns differ (IvE: the elevation
// class changes from cell to cell); transposed it is 64 lines per ENTRY.  The transpose of the
// (small: nf x nA_d or nE_d) input is a ~3 us pre-kernel.
typedef double sr_double2 __attribute__((ext_vector_type(2)));
template <bool NT, int G, bool REALIGN, bool XT>
__global__ __launch_bounds__(SR_THREADS) void spmm_shortrow_kernel(
    const int *__restrict__ rowptr, const int *__restrict__ colind, const double *__restrict__ vals,
    const BatchPtrs bp, const double *__restrict__ XT0, long xt_stride, long ldx, long ldy, int nrow, int nf, int fper,
    const double *__restrict__ wM, double fill, unsigned *chain_cnt = nullptr, unsigned chain_n = 0, int chain_ncol = 0)
{
    __shared__ double s_y[REALIGN ? G : 1][SR_THREADS];
    extern __shared__ double sr_chain[];                 // chain: this workgroup's fields of X, [fper][chain_ncol]
    // blockIdx.y = field batch of a batched launch (ibh_weighted_apply_many_device): the workgroups of batch q+1 start while
    // those of batch q still store, so launch ramp and drain are paid once per launch
    const double *__restrict__ X = XT ? XT0 + (long)blockIdx.y * xt_stride : bp.x[blockIdx.y];
    double *__restrict__ Y = bp.y[blockIdx.y];
    // 1-D grid, field chunk fastest, contiguous ranges per XCD: the nfy workgroups that walk the same
    // 256 rows (one per field chunk) are neighbours on one XCD, so rowptr/colind/vals/wM of those rows
    // come from HBM once and from that XCD's L2 for the other chunks.
    const int nfy = (nf + fper - 1) / fper;
    const int logical = xcd_contiguous(blockIdx.x, gridDim.x);
    const int rb = logical / nfy, fy = logical - rb * nfy;
    const int rbase = rb * (REALIGN ? SR_STEP : SR_THREADS);
    const int r = rbase + threadIdx.x;
    const bool live = r < nrow;
    const int fbeg = fy * fper;
    const int fend = min(nf, fbeg + fper);
    if (chain_cnt) {
        // last kernel of a chain: dispatched without a barrier behind the kernel that makes X.  Wait for that kernel's workgroups,
        // then take this workgroup's fields of the (small: an A-space result) X into LDS through device-coherent loads -- once
        // per workgroup; the gathers below read LDS (plain loads could hit stale lines, coherent ones bypass the L1 they live on)
        chain_wait(chain_cnt, chain_n);
        const int nel = (fend - fbeg) * chain_ncol;
        for (int k = threadIdx.x; k < nel; k += SR_THREADS) {
            const int ff = k / chain_ncol, c = k - ff * chain_ncol;
            sr_chain[k] = chain_load(X + (long)(fbeg + ff) * ldx + c);
        }
        __syncthreads();
    }
    if (!REALIGN && !live) return;
    int beg = 0, end = 0;
    bool dead = false;
    if (live) { beg = rowptr[r]; end = rowptr[r + 1]; dead = wM[r] == 0.0; }
    const int n = end - beg;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (n > 0) { c0 = colind[beg]; v0 = vals[beg]; }
    if (n > 1) { c1 = colind[beg + 1]; v1 = vals[beg + 1]; }
    if (n > 2) { c2 = colind[beg + 2]; v2 = vals[beg + 2]; }
    if (n > 3) { c3 = colind[beg + 3]; v3 = vals[beg + 3]; }
    // G fields in flight per thread: all G gathers are issued before the first store.  vmcnt counts
    // loads and stores together in issue order, so a gather issued after a store would wait for that
    // store's (long) write acknowledgement; with G == fields per thread there is one load phase and
    // one store phase per thread.
    // SHARED columns (XT, groups of <= 4 fields): rows of one GCM cell follow each other in a fresh ice dimension, so for every entry
    // slot most lanes of a wave name the SAME column.  That column's X values come through scalar loads and feed the FMAs as scalar
    // operands; only the lanes that differ (the ice cells along the cell's edge: two per grid-row segment) gather theirs, and a gather
    // costs the texture path by its ACTIVE lanes.  A full-wave gather of 16 bytes per lane is 16 cycles per CU whatever the addresses:
    // the time of this kernel followed the entries of the longest row of a wave, not the bytes (16.3 M rows, 128 fields: 3.0 / 4.0 /
    // 5.3 ms at 1 / 2 / 4 entries in every row; 4.9 with 2 and 4 in 7 % of the rows, scattered: the Antarctic IvE).
    constexpr bool SHARED = XT && G <= 4;
    int ucA[4] = {0, 0, 0, 0}, ucB[4] = {0, 0, 0, 0};
    bool mineA[4] = {false, false, false, false}, mineB[4] = {false, false, false, false};     // this lane's entry e sits in shared column A / B
    bool hasB[4] = {false, false, false, false};     // (wave-uniform) a second shared column
    bool leftover = false;                           // (wave-uniform) lanes in neither: the wave takes the gather form below
    int nmax = 0;
    bool shared_ok = SHARED && __ballot(n > 4) == 0;      // (a smoothed matrix has longer rows: the gather form, too)
    if (shared_ok) {
        const int cc[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long he = __ballot(n > e);
            if (!he) break;
            nmax = e + 1;
            // A: the column of the first or of the last lane that has the entry, whichever more lanes name; B: the column of the
            // first lane outside A (a class boundary or a cell edge inside the wave)
            const int ca = __builtin_amdgcn_readlane(cc[e], __builtin_ctzll(he)), cb = __builtin_amdgcn_readlane(cc[e], 63 - __builtin_clzll(he));
            const unsigned long long ma = __ballot(n > e && cc[e] == ca), mb = __ballot(n > e && cc[e] == cb);
            ucA[e] = __popcll(ma) >= __popcll(mb) ? ca : cb;
            mineA[e] = n > e && cc[e] == ucA[e];
            const unsigned long long rest = __ballot(n > e && !mineA[e]);
            if (rest) {
                hasB[e] = true;
                ucB[e] = __builtin_amdgcn_readlane(cc[e], __builtin_ctzll(rest));
                mineB[e] = n > e && !mineA[e] && cc[e] == ucB[e];
                leftover = leftover || __ballot(n > e && !mineA[e] && !mineB[e]) != 0;
            }
        }
    }
    shared_ok = shared_ok && !leftover;
    for (int f0 = fbeg; f0 < fend; f0 += G) {
        double acc[G];
        if (shared_ok) {
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = 0.0;
            const double vv[4] = {v0, v1, v2, v3};
            // all loads of the group first (scalar: wave-uniform addresses)
            double sa[4][G], sb[4][G];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= nmax) break;
                const double *__restrict__ pa = X + (long)ucA[e] * ldx + f0;     // (wave-uniform addresses: scalar loads)
#pragma unroll
                for (int g = 0; g < G; ++g) sa[e][g] = pa[g];
                if (hasB[e]) {
                    const double *__restrict__ pb = X + (long)ucB[e] * ldx + f0;
#pragma unroll
                    for (int g = 0; g < G; ++g) sb[e][g] = pb[g];
                }
            }
            // the products, entry by entry in every lane (the row's order): never multiplied by a padded zero (0*NaN)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= nmax) break;
                if (mineA[e]) {
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g] = e == 0 ? vv[e] * sa[e][g] : fma(vv[e], sa[e][g], acc[g]);
                }
                if (hasB[e] && mineB[e]) {
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g] = e == 0 ? vv[e] * sb[e][g] : fma(vv[e], sb[e][g], acc[g]);
                }
            }
            if (dead) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = fill;
            }
        } else if (XT) {
            // entry by entry, G fields at a time (f0 is a multiple of G, rows of XT are padded to 16 fields)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = 0.0;
            auto entry = [&](int c, double v, bool first) {
                const sr_double2 *p = reinterpret_cast<const sr_double2 *>(X + (long)c * ldx + f0);
#pragma unroll
                for (int g = 0; g < G; g += 2) {
                    const sr_double2 x = p[g >> 1];
                    acc[g] = first ? v * x.x : fma(v, x.x, acc[g]);
                    acc[g + 1] = first ? v * x.y : fma(v, x.y, acc[g + 1]);
                }
            };
            // predicated, never multiplied by a padded zero: 0*NaN must not leak into a row
            if (n > 0) entry(c0, v0, true);
            if (n > 1) entry(c1, v1, false);
            if (n > 2) entry(c2, v2, false);
            if (n > 3) entry(c3, v3, false);
            for (int k = beg + 4; k < end; ++k) entry(colind[k], vals[k], false);
            if (dead) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = fill;
            }
        } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int f = f0 + g < fend ? f0 + g : fend - 1;
            const double *xf = X + (long)f * ldx;
            // predicated, never multiplied by a padded zero: 0*NaN must not leak into a row
            double a = 0.0;
            if (chain_cnt) {                            // (X comes from the kernel this one was dispatched behind without a barrier: staged)
                const double *xs = sr_chain + (f - fbeg) * chain_ncol;
                if (n > 0) a = v0 * xs[c0];
                if (n > 1) a = fma(v1, xs[c1], a);
                if (n > 2) a = fma(v2, xs[c2], a);
                if (n > 3) a = fma(v3, xs[c3], a);
                for (int k = beg + 4; k < end; ++k) a = fma(vals[k], xs[colind[k]], a);
            } else {
                if (n > 0) a = v0 * xf[c0];
                if (n > 1) a = fma(v1, xf[c1], a);
                if (n > 2) a = fma(v2, xf[c2], a);
                if (n > 3) a = fma(v3, xf[c3], a);
                for (int k = beg + 4; k < end; ++k) a = fma(vals[k], xf[colind[k]], a);
            }
            acc[g] = dead ? fill : a;
        }
        }
        if (!REALIGN) {
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (f0 + g < fend) {
                    if (NT) __builtin_nontemporal_store(acc[g], &Y[(long)(f0 + g) * ldy + r]);
                    else Y[(long)(f0 + g) * ldy + r] = acc[g];
                }
        } else {
            if (f0 != fbeg) __syncthreads();               // the previous group's values have been read
#pragma unroll
            for (int g = 0; g < G; ++g) s_y[g][threadIdx.x] = acc[g];
            __syncthreads();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (f0 + g >= fend) break;
                double *yf = Y + (long)(f0 + g) * ldy;
                // first row >= rbase whose address is a multiple of 64 bytes (the very first run starts at row 0)
                const long a8 = (long)(reinterpret_cast<uintptr_t>(yf) >> 3);
                const int s0 = rb == 0 ? 0 : rbase + (int)((-(a8 + rbase)) & 7);
                int s1 = rbase + SR_STEP + (int)((-(a8 + rbase + SR_STEP)) & 7);
                if (s1 > nrow || rbase + SR_STEP >= nrow) s1 = nrow;
                const int rr = s0 + (int)threadIdx.x;
                if (rr < s1) {
                    const double v = s_y[g][rr - rbase];
                    if (NT) __builtin_nontemporal_store(v, &yf[rr]); else yf[rr] = v;
                }
            }
        }
    }
}

// (Round 5, tried and removed -- "fieldlane": I-row matrices on >= 64 fields with lane = FIELD, a wave owning 16 rows, the row
// structure wave-uniform (columns and weights through v_readlane into the scalar offset of one coalesced 512-byte load of the
// transposed input per entry, one FMA, no masks), row sums through an LDS tile that leaves transposed.  Bitwise the row kernel on
// every matrix tried, smoothed ones included, and slower everywhere on the Antarctic sheet, 128 fields, IvE / IvA against 4.1 /
// 2.8 ms: a tile per wave stored as 16 or 8 rows per plane (128- / 64-byte pieces) 6.8 / 5.8 and 8.6-9.0 / 7.5-8.0 ms -- with 22
// scalar instructions per entry or with 4, the same; HBM traffic exact (16.9 GB written, 1.6 fetched) -- and a tile per workgroup
// with the row kernel's own 512-byte stores 7.1 / 6.1 ms: four workgroups of 33 KB per CU and six to eight dependent round trips
// per 64 rows.  The thread-per-row kernel keeps many more loads in flight per wave; it stays.)
// ---- conservation correction (non-conservative matrices only) ------------------------------
// out[k] = sum_j w[j] * A[k*lda + j] over j with w[j] != 0, in a fixed order: the vector is cut into <= WD_MAXCHUNK chunks
// (one workgroup each per variable: a single workgroup per variable reads 10^5..10^7 elements at one CU's bandwidth --
// measured 35 us of an 85 us corrected apply at 5 km), the chunk sums are added sequentially by the last stage.
constexpr int WD_MAXCHUNK = 256, WD_MINLEN = 4096;
static int weight_dot_chunks(int n) { return n <= WD_MINLEN ? 1 : std::min(WD_MAXCHUNK, ceil_div(n, WD_MINLEN)); }
__global__ __launch_bounds__(1024) void weight_dot_kernel(const double *__restrict__ w, int n, int chunk_len,
                                                          const double *__restrict__ A, long lda,
                                                          double *__restrict__ part)
{
    __shared__ double s[16];
    const double *a = A + (long)blockIdx.x * lda;
    const int j0 = blockIdx.y * chunk_len, j1 = min(n, j0 + chunk_len);
    double acc = 0.0;
    for (int j = j0 + threadIdx.x; j < j1; j += 1024) {
        const double wj = w[j];
        if (wj != 0.0) acc = fma(wj, a[j], acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += s[q];
        part[(long)blockIdx.x * gridDim.y + blockIdx.y] = t;
    }
}
__global__ void weight_dot_final_kernel(const double *__restrict__ part, int nchunk, double *__restrict__ out) {
    const int k = blockIdx.x;                      // one variable per (single-thread) workgroup: <= WD_MAXCHUNK additions, in chunk order
    double t = 0.0;
    for (int c = 0; c < nchunk; ++c) t += part[(long)k * nchunk + c];
    out[k] = t;
}
__global__ void conserve_scale_kernel(double *__restrict__ Y, long ldy, int nrow, const double *__restrict__ wM,
                                      const double *__restrict__ TA, const double *__restrict__ TB)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (r >= nrow || wM[r] == 0.0) return;
    Y[(long)k * ldy + r] *= TA[k] / TB[k];
}

// dout[nvar]; part: scratch of nvar * weight_dot_chunks(n) doubles (nullptr allowed when n <= WD_MINLEN: one chunk goes
// straight to dout)
void weight_dot_launch(const double *dw, int n, const double *dA, int nvar, int64_t lda, double *dout, double *part,
                       hipStream_t stream)
{
    if (nvar <= 0) return;
    const int nchunk = part ? weight_dot_chunks(n) : 1;
    const int len = ceil_div(n, nchunk);
    hipLaunchKernelGGL(weight_dot_kernel, dim3(nvar, nchunk), dim3(1024), 0, stream, dw, n, len, dA, (long)lda, nchunk == 1 ? dout : part);
    if (nchunk > 1) hipLaunchKernelGGL(weight_dot_final_kernel, dim3(nvar), dim3(1), 0, stream, part, nchunk, dout);
    IBH_HIP(hipGetLastError());
}
size_t weight_dot_scratch(int n, int nvar) { return (size_t)nvar * (size_t)weight_dot_chunks(n); }

// ---- dispatch ------------------------------------------------------------------------------
void ensure_rowsum1(const ibh_weighted *w, hipStream_t stream);
// Scratch owned by the handle grows OUTSIDE stream capture only (hipMalloc is illegal while capturing
// and synchronises the device): ibh_weighted_reserve() sizes it up front; an apply that still finds a
// buffer too small grows it after a device synchronisation (an earlier apply may be reading the old
// block), or fails when its stream is being captured.
template <class T>
static void grow_scratch(DevBuf<T> &b, size_t count, hipStream_t stream, const char *what) {
    if (b.p && count * sizeof(T) <= b.granted) { b.n = count; return; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream) IBH_HIP(hipStreamIsCapturing(stream, &cs));
    IBH_CHECK(cs == hipStreamCaptureStatusNone,
              "apply inside a stream capture needs %zu more bytes of %s scratch: call ibh_weighted_reserve() before capturing",
              count * sizeof(T), what);
    if (b.p) IBH_HIP(hipDeviceSynchronize());
    b.alloc(count);
}

// rowperm[k] = the row with the k-th most entries (ties by index): longest-processing-time-first order
// of the (row, chunk) tasks.  Rank by counting, nrow <= LPT_MAX_ROWS.
constexpr int LPT_MAX_ROWS = 8192;
__global__ void rowperm_kernel(const int *__restrict__ rowptr, int nrow, int *__restrict__ perm) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    const int len = rowptr[r + 1] - rowptr[r];
    int rank = 0;
    for (int q = 0; q < nrow; ++q) {
        const int lq = rowptr[q + 1] - rowptr[q];
        rank += (lq > len || (lq == len && q < r)) ? 1 : 0;
    }
    perm[rank] = r;
}
static bool ensure_rowperm(const ibh_weighted *w, hipStream_t stream) {
    if (w->have_rowperm) return true;
    if (w->nrow > LPT_MAX_ROWS || w->nrow < 2) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream) IBH_HIP(hipStreamIsCapturing(stream, &cs));
    if (cs != hipStreamCaptureStatusNone && !(w->rowperm.p && (size_t)w->nrow * sizeof(int) <= w->rowperm.granted)) return false;
    grow_scratch(w->rowperm, (size_t)w->nrow, stream, "row-order");
    hipLaunchKernelGGL(rowperm_kernel, dim3(ceil_div(w->nrow, 256)), dim3(256), 0, stream, w->rowptr.p, w->nrow, w->rowperm.p);
    IBH_HIP(hipGetLastError());
    w->have_rowperm = true;
    return true;
}

static long rowblock_grid(int nrow, int nfc, int &xcd_mode) {
    xcd_mode = (nfc % 8 == 0 || nfc == 1 || nfc == 2 || nfc == 4) ? 1 : 0;
    xcd_mode = get_tuning("rowblock_xcd_mode", xcd_mode);
    if (xcd_mode == 1 && !(nfc % 8 == 0 || nfc == 1 || nfc == 2 || nfc == 4)) xcd_mode = 0;
    long nb = ((long)nrow * nfc + 7) & ~7l;
    if (xcd_mode == 1) {
        if (nfc < 8) {                        // 8/nfc XCDs per chunk, each a row range of <= ceil(nrow/m) rows
            const int m = 8 / nfc;
            nb = 8l * ((nrow + m - 1) / m + 1);
        } else nb = (long)nrow * nfc;         // nfc % 8 == 0
    }
    return nb;
}

template <int FPW, int WK, int NW>
static void launch_rowblock(const ibh_weighted *w, const BatchPtrs &bp, int nbatch, int nvar, long lda, long ldb,
                            double fill, hipStream_t stream)
{
    constexpr int FB = FPW * (NW / WK);
    const int nfc = ceil_div(nvar, FB);
    int xcd_mode;
    const long nb = rowblock_grid(w->nrow, nfc, xcd_mode);
    IBH_CHECK(nb < (1l << 31), "spmm grid too large (%ld blocks)", nb);
    IBH_CHECK((long)w->ncol * 8 < (1l << 31), "ncol too large for 32-bit buffer offsets");
    // loads in flight per lane and field: enough 64-entry slots to cover a typical row in one batch
    int unroll = get_tuning("rowblock_unroll", 0);
    if (unroll == 0) {
        const double mean = w->nrow ? (double)w->nnz / (double)w->nrow / (64.0 * WK) : 1.0;
        // (a row is one partly filled batch when the unroll overshoots it: 5 km EvI, 2.1 passes of 64 per row, 26.2 / 18.4 us
        // with four loads in flight, 23.6 / 16.6 with two)
        unroll = mean > 6.0 ? 8 : mean > 3.0 ? 4 : mean > 1.5 ? 2 : 1;
        // one field per wave: the weights are read when the gathers land, a slot in flight holds two registers -> a row of up to
        // 896 entries in ONE batch of 14 gathers per lane (measured, 5 km AvI, 32 applies per launch: 7.31 -> 7.03 us per apply;
        // 12 or 16 are slower: 8.5 / 7.4)
        if (FPW == 1 && WK == 1 && NW == 8 && mean > 8.0 && mean <= 14.0) unroll = 14;
    }
    // batches per workgroup: the staged row segment and the prologue are shared by qi batches
    int qi = 1;
    if (nbatch > 1) {
        qi = get_tuning("rowblock_many_qi", 0);
        if (qi <= 0) qi = nbatch >= 4 ? 2 : 1;       // measured at the 5 km headline shape: depth 16 7.5 (qi 2) / 8.2 (qi 1) / 8.0 us (qi 8)
        if (qi > nbatch) qi = nbatch;
    }
    const dim3 grid((unsigned)nb, (unsigned)ceil_div(nbatch, qi));
    const int *rowperm = nullptr;
    if (nbatch > 1 && get_tuning("rowblock_lpt", 0) && ensure_rowperm(w, stream)) rowperm = w->rowperm.p;
    hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
#define IBH_RB(U)                                                                                        \
    do {                                                                                                 \
        snprintf(w->last_sig, sizeof(w->last_sig), "spmm_rowblock_kernel<%d, %d, %d, %d, false>", FPW, WK, (int)(U), NW);  \
        hipExtLaunchKernelGGL((spmm_rowblock_kernel<FPW, WK, U, NW>), grid, dim3(NW * 64), 0, stream, ev0, ev1, 0,  \
                              w->rowptr.p, w->colind.p, w->val.p, bp, nbatch, qi, lda, w->ncol, ldb, w->nrow, nvar, nfc, xcd_mode, w->wM.p, fill, \
                              (const double *)nullptr, (double *)nullptr, (const int *)rowperm, 0l);     \
    } while (0)
    if (unroll == 1) IBH_RB(1);
    else if (unroll == 2) IBH_RB(2);
    else if (unroll == 8) IBH_RB(8);
    else if (unroll > 8 && FPW == 1 && WK == 1) {      // one batch covers a whole row of <= 768 / 896 / 1024 entries
        constexpr bool one = FPW == 1 && WK == 1;
        if (unroll <= 12) IBH_RB((one ? 12 : 8)); else if (unroll <= 14) IBH_RB((one ? 14 : 8)); else IBH_RB((one ? 16 : 8));
    }
    else IBH_RB(4);
#undef IBH_RB
    IBH_HIP(hipGetLastError());
}

template <int NW, int U>
static void launch_rowone(const ibh_weighted *w, const double *X, double *Y, int nvar, long lda, long ldb, double fill, hipStream_t stream)
{
    const int nfc = ceil_div(nvar, NW);
    int xcd_mode;
    const long nb = rowblock_grid(w->nrow, nfc, xcd_mode);
    IBH_CHECK(nb < (1l << 31), "spmm grid too large (%ld blocks)", nb);
    IBH_CHECK((long)w->ncol * 8 < (1l << 31), "ncol too large for 32-bit buffer offsets");
    hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    snprintf(w->last_sig, sizeof(w->last_sig), "spmm_rowone_kernel<%d, %d>", NW, U);
    hipExtLaunchKernelGGL((spmm_rowone_kernel<NW, U>), dim3((unsigned)nb), dim3(NW * 64), 0, stream, ev0, ev1, 0, w->rowptr.p, w->colind.p,
                          w->val.p, X, Y, lda, w->ncol, ldb, w->nrow, nvar, nfc, xcd_mode, w->wM.p, fill);
    IBH_HIP(hipGetLastError());
}

template <int NW, int U, int TW>
static void launch_rowgroup(const ibh_weighted *w, const BatchPtrs &bp, int nbatch, int nvar, long lda, long ldb, double fill, hipStream_t stream,
                            const PairView *pair = nullptr)
{
    const int nfc = ceil_div(nvar, NW);
    int xcd_mode;
    const long nb = rowblock_grid(w->grp_n, nfc, xcd_mode);
    IBH_CHECK(nb < (1l << 31), "spmm grid too large (%ld blocks)", nb);
    IBH_CHECK((long)w->ncol * 8 < (1l << 31), "ncol too large for 32-bit buffer offsets");
    GroupView gv{w->grp_ptr.p, w->grp_ns.p, w->grp_slotrow.p, w->grp_col.p, w->grp_meta.p, w->grp_v0.p, w->grp_v1.p, w->grp_n, w->grp_nslot};
    const size_t lds = (size_t)(3 * U * 64 + NW * w->grp_nslot * TW) * sizeof(double);
    hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    if (lds > 64 * 1024) {                               // beyond the default dynamic-LDS limit: raise it once per device
        static std::mutex mu;
        static bool raised[64] = {};
        std::lock_guard<std::mutex> lk(mu);
        if (!raised[w->device & 63]) {
            IBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_rowgroup_kernel<NW, U, TW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised[w->device & 63] = true;
        }
    }
    snprintf(w->last_sig, sizeof(w->last_sig), "spmm_rowgroup_kernel<%d, %d, %d, %s>", NW, U, TW, pair ? "true" : "false");
    if (pair) {
        IBH_CHECK(lds <= 64 * 1024 && nbatch == 1, "fused pair apply: %zu bytes of LDS / %d batches not supported", lds, nbatch);
        hipExtLaunchKernelGGL((spmm_rowgroup_kernel<NW, U, TW, true>), dim3((unsigned)nb, 1u), dim3(NW * 64), lds, stream, ev0, ev1, 0, gv, bp,
                              lda, w->ncol, ldb, nvar, nfc, xcd_mode, w->wM.p, fill, *pair);
    } else
        hipExtLaunchKernelGGL((spmm_rowgroup_kernel<NW, U, TW>), dim3((unsigned)nb, (unsigned)nbatch), dim3(NW * 64), lds, stream, ev0, ev1, 0, gv, bp,
                              lda, w->ncol, ldb, nvar, nfc, xcd_mode, w->wM.p, fill, PairView{});
    IBH_HIP(hipGetLastError());
}

template <int F, int NS, int SEG, int NW>
static void launch_grouptile(const ibh_weighted *w, const BatchPtrs &bp, int nbatch, int nvar, long lda, long ldb, double fill, hipStream_t stream,
                             const PairView *pair = nullptr)
{
    const int nfc = ceil_div(nvar, F);
    int xcd_mode;
    const long nb = rowblock_grid(w->grp_n, nfc, xcd_mode);
    IBH_CHECK(nb < (1l << 31), "spmm grid too large (%ld blocks)", nb);
    IBH_CHECK((long)w->ncol * 8 < (1l << 31), "ncol too large for 32-bit buffer offsets");
    IBH_CHECK(w->grp_nslot <= NS && w->gt_seg == SEG, "row groups of %d rows / tiles of %d in a kernel for %d / %d", w->grp_nslot, w->gt_seg, NS, SEG);
    TileView tv{w->grp_ptr.p, w->gt_ptr.p, w->gt_col.p, w->gt_ek.p, w->gt_eptr.p, w->gt_ev.p, w->grp_ns.p, w->grp_slotrow.p, w->grp_n, nullptr};
#ifdef GT_TIMELINE
    tv.dbg = reinterpret_cast<long long *>(((unsigned long long)(unsigned)get_tuning("gt_dbg_hi", 0) << 32) | (unsigned)get_tuning("gt_dbg_lo", 0));
#endif
    constexpr size_t lds = grouptile_lds<F, SEG>();
    static_assert(lds <= 64 * 1024, "within the default dynamic-LDS limit");
    hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    snprintf(w->last_sig, sizeof(w->last_sig), "spmm_grouptile_kernel<%d, %d, %d, %d, %s>", F, NS, SEG, NW, pair ? "true" : "false");
    if (pair) {
        IBH_CHECK(nbatch == 1, "fused pair apply: %d batches not supported", nbatch);
        hipExtLaunchKernelGGL((spmm_grouptile_kernel<F, NS, SEG, NW, true>), dim3((unsigned)nb, 1u), dim3(NW * 64), lds, stream, ev0, ev1, 0, tv, bp,
                              lda, w->ncol, ldb, nvar, nfc, xcd_mode, w->wM.p, fill, *pair);
    } else
        hipExtLaunchKernelGGL((spmm_grouptile_kernel<F, NS, SEG, NW, false>), dim3((unsigned)nb, (unsigned)nbatch), dim3(NW * 64), lds, stream, ev0, ev1, 0,
                              tv, bp, lda, w->ncol, ldb, nvar, nfc, xcd_mode, w->wM.p, fill, PairView{});
    IBH_HIP(hipGetLastError());
}
// (rows per group, tile size) -> instantiation; 16 fields per workgroup (32 lost everywhere it was measured: 1 km 269 against 218 us,
// the Antarctic sheet 3.67 against 3.62 ms, 5 km 22 against 18 us)
static void launch_grouptile_any(const ibh_weighted *w, const BatchPtrs &bp, int nbatch, int nvar, long lda, long ldb, double fill, hipStream_t stream,
                                 const PairView *pair = nullptr)
{
    const bool few = w->grp_nslot <= 16;
    if (w->gt_seg == 128) {
        if (few) launch_grouptile<16, 16, 128, 4>(w, bp, nbatch, nvar, lda, ldb, fill, stream, pair);
        else launch_grouptile<16, 32, 128, 4>(w, bp, nbatch, nvar, lda, ldb, fill, stream, pair);
    } else {
        if (few) launch_grouptile<16, 16, 256, 8>(w, bp, nbatch, nvar, lda, ldb, fill, stream, pair);
        else launch_grouptile<16, 32, 256, 8>(w, bp, nbatch, nvar, lda, ldb, fill, stream, pair);
    }
}
// Which form of the row groups: the tiles (grouptile) for bandwidth-sized matrices -- measured, one apply per launch, tiles
// against LDS atomics: the Antarctic sheet (35 M entries) 16 fields 503 against 672 us, 128 fields 3.62 against 4.34 ms; 1 km
// Greenland (4 M entries) 64 fields 218 against 220 us, 16 fields 82 against 74; 5 km (166 k entries: latency-bound, three
// tiles in sequence per GCM cell) 12.4 against 7.6 and 18.1 against 17.4 us.  ibh_set_tuning("rowgroup_form", 0 / 1) forces one.
static bool use_grouptile(const ibh_weighted *w, int nvar) {
    if (w->gt_ntile <= 0) return false;
    const int form = get_tuning("rowgroup_form", -1);
    if (form >= 0) return form == 1;
    return w->nnz >= (1l << 24) || (w->nnz >= (1l << 21) && nvar >= 48);
}

// B[f, r] = lower-class sum of band r + upper-class sum of the band below it (rb1[r], -1: none)
__global__ void dual_combine_kernel(const double *__restrict__ P0, const double *__restrict__ P1, long ldp, long pstride,
                                    const int *__restrict__ rb1, const double *__restrict__ wM, double fill,
                                    const BatchPtrs bp, long ldy, int nrow, int nf) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y, q = blockIdx.z;
    if (r >= nrow) return;
    const int b = rb1[r];
    double t = P0[(long)q * pstride + (long)f * ldp + r];
    if (b >= 0) t = t + P1[(long)q * pstride + (long)f * ldp + b];
    bp.y[q][(long)f * ldy + r] = wM[r] == 0.0 ? fill : t;
}

static size_t band_part_count(const ibh_weighted *w, int nvar, int nbatch) {
    const long ldp = ((long)w->nrow + 63) & ~63l;
    return 2 * (size_t)nbatch * (size_t)nvar * (size_t)ldp;
}
template <int FPW>
static void launch_rowdual(const ibh_weighted *w, const BatchPtrs &bp, int nbatch, int nvar, long lda, long ldb,
                           double fill, hipStream_t stream)
{
    constexpr int NW = 4, FB = FPW * NW;
    const int nfc = ceil_div(nvar, FB);
    int xcd_mode;
    const long nb = rowblock_grid(w->nrow, nfc, xcd_mode);
    IBH_CHECK(nb < (1l << 31), "spmm grid too large (%ld blocks)", nb);
    IBH_CHECK((long)w->ncol * 8 < (1l << 31), "ncol too large for 32-bit buffer offsets");
    const long ldp = ((long)w->nrow + 63) & ~63l;
    const long pstride = (long)nvar * ldp;                       // one batch's lower (or upper) partial sums
    grow_scratch(w->band_part, band_part_count(w, nvar, nbatch), stream, "band");
    double *P0 = w->band_part.p, *P1 = P0 + (size_t)nbatch * (size_t)pstride;
    const double mean = w->nrow ? (double)w->band_n / (double)w->nrow / 64.0 : 1.0;
    const int unroll = mean > 4.0 ? 8 : mean > 2.0 ? 4 : mean > 1.0 ? 2 : 1;
    BatchPtrs bq{};
    for (int q = 0; q < nbatch; ++q) { bq.x[q] = bp.x[q]; bq.y[q] = P0 + (size_t)q * (size_t)pstride; }
    const dim3 grid((unsigned)nb, (unsigned)nbatch);
#define IBH_RD(U)                                                                                                  \
    hipLaunchKernelGGL((spmm_rowblock_kernel<FPW, 1, U, NW, true>), grid, dim3(NW * 64), 0, stream,                 \
                       w->band_ptr.p, w->band_col.p, w->band_v0.p, bq, nbatch, 1, lda, w->ncol, ldp, w->nrow, nvar, nfc, xcd_mode, \
                       w->wM.p, fill, w->band_v1.p, P1, (const int *)nullptr, pstride)
    if (unroll == 1) IBH_RD(1); else if (unroll == 2) IBH_RD(2); else if (unroll == 8) IBH_RD(8); else IBH_RD(4);
#undef IBH_RD
    hipLaunchKernelGGL(dual_combine_kernel, dim3((unsigned)ceil_div(w->nrow, 256), (unsigned)nvar, (unsigned)nbatch), dim3(256), 0, stream,
                       P0, P1, ldp, pstride, w->band_rb1.p, w->wM.p, fill, bp, ldb, w->nrow, nvar);
    IBH_HIP(hipGetLastError());
}

// fields per lane group of the sweep: 64 (one batch per wave row) from 33 fields; below, the next power of two >= 8 so that
// 64 / nfl batches share the lanes
static int sweep_lg(int nvar) { return nvar > 32 ? 6 : nvar > 16 ? 5 : nvar > 8 ? 4 : 3; }
// lanes of the sweep that carry a (batch, field) pair for nvar fields in launches of nbatch
static int sweep_lanes(int nvar, int nbatch) { return nvar > 32 ? 64 : nvar * std::min(nbatch, 64 >> sweep_lg(nvar)); }
// ---- colsweep (sweep_kernel.inl): E-row matrices, >= 32 fields ------------------------------------------------
// Y[f, r] = the partial sums of the tasks that touch row r (comb_p[comb_ptr[r] .. comb_ptr[r+1])), in task order; rows no
// task touches are 0, rows with wM == 0 hold `fill` (mask_result, IceCoupler.cpp:186-201).
__global__ __launch_bounds__(256) void sweep_combine_kernel(const double *__restrict__ P, long pstride, long ldp,
                                                            const int *__restrict__ comb_ptr, const int *__restrict__ comb_p,
                                                            const double *__restrict__ wM, double fill, const BatchPtrs bp, long ldy,
                                                            int nrow, int nf, int lg)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.y * 64 + (threadIdx.x & 63);
    const int q = blockIdx.z;
    if (r >= nrow || f >= nf) return;
    const int k0 = comb_ptr[r], k1 = comb_ptr[r + 1];
    // the sweep's lanes are (batch, field) pairs (sweep_kernel.inl): lg = 6 -> one batch per slice, lane = field of the field
    // block; lg < 6 -> 64 >> lg batches per slice, lane = (batch % G) << lg | field
    const int G = 64 >> lg;
    const double *p = lg == 6 ? P + (long)q * pstride + f : P + (long)(q / G) * pstride + ((q % G) << lg) + f;
    // eight partial rows in flight (index, then value: two dependent loads each), added in task order
    double tot = 0.0;
    int k = k0;
    if (k < k1) { tot = p[(long)comb_p[k] * ldp]; ++k; }
    for (; k + 8 <= k1; k += 8) {
        int idx[8];
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) idx[u] = comb_p[k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long)idx[u] * ldp];
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += v[u];
    }
    for (; k < k1; ++k) tot += p[(long)comb_p[k] * ldp];
    bp.y[q][(long)f * ldy + r] = wM[r] == 0.0 ? fill : tot;
}
// fields per lane group of the sweep: 64 (one batch per wave row) from 33 fields; below, the next power of two >= 8 so that
// 64 / nfl batches share the lanes
static size_t sweep_part_count(const ibh_weighted *w, int nvar, int nbatch) {
    const int lg = sweep_lg(nvar);
    const size_t slices = lg == 6 ? (size_t)nbatch : (size_t)ceil_div(nbatch, 64 >> lg);
    const size_t ldp = lg == 6 ? (size_t)ceil_div(nvar, 64) * 64 : 64;
    return slices * (size_t)w->sweep_nprow * ldp;
}
static void launch_sweep(const ibh_weighted *w, const BatchPtrs &bp, int nbatch, int nvar, long lda, long ldb, double fill,
                         hipStream_t stream)
{
    const int lg = sweep_lg(nvar), G = 64 >> lg;
    const int nfb = lg == 6 ? ceil_div(nvar, 64) : 1, nz = lg == 6 ? nbatch : ceil_div(nbatch, G);
    const long ldp = (long)nfb * 64, pstride = (long)w->sweep_nprow * ldp;
    grow_scratch(w->sweep_part, sweep_part_count(w, nvar, nbatch), stream, "column-sweep");
    SweepView sv{w->sweep_task_p0.p, w->sweep_task_ns.p, w->sweep_col.p, w->sweep_meta.p, w->sweep_v0.p, w->sweep_v1.p,
                 w->sweep_tb, w->sweep_nblk, w->sweep_nitems};
    SweepBatch sb{};
    for (int q = 0; q < nbatch; ++q) sb.x[q] = bp.x[q];
    for (int zz = 0; zz < nz; ++zz) sb.p[zz] = w->sweep_part.p + (size_t)zz * (size_t)pstride;
    const size_t lds = sweep_lds_bytes(w->sweep_nslot);
    const bool full = lg == 6 ? nvar % 64 == 0 : (nvar == (1 << lg) && nbatch % G == 0);
    const bool ident = w->sweep_ident != 0;
    const void *fn = full ? (ident ? reinterpret_cast<const void *>(spmm_sweep_kernel<true, true, 0>) : reinterpret_cast<const void *>(spmm_sweep_kernel<true, false, 0>))
                          : (ident ? reinterpret_cast<const void *>(spmm_sweep_kernel<false, true, 0>) : reinterpret_cast<const void *>(spmm_sweep_kernel<false, false, 0>));
    if (lds > 64 * 1024) {                               // beyond the default dynamic-LDS limit: raise it once per device and variant
        static std::mutex mu;
        static bool raised[64][4] = {};
        std::lock_guard<std::mutex> lk(mu);
        const int dev = w->device & 63, var = (full ? 2 : 0) + (ident ? 1 : 0);
        if (!raised[dev][var]) {
            IBH_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised[dev][var] = true;
        }
    }
    const dim3 grid((unsigned)w->sweep_ntask, (unsigned)nfb, (unsigned)nz);
    hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    // launch timing: start = the sweep kernel begins, stop = the combine kernel ends (both belong to the apply)
#define IBH_SW(F, I) hipExtLaunchKernelGGL((spmm_sweep_kernel<F, I, 0>), grid, dim3(SWEEP_NW * 64), lds, stream, ev0, nullptr, 0, sv, sb, lda, nvar, w->sweep_nslot, ldp, lg, nbatch)
    if (full) { if (ident) IBH_SW(true, true); else IBH_SW(true, false); }
    else { if (ident) IBH_SW(false, true); else IBH_SW(false, false); }
#undef IBH_SW
    hipExtLaunchKernelGGL(sweep_combine_kernel, dim3((unsigned)ceil_div(w->nrow, 4), (unsigned)nfb, (unsigned)nbatch), dim3(256), 0, stream, nullptr, ev1, 0,
                       w->sweep_part.p, pstride, ldp, w->sweep_comb_ptr.p, w->sweep_comb_p.p, w->wM.p, fill,
                       bp, ldb, w->nrow, nvar, lg);
    IBH_HIP(hipGetLastError());
}

// which kernel serves (w, nvar): 1 rowblock, 2 shortrow, 3 rowdual, 4 colsweep, 5 rowgroup
static int pick_kernel(const ibh_weighted *w, int nvar, int nbatch = 1) {
    int kernel = w->kernel_override;
    if (kernel == 4 && w->sweep_ntask == 0) kernel = 0;       // no column-sweep structure: the automatic choice
    if (kernel == 0) {
        // rowblock = one workgroup per (row, field chunk): for FEW LONG rows.  Many rows of 6..63 entries (a smoothed IvE:
        // 76 k rows of ~16) are thread-per-row work (measured, 5 km smoothed IvE, 16 fields: 220 us as rowblock)
        const double mean = w->nrow ? (double)w->nnz / (double)w->nrow : 0.0;
        const bool few_rows = w->nrow <= get_tuning("rowblock_max_short_rows", 16384);
        kernel = (mean >= 64.0 || (few_rows && mean >= (double)get_tuning("rowblock_min_mean_nnz", 6))) ? 1 : 2;
    }
    // E-row matrices (EvI, EvX), once the structure exists: the row groups (every X element gathered once per GCM cell; measured
    // against the sweep at 1 km, 64 fields: 221 against 247 us one launch per apply, 199 against 203-223 batched) -- except
    // batched launches of FEWER than 32 fields, where the batches share the lanes of the column sweep (1 km, 16 fields, 16 per
    // launch: 51 against 63 us per apply).  The other long-row matrices (AvI, AvX) take the sweep in batched launches only
    // (1 km, 64 fields: 167 against 173 us per apply 32 deep, but 193 against 183 us one launch per apply).
    if (kernel == 1 && w->kernel_override == 0) {
        // (round 4, the Antarctic sheet -- 17.6 / 35.2 M entries -- one apply per launch: AvI, 128 fields, sweep 3 131 against 3 654 us;
        // EvI, 16 fields, row groups 641 against 812 (bands) / 747 (rows) / 1 994 us (sweep): scratch/kernel_choice.py)
        const bool huge = w->nnz >= (1l << 24);
        const bool sweep_ok = w->sweep_ntask > 0 && sweep_lanes(nvar, nbatch) >= get_tuning("sweep_min_nvar", 32) && get_tuning("sweep_auto", 1) &&
                              (w->band_eligible || nbatch >= get_tuning("sweep_min_batch", 4) || (huge && nvar >= 128));
        const bool grp_ok = w->grp_n > 0 && nvar >= 4 && get_tuning("rowgroup_auto", 1);
        // (round 5: on matrices of 2^24 entries and more the tiled row groups beat the sweep in batched launches of few fields too --
        // the Antarctic EvI, 16 fields, batches of 4: bench.py 0.487 of peak through the sweep, measured again below)
        const bool tiles_win = w->gt_ntile > 0 && w->nnz >= (1l << 24) && get_tuning("rowgroup_form", -1) != 0;
        if (grp_ok && (nvar >= 32 || !sweep_ok || tiles_win)) kernel = 5;
        else if (sweep_ok) kernel = 4;
    }
    if (kernel == 5 && w->grp_n == 0) kernel = 1;             // no row groups were built for this matrix
    if (kernel == 1 && w->kernel_override == 0 && w->band_n > 0 && nvar >= 4 && get_tuning("rowdual_auto", 1)) kernel = 3;
    if (kernel == 3 && w->band_n == 0) kernel = 1;            // no bands were built for this matrix
    return kernel;
}
struct ShortrowPlan { int fper, g, use_xt, ldt; bool one_entry, big; };
static ShortrowPlan shortrow_plan(const ibh_weighted *w, int nvar, int nbatch = 1) {
    // fields per thread.  At 1 km (1.9 M rows) the stores dominate and 16-32 fields per thread amortise the row's CSR reads.
    // Small problems (5 km: 76 k rows): round-2 sweep with the batched kernel (scratch/one_matrix.py, us per 64-field apply,
    // one launch / 16 per launch): one entry per row (IvA) 8 fields x transposed input 12.1 / 7.0 (4 fields, field-major:
    // 13.2 / 9.7); 2-3 entries (IvE) 16 fields 16.5 / 9.9, 32 fields 19.2 / 9.5 -- more fields per thread pay in deep launches.
    ShortrowPlan p;
    p.one_entry = (double)w->nnz <= 1.5 * (double)w->nrow;
    p.big = w->nrow >= (1 << 19);
    // (2-3 entries, one launch of <= 32 fields: 4 fields per thread, field-major 8.2 us against 14.2 through the transposed copy)
    // (32 fields: 13.7 against 16.0; rows of 2-3 entries only: a smoothed IvE -- ~16 entries per row -- needs the lines of the
    // transposed copy: 59.6 against 143 us at 16 fields)
    const bool few_fields_once = nbatch < 4 && nvar <= 32 && (double)w->nnz <= 4.0 * (double)w->nrow;
    const int small_multi = nbatch >= 4 ? 32 : few_fields_once ? 4 : 16;
    // (round 3, kernel durations by dispatch events instead of wall time: ONE launch of a one-entry matrix at 5 km is fastest
    // field-major with 4 fields per thread -- 64 fields 13.2 us against 15.8 through the transposed copy, whose second
    // launch costs more than its lines save; 16 fields 5.6 against 6.8; deep launches keep the transposed form: 7.0)
    const int small_one = w->nrow >= 16384 ? (nbatch >= 4 ? 8 : nvar >= 64 ? 16 : 4) : 4;
    // (round 4, the Antarctic sheet -- 17.2 M one-entry rows, results of 2.2 / 17.6 GB: groups of 4 fields and, from ~100 fields on,
    // 32 fields per thread: 16 fields 579 -> 546 us, 128 fields 4 761 -> 3 829 us = 0.45 -> 0.56 of 8 TB/s; scratch/tune_shortrow.py)
    const bool huge = w->nrow >= (1 << 23);
    p.fper = get_tuning("shortrow_fper", p.big ? (p.one_entry ? (huge && nvar >= 96 ? 32 : 16) : 32) : (p.one_entry ? small_one : small_multi));
    if (p.fper < 1) p.fper = 1;
    p.g = get_tuning("shortrow_group", p.big ? (p.one_entry ? (huge ? 4 : 8) : 4) : (p.fper >= 8 ? 8 : 4));
    if (p.g > p.fper) p.g = p.fper;
    // transposed input: the G fields of an entry are 8*G contiguous bytes per lane (one line per entry instead of one per
    // field): 5 km IvE 26.9 -> 18.5 us, 1 km IvE 302 -> 183 us, 1 km IvA 207 -> 176 us, 5 km IvA 13.2 -> 12.1 us
    p.use_xt = get_tuning("shortrow_xt", -1);
    // (one-entry matrices: the extra launch costs more than it saves for tiny matrices -- EvA: 4.9 -> 8.9 us -- and for a
    // single launch of few fields -- 5 km IvA, 16 fields: 7.5 -> 9.3 us)
    if (p.use_xt < 0) p.use_xt = ((!p.one_entry && !(few_fields_once && !p.big)) || p.big || (w->nrow >= 16384 && nbatch >= 4)) ? 1 : 0;
    if (p.fper % p.g != 0 || (p.g & 1)) p.use_xt = 0;
    p.ldt = (nvar + 15) & ~15;
    return p;
}

// ---- lazily built apply structures (column sweep, bands) ---------------------------------------------------
// Which structure applies of (nvar fields, nbatch per launch) would use, by the rules the apply path has always had;
// `seen` = the matrix has been applied before (an apply builds on the SECOND call only: the coupler's one build : one apply
// must not pay for a structure it never reuses; ibh_weighted_prepare builds at once).
static bool wants_sweep(const ibh_weighted *w, int nvar, int nbatch, bool seen) {
    if (w->sweep_tried || w->sweep_ntask > 0) return false;
    if (w->kernel_override == 4) return true;
    const bool long_rows = w->nrow > 0 && (double)w->nnz / (double)w->nrow >= 64.0 && w->nnz <= 2 * (int64_t)w->ncol;      // AvI, AvX
    const bool e_rows = w->band_eligible && (nvar < 32 || (w->grp_tried && w->grp_n == 0) || !get_tuning("rowgroup_auto", 1));
    const bool huge_wide = w->nnz >= (1l << 24) && nvar >= 128;      // (one launch of >= 128 fields on the Antarctic AvI: see launch_kernel_for)
    return (e_rows || (long_rows && (nbatch >= get_tuning("sweep_min_batch", 4) || huge_wide))) && w->kernel_override == 0 && seen &&
           sweep_lanes(nvar, nbatch) >= get_tuning("sweep_min_nvar", 32) && get_tuning("sweep_auto", 1) &&
           (double)w->nnz * sweep_lanes(nvar, nbatch) >= (double)get_tuning("sweep_min_work", 64 << 20);
}
static bool wants_bands(const ibh_weighted *w, int nvar, bool seen) {
    return w->band_eligible && !w->band_tried && w->band_n == 0 && w->sweep_ntask == 0 && seen && w->kernel_override == 0 && nvar >= 4 &&
           (double)w->nnz * nvar >= (double)get_tuning("rowdual_min_work", 128 << 20) && get_tuning("rowdual_auto", 1);
}
// Row groups (rowgroup kernel) of an E-row matrix: bandwidth-sized matrices applied to >= 32 fields get them like the sweep
// always got its structure -- on the SECOND apply, or at once in ibh_weighted_prepare; with fewer fields such a matrix takes the
// sweep (batches share its lanes).  Small matrices (5 km: the sweep is latency-bound there) get them from ibh_weighted_prepare
// only: the structure costs about as much as the matrix build itself, and an apply that switched kernels on its own would change
// the rounding of later results against earlier ones (ibh_set_tuning("rowgroup_after", n) asks for exactly that, from the
// n-th apply on).
static bool sweep_sized(const ibh_weighted *w, int nvar, int nbatch) {
    return (double)w->nnz * std::max(sweep_lanes(nvar, nbatch), std::min(nvar, 64)) >= (double)get_tuning("sweep_min_work", 64 << 20);
}
static bool wants_groups(const ibh_weighted *w, int nvar, int nbatch, long seen) {
    if (!w->band_eligible || w->grp_tried || w->grp_n > 0) return false;
    if (w->kernel_override == 5) return true;
    if (w->kernel_override != 0 || nvar < 4 || !get_tuning("rowgroup_auto", 1)) return false;
    // (bandwidth-sized matrices: from 32 fields -- fewer fields per launch share the lanes of the column sweep when launches are
    // batched; a single launch of a 2^24-entry matrix takes the groups from 4 fields on)
    if (sweep_sized(w, nvar, nbatch)) return (nvar >= 32 || (nbatch < 4 && w->nnz >= (1l << 24))) && seen >= 1;
    return seen >= get_tuning("rowgroup_after", 1 << 30);
}
static void drop_groups(const ibh_weighted *w) {
    w->grp_n = 0;
    w->pair_second = nullptr;                                 // (a pairing indexes the group table)
    w->grp_ptr.release(); w->grp_ns.release(); w->grp_slotrow.release(); w->grp_col.release(); w->grp_meta.release();
    w->grp_v0.release(); w->grp_v1.release();
    w->gt_ntile = 0; w->gt_seg = 0;
    w->gt_ptr.release(); w->gt_col.release(); w->gt_ek.release(); w->gt_eptr.release(); w->gt_ev.release();
}
static void drop_sweep(const ibh_weighted *w) {
    w->sweep_ntask = 0;
    w->sweep_task_p0.release(); w->sweep_task_ns.release(); w->sweep_col.release(); w->sweep_meta.release();
    w->sweep_v0.release(); w->sweep_v1.release(); w->sweep_comb_ptr.release(); w->sweep_comb_p.release();
}
static void drop_bands(const ibh_weighted *w) {
    w->band_n = 0;
    w->band_ptr.release(); w->band_col.release(); w->band_rb1.release(); w->band_v0.release(); w->band_v1.release();
}
// A structure that cannot be built (out of memory, not representable) is no reason to fail an apply the row-by-row kernel
// serves: the failure is swallowed, the structure marked as tried, the matrix keeps its kernel.  Never inside a capture.
static void build_structures(const ibh_weighted *w, int nvar, int nbatch, bool seen, hipStream_t stream, long nseen = 1l << 30) {
    const long ns = seen ? nseen : 0;
    if (!wants_groups(w, nvar, nbatch, ns) && !wants_sweep(w, nvar, nbatch, seen) && !wants_bands(w, nvar, seen)) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream) IBH_HIP(hipStreamIsCapturing(stream, &cs));
    if (cs != hipStreamCaptureStatusNone) return;
    if (wants_groups(w, nvar, nbatch, ns)) {
        w->grp_tried = 1;
        try { (void)build_groups_from_csr(w, stream); }
        catch (const Error &) { (void)hipGetLastError(); drop_groups(w); }
    }
    if (wants_sweep(w, nvar, nbatch, seen)) {           // (also: an E-row matrix whose groups were declined just now)
        w->sweep_tried = 1;
        try { (void)build_sweep_from_csr(w, stream); }
        catch (const Error &) { (void)hipGetLastError(); drop_sweep(w); }
    }
    if (w->grp_n == 0 && wants_bands(w, nvar, seen)) {
        w->band_tried = 1;
        try { build_bands_from_csr(w, stream); }
        catch (const Error &) { (void)hipGetLastError(); drop_bands(w); }
    }
}

// Scratch of every kernel that applies of (nvar, <= nbatch per launch) can pick -- all of them: a later
// ibh_weighted_set_kernel, or a tuning change, must not turn a captured apply into an allocation.
static void size_scratch(const ibh_weighted *w, int nvar, int nbatch) {
    if (w->band_n > 0) grow_scratch(w->band_part, band_part_count(w, nvar, nbatch), nullptr, "band");
    if (w->sweep_ntask > 0) grow_scratch(w->sweep_part, sweep_part_count(w, nvar, nbatch), nullptr, "column-sweep");
    if (pick_kernel(w, nvar, nbatch) == 2 || pick_kernel(w, nvar, 1) == 2 || w->kernel_override == 0) {
        for (int nb : {1, nbatch}) {
            const ShortrowPlan p = shortrow_plan(w, nvar, nb);
            const int qmax = std::max(1, get_tuning("shortrow_many", w->nrow >= (1 << 19) ? 1 : IBH_MAX_BATCH));
            if (p.use_xt && (pick_kernel(w, nvar, nb) == 2 || w->kernel_override == 2))
                grow_scratch(w->xt, (size_t)w->ncol * (size_t)p.ldt * (size_t)std::min(qmax, nb), nullptr, "transposed-input");
        }
    }
    grow_scratch(w->consv, 2 * (size_t)nvar + weight_dot_scratch(std::max(w->nrow, w->ncol), nvar), nullptr, "conservation");
    // apply_transformed: the small side holds nvar fields
    grow_scratch(w->scratch, (size_t)nvar * (size_t)std::min(w->nrow, w->ncol), nullptr, "transform");
    ensure_rowsum1(w, nullptr);
    if (get_tuning("rowblock_lpt", 0)) (void)ensure_rowperm(w, nullptr);
}

void weighted_reserve(const ibh_weighted *w, int nvar) {
    HandleScope hs_(w);
    if (nvar <= 0) return;
    size_scratch(w, nvar, 1);
}

void weighted_prepare(const ibh_weighted *w, int nvar, int nbatch) {
    HandleScope hs_(w);
    if (nvar <= 0 || nbatch <= 0 || w->nrow == 0) return;
    const int nb = std::min(nbatch, IBH_MAX_BATCH);           // deeper batches are split into launches of IBH_MAX_BATCH
    build_structures(w, nvar, nb, true, nullptr);
    size_scratch(w, nvar, nb);
    IBH_HIP(hipStreamSynchronize(nullptr));
}

// ---- fused pair: B1 = first * A, B2 = second * B1 in one launch (BASELINE config 3: EvI then AvE) --------------------------------
// Possible when every row of `second` reads rows of ONE row group of `first` only -- AvE after EvI: the value of a GCM cell is a
// combination of that cell's own elevation classes -- so the group's workgroup has all inputs of the second row in hand when
// it has its class sums.  The pairing is worked out on the host (the second matrix is tiny: one entry per class) through the
// SPARSE indices of the two matrices' shared dimension, so the two need not share a SparseSet object or a numbering.
void weighted_pair_prepare(const ibh_weighted *first, const ibh_weighted *second, int nvar) {
    HandleScope hs_(first);
    IBH_CHECK(first && second && first != second, "pair: two different matrices expected");
    IBH_CHECK(first->device == second->device, "pair: matrices live on different devices");
    IBH_CHECK(second->ncol == first->nrow || second->dims[1], "pair: the second matrix's columns are not the first one's rows");
    IBH_CHECK(second->nnz <= (1 << 22), "pair: the second matrix is too large (%ld entries) to be paired on the host", (long)second->nnz);
    first->pair_second = nullptr;
    if (first->grp_n == 0) {                                  // row groups first (small matrices get them from an explicit request only)
        const int keep = first->kernel_override;
        const_cast<ibh_weighted *>(first)->kernel_override = 5;
        try { weighted_prepare(first, nvar > 0 ? nvar : 16, 1); } catch (...) { const_cast<ibh_weighted *>(first)->kernel_override = keep; throw; }
        const_cast<ibh_weighted *>(first)->kernel_override = keep;
    }
    if (first->grp_n == 0) fail(IBH_ENOTIMPL, "pair: the first matrix has no row groups (not an elevation-class-row matrix, or its structure was declined)");
    const int ngrp = first->grp_n;
    std::vector<int32_t> slotrow((size_t)ngrp * IBH_GSLOTS), gns((size_t)ngrp), rp2((size_t)second->nrow + 1), ci2((size_t)second->nnz);
    std::vector<double> v2((size_t)second->nnz);
    IBH_HIP(hipMemcpy(slotrow.data(), first->grp_slotrow.p, slotrow.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    IBH_HIP(hipMemcpy(gns.data(), first->grp_ns.p, gns.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    IBH_HIP(hipMemcpy(rp2.data(), second->rowptr.p, rp2.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (second->nnz) {
        IBH_HIP(hipMemcpy(ci2.data(), second->colind.p, ci2.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        IBH_HIP(hipMemcpy(v2.data(), second->val.p, v2.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    // first row (dense) -> (group, slot)
    std::vector<int32_t> where_g((size_t)first->nrow, -1), where_s((size_t)first->nrow, -1);
    for (int g = 0; g < ngrp; ++g)
        for (int sl = 0; sl < gns[(size_t)g]; ++sl) {
            const int r = slotrow[(size_t)g * IBH_GSLOTS + sl];
            IBH_CHECK(r >= 0 && r < first->nrow, "pair: corrupt group table");
            where_g[(size_t)r] = g; where_s[(size_t)r] = sl;
        }
    // second column (dense) -> first row (dense), through the sparse indices of the shared dimension
    const ibh_sparse_set *d1 = first->dims[0], *d2 = second->dims[1];
    const bool same_numbering = d1 == d2 || !d1 || !d2;
    if (same_numbering) IBH_CHECK(second->ncol == first->nrow, "pair: extents differ (%d columns, %d rows)", second->ncol, first->nrow);
    else {
        if (d1->sparse_extent != d2->sparse_extent)
            fail(IBH_ENOTIMPL, "pair: the second matrix's columns (extent %ld) are not the first one's rows (extent %ld)",
                 (long)d2->sparse_extent, (long)d1->sparse_extent);
        d1->ensure_inverse(); d2->ensure_host();
    }
    std::vector<double> pw((size_t)ngrp * IBH_GSLOTS, 0.0);
    std::vector<uint32_t> pm((size_t)ngrp, 0u);
    std::vector<int32_t> pr((size_t)ngrp, -1);
    for (int a = 0; a < second->nrow; ++a) {
        if (rp2[(size_t)a] == rp2[(size_t)a + 1]) fail(IBH_ENOTIMPL, "pair: row %d of the second matrix has no entries", a);
        int grp = -1;
        for (int k = rp2[(size_t)a]; k < rp2[(size_t)a + 1]; ++k) {
            int r1 = ci2[(size_t)k];
            if (!same_numbering) {
                const auto it = d1->inv.find(d2->host[(size_t)r1]);
                if (it == d1->inv.end()) fail(IBH_ENOTIMPL, "pair: a column of the second matrix is no row of the first");
                r1 = it->second;
            }
            IBH_CHECK(r1 >= 0 && r1 < first->nrow, "pair: column out of range");
            const int g = where_g[(size_t)r1], sl = where_s[(size_t)r1];
            if (g < 0 || (grp >= 0 && g != grp)) fail(IBH_ENOTIMPL, "pair: row %d of the second matrix reads rows of more than one group of the first", a);
            grp = g;
            pw[(size_t)g * IBH_GSLOTS + sl] = v2[(size_t)k];
            pm[(size_t)g] |= 1u << sl;
        }
        if (pr[(size_t)grp] >= 0) fail(IBH_ENOTIMPL, "pair: two rows of the second matrix read the same group of the first");
        pr[(size_t)grp] = a;
    }
    first->pair_w.alloc(pw.size()); first->pair_mask.alloc(pm.size()); first->pair_row.alloc(pr.size());
    IBH_HIP(hipMemcpy(first->pair_w.p, pw.data(), pw.size() * sizeof(double), hipMemcpyHostToDevice));
    IBH_HIP(hipMemcpy(first->pair_mask.p, pm.data(), pm.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    IBH_HIP(hipMemcpy(first->pair_row.p, pr.data(), pr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    first->pair_second = second;
    first->pair_uid = second->uid;
}

static unsigned pair_grid_blocks(const ibh_weighted *first, int nvar);
void spmm_launch_pair(const ibh_weighted *first, const ibh_weighted *second, const double *dA, int nvar, int64_t lda, double *dB1,
                      int64_t ldb1, double *dB2, int64_t ldb2, double fill, hipStream_t stream, unsigned *done)
{
    HandleScope hs_(first);
    IBH_CHECK(first->pair_second == second && first->pair_uid == second->uid && first->grp_n > 0,
              "pair apply: call ibh_weighted_pair_prepare(first, second, nvar) first");
    if (nvar <= 0 || first->nrow == 0) return;
    IBH_CHECK(lda >= first->ncol && ldb1 >= first->nrow && ldb2 >= second->nrow, "pair apply: leading dimensions (%ld, %ld, %ld) too small",
              (long)lda, (long)ldb1, (long)ldb2);
    IBH_CHECK(first->conservative && second->conservative, "pair apply: smoothed (non-conservative) matrices take separate applies");
    BatchPtrs bp{};
    bp.x[0] = dA; bp.y[0] = dB1;
    const PairView pv{first->pair_w.p, first->pair_mask.p, first->pair_row.p, second->wM.p, dB2, (long)ldb2, done};
    const int nw = get_tuning("rowgroup_waves", nvar >= 32 ? 8 : 4), tw = get_tuning("rowgroup_tw", first->nnz < (1 << 20) ? 32 : 64);
#define IBH_RGP(N, TT) launch_rowgroup<N, 8, TT>(first, bp, 1, nvar, (long)lda, (long)ldb1, fill, stream, &pv)
    if (use_grouptile(first, nvar)) {
        launch_grouptile_any(first, bp, 1, nvar, (long)lda, (long)ldb1, fill, stream, &pv);
    } else if (nw == 8) { if (tw == 32) IBH_RGP(8, 32); else IBH_RGP(8, 64); }
    else { if (tw == 32) IBH_RGP(4, 32); else IBH_RGP(4, 64); }
#undef IBH_RGP
    first->last_kernel = 5;
    second->last_kernel = 5;
    ++first->napply; ++second->napply;
}
// workgroups of the pair launch of (first, nvar): what the chain's last kernel waits for
static unsigned pair_grid_blocks(const ibh_weighted *first, int nvar) {
    int xcd_mode;
    if (use_grouptile(first, nvar)) return (unsigned)rowblock_grid(first->grp_n, ceil_div(nvar, 16), xcd_mode);
    const int nw = get_tuning("rowgroup_waves", nvar >= 32 ? 8 : 4);
    return (unsigned)rowblock_grid(first->grp_n, ceil_div(nvar, nw == 8 ? 8 : 4), xcd_mode);
}
// B1 = first * A, B2 = second * B1 (the fused pair), B3 = third * B2: the chain ice -> elevation classes -> atmosphere -> ice of
// BASELINE config 3 (EvI, AvE, IvA): the pair launch and the third matrix's apply, stream-ordered.
// ibh_set_tuning("chain_overlap", 1): the second launch is dispatched WITHOUT a barrier behind the first (hipExtAnyOrderLaunch) --
// its launch latency runs while the first kernel computes -- and its workgroups wait on a device flag the first kernel's last
// workgroup raises once all results are out (chain_begin / chain_signal / chain_wait; the first kernel's packet precedes it in
// the queue, so its workgroups are placed first: the waiting ones cannot starve them).  Bitwise the ordered launches, and MEASURED
// SLOWER on MI355X / ROCm 7.2 (5 km, 16 fields, back-to-back chains: 33 us per chain against 15.9 ordered; 68 us with agent-scope
// acquire fences in the waiting workgroups -- every one of them invalidates its XCD's L2 --, 43 with device-coherent loads of X,
// 33 with X staged in LDS and the flag only read; the length of the sleep in the wait loop changes nothing; under rocprofv3 the
// pair kernel lasts 28 us instead of 9 with the waiting kernel resident): off by default, kept for a runtime where an unordered
// dispatch is cheap.
void spmm_launch_chain(const ibh_weighted *first, const ibh_weighted *second, const ibh_weighted *third, const double *dA, int nvar, int64_t lda,
                       double *dB1, int64_t ldb1, double *dB2, int64_t ldb2, double *dB3, int64_t ldb3, double fill, hipStream_t stream)
{
    IBH_CHECK(third->ncol == second->nrow, "chain: the third matrix reads %d columns, the second makes %d rows", third->ncol, second->nrow);
    if (nvar <= 0 || first->nrow == 0) return;
    bool overlap = get_tuning("chain_overlap", 0) != 0 && third->conservative && third->nrow > 0;
    if (overlap) {
        HandleScope hs_(third);
        const ShortrowPlan sp = shortrow_plan(third, nvar, 1);
        overlap = pick_kernel(third, nvar, 1) == 2 && !sp.use_xt && (size_t)sp.fper * (size_t)third->ncol * sizeof(double) <= 32 * 1024;
    }
    if (!overlap) {
        spmm_launch_pair(first, second, dA, nvar, lda, dB1, ldb1, dB2, ldb2, fill, stream, nullptr);
        spmm_launch(third, dB2, nvar, ldb2, dB3, ldb3, fill, 0, stream);
        return;
    }
    if (!first->chain_cnt.p) { first->chain_cnt.alloc(2); first->chain_cnt.zero(stream); }
    unsigned nblk;
    { HandleScope hs_(first); nblk = pair_grid_blocks(first, nvar); }
    spmm_launch_pair(first, second, dA, nvar, lda, dB1, ldb1, dB2, ldb2, fill, stream, first->chain_cnt.p);
    g_chain_cnt = first->chain_cnt.p; g_chain_n = nblk;
    spmm_launch(third, dB2, nvar, ldb2, dB3, ldb3, fill, 0, stream);
    IBH_CHECK(g_chain_cnt == nullptr, "internal: the chain's last apply did not take the counter");
}

static void launch_one_impl(const ibh_weighted *w, int kernel, const BatchPtrs &bp, int nbatch, int nvar, int64_t lda,
                            int64_t ldb, double fill, hipStream_t stream);
static void launch_one(const ibh_weighted *w, int kernel, const BatchPtrs &bp, int nbatch, int nvar, int64_t lda,
                       int64_t ldb, double fill, hipStream_t stream)
{
    // launch timing (ibh_set_launch_events): the single-kernel paths attach the events to their dispatch, the I-row path
    // to its first and last one (transposed input + row kernel); bands + combine records them around the sequence
    if (kernel == 3 && g_ev_start && g_ev_stop) {
        hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
        g_ev_start = g_ev_stop = nullptr;
        IBH_HIP(hipEventRecord(ev0, stream));
        launch_one_impl(w, kernel, bp, nbatch, nvar, lda, ldb, fill, stream);
        IBH_HIP(hipEventRecord(ev1, stream));
        return;
    }
    launch_one_impl(w, kernel, bp, nbatch, nvar, lda, ldb, fill, stream);
}
static void launch_one_impl(const ibh_weighted *w, int kernel, const BatchPtrs &bp, int nbatch, int nvar, int64_t lda,
                            int64_t ldb, double fill, hipStream_t stream)
{
    if (kernel == 5) {
        // 8 waves (fields) share a staged segment from 32 fields (5 km, 64 fields: 17.1 against 19.3 us with 4)
        const int u = get_tuning("rowgroup_unroll", 8), nw = get_tuning("rowgroup_waves", nvar >= 32 ? 8 : 4);
        // class tables of half width (two lanes per entry) for the small matrices: all workgroups of a 5 km launch fit the LDS at
        // once -- 16 applies per launch 13.5 -> 12.3 us (64 fields), 3.96 -> 3.37 (16 fields), one launch unchanged (17.6 / 17.8);
        // at 1 km the doubled atomic instructions cost 223 -> 238 us.  By the matrix alone, so one apply and a batch agree bitwise.
        const int tw = get_tuning("rowgroup_tw", w->nnz < (1 << 20) ? 32 : 64);
#define IBH_RG(N, UU, TT) launch_rowgroup<N, UU, TT>(w, bp, nbatch, nvar, (long)lda, (long)ldb, fill, stream)
        if (use_grouptile(w, nvar)) {
            launch_grouptile_any(w, bp, nbatch, nvar, (long)lda, (long)ldb, fill, stream);
        } else if (tw == 32) { if (nw == 8) { if (u <= 8) IBH_RG(8, 8, 32); else IBH_RG(8, 16, 32); } else { if (u <= 4) IBH_RG(4, 4, 32); else IBH_RG(4, 8, 32); } }
        else if (nw == 8) { if (u <= 8) IBH_RG(8, 8, 64); else IBH_RG(8, 16, 64); }
        else { if (u <= 4) IBH_RG(4, 4, 64); else if (u <= 8) IBH_RG(4, 8, 64); else IBH_RG(4, 16, 64); }
#undef IBH_RG
    } else if (kernel == 4) {
        launch_sweep(w, bp, nbatch, nvar, (long)lda, (long)ldb, fill, stream);
    } else if (kernel == 3) {
        const long pairs = (long)w->nrow * nvar;
        const int fpw = get_tuning("rowdual_fpw", pairs >= 4 * 8192 ? 4 : pairs >= 2 * 8192 ? 2 : 1);
        if (fpw >= 4) launch_rowdual<4>(w, bp, nbatch, nvar, (long)lda, (long)ldb, fill, stream);
        else if (fpw == 2) launch_rowdual<2>(w, bp, nbatch, nvar, (long)lda, (long)ldb, fill, stream);
        else launch_rowdual<1>(w, bp, nbatch, nvar, (long)lda, (long)ldb, fill, stream);
    } else if (kernel == 1) {
        int fpw = get_tuning(nbatch > 1 ? "rowblock_many_fpw" : "rowblock_fpw", 0), wk = get_tuning("rowblock_wk", 0);
        if (fpw == 0 || wk == 0) {
            // enough workgroups to give every CU ~8: small problems are latency-bound and want many
            // small tasks, big ones amortise the staged row segment over more fields
            const long pairs = (long)w->nrow * nvar;
            if (nvar >= 16 && pairs >= 4 * 8192) { fpw = 4; wk = 1; }
            else if (nvar >= 8 && pairs >= 2 * 8192) { fpw = 2; wk = 1; }
            else if (nvar >= 4) { fpw = 1; wk = 1; }
            else if (nvar >= 2) { fpw = 1; wk = 2; }
            else { fpw = 1; wk = 4; }
        }
        // deep batched launches: 8 waves (8 fields) per workgroup halve the workgroup count per batch
        // (measured at the 5 km headline shape, depth 16: 7.36 against 7.51 us per apply)
        // one apply per launch, >= 8 fields, rows of a few hundred entries: the lean kernel (same bits as rowblock with wk == 1)
        const double mean_len = w->nrow ? (double)w->nnz / (double)w->nrow : 0.0;
        const int rowone = get_tuning("rowone", nbatch == 1 && wk == 1 && nvar >= 32 && mean_len >= 192.0 && mean_len <= 1024.0 ? 1 : 0);
        if (rowone && nbatch == 1 && wk == 1) {
            int u = get_tuning("rowone_unroll", mean_len > 768.0 ? 16 : 14), nw = get_tuning("rowone_waves", 8);
            const double *X1 = bp.x[0];
            double *Y1 = bp.y[0];
#define IBH_R1(N, UU) launch_rowone<N, UU>(w, X1, Y1, nvar, (long)lda, (long)ldb, fill, stream)
            if (nw == 4) { if (u <= 8) IBH_R1(4, 8); else if (u <= 12) IBH_R1(4, 12); else if (u <= 14) IBH_R1(4, 14); else IBH_R1(4, 16); }
            else { if (u <= 8) IBH_R1(8, 8); else if (u <= 12) IBH_R1(8, 12); else if (u <= 14) IBH_R1(8, 14); else IBH_R1(8, 16); }
#undef IBH_R1
            return;
        }
        const int nw = get_tuning("rowblock_waves", nbatch >= 8 && fpw == 1 && wk == 1 ? 8 : 4);
#define IBH_L(F, K, N) launch_rowblock<F, K, N>(w, bp, nbatch, nvar, (long)lda, (long)ldb, fill, stream)
        if (nw == 8 && wk == 1) {
            if (fpw == 1) IBH_L(1, 1, 8); else if (fpw == 2) IBH_L(2, 1, 8); else IBH_L(4, 1, 8);
        } else if (fpw == 4 && wk == 1) IBH_L(4, 1, 4);
        else if (fpw == 8 && wk == 1) IBH_L(8, 1, 4);
        else if (fpw == 2 && wk == 1) IBH_L(2, 1, 4);
        else if (fpw == 1 && wk == 1) IBH_L(1, 1, 4);
        else if (fpw == 4 && wk == 2) IBH_L(4, 2, 4);
        else if (fpw == 2 && wk == 2) IBH_L(2, 2, 4);
        else if (fpw == 1 && wk == 2) IBH_L(1, 2, 4);
        else if (fpw == 2 && wk == 4) IBH_L(2, 4, 4);
        else if (fpw == 4 && wk == 4) IBH_L(4, 4, 4);
        else IBH_L(1, 4, 4);
#undef IBH_L
    } else {
        const ShortrowPlan p = shortrow_plan(w, nvar, nbatch);
        const int fper = p.fper, g = p.g;
        const long xt_stride = (long)w->ncol * p.ldt;
        // GB-sized results: deep launches cost the L2 locality of the row slices (measured at 1 km: 148 us per apply alone,
        // 181 us sixteen deep); they go out a few batches at a time
        int qmax = std::max(1, get_tuning("shortrow_many", w->nrow >= (1 << 19) ? 1 : IBH_MAX_BATCH));
        if (p.use_xt && qmax > 1 && (size_t)xt_stride * sizeof(double) * (size_t)std::min(qmax, nbatch) > w->xt.granted) {
            // the transposed copies of a deep launch do not fit the scratch the handle owns: grow it -- unless the stream is
            // being captured (no allocation there): then as many batches per launch as fit
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (stream) IBH_HIP(hipStreamIsCapturing(stream, &cs));
            if (cs != hipStreamCaptureStatusNone) qmax = std::max(1, (int)(w->xt.granted / ((size_t)xt_stride * sizeof(double))));
        }
        hipEvent_t ev0 = g_ev_start, ev1 = g_ev_stop;
        g_ev_start = g_ev_stop = nullptr;
        for (int q0 = 0; q0 < nbatch; q0 += qmax) {
            const int nq = std::min(qmax, nbatch - q0);
            hipEvent_t e_first = q0 == 0 ? ev0 : nullptr, e_last = q0 + qmax >= nbatch ? ev1 : nullptr;
            BatchPtrs bq{};
            bool misaligned = (ldb & 7) != 0;
            for (int q = 0; q < nq; ++q) {
                bq.x[q] = bp.x[q0 + q]; bq.y[q] = bp.y[q0 + q];
                misaligned = misaligned || (reinterpret_cast<uintptr_t>(bq.y[q]) & 63) != 0;
            }
            if (p.use_xt) grow_scratch(w->xt, (size_t)xt_stride * (size_t)nq, stream, "transposed-input");
            // planes of B that do not start on 64-byte lines are re-aligned through LDS (see the kernel)
            const bool realign = get_tuning("shortrow_realign", -1) >= 0 ? get_tuning("shortrow_realign", -1) != 0
                               : misaligned && w->nrow >= (1 << 18);   // below: latency-bound, the two extra barriers cost more
            const long nblk = (long)ceil_div(w->nrow, realign ? SR_STEP : SR_THREADS) * ceil_div(nvar, fper);
            IBH_CHECK(nblk < (1l << 31), "spmm grid too large (%ld blocks)", nblk);
            dim3 grid((unsigned)nblk, (unsigned)nq);
            long xld = (long)lda;
            if (p.use_xt) {
                hipExtLaunchKernelGGL(transpose_fields_kernel, dim3((unsigned)ceil_div(w->ncol, 64), (unsigned)(p.ldt / 16), (unsigned)nq), dim3(256), 0, stream,
                                      e_first, nullptr, 0, bq, (long)lda, nvar, w->ncol, w->xt.p, p.ldt, xt_stride);
                e_first = nullptr;
                xld = p.ldt;
            }
            unsigned *ccnt = g_chain_cnt;
            const unsigned cn = g_chain_n;
            g_chain_cnt = nullptr; g_chain_n = 0;
            IBH_CHECK(!ccnt || (!p.use_xt && nbatch == 1), "internal: a chain's last apply must read its input in place");
            const size_t clds = ccnt ? (size_t)fper * (size_t)w->ncol * sizeof(double) : 0;
#define IBH_SR4(NT, GG, RA, XTT)                                                                                \
    hipExtLaunchKernelGGL((spmm_shortrow_kernel<NT, GG, RA, XTT>), grid, dim3(SR_THREADS), clds, stream, e_first, e_last,   \
                          ccnt ? hipExtAnyOrderLaunch : 0,                                                                  \
                          w->rowptr.p, w->colind.p, w->val.p, bq, (const double *)w->xt.p, xt_stride, xld, (long)ldb,       \
                          w->nrow, nvar, fper, w->wM.p, fill, ccnt, cn, w->ncol)
#define IBH_SR(NT, GG)                                                                                          \
    do {                                                                                                        \
        if (realign) { if (p.use_xt) IBH_SR4(NT, GG, true, true); else IBH_SR4(NT, GG, true, false); }          \
        else { if (p.use_xt) IBH_SR4(NT, GG, false, true); else IBH_SR4(NT, GG, false, false); }                \
    } while (0)
            const bool nt = get_tuning("shortrow_nt", 1) != 0;
            if (g >= 16) { if (nt) IBH_SR(true, 16); else IBH_SR(false, 16); }
            else if (g >= 8) { if (nt) IBH_SR(true, 8); else IBH_SR(false, 8); }
            else { if (nt) IBH_SR(true, 4); else IBH_SR(false, 4); }
#undef IBH_SR4
#undef IBH_SR
        }
        IBH_HIP(hipGetLastError());
    }
}

void spmm_launch_many(const ibh_weighted *w, int nbatch, const double *const *dA, int nvar, int64_t lda,
                      double *const *dB, int64_t ldb, double fill, int force_conservation, hipStream_t stream)
{
    HandleScope hs_(w);
    if (nvar <= 0 || w->nrow == 0 || nbatch <= 0) return;
    IBH_CHECK(lda >= w->ncol && ldb >= w->nrow, "apply: leading dimensions (%ld, %ld) smaller than (%d, %d)",
              (long)lda, (long)ldb, w->ncol, w->nrow);
    // An E-row matrix that is applied again gets its band structure now (every ice cell carries the weights of BOTH
    // classes it lies between and is read once instead of twice): pays for a matrix that meets many field batches, not
    // for the coupler's one build : one apply, hence on the second apply and only for bandwidth-sized work (measured, 64
    // fields: 1 km EvI 292 -> 255 us; at 5 km the extra combine pass costs more than the halved traffic saves, 18.5 -> 21.9).
    // ... or, with >= 32 fields, the column-sweep structure (sweep_kernel.inl): every X element read once, in whole lines.
    // This is the one place where an apply synchronises and allocates: ibh_weighted_prepare() does it up front,
    // ibh_set_tuning("lazy_structures", 0) switches it off.
    if (get_tuning("lazy_structures", 1)) build_structures(w, nvar, std::min(nbatch, IBH_MAX_BATCH), w->napply >= 1, stream, (long)w->napply);
    ++w->napply;
    int kernel = pick_kernel(w, nvar, nbatch);
    // the column sweep addresses a wave's 16 field planes through one buffer descriptor (32-bit offsets)
    if (kernel == 4 && ((uint64_t)16 * (uint64_t)lda * 8 + (uint64_t)w->ncol * 8 >= (1ull << 32))) kernel = w->band_n > 0 ? 3 : 1;
    w->last_kernel = kernel;
    w->last_sig[0] = 0;
    const bool correct = !w->conservative && force_conservation;
    if (correct) grow_scratch(w->consv, 2 * (size_t)nvar + weight_dot_scratch(std::max(w->nrow, w->ncol), nvar), stream, "conservation");
    for (int b0 = 0; b0 < nbatch; b0 += IBH_MAX_BATCH) {
        const int nb = std::min(IBH_MAX_BATCH, nbatch - b0);
        BatchPtrs bp{};
        for (int q = 0; q < nb; ++q) {
            IBH_CHECK(dA[b0 + q] && dB[b0 + q], "apply: null field pointer in batch %d", b0 + q);
            bp.x[q] = dA[b0 + q]; bp.y[q] = dB[b0 + q];
        }
        launch_one(w, kernel, bp, nb, nvar, lda, ldb, fill, stream);
        if (correct) {
            // factor_k = (Mw . A_k) / (wM . B_k); rows with wM == 0 hold `fill` and are skipped.  The two
            // dot products live in handle-owned scratch: stream-ordered, no allocation, no host sync.
            for (int q = 0; q < nb; ++q) {
                double *T = w->consv.p;
                weight_dot_launch(w->Mw.p, w->ncol, bp.x[q], nvar, lda, T, T + 2 * nvar, stream);
                weight_dot_launch(w->wM.p, w->nrow, bp.y[q], nvar, ldb, T + nvar, T + 2 * nvar, stream);
                dim3 grid((unsigned)ceil_div(w->nrow, 256), (unsigned)nvar);
                hipLaunchKernelGGL(conserve_scale_kernel, grid, dim3(256), 0, stream, bp.y[q], (long)ldb, w->nrow, w->wM.p,
                                   T, T + nvar);
            }
            IBH_HIP(hipGetLastError());
        }
    }
}

void spmm_launch(const ibh_weighted *w, const double *dA, int nvar, int64_t lda, double *dB, int64_t ldb,
                 double fill, int force_conservation, hipStream_t stream)
{
    spmm_launch_many(w, 1, &dA, nvar, lda, &dB, ldb, fill, force_conservation, stream);
}

// ---- legacy coo_matvec contract (pylib/icebin_cython.cpp:158-192) ----------------------------------
// yy = M xx with "REPLACE" semantics: a row is written only if at least one of its terms survives;
// with ignore_nan, terms whose input is NaN are skipped; rows with no surviving term keep whatever
// the caller put in yy (icebin.coo_multiply presets `fill`).  No wM involved.  One wave per (row,
// field) for long rows, one thread for short ones; fixed summation order.
template <bool WAVE>
__global__ void matvec_legacy_kernel(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                     const double *__restrict__ vals, const double *__restrict__ X, long ldx,
                                     double *__restrict__ Y, long ldy, int nrow, int ignore_nan)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(WAVE ? t >> 6 : t);
    const int lane = WAVE ? (int)(threadIdx.x & 63) : 0;
    if (r >= nrow) return;
    const double *x = X + (long)blockIdx.y * ldx;
    double acc = 0.0;
    int live = 0;
    for (int k = rowptr[r] + lane; k < rowptr[r + 1]; k += WAVE ? 64 : 1) {
        const double xv = x[colind[k]];
        if (ignore_nan && xv != xv) continue;
        acc = fma(vals[k], xv, acc);
        ++live;
    }
    if (WAVE) {
        acc = wave_sum(acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) live += __shfl_xor(live, off, 64);
    }
    if (lane == 0 && live > 0) Y[(long)blockIdx.y * ldy + r] = acc;
}
void matvec_legacy_launch(const ibh_weighted *w, const double *dx, int nvar, int64_t ldx, double *dy, int64_t ldy,
                          int ignore_nan, hipStream_t stream)
{
    if (nvar <= 0 || w->nrow == 0) return;
    IBH_CHECK(ldx >= w->ncol && ldy >= w->nrow, "coo_matvec: leading dimensions too small");
    const bool wave = w->nnz >= 8 * (int64_t)w->nrow;
    if (wave)
        hipLaunchKernelGGL(matvec_legacy_kernel<true>, dim3((unsigned)ceil_div((long)w->nrow * 64, 256), (unsigned)nvar), dim3(256), 0,
                           stream, w->rowptr.p, w->colind.p, w->val.p, dx, (long)ldx, dy, (long)ldy, w->nrow, ignore_nan);
    else
        hipLaunchKernelGGL(matvec_legacy_kernel<false>, dim3((unsigned)ceil_div(w->nrow, 256), (unsigned)nvar), dim3(256), 0,
                           stream, w->rowptr.p, w->colind.p, w->val.p, dx, (long)ldx, dy, (long)ldy, w->nrow, ignore_nan);
    IBH_HIP(hipGetLastError());
}

// ---- coupler-shaped product M * (V*T + b) ---------------------------------------------------
// out[k, j] = sum_l T[l,k] * in[l, j] + b[k] * (scale ? scale[j] : 1): the variable transform, applied
// on whichever side of M is small.  T entries that are exactly 0 are structural (skipped: a NaN in an
// unused input variable must not reach the outputs).  One thread per column j.
// T (row-major nvar_in x nvar_out) and b travel in the kernarg segment (TB_MAX doubles): the caller's
// host arrays are consumed at launch time, nothing is copied asynchronously from them and the call
// stays stream-ordered and graph-capturable.  Larger transforms go through a device copy.
constexpr int TB_MAX = 384;
struct TransformArgs { double tb[TB_MAX]; };
template <bool INLINE>
__global__ void transform_kernel(const double *__restrict__ in, long ldin, int nin, const TransformArgs ta,
                                 const double *__restrict__ Tdev, const double *__restrict__ scale,
                                 const double *__restrict__ wM, double fill, double *__restrict__ out, long ldout,
                                 int nout, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (j >= n) return;
    const double *Tm = INLINE ? ta.tb : Tdev;
    const double *b = Tm + (size_t)nin * nout;
    double acc = 0.0;
    for (int l = 0; l < nin; ++l) {
        const double t = Tm[l * nout + k];
        if (t != 0.0) acc = fma(t, in[(long)l * ldin + j], acc);
    }
    acc += scale ? b[k] * scale[j] : b[k];
    if (wM && wM[j] == 0.0) acc = fill;
    out[(long)k * ldout + j] = acc;
}
// rowsum1[r] = sum of row r of M (= M * 1, the factor of the offset term b): one wave per row
__global__ void rowsum_kernel(const int *__restrict__ rowptr, const double *__restrict__ vals, int nrow,
                              double *__restrict__ out)
{
    const int r = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= nrow) return;
    double acc = 0.0;
    for (int k = rowptr[r] + lane; k < rowptr[r + 1]; k += 64) acc += vals[k];
    acc = wave_sum(acc);
    if (lane == 0) out[r] = acc;
}
void ensure_rowsum1(const ibh_weighted *w, hipStream_t stream) {
    if (w->have_rowsum1 || w->nrow == 0) return;
    grow_scratch(w->rowsum1, (size_t)w->nrow, stream, "row-sum");
    hipLaunchKernelGGL(rowsum_kernel, dim3(ceil_div((long)w->nrow * 64, 256)), dim3(256), 0, stream, w->rowptr.p, w->val.p,
                       w->nrow, w->rowsum1.p);
    IBH_HIP(hipGetLastError());
    w->have_rowsum1 = true;
}

void spmm_transformed_launch(const ibh_weighted *w, const double *dA, int nvar_in, int64_t lda, const double *T,
                             const double *b, int nvar_out, double *dB, int64_t ldb, double fill, hipStream_t stream)
{
    if (nvar_out <= 0 || w->nrow == 0) return;
    IBH_CHECK(nvar_in > 0 && T && b, "apply_transformed: empty transform");
    IBH_CHECK(lda >= w->ncol && ldb >= w->nrow, "apply_transformed: leading dimensions too small");
    const size_t ntb = (size_t)nvar_in * nvar_out + nvar_out;
    const bool inl = ntb <= (size_t)TB_MAX;
    TransformArgs ta;
    const double *dT = nullptr;
    if (inl) {
        std::copy(T, T + (size_t)nvar_in * nvar_out, ta.tb);
        std::copy(b, b + nvar_out, ta.tb + (size_t)nvar_in * nvar_out);
    } else {
        // more than TB_MAX coefficients: staged through a device copy (synchronises; not capturable)
        std::vector<double> tb(ntb);
        std::copy(T, T + (size_t)nvar_in * nvar_out, tb.begin());
        std::copy(b, b + nvar_out, tb.begin() + (size_t)nvar_in * nvar_out);
        grow_scratch(w->tbuf, ntb, stream, "transform-coefficient");
        IBH_HIP(hipStreamSynchronize(stream));      // an earlier apply may still read tbuf
        IBH_HIP(hipMemcpyAsync(w->tbuf.p, tb.data(), sizeof(double) * ntb, hipMemcpyHostToDevice, stream));
        IBH_HIP(hipStreamSynchronize(stream));      // tb is a stack-lifetime staging buffer
        dT = w->tbuf.p;
    }
    auto transform = [&](const double *in, long ldin, const double *scale, const double *wM, double *out, long ldout, int n) {
        dim3 grid((unsigned)ceil_div(n, 256), (unsigned)nvar_out);
        if (inl) hipLaunchKernelGGL(transform_kernel<true>, grid, dim3(256), 0, stream, in, ldin, nvar_in, ta, dT, scale, wM, fill, out, ldout, nvar_out, n);
        else hipLaunchKernelGGL(transform_kernel<false>, grid, dim3(256), 0, stream, in, ldin, nvar_in, ta, dT, scale, wM, fill, out, ldout, nvar_out, n);
        IBH_HIP(hipGetLastError());
    };
    if (w->ncol <= w->nrow) {
        // inputs are the small side: X' = V*T + b, then B = M * X'
        grow_scratch(w->scratch, (size_t)nvar_out * (size_t)w->ncol, stream, "transform");
        transform(dA, (long)lda, nullptr, nullptr, w->scratch.p, (long)w->ncol, w->ncol);
        spmm_launch(w, w->scratch.p, nvar_out, w->ncol, dB, ldb, fill, 0, stream);
    } else {
        // outputs are the small side: Z = M * V, then B = T^T Z + b * (M * 1), rows with wM == 0 -> fill
        ensure_rowsum1(w, stream);
        grow_scratch(w->scratch, (size_t)nvar_in * (size_t)w->nrow, stream, "transform");
        spmm_launch(w, dA, nvar_in, lda, w->scratch.p, w->nrow, 0.0, 0, stream);
        transform(w->scratch.p, (long)w->nrow, w->rowsum1.p, w->wM.p, dB, (long)ldb, w->nrow);
    }
}

}  // namespace ibh
