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

namespace ibh {

// ---- tuning knobs (ibh_set_tuning) ---------------------------------------------------------
static std::unordered_map<std::string, int> &tuning() {
    static std::unordered_map<std::string, int> t;
    return t;
}
int get_tuning(const char *key, int dflt) {
    auto it = tuning().find(key);
    return it == tuning().end() ? dflt : it->second;
}
void set_tuning(const char *key, int value) { tuning()[key] = value; }

// ---- helpers -------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).  Give each
// XCD a contiguous range of logical ids, so neighbours in logical order share an L2.
__device__ __forceinline__ int xcd_contiguous(int b, int nb) {
    const int q = nb >> 3, rem = nb & 7;
    const int x = b & 7, slot = b >> 3;
    const int base = x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q;
    return base + slot;
}

constexpr int RB_THREADS = 256;
constexpr int RB_SEG = 2048;    // nnz staged per pass: 24 KB of LDS

// FPW = fields per wave, WK = waves that split the nnz range of the row.
// The 4 waves form WF = 4/WK groups over fields; a block covers FB = FPW*WF fields.
template <int FPW, int WK, int UNROLL>
__global__ __launch_bounds__(RB_THREADS) void spmm_rowblock_kernel(
    const int *__restrict__ rowptr, const int *__restrict__ colind, const double *__restrict__ vals,
    const double *__restrict__ X, long ldx, double *__restrict__ Y, long ldy, int nrow, int nf, int nfc,
    const double *__restrict__ wM, double fill)
{
    constexpr int WF = 4 / WK;
    constexpr int FB = FPW * WF;
    __shared__ int s_col[RB_SEG];
    __shared__ double s_val[RB_SEG];
    __shared__ double s_part[WK][FB];

    const int nb = nrow * nfc;
    const int logical = xcd_contiguous(blockIdx.x, nb);
    const int r = logical / nfc;
    const int fc = logical - r * nfc;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wf = wave % WF, wk = wave / WF;
    const int f0 = fc * FB + wf * FPW;

    const double *xf[FPW];
#pragma unroll
    for (int j = 0; j < FPW; ++j) {
        int f = f0 + j < nf ? f0 + j : nf - 1;    // clamp: tail fields read valid memory, never stored
        xf[j] = X + (long)f * ldx;
    }
    double acc[FPW];
#pragma unroll
    for (int j = 0; j < FPW; ++j) acc[j] = 0.0;

    const int beg = rowptr[r], end = rowptr[r + 1];
    constexpr int STEP = WK * 64;
    for (int seg = beg; seg < end; seg += RB_SEG) {
        const int n = min(RB_SEG, end - seg);
        if (seg != beg) __syncthreads();
        for (int k = threadIdx.x; k < n; k += RB_THREADS) {
            s_col[k] = colind[seg + k];
            s_val[k] = vals[seg + k];
        }
        __syncthreads();
        // Uniform batches of UNROLL*STEP entries: every lane issues all UNROLL*FPW loads of a batch
        // before the first FMA, so a row costs ceil(n / (UNROLL*STEP)) memory round trips.  Lanes past
        // the end re-read the last entry and are masked at the FMA (never multiplied by 0: 0*NaN).
        for (int k0 = wk * 64 + lane; k0 - lane < n; k0 += UNROLL * STEP) {
            int c[UNROLL];
            double v[UNROLL];
            bool ok[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int k = k0 + u * STEP;
                ok[u] = k < n;
                const int kk = ok[u] ? k : n - 1;
                c[u] = s_col[kk];
                v[u] = s_val[kk];
            }
            double x[FPW][UNROLL];
#pragma unroll
            for (int j = 0; j < FPW; ++j)
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) x[j][u] = xf[j][c[u]];
#pragma unroll
            for (int j = 0; j < FPW; ++j)
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) acc[j] = ok[u] ? fma(v[u], x[j][u], acc[j]) : acc[j];
        }
    }
#pragma unroll
    for (int j = 0; j < FPW; ++j) acc[j] = wave_sum(acc[j]);

    const bool dead = wM[r] == 0.0;     // mask_result, IceCoupler.cpp:186-201
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
            for (int q = 1; q < WK; ++q) s += s_part[q][threadIdx.x];
            if (f < nf) Y[(long)f * ldy + r] = dead ? fill : s;
        }
    }
}

constexpr int SR_THREADS = 256;

__global__ __launch_bounds__(SR_THREADS) void spmm_shortrow_kernel(
    const int *__restrict__ rowptr, const int *__restrict__ colind, const double *__restrict__ vals,
    const double *__restrict__ X, long ldx, double *__restrict__ Y, long ldy, int nrow, int nf, int fper,
    const double *__restrict__ wM, double fill)
{
    const int r = blockIdx.x * SR_THREADS + threadIdx.x;
    if (r >= nrow) return;
    const int fbeg = blockIdx.y * fper;
    const int fend = min(nf, fbeg + fper);
    const int beg = rowptr[r], end = rowptr[r + 1];
    const int n = end - beg;
    const bool dead = wM[r] == 0.0;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (n > 0) { c0 = colind[beg]; v0 = vals[beg]; }
    if (n > 1) { c1 = colind[beg + 1]; v1 = vals[beg + 1]; }
    if (n > 2) { c2 = colind[beg + 2]; v2 = vals[beg + 2]; }
    if (n > 3) { c3 = colind[beg + 3]; v3 = vals[beg + 3]; }
    for (int f = fbeg; f < fend; ++f) {
        const double *xf = X + (long)f * ldx;
        double acc = 0.0;
        // predicated, never multiplied by a padded zero: 0*NaN must not leak into a row
        if (n > 0) acc = v0 * xf[c0];
        if (n > 1) acc = fma(v1, xf[c1], acc);
        if (n > 2) acc = fma(v2, xf[c2], acc);
        if (n > 3) acc = fma(v3, xf[c3], acc);
        for (int k = beg + 4; k < end; ++k) acc = fma(vals[k], xf[colind[k]], acc);
        __builtin_nontemporal_store(dead ? fill : acc, &Y[(long)f * ldy + r]);
    }
}

// ---- conservation correction (non-conservative matrices only) ------------------------------
// out[k] = sum_j w[j] * A[k*lda + j] over j with w[j] != 0; one block per variable, fixed order.
__global__ __launch_bounds__(1024) void weight_dot_kernel(const double *__restrict__ w, int n,
                                                          const double *__restrict__ A, long lda,
                                                          double *__restrict__ out)
{
    __shared__ double s[16];
    const double *a = A + (long)blockIdx.x * lda;
    double acc = 0.0;
    for (int j = threadIdx.x; j < n; j += 1024) {
        const double wj = w[j];
        if (wj != 0.0) acc = fma(wj, a[j], acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += s[q];
        out[blockIdx.x] = t;
    }
}
__global__ void conserve_scale_kernel(double *__restrict__ Y, long ldy, int nrow, const double *__restrict__ wM,
                                      const double *__restrict__ TA, const double *__restrict__ TB)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (r >= nrow || wM[r] == 0.0) return;
    Y[(long)k * ldy + r] *= TA[k] / TB[k];
}

void weight_dot_launch(const double *dw, int n, const double *dA, int nvar, int64_t lda, double *dout,
                       hipStream_t stream)
{
    if (nvar <= 0) return;
    hipLaunchKernelGGL(weight_dot_kernel, dim3(nvar), dim3(1024), 0, stream, dw, n, dA, (long)lda, dout);
    IBH_HIP(hipGetLastError());
}

// ---- dispatch ------------------------------------------------------------------------------
template <int FPW, int WK>
static void launch_rowblock(const ibh_weighted *w, const double *dA, int nvar, long lda, double *dB, long ldb,
                            double fill, hipStream_t stream)
{
    constexpr int FB = FPW * (4 / WK);
    const int nfc = ceil_div(nvar, FB);
    const long nb = (long)w->nrow * nfc;
    IBH_CHECK(nb < (1l << 31), "spmm grid too large (%ld blocks)", nb);
    const int unroll = get_tuning("rowblock_unroll", 8);
#define IBH_RB(U)                                                                                        \
    hipLaunchKernelGGL((spmm_rowblock_kernel<FPW, WK, U>), dim3((unsigned)nb), dim3(RB_THREADS), 0, stream, \
                       w->rowptr.p, w->colind.p, w->val.p, dA, lda, dB, ldb, w->nrow, nvar, nfc, w->wM.p, fill)
    if (unroll == 1) IBH_RB(1);
    else if (unroll == 2) IBH_RB(2);
    else if (unroll == 6) IBH_RB(6);
    else if (unroll == 8) IBH_RB(8);
    else if (unroll == 12) IBH_RB(12);
    else IBH_RB(4);
#undef IBH_RB
    IBH_HIP(hipGetLastError());
}

void spmm_launch(const ibh_weighted *w, const double *dA, int nvar, int64_t lda, double *dB, int64_t ldb,
                 double fill, int force_conservation, hipStream_t stream)
{
    if (nvar <= 0 || w->nrow == 0) return;
    IBH_CHECK(lda >= w->ncol && ldb >= w->nrow, "apply: leading dimensions (%ld, %ld) smaller than (%d, %d)",
              (long)lda, (long)ldb, w->ncol, w->nrow);
    int kernel = w->kernel_override;
    if (kernel == 0) {
        const double mean = w->nrow ? (double)w->nnz / (double)w->nrow : 0.0;
        kernel = mean >= (double)get_tuning("rowblock_min_mean_nnz", 12) ? 1 : 2;
    }
    w->last_kernel = kernel;
    if (kernel == 1) {
        int fpw = get_tuning("rowblock_fpw", 0), wk = get_tuning("rowblock_wk", 0);
        if (fpw == 0 || wk == 0) {
            if (nvar >= 8) { fpw = 2; wk = 1; }
            else if (nvar >= 4) { fpw = 1; wk = 1; }
            else if (nvar >= 2) { fpw = 1; wk = 2; }
            else { fpw = 1; wk = 4; }
        }
        if (fpw == 4 && wk == 1) launch_rowblock<4, 1>(w, dA, nvar, lda, dB, ldb, fill, stream);
        else if (fpw == 8 && wk == 1) launch_rowblock<8, 1>(w, dA, nvar, lda, dB, ldb, fill, stream);
        else if (fpw == 2 && wk == 1) launch_rowblock<2, 1>(w, dA, nvar, lda, dB, ldb, fill, stream);
        else if (fpw == 1 && wk == 1) launch_rowblock<1, 1>(w, dA, nvar, lda, dB, ldb, fill, stream);
        else if (fpw == 4 && wk == 2) launch_rowblock<4, 2>(w, dA, nvar, lda, dB, ldb, fill, stream);
        else if (fpw == 2 && wk == 2) launch_rowblock<2, 2>(w, dA, nvar, lda, dB, ldb, fill, stream);
        else if (fpw == 1 && wk == 2) launch_rowblock<1, 2>(w, dA, nvar, lda, dB, ldb, fill, stream);
        else if (fpw == 2 && wk == 4) launch_rowblock<2, 4>(w, dA, nvar, lda, dB, ldb, fill, stream);
        else if (fpw == 4 && wk == 4) launch_rowblock<4, 4>(w, dA, nvar, lda, dB, ldb, fill, stream);
        else launch_rowblock<1, 4>(w, dA, nvar, lda, dB, ldb, fill, stream);
    } else {
        int fper = get_tuning("shortrow_fper", 16);
        if (fper < 1) fper = 1;
        dim3 grid((unsigned)ceil_div(w->nrow, SR_THREADS), (unsigned)ceil_div(nvar, fper));
        hipLaunchKernelGGL(spmm_shortrow_kernel, grid, dim3(SR_THREADS), 0, stream, w->rowptr.p, w->colind.p,
                           w->val.p, dA, (long)lda, dB, (long)ldb, w->nrow, nvar, fper, w->wM.p, fill);
        IBH_HIP(hipGetLastError());
    }
    if (!w->conservative && force_conservation) {
        // factor_k = (Mw . A_k) / (wM . B_k); rows with wM == 0 hold `fill` and are skipped
        DevBuf<double> T(2 * (size_t)nvar);
        weight_dot_launch(w->Mw.p, w->ncol, dA, nvar, lda, T.p, stream);
        weight_dot_launch(w->wM.p, w->nrow, dB, nvar, ldb, T.p + nvar, stream);
        dim3 grid((unsigned)ceil_div(w->nrow, 256), (unsigned)nvar);
        hipLaunchKernelGGL(conserve_scale_kernel, grid, dim3(256), 0, stream, dB, (long)ldb, w->nrow, w->wM.p,
                           T.p, T.p + nvar);
        IBH_HIP(hipGetLastError());
        IBH_HIP(hipStreamSynchronize(stream));   // T is freed on return
    }
}

}  // namespace ibh
