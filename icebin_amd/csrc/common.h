// common.h -- internal helpers of libicebin_hip.so (error channel, device buffers, handle layouts).
#pragma once
#include <atomic>
#include <mutex>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/icebin_hip.h"

namespace ibh {

// ---- error channel -------------------------------------------------------------------------
struct Error : std::exception {
    int code;
    std::string msg;
    Error(int c, std::string m) : code(c), msg(std::move(m)) {}
    const char *what() const noexcept override { return msg.c_str(); }
};
[[noreturn]] void fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void set_last_error(const char *msg);

#define IBH_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            ::ibh::fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? IBH_ENODEVICE : IBH_EHIP, \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define IBH_CHECK(cond, ...)                                  \
    do {                                                      \
        if (!(cond)) ::ibh::fail(IBH_EINVAL, __VA_ARGS__);    \
    } while (0)

// Wrap a C-ABI body: exceptions -> status code + thread-local message.
template <class F>
int guarded(F &&f) noexcept {
    try {
        f();
        return IBH_OK;
    } catch (const Error &e) {
        set_last_error(e.msg.c_str());
        return e.code;
    } catch (const std::bad_alloc &) {
        set_last_error("out of host memory");
        return IBH_EINVAL;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return IBH_EINVAL;
    }
}

// ---- device memory -------------------------------------------------------------------------
// Caching allocator (capi.hip): freed blocks are kept per device and size class and handed out again,
// so that rebuilding the same matrices every coupling step does no hipMalloc/hipFree (both
// synchronise the device).  ibh_release_cached_memory() returns everything to the driver.
void *dev_alloc(size_t bytes, size_t *granted, int *device);
void dev_free(void *p, size_t granted, int device);     // filed under the device the block was allocated on
void release_cached_memory();

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t granted = 0;     // bytes actually reserved (size class)
    int device = 0;         // device the block lives on (the current one at alloc time)
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), granted(o.granted), device(o.device) { o.p = nullptr; o.n = 0; o.granted = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; granted = o.granted; device = o.device; o.p = nullptr; o.n = 0; o.granted = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        if (p && count * sizeof(T) <= granted) { n = count; return; }
        release();
        n = count;
        p = static_cast<T *>(dev_alloc((count ? count : 1) * sizeof(T), &granted, &device));
    }
    void release() {
        if (p) dev_free(p, granted, device);
        p = nullptr; n = 0; granted = 0;
    }
    void upload(const T *host, size_t count, hipStream_t s = nullptr) {
        if (count > n || !p) alloc(count);
        if (count) IBH_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T *host, size_t count, hipStream_t s = nullptr) const {
        if (count) IBH_HIP(hipMemcpyAsync(host, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
        IBH_HIP(hipStreamSynchronize(s));
    }
    void zero(hipStream_t s = nullptr) { if (n) IBH_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
constexpr int IBH_GSLOTS = 32;      // most rows (elevation classes of one GCM cell) in a row group (spmm.hip rowgroup)
// tiles of a row group (spmm.hip grouptile): seg = items (distinct columns) of a tile, 256 or 128 by the matrix (ibh_weighted::gt_seg)
constexpr int ibh_gt_ecap(int seg) { return 2 * seg + 4 * IBH_GSLOTS; }   // entries of a tile: <= 2 per item, every slot's list padded to a multiple of 4
constexpr int IBH_GT_EP = IBH_GSLOTS + 2;   // u16 entry offsets of a tile: one per slot + the end, padded to whole dwords
inline int bits_for(uint64_t n) {   // bits needed to represent values in [0, n)
    int b = 0;
    while (b < 64 && (n > (1ull << b))) ++b;
    return b;
}

}  // namespace ibh

// ---- handle layouts (opaque to the C-ABI user) ----------------------------------------------
struct ibh_sparse_set {
    int64_t sparse_extent = -1;
    int32_t n = 0;                       // dense extent
    bool identity = false;               // to_sparse[i] == i for i < n; neither copy is materialised until asked for
    // dense -> sparse table in first-seen order.  A matrix build appends on the DEVICE; the host copy
    // is completed lazily (ensure_host) so that a 10^7-entry dims table never crosses PCIe unless a
    // caller actually reads it.
    mutable std::vector<int64_t> host;   // entries [0, host_n) valid
    mutable int32_t host_n = 0;
    ibh::DevBuf<int64_t> dev;            // entries [0, dev_n) valid
    int32_t dev_n = 0;
    ibh::DevBuf<int32_t> tab;            // device sparse -> dense table (-1 missing) of entries [0, tab_n), built on demand
    int32_t tab_n = -1;
    int64_t tab_extent = -1;
    mutable std::unordered_map<int64_t, int32_t> inv;    // sparse -> dense of entries [0, inv_n): host-side to_dense / add_dense
    mutable int32_t inv_n = 0;
    void ensure_inverse() const {
        ensure_host();
        for (int32_t i = inv_n; i < n; ++i) inv[host[(size_t)i]] = i;
        inv_n = n;
    }
    int32_t dense_extent() const { return n; }
    void ensure_host() const {
        if (host_n >= n) return;
        host.resize((size_t)n);
        if (identity) {
            for (int32_t i = host_n; i < n; ++i) host[(size_t)i] = i;
        } else {
            if (dev_n < n) ibh::fail(IBH_EINVAL, "internal: sparse set has no valid copy of entries [%d,%d)", host_n, n);
            IBH_HIP(hipMemcpy(host.data() + host_n, dev.p + host_n, sizeof(int64_t) * (size_t)(n - host_n), hipMemcpyDeviceToHost));
        }
        host_n = n;
    }
};

// Mask-independent structure of one exchange grid (fastasm.inl): built once, on the device, the first
// time a matrix of this sheet is assembled.  Only for grids sorted by (iA, iI) -- the order
// ExchangeGrid's constructor leaves them in (AbbrGrid.cpp:10-21).
struct ibh_plan {
    bool tried = false, ok = false;
    int32_t nAr = 0;                     // atmosphere cells with exchange cells ("ranges" of consecutive x)
    ibh::DevBuf<int32_t> arng;           // [nAr+1] first exchange cell of every range
    ibh::DevBuf<int32_t> aidx;           // [nX] range of every exchange cell
    ibh::DevBuf<int32_t> ilptr, ilist;   // [nI+1], [nX] exchange cells of every ice cell, ascending
    ibh::DevBuf<int32_t> ifirst;         // [nI] first exchange cell of the ice cell with area != 0, -1: none
    ibh::DevBuf<uint8_t> isdup;          // [nX] bit 0: same (iA, iI) as the cell before; bit 1: this is the ONLY exchange cell of
                                         //      its ice cell (the ice cell straddles no GCM-cell edge and has no duplicates); bit 2: this
                                         //      cell is the first-seen exchange cell of its ice cell (ifirst[iI] == x); bits 3 / 4: area > 0 /
                                         //      area != 0 and not > 0
    ibh::DevBuf<int32_t> mlist;          // [nmulti] ice cells with more than one exchange cell (several ranges, or duplicates)
    int32_t nmulti = 0;
    // [nI] matrix entries an UNMASKED ice cell has per elevation-class slot: groups of duplicate exchange cells (same (iA, iI))
    // with a member of area > 0 (IvA: rows of GvAp) / area != 0 (IvE: rows of GvI); static, so the I-row builds count their
    // rows without visiting the exchange cells.  Empty when an area is so small that area * weight could underflow
    // (the count would then depend on the elevation): the builds count by visiting, as before.
    ibh::DevBuf<uint8_t> icnt_pos, icnt_nz;
    // [nX] the ice-cell index of every exchange cell on its own (ex_indices interleaves it with the atmosphere index, which is
    // constant over a range): the per-range kernels and the streamed count read 4 bytes per cell instead of 8
    ibh::DevBuf<int32_t> exI;
    // for the streamed build (streamasm.inl): isdup bits 5 (first cell of its range), 6 (its ice cell was first seen in an
    // EARLIER range: a straddler) and 7 (the next cell is a duplicate of this one); the atmosphere cell of every range; the
    // longest range (positions inside a range's block of entries are stored in 16 bits when they fit); an area so small that
    // area * class weight could underflow was seen (the streamed elevation-class builds count by sign classes: not for it)
    ibh::DevBuf<int32_t> riA;
    int32_t maxrange = 0;
    bool tiny = false;
    // the ice cells with several exchange cells, split for the streamed build: PAIRS -- exactly two exchange cells, in two ranges,
    // the first one the first-seen cell (an ice cell across one GCM-cell edge: nearly all of them) -- as three parallel arrays
    // (first cell, second cell, ice cell), served by a lean kernel without list walks; the rest (corners, duplicates) in mlist3
    ibh::DevBuf<int32_t> px1, px2, piI, mlist3;
    ibh::DevBuf<double> pa1, pa2;        // the (static) overlap areas of a pair's two cells, beside them: no gathers in k_sa_pairs
    int32_t npair = 0, nmulti3 = 0;
    std::vector<int32_t> arng_h;         // host copy of arng (the sharded build deals ranges to ranks), filled on first use
};

struct ibh_regridder {
    int device = 0;
    mutable ibh_plan plan;
    int64_t nX = 0, nI = 0, nA = 0;
    int32_t nA_dense = 0, nhc = 0, interp_style = 0;
    int64_t hc_stride_A = 1, hc_stride_HC = 0;
    ibh::DevBuf<int32_t> ex_indices;    // [2*nX]
    ibh::DevBuf<double> ex_area;        // [nX]
    ibh::DevBuf<double> hcdefs;         // [nhc]
    ibh::DevBuf<double> A_ratio_s;      // [nA] native/proj by SPARSE A index, 0 where A cell not realised
    std::vector<int64_t> A_to_sparse;   // host copies for wA()
    std::vector<double> A_native, A_proj, hcdefs_h;
    double hc_last = 0;
    ibh::DevBuf<double> I_centroid;     // [2*nI] (x,y) by sparse ice index; empty when not supplied
    bool has_centroid = false;
    double cmin[2] = {0, 0}, cmax[2] = {0, 0};   // bounding box of the centroids
};

struct ibh_regrid_matrices {
    const ibh_regridder *rg = nullptr;
    ibh::DevBuf<double> elevmaskI;      // [nI] copy (RegridMatrices_Dynamic.cpp:352)
    int scale = 1, correctA = 0;
    double sigma[3] = {0, 0, 0};
    // one byte per ice cell, a function of its elevation and the regridder's elevation classes alone (ibh::elevmask_classes,
    // streamasm.inl): 0xFF masked, 0xFE elevation beyond the last class (linterp_1d_b's error), else first class | classes << 6.
    // Made once per elevmask -- at creation for grids the streamed build serves, else by the first build that wants it -- and read by
    // every matrix of this object in place of the 8-byte elevation wherever only the mask or the class pattern matters.
    mutable ibh::DevBuf<uint8_t> em_cls;
    mutable int em_cls_nhc = -1, em_cls_interp = -1;
    mutable std::mutex em_cls_mu;
};

namespace ibh {
inline uint64_t next_weighted_uid() {
    static std::atomic<uint64_t> n{0};
    return ++n;
}
}  // namespace ibh
struct ibh_weighted {
    const uint64_t uid = ibh::next_weighted_uid();     // never reused: a pairing names its second matrix by (address, uid)
    int device = 0;
    int32_t nrow = 0, ncol = 0;
    int64_t nnz = 0;
    ibh::DevBuf<int32_t> rowptr, colind;
    ibh::DevBuf<double> val, wM, Mw;
    ibh_sparse_set *dims[2] = {nullptr, nullptr};
    bool owns[2] = {false, false};
    int conservative = 1, scaled = 1;
    int built_fast = 0;                 // assembled by the plan-based fast path (fastasm.inl); introspection only
    // SpMM dispatch
    int kernel_override = 0;            // 0 auto, 1 rowblock, 2 shortrow, 3 rowdual
    mutable char last_sig[64] = {0};    // the kernel instantiation the last apply launched, as rocprofv3 names it (rowblock / rowone; else the kernel family)
    // per-handle launch options (ibh_weighted_set_option): looked up before the process-wide ibh_set_tuning map by every apply of
    // THIS matrix, so two host threads tuning different handles do not interfere
    mutable std::unordered_map<std::string, int> opts;
    mutable int last_kernel = 0;
    // apply_transformed: scratch fields + small transform, and M*1 (row sums) for the offset term
    mutable ibh::DevBuf<double> scratch, tbuf, rowsum1;
    mutable ibh::DevBuf<double> consv;  // force_conservation: the two dot products per variable [2*nvar] + chunk sums
    mutable ibh::DevBuf<double> xt;     // shortrow: transposed copy of the (small) input fields
    // rowdual (EvI, EvX): the CSR filtered to one entry per (GCM cell, ice cell) carrying the weights of
    // BOTH elevation classes the cell lies between (assemble.hip build_bands); band r = row r
    mutable int64_t band_n = 0;         // number of band entries, 0: not built
    mutable ibh::DevBuf<int32_t> band_ptr, band_col, band_rb1;   // [nrow+1]; [band_n] column | bit30 lower exists | bit31 upper exists; [nrow]
    mutable ibh::DevBuf<double> band_v0, band_v1;                // [band_n] lower-class / upper-class weight
    // E-row matrices over ice / exchange columns can get the band structure later, from their CSR (ensure_bands):
    // how the row keys decode into (GCM cell, class), and how many applies the matrix has seen
    int band_eligible = 0;
    int64_t band_sA = 0, band_sHC = 0;
    mutable int band_tried = 0;
    mutable int64_t napply = 0;
    mutable ibh::DevBuf<double> band_part;               // per-apply partial sums [2][nvar][nrow padded]
    // colsweep (EvI, EvX; sweep_kernel.inl): the entries in column order, paired per column into items, 64 items a block,
    // tb blocks a task with its local row table; built lazily from the CSR (assemble.hip build_sweep_from_csr)
    mutable int sweep_tried = 0;
    mutable int32_t sweep_ntask = 0, sweep_nblk = 0, sweep_nprow = 0, sweep_nslot = 0;   // ntask == 0: not built
    mutable int32_t sweep_tb = 0, sweep_nitems = 0, sweep_ident = 0;                      // blocks per task, items, column == item index
    mutable ibh::DevBuf<int32_t> sweep_task_p0, sweep_task_ns, sweep_col;
    mutable ibh::DevBuf<uint32_t> sweep_meta;
    mutable ibh::DevBuf<double> sweep_v0, sweep_v1;
    mutable ibh::DevBuf<int32_t> sweep_comb_ptr, sweep_comb_p;     // [nrow+1], [nprow]: the partial-sum rows that make up row r, in task order
    mutable ibh::DevBuf<double> sweep_part;              // per-apply partial sums [nbatch][nprow][fields padded to 64]
    // rowgroup (EvI, EvX; spmm.hip): the rows of one GCM cell (its elevation classes) form a GROUP; the group's entries are
    // listed once per distinct column, ascending, as ITEMS {col, meta = slot0 | slot1 << 8 | has-bits, v0, v1} -- the column
    // set of a group is the AvI row of its GCM cell, so X is gathered once per (GCM cell, ice cell) instead of once per class;
    // slot s of group g is row grp_slotrow[g * IBH_GSLOTS + s].  Built from the CSR (assemble.hip build_groups_from_csr).
    mutable int grp_tried = 0;
    // fused pair (ibh_weighted_pair_prepare): a second matrix whose every row reads rows of ONE group of this matrix only (AvE
    // after EvI: a GCM cell's value is a combination of that cell's elevation classes) rides in the row-group kernel's epilogue.
    // pair_w[g * IBH_GSLOTS + s]: the second matrix's weight of slot s of group g (pair_mask[g] bit s: it has one);
    // pair_row[g]: the second matrix's row fed by group g, -1 none.
    mutable const ibh_weighted *pair_second = nullptr;
    mutable uint64_t pair_uid = 0;
    mutable ibh::DevBuf<double> pair_w;
    mutable ibh::DevBuf<uint32_t> pair_mask;
    mutable ibh::DevBuf<int32_t> pair_row;
    mutable ibh::DevBuf<uint32_t> chain_cnt;             // chain (ibh_weighted_apply_chain_device): {workgroups of the pair kernel done, of the last kernel through}
    mutable int32_t grp_n = 0, grp_nslot = 0, grp_nitems = 0;          // groups (0: not built), most rows in a group, items
    mutable ibh::DevBuf<int32_t> grp_ptr, grp_ns, grp_slotrow, grp_col; // [grp_n+1] items of a group; [grp_n] rows of a group; [grp_n*IBH_GSLOTS]; [nitems]
    mutable ibh::DevBuf<uint32_t> grp_meta;
    mutable ibh::DevBuf<double> grp_v0, grp_v1;
    // grouptile (spmm.hip): the items of a group cut into TILES of gt_seg columns; per tile the columns (padded by repeating
    // the last) and the entries sorted by (slot, column) as {8 x local item index, weight} with u16 offsets per slot -- the rows of
    // the CSR restricted to the tile, in CSR order.  Entries of tile t start at ibh_gt_ecap(gt_seg) * t.
    mutable int32_t gt_ntile = 0, gt_seg = 0;            // 0: not built
    mutable ibh::DevBuf<int32_t> gt_ptr, gt_col;         // [grp_n+1] tiles of a group; [gt_ntile * gt_seg]
    mutable ibh::DevBuf<uint16_t> gt_ek, gt_eptr;        // [gt_ntile * ecap]; [gt_ntile * IBH_GT_EP]
    mutable ibh::DevBuf<double> gt_ev;                   // [gt_ntile * ecap]
    mutable bool have_rowsum1 = false;
    mutable ibh::DevBuf<int32_t> rowperm;                // rows by descending length (batched rowblock launches)
    mutable bool have_rowperm = false;
    ~ibh_weighted() {
        for (int k = 0; k < 2; ++k)
            if (owns[k]) delete dims[k];
    }
};

struct ibh_exgrid {
    int64_t nX = 0;
    ibh::DevBuf<int32_t> indices;       // [2*nX] interleaved (iA, iI), sorted by (iA, iI)
    ibh::DevBuf<double> overlaps;       // [nX]
};

namespace ibh {
// spmm.hip
// comm.hip: the exchanges of the sharded assembly (streamasm.inl), enqueued on the caller's stream
int comm_world(const ibh_comm *c);
int comm_rank(const ibh_comm *c);
void comm_exchange_blocks(ibh_comm *c, double *base, int64_t count, int64_t stride, hipStream_t st);
void comm_gatherv(ibh_comm *c, int n, void *const *bases, const int64_t *const *offs, hipStream_t st);
void spmm_launch(const ibh_weighted *w, const double *dA, int nvar, int64_t lda, double *dB, int64_t ldb,
                 double fill, int force_conservation, hipStream_t stream);
void spmm_launch_many(const ibh_weighted *w, int nbatch, const double *const *dA, int nvar, int64_t lda,
                      double *const *dB, int64_t ldb, double fill, int force_conservation, hipStream_t stream);
void weighted_reserve(const ibh_weighted *w, int nvar);
void weighted_prepare(const ibh_weighted *w, int nvar, int nbatch);
void weighted_pair_prepare(const ibh_weighted *first, const ibh_weighted *second, int nvar);
void spmm_launch_pair(const ibh_weighted *first, const ibh_weighted *second, const double *dA, int nvar, int64_t lda, double *dB1,
                      int64_t ldb1, double *dB2, int64_t ldb2, double fill, hipStream_t stream, unsigned *done = nullptr);
void spmm_launch_chain(const ibh_weighted *first, const ibh_weighted *second, const ibh_weighted *third, const double *dA, int nvar, int64_t lda,
                       double *dB1, int64_t ldb1, double *dB2, int64_t ldb2, double *dB3, int64_t ldb3, double fill, hipStream_t stream);
// assemble.hip: the band structure of an E-row matrix from its CSR (same result as building it with the matrix)
void build_bands_from_csr(const ibh_weighted *w, hipStream_t st);
// assemble.hip: the column-sweep structure of an E-row matrix from its CSR (sweep_kernel.inl); false: not representable
bool build_sweep_from_csr(const ibh_weighted *w, hipStream_t st);
// assemble.hip: the row-group structure of an E-row matrix from its CSR (spmm.hip rowgroup); false: not representable
bool build_groups_from_csr(const ibh_weighted *w, hipStream_t st);
void matvec_legacy_launch(const ibh_weighted *w, const double *dx, int nvar, int64_t ldx, double *dy, int64_t ldy,
                          int ignore_nan, hipStream_t stream);
void spmm_transformed_launch(const ibh_weighted *w, const double *dA, int nvar_in, int64_t lda, const double *T,
                              const double *b, int nvar_out, double *dB, int64_t ldb, double fill, hipStream_t stream);
void weight_dot_launch(const double *dw, int n, const double *dA, int nvar, int64_t lda, double *dout, double *part,
                       hipStream_t stream);
size_t weight_dot_scratch(int n, int nvar);
void set_launch_events(hipEvent_t start, hipEvent_t stop);
int get_tuning(const char *key, int dflt);
void set_tuning(const char *key, int value);
}  // namespace ibh
