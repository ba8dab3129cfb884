// assemble.hip -- K2-K4: regrid-matrix assembly on gfx950 (COO -> CSR, weights, scaling).
//
// Replaces, for IceRegridder_L0 exchange grids, the whole of compute_AEvI /
// compute_IvAE / compute_EvA (RegridMatrices_Dynamic.cpp:50-332): the Ur-matrix
// generators GvAp / GvI / GvEp (IceRegridder_L0.cpp:100-214), spsparse's
// first-seen dense numbering (MakeDenseEigenT ... ADD_DENSE), Eigen's
// setFromTriplets, the sparse*diag*sparse products, sum(), and the diagonal
// scalings.  For an L0 exchange grid every exchange cell x has exactly one
// atmosphere cell and one ice cell, so each product collapses to a keyed sum
//     T[r,c] = sum over x ASCENDING of  fl( fl(lhs_x * sinv_x) * rhs_x )
// (the order and rounding of Eigen's conservative sparse product, see
// oracle/icebin_oracle.c), which is computed here as: emit contributions in x
// order -> stable radix sort by (row_d, col_d) -> sequential sum per segment.
// Row sums run sequentially along CSR rows (ascending column) and column sums
// along a stable by-column reordering (ascending row): the orders of spsparse
// sum().  All of it is index/byte work bound by HBM traffic; no float atomics,
// no FMA contraction (this file is compiled with -ffp-contract=off), so the
// CSR structure, dims, wM, Mw and M are bit-identical to the oracle.
#include "assemble.h"
#include "prims.h"
#include "sweep_kernel.inl"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace ibh {

enum { LIST_AP = 0, LIST_I = 1, LIST_EP = 2 };
enum { KEY_A = 0, KEY_I = 1, KEY_E = 2, KEY_X = 3 };
enum { FAM_AEVI = 0, FAM_IVAE = 1, FAM_EVA = 2 };
// contribution formulas: t = fl(fl(lhs * sinv) * rhs)
enum { TERM_PLAIN_A = 0,   // a                       (AvX, XvA)
       TERM_PLAIN_E,       // vE                      (EvX, XvE)
       TERM_A_A,           // lhs=a,  sinv=1/a,  rhs=a    (AvI, IvA), needs x in GvAp and GvI
       TERM_E_A,           // lhs=vE, sinv=1/a,  rhs=a    (EvI needs GvI; EvA needs GvAp)
       TERM_A_E };         // lhs=a,  sinv=1/rs, rhs=vE   (IvE needs GvI; AvE needs GvAp)

struct MatSpec {
    const char *name;
    int family;
    int row_list, row_key, col_list, col_key;
    int term;
    int need_list;     // extra list the contributing x must belong to (LIST_AP / LIST_I), -1 none
    int epx;           // contributions per x: 1 or 2 (2 when GvEp entries are enumerated)
};
static const MatSpec SPECS[] = {
    // name  family    rows               cols               term          need     epx
    {"AvI", FAM_AEVI, LIST_AP, KEY_A, LIST_I,  KEY_I, TERM_A_A,     LIST_I,  1},
    {"EvI", FAM_AEVI, LIST_EP, KEY_E, LIST_I,  KEY_I, TERM_E_A,     LIST_I,  2},
    {"AvX", FAM_AEVI, LIST_AP, KEY_A, LIST_AP, KEY_X, TERM_PLAIN_A, -1,      1},
    {"EvX", FAM_AEVI, LIST_EP, KEY_E, LIST_EP, KEY_X, TERM_PLAIN_E, -1,      2},
    {"IvA", FAM_IVAE, LIST_I,  KEY_I, LIST_AP, KEY_A, TERM_A_A,     LIST_I,  1},
    {"IvE", FAM_IVAE, LIST_I,  KEY_I, LIST_EP, KEY_E, TERM_A_E,     LIST_I,  2},
    {"XvA", FAM_IVAE, LIST_AP, KEY_X, LIST_AP, KEY_A, TERM_PLAIN_A, -1,      1},
    {"XvE", FAM_IVAE, LIST_EP, KEY_X, LIST_EP, KEY_E, TERM_PLAIN_E, -1,      2},
    {"EvA", FAM_EVA,  LIST_EP, KEY_E, LIST_AP, KEY_A, TERM_E_A,     LIST_AP, 2},
    {"AvE", FAM_EVA,  LIST_AP, KEY_A, LIST_EP, KEY_E, TERM_A_E,     LIST_AP, 2},
};

struct RgView {
    const int32_t *exi;
    const double *area, *em, *hc;
    const double *ratioA;   // by sparse A index; 0 = A cell not realised
    long nX, nA;
    int nhc, interp;
    long sA, sHC;
};

// ---- per-exchange-cell generators (device restatement of IceRegridder_L0.cpp:100-214) ----------
struct XCell {
    long iA, iI;
    double a;
    bool unmasked, inAp, inI;
    int nep;            // GvEp entries actually emitted (0..2)
    long iE0, iE1;      // scalars, not arrays: dynamically indexed private arrays live in scratch memory
    double vE0, vE1;
    double rsE;         // sum(GvEp row x): one entry -> v, two -> fl(v0 + v1)
    bool range_error;
};

__device__ __forceinline__ int dev_lower_bound(const double *__restrict__ xp, int n, double xx) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (xp[mid] < xx) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// hcdefs staged in LDS for the elevation-class generators (binary search per cell); falls back to
// global memory for very long tables
constexpr int HC_LDS = 512;
template <bool WITH_EP>
__device__ __forceinline__ void stage_hc(RgView &rg, double *s_hc) {
    if (WITH_EP && rg.nhc <= HC_LDS) {
        for (int i = threadIdx.x; i < rg.nhc; i += blockDim.x) s_hc[i] = rg.hc[i];
        __syncthreads();
        rg.hc = s_hc;
    }
}
// the arithmetic of a cell from its four inputs (callers that stage the loads of several cells themselves use this form)
// (hint: the result of the class search when the caller already has it -- the upper class index of Z_INTERP, the class of
// ELEV_CLASS_INTERP; < 0: search)
template <bool WITH_EP>
__device__ __forceinline__ XCell make_cell(const RgView &rg, long iA, long iI, double a, double e, int hint = -1) {
    XCell c;
    c.iA = iA;
    c.iI = iI;
    c.a = a;
    c.unmasked = !(e != e);                       // !std::isnan, IceRegridder_L0.cpp:121,186,208
    c.inAp = c.unmasked && c.a > 0;               // :208-209
    c.inI = c.unmasked && c.a != 0;               // :186-187 (+ include_zero=false)
    c.nep = 0;
    c.iE0 = c.iE1 = -1;
    c.vE0 = c.vE1 = 0.0;
    c.rsE = 0.0;
    c.range_error = false;
    if (WITH_EP && c.unmasked) {
        const double elevation = e < 0.0 ? 0.0 : e;          // std::max(elev, 0.0), :123
        if (rg.interp == 0) {                                // Z_INTERP, :127-145
            int i1 = hint >= 0 ? hint : dev_lower_bound(rg.hc, rg.nhc, elevation);
            if (i1 <= 0) i1 = 1;
            if (i1 >= rg.nhc) { c.range_error = true; return c; }      // :84-85
            const int i0 = i1 - 1;
            const double ratio = (elevation - rg.hc[i0]) / (rg.hc[i1] - rg.hc[i0]);
            const double w0 = 1.0 - ratio, w1 = ratio;
            bool h0 = false, h1 = false;
            double v0 = 0.0, v1 = 0.0;
            if (w0 != 0) { v0 = c.a * w0; h0 = v0 != 0; }
            if (w1 != 0) { v1 = c.a * w1; h1 = v1 != 0; }
            const long k0 = c.iA * rg.sA + (long)i0 * rg.sHC, k1 = c.iA * rg.sA + (long)i1 * rg.sHC;
            if (h0) {
                c.iE0 = k0; c.vE0 = v0; c.nep = 1;
                if (h1) { c.iE1 = k1; c.vE1 = v1; c.nep = 2; }
            } else if (h1) {
                c.iE0 = k1; c.vE0 = v1; c.nep = 1;
            }
        } else {                                             // ELEV_CLASS_INTERP, :146-150 + nearest_1d :43-67
            const int n = rg.nhc;
            int i1 = hint >= 0 ? 0 : dev_lower_bound(rg.hc, n, elevation);
            int ih;
            if (hint >= 0) ih = hint;
            else if (i1 <= 0) ih = 0;
            else if (i1 >= n) ih = n - 1;
            else {
                const int i0 = i1 - 1;
                const double d0 = fabs(elevation - rg.hc[i0]), d1 = fabs(rg.hc[i1] - elevation);
                ih = d0 <= d1 ? i0 : i1;
            }
            if (c.a != 0) { c.iE0 = c.iA * rg.sA + (long)ih * rg.sHC; c.vE0 = c.a; c.nep = 1; }
        }
        if (c.nep == 1) c.rsE = c.vE0;
        else if (c.nep == 2) c.rsE = c.vE0 + c.vE1;
    }
    return c;
}
// the class search of make_cell for one elevation, to be handed back to it as `hint` by a caller that evaluates several exchange
// cells of the same ice cell
__device__ __forceinline__ int cell_hint(const RgView &rg, double e) {
    if (e != e || rg.nhc < 1) return -1;
    const double elevation = e < 0.0 ? 0.0 : e;
    const int i1 = dev_lower_bound(rg.hc, rg.nhc, elevation);
    if (rg.interp == 0) return i1;
    const int n = rg.nhc;
    if (i1 <= 0) return 0;
    if (i1 >= n) return n - 1;
    const int i0 = i1 - 1;
    const double d0 = fabs(elevation - rg.hc[i0]), d1 = fabs(rg.hc[i1] - elevation);
    return d0 <= d1 ? i0 : i1;
}
template <bool WITH_EP>
__device__ __forceinline__ XCell load_cell(const RgView &rg, long x) {
    const long iA = rg.exi[2 * x], iI = rg.exi[2 * x + 1];
    return make_cell<WITH_EP>(rg, iA, iI, rg.area[x], rg.em[iI]);
}

// entries of `list` at cell c: count and sparse keys of kind `key`
__device__ __forceinline__ int list_entries(const XCell &c, long x, int list, int key, long &k0, long &k1) {
    int n;
    if (list == LIST_AP) n = c.inAp ? 1 : 0;
    else if (list == LIST_I) n = c.inI ? 1 : 0;
    else n = c.nep;
    const long base = key == KEY_A ? c.iA : key == KEY_I ? c.iI : x;
    k0 = key == KEY_E ? c.iE0 : base;
    k1 = key == KEY_E ? c.iE1 : base;
    // an X key is emitted once per list entry; duplicates are harmless to first-seen numbering
    return n;
}

// ---- dense numbering (spsparse::SparseSet::add_dense in emission order) ------------------------
__global__ void k_fill_i32(int32_t *p, size_t n, int32_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_iota_i64(int64_t *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int64_t)i;
}
__global__ void k_scatter_existing(int32_t *tab, const int64_t *to_sparse, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[to_sparse[i]] = i;
}

// The generator passes handle BOTH sets of a matrix (rows and columns) per exchange-cell visit:
//   pass 1 (k_first2)  first[key] = smallest emission position 2x+j naming the key (atomicMin)
//   pass 2 (k_flag2)   pk[x] = which of the cell's entries are first occurrences + its number of
//                      contributions; ONE packed scan then ranks rows, columns and contributions
//   pass 3 (k_contrib_emit) dense id of any key = old id, or base + rank of its first occurrence
//                      (looked up through first[]); first occurrences also record to_sparse.
struct SetArgs {
    int enabled, list, key, base;
    long ident_n;               // >= 0: the set's old part is the identity on [0, ident_n)
    const int32_t *tab;         // old sparse -> dense table (-1 missing); nullptr when there is no old part / identity
    uint32_t *first;            // [sparse_extent] smallest emission position 2x+j that names the key
    int64_t *to_sparse;         // [capacity] dense -> sparse
    const uint32_t *off;        // [nX] new keys of this set emitted before cell x (after the scan)
    int pkshift;                // this set's flag pair inside pk[x]
};
__device__ __forceinline__ int old_dense(const SetArgs &a, long key) {
    if (a.ident_n >= 0) return key < a.ident_n ? (int)key : -1;
    return a.tab ? a.tab[key] : -1;
}

// Keys shared by many cells (atmosphere cells, elevation classes: ~10^2..10^3 cells name each one)
// are de-duplicated inside the wave first: lanes holding the same key elect the lowest lane, whose
// position is the smallest (positions grow with the lane), and only that lane issues the global
// atomicMin.  Device-scope atomics are resolved beyond the per-XCD L2; a 38.8 M-cell grid would
// otherwise issue 7.7e7 of them on ~10^5 addresses.  Keys that are (almost) unique per cell (ice
// cells, exchange cells) go to memory directly.
__device__ __forceinline__ bool shared_keys(const SetArgs &a) { return a.key == KEY_A || a.key == KEY_E; }
__device__ __forceinline__ void first_entry(const SetArgs &a, long key, bool has, uint32_t pos, bool dedupe, int lane) {
    if (!dedupe) {
        if (has) atomicMin(&a.first[key], pos);
        return;
    }
    unsigned long long todo = __ballot(has);          // every lane takes part in the ballots
    bool is_leader = false;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const long k = __shfl(key, leader, 64);
        const unsigned long long same = __ballot(has && key == k);
        if (lane == leader) is_leader = true;
        todo &= ~same;
    }
    // ONE atomic instruction for all elected lanes
    if (is_leader) atomicMin(&a.first[key], pos);
}
__device__ __forceinline__ void first_one(const SetArgs &a, const XCell &c, long x, bool valid) {
    long k0 = -1, k1 = -1;
    const int n = valid ? list_entries(c, x, a.list, a.key, k0, k1) : 0;
    const bool dedupe = shared_keys(a);
    const int lane = threadIdx.x & 63;
    first_entry(a, k0, n > 0 && old_dense(a, k0) < 0, (uint32_t)(2 * x), dedupe, lane);
    if (a.list == LIST_EP)          // uniform: only elevation-class lists have second entries
        first_entry(a, k1, n > 1 && old_dense(a, k1) < 0, (uint32_t)(2 * x + 1), dedupe, lane);
}
template <bool WITH_EP>
__global__ __launch_bounds__(256) void k_first2(RgView rg, SetArgs a, SetArgs b, uint32_t *__restrict__ err_x) {
    __shared__ double s_hc[WITH_EP ? HC_LDS : 1];
    stage_hc<WITH_EP>(rg, s_hc);
    const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
    XCell c;
    bool valid = false;                         // no early return: every lane takes part in the ballots
    if (x < rg.nX) {
        c = load_cell<WITH_EP>(rg, x);
        if (WITH_EP && c.range_error) atomicMin(err_x, (uint32_t)x);
        else valid = true;
    }
    if (a.enabled) first_one(a, c, x, valid);
    if (b.enabled) first_one(b, c, x, valid);
}

__device__ __forceinline__ uint32_t flag_one(const SetArgs &a, const XCell &c, long x) {
    long k0, k1;
    const int n = list_entries(c, x, a.list, a.key, k0, k1);
    uint32_t f = 0;
    if (n > 0 && old_dense(a, k0) < 0 && a.first[k0] == (uint32_t)(2 * x)) f |= 1u;
    if (n > 1 && old_dense(a, k1) < 0 && a.first[k1] == (uint32_t)(2 * x + 1)) f |= 2u;
    return f;
}
// dense id of `key` (any cell may ask): old id, else base + rank of the key's first occurrence
__device__ __forceinline__ int dense_of(const SetArgs &a, long key, const uint32_t *__restrict__ pk) {
    if (!a.enabled) return (int)key;                       // identity over the whole sparse extent
    const int t = old_dense(a, key);
    if (t >= 0) return t;
    const uint32_t p = a.first[key];
    const uint32_t xf = p >> 1;
    uint32_t r = a.off[xf];
    if (p & 1u) r += (pk[xf] >> a.pkshift) & 1u;           // second entry of its cell: after the first if that one was new too
    return a.base + (int)r;
}
// the cell's own first occurrences record dense -> sparse
__device__ __forceinline__ void record_new(const SetArgs &a, const XCell &c, long x, uint32_t pkx) {
    const uint32_t f = (pkx >> a.pkshift) & 3u;
    if (!f) return;
    long k0, k1;
    (void)list_entries(c, x, a.list, a.key, k0, k1);
    const uint32_t r = a.off[x];
    if (f & 1u) a.to_sparse[a.base + r] = k0;
    if (f & 2u) a.to_sparse[a.base + r + (f & 1u)] = k1;
}

struct DeviceSet {
    int64_t *to_sparse = nullptr; // [capacity] dense -> sparse
    int n_old = 0, n = 0;
};

// Host side of the dense numbering of one set (spsparse::SparseSet::add_dense in emission order,
// appending to whatever the caller's set already holds): prepare() builds the device tables,
// the fused kernels above do the work, finish() adopts the new keys once the host knows how many.
struct Numbering {
    DeviceSet ds;
    ibh_sparse_set *set = nullptr;
    SetArgs args{};
    int64_t max_new = 0;
};

static Numbering number_set_prepare(const RgView &rg, ibh_sparse_set *set, int64_t sparse_extent, int list, int key,
                                    int64_t max_new, int pkshift, uint32_t *first_storage, hipStream_t st) {
    Arena &A = arena();
    set->sparse_extent = sparse_extent;                       // set_sparse_extent, RegridMatrices_Dynamic.cpp:69-72
    Numbering nb;
    nb.set = set;
    DeviceSet &ds = nb.ds;
    ds.n_old = set->dense_extent();
    if (set->identity) {
        IBH_CHECK(ds.n_old <= sparse_extent, "identity dims larger than sparse extent");
    } else if (set->host_n == ds.n_old) {
        for (int32_t i = 0; i < ds.n_old; ++i)
            IBH_CHECK(set->host[(size_t)i] >= 0 && set->host[(size_t)i] < sparse_extent, "dims entry %ld outside sparse extent %ld",
                      (long)set->host[(size_t)i], (long)sparse_extent);
    }
    IBH_CHECK(sparse_extent < 0xffffffffll, "sparse extent %ld does not fit 32 bits", (long)sparse_extent);
    if (max_new > sparse_extent - ds.n_old) max_new = sparse_extent - ds.n_old;
    nb.max_new = max_new;
    const int64_t cap = (int64_t)ds.n_old + max_new;
    ds.to_sparse = A.get<int64_t>((size_t)cap);
    const int T = 256;
    SetArgs &a = nb.args;
    a.list = list; a.key = key; a.base = ds.n_old; a.to_sparse = ds.to_sparse; a.pkshift = pkshift;
    a.ident_n = -1; a.tab = nullptr;
    if (ds.n_old) {
        if (set->identity) {
            a.ident_n = ds.n_old;
            hipLaunchKernelGGL(k_iota_i64, dim3(ceil_div(ds.n_old, T)), dim3(T), 0, st, ds.to_sparse, (size_t)ds.n_old);
        } else {
            if (set->dev_n == ds.n_old)
                IBH_HIP(hipMemcpyAsync(ds.to_sparse, set->dev.p, sizeof(int64_t) * (size_t)ds.n_old, hipMemcpyDeviceToDevice, st));
            else {
                set->ensure_host();
                IBH_HIP(hipMemcpyAsync(ds.to_sparse, set->host.data(), sizeof(int64_t) * (size_t)ds.n_old, hipMemcpyHostToDevice, st));
            }
            int32_t *tab = A.get<int32_t>((size_t)sparse_extent);
            hipLaunchKernelGGL(k_fill_i32, dim3(ceil_div(sparse_extent, T)), dim3(T), 0, st, tab, (size_t)sparse_extent, -1);
            hipLaunchKernelGGL(k_scatter_existing, dim3(ceil_div(ds.n_old, T)), dim3(T), 0, st, tab, ds.to_sparse, ds.n_old);
            a.tab = tab;
        }
    }
    ds.n = ds.n_old;
    a.enabled = !(set->identity && ds.n_old == sparse_extent);      // an identity set that covers everything gains nothing
    a.first = first_storage;            // [sparse_extent], preset to 0xFFFFFFFF by the caller (one fill for both sets)
    return nb;
}

static void number_set_finish(Numbering &nb, uint32_t n_new, hipStream_t st) {
    DeviceSet &ds = nb.ds;
    ibh_sparse_set *set = nb.set;
    if (!nb.args.enabled || n_new == 0) return;
    ds.n = ds.n_old + (int)n_new;
    // the set keeps its table on the device; the host copy is completed only when somebody asks
    DevBuf<int64_t> grown((size_t)ds.n);
    IBH_HIP(hipMemcpyAsync(grown.p, ds.to_sparse, sizeof(int64_t) * (size_t)ds.n, hipMemcpyDeviceToDevice, st));
    if (set->identity) { set->host.clear(); set->host_n = 0; }     // an identity prefix was never materialised
    set->dev = std::move(grown);
    set->dev_n = set->n = ds.n;
    set->identity = false;
}

// ---- contributions ---------------------------------------------------------------------------
struct Contrib { long r0, r1, c0, c1; double t0, t1; };
__device__ __forceinline__ int contributions(const XCell &c, long x, const MatSpec &s, Contrib &o) {
    if (s.need_list == LIST_I && !c.inI) return 0;
    if (s.need_list == LIST_AP && !c.inAp) return 0;
    switch (s.term) {
        case TERM_PLAIN_A:
            if (!c.inAp) return 0;
            o.r0 = s.row_key == KEY_X ? x : c.iA; o.c0 = s.col_key == KEY_X ? x : c.iA; o.t0 = c.a;
            return 1;
        case TERM_PLAIN_E:
            o.r0 = s.row_key == KEY_X ? x : c.iE0; o.c0 = s.col_key == KEY_X ? x : c.iE0; o.t0 = c.vE0;
            o.r1 = s.row_key == KEY_X ? x : c.iE1; o.c1 = s.col_key == KEY_X ? x : c.iE1; o.t1 = c.vE1;
            return c.nep;
        case TERM_A_A: {
            if (!c.inAp) return 0;
            const double sinv = 1.0 / c.a;
            const double l = c.a * sinv;
            o.r0 = s.row_key == KEY_A ? c.iA : c.iI; o.c0 = s.col_key == KEY_A ? c.iA : c.iI; o.t0 = l * c.a;
            return 1;
        }
        case TERM_E_A: {      // rows E; cols I (EvI) or A (EvA)
            const double sinv = 1.0 / c.a;
            const long ck = s.col_key == KEY_A ? c.iA : c.iI;
            const double l0 = c.vE0 * sinv, l1 = c.vE1 * sinv;
            o.r0 = c.iE0; o.c0 = ck; o.t0 = l0 * c.a;
            o.r1 = c.iE1; o.c1 = ck; o.t1 = l1 * c.a;
            return c.nep;
        }
        default: {            // TERM_A_E: cols E; rows I (IvE) or A (AvE)
            if (c.nep == 0) return 0;
            const double sinv = 1.0 / c.rsE;
            const double l = c.a * sinv;
            const long rk = s.row_key == KEY_A ? c.iA : c.iI;
            o.r0 = rk; o.c0 = c.iE0; o.t0 = l * c.vE0;
            o.r1 = rk; o.c1 = c.iE1; o.t1 = l * c.vE1;
            return c.nep;
        }
    }
}

template <bool WITH_EP>
__global__ void k_flag2(RgView rg, SetArgs a, SetArgs b, MatSpec s, uint32_t *__restrict__ pk) {
    __shared__ double s_hc[WITH_EP ? HC_LDS : 1];
    stage_hc<WITH_EP>(rg, s_hc);
    const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= rg.nX) return;
    const XCell c = load_cell<WITH_EP>(rg, x);
    uint32_t v = 0;
    if (!(WITH_EP && c.range_error)) {
        if (a.enabled) v |= flag_one(a, c, x) << a.pkshift;
        if (b.enabled) v |= flag_one(b, c, x) << b.pkshift;
        Contrib o;
        v |= (uint32_t)contributions(c, x, s, o) << 4;
    }
    pk[x] = v;
}
template <bool WITH_EP>
__global__ void k_contrib_emit(RgView rg, MatSpec s, SetArgs a, SetArgs b, const uint32_t *__restrict__ pk,
                               const uint32_t *__restrict__ pos, uint64_t *__restrict__ keys, uint32_t *__restrict__ idx,
                               double *__restrict__ term) {
    __shared__ double s_hc[WITH_EP ? HC_LDS : 1];
    stage_hc<WITH_EP>(rg, s_hc);
    const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= rg.nX) return;
    const uint32_t pkx = pk[x];
    if (!pkx) return;                          // nothing new, nothing contributed (masked cells, range errors)
    const XCell c = load_cell<WITH_EP>(rg, x);
    if (a.enabled) record_new(a, c, x, pkx);
    if (b.enabled) record_new(b, c, x, pkx);
    Contrib o;
    const int n = contributions(c, x, s, o);
    const uint32_t p = pos[x];
    if (n == 0) return;
    uint64_t q0 = ((uint64_t)(uint32_t)dense_of(a, o.r0, pk) << 32) | (uint32_t)dense_of(b, o.c0, pk);
    if (n > 1) {
        // The two entries of a cell name different elevation classes, so they are never summed
        // together and may be emitted in either order: ascending key keeps the sequence closer to
        // sorted (the first-seen class numbering is not monotone in the class index).
        uint64_t q1 = ((uint64_t)(uint32_t)dense_of(a, o.r1, pk) << 32) | (uint32_t)dense_of(b, o.c1, pk);
        if (q1 < q0) { const uint64_t tq = q0; q0 = q1; q1 = tq; const double tt = o.t0; o.t0 = o.t1; o.t1 = tt; }
        keys[p + 1] = q1; idx[p + 1] = p + 1; term[p + 1] = o.t1;
    }
    keys[p] = q0; idx[p] = p; term[p] = o.t0;
}

// ---- sorted contributions -> unique CSR entries ------------------------------------------------
__global__ void k_head_flags(const uint64_t *__restrict__ keys, size_t n, uint32_t *__restrict__ head,
                             uint32_t *__restrict__ unsorted) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t key = keys[k], prev = k ? keys[k - 1] : 0;
    head[k] = (k == 0 || key != prev) ? 1u : 0u;
    if (unsorted && k && key < prev) *unsorted = 1u;        // verification of an optimistic (high-field-only) sort
}
// Duplicate (row,col) contributions are summed in sorted (== emission) order, the first term
// ASSIGNED (Eigen's product and setFromTriplets both start from the first value, which keeps a
// signed zero): k_segment_heads lists the segment boundaries, seg_sums<false> does the chains.
__global__ void k_segment_heads(const uint64_t *__restrict__ keys, size_t n, const uint32_t *__restrict__ upos,
                                uint32_t nseg, int32_t *__restrict__ segptr, int32_t *__restrict__ row,
                                int32_t *__restrict__ col) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t key = keys[k];
    if (k == 0) segptr[nseg] = (int32_t)n;
    if (k != 0 && keys[k - 1] == key) return;
    const uint32_t u = upos[k];
    segptr[u] = (int32_t)k;
    row[u] = (int32_t)(key >> 32);
    col[u] = (int32_t)(key & 0xffffffffu);
}
// rowptr from the (sorted) row index of every entry.  Rows that own entries get their start from the
// entry that opens them; rows WITHOUT entries (a caller-supplied identity set spans every ice cell,
// most of them masked: millions of consecutive empty rows) find theirs by binary search, each on
// its own thread -- a per-entry loop over the gap in front of it would serialise on exactly those runs.
__global__ void k_rowptr_heads(const int32_t *__restrict__ row, long nnz, int32_t *__restrict__ rowptr) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nnz) return;
    const int r = row[u];
    if (u == 0 || row[u - 1] != r) rowptr[r] = (int32_t)u;
}
__global__ void k_rowptr_fill(const int32_t *__restrict__ row, long nnz, int nrow, int32_t *__restrict__ rowptr) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nrow) return;
    if (q == nrow) { rowptr[q] = (int32_t)nnz; return; }
    if (rowptr[q] >= 0) return;
    long lo = 0, hi = nnz;                       // first entry whose row is >= q
    while (lo < hi) {
        const long mid = lo + ((hi - lo) >> 1);
        if (row[mid] < q) lo = mid + 1; else hi = mid;
    }
    rowptr[q] = (int32_t)lo;
}
__global__ void k_rowptr_search(const int32_t *__restrict__ row, long nnz, int nrow, int32_t *__restrict__ rowptr) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nrow) return;
    long lo = 0, hi = nnz;                       // first entry whose row is >= q
    while (lo < hi) {
        const long mid = lo + ((hi - lo) >> 1);
        if (row[mid] < q) lo = mid + 1; else hi = mid;
    }
    rowptr[q] = (int32_t)lo;
}
static void rowptr_from_rows(const int32_t *row, long nnz, int nrow, int32_t *rowptr, hipStream_t st) {
    if (nrow < (1 << 16)) {                      // few rows: one launch, every row searches (the small builds are launch-bound)
        hipLaunchKernelGGL(k_rowptr_search, dim3(ceil_div(nrow + 1, 256)), dim3(256), 0, st, row, nnz, nrow, rowptr);
        return;
    }
    IBH_HIP(hipMemsetAsync(rowptr, 0xFF, sizeof(int32_t) * ((size_t)nrow + 1), st));
    if (nnz) hipLaunchKernelGGL(k_rowptr_heads, dim3(ceil_div(nnz, 256)), dim3(256), 0, st, row, nnz, rowptr);
    hipLaunchKernelGGL(k_rowptr_fill, dim3(ceil_div(nrow + 1, 256)), dim3(256), 0, st, row, nnz, nrow, rowptr);
}
// spsparse sum(M,dim,'+'): one sequential chain per segment (a CSR row in ascending column order,
// or a column in ascending row order through the stable by-column permutation `idx`), exactly
// ret = ret + v starting from 0.  Thread-per-segment suits 1-3 entry segments; the wave form
// loads 64 values coalesced and replays the same sequential chain from registers (every lane
// computes the identical sum), which is what long rows (10^2..10^4 entries) need.
template <bool FROM_ZERO>
__global__ void k_seg_sums_thread(const int32_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
                                  const double *__restrict__ val, int nseg, double *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nseg) return;
    int k = ptr[r];
    const int end = ptr[r + 1];
    double s = 0.0;
    if (!FROM_ZERO && k < end) { s = val[idx ? idx[k] : (uint32_t)k]; ++k; }
    for (; k < end; ++k) s = s + val[idx ? idx[k] : (uint32_t)k];
    out[r] = s;
}
template <bool FROM_ZERO>
__global__ void k_seg_sums_wave(const int32_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
                                const double *__restrict__ val, int nseg, double *__restrict__ out) {
    const int r = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= nseg) return;
    const int beg = ptr[r], end = ptr[r + 1];
    double s = 0.0;
    for (int base = beg; base < end; base += 64) {
        const int k = base + lane;
        const double v = k < end ? val[idx ? idx[k] : (uint32_t)k] : 0.0;
        const int cnt = min(64, end - base);
        const int lo = __double2loint(v), hi = __double2hiint(v);
        int j = 0;
        if (!FROM_ZERO && base == beg) {      // first term assigned
            s = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
            j = 1;
        }
        if (cnt == 64 && j == 0) {
#pragma unroll
            for (int q = 0; q < 64; ++q)
                s = s + __hiloint2double(__builtin_amdgcn_readlane(hi, q), __builtin_amdgcn_readlane(lo, q));
        } else {
            for (; j < cnt; ++j)
                s = s + __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));
        }
    }
    if (lane == 0) out[r] = s;
}
template <bool FROM_ZERO>
static void seg_sums(const int32_t *ptr, const uint32_t *idx, const double *val, int nseg, long nnz, double *out,
                     hipStream_t st) {
    if (nseg == 0) return;
    if (nnz >= 8l * nseg)
        hipLaunchKernelGGL(k_seg_sums_wave<FROM_ZERO>, dim3(ceil_div((long)nseg * 64, 256)), dim3(256), 0, st, ptr, idx, val, nseg, out);
    else
        hipLaunchKernelGGL(k_seg_sums_thread<FROM_ZERO>, dim3(ceil_div(nseg, 256)), dim3(256), 0, st, ptr, idx, val, nseg, out);
}
__global__ void k_col_keys(const int32_t *__restrict__ col, long nnz, uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < nnz) { keys[u] = (uint64_t)(uint32_t)col[u]; idx[u] = (uint32_t)u; }
}
__global__ void k_keys_to_i32(const uint64_t *__restrict__ keys, long n, int32_t *__restrict__ out) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n) out[u] = (int32_t)keys[u];
}

// ---- column sums for matrices with short columns (AvI, EvI, AvX, EvX, AvE: <= a few rows per column) ----
// spsparse sum(M, 1, '+') visits a column in ascending row order.  Instead of reordering all nnz
// entries by column (three radix passes), the entries are dropped into per-column slots
// (integer atomics: which slot an entry gets is not deterministic, the SET in a column is) and each
// column is summed by one thread: 1 or 2 entries need no order at all ((0+a)+b == (0+b)+a bit for
// bit); longer columns pick their entries in ascending row order by repeated selection (rows are
// unique inside a column).  Columns of more than 64 entries go to a wave each.  Only used when
// nnz <= 4*ncol, so long columns are the exception (an ice cell under a GCM-cell corner: 4).
__global__ void k_col_count(const int32_t *__restrict__ col, long nnz, uint32_t *__restrict__ cnt) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < nnz) atomicAdd(&cnt[col[u]], 1u);
}
__global__ void k_col_scatter(const int32_t *__restrict__ row, const int32_t *__restrict__ col, long nnz,
                              const uint32_t *__restrict__ colptr, uint32_t *__restrict__ fillc,
                              int32_t *__restrict__ lrow, uint32_t *__restrict__ lidx) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nnz) return;
    const int c = col[u];
    const uint32_t p = colptr[c] + atomicAdd(&fillc[c], 1u);
    lrow[p] = row[u];
    lidx[p] = (uint32_t)u;
}
__global__ void k_col_sums(const uint32_t *__restrict__ colptr, int ncol, const int32_t *__restrict__ lrow,
                           const uint32_t *__restrict__ lidx, const double *__restrict__ val, double *__restrict__ cs,
                           uint32_t *__restrict__ nlong, int32_t *__restrict__ longcols) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const uint32_t b = colptr[c], e = colptr[c + 1];
    const uint32_t n = e - b;
    double s = 0.0;
    if (n <= 2) {
        if (n > 0) s = s + val[lidx[b]];
        if (n > 1) s = s + val[lidx[b + 1]];
    } else if (n <= 64) {
        int prev = -1;
        for (uint32_t k = 0; k < n; ++k) {
            int best = 0x7fffffff; uint32_t at = b;
            for (uint32_t j = b; j < e; ++j) {
                const int r = lrow[j];
                if (r > prev && r < best) { best = r; at = j; }
            }
            s = s + val[lidx[at]];
            prev = best;
        }
    } else {
        longcols[atomicAdd(nlong, 1u)] = c;
        return;
    }
    cs[c] = s;
}
__global__ void k_col_sums_long(const uint32_t *__restrict__ colptr, const int32_t *__restrict__ lrow,
                                const uint32_t *__restrict__ lidx, const double *__restrict__ val, double *__restrict__ cs,
                                const uint32_t *__restrict__ nlong, const int32_t *__restrict__ longcols) {
    const int lane = threadIdx.x & 63;
    const uint32_t nl = *nlong;
    for (uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; q < nl; q += (gridDim.x * blockDim.x) >> 6) {
        const int c = longcols[q];
        const uint32_t b = colptr[c], e = colptr[c + 1];
        double s = 0.0;
        int prev = -1;
        for (uint32_t k = b; k < e; ++k) {            // one selection round per entry, 64 candidates per step
            int best = 0x7fffffff; uint32_t at = b;
            for (uint32_t j = b + lane; j < e; j += 64) {
                const int r = lrow[j];
                if (r > prev && r < best) { best = r; at = j; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const int ob = __shfl_xor(best, off, 64); const uint32_t oa = __shfl_xor(at, off, 64);
                if (ob < best) { best = ob; at = oa; }
            }
            s = s + val[lidx[at]];
            prev = best;
        }
        if (lane == 0) cs[c] = s;
    }
}

// ---- weights and scaling (RegridMatrices_Dynamic.cpp:100-146, 201-233, 290-329) ----------------
__device__ __forceinline__ double ratio_of(const RgView &rg, int keykind, int64_t sparse) {
    long iA = sparse;
    if (keykind == KEY_E) iA = rg.sHC >= rg.sA ? (sparse % rg.sHC) / rg.sA : sparse / rg.sA;
    return rg.ratioA[iA];
}
struct FinalizeArgs {
    int family, scale, correctA, row_key, col_key;
    int nrow, ncol;
    const int64_t *row_s, *col_s;     // dense -> sparse
    const double *rs, *cs;
    double *wM, *Mw, *rowmul, *colmul; // rowmul/colmul: per-row / per-column factors applied to T
};
__global__ void k_weights(RgView rg, FinalizeArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.nrow) {
        const double rs = a.rs[i];
        double wM, mul = 1.0;
        if (a.family == FAM_AEVI) {
            if (a.correctA) {
                const double r = ratio_of(rg, a.row_key, a.row_s[i]);
                wM = r * rs;                                   // :112-115
                if (a.scale) mul = (1.0 / r) * (1.0 / rs);     // :118-122
            } else {
                wM = rs;                                       // :135-136
                if (a.scale) mul = 1.0 / rs;                   // :140
            }
        } else if (a.family == FAM_IVAE) {
            wM = rs;                                           // :201
            if (a.scale) mul = 1.0 / rs;                       // :218,228
        } else {
            if (a.correctA) {
                const double r = ratio_of(rg, a.row_key, a.row_s[i]);
                wM = r * rs;                                   // :303-304
                if (a.scale) mul = 1.0 / wM;                   // :313
            } else {
                wM = rs;                                       // :321
                if (a.scale) mul = 1.0 / rs;                   // :324
            }
        }
        a.wM[i] = wM;
        a.rowmul[i] = mul;
    }
    if (i < a.ncol) {
        const double cs = a.cs[i];
        double Mw = cs, cm = 1.0;                              // :100, :226, :322
        if (a.correctA && a.family != FAM_AEVI) {
            const double r = ratio_of(rg, a.col_key, a.col_s[i]);
            Mw = r * cs;                                       // :214-215, :309-310
            cm = r;                                            // * sApvA, :220,223,315,317
        }
        a.Mw[i] = Mw;
        a.colmul[i] = cm;
    }
}
// M = [diag(rowmul)] * T [* diag(colmul)], each product separately rounded, in the reference's order
__global__ void k_scale(const int32_t *__restrict__ row, const int32_t *__restrict__ col, double *__restrict__ val,
                        long nnz, const double *__restrict__ rowmul, const double *__restrict__ colmul,
                        int apply_row, int apply_col) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nnz) return;
    double v = val[u];
    if (apply_row) v = rowmul[row[u]] * v;
    if (apply_col) v = v * colmul[col[u]];
    val[u] = v;
}

// ---- shared: sorted (row,col,idx) contributions -> Weighted CSR ---------------------------------
struct Triplets { uint64_t *keys, *keys_alt; uint32_t *idx, *idx_alt; double *term; size_t n; };

// ---- contributions -> CSR for matrices with short rows (IvA, IvE, XvA, XvE: 1-8 contributions per row) ----
// No ordering of the whole list: contributions are dropped into per-row slots (integer atomics: the
// slot inside a row is arbitrary, the SET is not) and one thread per row lists its entries in
// ascending (column, emission index) order by repeated selection, summing equal columns in emission
// order with the first term assigned -- exactly what the sort-based path produces.  Rows of more than
// SR_MAX contributions set a flag and the caller falls back to that path.
constexpr int SRB_MAX = 32;
__global__ void k_srb_count(const uint64_t *__restrict__ keys, size_t n, uint32_t *__restrict__ cnt) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) atomicAdd(&cnt[keys[k] >> 32], 1u);
}
__global__ void k_srb_scatter(const uint64_t *__restrict__ keys, size_t n, const uint32_t *__restrict__ slotptr,
                              uint32_t *__restrict__ fillr, uint32_t *__restrict__ scol, uint32_t *__restrict__ sidx) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint64_t key = keys[k];
    const uint32_t r = (uint32_t)(key >> 32);
    const uint32_t p = slotptr[r] + atomicAdd(&fillr[r], 1u);
    scol[p] = (uint32_t)key;
    sidx[p] = (uint32_t)k;                   // emission index (contributions are still in emission order)
}
// next (col, idx) pair after (pc, pi) in the row's slots; returns false when there is none
__device__ __forceinline__ bool srb_next(const uint32_t *__restrict__ scol, const uint32_t *__restrict__ sidx, uint32_t b,
                                         uint32_t e, bool first, uint32_t pc, uint32_t pi, uint32_t &bc, uint32_t &bi) {
    bool any = false;
    for (uint32_t j = b; j < e; ++j) {
        const uint32_t c = scol[j], i = sidx[j];
        const bool after = first || c > pc || (c == pc && i > pi);
        if (after && (!any || c < bc || (c == bc && i < bi))) { bc = c; bi = i; any = true; }
    }
    return any;
}
__global__ void k_srb_unique(const uint32_t *__restrict__ slotptr, int nrow, const uint32_t *__restrict__ scol,
                             uint32_t *__restrict__ ucnt, uint32_t *__restrict__ too_long) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    const uint32_t b = slotptr[r], e = slotptr[r + 1];
    if (e - b > (uint32_t)SRB_MAX) { *too_long = 1u; ucnt[r] = 0; return; }
    uint32_t u = 0;
    for (uint32_t j = b; j < e; ++j) {        // distinct columns: count the first occurrence of each
        bool seen = false;
        for (uint32_t q = b; q < j; ++q) seen = seen || scol[q] == scol[j];
        u += seen ? 0u : 1u;
    }
    ucnt[r] = u;
}
__global__ void k_srb_emit(const uint32_t *__restrict__ slotptr, int nrow, const uint32_t *__restrict__ scol,
                           const uint32_t *__restrict__ sidx, const double *__restrict__ term,
                           const int32_t *__restrict__ rowptr, int32_t *__restrict__ row, int32_t *__restrict__ col,
                           double *__restrict__ val) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    const uint32_t b = slotptr[r], e = slotptr[r + 1];
    int out = rowptr[r] - 1;
    uint32_t pc = 0, pi = 0, bc = 0, bi = 0;
    bool first = true;
    double s = 0.0;
    while (srb_next(scol, sidx, b, e, first, pc, pi, bc, bi)) {
        if (first || bc != pc) {               // a new (row, col): the first term is assigned
            if (!first) val[out] = s;
            ++out;
            row[out] = r; col[out] = (int32_t)bc;
            s = term[bi];
        } else {
            s = s + term[bi];                  // duplicates in emission order
        }
        pc = bc; pi = bi; first = false;
    }
    if (!first) val[out] = s;
}
// returns false (nothing built) when a row is too long for the per-thread selection
static bool build_csr_short_rows(ibh_weighted *w, const Triplets &t, int nrow, int ncol, int32_t **row_out, hipStream_t st) {
    Arena &A = arena();
    const int T = 256;
    uint32_t *cnt = A.get<uint32_t>(2 * (size_t)nrow + 2), *fillr = cnt + nrow, *too_long = fillr + nrow, *d_nnz = too_long + 1;
    uint32_t *slotptr = A.get<uint32_t>((size_t)nrow + 1), *ucnt = A.get<uint32_t>((size_t)nrow);
    uint32_t *scol = A.get<uint32_t>(t.n), *sidx = A.get<uint32_t>(t.n);
    IBH_HIP(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * (2 * (size_t)nrow + 2), st));
    hipLaunchKernelGGL(k_srb_count, dim3(ceil_div(t.n, T)), dim3(T), 0, st, t.keys, t.n, cnt);
    exclusive_scan_u32(cnt, slotptr, (size_t)nrow, slotptr + nrow, st);
    hipLaunchKernelGGL(k_srb_scatter, dim3(ceil_div(t.n, T)), dim3(T), 0, st, t.keys, t.n, slotptr, fillr, scol, sidx);
    hipLaunchKernelGGL(k_srb_unique, dim3(ceil_div(nrow, T)), dim3(T), 0, st, slotptr, nrow, scol, ucnt, too_long);
    w->rowptr.alloc((size_t)nrow + 1);
    exclusive_scan_u32(ucnt, reinterpret_cast<uint32_t *>(w->rowptr.p), (size_t)nrow, d_nnz, st);
    uint32_t h[2];
    readback_sync(h, too_long, sizeof(h), st);
    if (h[0]) return false;
    const uint32_t nnz = h[1];
    IBH_CHECK(nnz < (1u << 31), "nnz overflows int32");
    w->nrow = nrow; w->ncol = ncol; w->nnz = nnz;
    IBH_HIP(hipMemcpyAsync(w->rowptr.p + nrow, d_nnz, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    w->colind.alloc(nnz); w->val.alloc(nnz);
    int32_t *row = A.get<int32_t>(nnz);
    hipLaunchKernelGGL(k_srb_emit, dim3(ceil_div(nrow, T)), dim3(T), 0, st, slotptr, nrow, scol, sidx, t.term, w->rowptr.p, row,
                       w->colind.p, w->val.p);
    IBH_HIP(hipGetLastError());
    *row_out = row;
    return true;
}

static void build_csr_from_contributions(ibh_weighted *w, Triplets t, int nrow, int ncol, int32_t **row_out,
                                         hipStream_t st, bool expect_local = true) {
    Arena &A = arena();
    const int T = 256;
    w->nrow = nrow; w->ncol = ncol;
    w->rowptr.alloc((size_t)nrow + 1);
    if (t.n == 0) {
        w->nnz = 0;
        w->rowptr.zero(st);
        w->colind.alloc(0); w->val.alloc(0);
        *row_out = nullptr;
        return;
    }
    // ice / exchange rows with a handful of contributions each: per-row slots, no ordering of the list
    // (measured: always for ~1 contribution per row; with 2+ per row -- IvE -- only while the build is
    // launch-bound, at 35 M contributions the radix passes are cheaper than the per-row selection)
    if (!expect_local && nrow > 0 && t.n <= 8 * (size_t)nrow && (2 * t.n <= 3 * (size_t)nrow || t.n < (4u << 20)) &&
        get_tuning("assemble_short_rows", 1) && build_csr_short_rows(w, t, nrow, ncol, row_out, st))
        return;
    // Order analysis + piece sort, then (speculatively, on the data as it stands) the duplicate
    // flags and their scan: ONE host synchronisation returns both the analysis and nnz.  Only when a
    // piece was too long for LDS (nothing was touched) does the radix sort run and the tail repeat.
    const int lo_bits = bits_for((uint64_t)ncol), hi_bits = bits_for((uint64_t)nrow);
    struct Readback { OrderInfo info; uint32_t total, unsorted; } rb;
    Readback *d_rb = A.get<Readback>(1);
    uint32_t *head = A.get<uint32_t>(t.n);
    uint32_t *d_total = &d_rb->total;
    auto flags_and_count = [&](bool verify) {
        hipLaunchKernelGGL(k_head_flags, dim3(ceil_div(t.n, T)), dim3(T), 0, st, t.keys, t.n, head, verify ? &d_rb->unsorted : nullptr);
        exclusive_scan_u32(head, head, t.n, d_total, st);
    };
    order_and_chunk_sort(t.keys, t.idx, t.n, &d_rb->info, st, expect_local);
    IBH_HIP(hipMemsetAsync(&d_rb->unsorted, 0, sizeof(uint32_t), st));
    if (expect_local) flags_and_count(false);
    readback_sync(&rb, d_rb, sizeof(rb), st);
    order_debug(rb.info, t.n, lo_bits, hi_bits);
    const bool resort = t.n >= 2 && !order_is_final(rb.info);
    if (resort) {
        // Ice-cell rows: the column field is not monotone along the whole sequence (elevation classes),
        // but inside each ROW it is when the exchange grid is sorted (later cells = later GCM cells =
        // later class ids).  Then a stable sort by the row field alone is the full order: do that
        // first and verify; only a failed check pays for the column passes.
        const bool optimistic = !expect_local && (rb.info.flags & ORD_LO_DEC) && hi_bits > 0;
        OrderInfo plan = rb.info;
        if (optimistic) plan.flags &= ~(uint32_t)ORD_LO_DEC;
        if (radix_after_analysis(plan, t.keys, t.keys_alt, t.idx, t.idx_alt, t.n, lo_bits, hi_bits, st)) {
            std::swap(t.keys, t.keys_alt); std::swap(t.idx, t.idx_alt);
        }
        flags_and_count(optimistic);
        readback_sync(&rb.total, d_total, 2 * sizeof(uint32_t), st);
        if (optimistic && rb.unsorted) {
            if (radix_after_analysis(rb.info, t.keys, t.keys_alt, t.idx, t.idx_alt, t.n, lo_bits, hi_bits, st)) {
                std::swap(t.keys, t.keys_alt); std::swap(t.idx, t.idx_alt);
            }
            flags_and_count(false);
            readback_sync(&rb.total, d_total, sizeof(uint32_t), st);
        }
    } else if (!expect_local) {
        flags_and_count(false);
        readback_sync(&rb.total, d_total, sizeof(uint32_t), st);
    }
    const uint32_t nnz = rb.total;
    IBH_CHECK(nnz < (1u << 31), "nnz overflows int32");
    w->nnz = nnz;
    w->colind.alloc(nnz); w->val.alloc(nnz);
    int32_t *row = A.get<int32_t>(nnz);
    int32_t *segptr = A.get<int32_t>((size_t)nnz + 1);
    hipLaunchKernelGGL(k_segment_heads, dim3(ceil_div(t.n, T)), dim3(T), 0, st, t.keys, t.n, head, nnz, segptr, row, w->colind.p);
    seg_sums<false>(segptr, t.idx, t.term, (int)nnz, (long)t.n, w->val.p, st);
    rowptr_from_rows(row, (long)nnz, nrow, w->rowptr.p, st);
    IBH_HIP(hipGetLastError());
    *row_out = row;
}

// ---- bands for the rowdual apply kernel (spmm.hip) -----------------------------------------------
// An ice cell between two elevation classes is a column of TWO rows of an E-row matrix: (cell a,
// class k) with weight 1-r and (a, k+1) with weight r.  Row by row its field value is fetched twice.
// The band structure lists every (GCM cell, ice cell) pair ONCE, in the lower row, with both weights:
// band r = the entries of row r that are not the upper partner of an entry of the row below, in CSR
// order (columns ascending); an upper partner is attached to its lower entry.  Pairing is decided per
// column from its few entries (per-column slots of the short-column path): within one GCM cell,
// classes pair up from the lowest (k, k+1), so the partner row of a band is always the same row.
// A filtered copy of the CSR: no sorting, exact copies of M's values.
__device__ __forceinline__ void e_decode(const RgView &rg, int64_t e, long &a, long &hc) {
    if (rg.sHC >= rg.sA) { hc = e / rg.sHC; a = (e % rg.sHC) / rg.sA; }
    else { a = e / rg.sA; hc = e % rg.sA; }
}
__global__ void k_band_roles(RgView rg, const int64_t *__restrict__ row_s, const uint32_t *__restrict__ colptr, int ncol,
                             const int32_t *__restrict__ lrow, const uint32_t *__restrict__ lidx,
                             uint32_t *__restrict__ lower, int32_t *__restrict__ partner) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const uint32_t b = colptr[c], e = colptr[c + 1];
    for (uint32_t i = b; i < e; ++i) {
        long ai, hi;
        e_decode(rg, row_s[lrow[i]], ai, hi);
        // number of consecutive classes directly below this entry in the same GCM cell: odd -> upper partner
        int below = 0;
        int32_t up = -1;                       // the entry one class above (its partner if this one is a lower entry)
        for (long t = hi - 1;; --t) {
            bool found = false;
            for (uint32_t j = b; j < e && !found; ++j) {
                long aj, hj;
                e_decode(rg, row_s[lrow[j]], aj, hj);
                found = aj == ai && hj == t;
            }
            if (!found) break;
            ++below;
        }
        for (uint32_t j = b; j < e; ++j) {
            long aj, hj;
            e_decode(rg, row_s[lrow[j]], aj, hj);
            if (aj == ai && hj == hi + 1) up = (int32_t)lidx[j];
        }
        const bool is_lower = (below & 1) == 0;
        lower[lidx[i]] = is_lower ? 1u : 0u;
        partner[lidx[i]] = is_lower ? up : -1;
    }
}
__global__ void k_band_emit(const int32_t *__restrict__ row, const int32_t *__restrict__ col, const double *__restrict__ val,
                            long nnz, const uint32_t *__restrict__ lower, const uint32_t *__restrict__ dpos,
                            const int32_t *__restrict__ partner, int32_t *__restrict__ bcol, double *__restrict__ bv0,
                            double *__restrict__ bv1, int32_t *__restrict__ rb1) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nnz || !lower[u]) return;
    const uint32_t d = dpos[u];
    const int32_t p = partner[u];
    bcol[d] = (int32_t)((uint32_t)col[u] | 0x40000000u | (p >= 0 ? 0x80000000u : 0u));
    bv0[d] = val[u];
    bv1[d] = p >= 0 ? val[p] : 0.0;
    if (p >= 0) rb1[row[p]] = row[u];          // every lower entry of this row names the same partner row
}
__global__ void k_band_ptr(const int32_t *__restrict__ rowptr, int nrow, long nnz, const uint32_t *__restrict__ dpos,
                           const uint32_t *__restrict__ d_total, int32_t *__restrict__ bptr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrow) return;
    const int k = rowptr[r];
    bptr[r] = k < nnz ? (int32_t)dpos[k] : (int32_t)*d_total;
}
// colptr/lrow/lidx: the per-column slots built for the column sums.  *d_total receives the number of band entries.
static void build_bands(ibh_weighted *w, const RgView &rg, const int64_t *row_s, const int32_t *row, const uint32_t *colptr,
                        const int32_t *lrow, const uint32_t *lidx, uint32_t *d_total, hipStream_t st) {
    Arena &A = arena();
    const int T = 256;
    const int nrow = w->nrow, ncol = w->ncol;
    const long nnz = w->nnz;
    uint32_t *lower = A.get<uint32_t>((size_t)nnz), *dpos = A.get<uint32_t>((size_t)nnz);
    int32_t *partner = A.get<int32_t>((size_t)nnz);
    hipLaunchKernelGGL(k_band_roles, dim3(ceil_div(ncol, T)), dim3(T), 0, st, rg, row_s, colptr, ncol, lrow, lidx, lower, partner);
    exclusive_scan_u32(lower, dpos, (size_t)nnz, d_total, st);
    w->band_ptr.alloc((size_t)nrow + 1); w->band_rb1.alloc((size_t)nrow);
    w->band_col.alloc((size_t)nnz); w->band_v0.alloc((size_t)nnz); w->band_v1.alloc((size_t)nnz);       // upper bound; >= nnz/2 are used
    IBH_HIP(hipMemsetAsync(w->band_rb1.p, 0xFF, sizeof(int32_t) * (size_t)nrow, st));
    hipLaunchKernelGGL(k_band_emit, dim3(ceil_div(nnz, T)), dim3(T), 0, st, row, w->colind.p, w->val.p, nnz, lower, dpos, partner,
                       w->band_col.p, w->band_v0.p, w->band_v1.p, w->band_rb1.p);
    hipLaunchKernelGGL(k_band_ptr, dim3(ceil_div(nrow + 1, T)), dim3(T), 0, st, w->rowptr.p, nrow, nnz, dpos, d_total, w->band_ptr.p);
    IBH_HIP(hipGetLastError());
}

// ---- smoothing (sigma != 0): M <- smoothI * M  (smoother.cpp:8-99, RegridMatrices_Dynamic.cpp:237-248) ------
// smoothI[i,j] = exp(-.5 d2(i,j)) * area_j / sum_j(...) over the unmasked ice cells j of dimI with
// d2 = sum_k ((c_j[k]-c_i[k])/sigma[k])^2 < 4, c = (x, y, elevation), area = wM.  The reference finds
// neighbours with an RTree; here: uniform bins of 2*sigma_x x 2*sigma_y, 3x3 bins searched per cell.
// Everything downstream reuses the assembly machinery: neighbour triplets -> sort by (i,j) -> row
// denominators (sequential) -> products with the rows of M as contributions -> sort by (i,a) ->
// sequential sums.  exp() is the device library's: entries agree with the oracle to rounding, not bitwise.
struct SmoothView {
    const int64_t *row_s;       // dense ice id -> sparse
    const double *em, *cen, *area;
    int n;                      // dense ice cells
    double sx, sy, sz, x0, y0, bw, bh;
    int nbx, nby;
};
__device__ __forceinline__ bool smooth_tuple(const SmoothView &v, int d, double c[3]) {
    const int64_t s = v.row_s[d];
    const double e = v.em[s];
    if (e != e) return false;               // masked cells are not tuples (smoother.cpp:83-84)
    c[0] = v.cen[2 * s]; c[1] = v.cen[2 * s + 1]; c[2] = e;
    return true;
}
__device__ __forceinline__ int smooth_bin(const SmoothView &v, const double c[3]) {
    int bx = (int)((c[0] - v.x0) / v.bw), by = (int)((c[1] - v.y0) / v.bh);
    bx = bx < 0 ? 0 : bx >= v.nbx ? v.nbx - 1 : bx;
    by = by < 0 ? 0 : by >= v.nby ? v.nby - 1 : by;
    return by * v.nbx + bx;
}
__global__ void k_smooth_bin_count(SmoothView v, uint32_t *__restrict__ bincnt, int *__restrict__ bad) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= v.n) return;
    double c[3];
    if (!smooth_tuple(v, d, c)) return;
    if (v.area[d] == 0.0) atomicMin(bad, d);          // "Area of cell %ld must be non-zero", smoother.cpp:89-90
    atomicAdd(&bincnt[smooth_bin(v, c)], 1u);
}
__global__ void k_smooth_bin_fill(SmoothView v, const uint32_t *__restrict__ binstart, uint32_t *__restrict__ cursor,
                                  int32_t *__restrict__ members) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= v.n) return;
    double c[3];
    if (!smooth_tuple(v, d, c)) return;
    const int b = smooth_bin(v, c);
    members[binstart[b] + atomicAdd(&cursor[b], 1u)] = d;      // order inside a bin is irrelevant: triplets are sorted later
}
// pass 0: count neighbours of cell d; pass 1: emit (d<<32|j, w) at off[d]
template <int PASS>
__global__ void k_smooth_neighbours(SmoothView v, const uint32_t *__restrict__ binstart, const int32_t *__restrict__ members,
                                    uint32_t *__restrict__ cnt, const uint32_t *__restrict__ off,
                                    uint64_t *__restrict__ keys, uint32_t *__restrict__ idx, double *__restrict__ wraw) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= v.n) return;
    double c[3];
    uint32_t n = 0;
    if (smooth_tuple(v, d, c)) {
        const int b = smooth_bin(v, c), bx = b % v.nbx, by = b / v.nbx;
        const uint32_t o = PASS ? off[d] : 0;
        for (int yy = max(by - 1, 0); yy <= min(by + 1, v.nby - 1); ++yy)
            for (int xx = max(bx - 1, 0); xx <= min(bx + 1, v.nbx - 1); ++xx) {
                const int bb = yy * v.nbx + xx;
                for (uint32_t q = binstart[bb]; q < binstart[bb + 1]; ++q) {
                    const int j = members[q];
                    double cj[3];
                    (void)smooth_tuple(v, j, cj);
                    const double d0 = (cj[0] - c[0]) / v.sx, d1 = (cj[1] - c[1]) / v.sy, d2 = (cj[2] - c[2]) / v.sz;
                    double nds = 0;
                    nds = nds + d0 * d0; nds = nds + d1 * d1; nds = nds + d2 * d2;      // smoother.cpp:29-33
                    if (nds < 4.0) {                                                      // nsigma^2
                        if (PASS) {
                            keys[o + n] = ((uint64_t)(uint32_t)d << 32) | (uint32_t)j;
                            idx[o + n] = o + n;
                            wraw[o + n] = exp(-.5 * nds) * v.area[j];                     // :35-36
                        }
                        ++n;
                    }
                }
            }
    }
    if (!PASS) cnt[d] = n;
}
__global__ void k_smooth_rowptr(const uint32_t *__restrict__ off, int n, uint32_t total, int32_t *__restrict__ rowptr) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n) rowptr[d] = (int32_t)off[d];
    if (d == 0) rowptr[n] = (int32_t)total;
}
// number of M entries each smoothing entry (i,j) multiplies = length of row j of M
__global__ void k_smooth_prod_count(const uint64_t *__restrict__ skeys, size_t ns, const int32_t *__restrict__ rowptr,
                                    uint32_t *__restrict__ cnt) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ns) return;
    const int j = (int)(skeys[q] & 0xffffffffu);
    cnt[q] = (uint32_t)(rowptr[j + 1] - rowptr[j]);
}
__global__ void k_smooth_prod_emit(const uint64_t *__restrict__ skeys, const uint32_t *__restrict__ sidx,
                                   const double *__restrict__ wraw, const double *__restrict__ denom, size_t ns,
                                   const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                   const double *__restrict__ val, const uint32_t *__restrict__ pos,
                                   uint64_t *__restrict__ keys, uint32_t *__restrict__ idx, double *__restrict__ term) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ns) return;
    const uint64_t k = skeys[q];
    const int i = (int)(k >> 32), j = (int)(k & 0xffffffffu);
    const double factor = 1.0 / denom[i];
    const double s = factor * wraw[sidx[q]];                     // factor * ii->second, smoother.cpp:62-64
    uint32_t p = pos[q];
    for (int e = rowptr[j]; e < rowptr[j + 1]; ++e, ++p) {
        keys[p] = ((uint64_t)(uint32_t)i << 32) | (uint32_t)colind[e];
        idx[p] = p;
        term[p] = s * val[e];
    }
}

// ---- smoothI * M without the triplets (round 2) ---------------------------------------------------------------------
// The pipeline below materialises smoothI (~10^8 neighbour triplets at 5 km) and the ~2 x 10^8 products, and sorts both.
// Row i of smoothI * M touches only the few columns its neighbours' rows of M hold (GCM cells / elevation classes within
// two sigmas), so ONE WAVE computes it directly: pass 1 walks the candidates of the 3 x 3 bins (three contiguous member
// ranges), sums the denominator and collects the distinct columns in a small LDS hash set; the set is ranked (ascending
// columns = the CSR order); pass 2 walks the candidates again, parks the hits (j, w_ij / denom) in LDS and lets lane c
// accumulate column c over the hit list, every lane reading the same rows of M (broadcast loads).  Sums run in candidate
// order instead of ascending j: entries agree with the triplet pipeline to rounding (the smoothing test's tolerance: the
// reference's RTree order is not reproducible either); the structure is identical.  Rows with more than SMD_MAXCOL columns
// send the build to the triplet pipeline.
constexpr int SMD_HASH = 256, SMD_MAXCOL = 128, SMD_HITS = 1024, SMD_WAVES = 4;
__device__ __forceinline__ double smd_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <bool EMIT>
__global__ __launch_bounds__(SMD_WAVES * 64) void k_smooth_direct(SmoothView v, const uint32_t *__restrict__ binstart, const int32_t *__restrict__ members,
                                                                  const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                                  const double *__restrict__ val, uint32_t *__restrict__ rowlen,
                                                                  const int32_t *__restrict__ orowptr, int32_t *__restrict__ ocol,
                                                                  double *__restrict__ oval, uint32_t *__restrict__ overflow) {
    __shared__ int s_hash[SMD_WAVES][SMD_HASH];
    __shared__ int s_cols[SMD_WAVES][SMD_MAXCOL];
    __shared__ int s_hj[EMIT ? SMD_WAVES : 1][EMIT ? SMD_HITS : 1];
    __shared__ double s_hw[EMIT ? SMD_WAVES : 1][EMIT ? SMD_HITS : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * SMD_WAVES + wave;
    if (i >= v.n) return;
    double ci[3];
    if (!smooth_tuple(v, i, ci)) {                   // a masked cell is no tuple: an empty row
        if (!EMIT && lane == 0) rowlen[i] = 0;
        return;
    }
    int *hk = s_hash[wave];
    for (int q = lane; q < SMD_HASH; q += 64) hk[q] = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int b = smooth_bin(v, ci), bx = b % v.nbx, by = b / v.nbx;
    const int x0 = max(bx - 1, 0), x1 = min(bx + 1, v.nbx - 1), y0 = max(by - 1, 0), y1 = min(by + 1, v.nby - 1);
    // weight of candidate j, < 0: not a neighbour (smoother.cpp:29-36)
    auto weight = [&](int j) {
        double cj[3];
        (void)smooth_tuple(v, j, cj);
        const double d0 = (cj[0] - ci[0]) / v.sx, d1 = (cj[1] - ci[1]) / v.sy, d2 = (cj[2] - ci[2]) / v.sz;
        double nds = 0;
        nds = nds + d0 * d0; nds = nds + d1 * d1; nds = nds + d2 * d2;
        return nds < 4.0 ? exp(-.5 * nds) * v.area[j] : -1.0;
    };
    // ---- pass 1: denominator and the set of columns
    double dpart = 0.0;
    for (int yy = y0; yy <= y1; ++yy) {
        const uint32_t qb = binstart[yy * v.nbx + x0], qe = binstart[yy * v.nbx + x1 + 1];
        for (uint32_t q = qb + lane; q < qe; q += 64) {
            const int j = members[q];
            const double wgt = weight(j);
            if (wgt >= 0.0) {
                dpart += wgt;
                for (int e = rowptr[j]; e < rowptr[j + 1]; ++e) {
                    const int c = colind[e];
                    unsigned h = ((unsigned)c * 2654435761u) >> 24;
                    for (int probe = 0; probe < SMD_HASH; ++probe) {
                        const int old = atomicCAS(&hk[h], -1, c);
                        if (old == -1 || old == c) break;
                        h = (h + 1) & (SMD_HASH - 1);
                    }
                }
            }
        }
    }
    const double denom = smd_wave_sum(dpart);
    // the set, compacted (slot order) and ranked (ascending)
    int *cols = s_cols[wave];
    int mykeys[SMD_HASH / 64], nk = 0;
    int K = 0;
#pragma unroll
    for (int r = 0; r < SMD_HASH / 64; ++r) {
        const int k = hk[r * 64 + lane];
        const unsigned long long m = __ballot(k >= 0);
        mykeys[r] = k;
        K += __popcll(m);
        (void)nk;
    }
    if (K > SMD_MAXCOL) {                            // (also a full hash table: K == SMD_HASH)
        if (lane == 0) { atomicOr(overflow, 1u); if (!EMIT) rowlen[i] = 0; }
        return;
    }
    if (!EMIT) { if (lane == 0) rowlen[i] = (uint32_t)K; return; }
    // rank of a key = number of smaller keys in the table (keys are distinct)
#pragma unroll
    for (int r = 0; r < SMD_HASH / 64; ++r) {
        const int k = mykeys[r];
        if (k >= 0) {
            int rank = 0;
            for (int q = 0; q < SMD_HASH; ++q) { const int o = hk[q]; rank += (o >= 0 && o < k) ? 1 : 0; }
            cols[rank] = k;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int c0 = lane < K ? cols[lane] : -2, c1 = 64 + lane < K ? cols[64 + lane] : -2;
    double a0 = 0.0, a1 = 0.0;
    const double factor = 1.0 / denom;                // smoother.cpp:62
    int *hj = s_hj[wave];
    double *hw = s_hw[wave];
    int nh = 0;
    auto drain = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int h = 0; h < nh; ++h) {
            const int j = hj[h];
            const double sc = hw[h];
            for (int e = rowptr[j]; e < rowptr[j + 1]; ++e) {
                const int c = colind[e];
                const double t = sc * val[e];
                if (c == c0) a0 = a0 + t;
                else if (c == c1) a1 = a1 + t;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        nh = 0;
    };
    // ---- pass 2: the hits, parked SMD_HITS at a time, accumulated column by lane
    for (int yy = y0; yy <= y1; ++yy) {
        const uint32_t qb = binstart[yy * v.nbx + x0], qe = binstart[yy * v.nbx + x1 + 1];
        for (uint32_t qq = qb; qq < qe; qq += 64) {
            const uint32_t q = qq + lane;
            double wgt = -1.0;
            int j = 0;
            if (q < qe) { j = members[q]; wgt = weight(j); }
            const unsigned long long m = __ballot(wgt >= 0.0);
            if (nh + 64 > SMD_HITS) drain();
            if (wgt >= 0.0) {
                const int pos = nh + __popcll(m & ((1ull << lane) - 1ull));
                hj[pos] = j; hw[pos] = factor * wgt;
            }
            nh += __popcll(m);
        }
    }
    drain();
    const int ob = orowptr[i];
    if (lane < K) { ocol[ob + lane] = c0; oval[ob + lane] = a0; }
    if (64 + lane < K) { ocol[ob + 64 + lane] = c1; oval[ob + 64 + lane] = a1; }
}

// ---- smoothI * M by spatial tiles (round 3): grids too large for one wave per row ---------------------------------------
// The direct kernel above lets every row walk its ~3 600 candidates alone (three passes, five gathers per candidate: 13 ms at
// 5 km); the triplet pipeline materialises ~10^8 neighbour pairs and sorts them twice (12 ms).  But the rows of one spatial bin
// share their candidates -- the members of the 3 x 3 bins around it.  So: ONE WAVE per 64 rows of a bin; the candidates are
// staged 64 at a time in LDS (compact records in bin order: position, elevation, area, and the candidate's row of M already
// translated to slots of the bin's column table) and every lane runs its own row against the staged candidate, which is
// wave-uniform: broadcast LDS reads, no gathers in the inner loop.  One pass sums the denominator and adds w_ij * M[j, c] into an
// LDS table acc[slot][lane] with ds_add_f64 (no return: nothing waits), noting the slot in a 128-bit presence mask per lane --
// row i holds column c iff a neighbour within two sigmas does, exactly the reference's structure; the sums are divided by the
// denominator at the end.  The bin's column table (the distinct columns of all members of its 3 x 3 neighbourhood, ascending) comes from a
// small pre-kernel; a compaction kernel turns (presence, table) into CSR rows.  Sums run in candidate order (bin by bin,
// member by member), fixed: reproducible; entries agree with the oracle's ascending-j order to rounding (the test's 1e-12).
// More than SMT_KMAX columns around a bin, or a member row of M with more than SMT_ROWMAX entries: the triplet pipeline serves
// the build.
constexpr int SMT_KMAX = 128, SMT_ROWMAX = 8, SMT_HASH = 1024;
struct SmtCand { double x, y, z, area; };
// compact records in member (bin) order: position / elevation / area, and the member's row of M entry-major
// (mcol[e * nmem + q], mval[...]): the tile kernel stages them with coalesced, independent loads
__global__ void k_smt_pack(SmoothView v, const int32_t *__restrict__ members, int nmem, const int32_t *__restrict__ rowptr,
                           const int32_t *__restrict__ colind, const double *__restrict__ val, SmtCand *__restrict__ cand,
                           unsigned char *__restrict__ mne, int32_t *__restrict__ mcol, double *__restrict__ mval) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nmem) return;
    const int d = members[q];
    double c[3];
    (void)smooth_tuple(v, d, c);
    cand[q] = SmtCand{c[0], c[1], c[2], v.area[d]};
    const int e0 = rowptr[d], ne = min(rowptr[d + 1] - e0, SMT_ROWMAX);
    mne[q] = (unsigned char)ne;
    for (int e = 0; e < SMT_ROWMAX; ++e) {
        mcol[(size_t)e * nmem + q] = e < ne ? colind[e0 + e] : 0;
        mval[(size_t)e * nmem + q] = e < ne ? val[e0 + e] : 0.0;
    }
}
// per bin: the distinct columns of the members of its 3 x 3 neighbourhood, ascending -> bincols[b * SMT_KMAX ..], binK[b];
// waves the bin needs (64 rows each) -> binwaves[b]
__global__ __launch_bounds__(256) void k_smt_bincols(SmoothView v, const uint32_t *__restrict__ binstart, const int32_t *__restrict__ members,
                                                     const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                                     int32_t *__restrict__ bincols, int32_t *__restrict__ binK, uint32_t *__restrict__ binwaves,
                                                     uint32_t *__restrict__ flags) {
    __shared__ int s_hash[SMT_HASH];
    __shared__ int s_n;
    const int b = blockIdx.x, bx = b % v.nbx, by = b / v.nbx;
    const uint32_t mine = binstart[b + 1] - binstart[b];
    if (threadIdx.x == 0) { binwaves[b] = (mine + 63) / 64; s_n = 0; }
    if (mine == 0) { if (threadIdx.x == 0) binK[b] = 0; return; }
    for (int q = threadIdx.x; q < SMT_HASH; q += blockDim.x) s_hash[q] = -1;
    __syncthreads();
    const int x0 = max(bx - 1, 0), x1 = min(bx + 1, v.nbx - 1), y0 = max(by - 1, 0), y1 = min(by + 1, v.nby - 1);
    for (int yy = y0; yy <= y1; ++yy) {
        const uint32_t qb = binstart[yy * v.nbx + x0], qe = binstart[yy * v.nbx + x1 + 1];
        for (uint32_t q = qb + threadIdx.x; q < qe; q += blockDim.x) {
            const int j = members[q];
            if (rowptr[j + 1] - rowptr[j] > SMT_ROWMAX) atomicOr(flags, 2u);
            for (int e = rowptr[j]; e < rowptr[j + 1]; ++e) {
                const int c = colind[e];
                unsigned h = ((unsigned)c * 2654435761u) >> 22;
                for (int probe = 0; probe < SMT_HASH; ++probe) {
                    const int old = atomicCAS(&s_hash[h], -1, c);
                    if (old == -1) { atomicAdd(&s_n, 1); break; }
                    if (old == c) break;
                    h = (h + 1) & (SMT_HASH - 1);
                }
            }
        }
    }
    __syncthreads();
    const int K = s_n;
    if (threadIdx.x == 0) binK[b] = K;
    if (K > SMT_KMAX) { if (threadIdx.x == 0) atomicOr(flags, 1u); return; }
    for (int q = threadIdx.x; q < SMT_HASH; q += blockDim.x) {
        const int k = s_hash[q];
        if (k < 0) continue;
        int rank = 0;
        for (int t = 0; t < SMT_HASH; ++t) { const int o = s_hash[t]; rank += (o >= 0 && o < k) ? 1 : 0; }
        bincols[(size_t)b * SMT_KMAX + rank] = k;
    }
}
// wavebin[w] = the bin of wave w (wstart: exclusive scan of binwaves)
__global__ void k_smt_wavebin(const uint32_t *__restrict__ wstart, const uint32_t *__restrict__ binwaves, int nbins, int32_t *__restrict__ wavebin) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbins) return;
    for (uint32_t k = 0; k < binwaves[b]; ++k) wavebin[wstart[b] + k] = b;
}
__global__ __launch_bounds__(64) void k_smt_tile(SmoothView v, const uint32_t *__restrict__ binstart, const int32_t *__restrict__ members,
                                                 const SmtCand *__restrict__ cand, const unsigned char *__restrict__ mne,
                                                 const int32_t *__restrict__ mcol, const double *__restrict__ mval, int nmem,
                                                 const int32_t *__restrict__ bincols, const int32_t *__restrict__ binK,
                                                 const uint32_t *__restrict__ wstart, const int32_t *__restrict__ wavebin, int kstride,
                                                 double *__restrict__ scratch, unsigned long long *__restrict__ pres, uint32_t *__restrict__ rowlen) {
    extern __shared__ double smt_lds[];
    double *acc = smt_lds;                                       // [K][64]
    SmtCand *s_c = reinterpret_cast<SmtCand *>(smt_lds + (size_t)kstride * 64);      // [64]
    double *s_val = reinterpret_cast<double *>(s_c + 64);        // [64][SMT_ROWMAX]
    int *s_cols = reinterpret_cast<int *>(s_val + 64 * SMT_ROWMAX);                  // [SMT_KMAX]
    unsigned char *s_slot = reinterpret_cast<unsigned char *>(s_cols + SMT_KMAX);    // [64][SMT_ROWMAX]
    unsigned char *s_ne = s_slot + 64 * SMT_ROWMAX;             // [64]
    const int w = blockIdx.x, lane = threadIdx.x;
    const int b = wavebin[w], bx = b % v.nbx, by = b / v.nbx;
    const int K = binK[b];
    const uint32_t q0 = binstart[b] + (uint32_t)(w - (int)wstart[b]) * 64u, qend = binstart[b + 1];
    const bool live = q0 + lane < qend;
    const SmtCand me = cand[live ? q0 + lane : qend - 1];
    for (int k = lane; k < K; k += 64) s_cols[k] = bincols[(size_t)b * SMT_KMAX + k];
    for (int k = 0; k < K; ++k) acc[k * 64 + lane] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int x0 = max(bx - 1, 0), x1 = min(bx + 1, v.nbx - 1), y0 = max(by - 1, 0), y1 = min(by + 1, v.nby - 1);
    // weight of the staged candidate for this lane's row, < 0: not a neighbour (smoother.cpp:29-36).  The reference divides by
    // the sigmas; three f64 divisions per pair are most of the arithmetic here, so the distance is formed with reciprocals and
    // only a pair within 1e-9 of the cut (nds == 4: cells exactly two sigmas apart on a regular grid) repeats the reference's
    // divisions to decide on which side it falls -- the structure stays exactly the reference's, the weights differ from the
    // divided form in the last bit at most.
    const double rx = 1.0 / v.sx, ry = 1.0 / v.sy, rz = 1.0 / v.sz;
    auto weight = [&](const SmtCand &c) {
        const double d0 = (c.x - me.x) * rx, d1 = (c.y - me.y) * ry, d2 = (c.z - me.z) * rz;
        double nds = 0;
        nds = nds + d0 * d0; nds = nds + d1 * d1; nds = nds + d2 * d2;
        bool in = nds < 4.0;
        if (fabs(nds - 4.0) < 1e-9) {
            const double e0 = (c.x - me.x) / v.sx, e1 = (c.y - me.y) / v.sy, e2 = (c.z - me.z) / v.sz;
            double ex = 0;
            ex = ex + e0 * e0; ex = ex + e1 * e1; ex = ex + e2 * e2;
            in = ex < 4.0;
        }
        return in ? exp(-.5 * nds) * c.area : -1.0;
    };
    // ONE pass: the denominator and the unnormalised sums together (the reference multiplies every term by factor = 1 / denom
    // first, smoother.cpp:62-64; factoring it out of the sum changes the rounding, not the value: the test's 1e-12)
    double denom = 0.0;
    unsigned long long p0 = 0ull, p1 = 0ull;
    for (int yy = y0; yy <= y1; ++yy) {
        const uint32_t qb = binstart[yy * v.nbx + x0], qe = binstart[yy * v.nbx + x1 + 1];
        // the records of chunk c+1 are loaded (coalesced, independent: one round trip) while chunk c is consumed
        auto fetch = [&](uint32_t qq, SmtCand &fc, int &fne, int &c0, int &c1, double &a0, double &a1) {
            const uint32_t q = min(qq + (uint32_t)lane, qe - 1);
            fc = cand[q]; fne = mne[q]; c0 = mcol[q]; c1 = mcol[(size_t)nmem + q]; a0 = mval[q]; a1 = mval[(size_t)nmem + q];
        };
        SmtCand fc; int fne, fc0, fc1; double fa0, fa1;
        if (qb < qe) fetch(qb, fc, fne, fc0, fc1, fa0, fa1);
        for (uint32_t qq = qb; qq < qe; qq += 64) {
            const int nst = (int)min(64u, qe - qq);
            __builtin_amdgcn_wave_barrier();                     // the previous chunk has been consumed
            auto slot_of = [&](int c) {                          // the column is in the bin's table by construction
                int lo = 0, hi = K - 1;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_cols[mid] < c) lo = mid + 1; else hi = mid; }
                return lo;
            };
            if (lane < nst) {
                s_c[lane] = fc;
                s_ne[lane] = (unsigned char)fne;
                if (fne > 0) { s_slot[lane * SMT_ROWMAX] = (unsigned char)slot_of(fc0); s_val[lane * SMT_ROWMAX] = fa0; }
                if (fne > 1) { s_slot[lane * SMT_ROWMAX + 1] = (unsigned char)slot_of(fc1); s_val[lane * SMT_ROWMAX + 1] = fa1; }
                for (int e = 2; e < fne; ++e) {                  // (an ice cell under several GCM cells: rare)
                    s_slot[lane * SMT_ROWMAX + e] = (unsigned char)slot_of(mcol[(size_t)e * nmem + qq + lane]);
                    s_val[lane * SMT_ROWMAX + e] = mval[(size_t)e * nmem + qq + lane];
                }
            }
            if (qq + 64 < qe) fetch(qq + 64, fc, fne, fc0, fc1, fa0, fa1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int t = 0; t < nst; ++t) {
                const double wgt = weight(s_c[t]);
                const bool hit = live && wgt >= 0.0;
                if (hit) denom += wgt;
                const int ne = s_ne[t];
                for (int e = 0; e < ne; ++e) {                   // (wave-uniform: every lane looks at the same candidate)
                    const int k = s_slot[t * SMT_ROWMAX + e];
                    if (hit) {
                        __hip_atomic_fetch_add(acc + k * 64 + lane, wgt * s_val[t * SMT_ROWMAX + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (k < 64) p0 |= 1ull << k; else p1 |= 1ull << (k - 64);
                    }
                }
            }
        }
    }
    const double factor = live ? 1.0 / denom : 0.0;              // smoother.cpp:62
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < K; ++k) scratch[((size_t)w * kstride + k) * 64 + lane] = factor * acc[k * 64 + lane];
    if (live) {
        pres[2 * (size_t)(q0 + lane)] = p0; pres[2 * (size_t)(q0 + lane) + 1] = p1;
        rowlen[members[q0 + lane]] = (uint32_t)(__popcll(p0) + __popcll(p1));
    }
}
__global__ void k_smt_emit(const uint32_t *__restrict__ binstart, const int32_t *__restrict__ members, int nmem, const uint32_t *__restrict__ wstart,
                           const int32_t *__restrict__ memberbin, const int32_t *__restrict__ bincols, int kstride, const double *__restrict__ scratch,
                           const unsigned long long *__restrict__ pres, const int32_t *__restrict__ orowptr, int32_t *__restrict__ ocol,
                           double *__restrict__ oval) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nmem) return;
    const int b = memberbin[q];
    const uint32_t rel = (uint32_t)q - binstart[b];
    const size_t w = wstart[b] + rel / 64;
    const int lane = (int)(rel & 63u);
    int o = orowptr[members[q]];
    for (int half = 0; half < 2; ++half) {
        unsigned long long m = pres[2 * (size_t)q + half];
        while (m) {
            const int k = __builtin_ctzll(m) + 64 * half;
            m &= m - 1;
            ocol[o] = bincols[(size_t)b * SMT_KMAX + k];
            oval[o] = scratch[(w * kstride + k) * 64 + lane];
            ++o;
        }
    }
}
// bin of every member position (for the compaction kernel)
__global__ void k_smt_memberbin(const uint32_t *__restrict__ binstart, int nbins, int32_t *__restrict__ memberbin) {
    const int b = blockIdx.x;
    for (uint32_t q = binstart[b] + threadIdx.x; q < binstart[b + 1]; q += blockDim.x) memberbin[q] = b;
}

static void smooth_matrix(ibh_weighted *w, const ibh_regrid_matrices *rm, const int64_t *row_s, const double sigma[3],
                          hipStream_t st) {
    const ibh_regridder *g = rm->rg;
    IBH_CHECK(g->has_centroid, "smoothing (sigma != 0) needs the ice grid's centroid_xy (ibh_regridder_desc.I_centroid_xy)");
    IBH_CHECK(sigma[0] > 0 && sigma[1] > 0 && sigma[2] > 0, "smoothing needs three positive sigmas, got (%g, %g, %g)", sigma[0], sigma[1], sigma[2]);
    if (w->nrow == 0) return;
    Arena &A = arena();
    const int T = 256, n = w->nrow;
    SmoothView v{row_s, rm->elevmaskI.p, g->I_centroid.p, w->wM.p, n, sigma[0], sigma[1], sigma[2],
                 g->cmin[0], g->cmin[1], 2.0 * sigma[0], 2.0 * sigma[1], 1, 1};
    v.nbx = (int)std::min(4096.0, std::floor((g->cmax[0] - g->cmin[0]) / v.bw) + 1.0);
    v.nby = (int)std::min(4096.0, std::floor((g->cmax[1] - g->cmin[1]) / v.bh) + 1.0);
    if (v.nbx == 4096) v.bw = (g->cmax[0] - g->cmin[0]) / 4096.0 * (1 + 1e-12);     // never finer than 4096 bins: bins only grow
    if (v.nby == 4096) v.bh = (g->cmax[1] - g->cmin[1]) / 4096.0 * (1 + 1e-12);
    const size_t nbins = (size_t)v.nbx * v.nby;
    uint32_t *binstart = A.get<uint32_t>(nbins + 1), *cursor = A.get<uint32_t>(nbins);
    int32_t *members = A.get<int32_t>((size_t)n);
    uint32_t *d_cnt = A.get<uint32_t>(4);
    int *d_bad = reinterpret_cast<int *>(d_cnt + 3);
    const int big = 0x7fffffff;
    IBH_HIP(hipMemsetAsync(binstart, 0, sizeof(uint32_t) * (nbins + 1), st));
    IBH_HIP(hipMemsetAsync(cursor, 0, sizeof(uint32_t) * nbins, st));
    IBH_HIP(hipMemcpyAsync(d_bad, &big, sizeof(int), hipMemcpyHostToDevice, st));
    const dim3 grid(ceil_div(n, T));
    hipLaunchKernelGGL(k_smooth_bin_count, grid, dim3(T), 0, st, v, binstart, d_bad);
    exclusive_scan_u32(binstart, binstart, nbins + 1, nullptr, st);
    hipLaunchKernelGGL(k_smooth_bin_fill, grid, dim3(T), 0, st, v, binstart, cursor, members);
    // (measured: 2 x faster than the triplet pipeline at 20 km -- 0.50 against 1.08 ms, that one is launch-bound there --
    // but slower at 5 km, 13 against 12 ms: three passes of ~3600 candidate evaluations per row, each five gathers; the
    // direct form is therefore taken for small problems only.  `smooth_direct`: 1 always, 0 never, -1 by size.)
    const int direct = get_tuning("smooth_direct", -1);
    const bool take_direct = direct > 0 || (direct < 0 && n <= get_tuning("smooth_direct_max_rows", 16384));
    const int tile = get_tuning("smooth_tile", -1);
    if (tile > 0 || (tile < 0 && !take_direct)) {
        // tiles (k_smt_*): column tables per bin -> one wave per 64 rows of a bin -> compaction; two read-backs
        const int nb = (int)nbins;
        uint32_t h4[4];
        IBH_HIP(hipMemsetAsync(d_cnt, 0, 3 * sizeof(uint32_t), st));
        int32_t *bincols = A.get<int32_t>(nbins * SMT_KMAX), *binK = A.get<int32_t>(nbins);
        uint32_t *binwaves = A.get<uint32_t>(nbins + 1), *wstart = A.get<uint32_t>(nbins + 1);
        // members of all bins = the unmasked rows: binstart[nbins]
        hipLaunchKernelGGL(k_smt_bincols, dim3(nb), dim3(256), 0, st, v, binstart, members, w->rowptr.p, w->colind.p, bincols, binK, binwaves, d_cnt + 1);
        exclusive_scan_u32(binwaves, wstart, nbins, d_cnt, st);
        IBH_HIP(hipMemcpyAsync(d_cnt + 2, binstart + nbins, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        readback_sync(h4, d_cnt, sizeof(h4), st);
        if ((int)h4[3] != big) {
            int64_t sc = 0;
            IBH_HIP(hipMemcpy(&sc, row_s + (int)h4[3], sizeof(int64_t), hipMemcpyDeviceToHost));
            fail(IBH_EINVAL, "Area of cell %ld must be non-zero", (long)sc);
        }
        if (h4[1] == 0) {
            const int nwaves = (int)h4[0], nmem = (int)h4[2];
            std::vector<int32_t> hK(nbins);
            IBH_HIP(hipMemcpy(hK.data(), binK, sizeof(int32_t) * nbins, hipMemcpyDeviceToHost));
            int kmax = 1;
            for (size_t q = 0; q < nbins; ++q) kmax = std::max(kmax, hK[q]);
            uint32_t *rowlen = A.get<uint32_t>((size_t)n + 1);
            IBH_HIP(hipMemsetAsync(rowlen, 0, sizeof(uint32_t) * ((size_t)n + 1), st));
            SmtCand *cand = A.get<SmtCand>((size_t)std::max(nmem, 1));
            int32_t *wavebin = A.get<int32_t>((size_t)std::max(nwaves, 1)), *memberbin = A.get<int32_t>((size_t)std::max(nmem, 1));
            double *scratch = A.get<double>((size_t)std::max(nwaves, 1) * kmax * 64);
            unsigned long long *pres = A.get<unsigned long long>(2 * (size_t)std::max(nmem, 1));
            unsigned char *mne = A.get<unsigned char>((size_t)std::max(nmem, 1));
            int32_t *mcol = A.get<int32_t>((size_t)std::max(nmem, 1) * SMT_ROWMAX);
            double *mval = A.get<double>((size_t)std::max(nmem, 1) * SMT_ROWMAX);
            DevBuf<int32_t> nrowptr((size_t)n + 1);
            if (nmem) {
                hipLaunchKernelGGL(k_smt_pack, dim3(ceil_div(nmem, T)), dim3(T), 0, st, v, members, nmem, w->rowptr.p, w->colind.p, w->val.p, cand, mne, mcol, mval);
                hipLaunchKernelGGL(k_smt_wavebin, dim3(ceil_div(nb, T)), dim3(T), 0, st, wstart, binwaves, nb, wavebin);
                hipLaunchKernelGGL(k_smt_memberbin, dim3(nb), dim3(64), 0, st, binstart, nb, memberbin);
                const size_t lds = (size_t)kmax * 64 * 8 + 64 * sizeof(SmtCand) + 64 * SMT_ROWMAX * 8 + SMT_KMAX * 4 + 64 * SMT_ROWMAX + 64;
                static std::mutex mu;
                static bool raised[64] = {};
                if (lds > 64 * 1024) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!raised[w->device & 63]) {
                        IBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_smt_tile), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        raised[w->device & 63] = true;
                    }
                }
                hipLaunchKernelGGL(k_smt_tile, dim3(nwaves), dim3(64), lds, st, v, binstart, members, cand, mne, mcol, mval, nmem, bincols, binK,
                                   wstart, wavebin, kmax, scratch, pres, rowlen);
            }
            exclusive_scan_u32(rowlen, reinterpret_cast<uint32_t *>(nrowptr.p), (size_t)n, d_cnt, st);
            readback_sync(h4, d_cnt, sizeof(h4), st);
            const uint32_t nnz2 = h4[0];
            IBH_CHECK(nnz2 < (1u << 31), "smoothed matrix too large (%u entries)", nnz2);
            IBH_HIP(hipMemcpyAsync(nrowptr.p + n, d_cnt, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            DevBuf<int32_t> ncol((size_t)nnz2);
            DevBuf<double> nval((size_t)nnz2);
            if (nmem) hipLaunchKernelGGL(k_smt_emit, dim3(ceil_div(nmem, T)), dim3(T), 0, st, binstart, members, nmem, wstart, memberbin, bincols, kmax, scratch,
                                         pres, nrowptr.p, ncol.p, nval.p);
            IBH_HIP(hipGetLastError());
            IBH_HIP(hipStreamSynchronize(st));
            w->rowptr = std::move(nrowptr); w->colind = std::move(ncol); w->val = std::move(nval);
            w->nnz = nnz2;
            w->conservative = 0;            // conservative = !smooth, RegridMatrices_Dynamic.cpp:167
            return;
        }
        // more than SMT_KMAX columns around a bin, or a member row longer than SMT_ROWMAX: the pipelines below serve the build
    }
    if (take_direct) {
        // direct rows (k_smooth_direct): count -> scan -> emit; one read-back (error word, overflow flag, nnz)
        uint32_t *rowlen = A.get<uint32_t>((size_t)n + 1);
        IBH_HIP(hipMemsetAsync(d_cnt, 0, 3 * sizeof(uint32_t), st));
        const dim3 gd(ceil_div(n, SMD_WAVES));
        hipLaunchKernelGGL(k_smooth_direct<false>, gd, dim3(SMD_WAVES * 64), 0, st, v, binstart, members, w->rowptr.p, w->colind.p, w->val.p, rowlen,
                           (const int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr, d_cnt + 1);
        DevBuf<int32_t> nrowptr((size_t)n + 1);
        exclusive_scan_u32(rowlen, reinterpret_cast<uint32_t *>(nrowptr.p), (size_t)n, d_cnt, st);
        uint32_t hd[4];
        readback_sync(hd, d_cnt, sizeof(hd), st);
        if ((int)hd[3] != big) {
            int64_t sc = 0;
            IBH_HIP(hipMemcpy(&sc, row_s + (int)hd[3], sizeof(int64_t), hipMemcpyDeviceToHost));
            fail(IBH_EINVAL, "Area of cell %ld must be non-zero", (long)sc);
        }
        if (hd[1] == 0) {
            const uint32_t nnz2 = hd[0];
            IBH_CHECK(nnz2 < (1u << 31), "smoothed matrix too large (%u entries)", nnz2);
            IBH_HIP(hipMemcpyAsync(nrowptr.p + n, d_cnt, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            DevBuf<int32_t> ncol((size_t)nnz2);
            DevBuf<double> nval((size_t)nnz2);
            hipLaunchKernelGGL(k_smooth_direct<true>, gd, dim3(SMD_WAVES * 64), 0, st, v, binstart, members, w->rowptr.p, w->colind.p, w->val.p,
                               (uint32_t *)nullptr, nrowptr.p, ncol.p, nval.p, d_cnt + 1);
            IBH_HIP(hipGetLastError());
            IBH_HIP(hipStreamSynchronize(st));          // the old CSR was read until here
            w->rowptr = std::move(nrowptr); w->colind = std::move(ncol); w->val = std::move(nval);
            w->nnz = nnz2;
            w->conservative = 0;            // conservative = !smooth, RegridMatrices_Dynamic.cpp:167
            return;
        }
        // a row with more than SMD_MAXCOL columns: the triplet pipeline below serves the build
    }
    uint32_t *ncnt = A.get<uint32_t>((size_t)n);
    hipLaunchKernelGGL(k_smooth_neighbours<0>, grid, dim3(T), 0, st, v, binstart, members, ncnt, (const uint32_t *)nullptr,
                       (uint64_t *)nullptr, (uint32_t *)nullptr, (double *)nullptr);
    exclusive_scan_u32(ncnt, ncnt, (size_t)n, d_cnt, st);
    uint32_t h[4];
    IBH_HIP(hipMemcpyAsync(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    IBH_HIP(hipStreamSynchronize(st));
    if ((int)h[3] != big) {
        int64_t s = 0;
        IBH_HIP(hipMemcpy(&s, row_s + (int)h[3], sizeof(int64_t), hipMemcpyDeviceToHost));
        fail(IBH_EINVAL, "Area of cell %ld must be non-zero", (long)s);
    }
    const size_t ns = h[0];
    IBH_CHECK(ns < (1ul << 31), "smoothing matrix too large (%zu entries)", ns);
    uint64_t *sk = A.get<uint64_t>(ns), *sk2 = A.get<uint64_t>(ns);
    uint32_t *si = A.get<uint32_t>(ns), *si2 = A.get<uint32_t>(ns);
    double *wraw = A.get<double>(ns);
    hipLaunchKernelGGL(k_smooth_neighbours<1>, grid, dim3(T), 0, st, v, binstart, members, (uint32_t *)nullptr, ncnt, sk, si, wraw);
    // order the triplets by (i, j): rows become contiguous with ascending columns
    KeyField sf[2] = {{0, bits_for((uint64_t)n)}, {32, bits_for((uint64_t)n)}};
    if (ns && sf[0].nbits > 0 && radix_sort_pairs(sk, sk2, si, si2, ns, sf, 2, st)) { std::swap(sk, sk2); std::swap(si, si2); }
    int32_t *srowptr = A.get<int32_t>((size_t)n + 1);
    hipLaunchKernelGGL(k_smooth_rowptr, grid, dim3(T), 0, st, ncnt, n, (uint32_t)ns, srowptr);
    double *denom = A.get<double>((size_t)n);
    seg_sums<true>(srowptr, si, wraw, n, (long)ns, denom, st);          // denom_sum, smoother.cpp:37,51
    // smoothI * M as contributions in (i, j ascending) order
    uint32_t *pcnt = A.get<uint32_t>(ns ? ns : 1);
    if (ns) hipLaunchKernelGGL(k_smooth_prod_count, dim3(ceil_div(ns, T)), dim3(T), 0, st, sk, ns, w->rowptr.p, pcnt);
    exclusive_scan_u32(pcnt, pcnt, ns, d_cnt + 1, st);
    IBH_HIP(hipMemcpyAsync(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, st));
    IBH_HIP(hipStreamSynchronize(st));
    Triplets t;
    t.n = h[1];
    IBH_CHECK(t.n < (1ul << 31), "smoothed matrix too large (%zu products)", t.n);
    t.keys = A.get<uint64_t>(t.n); t.keys_alt = A.get<uint64_t>(t.n);
    t.idx = A.get<uint32_t>(t.n); t.idx_alt = A.get<uint32_t>(t.n);
    t.term = A.get<double>(t.n);
    if (ns) hipLaunchKernelGGL(k_smooth_prod_emit, dim3(ceil_div(ns, T)), dim3(T), 0, st, sk, si, wraw, denom, ns, w->rowptr.p,
                               w->colind.p, w->val.p, pcnt, t.keys, t.idx, t.term);
    IBH_HIP(hipGetLastError());
    // the products read w's CSR while the new one is being built: build into fresh buffers, then swap
    ibh_weighted tmp;
    int32_t *row = nullptr;
    build_csr_from_contributions(&tmp, t, w->nrow, w->ncol, &row, st);
    IBH_HIP(hipStreamSynchronize(st));
    w->rowptr = std::move(tmp.rowptr); w->colind = std::move(tmp.colind); w->val = std::move(tmp.val);
    w->nnz = tmp.nnz;
    w->conservative = 0;            // conservative = !smooth, RegridMatrices_Dynamic.cpp:167
}

// The same structure for a matrix that already exists (either assembly path; spmm.hip calls this when an E-row matrix is
// applied again and again): per-column slots from the CSR, then build_bands.  One synchronisation (the band count).
__global__ void k_expand_rows(const int32_t *__restrict__ rowptr, int nrow, int32_t *__restrict__ row) {
    const int r = blockIdx.x;                         // one workgroup per row: rows of E matrices hold 10^1..10^4 entries
    for (int k = rowptr[r] + threadIdx.x; k < rowptr[r + 1]; k += blockDim.x) row[k] = r;
}
void build_bands_from_csr(const ibh_weighted *cw, hipStream_t st) {
    ibh_weighted *w = const_cast<ibh_weighted *>(cw);
    if (!w->band_eligible || w->band_n > 0 || w->nnz == 0 || w->nrow == 0 || w->ncol >= (1 << 28)) return;
    const ibh_sparse_set *rset = w->dims[0];
    if (!rset || rset->identity || rset->dev_n < w->nrow) return;         // the row keys must be on the device
    Arena &A = arena();
    A.reset();
    const int T = 256, nrow = w->nrow, ncol = w->ncol;
    const long nnz = w->nnz;
    RgView rg{};
    rg.sA = w->band_sA; rg.sHC = w->band_sHC;
    int32_t *row = A.get<int32_t>((size_t)nnz);
    hipLaunchKernelGGL(k_expand_rows, dim3(nrow), dim3(T), 0, st, w->rowptr.p, nrow, row);
    uint32_t *colptr = A.get<uint32_t>((size_t)ncol + 1);
    uint32_t *cntc = A.get<uint32_t>(2 * (size_t)ncol + 1), *fillc = cntc + ncol;
    int32_t *lrow = A.get<int32_t>((size_t)nnz);
    uint32_t *lidx = A.get<uint32_t>((size_t)nnz);
    IBH_HIP(hipMemsetAsync(cntc, 0, sizeof(uint32_t) * (2 * (size_t)ncol + 1), st));
    hipLaunchKernelGGL(k_col_count, dim3(ceil_div(nnz, T)), dim3(T), 0, st, w->colind.p, nnz, cntc);
    exclusive_scan_u32(cntc, colptr, (size_t)ncol, colptr + ncol, st);
    hipLaunchKernelGGL(k_col_scatter, dim3(ceil_div(nnz, T)), dim3(T), 0, st, row, w->colind.p, nnz, colptr, fillc, lrow, lidx);
    uint32_t *d_nb = A.get<uint32_t>(1);
    build_bands(w, rg, rset->dev.p, row, colptr, lrow, lidx, d_nb, st);
    uint32_t nb = 0;
    readback_sync(&nb, d_nb, sizeof(uint32_t), st);
    w->band_n = nb;
}

// ---- column-sweep structure (sweep_kernel.inl) of a matrix with short columns, from its CSR --------------------------
// ALL columns are swept once, in ascending order.  The entries are ordered by column (stable radix sort: a column's
// entries stay in row order) and paired up: one ITEM = a column + <= 2 of its entries (an ice cell of an E-row matrix has
// the two classes of its GCM cell: one item; a cell that straddles two GCM cells has four entries: two items).  64 items
// make a block, `tb` blocks a task.  A task keeps partial sums for the rows its columns touch: its local row table =
// the distinct rows of its entries in ascending order ("slots", <= SWEEP_MAX_SLOTS, else tb is halved), found by sorting
// the (task, row) pairs; the rows of the partial-sum array are exactly those distinct pairs.  The combine lists (which
// partial rows make up row r, in task order) come from one more sort by row.  Exact copies of M's values throughout.
// Host synchronisations for sizes; runs once per matrix, lazily (spmm.hip).
constexpr int SWEEP_MAX_SLOTS = 32;
__global__ void k_sw_expand(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind, int nrow, uint64_t *__restrict__ keys,
                            uint32_t *__restrict__ idx, int32_t *__restrict__ erow) {
    const int r = blockIdx.x;                       // one workgroup per row
    for (int k = rowptr[r] + threadIdx.x; k < rowptr[r + 1]; k += blockDim.x) { keys[k] = (uint64_t)(uint32_t)colind[k]; idx[k] = (uint32_t)k; erow[k] = r; }
}
// per sorted entry: is it the first of its column (-> colstart), then: does it open an item (even position in its column)
__global__ void k_sw_colstart(const uint64_t *__restrict__ keys, long n, int32_t *__restrict__ colstart) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || keys[i] != keys[i - 1]) colstart[(uint32_t)keys[i]] = (int32_t)i;
}
__global__ void k_sw_itemheads(const uint64_t *__restrict__ keys, long n, const int32_t *__restrict__ colstart, uint32_t *__restrict__ ih) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ih[i] = ((i - colstart[(uint32_t)keys[i]]) & 1) == 0 ? 1u : 0u;
}
// (task, row) of every sorted entry; payload = the entry's sorted position
__global__ void k_sw_taskrow(const uint32_t *__restrict__ ih, const uint32_t *__restrict__ iscan, const uint32_t *__restrict__ idx, long n,
                             const int32_t *__restrict__ erow, int items_per_task, uint64_t *__restrict__ tk, uint32_t *__restrict__ tv) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t item = iscan[i] + ih[i] - 1;
    tk[i] = ((uint64_t)(item / (uint32_t)items_per_task) << 32) | (uint32_t)erow[idx[i]];
    tv[i] = (uint32_t)i;
}
__global__ void k_sw_uniq(const uint64_t *__restrict__ tk, long n, uint32_t *__restrict__ u) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    u[k] = (k == 0 || tk[k] != tk[k - 1]) ? 1u : 0u;
}
// task_p0[t] = index of the task's first distinct (task, row) pair; uniq_row / uniq list for the combine
__global__ void k_sw_taskfirst(const uint64_t *__restrict__ tk, const uint32_t *__restrict__ u, const uint32_t *__restrict__ uscan, long n,
                               int32_t *__restrict__ task_p0, int32_t *__restrict__ urow) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || !u[k]) return;
    const uint32_t t = (uint32_t)(tk[k] >> 32);
    urow[uscan[k]] = (int32_t)(uint32_t)tk[k];
    if (k == 0 || (uint32_t)(tk[k - 1] >> 32) != t) task_p0[t] = (int32_t)uscan[k];
}
__global__ void k_sw_slots(const uint64_t *__restrict__ tk, const uint32_t *__restrict__ tv, const uint32_t *__restrict__ u,
                           const uint32_t *__restrict__ uscan, long n, const int32_t *__restrict__ task_p0, uint8_t *__restrict__ eslot) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t t = (uint32_t)(tk[k] >> 32);
    eslot[tv[k]] = (uint8_t)(uscan[k] + u[k] - 1 - (uint32_t)task_p0[t]);
}
__global__ void k_sw_task_ns(const int32_t *__restrict__ task_p0, int ntask, int nuniq, int32_t *__restrict__ task_ns,
                             uint32_t *__restrict__ d_maxns) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntask) return;
    const int ns = (t + 1 < ntask ? task_p0[t + 1] : nuniq) - task_p0[t];
    task_ns[t] = ns;
    atomicMax(d_maxns, (uint32_t)ns);
}
__global__ void k_sw_items(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ ih,
                           const uint32_t *__restrict__ iscan, long n, const uint8_t *__restrict__ eslot, const double *__restrict__ val,
                           int32_t *__restrict__ it_col, uint32_t *__restrict__ meta, double *__restrict__ v0, double *__restrict__ v1,
                           uint32_t *__restrict__ d_notident) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !ih[i]) return;
    const long it = (long)iscan[i];
    if ((long)(uint32_t)keys[i] != it) *d_notident = 1u;     // some column has no item, or more than one
    const uint32_t s0 = eslot[i];
    uint32_t m = s0 | (s0 << 8) | SWEEP_HAS0;
    double w1 = 0.0;
    if (i + 1 < n && keys[i + 1] == keys[i]) {      // the odd entry of the pair
        m = s0 | ((uint32_t)eslot[i + 1] << 8) | SWEEP_HAS0 | SWEEP_HAS1;
        w1 = val[idx[i + 1]];
    }
    it_col[it] = (int32_t)(uint32_t)keys[i]; meta[it] = m; v0[it] = val[idx[i]]; v1[it] = w1;
}
__global__ void k_sw_combkeys(const int32_t *__restrict__ urow, int nuniq, uint64_t *__restrict__ ck, uint32_t *__restrict__ cv) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nuniq) { ck[k] = (uint64_t)(uint32_t)urow[k]; cv[k] = (uint32_t)k; }
}
bool build_sweep_from_csr(const ibh_weighted *cw, hipStream_t st) {
    ibh_weighted *w = const_cast<ibh_weighted *>(cw);
    if (w->sweep_ntask > 0 || w->nnz == 0 || w->nrow == 0 || w->nnz >= (1ll << 31) - 64) return false;
    Arena &A = arena();
    A.reset();
    const int T = 256, nrow = w->nrow, ncol = w->ncol;
    const long nnz = w->nnz;
    uint32_t *d_cnt = A.get<uint32_t>(8);
    IBH_HIP(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * 8, st));
    // entries by column
    uint64_t *keys = A.get<uint64_t>((size_t)nnz), *keys2 = A.get<uint64_t>((size_t)nnz);
    uint32_t *idx = A.get<uint32_t>((size_t)nnz), *idx2 = A.get<uint32_t>((size_t)nnz);
    int32_t *erow = A.get<int32_t>((size_t)nnz);
    hipLaunchKernelGGL(k_sw_expand, dim3(nrow), dim3(T), 0, st, w->rowptr.p, w->colind.p, nrow, keys, idx, erow);
    KeyField kf{0, bits_for((uint64_t)ncol)};
    if (kf.nbits > 0 && radix_sort_pairs(keys, keys2, idx, idx2, (size_t)nnz, &kf, 1, st)) { std::swap(keys, keys2); std::swap(idx, idx2); }
    // items: the entries of a column in pairs
    int32_t *colstart = A.get<int32_t>((size_t)ncol);
    uint32_t *ih = A.get<uint32_t>((size_t)nnz), *iscan = A.get<uint32_t>((size_t)nnz);
    hipLaunchKernelGGL(k_sw_colstart, dim3(ceil_div(nnz, T)), dim3(T), 0, st, keys, nnz, colstart);
    hipLaunchKernelGGL(k_sw_itemheads, dim3(ceil_div(nnz, T)), dim3(T), 0, st, keys, nnz, colstart, ih);
    exclusive_scan_u32(ih, iscan, (size_t)nnz, d_cnt + 0, st);
    uint32_t h[8];
    readback_sync(h, d_cnt, sizeof(h), st);
    const long nitems = (long)h[0];
    const int nblk = (int)((nitems + 63) / 64);
    if (nitems <= 0 || (long)nblk * 64 >= (1ll << 31)) return false;
    // tasks and their local row tables; tb halves until every task touches <= SWEEP_MAX_SLOTS rows
    int tb = get_tuning("sweep_tb", 0);
    if (tb <= 0) {                                  // the largest power of two <= 16 that leaves >= 1024 tasks (two per CU and round: measured at
        tb = 16;                                    // 1 km, 64 fields: tb 2 / 4 / 6 / 8 -> 276 / 234 / 230 / 214 us; beyond 8 the row tables outgrow the LDS)
        while (tb > 1 && nblk / tb < 1024) tb /= 2;
    }
    uint64_t *tk = A.get<uint64_t>((size_t)nnz), *tk2 = A.get<uint64_t>((size_t)nnz);
    uint32_t *tv = A.get<uint32_t>((size_t)nnz), *tv2 = A.get<uint32_t>((size_t)nnz);
    uint32_t *u = A.get<uint32_t>((size_t)nnz), *uscan = A.get<uint32_t>((size_t)nnz);
    int32_t *urow = A.get<int32_t>((size_t)nnz);
    uint8_t *eslot = A.get<uint8_t>((size_t)nnz);
    int ntask = 0, nuniq = 0, maxns = 0;
    for (;; tb /= 2) {
        if (tb < 1) return false;                   // 64 items touch more rows than the kernel's table holds
        ntask = ceil_div(nblk, tb);
        uint64_t *a = tk, *b2 = tk2;
        uint32_t *av = tv, *bv = tv2;
        hipLaunchKernelGGL(k_sw_taskrow, dim3(ceil_div(nnz, T)), dim3(T), 0, st, ih, iscan, idx, nnz, erow, 64 * tb, a, av);
        KeyField tf[2] = {{0, bits_for((uint64_t)nrow)}, {32, bits_for((uint64_t)ntask)}};
        if (radix_sort_pairs(a, b2, av, bv, (size_t)nnz, tf, 2, st)) { std::swap(a, b2); std::swap(av, bv); }
        hipLaunchKernelGGL(k_sw_uniq, dim3(ceil_div(nnz, T)), dim3(T), 0, st, a, nnz, u);
        exclusive_scan_u32(u, uscan, (size_t)nnz, d_cnt + 1, st);
        w->sweep_task_p0.alloc((size_t)ntask + 1); w->sweep_task_ns.alloc((size_t)ntask + 1);
        hipLaunchKernelGGL(k_sw_taskfirst, dim3(ceil_div(nnz, T)), dim3(T), 0, st, a, u, uscan, nnz, w->sweep_task_p0.p, urow);
        readback_sync(h, d_cnt, sizeof(h), st);
        nuniq = (int)h[1];
        IBH_HIP(hipMemsetAsync(d_cnt + 2, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_sw_task_ns, dim3(ceil_div(ntask, T)), dim3(T), 0, st, w->sweep_task_p0.p, ntask, nuniq, w->sweep_task_ns.p, d_cnt + 2);
        readback_sync(h, d_cnt, sizeof(h), st);
        maxns = (int)h[2];
        // LDS per workgroup = tile (33 KB) + 4 waves x maxns x 512 B: <= 22 slots keep two workgroups on a CU, <= 9 three
        const int soft = get_tuning("sweep_soft_slots", 22);
        if (getenv("IBH_SWEEP_DEBUG")) fprintf(stderr, "sweep: nitems %ld nblk %d tb %d ntask %d nuniq %d maxns %d\n", nitems, nblk, tb, ntask, nuniq, maxns);
        if (maxns <= SWEEP_MAX_SLOTS && (maxns <= soft || tb == 1)) {
            hipLaunchKernelGGL(k_sw_slots, dim3(ceil_div(nnz, T)), dim3(T), 0, st, a, av, u, uscan, nnz, w->sweep_task_p0.p, eslot);
            break;
        }
    }
    // items
    w->sweep_col.alloc((size_t)nblk * 64);
    w->sweep_meta.alloc((size_t)nblk * 64); w->sweep_v0.alloc((size_t)nblk * 64); w->sweep_v1.alloc((size_t)nblk * 64);
    IBH_HIP(hipMemsetAsync(w->sweep_col.p, 0, sizeof(int32_t) * (size_t)nblk * 64, st));
    IBH_HIP(hipMemsetAsync(w->sweep_meta.p, 0, sizeof(uint32_t) * (size_t)nblk * 64, st));
    IBH_HIP(hipMemsetAsync(w->sweep_v0.p, 0, sizeof(double) * (size_t)nblk * 64, st));
    IBH_HIP(hipMemsetAsync(w->sweep_v1.p, 0, sizeof(double) * (size_t)nblk * 64, st));
    hipLaunchKernelGGL(k_sw_items, dim3(ceil_div(nnz, T)), dim3(T), 0, st, keys, idx, ih, iscan, nnz, eslot, w->val.p, w->sweep_col.p,
                       w->sweep_meta.p, w->sweep_v0.p, w->sweep_v1.p, d_cnt + 5);
    // combine lists: the distinct (task, row) pairs by row, tasks ascending inside a row
    uint64_t *ck = A.get<uint64_t>((size_t)nuniq), *ck2 = A.get<uint64_t>((size_t)nuniq);
    uint32_t *cv = A.get<uint32_t>((size_t)nuniq), *cv2 = A.get<uint32_t>((size_t)nuniq);
    hipLaunchKernelGGL(k_sw_combkeys, dim3(ceil_div(nuniq, T)), dim3(T), 0, st, urow, nuniq, ck, cv);
    KeyField cf{0, bits_for((uint64_t)nrow)};
    if (cf.nbits > 0 && radix_sort_pairs(ck, ck2, cv, cv2, (size_t)nuniq, &cf, 1, st)) { std::swap(ck, ck2); std::swap(cv, cv2); }
    int32_t *srow = A.get<int32_t>((size_t)nuniq);
    hipLaunchKernelGGL(k_keys_to_i32, dim3(ceil_div(nuniq, T)), dim3(T), 0, st, ck, (long)nuniq, srow);
    w->sweep_comb_ptr.alloc((size_t)nrow + 1); w->sweep_comb_p.alloc((size_t)nuniq);
    rowptr_from_rows(srow, (long)nuniq, nrow, w->sweep_comb_ptr.p, st);
    IBH_HIP(hipMemcpyAsync(w->sweep_comb_p.p, cv, sizeof(uint32_t) * (size_t)nuniq, hipMemcpyDeviceToDevice, st));
    IBH_HIP(hipGetLastError());
    readback_sync(h, d_cnt, sizeof(h), st);         // (the arena is reused by the next build)
    w->sweep_ident = (h[5] == 0 && nitems == (long)ncol) ? 1 : 0;
    if (w->sweep_ident) w->sweep_col.release();     // column = item index: the list is never read
    w->sweep_tb = tb; w->sweep_nitems = (int32_t)nitems;
    w->sweep_nprow = nuniq;
    w->sweep_nblk = nblk;
    w->sweep_nslot = maxns < 1 ? 1 : maxns;
    w->sweep_ntask = ntask;
    return true;
}

// ---- row-group structure (spmm.hip rowgroup) of an E-row matrix, from its CSR -------------------------------------------
// A group = the rows whose keys decode to the same GCM cell (its elevation classes), in ascending row order: slot s of
// group g.  The group's entries are ordered by column and paired per column into items (an ice cell lies between two classes
// of a GCM cell: one item with both weights).  Exact copies of M's values.  Declined (false, nothing kept) when a group has
// more than IBH_GSLOTS rows or a column more than two entries in one group.  Host synchronisations for sizes; runs once per
// matrix (ibh_weighted_prepare, or lazily from a later apply).
constexpr uint32_t GRP_HAS0 = 1u << 16, GRP_HAS1 = 1u << 17;
static const int64_t *dims_device_table(ibh_sparse_set *set, int n, hipStream_t st);
__global__ void k_rg_rowkeys(RgView rg, const int64_t *__restrict__ row_s, int nrow, uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    long a, hc;
    e_decode(rg, row_s[r], a, hc);
    keys[r] = (uint64_t)a; idx[r] = (uint32_t)r;
}
__global__ void k_rg_heads(const uint64_t *__restrict__ keys, long n, uint32_t *__restrict__ h) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) h[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// sorted rows -> (group, slot): gstart[g] = position of the group's first row
__global__ void k_rg_gstart(const uint32_t *__restrict__ h, const uint32_t *__restrict__ hscan, int nrow, int32_t *__restrict__ gstart) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrow && h[i]) gstart[hscan[i]] = i;
}
__global__ void k_rg_slots(const uint32_t *__restrict__ h, const uint32_t *__restrict__ hscan, const uint32_t *__restrict__ srow, int nrow,
                           int ngrp, const int32_t *__restrict__ gstart, int32_t *__restrict__ grp_of_row, int32_t *__restrict__ slot_of_row,
                           int32_t *__restrict__ slotrow, int32_t *__restrict__ grp_ns, uint32_t *__restrict__ d_maxns) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrow) return;
    const int g = (int)(hscan[i] + h[i]) - 1, s = i - gstart[g];
    const int r = (int)srow[i];
    grp_of_row[r] = g; slot_of_row[r] = s;
    if (s < IBH_GSLOTS) slotrow[g * IBH_GSLOTS + s] = r;
    if (i + 1 == nrow || h[i + 1]) { grp_ns[g] = s + 1; atomicMax(d_maxns, (uint32_t)(s + 1)); }
}
__global__ void k_rg_entrykeys(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind, int nrow, const int32_t *__restrict__ grp_of_row,
                               uint64_t *__restrict__ keys, uint32_t *__restrict__ idx, int32_t *__restrict__ erow) {
    const int r = blockIdx.x;                       // one workgroup per row
    const uint64_t g = (uint64_t)(uint32_t)grp_of_row[r] << 32;
    for (int k = rowptr[r] + threadIdx.x; k < rowptr[r + 1]; k += blockDim.x) { keys[k] = g | (uint32_t)colind[k]; idx[k] = (uint32_t)k; erow[k] = r; }
}
// item heads of the (group, column)-sorted entries: a run of equal keys is one item (two entries at most: else *d_bad)
__global__ void k_rg_itemheads(const uint64_t *__restrict__ keys, long n, uint32_t *__restrict__ h, uint32_t *__restrict__ d_bad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool head = i == 0 || keys[i] != keys[i - 1];
    h[i] = head ? 1u : 0u;
    if (i >= 2 && keys[i] == keys[i - 1] && keys[i] == keys[i - 2]) *d_bad = 1u;
}
__global__ void k_rg_items(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ h,
                           const uint32_t *__restrict__ hscan, long n, const int32_t *__restrict__ erow, const int32_t *__restrict__ slot_of_row,
                           const double *__restrict__ val, int32_t *__restrict__ it_col, uint32_t *__restrict__ meta, double *__restrict__ v0,
                           double *__restrict__ v1, int32_t *__restrict__ it_grp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !h[i]) return;
    const long it = (long)hscan[i];
    const uint32_t s0 = (uint32_t)slot_of_row[erow[idx[i]]];
    uint32_t m = s0 | GRP_HAS0;
    double w1 = 0.0;
    if (i + 1 < n && keys[i + 1] == keys[i]) {
        m |= ((uint32_t)slot_of_row[erow[idx[i + 1]]] << 8) | GRP_HAS1;
        w1 = val[idx[i + 1]];
    }
    it_col[it] = (int32_t)(uint32_t)keys[i]; meta[it] = m; v0[it] = val[idx[i]]; v1[it] = w1;
    it_grp[it] = (int32_t)(keys[i] >> 32);
}
// ---- tiles of the row groups (spmm.hip grouptile) ------------------------------------------------------------
// One workgroup per tile of SEG items: the columns (padded by repeating the last), and the entries of the tile sorted by
// (slot, item) -- ascending item = ascending column = the order of the CSR row -- with their offsets per slot.  Every slot's
// list is padded to a multiple of four entries with {item SEG, weight 0}: the kernel keeps a zero in that place of its
// X tile, so a padded entry adds 0 * 0 and the inner loop needs no predicate (a padded weight never meets a field value: 0 * NaN).
template <int SEG>
__global__ __launch_bounds__(SEG) void k_gt_build(const int32_t *__restrict__ tile_grp, const int32_t *__restrict__ tile_i0,
                                                          const int32_t *__restrict__ grp_ptr, const int32_t *__restrict__ it_col,
                                                          const uint32_t *__restrict__ it_meta, const double *__restrict__ it_v0,
                                                          const double *__restrict__ it_v1, int32_t *__restrict__ gt_col,
                                                          uint16_t *__restrict__ gt_ek, double *__restrict__ gt_ev, uint16_t *__restrict__ gt_eptr) {
    constexpr int NWV = SEG / 64;
    __shared__ uint16_t wc[NWV][IBH_GSLOTS];
    __shared__ uint16_t base[NWV][IBH_GSLOTS];
    __shared__ uint16_t s_end[IBH_GSLOTS], s_pend[IBH_GSLOTS];
    const long t = blockIdx.x;
    const int g = tile_grp[t], i0 = tile_i0[t];
    const int n = min(SEG, grp_ptr[g + 1] - i0);
    const int k = threadIdx.x, lane = k & 63, wv = k >> 6;
    const bool has = k < n;
    const long it = (long)i0 + (has ? k : n - 1);
    gt_col[t * SEG + k] = it_col[it];
    const uint32_t m = has ? it_meta[it] : 0u;
    const bool h0 = (m & GRP_HAS0) != 0, h1 = (m & GRP_HAS1) != 0;
    const int s0 = (int)(m & 0xffu), s1 = (int)((m >> 8) & 0xffu);
    int r0 = 0, r1 = 0;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int s = 0; s < IBH_GSLOTS; ++s) {
        const bool a = h0 && s0 == s, b = h1 && s1 == s;
        const unsigned long long mk = __ballot(a || b);       // (the two entries of an item sit in different rows)
        if (lane == 0) wc[wv][s] = (uint16_t)__popcll(mk);
        const int r = __popcll(mk & lt);
        if (a) r0 = r;
        if (b) r1 = r;
    }
    __syncthreads();
    const long e0 = t * ibh_gt_ecap(SEG);
    if (k == 0) {
        int off = 0;
        for (int s = 0; s < IBH_GSLOTS; ++s) {
            gt_eptr[t * IBH_GT_EP + s] = (uint16_t)off;
            for (int q = 0; q < NWV; ++q) { base[q][s] = (uint16_t)off; off += wc[q][s]; }
            s_end[s] = (uint16_t)off;
            off = (off + 3) & ~3;
            s_pend[s] = (uint16_t)off;
        }
        gt_eptr[t * IBH_GT_EP + IBH_GSLOTS] = (uint16_t)off;
        gt_eptr[t * IBH_GT_EP + IBH_GSLOTS + 1] = 0;
    }
    __syncthreads();
    if (h0) { const long e = e0 + base[wv][s0] + r0; gt_ek[e] = (uint16_t)(k * 8); gt_ev[e] = it_v0[it]; }
    if (h1) { const long e = e0 + base[wv][s1] + r1; gt_ek[e] = (uint16_t)(k * 8); gt_ev[e] = it_v1[it]; }
    if (k < IBH_GSLOTS)
        for (int e = s_end[k]; e < s_pend[k]; ++e) { gt_ek[e0 + e] = (uint16_t)(SEG * 8); gt_ev[e0 + e] = 0.0; }
}
static void build_group_tiles(ibh_weighted *w, const int32_t *d_grp_ptr, int ngrp, const int32_t *it_col, const uint32_t *it_meta,
                              const double *it_v0, const double *it_v1, long nitems, hipStream_t st) {
    // Tiles of 256 items walked by eight waves, or of 128 by four: many small groups (the Antarctic sheet under a half-degree grid:
    // ~600 items per GCM cell, tens of thousands of cells) leave less of a small tile empty and keep more workgroups on a CU --
    // measured, 128 fields: 3.62 against 3.76 ms; few or large groups (1 km Greenland: 3 400 items per cell) take the large one
    // (218 against 235 us at 64 fields).
    int seg = get_tuning("grouptile_seg", (ngrp >= 2048 && nitems < 1024l * ngrp) ? 128 : 256);
    if (seg != 128) seg = 256;
    const int ecap = ibh_gt_ecap(seg);
    std::vector<int32_t> gp((size_t)ngrp + 1), tp((size_t)ngrp + 1);
    IBH_HIP(hipMemcpyAsync(gp.data(), d_grp_ptr, sizeof(int32_t) * gp.size(), hipMemcpyDeviceToHost, st));
    IBH_HIP(hipStreamSynchronize(st));
    long nt = 0;
    for (int g = 0; g < ngrp; ++g) { tp[(size_t)g] = (int32_t)nt; nt += ((long)gp[(size_t)g + 1] - gp[(size_t)g] + seg - 1) / seg; }
    tp[(size_t)ngrp] = (int32_t)nt;
    if (nt <= 0 || nt * ecap >= (1l << 31)) return;
    std::vector<int32_t> tg((size_t)nt), ti((size_t)nt);
    for (int g = 0; g < ngrp; ++g)
        for (int t = tp[(size_t)g], i = gp[(size_t)g]; t < tp[(size_t)g + 1]; ++t, i += seg) { tg[(size_t)t] = g; ti[(size_t)t] = i; }
    DevBuf<int32_t> d_tg((size_t)nt), d_ti((size_t)nt), gt_ptr((size_t)ngrp + 1), gt_col((size_t)nt * seg);
    DevBuf<uint16_t> gt_ek((size_t)nt * ecap), gt_eptr((size_t)nt * IBH_GT_EP);
    DevBuf<double> gt_ev((size_t)nt * ecap);
    d_tg.upload(tg.data(), tg.size(), st); d_ti.upload(ti.data(), ti.size(), st); gt_ptr.upload(tp.data(), tp.size(), st);
    gt_ek.zero(st); gt_ev.zero(st);
    if (seg == 128)
        hipLaunchKernelGGL(k_gt_build<128>, dim3((unsigned)nt), dim3(128), 0, st, d_tg.p, d_ti.p, d_grp_ptr, it_col, it_meta, it_v0, it_v1, gt_col.p,
                           gt_ek.p, gt_ev.p, gt_eptr.p);
    else
        hipLaunchKernelGGL(k_gt_build<256>, dim3((unsigned)nt), dim3(256), 0, st, d_tg.p, d_ti.p, d_grp_ptr, it_col, it_meta, it_v0, it_v1, gt_col.p,
                           gt_ek.p, gt_ev.p, gt_eptr.p);
    IBH_HIP(hipGetLastError());
    IBH_HIP(hipStreamSynchronize(st));              // (the host vectors and the tile maps die here)
    w->gt_ptr = std::move(gt_ptr); w->gt_col = std::move(gt_col); w->gt_ek = std::move(gt_ek); w->gt_eptr = std::move(gt_eptr);
    w->gt_ev = std::move(gt_ev);
    w->gt_seg = seg;
    w->gt_ntile = (int32_t)nt;
}
bool build_groups_from_csr(const ibh_weighted *cw, hipStream_t st) {
    ibh_weighted *w = const_cast<ibh_weighted *>(cw);
    if (w->grp_n > 0 || !w->band_eligible || w->nnz == 0 || w->nrow == 0 || w->nnz >= (1ll << 31) - 64) return false;
    if (!w->dims[0]) return false;
    Arena &A = arena();
    A.reset();
    const int T = 256, nrow = w->nrow;
    const long nnz = w->nnz;
    const int64_t *row_s = dims_device_table(w->dims[0], nrow, st);       // the row keys (a caller-supplied set may live on the host only)
    RgView rg{};
    rg.sA = w->band_sA; rg.sHC = w->band_sHC;
    uint32_t *d_cnt = A.get<uint32_t>(8);
    IBH_HIP(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * 8, st));
    // rows by GCM cell (stable: ascending row inside a group)
    uint64_t *rk = A.get<uint64_t>((size_t)nrow), *rk2 = A.get<uint64_t>((size_t)nrow);
    uint32_t *ri = A.get<uint32_t>((size_t)nrow), *ri2 = A.get<uint32_t>((size_t)nrow);
    hipLaunchKernelGGL(k_rg_rowkeys, dim3(ceil_div(nrow, T)), dim3(T), 0, st, rg, row_s, nrow, rk, ri);
    KeyField rf{0, 32};
    if (radix_sort_pairs(rk, rk2, ri, ri2, (size_t)nrow, &rf, 1, st)) { std::swap(rk, rk2); std::swap(ri, ri2); }
    uint32_t *rh = A.get<uint32_t>((size_t)nrow), *rhs = A.get<uint32_t>((size_t)nrow);
    hipLaunchKernelGGL(k_rg_heads, dim3(ceil_div(nrow, T)), dim3(T), 0, st, rk, (long)nrow, rh);
    exclusive_scan_u32(rh, rhs, (size_t)nrow, d_cnt + 0, st);
    uint32_t h[8];
    readback_sync(h, d_cnt, sizeof(h), st);
    const int ngrp = (int)h[0];
    if (ngrp <= 0) return false;
    int32_t *gstart = A.get<int32_t>((size_t)ngrp);
    int32_t *grp_of_row = A.get<int32_t>((size_t)nrow), *slot_of_row = A.get<int32_t>((size_t)nrow);
    DevBuf<int32_t> slotrow((size_t)ngrp * IBH_GSLOTS), grp_ns((size_t)ngrp), grp_ptr((size_t)ngrp + 1);
    IBH_HIP(hipMemsetAsync(slotrow.p, 0, sizeof(int32_t) * (size_t)ngrp * IBH_GSLOTS, st));
    hipLaunchKernelGGL(k_rg_gstart, dim3(ceil_div(nrow, T)), dim3(T), 0, st, rh, rhs, nrow, gstart);
    hipLaunchKernelGGL(k_rg_slots, dim3(ceil_div(nrow, T)), dim3(T), 0, st, rh, rhs, ri, nrow, ngrp, gstart, grp_of_row, slot_of_row, slotrow.p,
                       grp_ns.p, d_cnt + 1);
    // entries by (group, column): stable, so the two entries of a column keep their row (= slot) order
    uint64_t *keys = A.get<uint64_t>((size_t)nnz), *keys2 = A.get<uint64_t>((size_t)nnz);
    uint32_t *idx = A.get<uint32_t>((size_t)nnz), *idx2 = A.get<uint32_t>((size_t)nnz);
    int32_t *erow = A.get<int32_t>((size_t)nnz);
    hipLaunchKernelGGL(k_rg_entrykeys, dim3(nrow), dim3(T), 0, st, w->rowptr.p, w->colind.p, nrow, grp_of_row, keys, idx, erow);
    KeyField ef[2] = {{0, bits_for((uint64_t)w->ncol)}, {32, bits_for((uint64_t)ngrp)}};
    if (radix_sort_pairs(keys, keys2, idx, idx2, (size_t)nnz, ef, 2, st)) { std::swap(keys, keys2); std::swap(idx, idx2); }
    uint32_t *ih = A.get<uint32_t>((size_t)nnz), *ihs = A.get<uint32_t>((size_t)nnz);
    hipLaunchKernelGGL(k_rg_itemheads, dim3(ceil_div(nnz, T)), dim3(T), 0, st, keys, nnz, ih, d_cnt + 3);
    exclusive_scan_u32(ih, ihs, (size_t)nnz, d_cnt + 2, st);
    readback_sync(h, d_cnt, sizeof(h), st);
    const int maxns = (int)h[1];
    const long nitems = (long)h[2];
    if (getenv("IBH_SWEEP_DEBUG")) fprintf(stderr, "groups: nrow %d ngrp %d maxns %d nitems %ld bad %u\n", nrow, ngrp, maxns, nitems, h[3]);
    if (maxns > IBH_GSLOTS || h[3] != 0 || nitems <= 0) return false;
    DevBuf<int32_t> it_col((size_t)nitems);
    DevBuf<uint32_t> it_meta((size_t)nitems);
    DevBuf<double> it_v0((size_t)nitems), it_v1((size_t)nitems);
    int32_t *it_grp = A.get<int32_t>((size_t)nitems);
    hipLaunchKernelGGL(k_rg_items, dim3(ceil_div(nnz, T)), dim3(T), 0, st, keys, idx, ih, ihs, nnz, erow, slot_of_row, w->val.p, it_col.p, it_meta.p,
                       it_v0.p, it_v1.p, it_grp);
    rowptr_from_rows(it_grp, nitems, ngrp, grp_ptr.p, st);
    IBH_HIP(hipGetLastError());
    IBH_HIP(hipStreamSynchronize(st));              // (the arena is reused by the next build)
    w->grp_ptr = std::move(grp_ptr); w->grp_ns = std::move(grp_ns); w->grp_slotrow = std::move(slotrow);
    w->grp_col = std::move(it_col); w->grp_meta = std::move(it_meta); w->grp_v0 = std::move(it_v0); w->grp_v1 = std::move(it_v1);
    w->grp_nslot = maxns; w->grp_nitems = (int32_t)nitems;
    w->grp_n = ngrp;
    if (get_tuning("grouptile_build", 1)) build_group_tiles(w, w->grp_ptr.p, ngrp, w->grp_col.p, w->grp_meta.p, w->grp_v0.p, w->grp_v1.p, nitems, st);
    return true;
}

#include "fastasm.inl"
#include "streamasm.inl"

bool elevmask_classes(const ibh_regrid_matrices *rm, hipStream_t stream, const double *d_src) {
    const ibh_regridder *gr = rm->rg;
    if (gr->nhc > 64 || !get_tuning("assemble_stream", gr->nX >= (1l << 20) ? 1 : 0)) return false;      // (not a grid of the streamed build: the first build that wants the bytes makes them)
    elevmask_classes_impl(rm, stream, d_src);
    return true;
}

// ---- RegridMatrices_Dynamic::matrix_d ----------------------------------------------------------
static const MatSpec *find_spec(const char *spec_name) {
    const MatSpec *sp = nullptr;
    for (const auto &s : SPECS) if (!strcmp(s.name, spec_name)) sp = &s;
    if (!sp) fail(IBH_ENOKEY, "unknown regrid matrix '%s' (expected one of AvI IvA AvX XvA EvI IvE EvX XvE EvA AvE)", spec_name);
    return sp;
}
// fast_only: build through the plan-based fast path or not at all (returns false, nothing touched) -- the mode of the
// concurrent builds of a batch, which must not append to a set another build of the same wave is reading
bool assemble_matrix(const ibh_regrid_matrices *rm, const char *spec_name, ibh_sparse_set *dim0, ibh_sparse_set *dim1,
                     int scale, int correctA, const double sigma[3], ibh_weighted **out, bool fast_only, ibh_comm *comm) {
    IBH_CHECK(rm && spec_name && out, "null argument");
    const MatSpec *sp = find_spec(spec_name);
    // RegridParams::smooth() (RegridMatrices.hpp:31): only compute_IvAE smooths (RegridMatrices_Dynamic.cpp:237-248)
    const bool smooth = sigma && sigma[0] != 0 && sp->family == FAM_IVAE;
    if (smooth && sp->row_key != KEY_I)
        fail(IBH_ENOTIMPL, "smoothing of '%s' (rows on the exchange grid) is not supported: the smoother needs ice-grid centroids", spec_name);
    const ibh_regridder *g = rm->rg;
    const bool uses_ep = sp->row_list == LIST_EP || sp->col_list == LIST_EP;
    if (uses_ep && g->nhc == 0) fail(IBH_EINVAL, "IceRegridder_L0::GvEp(): hcdefs is zero-length!");   // IceRegridder_L0.cpp:108-109
    // the calling thread's own stream: builds issued by different host threads (assemble_batch) overlap on the GPU
    hipStream_t st = hipStreamPerThread;
    Arena &A = arena();
    A.reset();

    RgView rg{g->ex_indices.p, g->ex_area.p, rm->elevmaskI.p, g->hcdefs.p, g->A_ratio_s.p, g->nX, g->nA,
              g->nhc, g->interp_style, g->hc_stride_A, g->hc_stride_HC};

    std::unique_ptr<ibh_weighted> w(new ibh_weighted);
    IBH_HIP(hipGetDevice(&w->device));
    ibh_sparse_set *dims[2] = {dim0, dim1};
    for (int k = 0; k < 2; ++k) {
        if (!dims[k]) { dims[k] = new ibh_sparse_set; w->owns[k] = true; }
        w->dims[k] = dims[k];
    }
    w->conservative = 1;              // :63, :167 (no smoothing), :258
    w->scaled = scale;                // :421
    if (sp->row_key == KEY_E && (sp->col_key == KEY_I || sp->col_key == KEY_X)) {       // EvI, EvX: bands may be built later (ensure_bands)
        w->band_eligible = 1; w->band_sA = g->hc_stride_A; w->band_sHC = g->hc_stride_HC;
    }

    auto extent_of = [&](int key) -> int64_t {
        return key == KEY_A ? g->nA : key == KEY_E ? g->nA * (int64_t)g->nhc : key == KEY_I ? g->nI : g->nX;
    };
    // sorted exchange grid + dims shapes it covers: the plan-based fast path (fastasm.inl); smoothing and the
    // band structure work on intermediates of the general pipeline
    const bool bands_wanted = sp->row_key == KEY_E && (sp->col_key == KEY_I || sp->col_key == KEY_X) && get_tuning("assemble_bands", 0);
    if (!smooth && !bands_wanted && fast_build(rm, sp, dims, scale, correctA, rg, w.get(), st, comm)) {
        if (!w->built_fast) w->built_fast = 1;          // (2: the streamed build, set by stream_build)
        *out = w.release();
        return true;
    }
    if (fast_only) {
        for (int k = 0; k < 2; ++k) if (w->owns[k]) { delete dims[k]; w->owns[k] = false; w->dims[k] = nullptr; }
        return false;
    }
    A.reset();
    // counters read back with ONE sync: [0] first out-of-range exchange cell, [1] new row keys,
    // [2] new column keys, [3] number of contributions
    // [first(rows) | first(cols) | counters] are one allocation, preset to 0xFF.. with one fill
    const size_t er = (size_t)extent_of(sp->row_key), ec = (size_t)extent_of(sp->col_key);
    uint32_t *first_r = A.get<uint32_t>(er + ec + 4), *first_c = first_r + er;
    uint32_t *d_cnt = first_c + ec;
    uint32_t *d_err = d_cnt;
    const uint32_t big = 0xffffffffu;
    IBH_HIP(hipMemsetAsync(first_r, 0xFF, sizeof(uint32_t) * (er + ec + 1), st));

    // dense numbering in emission order; each user-visible set is numbered by exactly one Ur matrix
    // (RegridMatrices_Dynamic.cpp:75-81, 86-90, 178-183, 187-190, 270-277), so the two are independent.
    Numbering rnum = number_set_prepare(rg, dims[0], extent_of(sp->row_key), sp->row_list, sp->row_key,
                                        (sp->row_list == LIST_EP ? 2 : 1) * g->nX, 0, first_r, st);
    Numbering cnum = number_set_prepare(rg, dims[1], extent_of(sp->col_key), sp->col_list, sp->col_key,
                                        (sp->col_list == LIST_EP ? 2 : 1) * g->nX, 2, first_c, st);
    const int T = 256;
    const dim3 grid(g->nX ? ceil_div(g->nX, T) : 1);
    uint32_t *pk = A.get<uint32_t>((size_t)g->nX);
    uint32_t *roff = A.get<uint32_t>((size_t)g->nX), *coff = A.get<uint32_t>((size_t)g->nX), *poff = A.get<uint32_t>((size_t)g->nX);
    rnum.args.off = roff; cnum.args.off = coff;
    if (uses_ep) {
        hipLaunchKernelGGL(k_first2<true>, grid, dim3(T), 0, st, rg, rnum.args, cnum.args, d_err);
        hipLaunchKernelGGL(k_flag2<true>, grid, dim3(T), 0, st, rg, rnum.args, cnum.args, *sp, pk);
    } else {
        hipLaunchKernelGGL(k_first2<false>, grid, dim3(T), 0, st, rg, rnum.args, cnum.args, d_err);
        hipLaunchKernelGGL(k_flag2<false>, grid, dim3(T), 0, st, rg, rnum.args, cnum.args, *sp, pk);
    }
    exclusive_scan3(pk, (size_t)g->nX, roff, coff, poff, d_cnt + 1, st);
    uint32_t h_cnt[4];
    readback_sync(h_cnt, d_cnt, sizeof(h_cnt), st);
    const uint32_t err_x = h_cnt[0];
    if (err_x != big) {
        // message of linterp_1d_b, IceRegridder_L0.cpp:84-85
        std::vector<int32_t> ij(2);
        IBH_HIP(hipMemcpy(ij.data(), g->ex_indices.p + 2 * (size_t)err_x, 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
        double e = 0;
        IBH_HIP(hipMemcpy(&e, rm->elevmaskI.p + ij[1], sizeof(double), hipMemcpyDeviceToHost));
        fail(IBH_ERANGE, "Elevation %g out of bounds (%g, %g)", e < 0 ? 0.0 : e, g->hcdefs_h.front(), g->hcdefs_h.back());
    }
    for (Numbering *nb : {&rnum, &cnum}) {
        const uint32_t n_new = nb == &rnum ? h_cnt[1] : h_cnt[2];
        IBH_CHECK((int64_t)n_new <= nb->max_new, "internal: more new keys (%u) than reserved (%ld)", n_new, (long)nb->max_new);
        IBH_CHECK((int64_t)nb->ds.n_old + n_new < (1ll << 31), "dense extent overflows int32");
    }
    const uint32_t ncontrib = h_cnt[3];
    Triplets t;
    t.n = ncontrib;
    t.keys = A.get<uint64_t>(t.n); t.keys_alt = A.get<uint64_t>(t.n);
    t.idx = A.get<uint32_t>(t.n); t.idx_alt = A.get<uint32_t>(t.n);
    t.term = A.get<double>(t.n);
    // emit also records the new keys' dense -> sparse entries, so it runs even with no contributions
    if (uses_ep) hipLaunchKernelGGL(k_contrib_emit<true>, grid, dim3(T), 0, st, rg, *sp, rnum.args, cnum.args, pk, poff, t.keys, t.idx, t.term);
    else hipLaunchKernelGGL(k_contrib_emit<false>, grid, dim3(T), 0, st, rg, *sp, rnum.args, cnum.args, pk, poff, t.keys, t.idx, t.term);
    IBH_HIP(hipGetLastError());
    number_set_finish(rnum, h_cnt[1], st);
    number_set_finish(cnum, h_cnt[2], st);
    const DeviceSet &rset = rnum.ds, &cset = cnum.ds;
    const int nrow = rset.n, ncol = cset.n;
    int32_t *row = nullptr;
    // ice-cell rows over shared columns (IvA, IvE): the re-visits of straddling ice cells are far apart
    // in the emission order, the pieces would span the whole sequence -> straight to the radix sort
    build_csr_from_contributions(w.get(), t, nrow, ncol, &row, st, sp->row_key != KEY_I);
    const long nnz = w->nnz;

    // weights
    double *rs = A.get<double>((size_t)nrow), *cs = A.get<double>((size_t)ncol);
    double *rowmul = A.get<double>((size_t)nrow), *colmul = A.get<double>((size_t)ncol);
    w->wM.alloc((size_t)nrow); w->Mw.alloc((size_t)ncol);
    seg_sums<true>(w->rowptr.p, nullptr, w->val.p, nrow, nnz, rs, st);
    uint32_t *colptr = nullptr, *lidx = nullptr;       // per-column slots (short-column path), reused by build_bands
    int32_t *lrow = nullptr;
    if (nnz && nnz <= 4l * ncol) {
        // short columns: per-column slots + one thread per column (see k_col_sums)
        colptr = A.get<uint32_t>((size_t)ncol + 1);
        uint32_t *cntc = A.get<uint32_t>(2 * (size_t)ncol + 1), *fillc = cntc + ncol, *nlong = fillc + ncol;    // zeroed together
        lrow = A.get<int32_t>((size_t)nnz);
        int32_t *longcols = A.get<int32_t>((size_t)ncol);
        lidx = A.get<uint32_t>((size_t)nnz);
        IBH_HIP(hipMemsetAsync(cntc, 0, sizeof(uint32_t) * (2 * (size_t)ncol + 1), st));
        hipLaunchKernelGGL(k_col_count, dim3(ceil_div(nnz, T)), dim3(T), 0, st, w->colind.p, nnz, cntc);
        exclusive_scan_u32(cntc, colptr, (size_t)ncol, colptr + ncol, st);
        hipLaunchKernelGGL(k_col_scatter, dim3(ceil_div(nnz, T)), dim3(T), 0, st, row, w->colind.p, nnz, colptr, fillc, lrow, lidx);
        hipLaunchKernelGGL(k_col_sums, dim3(ceil_div(ncol, T)), dim3(T), 0, st, colptr, ncol, lrow, lidx, w->val.p, cs, nlong, longcols);
        hipLaunchKernelGGL(k_col_sums_long, dim3(256), dim3(T), 0, st, colptr, lrow, lidx, w->val.p, cs, nlong, longcols);
    } else if (nnz) {
        uint64_t *ck = A.get<uint64_t>((size_t)nnz), *ck2 = A.get<uint64_t>((size_t)nnz);
        uint32_t *ci = A.get<uint32_t>((size_t)nnz), *ci2 = A.get<uint32_t>((size_t)nnz);
        hipLaunchKernelGGL(k_col_keys, dim3(ceil_div(nnz, T)), dim3(T), 0, st, w->colind.p, nnz, ck, ci);
        KeyField f{0, bits_for((uint64_t)ncol)};
        if (f.nbits > 0 && radix_sort_pairs(ck, ck2, ci, ci2, (size_t)nnz, &f, 1, st)) { std::swap(ck, ck2); std::swap(ci, ci2); }
        int32_t *scol = A.get<int32_t>((size_t)nnz), *colptr = A.get<int32_t>((size_t)ncol + 1);
        hipLaunchKernelGGL(k_keys_to_i32, dim3(ceil_div(nnz, T)), dim3(T), 0, st, ck, nnz, scol);
        rowptr_from_rows(scol, nnz, ncol, colptr, st);
        seg_sums<true>(colptr, ci, w->val.p, ncol, nnz, cs, st);
    } else if (ncol) {
        IBH_HIP(hipMemsetAsync(cs, 0, sizeof(double) * (size_t)ncol, st));
    }
    FinalizeArgs fa{sp->family, scale, correctA, sp->row_key, sp->col_key, nrow, ncol, rset.to_sparse, cset.to_sparse,
                    rs, cs, w->wM.p, w->Mw.p, rowmul, colmul};
    const int nmax = nrow > ncol ? nrow : ncol;
    if (nmax) hipLaunchKernelGGL(k_weights, dim3(ceil_div(nmax, T)), dim3(T), 0, st, rg, fa);
    const int apply_row = scale ? 1 : 0;
    const int apply_col = (correctA && sp->family != FAM_AEVI) ? 1 : 0;
    if (nnz && (apply_row || apply_col))
        hipLaunchKernelGGL(k_scale, dim3(ceil_div(nnz, T)), dim3(T), 0, st, row, w->colind.p, w->val.p, nnz, rowmul, colmul,
                           apply_row, apply_col);
    IBH_HIP(hipGetLastError());
    if (smooth) smooth_matrix(w.get(), rm, rset.to_sparse, sigma, st);
    // E-row matrices over ice / exchange columns: band structure for the apply (after the scaling: exact
    // copies of M).  Opt-in (ibh_set_tuning("assemble_bands", 1)): it costs +25-35 % of the build and
    // buys -16 % on a 1 km apply, nothing at 5 km -- right for a matrix that is applied to many field
    // batches, wrong for the coupler's one build : one apply per step.
    const bool want_bands = sp->row_key == KEY_E && (sp->col_key == KEY_I || sp->col_key == KEY_X) && colptr != nullptr &&
                            ncol < (1 << 28) && get_tuning("assemble_bands", 0);
    uint32_t nband_entries = 0;
    if (want_bands) {
        uint32_t *d_nb = A.get<uint32_t>(1);
        build_bands(w.get(), rg, rset.to_sparse, row, colptr, lrow, lidx, d_nb, st);
        readback_sync(&nband_entries, d_nb, sizeof(uint32_t), st);
        w->band_n = nband_entries;
    } else {
        IBH_HIP(hipStreamSynchronize(st));
    }
    *out = w.release();
    return true;
}

// ---- a batch of builds (the coupler's per-timestep set, IceCoupler.cpp:361-468) ----------------------------------
// Same results as calling matrix_d for the jobs one after the other, in order.  Jobs that share no set they could
// modify run CONCURRENTLY, each on a host thread of a small persistent pool and that thread's own stream: a 5 km build
// is ~250 workgroups, an eighth of the chip, so four of them overlap almost perfectly.  A job waits for every earlier
// job that may still append to one of its sets (EvI numbers dimE before IvE / XvE read it); jobs of one wave that share a
// pre-populated set run in fast-only mode (the fast path only reads such a set) and anything that needs the general
// pipeline -- which may append -- is redone sequentially afterwards, together with the later jobs that share its sets.
namespace {
class WorkerPool {
    // A coupling step at 5 km is ~0.2 ms in two waves of two builds: a worker that went to sleep on the condition variable between
    // the waves costs a futex wake-up (tens of us) each time.  Workers therefore poll for the next job for a short while after
    // finishing one, and wait() polls before it blocks; idle pools sleep as before.
    static constexpr long SPIN_NS = 400 * 1000;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::deque<std::function<void()>> queue;
    std::atomic<int> queued{0}, pending{0};
    bool stop = false;
    static long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
public:
    explicit WorkerPool(int n) {
        for (int i = 0; i < n; ++i)
            threads.emplace_back([this] {
                long spin_until = 0;
                for (;;) {
                    std::function<void()> job;
                    while (queued.load(std::memory_order_acquire) == 0 && now_ns() < spin_until) __builtin_ia32_pause();
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this] { return stop || !queue.empty(); });
                        if (stop && queue.empty()) return;
                        job = std::move(queue.front());
                        queue.pop_front();
                        queued.fetch_sub(1, std::memory_order_relaxed);
                    }
                    job();
                    if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                        std::lock_guard<std::mutex> lk(mu);      // (pairs with wait(): no wake-up is lost)
                        done_cv.notify_all();
                    }
                    spin_until = now_ns() + SPIN_NS;
                }
            });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : threads) t.detach();      // process exit: the HIP runtime may already be gone, do not join into it
    }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(mu); queue.push_back(std::move(f)); pending.fetch_add(1, std::memory_order_relaxed); queued.fetch_add(1, std::memory_order_release); }
        cv.notify_one();
    }
    void wait() {
        const long until = now_ns() + 4 * SPIN_NS;
        while (pending.load(std::memory_order_acquire) != 0 && now_ns() < until) __builtin_ia32_pause();
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [this] { return pending.load(std::memory_order_acquire) == 0; });
    }
};
WorkerPool &pool() { static WorkerPool *p = new WorkerPool(3); return *p; }
}  // namespace

void assemble_batch(const ibh_regrid_matrices *rm, int n, const char *const *specs, ibh_sparse_set *const *dim0,
                    ibh_sparse_set *const *dim1, const int32_t *scale, const int32_t *correctA, const double sigma[3],
                    ibh_weighted **out) {
    IBH_CHECK(rm && n >= 0 && (n == 0 || (specs && scale && correctA && out)), "bad arguments");
    struct Job { const MatSpec *sp; ibh_sparse_set *d[2]; int wave; bool done; int code; std::string err; };
    std::vector<Job> jobs((size_t)n);
    for (int j = 0; j < n; ++j) {
        IBH_CHECK(specs[j] != nullptr, "null matrix name in batch");
        jobs[(size_t)j] = Job{find_spec(specs[j]), {dim0 ? dim0[j] : nullptr, dim1 ? dim1[j] : nullptr}, 0, false, IBH_OK, std::string()};
        IBH_CHECK(jobs[(size_t)j].d[0] == nullptr || jobs[(size_t)j].d[0] != jobs[(size_t)j].d[1], "dims[0] and dims[1] must be distinct sets");
        out[j] = nullptr;
    }
    const ibh_regridder *g = rm->rg;
    auto extent_of = [&](int key) -> int64_t {
        return key == KEY_A ? g->nA : key == KEY_E ? g->nA * (int64_t)g->nhc : key == KEY_I ? g->nI : g->nX;
    };
    auto readonly = [&](const Job &jb, int k) {          // a set no build ever changes: the identity over its whole extent
        const ibh_sparse_set *s = jb.d[k];
        return s && s->identity && s->n == extent_of(k == 0 ? jb.sp->row_key : jb.sp->col_key);
    };
    // wave of a job: one after the PRODUCER of each of its sets -- the first job of the batch that uses a set which is
    // still empty numbers it; the later users of that set only read it (fast path) and do not wait for each other
    for (int j = 0; j < n; ++j)
        for (int a = 0; a < 2; ++a) {
            const ibh_sparse_set *s = jobs[(size_t)j].d[a];
            if (!s || s->n != 0 || readonly(jobs[(size_t)j], a)) continue;
            for (int i = 0; i < j; ++i)
                if (jobs[(size_t)i].d[0] == s || jobs[(size_t)i].d[1] == s) {      // i is the producer (the first user)
                    jobs[(size_t)j].wave = std::max(jobs[(size_t)j].wave, jobs[(size_t)i].wave + 1);
                    break;
                }
        }
    int dev = 0;
    IBH_HIP(hipGetDevice(&dev));
    int nwaves = 0;
    for (auto &jb : jobs) nwaves = std::max(nwaves, jb.wave + 1);
    auto run = [&](int j, bool fast_only) {
        Job &jb = jobs[(size_t)j];
        try {
            jb.done = assemble_matrix(rm, specs[j], jb.d[0], jb.d[1], scale[j], correctA[j], sigma, &out[j], fast_only);
        } catch (const Error &e) { jb.code = e.code; jb.err = e.msg; }
        catch (const std::exception &e) { jb.code = IBH_EINVAL; jb.err = e.what(); }
    };
    auto shares = [&](int i, int j) {
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                if (jobs[(size_t)i].d[a] && jobs[(size_t)i].d[a] == jobs[(size_t)j].d[b] && !readonly(jobs[(size_t)i], a)) return true;
        return false;
    };
    for (int wv = 0; wv < nwaves; ++wv) {
        std::vector<int> members;
        for (int j = 0; j < n; ++j) if (jobs[(size_t)j].wave == wv) members.push_back(j);
        // members that share a (by now pre-populated) set with another member: fast-only; the others are free to use either path
        std::vector<char> fo(members.size(), 0);
        for (size_t a = 0; a < members.size(); ++a)
            for (size_t b = 0; b < members.size(); ++b)
                if (a != b && shares(members[a], members[b])) fo[a] = 1;
        for (int j : members) { ibh_sparse_set *d[2] = {jobs[(size_t)j].d[0], jobs[(size_t)j].d[1]}; fast_prewarm(rm, jobs[(size_t)j].sp, d, hipStreamPerThread); }
        for (size_t a = 1; a < members.size(); ++a) {
            const int j = members[a];
            const bool f = fo[a] != 0;
            pool().submit([&, j, f, dev] { (void)hipSetDevice(dev); run(j, f); });
        }
        if (!members.empty()) run(members[0], fo[0] != 0);
        pool().wait();
        // anything the fast path declined is redone by the general pipeline, in order; a later member that shares a set
        // with it saw the set before that build could append to it and is redone too
        std::vector<int> redone;
        for (size_t a = 0; a < members.size(); ++a) {
            const int j = members[a];
            Job &jb = jobs[(size_t)j];
            if (jb.code != IBH_OK) continue;
            bool redo = !jb.done;
            for (int b : redone) redo = redo || shares(b, j);
            if (!redo) continue;
            if (out[j]) { delete out[j]; out[j] = nullptr; }
            redone.push_back(j);
            run(j, false);
        }
    }
    for (int j = 0; j < n; ++j)
        if (jobs[(size_t)j].code != IBH_OK) {
            const Job jb = jobs[(size_t)j];
            for (int q = 0; q < n; ++q) { delete out[q]; out[q] = nullptr; }
            throw Error(jb.code, jb.err);
        }
}

// ---- ibh_weighted_from_coo: Eigen setFromTriplets on device -----------------------------------
__global__ void k_coo_keys(const int32_t *__restrict__ row, const int32_t *__restrict__ col, size_t n,
                           uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { keys[k] = ((uint64_t)(uint32_t)row[k] << 32) | (uint32_t)col[k]; idx[k] = (uint32_t)k; }
}
void weighted_from_coo_device(ibh_weighted *w, int nrow, int ncol, int64_t n, const int32_t *row, const int32_t *col,
                              const double *val) {
    hipStream_t st = nullptr;
    Arena &A = arena();
    A.reset();
    Triplets t;
    t.n = (size_t)n;
    t.keys = A.get<uint64_t>(t.n); t.keys_alt = A.get<uint64_t>(t.n);
    t.idx = A.get<uint32_t>(t.n); t.idx_alt = A.get<uint32_t>(t.n);
    t.term = A.get<double>(t.n);
    int32_t *drow = A.get<int32_t>(t.n), *dcol = A.get<int32_t>(t.n);
    if (n) {
        IBH_HIP(hipMemcpyAsync(drow, row, sizeof(int32_t) * t.n, hipMemcpyHostToDevice, st));
        IBH_HIP(hipMemcpyAsync(dcol, col, sizeof(int32_t) * t.n, hipMemcpyHostToDevice, st));
        IBH_HIP(hipMemcpyAsync(t.term, val, sizeof(double) * t.n, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_coo_keys, dim3(ceil_div(n, 256)), dim3(256), 0, st, drow, dcol, t.n, t.keys, t.idx);
    }
    int32_t *r = nullptr;
    build_csr_from_contributions(w, t, nrow, ncol, &r, st);
    IBH_HIP(hipStreamSynchronize(st));
}

// ---- E1vE0 (slib/icebin/e1ve0.cpp:55-106 compute_E1vE0c) ------------------------------------------------
//   E1vE0c = diag(1 / sum_sheets Mw(XuE1)) * sum_sheets[ E1uX * (XvE0 - XvE1) ], consolidated,
// XvE = diag(1/wM) * XuE.  Every exchange cell x has <= a handful of entries in XuE1 and XuE0, so the
// product through X is again a keyed sum: per x, every (e1 of XuE1 row x) x (e of the merged row of
// XvE0 - XvE1) contributes u1 * d to key (e1, e), summed over x ascending with the first term assigned
// (Eigen's conservative sparse product) -- the assembly machinery (contributions -> order -> sequential
// sums) does the rest.  Keys are SPARSE E indices (the reference works on sparsified matrices).
struct XuEView {
    const int32_t *rowptr, *colind;
    const double *val, *wM;
    const int64_t *col_s;        // dense E -> sparse E
    const int32_t *rowmap;       // sparse x -> dense row, nullptr: identity
    int nrow;
};
__device__ __forceinline__ int xue_row(const XuEView &m, long x) {
    if (!m.rowmap) return x < m.nrow ? (int)x : -1;
    return m.rowmap[x];
}
constexpr int E1_MAXROW = 16;    // entries of one XuE row handled in registers (Z_INTERP gives <= 2)
template <int PASS>
__global__ void k_e1ve0(XuEView m1, XuEView m0, long nX, uint32_t *__restrict__ cnt, const uint32_t *__restrict__ pos,
                        uint64_t *__restrict__ keys, uint32_t *__restrict__ idx, double *__restrict__ term,
                        uint32_t *__restrict__ too_long)
{
    const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nX) return;
    const int r1 = xue_row(m1, x), r0 = xue_row(m0, x);
    const int b1 = r1 >= 0 ? m1.rowptr[r1] : 0, n1 = r1 >= 0 ? m1.rowptr[r1 + 1] - b1 : 0;
    const int b0 = r0 >= 0 ? m0.rowptr[r0] : 0, n0 = r0 >= 0 ? m0.rowptr[r0 + 1] - b0 : 0;
    if (n1 > E1_MAXROW || n0 > E1_MAXROW) { *too_long = 1u; if (!PASS) cnt[x] = 0; return; }
    // merged row of D = XvE0 - XvE1 by sparse column: entries of E0 first, then the E1 entries E0 lacks
    int nd = n0;
    for (int j = 0; j < n1; ++j) {
        const int64_t e = m1.col_s[m1.colind[b1 + j]];
        bool in0 = false;
        for (int i = 0; i < n0; ++i) in0 = in0 || m0.col_s[m0.colind[b0 + i]] == e;
        nd += in0 ? 0 : 1;
    }
    if (!PASS) { cnt[x] = (uint32_t)(n1 * nd); return; }
    if (n1 == 0 || nd == 0) return;
    const double s0 = r0 >= 0 ? 1. / m0.wM[r0] : 0.0, s1 = 1. / m1.wM[r1];
    uint32_t p = pos[x];
    for (int a = 0; a < n1; ++a) {
        const int64_t e1 = m1.col_s[m1.colind[b1 + a]];
        const double u1 = m1.val[b1 + a];
        for (int i = 0; i < n0; ++i) {                        // columns stored in E0 (a - b, or a)
            const int64_t e = m0.col_s[m0.colind[b0 + i]];
            double d = s0 * m0.val[b0 + i];
            for (int j = 0; j < n1; ++j)
                if (m1.col_s[m1.colind[b1 + j]] == e) d = d - s1 * m1.val[b1 + j];
            keys[p] = ((uint64_t)e1 << 32) | (uint64_t)e; idx[p] = p; term[p] = u1 * d; ++p;
        }
        for (int j = 0; j < n1; ++j) {                        // columns only E1 stores (0 - b)
            const int64_t e = m1.col_s[m1.colind[b1 + j]];
            bool in0 = false;
            for (int i = 0; i < n0; ++i) in0 = in0 || m0.col_s[m0.colind[b0 + i]] == e;
            if (in0) continue;
            keys[p] = ((uint64_t)e1 << 32) | (uint64_t)e; idx[p] = p; term[p] = u1 * (0.0 - s1 * m1.val[b1 + j]); ++p;
        }
    }
}
__global__ void k_scatter_rowmap(const int64_t *__restrict__ to_sparse, int n, int32_t *__restrict__ rowmap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rowmap[to_sparse[i]] = i;
}
__global__ void k_add_by_sparse(const double *__restrict__ v, const int64_t *__restrict__ to_sparse, int n, double *__restrict__ acc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int64_t s = to_sparse[i]; acc[s] = acc[s] + v[i]; }       // sparse ids of one matrix are distinct
}
__global__ void k_scale_rows_inv(const int32_t *__restrict__ rowptr, int nrow, double *__restrict__ val, const double *__restrict__ sum) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    const int b = rowptr[r], e = rowptr[r + 1];
    if (b == e) return;
    const double s = 1. / sum[r];
    for (int k = b; k < e; ++k) val[k] = val[k] * s;          // ii->value() *= sE1(iE1)
}
__global__ void k_csr_to_contrib(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind, const double *__restrict__ val,
                                 int nrow, uint32_t base, uint64_t *__restrict__ keys, uint32_t *__restrict__ idx, double *__restrict__ term) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const uint32_t p = base + (uint32_t)k;
        keys[p] = ((uint64_t)(uint32_t)r << 32) | (uint32_t)colind[k]; idx[p] = p; term[p] = val[k];
    }
}
__global__ void k_fill_f64(double *p, size_t n, double v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

static const int64_t *dims_device_table(ibh_sparse_set *set, int n, hipStream_t st) {
    // dense -> sparse table of a matrix's dims on the device (an identity set is materialised in the arena)
    Arena &A = arena();
    int64_t *t = A.get<int64_t>((size_t)n);
    if (set->identity) {
        hipLaunchKernelGGL(k_iota_i64, dim3(ceil_div(n, 256)), dim3(256), 0, st, t, (size_t)n);
    } else if (set->dev_n >= n) {
        IBH_HIP(hipMemcpyAsync(t, set->dev.p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    } else {
        set->ensure_host();
        IBH_HIP(hipMemcpyAsync(t, set->host.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st));
        IBH_HIP(hipStreamSynchronize(st));
    }
    return t;
}

void e1ve0_compute(int nsheets, const ibh_weighted *const *XuE1s, const ibh_weighted *const *XuE0s, int64_t nE, ibh_weighted *out) {
    IBH_CHECK(nsheets >= 1 && XuE1s && XuE0s && out, "bad arguments");
    IBH_CHECK(nE > 0 && nE < (1ll << 31), "nE=%ld out of range", (long)nE);
    hipStream_t st = nullptr;
    Arena &A = arena();
    A.reset();
    const int T = 256;
    double *sE1 = A.get<double>((size_t)nE);
    IBH_HIP(hipMemsetAsync(sE1, 0, sizeof(double) * (size_t)nE, st));
    std::vector<std::unique_ptr<ibh_weighted>> locals;
    for (int s = 0; s < nsheets; ++s) {
        const ibh_weighted *w1 = XuE1s[s], *w0 = XuE0s[s];
        IBH_CHECK(w1 && w0, "null matrix for sheet %d", s);
        IBH_CHECK(w1->dims[1]->sparse_extent <= nE && w0->dims[1]->sparse_extent <= nE, "sheet %d: E extent exceeds nE", s);
        const int64_t nX = w1->dims[0]->sparse_extent;
        IBH_CHECK(nX == w0->dims[0]->sparse_extent && nX >= 0 && nX < (1ll << 31), "sheet %d: XuE1 and XuE0 disagree on nX", s);
        XuEView v[2];
        const ibh_weighted *ws[2] = {w1, w0};
        for (int k = 0; k < 2; ++k) {
            const ibh_weighted *w = ws[k];
            v[k] = XuEView{w->rowptr.p, w->colind.p, w->val.p, w->wM.p, dims_device_table(w->dims[1], w->ncol, st), nullptr, w->nrow};
            const bool ident = w->dims[0]->identity && w->dims[0]->n >= w->nrow;
            if (!ident) {
                int32_t *rm = A.get<int32_t>((size_t)nX);
                IBH_HIP(hipMemsetAsync(rm, 0xFF, sizeof(int32_t) * (size_t)nX, st));
                const int64_t *rs = dims_device_table(w->dims[0], w->nrow, st);
                if (w->nrow) hipLaunchKernelGGL(k_scatter_rowmap, dim3(ceil_div(w->nrow, T)), dim3(T), 0, st, rs, w->nrow, rm);
                v[k].rowmap = rm;
            }
        }
        if (w1->ncol) hipLaunchKernelGGL(k_add_by_sparse, dim3(ceil_div(w1->ncol, T)), dim3(T), 0, st, w1->Mw.p, v[0].col_s, w1->ncol, sE1);
        uint32_t *cnt = A.get<uint32_t>((size_t)nX + 2), *pos = A.get<uint32_t>((size_t)nX + 1);
        uint32_t *d_total = cnt + nX, *too_long = cnt + nX + 1;
        IBH_HIP(hipMemsetAsync(cnt + nX, 0, 2 * sizeof(uint32_t), st));
        const dim3 grid(nX ? ceil_div(nX, T) : 1);
        hipLaunchKernelGGL(k_e1ve0<0>, grid, dim3(T), 0, st, v[0], v[1], (long)nX, cnt, (const uint32_t *)nullptr, (uint64_t *)nullptr,
                           (uint32_t *)nullptr, (double *)nullptr, too_long);
        exclusive_scan_u32(cnt, pos, (size_t)nX, d_total, st);
        uint32_t h[2];
        readback_sync(h, d_total, sizeof(h), st);
        IBH_CHECK(!h[1], "E1vE0: an XuE row has more than %d entries", E1_MAXROW);
        Triplets t;
        t.n = h[0];
        t.keys = A.get<uint64_t>(t.n); t.keys_alt = A.get<uint64_t>(t.n);
        t.idx = A.get<uint32_t>(t.n); t.idx_alt = A.get<uint32_t>(t.n);
        t.term = A.get<double>(t.n);
        if (t.n) hipLaunchKernelGGL(k_e1ve0<1>, grid, dim3(T), 0, st, v[0], v[1], (long)nX, cnt, pos, t.keys, t.idx, t.term, too_long);
        IBH_HIP(hipGetLastError());
        std::unique_ptr<ibh_weighted> loc(new ibh_weighted);
        int32_t *row = nullptr;
        build_csr_from_contributions(loc.get(), t, (int)nE, (int)nE, &row, st);
        locals.push_back(std::move(loc));
    }
    // scale by 1 / (sum over sheets of Mw(XuE1)), then consolidate across sheets
    for (auto &loc : locals)
        hipLaunchKernelGGL(k_scale_rows_inv, dim3(ceil_div(nE, T)), dim3(T), 0, st, loc->rowptr.p, (int)nE, loc->val.p, sE1);
    IBH_HIP(hipGetLastError());
    if (locals.size() == 1) {
        out->rowptr = std::move(locals[0]->rowptr); out->colind = std::move(locals[0]->colind); out->val = std::move(locals[0]->val);
        out->nnz = locals[0]->nnz;
    } else {
        Triplets t;
        t.n = 0;
        for (auto &loc : locals) t.n += (size_t)loc->nnz;
        IBH_CHECK(t.n < (1ul << 31), "E1vE0 too large");
        t.keys = A.get<uint64_t>(t.n); t.keys_alt = A.get<uint64_t>(t.n);
        t.idx = A.get<uint32_t>(t.n); t.idx_alt = A.get<uint32_t>(t.n);
        t.term = A.get<double>(t.n);
        uint32_t base = 0;
        for (auto &loc : locals) {          // sheet-major emission order: equal (iE1, iE0) are summed in sheet order
            hipLaunchKernelGGL(k_csr_to_contrib, dim3(ceil_div(nE, T)), dim3(T), 0, st, loc->rowptr.p, loc->colind.p, loc->val.p, (int)nE,
                               base, t.keys, t.idx, t.term);
            base += (uint32_t)loc->nnz;
        }
        int32_t *row = nullptr;
        build_csr_from_contributions(out, t, (int)nE, (int)nE, &row, st);
    }
    out->nrow = out->ncol = (int)nE;
    out->wM.alloc((size_t)nE); out->Mw.alloc((size_t)nE);
    hipLaunchKernelGGL(k_fill_f64, dim3(ceil_div(nE, T)), dim3(T), 0, st, out->wM.p, (size_t)nE, 1.0);
    hipLaunchKernelGGL(k_fill_f64, dim3(ceil_div(nE, T)), dim3(T), 0, st, out->Mw.p, (size_t)nE, 1.0);
    IBH_HIP(hipGetLastError());
    IBH_HIP(hipStreamSynchronize(st));
}

}  // namespace ibh
