// capi.hip -- extern "C" entry points declared in include/icebin_hip.h.
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>

#include "assemble.h"
#include "common.h"
#include "prims.h"

namespace ibh {

static thread_local std::string g_last_error;
void set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }
void fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

// ---- caching device allocator ----------------------------------------------------------------
namespace {
struct Pool {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_blocks;   // (device, size class) -> blocks
    size_t cached_bytes = 0;
};
Pool &pool() { static Pool *p = new Pool; return *p; }      // leaked on purpose: outlives the HIP runtime teardown
constexpr size_t kMaxCached = size_t(32) << 30;

size_t size_class(size_t bytes) {      // 512 B granules below 64 KiB, then eighth-of-a-power-of-two steps (<= 12.5 % slack)
    if (bytes <= 65536) return (bytes + 511) & ~size_t(511);
    size_t p2 = 1;
    while ((p2 << 1) <= bytes) p2 <<= 1;
    const size_t step = p2 >> 3;
    return (bytes + step - 1) / step * step;
}
}  // namespace

void *dev_alloc(size_t bytes, size_t *granted, int *device) {
    const size_t cls = size_class(bytes);
    int dev = 0;
    IBH_HIP(hipGetDevice(&dev));
    *device = dev;
    {
        std::lock_guard<std::mutex> lk(pool().mu);
        auto it = pool().free_blocks.find({dev, cls});
        if (it != pool().free_blocks.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            pool().cached_bytes -= cls;
            *granted = cls;
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, cls);
    if (e == hipErrorOutOfMemory) {          // give the cache back and retry once
        (void)hipGetLastError();
        release_cached_memory();
        e = hipMalloc(&p, cls);
    }
    if (e != hipSuccess) fail(e == hipErrorNoDevice ? IBH_ENODEVICE : IBH_EHIP, "hipMalloc(%zu) failed: %s", cls, hipGetErrorString(e));
    *granted = cls;
    return p;
}
void dev_free(void *p, size_t granted, int device) {
    if (!p) return;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return;      // runtime already gone (process exit)
    std::lock_guard<std::mutex> lk(pool().mu);
    if (pool().cached_bytes + granted > kMaxCached) { (void)hipFree(p); return; }
    pool().free_blocks[{device, granted}].push_back(p);   // the block's own device, whichever is current now
    pool().cached_bytes += granted;
}
void release_cached_memory() {
    std::lock_guard<std::mutex> lk(pool().mu);
    for (auto &kv : pool().free_blocks)
        for (void *p : kv.second) (void)hipFree(p);
    pool().free_blocks.clear();
    pool().cached_bytes = 0;
}

static void require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        fail(IBH_ENODEVICE, "no HIP device available (%s); libicebin_hip has no CPU fallback",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
}

static void check_weighted_device(const ibh_weighted *w) {
    IBH_CHECK(w != nullptr, "null Weighted handle");
    int dev = -1;
    IBH_HIP(hipGetDevice(&dev));
    IBH_CHECK(dev == w->device, "Weighted handle belongs to device %d, current device is %d", w->device, dev);
}

static void make_identity(ibh_sparse_set *s, int64_t n) {
    s->sparse_extent = n;
    s->n = (int32_t)n;
    s->identity = true;
}

}  // namespace ibh

using namespace ibh;

extern "C" {

const char *ibh_last_error(void) { return g_last_error.c_str(); }
int ibh_version(void) { return 100; }

int ibh_device_count(int *count) {
    return guarded([&] {
        IBH_CHECK(count != nullptr, "null argument");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        *count = e == hipSuccess ? n : 0;
    });
}
int ibh_set_device(int device) {
    return guarded([&] { IBH_HIP(hipSetDevice(device)); });
}

// ---- SparseSet -----------------------------------------------------------------------------
int ibh_sparse_set_create(int64_t sparse_extent, ibh_sparse_set **out) {
    return guarded([&] {
        IBH_CHECK(out != nullptr, "null argument");
        auto *s = new ibh_sparse_set;
        s->sparse_extent = sparse_extent;
        *out = s;
    });
}
int ibh_sparse_set_create_identity(int64_t n, ibh_sparse_set **out) {
    return guarded([&] {
        IBH_CHECK(out != nullptr && n >= 0 && n < (1ll << 31), "bad identity set size %ld", (long)n);
        auto *s = new ibh_sparse_set;
        make_identity(s, n);
        *out = s;
    });
}
int ibh_sparse_set_from_array(int64_t sparse_extent, const int64_t *to_sparse, int32_t n, ibh_sparse_set **out) {
    return guarded([&] {
        IBH_CHECK(out != nullptr && n >= 0 && (n == 0 || to_sparse != nullptr), "bad arguments");
        std::unique_ptr<ibh_sparse_set> s(new ibh_sparse_set);
        s->sparse_extent = sparse_extent;
        s->host.assign(to_sparse, to_sparse + n);
        s->n = s->host_n = n;
        std::vector<int64_t> sorted(s->host);
        std::sort(sorted.begin(), sorted.end());
        for (int32_t i = 0; i < n; ++i) {
            IBH_CHECK(sorted[(size_t)i] >= 0 && (sparse_extent < 0 || sorted[(size_t)i] < sparse_extent),
                      "sparse index %ld outside extent %ld", (long)sorted[(size_t)i], (long)sparse_extent);
            IBH_CHECK(i == 0 || sorted[(size_t)i] != sorted[(size_t)i - 1], "duplicate sparse index %ld", (long)sorted[(size_t)i]);
        }
        *out = s.release();
    });
}
int ibh_sparse_set_destroy(ibh_sparse_set *s) { delete s; return IBH_OK; }
int ibh_sparse_set_sparse_extent(const ibh_sparse_set *s, int64_t *out) {
    return guarded([&] { IBH_CHECK(s && out, "null argument"); *out = s->sparse_extent; });
}
int ibh_sparse_set_dense_extent(const ibh_sparse_set *s, int32_t *out) {
    return guarded([&] { IBH_CHECK(s && out, "null argument"); *out = s->dense_extent(); });
}
int ibh_sparse_set_to_sparse(const ibh_sparse_set *s, int64_t *out) {
    return guarded([&] {
        IBH_CHECK(s && (out || s->n == 0), "null argument");
        s->ensure_host();
        std::copy(s->host.begin(), s->host.begin() + s->n, out);
    });
}

int ibh_sparse_set_to_dense(const ibh_sparse_set *s, int64_t sparse, int32_t *dense) {
    return guarded([&] {
        IBH_CHECK(s && dense, "null argument");
        if (s->identity) { *dense = sparse >= 0 && sparse < s->n ? (int32_t)sparse : -1; return; }
        s->ensure_inverse();
        auto it = s->inv.find(sparse);
        *dense = it == s->inv.end() ? -1 : it->second;
    });
}
int ibh_sparse_set_add_dense(ibh_sparse_set *s, int64_t sparse, int32_t *dense) {
    return guarded([&] {
        IBH_CHECK(s && dense, "null argument");
        IBH_CHECK(sparse >= 0 && (s->sparse_extent < 0 || sparse < s->sparse_extent), "sparse index %ld outside extent %ld",
                  (long)sparse, (long)s->sparse_extent);
        if (s->identity && sparse < s->n) { *dense = (int32_t)sparse; return; }
        s->ensure_inverse();                     // materialises an identity prefix on the host
        auto it = s->inv.find(sparse);
        if (it != s->inv.end()) { *dense = it->second; return; }
        IBH_CHECK(s->n < 0x7fffffff, "dense extent overflows int32");
        s->identity = false;
        s->host.push_back(sparse);
        s->inv[sparse] = s->n;
        *dense = s->n++;
        s->host_n = s->inv_n = s->n;             // the device copy (entries [0, dev_n)) is completed by the next build
    });
}

// ---- exchange-grid generation ----------------------------------------------------------------
int ibh_exgrid_generate(const ibh_exgrid_desc *d, ibh_exgrid **out) {
    return guarded([&] {
        IBH_CHECK(d && out, "null argument");
        require_device();
        std::unique_ptr<ibh_exgrid> ex(new ibh_exgrid);
        exgrid_generate(d, ex.get());
        *out = ex.release();
    });
}
int ibh_exgrid_size(const ibh_exgrid *ex, int64_t *nX) {
    return guarded([&] { IBH_CHECK(ex && nX, "null argument"); *nX = ex->nX; });
}
int ibh_exgrid_get(const ibh_exgrid *ex, int32_t *indices, double *overlaps) {
    return guarded([&] {
        IBH_CHECK(ex && (ex->nX == 0 || (indices && overlaps)), "null argument");
        if (ex->nX == 0) return;
        ex->indices.download(indices, 2 * (size_t)ex->nX);
        ex->overlaps.download(overlaps, (size_t)ex->nX);
    });
}
int ibh_exgrid_destroy(ibh_exgrid *ex) { delete ex; return IBH_OK; }

// ---- Regridder -----------------------------------------------------------------------------
int ibh_regridder_create(const ibh_regridder_desc *d, ibh_regridder **out) {
    return guarded([&] {
        IBH_CHECK(d && out, "null argument");
        require_device();
        IBH_CHECK(d->nX >= 0 && d->nX < (1ll << 31) - 1, "nX=%ld out of range", (long)d->nX);
        IBH_CHECK(d->nI > 0 && d->nI < (1ll << 31), "nI=%ld out of range", (long)d->nI);
        IBH_CHECK(d->nA > 0 && d->nA < (1ll << 31), "nA=%ld out of range", (long)d->nA);
        IBH_CHECK(d->nA_dense >= 0 && d->nhc >= 0, "negative size");
        IBH_CHECK(d->nA * (int64_t)(d->nhc > 0 ? d->nhc : 1) < (1ll << 31), "nE = nA*nhc overflows int32 dense ids");
        IBH_CHECK(d->nX == 0 || (d->ex_indices && d->ex_area), "null exchange grid arrays");
        IBH_CHECK(d->nA_dense == 0 || (d->A_to_sparse && d->A_native_area && d->A_proj_area), "null A grid arrays");
        IBH_CHECK(d->interp_style == 0 || d->interp_style == 1, "unknown interp_style %d", d->interp_style);
        if (d->nhc > 0) {
            IBH_CHECK(d->hcdefs != nullptr, "null hcdefs");
            for (int k = 1; k < d->nhc; ++k) IBH_CHECK(d->hcdefs[k] > d->hcdefs[k - 1], "hcdefs must be ascending");
            const bool hc_slowest = d->hc_stride_A == 1 && d->hc_stride_HC == d->nA;
            const bool hc_fastest = d->hc_stride_HC == 1 && d->hc_stride_A == d->nhc;
            IBH_CHECK(hc_slowest || hc_fastest,
                      "indexingHC strides (%ld,%ld) are neither (1,nA) nor (nhc,1)", (long)d->hc_stride_A, (long)d->hc_stride_HC);
        }
        std::unique_ptr<ibh_regridder> g(new ibh_regridder);
        IBH_HIP(hipGetDevice(&g->device));
        g->nX = d->nX; g->nI = d->nI; g->nA = d->nA; g->nA_dense = d->nA_dense; g->nhc = d->nhc;
        g->interp_style = d->interp_style; g->hc_stride_A = d->hc_stride_A; g->hc_stride_HC = d->hc_stride_HC;
        g->A_to_sparse.assign(d->A_to_sparse, d->A_to_sparse + d->nA_dense);
        g->A_native.assign(d->A_native_area, d->A_native_area + d->nA_dense);
        g->A_proj.assign(d->A_proj_area, d->A_proj_area + d->nA_dense);
        g->hcdefs_h.assign(d->hcdefs, d->hcdefs + d->nhc);
        // ratio native/proj by sparse A index (the diagonal of sApvA, IceRegridder.cpp:41-48)
        std::vector<double> ratio((size_t)d->nA, 0.0);
        for (int32_t id = 0; id < d->nA_dense; ++id) {
            const int64_t s = d->A_to_sparse[id];
            IBH_CHECK(s >= 0 && s < d->nA, "A_to_sparse[%d]=%ld outside [0,%ld)", id, (long)s, (long)d->nA);
            IBH_CHECK(ratio[(size_t)s] == 0.0, "duplicate atmosphere cell %ld", (long)s);
            const double r = d->A_native_area[id] / d->A_proj_area[id];
            IBH_CHECK(std::isfinite(r) && r > 0, "atmosphere cell %ld: native/proj area ratio %g is not positive", (long)s, r);
            ratio[(size_t)s] = r;
        }
        for (int64_t x = 0; x < d->nX; ++x) {
            const int32_t iA = d->ex_indices[2 * x], iI = d->ex_indices[2 * x + 1];
            IBH_CHECK(iA >= 0 && iA < d->nA && iI >= 0 && iI < d->nI, "exchange cell %ld: (iA,iI)=(%d,%d) out of range", (long)x, iA, iI);
            IBH_CHECK(ratio[(size_t)iA] != 0.0, "exchange cell %ld overlaps atmosphere cell %d, which is not in agridA", (long)x, iA);
        }
        g->ex_indices.upload(d->ex_indices, (size_t)(2 * d->nX));
        g->ex_area.upload(d->ex_area, (size_t)d->nX);
        g->hcdefs.upload(d->hcdefs, (size_t)d->nhc);
        g->A_ratio_s.upload(ratio.data(), ratio.size());
        if (d->I_centroid_xy) {
            g->has_centroid = true;
            for (int k = 0; k < 2; ++k) { g->cmin[k] = 1e300; g->cmax[k] = -1e300; }
            for (int64_t i = 0; i < d->nI; ++i)
                for (int k = 0; k < 2; ++k) {
                    const double v = d->I_centroid_xy[2 * i + k];
                    IBH_CHECK(std::isfinite(v), "ice cell %ld has a non-finite centroid", (long)i);
                    g->cmin[k] = std::min(g->cmin[k], v); g->cmax[k] = std::max(g->cmax[k], v);
                }
            g->I_centroid.upload(d->I_centroid_xy, (size_t)(2 * d->nI));
        }
        IBH_HIP(hipStreamSynchronize(nullptr));
        *out = g.release();
    });
}
int ibh_regridder_destroy(ibh_regridder *rg) { delete rg; return IBH_OK; }
int ibh_regridder_sizes(const ibh_regridder *g, int64_t *nA, int64_t *nE, int64_t *nI, int64_t *nX, int32_t *nhc) {
    return guarded([&] {
        IBH_CHECK(g != nullptr, "null argument");
        if (nA) *nA = g->nA;
        if (nE) *nE = g->nA * (int64_t)g->nhc;      // GCMRegridder.hpp:273
        if (nI) *nI = g->nI;
        if (nX) *nX = g->nX;
        if (nhc) *nhc = g->nhc;
    });
}
int ibh_regridder_wA(const ibh_regridder *g, int native, double fill, double *out) {
    return guarded([&] {
        IBH_CHECK(g && out, "null argument");
        std::fill(out, out + g->nA, fill);
        const auto &areas = native ? g->A_native : g->A_proj;
        for (int32_t id = 0; id < g->nA_dense; ++id) out[g->A_to_sparse[(size_t)id]] = areas[(size_t)id];
    });
}

// ---- RegridMatrices ------------------------------------------------------------------------
int ibh_regrid_matrices_create(const ibh_regridder *rg, const double *elevmaskI, int64_t n, int scale, int correctA,
                               const double sigma[3], ibh_regrid_matrices **out) {
    return guarded([&] {
        IBH_CHECK(rg && elevmaskI && out, "null argument");
        // np_to_blitz shape check {nI}, icebin_cython.cpp:231
        IBH_CHECK(n == rg->nI, "elevmaskI has %ld elements, the ice grid has nI=%ld", (long)n, (long)rg->nI);
        int dev = -1;
        IBH_HIP(hipGetDevice(&dev));
        IBH_CHECK(dev == rg->device, "regridder belongs to device %d, current device is %d", rg->device, dev);
        std::unique_ptr<ibh_regrid_matrices> rm(new ibh_regrid_matrices);
        rm->rg = rg;
        rm->scale = scale; rm->correctA = correctA;
        if (sigma) for (int k = 0; k < 3; ++k) rm->sigma[k] = sigma[k];
        rm->elevmaskI.upload(elevmaskI, (size_t)n);
        IBH_HIP(hipStreamSynchronize(nullptr));
        elevmask_classes(rm.get(), nullptr);
        *out = rm.release();
    });
}
int ibh_regrid_matrices_create_device(const ibh_regridder *rg, const double *d_elevmaskI, int64_t n, int scale,
                                      int correctA, const double sigma[3], void *stream, ibh_regrid_matrices **out) {
    return guarded([&] {
        IBH_CHECK(rg && d_elevmaskI && out, "null argument");
        IBH_CHECK(n == rg->nI, "elevmaskI has %ld elements, the ice grid has nI=%ld", (long)n, (long)rg->nI);
        int dev = -1;
        IBH_HIP(hipGetDevice(&dev));
        IBH_CHECK(dev == rg->device, "regridder belongs to device %d, current device is %d", rg->device, dev);
        std::unique_ptr<ibh_regrid_matrices> rm(new ibh_regrid_matrices);
        rm->rg = rg;
        rm->scale = scale; rm->correctA = correctA;
        if (sigma) for (int k = 0; k < 3; ++k) rm->sigma[k] = sigma[k];
        hipStream_t st = static_cast<hipStream_t>(stream);
        rm->elevmaskI.alloc((size_t)n);
        if (!elevmask_classes(rm.get(), st, d_elevmaskI))       // (large grids: the copy and the class bytes in one pass)
            IBH_HIP(hipMemcpyAsync(rm->elevmaskI.p, d_elevmaskI, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
        IBH_HIP(hipStreamSynchronize(st));
        *out = rm.release();
    });
}
int ibh_regrid_matrices_destroy(ibh_regrid_matrices *rm) { delete rm; return IBH_OK; }

static void check_rm_device(const ibh_regrid_matrices *rm) {
    int dev = -1;
    IBH_HIP(hipGetDevice(&dev));
    IBH_CHECK(dev == rm->rg->device, "regridder belongs to device %d, current device is %d", rm->rg->device, dev);
}
int ibh_regrid_matrices_matrix_d(const ibh_regrid_matrices *rm, const char *spec, ibh_sparse_set *dim0,
                                 ibh_sparse_set *dim1, int scale, int correctA, const double sigma[3],
                                 ibh_weighted **out) {
    return guarded([&] {
        IBH_CHECK(rm && spec && out, "null argument");
        check_rm_device(rm);
        IBH_CHECK(dim0 == nullptr || dim0 != dim1, "dims[0] and dims[1] must be distinct sets");
        assemble_matrix(rm, spec, dim0, dim1, scale, correctA, sigma, out);
    });
}
int ibh_regrid_matrices_matrix_d_sharded(const ibh_regrid_matrices *rm, ibh_comm *c, const char *spec, ibh_sparse_set *dim0,
                                         ibh_sparse_set *dim1, int scale, int correctA, ibh_weighted **out) {
    return guarded([&] {
        IBH_CHECK(rm && c && spec && out, "null argument");
        check_rm_device(rm);
        IBH_CHECK(dim0 == nullptr || dim0 != dim1, "dims[0] and dims[1] must be distinct sets");
        const double sigma[3] = {0, 0, 0};
        assemble_matrix(rm, spec, dim0, dim1, scale, correctA, sigma, out, false, c);
    });
}
int ibh_regrid_matrices_matrix_batch(const ibh_regrid_matrices *rm, int32_t n, const char *const *specs,
                                     ibh_sparse_set *const *dim0, ibh_sparse_set *const *dim1, const int32_t *scale,
                                     const int32_t *correctA, const double sigma[3], ibh_weighted **out) {
    return guarded([&] {
        IBH_CHECK(rm && out, "null argument");
        check_rm_device(rm);
        assemble_batch(rm, n, specs, dim0, dim1, scale, correctA, sigma, out);
    });
}
int ibh_regrid_matrices_matrix(const ibh_regrid_matrices *rm, const char *spec, ibh_weighted **out) {
    return guarded([&] {
        IBH_CHECK(rm && spec && out, "null argument");
        check_rm_device(rm);
        assemble_matrix(rm, spec, nullptr, nullptr, rm->scale, rm->correctA, rm->sigma, out);
    });
}

// ---- Weighted ------------------------------------------------------------------------------
static std::unique_ptr<ibh_weighted> new_loaded(int32_t nrow, int32_t ncol, const double *wM, const double *Mw,
                                                int conservative, int scaled) {
    require_device();
    IBH_CHECK(nrow >= 0 && ncol >= 0, "negative shape");
    IBH_CHECK((nrow == 0 || wM) && (ncol == 0 || Mw), "null weight vector");
    std::unique_ptr<ibh_weighted> w(new ibh_weighted);
    IBH_HIP(hipGetDevice(&w->device));
    w->nrow = nrow; w->ncol = ncol;
    w->conservative = conservative; w->scaled = scaled;
    for (int k = 0; k < 2; ++k) { w->dims[k] = new ibh_sparse_set; w->owns[k] = true; }
    make_identity(w->dims[0], nrow);
    make_identity(w->dims[1], ncol);
    w->wM.upload(wM, (size_t)nrow);
    w->Mw.upload(Mw, (size_t)ncol);
    return w;
}

int ibh_weighted_from_coo(int32_t nrow, int32_t ncol, int64_t nnz, const int32_t *row, const int32_t *col,
                          const double *val, const double *wM, const double *Mw, int conservative, int scaled,
                          ibh_weighted **out) {
    return guarded([&] {
        IBH_CHECK(out != nullptr && nnz >= 0 && nnz < (1ll << 31), "bad nnz %ld", (long)nnz);
        IBH_CHECK(nnz == 0 || (row && col && val), "null triplet arrays");
        for (int64_t k = 0; k < nnz; ++k)
            IBH_CHECK(row[k] >= 0 && row[k] < nrow && col[k] >= 0 && col[k] < ncol,
                      "triplet %ld: (%d,%d) outside %d x %d", (long)k, row[k], col[k], nrow, ncol);
        auto w = new_loaded(nrow, ncol, wM, Mw, conservative, scaled);
        weighted_from_coo_device(w.get(), nrow, ncol, nnz, row, col, val);
        *out = w.release();
    });
}
int ibh_weighted_from_csr(int32_t nrow, int32_t ncol, const int32_t *rowptr, const int32_t *colind, const double *val,
                          const double *wM, const double *Mw, int conservative, int scaled, ibh_weighted **out) {
    return guarded([&] {
        IBH_CHECK(out && rowptr, "null argument");
        IBH_CHECK(rowptr[0] == 0, "rowptr[0] must be 0");
        for (int32_t r = 0; r < nrow; ++r) IBH_CHECK(rowptr[r + 1] >= rowptr[r], "rowptr not monotone at row %d", r);
        const int64_t nnz = rowptr[nrow];
        IBH_CHECK(nnz == 0 || (colind && val), "null CSR arrays");
        for (int64_t k = 0; k < nnz; ++k) IBH_CHECK(colind[k] >= 0 && colind[k] < ncol, "colind[%ld]=%d outside [0,%d)", (long)k, colind[k], ncol);
        auto w = new_loaded(nrow, ncol, wM, Mw, conservative, scaled);
        w->nnz = nnz;
        w->rowptr.upload(rowptr, (size_t)nrow + 1);
        w->colind.upload(colind, (size_t)nnz);
        w->val.upload(val, (size_t)nnz);
        IBH_HIP(hipStreamSynchronize(nullptr));
        *out = w.release();
    });
}
int ibh_e1ve0_compute(int32_t nsheets, const ibh_weighted *const *XuE1s, const ibh_weighted *const *XuE0s, int64_t nE,
                      ibh_weighted **out) {
    return guarded([&] {
        IBH_CHECK(out && nsheets >= 1 && XuE1s && XuE0s, "bad arguments");
        require_device();
        for (int s = 0; s < nsheets; ++s) { check_weighted_device(XuE1s[s]); check_weighted_device(XuE0s[s]); }
        std::unique_ptr<ibh_weighted> w(new ibh_weighted);
        IBH_HIP(hipGetDevice(&w->device));
        for (int k = 0; k < 2; ++k) { w->dims[k] = new ibh_sparse_set; w->owns[k] = true; make_identity(w->dims[k], nE); }
        e1ve0_compute(nsheets, XuE1s, XuE0s, nE, w.get());
        *out = w.release();
    });
}
int ibh_weighted_destroy(ibh_weighted *w) { delete w; return IBH_OK; }

int ibh_weighted_shape(const ibh_weighted *w, int32_t *nrow, int32_t *ncol, int64_t *nnz) {
    return guarded([&] {
        IBH_CHECK(w != nullptr, "null argument");
        if (nrow) *nrow = w->nrow;
        if (ncol) *ncol = w->ncol;
        if (nnz) *nnz = w->nnz;
    });
}
int ibh_weighted_flags(const ibh_weighted *w, int *conservative, int *scaled) {
    return guarded([&] {
        IBH_CHECK(w != nullptr, "null argument");
        if (conservative) *conservative = w->conservative;
        if (scaled) *scaled = w->scaled;
    });
}
int ibh_weighted_dim(const ibh_weighted *w, int k, int64_t *sparse_extent, int32_t *dense_extent) {
    return guarded([&] {
        IBH_CHECK(w && (k == 0 || k == 1), "bad argument");
        if (sparse_extent) *sparse_extent = w->dims[k]->sparse_extent;
        if (dense_extent) *dense_extent = k == 0 ? w->nrow : w->ncol;
    });
}
int ibh_weighted_dim_to_sparse(const ibh_weighted *w, int k, int64_t *out) {
    return guarded([&] {
        IBH_CHECK(w && (k == 0 || k == 1), "bad argument");
        // a shared dims set may have grown after this matrix was built: report this matrix's extent
        const int32_t n = k == 0 ? w->nrow : w->ncol;
        IBH_CHECK(n == 0 || out, "null output");
        w->dims[k]->ensure_host();
        std::copy(w->dims[k]->host.begin(), w->dims[k]->host.begin() + n, out);
    });
}
int ibh_weighted_get_wM(const ibh_weighted *w, double *out) {
    return guarded([&] { check_weighted_device(w); w->wM.download(out, (size_t)w->nrow); });
}
int ibh_weighted_get_Mw(const ibh_weighted *w, double *out) {
    return guarded([&] { check_weighted_device(w); w->Mw.download(out, (size_t)w->ncol); });
}
int ibh_weighted_get_csr(const ibh_weighted *w, int32_t *rowptr, int32_t *colind, double *val) {
    return guarded([&] {
        check_weighted_device(w);
        if (rowptr) w->rowptr.download(rowptr, (size_t)w->nrow + 1);
        if (colind) w->colind.download(colind, (size_t)w->nnz);
        if (val) w->val.download(val, (size_t)w->nnz);
    });
}
int ibh_weighted_get_coo(const ibh_weighted *w, int32_t *row, int32_t *col, double *val) {
    return guarded([&] {
        check_weighted_device(w);
        if (row) {
            std::vector<int32_t> rp((size_t)w->nrow + 1);
            w->rowptr.download(rp.data(), rp.size());
            for (int32_t r = 0; r < w->nrow; ++r)
                for (int32_t k = rp[(size_t)r]; k < rp[(size_t)r + 1]; ++k) row[k] = r;
        }
        if (col) w->colind.download(col, (size_t)w->nnz);
        if (val) w->val.download(val, (size_t)w->nnz);
    });
}

int ibh_weighted_apply_device(const ibh_weighted *w, const double *dA, int32_t nvar, int64_t lda, double *dB,
                              int64_t ldb, double fill, int force_conservation, void *stream) {
    return guarded([&] {
        check_weighted_device(w);
        IBH_CHECK(nvar >= 0 && (nvar == 0 || (dA && dB)), "bad arguments");
        spmm_launch(w, dA, nvar, lda, dB, ldb, fill, force_conservation, static_cast<hipStream_t>(stream));
    });
}
int ibh_weighted_apply_many_device(const ibh_weighted *w, int32_t nbatch, const double *const *dA, int32_t nvar,
                                   int64_t lda, double *const *dB, int64_t ldb, double fill, int force_conservation,
                                   void *stream) {
    return guarded([&] {
        check_weighted_device(w);
        IBH_CHECK(nbatch >= 0 && nvar >= 0 && (nbatch == 0 || nvar == 0 || (dA && dB)), "bad arguments");
        spmm_launch_many(w, nbatch, dA, nvar, lda, dB, ldb, fill, force_conservation, static_cast<hipStream_t>(stream));
    });
}
int ibh_weighted_reserve(const ibh_weighted *w, int32_t nvar) {
    return guarded([&] {
        check_weighted_device(w);
        IBH_CHECK(nvar >= 0, "bad arguments");
        weighted_reserve(w, nvar);
    });
}
int ibh_weighted_prepare(const ibh_weighted *w, int32_t nvar, int32_t nbatch) {
    return guarded([&] {
        check_weighted_device(w);
        IBH_CHECK(nvar >= 0 && nbatch >= 0, "bad arguments");
        weighted_prepare(w, nvar, nbatch);
    });
}
int ibh_weighted_pair_prepare(const ibh_weighted *first, const ibh_weighted *second, int32_t nvar) {
    return guarded([&] {
        check_weighted_device(first);
        check_weighted_device(second);
        IBH_CHECK(nvar >= 0, "bad arguments");
        weighted_pair_prepare(first, second, nvar);
    });
}
int ibh_weighted_apply_pair_device(const ibh_weighted *first, const ibh_weighted *second, const double *dA, int32_t nvar, int64_t lda,
                                   double *dB1, int64_t ldb1, double *dB2, int64_t ldb2, double fill, void *stream) {
    return guarded([&] {
        check_weighted_device(first);
        check_weighted_device(second);
        IBH_CHECK(nvar >= 0 && (nvar == 0 || (dA && dB1 && dB2)), "bad arguments");
        spmm_launch_pair(first, second, dA, nvar, lda, dB1, ldb1, dB2, ldb2, fill, static_cast<hipStream_t>(stream));
    });
}
int ibh_weighted_apply_chain_device(const ibh_weighted *first, const ibh_weighted *second, const ibh_weighted *third, const double *dA, int32_t nvar,
                                    int64_t lda, double *dB1, int64_t ldb1, double *dB2, int64_t ldb2, double *dB3, int64_t ldb3, double fill,
                                    void *stream) {
    return guarded([&] {
        check_weighted_device(first);
        check_weighted_device(second);
        check_weighted_device(third);
        IBH_CHECK(nvar >= 0 && (nvar == 0 || (dA && dB1 && dB2 && dB3)), "bad arguments");
        IBH_CHECK(nvar == 0 || ldb3 >= third->nrow, "chain apply: leading dimension %ld smaller than %d rows", (long)ldb3, third->nrow);
        spmm_launch_chain(first, second, third, dA, nvar, lda, dB1, ldb1, dB2, ldb2, dB3, ldb3, fill, static_cast<hipStream_t>(stream));
    });
}
int ibh_weighted_apply_host(const ibh_weighted *w, const double *A_b, int32_t nvar, int64_t lda, double *B_b,
                            int64_t ldb, double fill, int force_conservation) {
    return guarded([&] {
        check_weighted_device(w);
        IBH_CHECK(nvar >= 0 && (nvar == 0 || (A_b && B_b)), "bad arguments");
        IBH_CHECK(lda >= w->ncol && ldb >= w->nrow, "leading dimensions too small");
        if (nvar == 0) return;
        // device staging with 512-byte-aligned field planes (whole-line wave stores, see spmm.hip shortrow)
        const size_t dlda = ((size_t)w->ncol + 63) & ~size_t(63), dldb = ((size_t)w->nrow + 63) & ~size_t(63);
        DevBuf<double> dA((size_t)nvar * dlda), dB((size_t)nvar * dldb);
        IBH_HIP(hipMemcpy2DAsync(dA.p, sizeof(double) * dlda, A_b, sizeof(double) * (size_t)lda,
                                 sizeof(double) * (size_t)w->ncol, (size_t)nvar, hipMemcpyHostToDevice, nullptr));
        spmm_launch(w, dA.p, nvar, (int64_t)dlda, dB.p, (int64_t)dldb, fill, force_conservation, nullptr);
        IBH_HIP(hipMemcpy2DAsync(B_b, sizeof(double) * (size_t)ldb, dB.p, sizeof(double) * dldb,
                                 sizeof(double) * (size_t)w->nrow, (size_t)nvar, hipMemcpyDeviceToHost, nullptr));
        IBH_HIP(hipStreamSynchronize(nullptr));
    });
}
int ibh_weighted_apply_transformed_device(const ibh_weighted *w, const double *dA, int32_t nvar_in, int64_t lda,
                                          const double *T, const double *b, int32_t nvar_out, double *dB,
                                          int64_t ldb, double fill, void *stream) {
    return guarded([&] {
        check_weighted_device(w);
        IBH_CHECK(nvar_in >= 0 && nvar_out >= 0 && (nvar_out == 0 || (dA && dB && T && b)), "bad arguments");
        spmm_transformed_launch(w, dA, nvar_in, lda, T, b, nvar_out, dB, ldb, fill, static_cast<hipStream_t>(stream));
    });
}
int ibh_weighted_matvec_device(const ibh_weighted *w, const double *dxx, int32_t nvar, int64_t ldx, double *dyy,
                               int64_t ldy, int ignore_nan, void *stream) {
    return guarded([&] {
        check_weighted_device(w);
        IBH_CHECK(nvar >= 0 && (nvar == 0 || (dxx && dyy)), "bad arguments");
        matvec_legacy_launch(w, dxx, nvar, ldx, dyy, ldy, ignore_nan, static_cast<hipStream_t>(stream));
    });
}
int ibh_coo_matvec(double *yy, const double *xx, int ignore_nan, int64_t nrow, int64_t ncol, int64_t nnz,
                   const int32_t *row, const int32_t *col, const double *data) {
    return guarded([&] {
        IBH_CHECK(nrow >= 0 && ncol >= 0 && nrow < (1ll << 31) && ncol < (1ll << 31), "bad shape");
        IBH_CHECK(nnz >= 0 && nnz < (1ll << 31) && (nnz == 0 || (row && col && data)), "bad triplets");
        IBH_CHECK((nrow == 0 || yy) && (ncol == 0 || xx), "null vector");
        for (int64_t k = 0; k < nnz; ++k)
            IBH_CHECK(row[k] >= 0 && row[k] < nrow && col[k] >= 0 && col[k] < ncol,
                      "triplet %ld: (%d,%d) outside %ld x %ld", (long)k, row[k], col[k], (long)nrow, (long)ncol);
        if (nrow == 0 || nnz == 0) return;
        std::vector<double> ones_r((size_t)nrow, 1.0), ones_c((size_t)ncol, 1.0);
        auto w = new_loaded((int32_t)nrow, (int32_t)ncol, ones_r.data(), ones_c.data(), 1, 1);
        weighted_from_coo_device(w.get(), (int)nrow, (int)ncol, nnz, row, col, data);
        DevBuf<double> dx((size_t)ncol), dy((size_t)nrow);
        dx.upload(xx, (size_t)ncol);
        dy.upload(yy, (size_t)nrow);
        matvec_legacy_launch(w.get(), dx.p, 1, ncol, dy.p, nrow, ignore_nan, nullptr);
        dy.download(yy, (size_t)nrow);
    });
}
int ibh_weighted_apply_weight_host(const ibh_weighted *w, int dim, const double *A_b, int32_t nvar, int64_t lda,
                                   double *out) {
    return guarded([&] {
        check_weighted_device(w);
        IBH_CHECK((dim == 0 || dim == 1) && nvar >= 0 && (nvar == 0 || (A_b && out)), "bad arguments");
        const int n = dim == 0 ? w->nrow : w->ncol;
        IBH_CHECK(lda >= n, "leading dimension too small");
        if (nvar == 0) return;
        DevBuf<double> dA((size_t)nvar * (size_t)n), dout((size_t)nvar), dpart(weight_dot_scratch(n, nvar));
        IBH_HIP(hipMemcpy2DAsync(dA.p, sizeof(double) * (size_t)n, A_b, sizeof(double) * (size_t)lda,
                                 sizeof(double) * (size_t)n, (size_t)nvar, hipMemcpyHostToDevice, nullptr));
        weight_dot_launch(dim == 0 ? w->wM.p : w->Mw.p, n, dA.p, nvar, n, dout.p, dpart.p, nullptr);
        dout.download(out, (size_t)nvar);
    });
}

int ibh_weighted_device_view_get(const ibh_weighted *w, ibh_weighted_device_view *out) {
    return guarded([&] {
        IBH_CHECK(w && out, "null argument");
        *out = ibh_weighted_device_view{w->nrow, w->ncol, w->nnz, w->rowptr.p, w->colind.p, w->val.p, w->wM.p, w->Mw.p};
    });
}

int ibh_weighted_set_kernel(ibh_weighted *w, const char *name) {
    return guarded([&] {
        IBH_CHECK(w && name, "null argument");
        if (!strcmp(name, "auto")) w->kernel_override = 0;
        else if (!strcmp(name, "rowblock")) w->kernel_override = 1;
        else if (!strcmp(name, "shortrow")) w->kernel_override = 2;
        else if (!strcmp(name, "colsweep")) w->kernel_override = 4;       // falls back to rowblock when the matrix has no column-sweep structure
        else if (!strcmp(name, "rowdual")) w->kernel_override = 3;        // falls back to rowblock when the matrix has no bands
        else if (!strcmp(name, "rowgroup")) w->kernel_override = 5;       // falls back to rowblock when the matrix has no row groups
        else fail(IBH_EINVAL, "unknown kernel '%s'", name);
    });
}
int ibh_weighted_last_kernel(const ibh_weighted *w, char *buf, int buflen) {
    return guarded([&] {
        IBH_CHECK(w && buf && buflen > 0, "bad argument");
        snprintf(buf, (size_t)buflen, "%s", w->last_kernel == 1 ? "rowblock" : w->last_kernel == 2 ? "shortrow" : w->last_kernel == 3 ? "rowdual" : w->last_kernel == 4 ? "colsweep" : w->last_kernel == 5 ? "rowgroup" : "none");
    });
}
int ibh_weighted_last_launch(const ibh_weighted *w, char *buf, int buflen) {
    return guarded([&] {
        IBH_CHECK(w && buf && buflen > 0, "bad arguments");
        snprintf(buf, (size_t)buflen, "%s", w->last_sig);
    });
}
int ibh_weighted_set_option(ibh_weighted *w, const char *key, int value) {
    return guarded([&] {
        IBH_CHECK(w && key, "null argument");
        if (value == INT32_MIN) w->opts.erase(key);
        else w->opts[key] = value;
    });
}
int ibh_weighted_built_fast(const ibh_weighted *w, int *out) {
    return guarded([&] { IBH_CHECK(w && out, "null argument"); *out = w->built_fast; });
}
int ibh_selftest_sort(const uint64_t *keys, int64_t n, int lo_bits, int hi_bits, uint32_t *perm_out, int *path_out) {
    return guarded([&] {
        IBH_CHECK(n >= 0 && n < (1ll << 31) && (n == 0 || (keys && perm_out)) && path_out, "bad arguments");
        Arena &A = arena();
        A.reset();
        *path_out = 0;
        if (n == 0) return;
        hipStream_t st = nullptr;
        uint64_t *k = A.get<uint64_t>((size_t)n), *k2 = A.get<uint64_t>((size_t)n);
        uint32_t *v = A.get<uint32_t>((size_t)n), *v2 = A.get<uint32_t>((size_t)n);
        OrderInfo *info = A.get<OrderInfo>(1);
        IBH_HIP(hipMemcpyAsync(k, keys, sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, st));
        std::vector<uint32_t> iota((size_t)n);
        for (int64_t i = 0; i < n; ++i) iota[(size_t)i] = (uint32_t)i;
        IBH_HIP(hipMemcpyAsync(v, iota.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, st));
        const bool alt = adaptive_sort_pairs(k, k2, v, v2, (size_t)n, lo_bits, hi_bits, info, st);
        OrderInfo h{};
        if (n >= 2) IBH_HIP(hipMemcpyAsync(&h, info, sizeof(h), hipMemcpyDeviceToHost, st));
        IBH_HIP(hipMemcpyAsync(perm_out, alt ? v2 : v, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
        IBH_HIP(hipStreamSynchronize(st));
        *path_out = !(h.flags & ORD_FULL_DEC) ? 0 : h.maxlen <= (uint32_t)CS_BIG ? 1 : 2;
    });
}
int ibh_event_create(void **out) {
    return guarded([&] {
        IBH_CHECK(out != nullptr, "null argument");
        require_device();
        hipEvent_t e;
        IBH_HIP(hipEventCreate(&e));
        *out = e;
    });
}
int ibh_event_destroy(void *ev) {
    return guarded([&] { if (ev) IBH_HIP(hipEventDestroy(static_cast<hipEvent_t>(ev))); });
}
int ibh_event_elapsed_ms(void *start, void *stop, float *ms) {
    return guarded([&] {
        IBH_CHECK(start && stop && ms, "null argument");
        IBH_HIP(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
        IBH_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    });
}
int ibh_set_launch_events(void *start, void *stop) {
    return guarded([&] { set_launch_events(static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)); });
}
int ibh_release_cached_memory(void) {
    return guarded([&] { release_workspace(); release_cached_memory(); });
}
int ibh_set_tuning(const char *key, int value) {
    return guarded([&] { IBH_CHECK(key != nullptr, "null key"); set_tuning(key, value); });
}

}  // extern "C"
