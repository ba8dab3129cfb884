// gridgen.hip -- exchange-grid generation for a rectilinear XY ice grid under convex (projected) GCM cells.
//
// Replaces make_exchange_grid (slib/icebin/gridgen/GridGen_Exchange.cpp:175-284) for the case the
// regrid path is quoted on: ice cells are axis-aligned rectangles in the projected plane
// (gridgen/searise_grid.cpp), GCM cells are lon/lat boxes whose corners the caller has projected to
// that plane (OGrid, GridGen_Exchange.cpp:120-166) -- convex polygons of a few vertices.  The reference
// finds candidate pairs with an RTree over the ice cells and intersects polygons with CGAL; here the
// candidates of a GCM cell are the index ranges its bounding box covers (the ice grid is rectilinear:
// two binary searches), one thread clips the polygon against one ice cell (Sutherland-Hodgman against
// four axis-aligned half-planes) and takes the area with the reference's own formula
// (Cell::proj_area, Grid.cpp:42-70).  Cells are emitted ordered by (iA, iI) ascending -- the order of
// ExchangeGrid's constructor (AbbrGrid.cpp:10-21) -- by construction: candidates are enumerated in that
// order and compacted with a scan.  Byte/index work bound by HBM traffic; double arithmetic without
// contraction (the CGAL result is exact rationals rounded to double: areas agree to rounding).
#include "assemble.h"
#include "prims.h"

namespace ibh {

constexpr int GG_MAXV = 16;          // vertices of a GCM-cell polygon
constexpr int GG_CLIPV = GG_MAXV + 4;

struct GridGenView {
    const double *xe, *ye;           // ice-cell edges [nx+1], [ny+1], ascending
    int nx, ny, x_fastest;
    const int32_t *polyptr;          // [npoly+1]
    const double *vx, *vy;           // polygon vertices, counter-clockwise
    int npoly;
};
struct CandRange { int ix0, nxr, iy0, nyr; };

__device__ __forceinline__ int first_edge_gt(const double *e, int n, double v) {     // first k with e[k] > v
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (e[mid] > v) hi = mid; else lo = mid + 1; }
    return lo;
}
__device__ __forceinline__ int first_edge_ge(const double *e, int n, double v) {     // first k with e[k] >= v
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (e[mid] >= v) hi = mid; else lo = mid + 1; }
    return lo;
}
// ice cells whose open interior can meet (lo, hi): cell k = [e[k], e[k+1]] needs e[k+1] > lo and e[k] < hi
__device__ __forceinline__ void cell_range(const double *e, int n, double lo, double hi, int &k0, int &cnt) {
    k0 = first_edge_gt(e, n + 1, lo) - 1;
    if (k0 < 0) k0 = 0;
    int k1 = first_edge_ge(e, n + 1, hi);        // cells k < k1 have e[k] < hi
    if (k1 > n) k1 = n;
    cnt = k1 > k0 ? k1 - k0 : 0;
}
__global__ void k_gg_ranges(GridGenView g, CandRange *__restrict__ rng, uint32_t *__restrict__ cnt) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.npoly) return;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int k = g.polyptr[p]; k < g.polyptr[p + 1]; ++k) {
        xmin = fmin(xmin, g.vx[k]); xmax = fmax(xmax, g.vx[k]);
        ymin = fmin(ymin, g.vy[k]); ymax = fmax(ymax, g.vy[k]);
    }
    CandRange r;
    cell_range(g.xe, g.nx, xmin, xmax, r.ix0, r.nxr);
    cell_range(g.ye, g.ny, ymin, ymax, r.iy0, r.nyr);
    rng[p] = r;
    cnt[p] = (uint32_t)r.nxr * (uint32_t)r.nyr;
}
// keep the part of the polygon on the inner side of one axis-aligned line: coordinate `axis` (0 x, 1 y),
// inside = (c >= bound) when lower, (c <= bound) otherwise
__device__ __forceinline__ int clip_axis(const double *ix, const double *iy, int n, double *ox, double *oy, int axis,
                                         double bound, bool lower) {
    int m = 0;
    for (int k = 0; k < n; ++k) {
        const int kn = k + 1 == n ? 0 : k + 1;
        const double ax = ix[k], ay = iy[k], bx = ix[kn], by = iy[kn];
        const double ca = axis ? ay : ax, cb = axis ? by : bx;
        const bool ina = lower ? ca >= bound : ca <= bound, inb = lower ? cb >= bound : cb <= bound;
        if (ina) { ox[m] = ax; oy[m] = ay; ++m; }
        if (ina != inb) {
            const double t = (bound - ca) / (cb - ca);
            if (axis) { ox[m] = ax + t * (bx - ax); oy[m] = bound; }
            else      { ox[m] = bound; oy[m] = ay + t * (by - ay); }
            ++m;
        }
    }
    return m;
}
__global__ void k_gg_clip(GridGenView g, const CandRange *__restrict__ rng, const uint32_t *__restrict__ candptr, uint32_t ncand,
                          double *__restrict__ area, uint32_t *__restrict__ keep, int32_t *__restrict__ cellI) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncand) return;
    int lo = 0, hi = g.npoly;                      // polygon p with candptr[p] <= t < candptr[p+1]
    while (lo + 1 < hi) { const int mid = (lo + hi) >> 1; if (candptr[mid] <= t) lo = mid; else hi = mid; }
    const int p = lo;
    const CandRange r = rng[p];
    const uint32_t local = t - candptr[p];
    int ix, iy;
    if (g.x_fastest) { iy = r.iy0 + (int)(local / (uint32_t)r.nxr); ix = r.ix0 + (int)(local % (uint32_t)r.nxr); }
    else             { ix = r.ix0 + (int)(local / (uint32_t)r.nyr); iy = r.iy0 + (int)(local % (uint32_t)r.nyr); }
    double ax[GG_CLIPV], ay[GG_CLIPV], bx[GG_CLIPV], by[GG_CLIPV];
    int n = g.polyptr[p + 1] - g.polyptr[p];
    for (int k = 0; k < n; ++k) { ax[k] = g.vx[g.polyptr[p] + k]; ay[k] = g.vy[g.polyptr[p] + k]; }
    n = clip_axis(ax, ay, n, bx, by, 0, g.xe[ix], true);
    n = clip_axis(bx, by, n, ax, ay, 0, g.xe[ix + 1], false);
    n = clip_axis(ax, ay, n, bx, by, 1, g.ye[iy], true);
    n = clip_axis(bx, by, n, ax, ay, 1, g.ye[iy + 1], false);
    double ret = 0;                                // Cell::proj_area, Grid.cpp:42-70
    if (n >= 3) {
        double x0 = ax[n - 1], y0 = ay[n - 1];
        for (int k = 0; k < n; ++k) {
            const double x1 = ax[k], y1 = ay[k];
            ret += (x0 * y1) - (x1 * y0);
            x0 = x1; y0 = y1;
        }
        ret *= .5;
    }
    area[t] = ret;
    keep[t] = ret > 0 ? 1u : 0u;                   // an empty / degenerate overlap is no exchange cell (:183-184)
    cellI[t] = g.x_fastest ? iy * g.nx + ix : ix * g.ny + iy;
}
__global__ void k_gg_emit(const int64_t *__restrict__ iA, const uint32_t *__restrict__ candptr, int npoly, uint32_t ncand,
                          const double *__restrict__ area, const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos,
                          const int32_t *__restrict__ cellI, int32_t *__restrict__ indices, double *__restrict__ overlaps) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncand || !keep[t]) return;
    int lo = 0, hi = npoly;
    while (lo + 1 < hi) { const int mid = (lo + hi) >> 1; if (candptr[mid] <= t) lo = mid; else hi = mid; }
    const uint32_t q = pos[t];
    indices[2 * (size_t)q] = (int32_t)iA[lo];
    indices[2 * (size_t)q + 1] = cellI[t];
    overlaps[q] = area[t];
}

void exgrid_generate(const ibh_exgrid_desc *d, ibh_exgrid *out) {
    IBH_CHECK(d->nx > 0 && d->ny > 0 && d->xedges && d->yedges, "ice grid: bad sizes / null edges");
    IBH_CHECK((int64_t)d->nx * d->ny < (1ll << 31), "nI overflows int32");
    IBH_CHECK(d->npoly >= 0 && (d->npoly == 0 || (d->polyptr && d->vx && d->vy && d->iA)), "null polygon arrays");
    for (int k = 0; k < d->nx; ++k) IBH_CHECK(d->xedges[k + 1] > d->xedges[k], "x edges must be ascending");
    for (int k = 0; k < d->ny; ++k) IBH_CHECK(d->yedges[k + 1] > d->yedges[k], "y edges must be ascending");
    for (int p = 0; p < d->npoly; ++p) {
        const int n = d->polyptr[p + 1] - d->polyptr[p];
        IBH_CHECK(n >= 3 && n <= GG_MAXV, "polygon %d has %d vertices (3..%d supported)", p, n, GG_MAXV);
        IBH_CHECK(d->iA[p] >= 0 && d->iA[p] < (1ll << 31), "polygon %d: iA out of range", p);
        IBH_CHECK(p == 0 || d->iA[p] > d->iA[p - 1], "polygons must come in ascending iA order (cells.sorted(), AbbrGrid.cpp:15)");
    }
    hipStream_t st = nullptr;
    Arena &A = arena();
    A.reset();
    const int T = 256, np = d->npoly;
    out->nX = 0;
    if (np == 0) return;
    const int nv = d->polyptr[np];
    double *xe = A.get<double>((size_t)d->nx + 1), *ye = A.get<double>((size_t)d->ny + 1);
    double *vx = A.get<double>((size_t)nv), *vy = A.get<double>((size_t)nv);
    int32_t *pp = A.get<int32_t>((size_t)np + 1);
    int64_t *iA = A.get<int64_t>((size_t)np);
    IBH_HIP(hipMemcpyAsync(xe, d->xedges, sizeof(double) * ((size_t)d->nx + 1), hipMemcpyHostToDevice, st));
    IBH_HIP(hipMemcpyAsync(ye, d->yedges, sizeof(double) * ((size_t)d->ny + 1), hipMemcpyHostToDevice, st));
    IBH_HIP(hipMemcpyAsync(vx, d->vx, sizeof(double) * (size_t)nv, hipMemcpyHostToDevice, st));
    IBH_HIP(hipMemcpyAsync(vy, d->vy, sizeof(double) * (size_t)nv, hipMemcpyHostToDevice, st));
    IBH_HIP(hipMemcpyAsync(pp, d->polyptr, sizeof(int32_t) * ((size_t)np + 1), hipMemcpyHostToDevice, st));
    IBH_HIP(hipMemcpyAsync(iA, d->iA, sizeof(int64_t) * (size_t)np, hipMemcpyHostToDevice, st));
    GridGenView g{xe, ye, d->nx, d->ny, d->x_fastest, pp, vx, vy, np};
    CandRange *rng = A.get<CandRange>((size_t)np);
    uint32_t *cnt = A.get<uint32_t>((size_t)np + 1), *candptr = A.get<uint32_t>((size_t)np + 1);
    hipLaunchKernelGGL(k_gg_ranges, dim3(ceil_div(np, T)), dim3(T), 0, st, g, rng, cnt);
    exclusive_scan_u32(cnt, candptr, (size_t)np, candptr + np, st);
    uint32_t ncand = 0;
    readback_sync(&ncand, candptr + np, sizeof(uint32_t), st);
    IBH_CHECK(ncand < (1u << 31), "too many candidate pairs");
    if (ncand == 0) return;
    double *area = A.get<double>(ncand);
    uint32_t *keep = A.get<uint32_t>((size_t)ncand + 1), *pos = A.get<uint32_t>((size_t)ncand + 1);
    int32_t *cellI = A.get<int32_t>(ncand);
    hipLaunchKernelGGL(k_gg_clip, dim3(ceil_div(ncand, T)), dim3(T), 0, st, g, rng, candptr, ncand, area, keep, cellI);
    exclusive_scan_u32(keep, pos, ncand, pos + ncand, st);
    uint32_t nX = 0;
    readback_sync(&nX, pos + ncand, sizeof(uint32_t), st);
    out->nX = nX;
    out->indices.alloc(2 * (size_t)nX); out->overlaps.alloc(nX);
    if (nX) hipLaunchKernelGGL(k_gg_emit, dim3(ceil_div(ncand, T)), dim3(T), 0, st, iA, candptr, np, ncand, area, keep, pos, cellI,
                               out->indices.p, out->overlaps.p);
    IBH_HIP(hipGetLastError());
    IBH_HIP(hipStreamSynchronize(st));
}

}  // namespace ibh
