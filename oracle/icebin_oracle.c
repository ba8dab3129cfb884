/*
 * icebin_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of IceBin's regrid-matrix assembly and Weighted apply.
 * See icebin_oracle.h for scope and parity status.  Compile with
 * -ffp-contract=off: the reference (gcc, generic x86-64, no -march) performs
 * every multiply and add as a separately rounded IEEE operation, and so does
 * this file.  Canonical floating-point evaluation orders restated here:
 *   - setFromTriplets: duplicates summed in emission order      [Eigen 3.3
 *     SparseMatrix.h set_from_triplets/collapseDuplicates]
 *   - sparse*sparse: res(i,j) = sum over k ASCENDING of lhs(i,k)*rhs(k,j),
 *     the first term assigned, later terms added                [Eigen 3.3
 *     ConservativeSparseSparseProduct.h]
 *   - sparse*diag / diag*sparse: one rounded multiply per stored entry, the
 *     result materialised before it is used in a further product
 *   - spsparse sum(M,dim,'+'|'-'): stored entries visited column-major
 *     (column ascending, row ascending inside), ret += v; '-' then 1/ret
 *   - Weighted_Eigen::apply: SparseMatrix<ColMajor>*Dense, i.e. per variable,
 *     per column j ascending: y[row] += val * x[j]
 * [INFERRED] marks semantics of the absent ibmisc/spsparse sources deduced
 * from IceBin's call sites.
 */
#define _GNU_SOURCE
#include "icebin_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ========================================================================= */
/* small utilities                                                           */
static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n); abort(); }
    return p;
}
static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz ? sz : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
typedef struct { char *buf; int len; int failed; } errctx;
static void set_err(errctx *e, const char *fmt, ...) {
    e->failed = 1;
    if (!e->buf || e->len <= 0) return;
    va_list ap; va_start(ap, fmt); vsnprintf(e->buf, (size_t)e->len, fmt, ap); va_end(ap);
}

/* ========================================================================= */
/* spsparse::SparseSet<long,int> [INFERRED from AbbrGrid.cpp:108,
 * IceCoupler.cpp:298, RegridMatrices_Dynamic.cpp:69-72]: bijection
 * sparse(long) <-> dense(int); add_dense returns the existing id or the next
 * free one, i.e. ids are handed out in first-seen order. */
struct orc_sset {
    long  sparse_extent;
    int   ndense;
    long  cap;
    long *to_sparse;
    long  hcap;      /* power of two */
    long *hk;        /* keys, -1 = empty */
    int  *hv;
};
static unsigned long mix64(long k) {
    unsigned long x = (unsigned long)k;
    x ^= x >> 33; x *= 0xff51afd7ed558ccdUL; x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53UL; x ^= x >> 33;
    return x;
}
orc_sset *orc_sset_new(long sparse_extent) {
    orc_sset *s = (orc_sset *)xcalloc(1, sizeof(*s));
    s->sparse_extent = sparse_extent;
    s->cap = 16; s->to_sparse = (long *)xmalloc(sizeof(long) * (size_t)s->cap);
    s->hcap = 64;
    s->hk = (long *)xmalloc(sizeof(long) * (size_t)s->hcap);
    s->hv = (int *)xmalloc(sizeof(int) * (size_t)s->hcap);
    for (long i = 0; i < s->hcap; ++i) s->hk[i] = -1;
    return s;
}
void orc_sset_free(orc_sset *s) {
    if (!s) return;
    free(s->to_sparse); free(s->hk); free(s->hv); free(s);
}
void orc_sset_set_sparse_extent(orc_sset *s, long n) { s->sparse_extent = n; }
long orc_sset_sparse_extent(const orc_sset *s) { return s->sparse_extent; }
int  orc_sset_dense_extent(const orc_sset *s) { return s->ndense; }
long orc_sset_to_sparse(const orc_sset *s, int d) { return s->to_sparse[d]; }
const long *orc_sset_to_sparse_ptr(const orc_sset *s) { return s->to_sparse; }
static void sset_rehash(orc_sset *s) {
    long ncap = s->hcap * 2;
    long *nk = (long *)xmalloc(sizeof(long) * (size_t)ncap);
    int  *nv = (int *)xmalloc(sizeof(int) * (size_t)ncap);
    for (long i = 0; i < ncap; ++i) nk[i] = -1;
    for (long i = 0; i < s->hcap; ++i) {
        if (s->hk[i] < 0) continue;
        unsigned long h = mix64(s->hk[i]) & (unsigned long)(ncap - 1);
        while (nk[h] >= 0) h = (h + 1) & (unsigned long)(ncap - 1);
        nk[h] = s->hk[i]; nv[h] = s->hv[i];
    }
    free(s->hk); free(s->hv);
    s->hk = nk; s->hv = nv; s->hcap = ncap;
}
int orc_sset_to_dense(const orc_sset *s, long sp) {
    unsigned long h = mix64(sp) & (unsigned long)(s->hcap - 1);
    while (s->hk[h] >= 0) {
        if (s->hk[h] == sp) return s->hv[h];
        h = (h + 1) & (unsigned long)(s->hcap - 1);
    }
    return -1;
}
int orc_sset_add_dense(orc_sset *s, long sp) {
    int d = orc_sset_to_dense(s, sp);
    if (d >= 0) return d;
    if ((long)(s->ndense + 1) * 2 > s->hcap) sset_rehash(s);
    unsigned long h = mix64(sp) & (unsigned long)(s->hcap - 1);
    while (s->hk[h] >= 0) h = (h + 1) & (unsigned long)(s->hcap - 1);
    d = s->ndense++;
    s->hk[h] = sp; s->hv[h] = d;
    if (s->ndense > s->cap) {
        s->cap *= 2;
        s->to_sparse = (long *)xrealloc(s->to_sparse, sizeof(long) * (size_t)s->cap);
    }
    s->to_sparse[d] = sp;
    return d;
}

/* ========================================================================= */
/* triplet accumulator: what a generator writes into (spsparse accum::*)     */
typedef struct { long n, cap; long *i, *j; double *v; } trip;
static void trip_init(trip *t) { memset(t, 0, sizeof(*t)); }
static void trip_free(trip *t) { free(t->i); free(t->j); free(t->v); memset(t, 0, sizeof(*t)); }
static void trip_add(trip *t, long i, long j, double v) {
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->i = (long *)xrealloc(t->i, sizeof(long) * (size_t)t->cap);
        t->j = (long *)xrealloc(t->j, sizeof(long) * (size_t)t->cap);
        t->v = (double *)xrealloc(t->v, sizeof(double) * (size_t)t->cap);
    }
    t->i[t->n] = i; t->j[t->n] = j; t->v[t->n] = v; t->n++;
}

/* dense-indexed triplets in emission order */
typedef struct { long n; int *r, *c; double *v; } dtrip;
static void dtrip_free(dtrip *t) { free(t->r); free(t->c); free(t->v); memset(t, 0, sizeof(*t)); }

/* Eigen::SparseMatrix<double, ColMajor, int> */
typedef struct { int nrow, ncol; long nnz; long *colptr; int *row; double *val; } csc;
static csc *csc_alloc(int nrow, int ncol, long nnz) {
    csc *m = (csc *)xcalloc(1, sizeof(*m));
    m->nrow = nrow; m->ncol = ncol; m->nnz = nnz;
    m->colptr = (long *)xcalloc((size_t)ncol + 1, sizeof(long));
    m->row = (int *)xmalloc(sizeof(int) * (size_t)nnz);
    m->val = (double *)xmalloc(sizeof(double) * (size_t)nnz);
    return m;
}
static void csc_free(csc *m) { if (!m) return; free(m->colptr); free(m->row); free(m->val); free(m); }
static csc *csc_copy(const csc *a) {
    csc *m = csc_alloc(a->nrow, a->ncol, a->nnz);
    memcpy(m->colptr, a->colptr, sizeof(long) * ((size_t)a->ncol + 1));
    memcpy(m->row, a->row, sizeof(int) * (size_t)a->nnz);
    memcpy(m->val, a->val, sizeof(double) * (size_t)a->nnz);
    return m;
}

/* SparsifyTransform (eigen_types.hpp:16-32 via spsparse) [INFERRED] */
enum { TF_ADD_DENSE = 0, TF_TO_DENSE_IGNORE_MISSING = 1 };

/* MakeDenseEigenT(fn, {transform}, dims, transpose, include_zero), phase 1:
 * map every emitted triplet's sparse indices through dims (numbering new
 * indices first-seen for ADD_DENSE; dropping the triplet when an index is
 * missing for TO_DENSE_IGNORE_MISSING), then optionally transpose.
 * `dims[k]` belongs to the generator's k-th index (before transposition),
 * as at RegridMatrices_Dynamic.cpp:75-81.  include_zero==0 drops triplets
 * whose value is exactly 0 before they are numbered [INFERRED].  The
 * reference passes include_zero=false explicitly at :270-277 and leaves the
 * (absent) spsparse default elsewhere; this restatement takes that default to
 * be false as well (INCLUDE_ZERO_DEFAULT): with zeros retained, a zero-area
 * exchange cell under an unmasked ice cell would give EpvG(e,x)=0 times
 * sGvI[x]=1/0 = NaN inside EvI, which real IceBin runs do not show. */
#define INCLUDE_ZERO_DEFAULT 0
static void densify(const trip *t, int transform, orc_sset *dims[2], char transpose,
                    int include_zero, dtrip *out)
{
    out->n = 0;
    out->r = (int *)xmalloc(sizeof(int) * (size_t)t->n);
    out->c = (int *)xmalloc(sizeof(int) * (size_t)t->n);
    out->v = (double *)xmalloc(sizeof(double) * (size_t)t->n);
    for (long k = 0; k < t->n; ++k) {
        if (!include_zero && t->v[k] == 0.0) continue;
        int d0, d1;
        if (transform == TF_ADD_DENSE) {
            d0 = orc_sset_add_dense(dims[0], t->i[k]);
            d1 = orc_sset_add_dense(dims[1], t->j[k]);
        } else {
            d0 = orc_sset_to_dense(dims[0], t->i[k]);
            d1 = orc_sset_to_dense(dims[1], t->j[k]);
            if (d0 < 0 || d1 < 0) continue;
        }
        long n = out->n++;
        if (transpose == 'T') { out->r[n] = d1; out->c[n] = d0; }
        else                  { out->r[n] = d0; out->c[n] = d1; }
        out->v[n] = t->v[k];
    }
}

/* MakeDenseEigenT::to_eigen(), phase 2: Eigen setFromTriplets.  Stable
 * (col,row) ordering by two counting sorts, duplicates summed in emission
 * order (collapseDuplicates keeps the first slot and does value += dup). */
static csc *to_eigen(const dtrip *t, int nrow, int ncol)
{
    long n = t->n;
    long *ord1 = (long *)xmalloc(sizeof(long) * (size_t)n);
    long *ord2 = (long *)xmalloc(sizeof(long) * (size_t)n);
    long *cnt = (long *)xcalloc((size_t)(nrow > ncol ? nrow : ncol) + 2, sizeof(long));
    /* pass 1: stable by row */
    for (long k = 0; k < n; ++k) cnt[t->r[k] + 1]++;
    for (int i = 0; i < nrow; ++i) cnt[i + 1] += cnt[i];
    for (long k = 0; k < n; ++k) ord1[cnt[t->r[k]]++] = k;
    /* pass 2: stable by col */
    memset(cnt, 0, sizeof(long) * ((size_t)(nrow > ncol ? nrow : ncol) + 2));
    for (long k = 0; k < n; ++k) cnt[t->c[k] + 1]++;
    for (int j = 0; j < ncol; ++j) cnt[j + 1] += cnt[j];
    for (long q = 0; q < n; ++q) { long k = ord1[q]; ord2[cnt[t->c[k]]++] = k; }
    /* collapse */
    long nnz = 0;
    for (long q = 0; q < n; ++q) {
        long k = ord2[q];
        if (q == 0 || t->c[ord2[q - 1]] != t->c[k] || t->r[ord2[q - 1]] != t->r[k]) nnz++;
    }
    csc *m = csc_alloc(nrow, ncol, nnz);
    long w = -1;
    for (long q = 0; q < n; ++q) {
        long k = ord2[q];
        if (q == 0 || t->c[ord2[q - 1]] != t->c[k] || t->r[ord2[q - 1]] != t->r[k]) {
            ++w; m->row[w] = t->r[k]; m->val[w] = t->v[k]; m->colptr[t->c[k] + 1]++;
        } else {
            m->val[w] = m->val[w] + t->v[k];
        }
    }
    for (int j = 0; j < ncol; ++j) m->colptr[j + 1] += m->colptr[j];
    free(ord1); free(ord2); free(cnt);
    return m;
}

/* spsparse sum(M, dim, op) (use sites RegridMatrices_Dynamic.cpp:91,100,110,
 * 135,140,...) [INFERRED]: dim==0 -> vector over rows (row sums), dim==1 ->
 * vector over columns; entries visited in Eigen storage order; '-' inverts
 * every element afterwards (1/0 = inf is NOT special-cased). */
static double *sum_csc(const csc *m, int dim, char op)
{
    int n = dim == 0 ? m->nrow : m->ncol;
    double *ret = (double *)xcalloc((size_t)n, sizeof(double));
    for (int j = 0; j < m->ncol; ++j)
        for (long p = m->colptr[j]; p < m->colptr[j + 1]; ++p) {
            int idx = dim == 0 ? m->row[p] : j;
            ret[idx] = ret[idx] + m->val[p];
        }
    if (op == '-') for (int i = 0; i < n; ++i) ret[i] = 1.0 / ret[i];
    return ret;
}
/* invert1 (RegridMatrices_Dynamic.cpp:19-24) */
static double *invert1(const double *a, int n) {
    double *b = (double *)xmalloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) b[i] = 1.0 / a[i];
    return b;
}
/* M * map_eigen_diagonal(d): new matrix, val = M(i,j) * d[j] */
static csc *csc_scale_cols(const csc *a, const double *d) {
    csc *m = csc_copy(a);
    for (int j = 0; j < m->ncol; ++j)
        for (long p = m->colptr[j]; p < m->colptr[j + 1]; ++p) m->val[p] = a->val[p] * d[j];
    return m;
}
/* map_eigen_diagonal(d) * M: new matrix, val = d[i] * M(i,j) */
static csc *csc_scale_rows(const csc *a, const double *d) {
    csc *m = csc_copy(a);
    for (long p = 0; p < m->nnz; ++p) m->val[p] = d[a->row[p]] * a->val[p];
    return m;
}
/* Eigen conservative sparse*sparse product (no pruning): for each result
 * column j, for each stored rhs(k,j) with k ascending, for each stored
 * lhs(i,k): first hit assigns x*y, later hits add x*y.  Result rows sorted. */
static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }
static csc *spgemm(const csc *lhs, const csc *rhs)
{
    int nrow = lhs->nrow, ncol = rhs->ncol;
    double *acc = (double *)xmalloc(sizeof(double) * (size_t)nrow);
    char *mark = (char *)xcalloc((size_t)nrow, 1);
    int *touched = (int *)xmalloc(sizeof(int) * (size_t)nrow);
    long cap = lhs->nnz + rhs->nnz + 16, nnz = 0;
    int *rrow = (int *)xmalloc(sizeof(int) * (size_t)cap);
    double *rval = (double *)xmalloc(sizeof(double) * (size_t)cap);
    long *colptr = (long *)xcalloc((size_t)ncol + 1, sizeof(long));
    for (int j = 0; j < ncol; ++j) {
        int nt = 0;
        for (long q = rhs->colptr[j]; q < rhs->colptr[j + 1]; ++q) {
            int k = rhs->row[q]; double y = rhs->val[q];
            for (long p = lhs->colptr[k]; p < lhs->colptr[k + 1]; ++p) {
                int i = lhs->row[p]; double x = lhs->val[p];
                if (!mark[i]) { mark[i] = 1; acc[i] = x * y; touched[nt++] = i; }
                else acc[i] = acc[i] + x * y;
            }
        }
        qsort(touched, (size_t)nt, sizeof(int), cmp_int);
        if (nnz + nt > cap) {
            cap = (nnz + nt) * 2;
            rrow = (int *)xrealloc(rrow, sizeof(int) * (size_t)cap);
            rval = (double *)xrealloc(rval, sizeof(double) * (size_t)cap);
        }
        for (int t = 0; t < nt; ++t) { int i = touched[t]; rrow[nnz] = i; rval[nnz] = acc[i]; nnz++; mark[i] = 0; }
        colptr[j + 1] = nnz;
    }
    csc *m = (csc *)xcalloc(1, sizeof(*m));
    m->nrow = nrow; m->ncol = ncol; m->nnz = nnz; m->colptr = colptr; m->row = rrow; m->val = rval;
    free(acc); free(mark); free(touched);
    return m;
}

/* ========================================================================= */
/* known-answer helpers                                                      */
static int lower_bound_d(const double *x, int n, double xx) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = lo + (hi - lo) / 2; if (x[mid] < xx) lo = mid + 1; else hi = mid; }
    return lo;
}
/* IceRegridder_L0.cpp:43-67 */
int orc_nearest_1d(const double *xp, int n, double xx) {
    int i1 = lower_bound_d(xp, n, xx);
    if (i1 <= 0) return 0;
    if (i1 >= n) return n - 1;
    int i0 = i1 - 1;
    double d0 = fabs(xx - xp[i0]), d1 = fabs(xp[i1] - xx);
    return d0 <= d1 ? i0 : i1;
}
/* IceRegridder_L0.cpp:71-93: clamps below, errors above the last point */
int orc_linterp_1d_b(const double *xp, int n, double xx, long idx[2], double w[2]) {
    int i1 = lower_bound_d(xp, n, xx);
    if (i1 <= 0) i1 = 1;
    if (i1 >= n) return -1;
    int i0 = i1 - 1;
    idx[0] = i0; idx[1] = i1;
    double ratio = (xx - xp[i0]) / (xp[i1] - xp[i0]);
    w[0] = 1.0 - ratio; w[1] = ratio;
    return 0;
}
/* IceRegridder.cpp:181-204: clamps at both ends */
void orc_linterp_1d(const double *xp, int n, double xx, int idx[2], double w[2]) {
    int i1 = lower_bound_d(xp, n, xx);
    if (i1 <= 0) i1 = 1;
    if (i1 >= n) i1 = n - 1;
    int i0 = i1 - 1;
    idx[0] = i0; idx[1] = i1;
    double ratio = (xx - xp[i0]) / (xp[i1] - xp[i0]);
    w[0] = 1.0 - ratio; w[1] = ratio;
}
/* ElevMask.cpp:57-70; PISM mask codes from the fixture's flag_values
 * (elev_mask.cdl: 0 bedrock, 2 grounded, 3 floating, 4 ocean); UNKNOWN -> NaN */
void orc_elevmask_pism(long n, const signed char *mask, const double *thk,
                       const double *topg, double *emI_land, double *emI_ice) {
    for (long i = 0; i < n; ++i) {
        switch (mask[i]) {
            case 2: case 3: emI_land[i] = emI_ice[i] = topg[i] + thk[i]; break;
            case 0: emI_land[i] = topg[i]; emI_ice[i] = NAN; break;
            default: emI_land[i] = emI_ice[i] = NAN; break;
        }
    }
}
/* pylib/icebin_cython.cpp:158-192 */
void orc_coo_matvec(double *yy, const double *xx, int ignore_nan, long nrow, long ncol,
                    long nnz, const int *row, const int *col, const double *data) {
    (void)ncol;
    char *written = (char *)xcalloc((size_t)nrow, 1);
    for (long n = 0; n < nnz; ++n) {
        if (ignore_nan && isnan(xx[col[n]])) continue;
        double old = written[row[n]] ? yy[row[n]] : 0.0;
        written[row[n]] = 1;
        yy[row[n]] = old + data[n] * xx[col[n]];
    }
    free(written);
}

/* ========================================================================= */
/* Ur-matrix generators (COO emitters), gridG == 'X' forms                    */

/* IceRegridder_L0::GvAp, IceRegridder_L0.cpp:192-214 */
static void gen_GvAp(const orc_regridder *rg, const double *em, trip *out, errctx *e) {
    (void)e;
    for (long id = 0; id < rg->nX; ++id) {
        long iA = rg->ex_indices[2 * id], iI = rg->ex_indices[2 * id + 1];
        if (!isnan(em[iI]) && rg->ex_area[id] > 0) trip_add(out, id, iA, rg->ex_area[id]);
    }
}
/* IceRegridder_L0::GvI, 'X' branch, IceRegridder_L0.cpp:176-188 */
static void gen_GvI(const orc_regridder *rg, const double *em, trip *out) {
    for (long id = 0; id < rg->nX; ++id) {
        long iI = rg->ex_indices[2 * id + 1];
        if (!isnan(em[iI])) trip_add(out, id, iI, rg->ex_area[id]);
    }
}
/* IceRegridder_L0::GvEp, IceRegridder_L0.cpp:100-155 */
static void gen_GvEp(const orc_regridder *rg, const double *em, trip *out, errctx *e) {
    if (rg->nhc == 0) { set_err(e, "IceRegridder_L0::GvEp(): hcdefs is zero-length!"); return; }
    for (long id = 0; id < rg->nX; ++id) {
        long iA = rg->ex_indices[2 * id], iI = rg->ex_indices[2 * id + 1];
        if (isnan(em[iI])) continue;
        double elevation = em[iI] < 0.0 ? 0.0 : em[iI];   /* std::max(elev, 0.0), :123 */
        if (rg->interp_style == ORC_Z_INTERP) {
            long ih[2]; double wh[2];
            if (orc_linterp_1d_b(rg->hcdefs, rg->nhc, elevation, ih, wh) != 0) {
                set_err(e, "Elevation %g out of bounds (%g, %g)", elevation,
                        rg->hcdefs[0], rg->hcdefs[rg->nhc - 1]);
                return;
            }
            if (wh[0] != 0) trip_add(out, id, iA * rg->hc_stride_A + ih[0] * rg->hc_stride_HC, rg->ex_area[id] * wh[0]);
            if (wh[1] != 0) trip_add(out, id, iA * rg->hc_stride_A + ih[1] * rg->hc_stride_HC, rg->ex_area[id] * wh[1]);
        } else {
            int ih0 = orc_nearest_1d(rg->hcdefs, rg->nhc, elevation);
            trip_add(out, id, iA * rg->hc_stride_A + ih0 * rg->hc_stride_HC, rg->ex_area[id]);
        }
    }
}
/* IceRegridder::sApvA, IceRegridder.cpp:41-48 */
static void gen_sApvA(const orc_regridder *rg, trip *out) {
    for (int id = 0; id < rg->nA_dense; ++id) {
        long index = rg->A_to_sparse[id];
        trip_add(out, index, index, rg->A_native_area[id] / rg->A_proj_area[id]);
    }
}
/* IceRegridder::sEpvE, IceRegridder.cpp:52-65 */
static void gen_sEpvE(const orc_regridder *rg, trip *out) {
    for (int id = 0; id < rg->nA_dense; ++id) {
        long index = rg->A_to_sparse[id];
        for (long ihp = 0; ihp < rg->nhc; ++ihp) {
            long indexE = index * rg->hc_stride_A + ihp * rg->hc_stride_HC;
            trip_add(out, indexE, indexE, rg->A_native_area[id] / rg->A_proj_area[id]);
        }
    }
}

/* UrAE (RegridMatrices_Dynamic.cpp:32-45): 'A' or 'E' flavour of the Ur pair */
typedef struct { char which; long nfull; } urae;
static void ur_GvAp(const urae *u, const orc_regridder *rg, const double *em, trip *t, errctx *e) {
    if (u->which == 'A') gen_GvAp(rg, em, t, e); else gen_GvEp(rg, em, t, e);
}
static void ur_sApvA(const urae *u, const orc_regridder *rg, trip *t) {
    if (u->which == 'A') gen_sApvA(rg, t); else gen_sEpvE(rg, t);
}

/* ========================================================================= */
struct orc_weighted {
    orc_sset *dims[2];
    int       owns[2];
    csc      *M;
    double   *wM, *Mw;
    int       conservative, scaled;
    /* row-major copy for the threaded baseline, built lazily */
    long     *rowptr; int *colind; double *rval;
};
static orc_weighted *weighted_new(orc_sset *d0, orc_sset *d1, int conservative) {
    orc_weighted *w = (orc_weighted *)xcalloc(1, sizeof(*w));
    w->dims[0] = d0 ? d0 : orc_sset_new(-1); w->owns[0] = d0 == NULL;
    w->dims[1] = d1 ? d1 : orc_sset_new(-1); w->owns[1] = d1 == NULL;
    w->conservative = conservative;
    return w;
}
void orc_weighted_free(orc_weighted *w) {
    if (!w) return;
    for (int k = 0; k < 2; ++k) if (w->owns[k]) orc_sset_free(w->dims[k]);
    csc_free(w->M); free(w->wM); free(w->Mw);
    free(w->rowptr); free(w->colind); free(w->rval);
    free(w);
}
int  orc_weighted_nrow(const orc_weighted *w) { return w->M->nrow; }
int  orc_weighted_ncol(const orc_weighted *w) { return w->M->ncol; }
long orc_weighted_nnz(const orc_weighted *w) { return w->M->nnz; }
int  orc_weighted_conservative(const orc_weighted *w) { return w->conservative; }
int  orc_weighted_scaled(const orc_weighted *w) { return w->scaled; }
const orc_sset *orc_weighted_dim(const orc_weighted *w, int k) { return w->dims[k]; }
const double *orc_weighted_wM(const orc_weighted *w) { return w->wM; }
const double *orc_weighted_Mw(const orc_weighted *w) { return w->Mw; }

static void build_csr(orc_weighted *w) {
    if (w->rowptr) return;
    const csc *m = w->M;
    w->rowptr = (long *)xcalloc((size_t)m->nrow + 1, sizeof(long));
    w->colind = (int *)xmalloc(sizeof(int) * (size_t)m->nnz);
    w->rval = (double *)xmalloc(sizeof(double) * (size_t)m->nnz);
    for (long p = 0; p < m->nnz; ++p) w->rowptr[m->row[p] + 1]++;
    for (int i = 0; i < m->nrow; ++i) w->rowptr[i + 1] += w->rowptr[i];
    long *pos = (long *)xmalloc(sizeof(long) * ((size_t)m->nrow + 1));
    memcpy(pos, w->rowptr, sizeof(long) * ((size_t)m->nrow + 1));
    for (int j = 0; j < m->ncol; ++j)
        for (long p = m->colptr[j]; p < m->colptr[j + 1]; ++p) {
            long q = pos[m->row[p]]++;
            w->colind[q] = j; w->rval[q] = m->val[p];
        }
    free(pos);
}
void orc_weighted_coo(const orc_weighted *cw, int *row, int *col, double *val) {
    orc_weighted *w = (orc_weighted *)cw;
    build_csr(w);
    for (int i = 0; i < w->M->nrow; ++i)
        for (long q = w->rowptr[i]; q < w->rowptr[i + 1]; ++q) { row[q] = i; col[q] = w->colind[q]; val[q] = w->rval[q]; }
}
orc_weighted *orc_weighted_from_coo(int nrow, int ncol, long nnz, const int *row,
                                    const int *col, const double *val,
                                    const double *wM, const double *Mw,
                                    int conservative, int scaled) {
    orc_weighted *w = weighted_new(NULL, NULL, conservative);
    /* identity dims: dense == sparse */
    orc_sset_set_sparse_extent(w->dims[0], nrow);
    orc_sset_set_sparse_extent(w->dims[1], ncol);
    for (int i = 0; i < nrow; ++i) orc_sset_add_dense(w->dims[0], i);
    for (int j = 0; j < ncol; ++j) orc_sset_add_dense(w->dims[1], j);
    dtrip t; t.n = nnz;
    t.r = (int *)xmalloc(sizeof(int) * (size_t)nnz); memcpy(t.r, row, sizeof(int) * (size_t)nnz);
    t.c = (int *)xmalloc(sizeof(int) * (size_t)nnz); memcpy(t.c, col, sizeof(int) * (size_t)nnz);
    t.v = (double *)xmalloc(sizeof(double) * (size_t)nnz); memcpy(t.v, val, sizeof(double) * (size_t)nnz);
    w->M = to_eigen(&t, nrow, ncol);
    dtrip_free(&t);
    w->wM = (double *)xmalloc(sizeof(double) * (size_t)nrow); memcpy(w->wM, wM, sizeof(double) * (size_t)nrow);
    w->Mw = (double *)xmalloc(sizeof(double) * (size_t)ncol); memcpy(w->Mw, Mw, sizeof(double) * (size_t)ncol);
    w->scaled = scaled;
    return w;
}

/* diagonal Ur matrix densified with TO_DENSE_IGNORE_MISSING over {dim,dim}
 * (RegridMatrices_Dynamic.cpp:105-108, 206-209, 292-300) */
static csc *make_diag(const urae *u, const orc_regridder *rg, orc_sset *dim) {
    trip t; trip_init(&t);
    ur_sApvA(u, rg, &t);
    orc_sset *dd[2] = { dim, dim };
    dtrip d; densify(&t, TF_TO_DENSE_IGNORE_MISSING, dd, '.', INCLUDE_ZERO_DEFAULT, &d);
    csc *m = to_eigen(&d, orc_sset_dense_extent(dim), orc_sset_dense_extent(dim));
    dtrip_free(&d); trip_free(&t);
    return m;
}

static csc *smoothing_matrix(const orc_regridder *rg, const orc_sset *dimI, const double *em,
                             const double *area_d, const double sigma[3], errctx *e);

/* compute_AEvI, RegridMatrices_Dynamic.cpp:50-151 -- AvI, EvI, AvX, EvX */
static orc_weighted *compute_AEvI(const orc_regridder *rg, orc_sset *dims[2], int scale,
                                  int correctA, const double *em, char Igrid,
                                  const urae *AE, errctx *e)
{
    orc_weighted *ret = weighted_new(dims[0], dims[1], 1);            /* :63 */
    orc_sset *dimA = ret->dims[0], *dimI = ret->dims[1];
    orc_sset *dimG = orc_sset_new(-1);
    orc_sset_set_sparse_extent(dimA, AE->nfull);                      /* :69 */
    orc_sset_set_sparse_extent(dimI, Igrid == 'I' ? rg->nI : rg->nX); /* :70-71 */
    orc_sset_set_sparse_extent(dimG, rg->nX);                         /* :72 */

    /* Ur matrices: densify in the reference's order (GvAp first, then GvI),
     * build the Eigen matrices once both have numbered dimG so their shapes
     * agree (the reference relies on both generators seeing the same X set). */
    trip tA; trip_init(&tA); ur_GvAp(AE, rg, em, &tA, e);
    dtrip dA; memset(&dA, 0, sizeof(dA));
    dtrip dI; memset(&dI, 0, sizeof(dI));
    csc *ApvI = NULL;
    if (!e->failed) {
        orc_sset *dd[2] = { Igrid == 'I' ? dimG : dimI, dimA };
        densify(&tA, TF_ADD_DENSE, dd, 'T', INCLUDE_ZERO_DEFAULT, &dA);                  /* :75-81 */
        if (Igrid == 'I') {
            trip tI; trip_init(&tI); gen_GvI(rg, em, &tI);
            orc_sset *di[2] = { dimG, dimI };
            densify(&tI, TF_ADD_DENSE, di, '.', INCLUDE_ZERO_DEFAULT, &dI);              /* :86-90 */
            trip_free(&tI);
            csc *ApvG = to_eigen(&dA, orc_sset_dense_extent(dimA), orc_sset_dense_extent(dimG));
            csc *GvI = to_eigen(&dI, orc_sset_dense_extent(dimG), orc_sset_dense_extent(dimI));
            double *sGvI = sum_csc(GvI, 0, '-');                      /* :91 */
            csc *lhs = csc_scale_cols(ApvG, sGvI);                    /* ApvG * diag(sGvI) */
            ApvI = spgemm(lhs, GvI);                                  /* :93-94 */
            csc_free(lhs); free(sGvI); csc_free(ApvG); csc_free(GvI);
        } else {
            ApvI = to_eigen(&dA, orc_sset_dense_extent(dimA), orc_sset_dense_extent(dimI)); /* :97 */
        }
    }
    trip_free(&tA); dtrip_free(&dA); dtrip_free(&dI);
    orc_sset_free(dimG);
    if (e->failed) { orc_weighted_free(ret); return NULL; }

    ret->Mw = sum_csc(ApvI, 1, '+');                                  /* :100 */
    if (correctA) {
        csc *wAvAp = make_diag(AE, rg, dimA);                         /* :105-108 */
        double *wApvI = sum_csc(ApvI, 0, '+');                        /* :110 */
        csc *wAvI = csc_scale_cols(wAvAp, wApvI);                     /* :112 */
        ret->wM = sum_csc(wAvI, 0, '+');                              /* :115 */
        double *sAvAp = sum_csc(wAvAp, 0, '-');                       /* :118 */
        if (scale) {
            int n = ApvI->nrow;
            double *sApvI = invert1(wApvI, n);                        /* :121 */
            double *mul = (double *)xmalloc(sizeof(double) * (size_t)n);
            for (int i = 0; i < n; ++i) mul[i] = sAvAp[i] * sApvI[i]; /* :122 */
            ret->M = csc_scale_rows(ApvI, mul);                       /* :123-124 */
            csc_free(ApvI); free(sApvI); free(mul);
        } else {
            ret->M = ApvI;                                            /* :128 */
        }
        csc_free(wAvAp); csc_free(wAvI); free(wApvI); free(sAvAp);
    } else {
        ret->wM = sum_csc(ApvI, 0, '+');                              /* :135-136 */
        if (scale) {
            double *sApvI = sum_csc(ApvI, 0, '-');                    /* :140 */
            ret->M = csc_scale_rows(ApvI, sApvI);                     /* :142-143 */
            csc_free(ApvI); free(sApvI);
        } else {
            ret->M = ApvI;                                            /* :145 */
        }
    }
    return ret;
}

/* compute_IvAE, RegridMatrices_Dynamic.cpp:154-252 -- IvA, IvE, XvA, XvE */
static orc_weighted *compute_IvAE(const orc_regridder *rg, orc_sset *dims[2], int scale,
                                  int correctA, const double *sigma, const double *em, char Igrid,
                                  const urae *AE, errctx *e)
{
    const int smooth = sigma && sigma[0] != 0;              /* RegridParams::smooth(), RegridMatrices.hpp:31 */
    orc_weighted *ret = weighted_new(dims[0], dims[1], !smooth);  /* conservative = !smooth, :167 */
    orc_sset *dimA = ret->dims[1], *dimI = ret->dims[0];              /* :168 */
    orc_sset *dimG = orc_sset_new(-1);
    orc_sset_set_sparse_extent(dimA, AE->nfull);
    orc_sset_set_sparse_extent(dimI, Igrid == 'I' ? rg->nI : rg->nX);
    orc_sset_set_sparse_extent(dimG, rg->nX);

    trip tA; trip_init(&tA); ur_GvAp(AE, rg, em, &tA, e);
    dtrip dA; memset(&dA, 0, sizeof(dA));
    dtrip dI; memset(&dI, 0, sizeof(dI));
    csc *IvAp = NULL;
    if (!e->failed) {
        orc_sset *dd[2] = { Igrid == 'I' ? dimG : dimI, dimA };
        densify(&tA, TF_ADD_DENSE, dd, '.', INCLUDE_ZERO_DEFAULT, &dA);                  /* :178-183 */
        if (Igrid == 'I') {
            trip tI; trip_init(&tI); gen_GvI(rg, em, &tI);
            orc_sset *di[2] = { dimG, dimI };
            densify(&tI, TF_ADD_DENSE, di, 'T', INCLUDE_ZERO_DEFAULT, &dI);              /* :187-190 */
            trip_free(&tI);
            csc *GvAp = to_eigen(&dA, orc_sset_dense_extent(dimG), orc_sset_dense_extent(dimA));
            csc *IvG = to_eigen(&dI, orc_sset_dense_extent(dimI), orc_sset_dense_extent(dimG));
            double *sGvAp = sum_csc(GvAp, 0, '-');                    /* :192 */
            csc *lhs = csc_scale_cols(IvG, sGvAp);
            IvAp = spgemm(lhs, GvAp);                                 /* :193-194 */
            csc_free(lhs); free(sGvAp); csc_free(GvAp); csc_free(IvG);
        } else {
            IvAp = to_eigen(&dA, orc_sset_dense_extent(dimI), orc_sset_dense_extent(dimA)); /* :196 */
        }
    }
    trip_free(&tA); dtrip_free(&dA); dtrip_free(&dI);
    orc_sset_free(dimG);
    if (e->failed) { orc_weighted_free(ret); return NULL; }

    ret->wM = sum_csc(IvAp, 0, '+');                                  /* :201 */
    if (correctA) {
        csc *sApvA = make_diag(AE, rg, dimA);                         /* :206-209 */
        double *IvApw = sum_csc(IvAp, 1, '+');                        /* :212 */
        csc *Aw = csc_scale_cols(sApvA, IvApw);                       /* :214 */
        ret->Mw = sum_csc(Aw, 0, '+');                                /* :215 */
        if (scale) {
            double *sIvAp = sum_csc(IvAp, 0, '-');                    /* :218 */
            csc *t1 = csc_scale_rows(IvAp, sIvAp);
            ret->M = spgemm(t1, sApvA);                               /* :219-220 */
            csc_free(t1); free(sIvAp);
        } else {
            ret->M = spgemm(IvAp, sApvA);                             /* :222-223 */
        }
        csc_free(IvAp); csc_free(sApvA); csc_free(Aw); free(IvApw);
    } else {
        ret->Mw = sum_csc(IvAp, 1, '+');                              /* :226 */
        if (scale) {
            double *sIvAp = sum_csc(IvAp, 0, '-');                    /* :228 */
            ret->M = csc_scale_rows(IvAp, sIvAp);                     /* :229-230 */
            csc_free(IvAp); free(sIvAp);
        } else {
            ret->M = IvAp;                                            /* :232 */
        }
    }
    if (smooth) {                                                     /* :237-248 */
        csc *smoothI = smoothing_matrix(rg, dimI, em, ret->wM, sigma, e);
        if (!smoothI) { orc_weighted_free(ret); return NULL; }
        csc *Ms = spgemm(smoothI, ret->M);                            /* smoothI * M */
        csc_free(smoothI); csc_free(ret->M);
        ret->M = Ms;
    }
    return ret;
}

/* smoothing_matrix + Smoother::matrix, smoother.cpp:43-99 (hook RegridMatrices_Dynamic.cpp:237-248).
 * Tuples: the ice cells of agridI (all cells of the rectilinear sheet, ascending sparse index) that are
 * in dimI and unmasked, with centroid (x, y, elevation) and area = wM[dense id].  Row of cell t0:
 * every tuple t with sum_k ((t.c[k]-t0.c[k])/sigma[k])^2 < nsigma^2 = 4 gets w = exp(-.5*d2) * t.area,
 * the row is scaled by 1/sum(w).  The reference finds the neighbours with an RTree (ibmisc, absent) and
 * sums the denominator in the RTree's visiting order; here neighbours are visited in ascending tuple
 * order [PARITY UNPINNED at rounding level].  Brute-force search: the oracle is for test sizes. */
static csc *smoothing_matrix(const orc_regridder *rg, const orc_sset *dimI, const double *em,
                             const double *area_d, const double sigma[3], errctx *e)
{
    int nd = orc_sset_dense_extent(dimI);
    int nt = 0;
    int *tid = (int *)xmalloc(sizeof(int) * (size_t)(nd ? nd : 1));
    double *cx = (double *)xmalloc(sizeof(double) * 3 * (size_t)(nd ? nd : 1));
    if (!rg->I_centroid_xy) { set_err(e, "smoothing needs the ice grid's centroid_xy"); free(tid); free(cx); return NULL; }
    for (long iXs = 0; iXs < rg->nI; ++iXs) {               /* agridX.dim: dense == sparse for the sheet grid */
        int d = orc_sset_to_dense(dimI, iXs);
        if (d < 0) continue;
        double elev = em[iXs];
        if (isnan(elev)) continue;
        if (area_d[d] == 0.) { set_err(e, "Area of cell %ld must be non-zero\n", iXs); free(tid); free(cx); return NULL; }
        tid[nt] = d;
        cx[3 * nt] = rg->I_centroid_xy[2 * iXs]; cx[3 * nt + 1] = rg->I_centroid_xy[2 * iXs + 1]; cx[3 * nt + 2] = elev;
        nt++;
    }
    dtrip t; t.n = 0;
    long cap = 1024;
    t.r = (int *)xmalloc(sizeof(int) * (size_t)cap); t.c = (int *)xmalloc(sizeof(int) * (size_t)cap);
    t.v = (double *)xmalloc(sizeof(double) * (size_t)cap);
    const double nsigma_squared = 2. * 2.;
    for (int i = 0; i < nt; ++i) {
        long start = t.n;
        double denom_sum = 0;
        for (int j = 0; j < nt; ++j) {
            double nds = 0;
            for (int k = 0; k < 3; ++k) {
                double d = (cx[3 * j + k] - cx[3 * i + k]) / sigma[k];
                nds = nds + d * d;
            }
            if (nds < nsigma_squared) {
                double w = exp(-.5 * nds) * area_d[tid[j]];
                if (t.n == cap) {
                    cap *= 2;
                    t.r = (int *)xrealloc(t.r, sizeof(int) * (size_t)cap); t.c = (int *)xrealloc(t.c, sizeof(int) * (size_t)cap);
                    t.v = (double *)xrealloc(t.v, sizeof(double) * (size_t)cap);
                }
                t.r[t.n] = tid[i]; t.c[t.n] = tid[j]; t.v[t.n] = w; t.n++;
                denom_sum = denom_sum + w;
            }
        }
        double factor = 1. / denom_sum;
        for (long q = start; q < t.n; ++q) t.v[q] = factor * t.v[q];
    }
    csc *m = to_eigen(&t, nd, nd);
    dtrip_free(&t); free(tid); free(cx);
    return m;
}

/* compute_EvA, RegridMatrices_Dynamic.cpp:254-332 -- EvA, and AvE with the
 * roles of the two UrAE swapped (:387-390) */
static orc_weighted *compute_EvA(const orc_regridder *rg, orc_sset *dims[2], int scale,
                                 int correctA, const double *em,
                                 const urae *E, const urae *A, errctx *e)
{
    orc_weighted *ret = weighted_new(dims[0], dims[1], 1);            /* :258 */
    orc_sset *dimE = ret->dims[0], *dimA = ret->dims[1];
    orc_sset *dimG = orc_sset_new(-1);
    orc_sset_set_sparse_extent(dimA, A->nfull);
    orc_sset_set_sparse_extent(dimE, E->nfull);
    orc_sset_set_sparse_extent(dimG, rg->nX);

    trip tA; trip_init(&tA); ur_GvAp(A, rg, em, &tA, e);
    trip tE; trip_init(&tE); if (!e->failed) ur_GvAp(E, rg, em, &tE, e);
    if (e->failed) { trip_free(&tA); trip_free(&tE); orc_sset_free(dimG); orc_weighted_free(ret); return NULL; }
    dtrip dA, dE;
    orc_sset *da[2] = { dimG, dimA }; densify(&tA, TF_ADD_DENSE, da, '.', 0, &dA);  /* :270-273 */
    orc_sset *de[2] = { dimG, dimE }; densify(&tE, TF_ADD_DENSE, de, 'T', 0, &dE);  /* :274-277 */
    csc *GvAp = to_eigen(&dA, orc_sset_dense_extent(dimG), orc_sset_dense_extent(dimA)); /* :280 */
    csc *EpvG = to_eigen(&dE, orc_sset_dense_extent(dimE), orc_sset_dense_extent(dimG)); /* :281 */
    trip_free(&tA); trip_free(&tE); dtrip_free(&dA); dtrip_free(&dE);
    orc_sset_free(dimG);

    double *sGvAp = sum_csc(GvAp, 0, '-');                            /* :283 */
    csc *lhs = csc_scale_cols(EpvG, sGvAp);
    csc *EpvAp = spgemm(lhs, GvAp);                                   /* :286-287 */
    csc_free(lhs); free(sGvAp); csc_free(GvAp); csc_free(EpvG);

    double *wEpvAp = sum_csc(EpvAp, 0, '+');                          /* :290 */
    if (correctA) {
        csc *sApvA = make_diag(A, rg, dimA);                          /* :292-295 */
        csc *wEvEp = make_diag(E, rg, dimE);                          /* :297-300 */
        csc *wEvAp = csc_scale_cols(wEvEp, wEpvAp);                   /* :303 */
        ret->wM = sum_csc(wEvAp, 0, '+');                             /* :304 */
        double *EpvApw = sum_csc(EpvAp, 1, '+');                      /* :307 */
        csc *Aw = csc_scale_cols(sApvA, EpvApw);                      /* :309 */
        ret->Mw = sum_csc(Aw, 0, '+');                                /* :310 */
        if (scale) {
            double *sEvAp = sum_csc(wEvAp, 0, '-');                   /* :313 */
            csc *t1 = csc_scale_rows(EpvAp, sEvAp);
            ret->M = spgemm(t1, sApvA);                               /* :314-315 */
            csc_free(t1); free(sEvAp);
        } else {
            ret->M = spgemm(EpvAp, sApvA);                            /* :317 */
        }
        csc_free(EpvAp); csc_free(sApvA); csc_free(wEvEp); csc_free(wEvAp); csc_free(Aw);
        free(EpvApw); free(wEpvAp);
    } else {
        ret->wM = wEpvAp;                                             /* :321 */
        ret->Mw = sum_csc(EpvAp, 1, '+');                             /* :322 */
        if (scale) {
            double *sEpvAp = invert1(wEpvAp, EpvAp->nrow);            /* :324 */
            ret->M = csc_scale_rows(EpvAp, sEpvAp);                   /* :325 */
            csc_free(EpvAp); free(sEpvAp);
        } else {
            ret->M = EpvAp;                                           /* :327 */
        }
    }
    return ret;
}

/* GCMRegridder_Standard::regrid_matrices name table (RegridMatrices_Dynamic.cpp:
 * 354-390) + RegridMatrices_Dynamic::matrix_d (:412-423) */
int orc_matrix_d(const orc_regridder *rg, const double *elevmaskI,
                 const char *spec, orc_sset *dim0, orc_sset *dim1,
                 int scale, int correctA, const double sigma[3],
                 orc_weighted **out, char *err, int errlen)
{
    errctx e = { err, errlen, 0 };
    *out = NULL;
    urae urA = { 'A', rg->nA };                                       /* :354-356 */
    urae urE = { 'E', rg->nA * (long)rg->nhc };                       /* :358-360; nE = nA*nhc, GCMRegridder.hpp:273 */
    orc_sset *dims[2] = { dim0, dim1 };
    orc_weighted *w = NULL;
    if      (!strcmp(spec, "AvI")) w = compute_AEvI(rg, dims, scale, correctA, elevmaskI, 'I', &urA, &e);
    else if (!strcmp(spec, "IvA")) w = compute_IvAE(rg, dims, scale, correctA, sigma, elevmaskI, 'I', &urA, &e);
    else if (!strcmp(spec, "AvX")) w = compute_AEvI(rg, dims, scale, correctA, elevmaskI, 'X', &urA, &e);
    else if (!strcmp(spec, "XvA")) w = compute_IvAE(rg, dims, scale, correctA, sigma, elevmaskI, 'X', &urA, &e);
    else if (!strcmp(spec, "EvI")) w = compute_AEvI(rg, dims, scale, correctA, elevmaskI, 'I', &urE, &e);
    else if (!strcmp(spec, "IvE")) w = compute_IvAE(rg, dims, scale, correctA, sigma, elevmaskI, 'I', &urE, &e);
    else if (!strcmp(spec, "EvX")) w = compute_AEvI(rg, dims, scale, correctA, elevmaskI, 'X', &urE, &e);
    else if (!strcmp(spec, "XvE")) w = compute_IvAE(rg, dims, scale, correctA, sigma, elevmaskI, 'X', &urE, &e);
    else if (!strcmp(spec, "EvA")) w = compute_EvA(rg, dims, scale, correctA, elevmaskI, &urE, &urA, &e);
    else if (!strcmp(spec, "AvE")) w = compute_EvA(rg, dims, scale, correctA, elevmaskI, &urA, &urE, &e);
    else { set_err(&e, "unknown regrid matrix '%s'", spec); return -1; }  /* regrids.at() throws, :419 */
    if (!w) return -1;
    w->scaled = scale;                                                /* :421 */
    *out = w;
    return 0;
}

/* ========================================================================= */
/* Weighted_Eigen::apply [INFERRED; SURVEY.md 8a row 14, matrix_formats.rst:
 * 156-165, dead helper mask_result IceCoupler.cpp:186-201]:
 *   B0 = M * A  (Eigen ColMajor sparse * dense: column scatter);
 *   rows with wM == 0 receive `fill`;
 *   if !conservative && force_conservation: every variable is multiplied by
 *   (Mw . A) / (wM . B0), the dot products skipping rows/cols of zero weight.
 * The fill and force_conservation branches are PARITY UNPINNED. */
void orc_apply(const orc_weighted *w, const double *A_b, int nvar, double fill,
               int force_conservation, double *B_b)
{
    const csc *m = w->M;
    int nrow = m->nrow, ncol = m->ncol;
    for (int k = 0; k < nvar; ++k) {
        const double *x = A_b + (size_t)k * (size_t)ncol;
        double *y = B_b + (size_t)k * (size_t)nrow;
        for (int i = 0; i < nrow; ++i) y[i] = 0.0;
        for (int j = 0; j < ncol; ++j) {
            double xj = x[j];
            for (long p = m->colptr[j]; p < m->colptr[j + 1]; ++p)
                y[m->row[p]] = y[m->row[p]] + m->val[p] * xj;
        }
        if (!w->conservative && force_conservation) {
            double TA = 0.0, TB = 0.0;
            for (int j = 0; j < ncol; ++j) if (w->Mw[j] != 0.0) TA = TA + w->Mw[j] * x[j];
            for (int i = 0; i < nrow; ++i) if (w->wM[i] != 0.0) TB = TB + w->wM[i] * y[i];
            double factor = TA / TB;
            for (int i = 0; i < nrow; ++i) y[i] = y[i] * factor;
        }
        for (int i = 0; i < nrow; ++i) if (w->wM[i] == 0.0) y[i] = fill;
    }
}
void orc_apply_transformed(const orc_weighted *w, const double *V_b, int nvar_in, const double *T,
                           const double *b, int nvar_out, double fill, double *B_b)
{
    int ncol = w->M->ncol;
    double *X = (double *)xmalloc(sizeof(double) * (size_t)nvar_out * (size_t)ncol);
    for (int k = 0; k < nvar_out; ++k) {
        double *x = X + (size_t)k * (size_t)ncol;
        for (int j = 0; j < ncol; ++j) x[j] = 0.0;
        for (int l = 0; l < nvar_in; ++l) {                 /* gcm_ovalsE0_e * T.M: stored entries only */
            double t = T[(size_t)l * (size_t)nvar_out + k];
            if (t == 0.0) continue;
            const double *v = V_b + (size_t)l * (size_t)ncol;
            for (int j = 0; j < ncol; ++j) x[j] = x[j] + t * v[j];
        }
        for (int j = 0; j < ncol; ++j) x[j] = x[j] + b[k];  /* + b.replicate(n,1) */
    }
    orc_apply(w, X, nvar_out, fill, 0, B_b);
    free(X);
}
void orc_apply_weight(const orc_weighted *w, int dim, const double *A_b, int nvar, double *out) {
    int n = dim == 0 ? w->M->nrow : w->M->ncol;
    const double *wt = dim == 0 ? w->wM : w->Mw;
    for (int k = 0; k < nvar; ++k) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s = s + wt[j] * A_b[(size_t)k * (size_t)n + j];
        out[k] = s;
    }
}

/* ---- timed CPU baseline legs (bench.py cpu_baseline) ---------------------- */
typedef struct { const orc_weighted *w; const double *A; double *B; int nvar; int r0, r1; } tjob;
static void *csr_rows_worker(void *p) {
    tjob *j = (tjob *)p; const orc_weighted *w = j->w;
    int nrow = w->M->nrow, ncol = w->M->ncol;
    for (int k = 0; k < j->nvar; ++k) {
        const double *x = j->A + (size_t)k * (size_t)ncol;
        double *y = j->B + (size_t)k * (size_t)nrow;
        for (int i = j->r0; i < j->r1; ++i) {
            double s = 0.0;
            for (long q = w->rowptr[i]; q < w->rowptr[i + 1]; ++q) s += w->rval[q] * x[w->colind[q]];
            y[i] = s;
        }
    }
    return NULL;
}
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
double orc_time_apply(const orc_weighted *cw, const double *A_b, int nvar, double *B_b, int reps, int threads)
{
    orc_weighted *w = (orc_weighted *)cw;
    if (threads <= 1) {
        double t0 = now_s();
        for (int r = 0; r < reps; ++r) orc_apply(w, A_b, nvar, NAN, 0, B_b);
        return (now_s() - t0) / reps;
    }
    build_csr(w);
    /* balance rows by nnz */
    int nrow = w->M->nrow;
    int *cut = (int *)xmalloc(sizeof(int) * ((size_t)threads + 1));
    cut[0] = 0;
    for (int t = 1; t < threads; ++t) {
        long target = w->M->nnz * t / threads; int lo = cut[t - 1], hi = nrow;
        while (lo < hi) { int mid = lo + (hi - lo) / 2; if (w->rowptr[mid] < target) lo = mid + 1; else hi = mid; }
        cut[t] = lo;
    }
    cut[threads] = nrow;
    pthread_t *th = (pthread_t *)xmalloc(sizeof(pthread_t) * (size_t)threads);
    tjob *jobs = (tjob *)xmalloc(sizeof(tjob) * (size_t)threads);
    double t0 = now_s();
    for (int r = 0; r < reps; ++r) {
        for (int t = 0; t < threads; ++t) {
            jobs[t] = (tjob){ w, A_b, B_b, nvar, cut[t], cut[t + 1] };
            pthread_create(&th[t], NULL, csr_rows_worker, &jobs[t]);
        }
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    }
    double dt = (now_s() - t0) / reps;
    free(cut); free(th); free(jobs);
    return dt;
}

/* ========================================================================= */
/* E1vE0 (slib/icebin/e1ve0.cpp:55-106 compute_E1vE0c): the matrix that carries fields on the
 * elevation grid of the previous coupling step (E0) to that of the current one (E1),
 *     E1vE0c = diag(1 / sum_sheets Mw(XuE1)) * sum_sheets[ E1uX * (XvE0 - XvE1) ],   consolidated.
 * The reference takes "sparsified" XuE matrices (sparse X and E indices); here the Weighted inputs
 * are mapped through their dims first.  Output: tuples sorted by (iE1, iE0), duplicates summed. */
static csc *sparsify_csc(const orc_weighted *w, long nrow_s, long ncol_s, int transpose) {
    const csc *m = w->M;
    dtrip t; t.n = m->nnz;
    t.r = (int *)xmalloc(sizeof(int) * (size_t)(m->nnz ? m->nnz : 1));
    t.c = (int *)xmalloc(sizeof(int) * (size_t)(m->nnz ? m->nnz : 1));
    t.v = (double *)xmalloc(sizeof(double) * (size_t)(m->nnz ? m->nnz : 1));
    long k = 0;
    for (int j = 0; j < m->ncol; ++j)
        for (long p = m->colptr[j]; p < m->colptr[j + 1]; ++p, ++k) {
            int rs = (int)orc_sset_to_sparse(w->dims[0], m->row[p]), cs = (int)orc_sset_to_sparse(w->dims[1], j);
            t.r[k] = transpose ? cs : rs; t.c[k] = transpose ? rs : cs; t.v[k] = m->val[p];
        }
    csc *out = transpose ? to_eigen(&t, (int)ncol_s, (int)nrow_s) : to_eigen(&t, (int)nrow_s, (int)ncol_s);
    dtrip_free(&t);
    return out;
}
/* Eigen sparse a - b: union of the stored entries per column, rows ascending; a-b where both are
 * stored, a where only a is, 0-b where only b is */
static csc *csc_sub(const csc *a, const csc *b) {
    csc *m = csc_alloc(a->nrow, a->ncol, a->nnz + b->nnz);
    long n = 0;
    for (int j = 0; j < a->ncol; ++j) {
        long p = a->colptr[j], q = b->colptr[j];
        const long pe = a->colptr[j + 1], qe = b->colptr[j + 1];
        while (p < pe || q < qe) {
            if (q >= qe || (p < pe && a->row[p] < b->row[q])) { m->row[n] = a->row[p]; m->val[n] = a->val[p]; ++p; }
            else if (p >= pe || b->row[q] < a->row[p]) { m->row[n] = b->row[q]; m->val[n] = 0.0 - b->val[q]; ++q; }
            else { m->row[n] = a->row[p]; m->val[n] = a->val[p] - b->val[q]; ++p; ++q; }
            ++n;
        }
        m->colptr[j + 1] = n;
    }
    m->nnz = n;
    return m;
}
typedef struct { long i, j; double v; long seq; } e_tuple;
static int cmp_tuple(const void *pa, const void *pb) {
    const e_tuple *a = (const e_tuple *)pa, *b = (const e_tuple *)pb;
    if (a->i != b->i) return a->i < b->i ? -1 : 1;
    if (a->j != b->j) return a->j < b->j ? -1 : 1;
    return a->seq < b->seq ? -1 : a->seq > b->seq;      /* std::sort leaves equal tuples unordered: taken in sheet order */
}
long orc_e1ve0c(int nsheets, const orc_weighted *const *XuE1s, const orc_weighted *const *XuE0s, long nE,
                long **out_i, long **out_j, double **out_v)
{
    e_tuple *tl = NULL; long nt = 0, cap = 0;
    double *sE1 = (double *)xcalloc((size_t)nE, sizeof(double));
    for (int s = 0; s < nsheets; ++s) {
        const orc_weighted *XuE1 = XuE1s[s], *XuE0 = XuE0s[s];
        long nX = orc_sset_sparse_extent(XuE1->dims[0]);
        /* sXuE0 = 1 / XuE0->wM, sXuE1 = 1 / XuE1->wM  (sparse X indexing) */
        double *s0 = (double *)xcalloc((size_t)nX, sizeof(double)), *s1 = (double *)xcalloc((size_t)nX, sizeof(double));
        for (int d = 0; d < XuE0->M->nrow; ++d) s0[orc_sset_to_sparse(XuE0->dims[0], d)] = XuE0->wM[d];
        for (int d = 0; d < XuE1->M->nrow; ++d) s1[orc_sset_to_sparse(XuE1->dims[0], d)] = XuE1->wM[d];
        for (long x = 0; x < nX; ++x) { s0[x] = 1. / s0[x]; s1[x] = 1. / s1[x]; }
        csc *E1uX = sparsify_csc(XuE1, nX, nE, 1);                    /* XuE1->M->transpose() */
        csc *M1 = sparsify_csc(XuE1, nX, nE, 0), *M0 = sparsify_csc(XuE0, nX, nE, 0);
        csc *XvE1 = csc_scale_rows(M1, s1), *XvE0 = csc_scale_rows(M0, s0);
        csc *D = csc_sub(XvE0, XvE1);
        csc *L = spgemm(E1uX, D);                                     /* E1uX * (XvE0 - XvE1) */
        for (int j = 0; j < L->ncol; ++j)                             /* spcopy(accum::ref(E1vE0c), local) */
            for (long p = L->colptr[j]; p < L->colptr[j + 1]; ++p) {
                if (nt == cap) { cap = cap ? 2 * cap : 1024; tl = (e_tuple *)xrealloc(tl, sizeof(e_tuple) * (size_t)cap); }
                tl[nt].i = L->row[p]; tl[nt].j = j; tl[nt].v = L->val[p]; tl[nt].seq = nt; ++nt;
            }
        for (int d = 0; d < XuE1->M->ncol; ++d) { long e = orc_sset_to_sparse(XuE1->dims[1], d); sE1[e] += XuE1->Mw[d]; }
        csc_free(E1uX); csc_free(M1); csc_free(M0); csc_free(XvE1); csc_free(XvE0); csc_free(D); csc_free(L);
        free(s0); free(s1);
    }
    for (long e = 0; e < nE; ++e) sE1[e] = 1. / sE1[e];
    for (long k = 0; k < nt; ++k) tl[k].v *= sE1[tl[k].i];
    /* consolidate(): sort by index, sum equal indices */
    qsort(tl, (size_t)nt, sizeof(e_tuple), cmp_tuple);
    long n = 0;
    for (long k = 0; k < nt; ++k) {
        if (n && tl[n - 1].i == tl[k].i && tl[n - 1].j == tl[k].j) tl[n - 1].v += tl[k].v;
        else tl[n++] = tl[k];
    }
    *out_i = (long *)xmalloc(sizeof(long) * (size_t)(n ? n : 1));
    *out_j = (long *)xmalloc(sizeof(long) * (size_t)(n ? n : 1));
    *out_v = (double *)xmalloc(sizeof(double) * (size_t)(n ? n : 1));
    for (long k = 0; k < n; ++k) { (*out_i)[k] = tl[k].i; (*out_j)[k] = tl[k].j; (*out_v)[k] = tl[k].v; }
    free(tl); free(sE1);
    return n;
}
void orc_free(void *p) { free(p); }
