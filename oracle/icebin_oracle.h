/*
 * icebin_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of IceBin's conservative-regridding hot path
 * (RegridMatrices assembly + Weighted apply), used ONLY as the checker in
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in
 * icebin_amd/ (the product) may include, link or call this file.
 *
 * Parity status: the reference's arithmetic lives in the un-vendored
 * ibmisc/spsparse/Eigen3 libraries and cannot be built here, so
 * the oracle follows IceBin's own call sites line by line and restates the
 * published Eigen/spsparse algorithms they call (each function cites the
 * reference file:line).  It is pinned by every known-answer the reference's
 * tests hold for this path (tests/test_oracle_*.py: linterp_1d known answer,
 * the elev_mask fixture, and the test_conserv.py invariants).  The
 * `apply()` fill / force_conservation branches have no reference test:
 * PARITY UNPINNED for those two branches.
 */
#ifndef ICEBIN_ORACLE_H
#define ICEBIN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- spsparse::SparseSet<long,int>  (eigen_types.hpp:16-32; use sites
 *      AbbrGrid.cpp:108, RegridMatrices_Dynamic.cpp:69-72) ------------------ */
typedef struct orc_sset orc_sset;
orc_sset *orc_sset_new(long sparse_extent);
void      orc_sset_free(orc_sset *s);
void      orc_sset_set_sparse_extent(orc_sset *s, long n);
long      orc_sset_sparse_extent(const orc_sset *s);
int       orc_sset_dense_extent(const orc_sset *s);
int       orc_sset_add_dense(orc_sset *s, long sparse);      /* first-seen numbering */
int       orc_sset_to_dense(const orc_sset *s, long sparse); /* -1 if missing */
long      orc_sset_to_sparse(const orc_sset *s, int dense);
const long *orc_sset_to_sparse_ptr(const orc_sset *s);

/* ---- the state of GCMRegridder_Standard + IceRegridder_L0 the path reads --- */
enum { ORC_Z_INTERP = 0, ORC_ELEV_CLASS_INTERP = 1 };  /* IceRegridder.hpp:36-39 */

typedef struct orc_regridder {
    /* ExchangeGrid (AbbrGrid.hpp:40-89) */
    long          nX;
    const int    *ex_indices;   /* [2*nX] interleaved (iA_sparse, iI_sparse) */
    const double *ex_area;      /* [nX] native_area */
    /* ice grid */
    long          nI;           /* sparse extent of the ice grid */
    /* agridA (AbbrGrid.hpp:93-109) + IceRegridder::gridA_proj_area (IceRegridder.cpp:106-118) */
    long          nA;           /* sparse extent of the atmosphere grid */
    int           nA_dense;
    const long   *A_to_sparse;  /* [nA_dense] agridA->dim dense->sparse */
    const double *A_native_area;/* [nA_dense] */
    const double *A_proj_area;  /* [nA_dense] gridA_proj_area */
    /* elevation classes (GCMRegridder.hpp:249-273) */
    int           nhc;
    const double *hcdefs;       /* [nhc], ascending */
    long          hc_stride_A;  /* indexingHC: iE = iA*hc_stride_A + ihc*hc_stride_HC */
    long          hc_stride_HC;
    int           interp_style; /* ORC_Z_INTERP | ORC_ELEV_CLASS_INTERP */
    /* agridI.centroid_xy (AbbrGrid.hpp:108-109) by SPARSE ice index, [2*nI] (x,y); may be NULL when
     * no smoothing is requested (only smoother.cpp reads it) */
    const double *I_centroid_xy;
} orc_regridder;

/* ---- ibmisc::linear::Weighted_Eigen (shape visible at RegridMatrices_Dynamic.cpp:63-65,100,115,123) */
typedef struct orc_weighted orc_weighted;

/* RegridMatrices_Dynamic::matrix_d (RegridMatrices_Dynamic.cpp:412-423).
 * dims may be pre-populated (IceCoupler.cpp:370-371); NULL -> a fresh set is
 * allocated and owned by the result (RegridMatrices_Dynamic::matrix, :425-437).
 * Returns 0 on success; on error writes a message into err. */
int orc_matrix_d(const orc_regridder *rg, const double *elevmaskI,
                 const char *spec_name, orc_sset *dim0, orc_sset *dim1,
                 int scale, int correctA, const double sigma[3],
                 orc_weighted **out, char *err, int errlen);
void orc_weighted_free(orc_weighted *w);

int  orc_weighted_nrow(const orc_weighted *w);
int  orc_weighted_ncol(const orc_weighted *w);
long orc_weighted_nnz(const orc_weighted *w);
int  orc_weighted_conservative(const orc_weighted *w);
int  orc_weighted_scaled(const orc_weighted *w);
const orc_sset *orc_weighted_dim(const orc_weighted *w, int k);
const double *orc_weighted_wM(const orc_weighted *w);
const double *orc_weighted_Mw(const orc_weighted *w);
/* M as row-major-sorted COO (row asc, col asc) */
void orc_weighted_coo(const orc_weighted *w, int *row, int *col, double *val);
/* Build a Weighted from dense-indexed COO (the to_eigen_M route, eigen_types.cpp:9-34) */
orc_weighted *orc_weighted_from_coo(int nrow, int ncol, long nnz, const int *row,
                                    const int *col, const double *val,
                                    const double *wM, const double *Mw,
                                    int conservative, int scaled);

/* Weighted_Eigen::apply (call sites modele/merge_topo.cpp:65, modele/icebin22m.cpp:153).
 * A_b [nvar x ncol] row-major (each variable contiguous), B_b [nvar x nrow].
 * Summation order = Eigen's SparseMatrix<ColMajor> * Dense (column scatter). */
void orc_apply(const orc_weighted *w, const double *A_b, int nvar, double fill,
               int force_conservation, double *B_b);
/* The coupler's product M * (V*T + b) (IceCoupler.cpp:203-252 construct_ice_ivalsI, :445 gcm_ivalsX):
 * V_b [nvar_in x ncol] field-major, T [nvar_in x nvar_out] row-major dense holding the SPARSE
 * variable transform (exact zeros are structural: they are skipped, as Eigen's dense*sparse
 * product skips them), b [nvar_out].  X' = V*T + b is formed first, in Eigen's order (per output
 * variable, stored T entries by ascending input variable, then + b), then M is applied. */
void orc_apply_transformed(const orc_weighted *w, const double *V_b, int nvar_in, const double *T,
                           const double *b, int nvar_out, double fill, double *B_b);
/* apply_weight(dim, A) (matrix_formats.rst:167-186): out[k] = sum_j w[j]*A[k,j] */
void orc_apply_weight(const orc_weighted *w, int dim, const double *A_b, int nvar, double *out);

/* Timed CPU baseline legs (bench.py cpu_baseline only).  threads==1 is the
 * reference-faithful Eigen loop; threads>1 is a row-partitioned CSR SpMM. */
double orc_time_apply(const orc_weighted *w, const double *A_b, int nvar,
                      double *B_b, int reps, int threads);

/* ---- small known-answer helpers -------------------------------------------- */
/* IceRegridder_L0.cpp:71-93; returns 0, or -1 if xx is above the last point */
int  orc_linterp_1d_b(const double *xpoints, int n, double xx, long idx[2], double w[2]);
/* IceRegridder.cpp:181-204 */
void orc_linterp_1d(const double *xpoints, int n, double xx, int idx[2], double w[2]);
/* IceRegridder_L0.cpp:43-67 */
int  orc_nearest_1d(const double *xpoints, int n, double xx);
/* ElevMask.cpp:57-70: mask/thk/topg -> (emI_land, emI_ice) */
void orc_elevmask_pism(long n, const signed char *mask, const double *thk,
                       const double *topg, double *emI_land, double *emI_ice);
/* legacy coo_matvec (pylib/icebin_cython.cpp:158-192) */
void orc_coo_matvec(double *yy, const double *xx, int ignore_nan, long nrow, long ncol,
                    long nnz, const int *row, const int *col, const double *data);

/* compute_E1vE0c (slib/icebin/e1ve0.cpp:55-106): returns the number of tuples; arrays are malloc'd
 * (release with orc_free), sorted by (iE1, iE0) sparse indices. */
long orc_e1ve0c(int nsheets, const orc_weighted *const *XuE1s, const orc_weighted *const *XuE0s, long nE,
                long **out_i, long **out_j, double **out_v);
void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
