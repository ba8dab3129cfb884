/*
 * icebin_hip.h -- C-ABI of libicebin_hip.so, the MI355X (gfx950) implementation
 * of IceBin's conservative-regridding hot path.
 *
 * This is the drop-in boundary: plain C, opaque handles, plain
 * pointers and sizes, int status codes.  Every entry point replaces one piece
 * of the reference's C++ interface, cited as file:line under /root/reference.
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md;
 * icebin_amd/host/ (C++) and the icebin_amd Python package (ctypes) are two users of it.
 *
 * Conventions
 *   - All functions return IBH_OK (0) or a negative IBH_E* code; the message is
 *     available from ibh_last_error() (thread-local).  This replaces
 *     (*icebin_error)(-1, fmt, ...) (slib/icebin/error.hpp:28-32): the host
 *     wrappers re-throw it as a C++ exception / Python RuntimeError.
 *   - Host arrays are borrowed for the duration of the call only.
 *   - "_d" = dense (renumbered) index, "_s" = sparse (native) index
 *     (sphinx/source/sparse_matrices.rst).  Dense ids are int32, sparse ids
 *     int64, values double (slib/icebin/eigen_types.hpp:16-18).
 *   - Fields ("variables") are stored field-major: A_b[k*lda + j], the
 *     blitz::Array<double,2>(nvar, n) layout of modele/icebin22m.cpp:142.
 *   - Single-threaded per handle; the stream argument is a hipStream_t passed
 *     as void* (NULL = the default stream).  The library uses the current HIP
 *     device of the calling thread; handles are bound to the device they were
 *     created on.
 *   - There is no CPU fallback: without a HIP device every compute entry
 *     point fails with IBH_ENODEVICE.
 */
#ifndef ICEBIN_HIP_H
#define ICEBIN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IBH_OK          0
#define IBH_EINVAL     -1   /* bad argument / inconsistent sizes                       */
#define IBH_ENODEVICE  -2   /* no usable HIP device                                    */
#define IBH_EHIP       -3   /* a HIP runtime call failed                               */
#define IBH_ERANGE     -4   /* elevation above the last height class                   */
                            /*   (IceRegridder_L0.cpp:84-85)                           */
#define IBH_ENOTIMPL   -5   /* feature outside the hot-path scope                      */
#define IBH_ENOKEY     -6   /* unknown matrix name (regrids.at(), RegridMatrices_Dynamic.cpp:419) */

const char *ibh_last_error(void);
/* Library / device info.  ibh_device_count does not initialise the GPU context. */
int ibh_version(void);
int ibh_device_count(int *count);
int ibh_set_device(int device);

/* ------------------------------------------------------------------------- */
/* SparseSet: spsparse::SparseSet<long,int> (eigen_types.hpp:24; used as the
 * `dims` of every matrix, RegridMatrices_Dynamic.hpp:51-54).  Host-side value
 * type: dense -> sparse table in first-seen order.  Passed IN/OUT to matrix_d:
 * a pre-populated set is appended to, never reset (IceCoupler.cpp:366-371). */
typedef struct ibh_sparse_set ibh_sparse_set;
int ibh_sparse_set_create(int64_t sparse_extent, ibh_sparse_set **out);
int ibh_sparse_set_create_identity(int64_t n, ibh_sparse_set **out);          /* ibmisc id_sparse_set, modele/merge_topo.cpp:48 */
int ibh_sparse_set_from_array(int64_t sparse_extent, const int64_t *to_sparse, int32_t n, ibh_sparse_set **out);
int ibh_sparse_set_destroy(ibh_sparse_set *s);
int ibh_sparse_set_sparse_extent(const ibh_sparse_set *s, int64_t *out);
int ibh_sparse_set_dense_extent(const ibh_sparse_set *s, int32_t *out);
int ibh_sparse_set_to_sparse(const ibh_sparse_set *s, int64_t *out /* [dense_extent] */);
/* SparseSet::to_dense (-1 when the key is absent: in_sparse() == false) and add_dense (existing id or the
 * next one, first-seen order: AbbrGrid.cpp:108, IceCoupler.cpp:298), host side. */
int ibh_sparse_set_to_dense(const ibh_sparse_set *s, int64_t sparse, int32_t *dense);
int ibh_sparse_set_add_dense(ibh_sparse_set *s, int64_t sparse, int32_t *dense);

/* ------------------------------------------------------------------------- */
/* Exchange-grid generation: make_exchange_grid (slib/icebin/gridgen/GridGen_Exchange.cpp:175-284) for a
 * rectilinear ice grid in the projected plane (gridgen/searise_grid.cpp) under convex GCM-cell polygons
 * whose vertices the caller has projected to that plane (OGrid, GridGen_Exchange.cpp:120-166; counter-
 * clockwise, 3..16 vertices, ascending iA).  Every non-empty overlap becomes one exchange cell
 * (iA, iI, area) with area = Cell::proj_area of the overlap polygon (Grid.cpp:42-70); cells come out
 * sorted by (iA, iI) like ExchangeGrid's constructor leaves them (AbbrGrid.cpp:10-21).  The result
 * stays in HBM; ibh_exgrid_get copies it out (the layout ibh_regridder_desc takes). */
typedef struct ibh_exgrid_desc {
    int32_t        nx, ny;          /* ice cells along x and y                                   */
    const double  *xedges, *yedges; /* [nx+1], [ny+1] cell edges, ascending                      */
    int32_t        x_fastest;       /* iI = iy*nx + ix when non-zero, else ix*ny + iy            */
    int32_t        npoly;           /* realised GCM cells                                        */
    const int32_t *polyptr;         /* [npoly+1] vertex ranges                                   */
    const double  *vx, *vy;         /* projected vertices                                        */
    const int64_t *iA;              /* [npoly] sparse atmosphere index, ascending                */
} ibh_exgrid_desc;
typedef struct ibh_exgrid ibh_exgrid;
int ibh_exgrid_generate(const ibh_exgrid_desc *desc, ibh_exgrid **out);
int ibh_exgrid_size(const ibh_exgrid *ex, int64_t *nX);
int ibh_exgrid_get(const ibh_exgrid *ex, int32_t *indices /* [2*nX] */, double *overlaps /* [nX] */);
int ibh_exgrid_destroy(ibh_exgrid *ex);

/* ------------------------------------------------------------------------- */
/* Regridder: the state of GCMRegridder_Standard (GCMRegridder.hpp:207-302) and
 * one IceRegridder_L0 (IceRegridder.hpp:46-133) that the path reads, uploaded
 * to HBM once.  Replaces GCMRegridder_Standard::init + add_sheet
 * (GCMRegridder.cpp:65-86, GCMRegridder.hpp:353-367) and IceRegridder::init
 * (IceRegridder.cpp:93-119) for the data they keep. */
typedef struct ibh_regridder_desc {
    /* ExchangeGrid (AbbrGrid.hpp:40-89) */
    int64_t        nX;
    const int32_t *ex_indices;    /* [2*nX] interleaved (iA_s, iI_s), AbbrGrid.hpp:42 */
    const double  *ex_area;       /* [nX] overlaps, AbbrGrid.hpp:43                  */
    int64_t        nI;            /* ice-grid sparse extent (IceRegridder::nI)       */
    /* agridA (AbbrGrid.hpp:93-109) and IceRegridder::gridA_proj_area */
    int64_t        nA;            /* atmosphere sparse extent (GCMRegridder::nA)     */
    int32_t        nA_dense;
    const int64_t *A_to_sparse;   /* [nA_dense] agridA->dim                          */
    const double  *A_native_area; /* [nA_dense]                                      */
    const double  *A_proj_area;   /* [nA_dense] (== native when sproj=="", IceRegridder.cpp:106-108) */
    /* elevation classes */
    int32_t        nhc;
    const double  *hcdefs;        /* [nhc] ascending (GCMRegridder.hpp:253-256)      */
    int64_t        hc_stride_A;   /* indexingHC.tuple_to_index: iE = iA*stride_A + ihc*stride_HC */
    int64_t        hc_stride_HC;  /*   (icebin_cython.cpp:69 -> 1 and nA)            */
    int32_t        interp_style;  /* 0 Z_INTERP, 1 ELEV_CLASS_INTERP (IceRegridder.hpp:36-39) */
    /* agridI.centroid_xy (AbbrGrid.hpp:108-109) by SPARSE ice index, [2*nI] (x, y); only the smoother
     * (sigma != 0, smoother.cpp:69-99) reads it; may be NULL */
    const double  *I_centroid_xy;
} ibh_regridder_desc;

typedef struct ibh_regridder ibh_regridder;
int ibh_regridder_create(const ibh_regridder_desc *desc, ibh_regridder **out);
int ibh_regridder_destroy(ibh_regridder *rg);
int ibh_regridder_sizes(const ibh_regridder *rg, int64_t *nA, int64_t *nE, int64_t *nI, int64_t *nX, int32_t *nhc);
/* GCMRegridder::wA (GCMRegridder.hpp:305-315, icebin_cython.cpp:103-117):
 * out[nA] = fill, then out[A_to_sparse[id]] = native or projected area. */
int ibh_regridder_wA(const ibh_regridder *rg, int native, double fill, double *out /* [nA] */);

/* ------------------------------------------------------------------------- */
/* Weighted: ibmisc::linear::Weighted_Eigen {dims, M, wM, Mw, conservative,
 * scaled} (RegridMatrices_Dynamic.cpp:63-65,100,115,123,421).  M lives in HBM
 * as CSR (int32 rowptr/colind, f64 values, columns ascending inside a row). */
typedef struct ibh_weighted ibh_weighted;

/* RegridMatrices: GCMRegridder_Standard::regrid_matrices
 * (RegridMatrices_Dynamic.cpp:334-402; Cython shim new_regrid_matrices,
 * icebin_cython.cpp:215-236).  Copies elevmaskI (length must equal nI) to HBM.  For a grid of 2^20
 * exchange cells and more the call also derives one byte per ice cell from its elevation (masked /
 * beyond the last elevation class / first class and class count): every matrix built from this
 * object reads that byte where only the mask or the class pattern matters (36 M ice cells: 0.22 ms
 * for copy + bytes against 0.13 ms for the copy alone; smaller grids make the bytes when a build
 * first asks).  Results are unaffected. */
typedef struct ibh_regrid_matrices ibh_regrid_matrices;
int ibh_regrid_matrices_create(const ibh_regridder *rg, const double *elevmaskI, int64_t n,
                               int scale, int correctA, const double sigma[3],
                               ibh_regrid_matrices **out);
/* Same with the elevation mask already in HBM (an ice model that runs on the GPU: the host form
 * moves 8*nI bytes over PCIe per coupling step, 4 ms at 1 km).  Copied device-to-device on `stream`,
 * which is synchronised before returning. */
int ibh_regrid_matrices_create_device(const ibh_regridder *rg, const double *d_elevmaskI, int64_t n,
                                      int scale, int correctA, const double sigma[3], void *stream,
                                      ibh_regrid_matrices **out);
int ibh_regrid_matrices_destroy(ibh_regrid_matrices *rm);

/* RegridMatrices_Dynamic::matrix_d(spec, dims, params) (:412-423).
 * spec in {AvI IvA AvX XvA EvI IvE EvX XvE EvA AvE}.  dim0/dim1 are IN/OUT;
 * they must outlive the result unless passed as NULL, in which case fresh
 * sets owned by the result are used (== RegridMatrices_Dynamic::matrix, :425-437). */
int ibh_regrid_matrices_matrix_d(const ibh_regrid_matrices *rm, const char *spec,
                                 ibh_sparse_set *dim0, ibh_sparse_set *dim1,
                                 int scale, int correctA, const double sigma[3],
                                 ibh_weighted **out);
/* The matrices of one coupling step in ONE call (IceCoupler.cpp:361-468 builds EvI, AvI, IvE, XvE per step with a
 * shared dimE and identity dimI / dimX): the results -- matrices and dims -- are those of n ibh_regrid_matrices_matrix_d
 * calls in the order given (sigma[3], or NULL for no smoothing, applies to all); builds that cannot influence each
 * other run concurrently on the library's worker threads and their own streams, a build that reads a set an earlier
 * one numbers waits for it.  On error nothing is returned (out[] all NULL) and the first failing job's status is. */
int ibh_regrid_matrices_matrix_batch(const ibh_regrid_matrices *rm, int32_t n, const char *const *specs,
                                     ibh_sparse_set *const *dim0, ibh_sparse_set *const *dim1,
                                     const int32_t *scale, const int32_t *correctA, const double sigma[3],
                                     ibh_weighted **out /* [n] */);
/* RegridMatrices::matrix(spec) (RegridMatrices.hpp:60-61): own dims, params of rm. */
int ibh_regrid_matrices_matrix(const ibh_regrid_matrices *rm, const char *spec, ibh_weighted **out);

/* Load a Weighted from dense-indexed triplets: to_eigen_M (eigen_types.cpp:9-34)
 * / ibmisc.nc_read_weighted (matrix_formats.rst:139-147).  Duplicates are
 * summed in input order (Eigen setFromTriplets).  dims are identity. */
int ibh_weighted_from_coo(int32_t nrow, int32_t ncol, int64_t nnz,
                          const int32_t *row, const int32_t *col, const double *val,
                          const double *wM /* [nrow] */, const double *Mw /* [ncol] */,
                          int conservative, int scaled, ibh_weighted **out);
/* Same from host CSR arrays (columns need not be sorted; no duplicate merging). */
int ibh_weighted_from_csr(int32_t nrow, int32_t ncol, const int32_t *rowptr,
                          const int32_t *colind, const double *val,
                          const double *wM, const double *Mw,
                          int conservative, int scaled, ibh_weighted **out);
int ibh_weighted_destroy(ibh_weighted *w);

/* compute_E1vE0c (slib/icebin/e1ve0.cpp:55-106): the matrix that carries fields on last step's elevation
 * grid (E0) to this step's (E1), E1vE0c = diag(1/sum_sheets Mw(XuE1)) * sum_sheets[E1uX * (XvE0 - XvE1)].
 * XuE1s / XuE0s: one unscaled XvE matrix per ice sheet for the new and the old elevation mask
 * (IceCoupler.cpp:464-468 builds them; dims[0] over the exchange grid, dims[1] over E).  The result is a
 * Weighted over the SPARSE E space (identity dims of extent nE, rows iE1, columns iE0, entries sorted
 * by (iE1, iE0), duplicates summed; wM = Mw = 1): the reference returns the same tuples as a TupleList. */
int ibh_e1ve0_compute(int32_t nsheets, const ibh_weighted *const *XuE1s, const ibh_weighted *const *XuE0s,
                      int64_t nE, ibh_weighted **out);

/* Public members of Weighted_Eigen, read back to host. */
int ibh_weighted_shape(const ibh_weighted *w, int32_t *nrow_d, int32_t *ncol_d, int64_t *nnz);
int ibh_weighted_flags(const ibh_weighted *w, int *conservative, int *scaled);
int ibh_weighted_dim(const ibh_weighted *w, int k, int64_t *sparse_extent, int32_t *dense_extent);
int ibh_weighted_dim_to_sparse(const ibh_weighted *w, int k, int64_t *out /* [dense_extent] */);
int ibh_weighted_get_wM(const ibh_weighted *w, double *out /* [nrow_d] */);
int ibh_weighted_get_Mw(const ibh_weighted *w, double *out /* [ncol_d] */);
/* ->M in row-major COO order (row asc, col asc); any of row/col/val may be NULL. */
int ibh_weighted_get_coo(const ibh_weighted *w, int32_t *row, int32_t *col, double *val);
int ibh_weighted_get_csr(const ibh_weighted *w, int32_t *rowptr, int32_t *colind, double *val);

/* Weighted_Eigen::apply(A_b, fill, force_conservation, tmp)
 * (call sites modele/merge_topo.cpp:65, modele/icebin22m.cpp:153; inline Eigen
 * products IceCoupler.cpp:237,445).  B_b[k, :] = M * A_b[k, :] for k < nvar in
 * dense index spaces; rows with wM == 0 receive `fill`; when the matrix is
 * not conservative and force_conservation != 0 each variable is rescaled by
 * (Mw . A) / (wM . B).  _host takes host pointers (copies over PCIe);
 * _device takes device pointers and only enqueues work on `stream`: no allocation, copy or
 * synchronisation once the handle has been prepared (ibh_weighted_prepare).  On a handle that was
 * NOT prepared an apply may (a) grow scratch (allocates; an error inside a stream capture) and (b) on
 * the SECOND apply of a large elevation-class matrix build the column-sweep / band structure the faster
 * kernels read (allocates, synchronises `stream` a few times; skipped inside a capture; a build that
 * fails leaves the matrix on its row-by-row kernel, the apply still succeeds) --
 * ibh_set_tuning("lazy_structures", 0) switches (b) off for the process.  lda/ldb:
 * distance in doubles between consecutive variables (>= dense extents); any value
 * works, ldb a multiple of 64 (512-byte planes) is fastest for the I-row matrices.
 * The handle keeps small per-apply scratch buffers: use one stream at a time per handle. */
int ibh_weighted_apply_host(const ibh_weighted *w, const double *A_b, int32_t nvar, int64_t lda,
                            double *B_b, int64_t ldb, double fill, int force_conservation);
int ibh_weighted_apply_device(const ibh_weighted *w, const double *dA_b, int32_t nvar, int64_t lda,
                              double *dB_b, int64_t ldb, double fill, int force_conservation,
                              void *stream);
/* Several field batches through ONE launch: for q < nbatch, dB_b[q] = M * dA_b[q], every batch
 * [nvar x ncol_d] -> [nvar x nrow_d] with the same nvar / lda / ldb -- what a caller that applies one
 * matrix to several groups of variables in a row does (Weighted_Eigen::apply call sites
 * modele/merge_topo.cpp:65, modele/icebin22m.cpp:153; the coupler's per-sheet products IceCoupler.cpp:237,445).
 * dA_b / dB_b are HOST arrays of nbatch DEVICE pointers, read during the call (they travel in the
 * kernel arguments: no device-side table, nothing left pending on them).  Results are bitwise those of
 * nbatch separate ibh_weighted_apply_device calls.  A 64-field 5 km apply is a latency-sized problem
 * (launch + dependent loads are a third of its 11 us); batched, that cost is paid once per launch.
 * More than IBH_MAX_BATCH batches are split into several launches. */
#define IBH_MAX_BATCH 32
int ibh_weighted_apply_many_device(const ibh_weighted *w, int32_t nbatch, const double *const *dA_b, int32_t nvar,
                                   int64_t lda, double *const *dB_b, int64_t ldb, double fill,
                                   int force_conservation, void *stream);
/* Make applies of up to nvar variables, up to nbatch field batches per ibh_weighted_apply_many_device
 * call, pure enqueues: builds NOW (synchronously, on the default stream) whatever structure those applies
 * would otherwise build lazily (column sweep / bands: see above) and sizes every per-apply scratch buffer
 * (transposed inputs of the I-row kernels, partial sums of the sweep / band kernels for the batch depth,
 * conservation factors, transform scratch).  Call it once after a matrix is built and before capturing
 * applies into a hipGraph, or whenever the latency of the first applies matters.  Eager and captured
 * applies of a prepared handle run the same kernels and are bitwise equal.
 * ibh_weighted_reserve(w, nvar) only sizes scratch for single applies (nbatch = 1) and builds nothing. */
int ibh_weighted_prepare(const ibh_weighted *w, int32_t nvar, int32_t nbatch);

/* Fused pair: B1 = first * A and B2 = second * B1 in ONE launch -- the step "ice -> elevation classes -> atmosphere" of a
 * coupler (E = EvI * I, A = AvE * E; the reference makes two Weighted_Eigen::apply calls, ibmisc linear/eigen.cpp, one per
 * matrix: RegridMatrices_Dynamic.cpp:354-390 hands out the two matrices).  Possible when every row of `second` reads rows of one
 * row group of `first` only, which AvE after EvI does by construction (a GCM cell's value combines that cell's own elevation
 * classes); the rows need not be numbered alike (the pairing goes through the sparse indices of the shared dimension).
 * ibh_weighted_pair_prepare builds `first`'s row groups if it has none, works the pairing out on the host (synchronises) and
 * returns IBH_ENOTIMPL when the two matrices do not pair -- the caller then makes two applies.  ibh_weighted_apply_pair_device is
 * a pure enqueue (graph-capturable); B1 is bitwise what ibh_weighted_apply_device(first) writes, B2 is the second matrix's
 * result with the same fill / wM == 0 rule, summed in a fixed order (1e-12 of the separate apply, bitwise when `second` lists
 * its columns in the order of `first`'s rows).  Conservative matrices only (no smoothing). */
int ibh_weighted_pair_prepare(const ibh_weighted *first, const ibh_weighted *second, int32_t nvar);
int ibh_weighted_apply_pair_device(const ibh_weighted *first, const ibh_weighted *second, const double *dA_b, int32_t nvar, int64_t lda,
                                   double *dB1_b, int64_t ldb1, double *dB2_b, int64_t ldb2, double fill, void *stream);
/* The chain of BASELINE config 3, ice -> elevation classes -> atmosphere -> ice, in one call: B1 = first * A and B2 = second * B1
 * as the fused pair above (ibh_weighted_pair_prepare(first, second) first), B3 = third * B2 (IvA; the reference: three
 * Weighted_Eigen::apply calls, IceCoupler.cpp:203-252 builds its inputs the same way).  Two stream-ordered launches; B1, B2, B3 are
 * bitwise what the pair apply followed by ibh_weighted_apply_device(third) writes; a pure enqueue (graph-capturable).
 * ibh_set_tuning("chain_overlap", 1) dispatches the second launch without a queue barrier behind the pair kernel and lets its
 * workgroups wait on a device flag instead (same bits; measured slower on MI355X / ROCm 7.2: DESIGN.md K1f). */
int ibh_weighted_apply_chain_device(const ibh_weighted *first, const ibh_weighted *second, const ibh_weighted *third, const double *dA_b,
                                    int32_t nvar, int64_t lda, double *dB1_b, int64_t ldb1, double *dB2_b, int64_t ldb2, double *dB3_b,
                                    int64_t ldb3, double fill, void *stream);
int ibh_weighted_reserve(const ibh_weighted *w, int32_t nvar);
/* The coupler's fused product B = M * (A*T + b) (IceCoupler.cpp:203-252 construct_ice_ivalsI and
 * :445 gcm_ivalsX = M * (ice_ovalsI*T + b)): dA_b [nvar_in x ncol_d] field-major device pointer,
 * T [nvar_in x nvar_out] row-major HOST array holding the sparse variable transform (exact zeros are
 * structural and skipped, as Eigen's dense*sparse product does), b [nvar_out] host, dB_b
 * [nvar_out x nrow_d] device.  The small dense transform is applied on the SMALL side of M (inputs
 * when ncol_d <= nrow_d, outputs otherwise), so no transformed copy of the large field array is
 * ever written.  Uses scratch owned by the handle: calls on one handle must be stream-ordered.
 * T and b are consumed during the call (kernel arguments) when nvar_in*nvar_out + nvar_out <= 384;
 * larger transforms are staged through a device copy and synchronise the stream. */
int ibh_weighted_apply_transformed_device(const ibh_weighted *w, const double *dA_b, int32_t nvar_in, int64_t lda,
                                          const double *T, const double *b, int32_t nvar_out,
                                          double *dB_b, int64_t ldb, double fill, void *stream);
/* The legacy COO product behind icebin.coo_multiply(M, x, fill, ignore_nan) (coo_matvec,
 * pylib/icebin_cython.cpp:158-192; used by tests/test_conserv/test_conserv.py:139-205 and
 * pylib/icebin/ibplotter.py:88): yy[row] = sum of data*xx[col] over the row's entries, skipping entries
 * whose input is NaN when ignore_nan != 0; a row with no surviving entry is NOT written (yy keeps what
 * the caller preset, i.e. `fill`).  No wM test, no conservation correction.
 * _device: on the CSR of a Weighted, field-major device arrays, only enqueues work on `stream`.
 * ibh_coo_matvec: the reference's signature (host arrays, arbitrary triplet order, duplicates add up). */
int ibh_weighted_matvec_device(const ibh_weighted *w, const double *dxx, int32_t nvar, int64_t ldx,
                               double *dyy, int64_t ldy, int ignore_nan, void *stream);
int ibh_coo_matvec(double *yy /* [nrow] in/out */, const double *xx /* [ncol] */, int ignore_nan,
                   int64_t nrow, int64_t ncol, int64_t nnz, const int32_t *row, const int32_t *col,
                   const double *data);
/* linear_Weighted.apply_weight(dim, A) (matrix_formats.rst:167-186):
 * out[k] = sum_j w[j] * A_b[k*lda + j], w = wM (dim 0) or Mw (dim 1); host pointers. */
int ibh_weighted_apply_weight_host(const ibh_weighted *w, int dim, const double *A_b,
                                   int32_t nvar, int64_t lda, double *out /* [nvar] */);

/* ------------------------------------------------------------------------- */
/* Field-sharded applies across the GPUs of one node (BASELINE.json north_star).  One process per GPU:
 * rank r owns the fields [r*nvar_local, (r+1)*nvar_local) of a regrid of world*nvar_local fields (contiguous in the
 * field-major arrays), every rank holds the same matrix (built redundantly: a build is cheaper than broadcasting it),
 * the SpMM needs no communication and the results are reassembled on EVERY rank.  The reference has no counterpart:
 * it gathers everything to MPI rank 0 and regrids there (modele/GCMCoupler_ModelE.cpp:764-792).
 *   ibh_comm_unique_id   rank 0 obtains an id (ncclGetUniqueId) and ships it to the other ranks by whatever the host
 *                        program has (MPI_Bcast in ModelE: GCMCoupler_ModelE.cpp; a file; torch.distributed);
 *   ibh_comm_create      every rank: the communicator on its CURRENT device (ncclCommInitRank; id may be NULL for
 *                        world == 1, which needs no RCCL).  RCCL is loaded at run time (dlopen; ICEBIN_RCCL_LIB
 *                        overrides the search): without it only world == 1 and custom transports work;
 *   ibh_comm_create_custom  the same choreography over a transport the caller supplies (tests; other fabrics):
 *                        fn(user, d_base, count, stride, world, rank, stream) must deliver every rank's `count` doubles at
 *                        d_base + rank*stride to d_base + rank*stride on every peer, ordered on `stream`; 0 = success. */
#define IBH_UNIQUE_ID_BYTES 128
typedef struct ibh_comm ibh_comm;
typedef int (*ibh_exchange_fn)(void *user, double *d_base, int64_t count, int64_t stride, int world, int rank, void *stream);
int ibh_comm_unique_id(char id[IBH_UNIQUE_ID_BYTES]);
int ibh_comm_create(int world, int rank, const char id[IBH_UNIQUE_ID_BYTES], ibh_comm **out);
int ibh_comm_create_custom(int world, int rank, ibh_exchange_fn fn, void *user, ibh_comm **out);
/* A custom transport that also carries pieces of UNEQUAL size (the sharded assembly, ibh_regrid_matrices_matrix_d_sharded):
 * fn(user, d_base, offsets[world + 1] (bytes), world, rank, stream) must deliver every rank q's bytes
 * [offsets[q], offsets[q+1]) of d_base to the same place on every peer, ordered on `stream`; 0 = success. */
typedef int (*ibh_gatherv_fn)(void *user, void *d_base, const int64_t *offsets, int world, int rank, void *stream);
int ibh_comm_set_custom_gatherv(ibh_comm *c, ibh_gatherv_fn fn);
/* The exchange stream.  By default the communicator creates its own (non-blocking) stream and DESTROYS it in ibh_comm_destroy:
 * the `stream` a custom transport's callbacks receive for the field exchanges is that stream, and it is invalid once
 * ibh_comm_destroy has returned -- a transport whose allocator keeps per-stream state (torch's pinned-host cache records the
 * streams a block was used on) must drop that state before it destroys the communicator, or hand in a stream of its own:
 * ibh_comm_set_stream makes the communicator enqueue its exchanges on the CALLER's stream, which the caller keeps alive until
 * after ibh_comm_destroy and destroys itself (the library synchronises it in ibh_comm_destroy, never destroys it).  Call it
 * before the first apply; a stream that also carries the SpMMs serialises exchange and compute.  (The callbacks of the sharded
 * ASSEMBLY, ibh_gatherv_fn and the 32-byte ibh_exchange_fn call of ibh_regrid_matrices_matrix_d_sharded, always receive the
 * caller's build stream.) */
int ibh_comm_set_stream(ibh_comm *c, void *stream);
/* Options of one communicator.  "planes_padded" (default 0): 1 = the caller owns the gap [nrow_d, ldb) of every result plane of
 * the sharded applies on this communicator as padding (planes rounded up to whole 512-byte lines, as the library's own wrappers
 * allocate them): the planes of a field block then travel as ONE piece and the padding of the peers' planes is overwritten.
 * With 0 nothing outside [0, nrow_d) of a plane is touched on any rank: planes travel one by one unless ldb == nrow_d.
 * Unknown key: IBH_ENOKEY. */
int ibh_comm_set_option(ibh_comm *c, const char *key, int value);
int ibh_comm_destroy(ibh_comm *c);
int ibh_comm_info(const ibh_comm *c, int *world, int *rank);
/* Weighted::apply of world*nvar_local fields, sharded by field: dB_all [world*nvar_local x ldb] (device, the same
 * shape on every rank) receives rank r's results in rows [r*nvar_local, (r+1)*nvar_local).  The local SpMM is enqueued on
 * `stream` and writes straight into this rank's rows; the exchange -- direct peer-to-peer sends (xGMI is a full mesh:
 * seven concurrent transfers use all links) -- runs on a stream the communicator owns, `block_fields` fields at a time
 * (0: chosen by size -- one exchange for KB-sized results, blocks of >= 8 fields and ~128 MB for the GB-sized results of
 * the I-row matrices) so that it overlaps the SpMM of the following block and of the following apply.  dB_all is complete once
 * ibh_comm_wait(c, s) has made stream s wait for the exchanges enqueued so far.  Same results as ibh_weighted_apply_device
 * on each rank's fields (bitwise), conservative matrices only (no force_conservation).
 * ldb > nrow_d: a true leading dimension (a column view of a larger array) unless the communicator says otherwise -- every plane
 * travels by itself and nothing outside [0, nrow_d) of a plane is touched on any rank; ibh_comm_set_option(c, "planes_padded", 1)
 * declares the gap padding owned by the caller, the planes of a block then travel as one piece (fewer, larger transfers).
 * Streams: any stream may issue applies and waits on one communicator; an apply whose results overlap an exchange still in
 * flight is ordered behind it whichever stream enqueued that exchange (the last 8 exchanges are tracked individually, older
 * unfinished ones through the range hull of their successor). */
int ibh_weighted_apply_sharded_device(const ibh_weighted *w, ibh_comm *c, const double *dA_local, int32_t nvar_local,
                                      int64_t lda, double *dB_all, int64_t ldb, double fill, int32_t block_fields,
                                      void *stream);
/* The same for nbatch independent field batches (ibh_weighted_apply_many_device: ONE SpMM launch) whose results travel in
 * ONE grouped exchange -- for KB-sized results (A- and E-row matrices: a [64, 122] AvI result is 62 KB) an exchange per
 * apply is latency-bound, a group of 32 is not.  dA_local / dB_all: host arrays of nbatch device pointers. */
int ibh_weighted_apply_many_sharded_device(const ibh_weighted *w, ibh_comm *c, int32_t nbatch, const double *const *dA_local,
                                           int32_t nvar_local, int64_t lda, double *const *dB_all, int64_t ldb, double fill,
                                           void *stream);
int ibh_comm_wait(ibh_comm *c, void *stream);
/* RegridMatrices_Dynamic::matrix_d with the ASSEMBLY shared by the ranks of a communicator (BASELINE.json config 5: "overlap
 * COO->CSR assembly + apply, 8 x MI355X"; the reference rebuilds its matrices every coupling step, IceCoupler.cpp:361-468, on one
 * core).  Collective: every rank calls it with the same arguments on the same regridder / elevation mask and receives the WHOLE
 * matrix -- bitwise the result of ibh_regrid_matrices_matrix_d.  The exchange grid is sorted by atmosphere cell
 * (AbbrGrid.cpp:10-21), so rank k runs the streamed passes over a contiguous block of ranges holding ~1/world of the exchange
 * cells; three exchanges on the caller's thread stream (counters; the first-seen flags and class ranks the ranks look up in each
 * other's blocks, ~1 byte per exchange cell; the pieces of the CSR / weights / dims tables) complete it.  Served this way: the
 * eight A/E/I/X matrices on sets numbered by the build (dims NULL or empty) AND on the sets the reference's coupler passes
 * (IceCoupler.cpp:366-377, 462-467) -- an identity dimI / dimX on the ice / exchange side (A/E-row matrices: Mw travels as
 * {position, value} pairs; X rows: a block of cells is a block of rows; I rows: the row lengths are merged before the row pointer
 * is scanned, the rows travel in first-seen order and are copied to their places), a pre-populated dimE as the column set (looked
 * up in the table every rank has; the column sums travel as pairs) --, sorted grids, at most 8 ranks; anything else (EvA / AvE, a
 * pre-populated row set, unsorted grids) is built redundantly on every rank (same result).  ibh_weighted_built_fast reports 3 for a shared build.  A custom transport needs
 * ibh_comm_set_custom_gatherv. */
int ibh_regrid_matrices_matrix_d_sharded(const ibh_regrid_matrices *rm, ibh_comm *c, const char *spec, ibh_sparse_set *dim0,
                                         ibh_sparse_set *dim1, int scale, int correctA, ibh_weighted **out);

/* Device pointers of the CSR and weights, for callers that keep fields resident
 * (IceCoupler.cpp:408,445,456 read ->M and ->wM directly). */
typedef struct ibh_weighted_device_view {
    int32_t nrow, ncol; int64_t nnz;
    const int32_t *rowptr, *colind; const double *val, *wM, *Mw;
} ibh_weighted_device_view;
int ibh_weighted_device_view_get(const ibh_weighted *w, ibh_weighted_device_view *out);

/* Tuning / introspection (not part of the reference interface). */
int ibh_weighted_set_kernel(ibh_weighted *w, const char *name_or_auto);   /* "auto", "rowblock", "shortrow", "rowdual", "colsweep", "rowgroup" */
int ibh_weighted_last_kernel(const ibh_weighted *w, char *buf, int buflen);
/* The kernel INSTANTIATION the last apply launched, spelled as rocprofv3 prints it ("spmm_rowblock_kernel<1, 1, 14, 8, false>",
 * "spmm_rowone_kernel<8, 14>"; "" for the other kernel families): bench.py checks it against the kernel a committed PMC profile
 * was taken on before quoting that profile's traffic. */
int ibh_weighted_last_launch(const ibh_weighted *w, char *buf, int buflen);
/* A launch option of THIS matrix (the apply-side keys of ibh_set_tuning: rowblock_*, rowone*, rowgroup_*, shortrow_*, sweep_*,
 * rowdual_*, lazy_structures): read before the process-wide map by every apply / prepare of this handle, so host threads that
 * tune different matrices do not interfere.  value INT32_MIN removes the option.  Like every call on a handle: not concurrently
 * with an apply of the same handle. */
int ibh_weighted_set_option(ibh_weighted *w, const char *key, int value);
/* 1 when the matrix was assembled by the plan-based fast path for sorted exchange grids (fastasm.inl), 3 when the ranks of a
 * communicator shared the streamed build (ibh_regrid_matrices_matrix_d_sharded), 2 when its streamed
 * variant did (streamasm.inl: grids of 2^20 exchange cells and more; ibh_set_tuning("assemble_stream", 0 | 1) overrides),
 * 0 when the general pipeline built it (ibh_set_tuning("assemble_fast", 0) forces the latter).  Results are bit-identical. */
int ibh_weighted_built_fast(const ibh_weighted *w, int *out);
/* PROCESS-WIDE launch-heuristic overrides for measurements and tests -- experiments only; a product caller that needs a knob sets it
 * on the handle (ibh_weighted_set_option).  (README.md lists the keys: assemble_fast, assemble_fast_eva,
 * assemble_range_shape, assemble_stream_count, assemble_static_count, rowgroup_*, rowone*, rowblock_*, shortrow_*, sweep_*,
 * lazy_structures ...).  No key changes a result beyond the documented tolerance of the kernel it selects; the assembly keys
 * change no bit.  value INT32_MIN: back to the built-in default. */
int ibh_set_tuning(const char *key, int value);
/* DIAGNOSTIC measurement hooks (bench.py only; not for product code: ibh_set_launch_events is thread-local
 * one-shot state that changes which launch API the calling thread's next apply uses): HIP events owned by
 * the library, and a one-shot request to attach a pair of them to the NEXT SpMM launch of the calling thread (hipExtLaunchKernel: start = the
 * kernel begins, stop = it ends -- the kernel's own duration, as rocprofv3's kernel trace reports it,
 * without the host's submission latency in front).  ibh_event_elapsed_ms waits for `stop`. */
int ibh_event_create(void **out);
int ibh_event_destroy(void *ev);
int ibh_event_elapsed_ms(void *start, void *stop, float *ms);
int ibh_set_launch_events(void *start, void *stop);
/* Diagnostic: run the assembly's ordering primitive (order analysis + independent-piece LDS sort,
 * falling back to the device-wide radix sort) on host keys with payload 0..n-1 and return the
 * resulting permutation, which must equal a stable sort by key.  key = (hi field << 32) | lo field,
 * fields below 2^hi_bits / 2^lo_bits.  *path_out: 0 already ordered, 1 pieces sorted in LDS,
 * 2 device-wide radix sort.  Used by tests/test_gpu_parity.py. */
int ibh_selftest_sort(const uint64_t *keys, int64_t n, int lo_bits, int hi_bits, uint32_t *perm_out, int *path_out);
/* Return the build workspaces of ALL host threads (the caller's and the library's worker threads') and
 * all cached device blocks to the driver (the library keeps freed device memory for reuse: a coupler
 * rebuilds the same matrices every step).  Call it while no build is in flight. */
int ibh_release_cached_memory(void);

#ifdef __cplusplus
}
#endif
#endif
