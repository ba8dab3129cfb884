// assemble.h -- entry points of the assembly path (assemble.hip).
#pragma once
#include <memory>

#include "common.h"
#include "prims.h"

namespace ibh {
// RegridMatrices_Dynamic::matrix_d (RegridMatrices_Dynamic.cpp:412-423) on device.
// comm: the sharded build (streamasm.inl) -- every rank of the communicator makes this call with the same arguments and gets the
// whole matrix; the ranks share the work when the build is one the streamed path serves, and build redundantly otherwise
bool assemble_matrix(const ibh_regrid_matrices *rm, const char *spec_name, ibh_sparse_set *dim0, ibh_sparse_set *dim1,
                     int scale, int correctA, const double sigma[3], ibh_weighted **out, bool fast_only = false, ibh_comm *comm = nullptr);
// the per-ice-cell class bytes of an elevmask (ibh_regrid_matrices::em_cls), made on `stream` when the streamed build will serve this grid
// (returns false otherwise); d_src != nullptr: read the elevations there and fill rm->elevmaskI (allocated) in the same pass
bool elevmask_classes(const ibh_regrid_matrices *rm, hipStream_t stream, const double *d_src = nullptr);
// n builds with the results of n matrix_d calls in order; independent ones run concurrently (assemble.hip assemble_batch)
void assemble_batch(const ibh_regrid_matrices *rm, int n, const char *const *specs, ibh_sparse_set *const *dim0,
                    ibh_sparse_set *const *dim1, const int32_t *scale, const int32_t *correctA, const double sigma[3],
                    ibh_weighted **out);
// Eigen setFromTriplets (to_eigen_M, eigen_types.cpp:9-34) on device: fills w's CSR from host COO.
void weighted_from_coo_device(ibh_weighted *w, int nrow, int ncol, int64_t n, const int32_t *row, const int32_t *col,
                              const double *val);
// compute_E1vE0c (e1ve0.cpp:55-106) on device; `out` comes with identity dims over nE.
void e1ve0_compute(int nsheets, const ibh_weighted *const *XuE1s, const ibh_weighted *const *XuE0s, int64_t nE, ibh_weighted *out);
// make_exchange_grid (gridgen/GridGen_Exchange.cpp:175-284) for a rectilinear XY ice grid (gridgen.hip)
void exgrid_generate(const ibh_exgrid_desc *d, ibh_exgrid *out);
}  // namespace ibh
