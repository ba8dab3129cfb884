// assemble.h -- entry points of the assembly path (assemble.hip).
#pragma once
#include <memory>

#include "common.h"
#include "prims.h"

namespace ibh {
// RegridMatrices_Dynamic::matrix_d (RegridMatrices_Dynamic.cpp:412-423) on device.
void assemble_matrix(const ibh_regrid_matrices *rm, const char *spec_name, ibh_sparse_set *dim0, ibh_sparse_set *dim1,
                     int scale, int correctA, const double sigma[3], ibh_weighted **out);
// Eigen setFromTriplets (to_eigen_M, eigen_types.cpp:9-34) on device: fills w's CSR from host COO.
void weighted_from_coo_device(ibh_weighted *w, int nrow, int ncol, int64_t n, const int32_t *row, const int32_t *col,
                              const double *val);
}  // namespace ibh
