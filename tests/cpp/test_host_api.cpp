// C++ host-API test: uses icebin_amd/host/icebin_hip.hpp the way the reference's callers use
// GCMRegridder / RegridMatrices_Dynamic / Weighted_Eigen (modele/merge_topo.cpp:36-70,
// IceCoupler.cpp:361-468) and checks the invariants of tests/test_conserv/test_conserv.py.
// Exit code 0 = pass, 3 = no GPU (the no-fallback error path was verified instead).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../icebin_amd/host/icebin_hip.hpp"

using namespace icebin;
static const double NaN = std::nan("");

#define REQUIRE(cond)                                                        \
    do {                                                                     \
        if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    // ---- a 12 x 10 ice grid (cell 1 x 1) under a 3 x 2 atmosphere block (cell 4.5 x 5.5), nA = 8 x 4
    const int nx = 12, ny = 10, im = 8, i0 = 2, j0 = 1;
    const double Wx = 4.5, Wy = 5.5;
    ExchangeGrid exgrid;
    for (int ax = 0; ax < 3; ++ax) for (int ay = 0; ay < 2; ++ay)
        for (int ix = 0; ix < nx; ++ix) for (int iy = 0; iy < ny; ++iy) {
            double ox = std::fmin(ix + 1.0, (ax + 1) * Wx) - std::fmax((double)ix, ax * Wx);
            double oy = std::fmin(iy + 1.0, (ay + 1) * Wy) - std::fmax((double)iy, ay * Wy);
            if (ox > 0 && oy > 0) exgrid.add({{(j0 + ay) * im + (i0 + ax), ix * ny + iy}}, ox * oy);
        }
    AbbrGrid agridA;
    agridA.sparse_extent = 32;
    for (int ay = 0; ay < 2; ++ay) for (int ax = 0; ax < 3; ++ax) {
        agridA.dim_to_sparse.push_back((j0 + ay) * im + (i0 + ax));
        agridA.native_area.push_back(Wx * Wy * (1.0 + 0.02 * ax));
    }
    std::vector<double> proj_area(6, Wx * Wy);
    std::vector<double> hcdefs;
    for (int k = 0; k < 40; ++k) hcdefs.push_back(k * 100.0 - 50.0);     // write_icebin_in_base.py:44

    GCMRegridder_Standard gcm;
    gcm.init(std::move(agridA), std::move(hcdefs), true);
    REQUIRE(gcm.nA() == 32 && gcm.nhc() == 40 && gcm.nE() == 32 * 40);

    try {
        gcm.add_sheet("greenland", nx * ny, exgrid, proj_area, InterpStyle::Z_INTERP);
    } catch (Exception const &e) {
        if (e.code == IBH_ENODEVICE) { std::printf("no GPU: got the expected exception: %s\n", e.what()); return 3; }
        std::printf("unexpected exception: %s\n", e.what());
        return 1;
    }

    std::vector<double> elevmaskI(nx * ny);
    for (int ix = 0; ix < nx; ++ix) for (int iy = 0; iy < ny; ++iy)
        elevmaskI[ix * ny + iy] = (ix == 0 || iy == ny - 1) ? NaN : 100.0 * ix + 37.0 * iy;   // one masked edge row/col

    // ---- get_sheet_elevO, modele/merge_topo.cpp:36-70
    RegridParams paramsO(true, true, {{0., 0., 0.}});
    SparseSetT dimO, dimI(SparseSetT::identity(nx * ny));
    std::unique_ptr<RegridMatrices_Dynamic> rmO(gcm.regrid_matrices(0, elevmaskI, paramsO));
    auto OvI = rmO->matrix_d("AvI", {{&dimO, &dimI}}, paramsO);
    auto elevO = OvI->apply(ArrayView<const double>(elevmaskI.data(), (long)elevmaskI.size()), NaN, false);
    REQUIRE((int)elevO.size() == dimO.dense_extent() && dimO.dense_extent() == 6);
    for (double v : elevO) REQUIRE(std::isfinite(v) && v > 0);       // NaNs sit in empty columns only
    REQUIRE(OvI->conservative && OvI->scaled);

    // ---- the Python/Cython entry points (icebin_cython.cpp:195-198, 215-236)
    std::unique_ptr<RegridMatrices> rm(cython::new_regrid_matrices(&gcm, "greenland", elevmaskI.data(),
                                                                   (long)elevmaskI.size(), true, false, 0, 0, 0, true));
    std::unique_ptr<linear::Weighted> AvI(cython::RegridMatrices_matrix(rm.get(), "AvI"));
    std::unique_ptr<linear::Weighted> IvA(cython::RegridMatrices_matrix(rm.get(), "IvA"));
    std::unique_ptr<linear::Weighted> EvI(cython::RegridMatrices_matrix(rm.get(), "EvI"));
    std::unique_ptr<linear::Weighted> IvE(cython::RegridMatrices_matrix(rm.get(), "IvE"));
    REQUIRE(AvI->shape()[0] == 32 && AvI->shape()[1] == nx * ny);

    // test_constant_regrid (test_conserv.py:132-172): 1 survives I -> A -> I and I -> E -> I
    std::vector<double> onesI((size_t)AvI->shape_d()[1], 1.0);
    auto valA = AvI->apply(onesI, NaN, true);
    auto valIAI = IvA->apply(valA, NaN, true);
    for (double v : valIAI) REQUIRE(std::fabs(v - 1.0) < 1e-12);
    auto valE = EvI->apply(onesI);
    auto valIEI = IvE->apply(valE);
    for (double v : valIEI) REQUIRE(std::fabs(v - 1.0) < 1e-12);

    // test_conserv (test_conserv.py:174-205): sum(x * Mw) == sum(Mx * wM)
    std::vector<double> ramp(onesI.size());
    auto dimI2 = AvI->dim_to_sparse(1);
    for (size_t j = 0; j < ramp.size(); ++j) ramp[j] = (double)(dimI2[j] / ny + dimI2[j] % ny);
    auto y = AvI->apply(ramp);
    auto wM = AvI->wM(), Mw = AvI->Mw();
    double lhs = 0, rhs = 0;
    for (size_t j = 0; j < ramp.size(); ++j) lhs += Mw[j] * ramp[j];
    for (size_t i = 0; i < y.size(); ++i) rhs += wM[i] * y[i];
    REQUIRE(std::fabs(lhs / rhs - 1.0) < 1e-12);

    // scaled vs unscaled (test_conserv.py:89-99): BvA == diag(wM) * BvA_scaled
    auto unscaled = rmO->matrix_d("AvI", {{nullptr, nullptr}}, RegridParams(false, false, {{0., 0., 0.}}));
    std::vector<int> r0, c0, r1, c1; std::vector<double> v0, v1;
    AvI->M_coo(r0, c0, v0); unscaled->M_coo(r1, c1, v1);
    REQUIRE(r0 == r1 && c0 == c1);
    for (size_t k = 0; k < v0.size(); ++k) REQUIRE(std::fabs(wM[(size_t)r0[k]] * v0[k] / v1[k] - 1.0) < 1e-14);

    // the coupler's read-out loop (IceCoupler.cpp:449-457): dims[0]->to_sparse(jj) and wM(jj) per row, O(1) each
    {
        auto const &rows = dimO.to_sparse_all();
        REQUIRE((int)rows.size() == dimO.dense_extent());
        double s = 0;
        for (int jj = 0; jj < dimO.dense_extent(); ++jj) {
            REQUIRE(dimO.to_sparse(jj) == rows[(size_t)jj]);
            REQUIRE(dimO.to_dense(dimO.to_sparse(jj)) == jj && dimO.in_sparse(rows[(size_t)jj]));
            s += OvI->wM(jj);
        }
        REQUIRE(s > 0 && !dimO.in_sparse(0));
        // SparseSet::add_dense on the host: an existing key keeps its id, a new one gets the next
        SparseSetT extra(32);
        REQUIRE(extra.add_dense(7) == 0 && extra.add_dense(3) == 1 && extra.add_dense(7) == 0 && extra.dense_extent() == 2);
        REQUIRE(extra.to_sparse(1) == 3);
        // a set pre-populated that way is appended to by a build (IceCoupler.cpp:366-371 shares dims between builds)
        SparseSetT dimA2(32);
        dimA2.add_dense(rows[2]);
        auto M2 = rmO->matrix_d("AvI", {{&dimA2, &dimI}}, paramsO);
        REQUIRE(dimA2.dense_extent() == 6 && dimA2.to_sparse(0) == rows[2] && dimA2.to_dense(rows[2]) == 0);
        // ->M (IceCoupler.cpp:408 iterates the entries)
        REQUIRE(M2->M.rows() == 6 && M2->M.cols() == nx * ny && M2->M.nonZeros() == M2->nnz());
        auto trip = OvI->M.triplets();
        REQUIRE((long)trip.size() == OvI->nnz() && trip.front().row == 0 && std::isfinite(trip.front().value));
    }

    // one coupling step's matrices in one call (IceCoupler.cpp:361-468), then E1vE0 against an unchanged mask (e1ve0.cpp:55-106)
    {
        SparseSetT dI(SparseSetT::identity(nx * ny)), dX(SparseSetT::identity((long)exgrid.overlaps.size())), dE((long)gcm.nE()), dA;
        auto step = rmO->matrix_batch({"EvI", "AvI", "IvE", "XvE"},
                                      {{{&dE, &dI}}, {{&dA, &dI}}, {{&dI, &dE}}, {{&dX, &dE}}},
                                      {RegridParams(false, false, {{0., 0., 0.}}), RegridParams(false, true, {{0., 0., 0.}}),
                                       RegridParams(true, true, {{0., 0., 0.}}), RegridParams(false, true, {{0., 0., 0.}})});
        REQUIRE(step.size() == 4 && step[0]->shape_d()[0] == dE.dense_extent() && step[2]->shape_d()[1] == dE.dense_extent());
        REQUIRE(step[1]->shape_d()[0] == dA.dense_extent() && dA.dense_extent() == 6);
        auto one = rmO->matrix_d("XvE", {{&dX, &dE}}, RegridParams(false, true, {{0., 0., 0.}}));
        REQUIRE(one->nnz() == step[3]->nnz());
        auto E1vE0 = e1ve0::compute_E1vE0c({step[3].get()}, {one.get()}, gcm.nE());
        REQUIRE(E1vE0->shape()[0] == (long)gcm.nE() && E1vE0->nnz() > 0);
        for (auto const &t : E1vE0->M.triplets()) REQUIRE(t.value == 0.0);          // E1 == E0: no correction
        // the exchange grid regenerated on the device (gridgen/GridGen_Exchange.cpp:175-284) equals the hand-built one
        std::vector<double> xe, ye, vx, vy; std::vector<int> pp{0}; std::vector<long> ia;
        for (int k = 0; k <= nx; ++k) xe.push_back(k);
        for (int k = 0; k <= ny; ++k) ye.push_back(k);
        for (int ay = 0; ay < 2; ++ay) for (int ax = 0; ax < 3; ++ax) {
            double x0 = ax * Wx, x1 = (ax + 1) * Wx, y0 = ay * Wy, y1 = (ay + 1) * Wy;
            vx.insert(vx.end(), {x0, x1, x1, x0}); vy.insert(vy.end(), {y0, y0, y1, y1});
            pp.push_back((int)vx.size()); ia.push_back((j0 + ay) * im + (i0 + ax));
        }
        ExchangeGrid gen = make_exchange_grid(xe, ye, false, pp, vx, vy, ia);
        REQUIRE(gen.overlaps.size() == exgrid.overlaps.size());
        double tot = 0, tot0 = 0;
        for (double v : gen.overlaps) tot += v;
        for (double v : exgrid.overlaps) tot0 += v;
        REQUIRE(std::fabs(tot - tot0) < 1e-12 * tot0);
    }

    // on-disk formats from C++ (GCMRegridder.cpp:104-150, modele/global_ec.cpp:539-605): write the regridder and two matrices the
    // way global_ec appends them, read everything back into fresh objects, rebuild and compare bitwise
    {
        const std::string fn = argc > 1 ? std::string(argv[1]) : std::string("/tmp/icebin_hip_host_api.nc");
        { NcIO ncio(fn, 'w'); gcm.ncio(ncio, "m"); }
        { NcIO ncio(fn, 'a'); AvI->ncio(ncio, "AvI", {{"dimA", "dimI"}}); }
        { NcIO ncio(fn, 'a'); IvA->ncio(ncio, "IvA", {{"dimI", "dimA"}}); }      // shares dimA / dimI with AvI: written once
        NcIO rd(fn, 'r');
        GCMRegridder_Standard gcm2;
        gcm2.ncio(rd, "m");
        REQUIRE(gcm2.nA() == gcm.nA() && gcm2.nhc() == gcm.nhc() && gcm2.correctA == gcm.correctA && gcm2.nI(0) == gcm.nI(0));
        std::unique_ptr<RegridMatrices_Dynamic> rm2(gcm2.regrid_matrices((int)gcm2.sheet_index("greenland"), elevmaskI, RegridParams(true, false, {{0., 0., 0.}})));
        auto AvI2 = rm2->matrix("AvI");
        std::vector<int> ra, ca, rb, cb; std::vector<double> va, vb;
        AvI->M_coo(ra, ca, va); AvI2->M_coo(rb, cb, vb);
        REQUIRE(ra == rb && ca == cb && va == vb && AvI2->wM() == AvI->wM() && AvI2->Mw() == AvI->Mw());
        linear::Weighted L;
        L.ncio(rd, "AvI");
        L.M_coo(rb, cb, vb);
        REQUIRE(ra == rb && ca == cb && va == vb && L.wM() == AvI->wM() && L.Mw() == AvI->Mw());
        REQUIRE(L.shape() == AvI->shape() && L.dim_to_sparse(0) == AvI->dim_to_sparse(0) && L.dim_to_sparse(1) == AvI->dim_to_sparse(1));
        REQUIRE(L.conservative == AvI->conservative && L.scaled == AvI->scaled);
        REQUIRE(L.apply(ramp) == AvI->apply(ramp));
        linear::Weighted L2;
        L2.ncio(rd, "IvA");
        REQUIRE(L2.shape() == IvA->shape() && L2.nnz() == IvA->nnz());
        SparseSetT dA3;
        dA3.ncio(rd, "dimA");
        REQUIRE(dA3.dense_extent() == AvI->shape_d()[0] && dA3.to_sparse(0) == AvI->dim_to_sparse(0)[0]);
    }

    // errors surface as exceptions (error.hpp:28-32)
    bool threw = false;
    try { rm->matrix("BvA"); } catch (Exception const &e) { threw = e.code == IBH_ENOKEY; }
    REQUIRE(threw);
    threw = false;
    std::vector<double> shortmask(5, 0.0);
    try { gcm.regrid_matrices(0, shortmask); } catch (Exception const &e) { threw = e.code == IBH_EINVAL; }
    REQUIRE(threw);
    std::printf("C++ host API: all checks passed (AvI %d x %d, nnz %ld)\n", AvI->shape_d()[0], AvI->shape_d()[1], AvI->nnz());
    return 0;
}
