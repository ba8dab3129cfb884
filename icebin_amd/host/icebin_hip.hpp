// icebin_hip.hpp -- host-side C++ mirror of IceBin's regrid-matrix interface over the C-ABI.
//
// Same class names, method names, argument order and defaults as the reference, so a caller of
//   GCMRegridder_Standard::regrid_matrices()  (slib/icebin/GCMRegridder.hpp:290-293, 369-372)
//   RegridMatrices_Dynamic::matrix_d()/matrix() (slib/icebin/RegridMatrices_Dynamic.hpp:51-59)
//   linear::Weighted_Eigen::apply()           (call sites modele/merge_topo.cpp:65, icebin22m.cpp:153)
// can switch to this header and link libicebin_hip.so instead of ibmisc/spsparse/Eigen.
// Differences, all forced by the absent third-party types:
//   - blitz::Array<double,N>  -> icebin::ArrayView<double> (pointer + extents, row-major, borrowed)
//     for inputs and std::vector<double> for results;
//   - Eigen matrix `M`        -> device CSR owned by the handle; host copies via M_coo();
//   - errors: (*icebin_error)(-1, ...) -> icebin::Exception (an std::runtime_error), like
//     everytrace::Exception (slib/icebin/error.hpp:28).
// Header-only; C++14.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/icebin_hip.h"
#include "ncio.hpp"

namespace icebin {

struct Exception : std::runtime_error {
    int code;
    Exception(int c, std::string const &msg) : std::runtime_error(msg), code(c) {}
};
inline void check(int rc) {
    if (rc != IBH_OK) throw Exception(rc, ibh_last_error());
}

/** Borrowed row-major view: stands in for blitz::Array<T,1> / <T,2> arguments. */
template <class T>
struct ArrayView {
    T *data;
    long extent0, extent1;     // extent1 == 1 for rank-1 views
    int rank;
    ArrayView() : data(nullptr), extent0(0), extent1(1), rank(1) {}
    ArrayView(T *p, long n) : data(p), extent0(n), extent1(1), rank(1) {}
    ArrayView(T *p, long n0, long n1) : data(p), extent0(n0), extent1(n1), rank(2) {}
    ArrayView(std::vector<typename std::remove_const<T>::type> const &v)
        : data(const_cast<T *>(v.data())), extent0((long)v.size()), extent1(1), rank(1) {}
    long size() const { return extent0 * extent1; }
};

// ---- eigen_types.hpp:16-24 -------------------------------------------------------------------
typedef long sparse_index_type;
typedef int dense_index_type;
typedef double val_type;

/** spsparse::SparseSet<long,int> (eigen_types.hpp:24). */
class SparseSetT {
    ibh_sparse_set *h_;
    SparseSetT(ibh_sparse_set *h) : h_(h) {}
public:
    SparseSetT() : h_(nullptr) { check(ibh_sparse_set_create(-1, &h_)); }
    explicit SparseSetT(long sparse_extent) : h_(nullptr) { check(ibh_sparse_set_create(sparse_extent, &h_)); }
    SparseSetT(SparseSetT const &) = delete;
    SparseSetT &operator=(SparseSetT const &) = delete;
    SparseSetT(SparseSetT &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~SparseSetT() { if (h_) ibh_sparse_set_destroy(h_); }
    /** ibmisc id_sparse_set<SparseSetT>(n) (modele/merge_topo.cpp:48, IceCoupler.cpp:366) */
    static SparseSetT identity(long n) { ibh_sparse_set *h; check(ibh_sparse_set_create_identity(n, &h)); return SparseSetT(h); }
    long sparse_extent() const { int64_t v; check(ibh_sparse_set_sparse_extent(h_, &v)); return (long)v; }
    int dense_extent() const { int32_t v; check(ibh_sparse_set_dense_extent(h_, &v)); return v; }
    /** The whole dense -> sparse table.  Cached on the host: a set only ever grows (add_dense, or a matrix
        build appending on the device), so the cache is refreshed when the dense extent has changed and
        to_sparse(id) is O(1) in the loop shape of IceCoupler.cpp:449-457. */
    std::vector<long> const &to_sparse_all() const {
        const int n = dense_extent();
        if ((int)cache_.size() != n) {
            std::vector<int64_t> t((size_t)n);
            check(ibh_sparse_set_to_sparse(h_, t.data()));
            cache_.assign(t.begin(), t.end());
        }
        return cache_;
    }
    long to_sparse(int id) const { return to_sparse_all().at((size_t)id); }
    /** spsparse::SparseSet::{add_dense, to_dense, in_sparse} (AbbrGrid.cpp:108, IceCoupler.cpp:298) */
    int add_dense(long sparse) { int32_t d; check(ibh_sparse_set_add_dense(h_, sparse, &d)); return d; }
    bool in_sparse(long sparse) const { int32_t d; check(ibh_sparse_set_to_dense(h_, sparse, &d)); return d >= 0; }
    int to_dense(long sparse) const {
        int32_t d; check(ibh_sparse_set_to_dense(h_, sparse, &d));
        if (d < 0) throw Exception(IBH_ENOKEY, "SparseSet: sparse index " + std::to_string(sparse) + " is not in the set");
        return d;
    }
    ibh_sparse_set *handle() const { return h_; }
    /** SparseSet::ncio (matrix_formats.rst:23-27): `int64 <vname>(<vname>.dense_extent)` with attribute `sparse_extent`.
        Reading replaces the contents of this set (it must not be in use by a matrix). */
    void ncio(NcIO &ncio, std::string const &vname) {
        if (ncio.reading()) {
            nc::Var const &v = ncio.file.var(vname);
            std::vector<int64_t> t = v.data.as<int64_t>();
            ibh_sparse_set *h = nullptr;
            check(ibh_sparse_set_from_array(v.att("sparse_extent").at<int64_t>(0), t.data(), (int32_t)t.size(), &h));
            if (h_) ibh_sparse_set_destroy(h_);
            h_ = h; cache_.clear();
        } else {
            if (ncio.file.has_var(vname)) return;            // dims shared by several matrices are written once (IceCoupler.cpp:479-487)
            std::vector<long> const &t = to_sparse_all();
            std::vector<int64_t> t64(t.begin(), t.end());
            const std::string d = ncio.file.add_dim(vname + ".dense_extent", (int64_t)t64.size());
            const int64_t ext = sparse_extent();
            ncio.file.add_var(vname, {d}, nc::Array::of(t64), {{"sparse_extent", nc::Array::of(&ext, 1)}});
            ncio.touch();
        }
    }
private:
    mutable std::vector<long> cache_;
};

// ---- RegridMatrices.hpp:17-37 ----------------------------------------------------------------
struct RegridParams {
    bool scale;
    bool correctA;
    std::array<double, 3> sigma;
    bool smooth() const { return sigma[0] != 0; }
    RegridParams() : scale(true), correctA(false), sigma({0., 0., 0.}) {}
    RegridParams(bool _scale, bool _correctA, std::array<double, 3> const &_sigma)
        : scale(_scale), correctA(_correctA), sigma(_sigma) {}
};

/** The ranks of a field-sharded regrid, one process per GPU (ibh_comm; RCCL over xGMI underneath).  Rank 0 calls
    Communicator::unique_id() and ships the 128 bytes to the others with whatever the host program has (ModelE: MPI_Bcast);
    then every rank constructs Communicator(world, rank, id) on its own device.  world == 1 needs neither id nor RCCL. */
class Communicator {
    ibh_comm *h_ = nullptr;
public:
    static std::array<char, IBH_UNIQUE_ID_BYTES> unique_id() {
        std::array<char, IBH_UNIQUE_ID_BYTES> id;
        check(ibh_comm_unique_id(id.data()));
        return id;
    }
    Communicator() { check(ibh_comm_create(1, 0, nullptr, &h_)); }
    Communicator(int world, int rank, std::array<char, IBH_UNIQUE_ID_BYTES> const &id) { check(ibh_comm_create(world, rank, id.data(), &h_)); }
    Communicator(int world, int rank, ibh_exchange_fn fn, void *user) { check(ibh_comm_create_custom(world, rank, fn, user, &h_)); }
    Communicator(Communicator const &) = delete;
    Communicator &operator=(Communicator const &) = delete;
    ~Communicator() { if (h_) ibh_comm_destroy(h_); }
    int world() const { int w; check(ibh_comm_info(h_, &w, nullptr)); return w; }
    int rank() const { int r; check(ibh_comm_info(h_, nullptr, &r)); return r; }
    /** make `stream` wait for the exchanges enqueued so far: the gathered results are complete behind it */
    void wait(void *stream) const { check(ibh_comm_wait(h_, stream)); }
    /** run the exchanges on the caller's stream (which outlives this object and is never destroyed by it) */
    void set_stream(void *stream) { check(ibh_comm_set_stream(h_, stream)); }
    /** "planes_padded": the gap between nrow_d and ldb of the result planes is padding the caller owns */
    void set_option(const char *key, int value) { check(ibh_comm_set_option(h_, key, value)); }
    ibh_comm *handle() const { return h_; }
};

namespace linear {
/** ibmisc::linear::Weighted / Weighted_Eigen: M plus wM, Mw, dims, conservative, scaled. */
class Weighted {
    ibh_weighted *h_;
    mutable std::vector<double> wM_, Mw_;
    mutable bool have_wM_ = false, have_Mw_ = false;
    bool file_dims_ = false;                       // loaded by ncio(): the dims are those of the file (the handle's are identities)
    std::array<std::vector<long>, 2> file_dim_;
    std::array<long, 2> file_extent_ = {{0, 0}};
public:
    bool conservative = true, scaled = true;
    /** an empty matrix, to be filled by ncio() from a file */
    Weighted() : h_(nullptr) {}
    explicit Weighted(ibh_weighted *h) : h_(h) {
        int c, s;
        check(ibh_weighted_flags(h, &c, &s));
        conservative = c != 0; scaled = s != 0;
    }
    Weighted(Weighted const &) = delete;
    Weighted &operator=(Weighted const &) = delete;
    virtual ~Weighted() { if (h_) ibh_weighted_destroy(h_); }
    ibh_weighted *handle() const { return h_; }

    /** Dense shape {nrow_d, ncol_d} and number of stored entries. */
    std::array<int, 2> shape_d() const { int32_t r, c; check(ibh_weighted_shape(h_, &r, &c, nullptr)); return {{r, c}}; }
    long nnz() const { int64_t n; check(ibh_weighted_shape(h_, nullptr, nullptr, &n)); return (long)n; }
    /** Sparse shape (dims[k]->sparse_extent()). */
    std::array<long, 2> shape() const {
        if (file_dims_) return {{file_extent_[0], file_extent_[1]}};
        int64_t a, b;
        check(ibh_weighted_dim(h_, 0, &a, nullptr)); check(ibh_weighted_dim(h_, 1, &b, nullptr));
        return {{(long)a, (long)b}};
    }
    /** dims[k]->to_sparse(j) for all j */
    std::vector<long> dim_to_sparse(int k) const {
        if (file_dims_) return file_dim_[(size_t)k];
        std::vector<int64_t> t((size_t)shape_d()[(size_t)k]);
        check(ibh_weighted_dim_to_sparse(h_, k, t.data()));
        return std::vector<long>(t.begin(), t.end());
    }
    /** Weighted_Eigen::ncio(ncio, vname, {dimB, dimA}) (modele/global_ec.cpp:571-605, IceCoupler.cpp:473-488): the "Eigen
        format" of sphinx/source/matrix_formats.rst:9-63 -- the dims under their own names (written once when several matrices
        share them), `<v>.info` {type "EIGEN", conservative, scaled, dim_names}, `<v>.M.info` {shape, conservative},
        `<v>.M.indices(nnz, rank)`, `<v>.M.values`, `<v>.Mw`, `<v>.wM`.  With a reading NcIO the matrix is LOADED from the file
        (to_eigen_M, eigen_types.cpp:9-34: ibh_weighted_from_coo) and keeps the file's dims. */
    void ncio(NcIO &ncio, std::string const &vname, std::array<std::string, 2> const &dim_names = {{"", ""}}) {
        if (ncio.reading()) {
            nc::File const &f = ncio.file;
            nc::Var const &info = f.var(vname + ".info");
            if (nc::find(info.attrs, "type") && info.att("type").str() != "EIGEN")
                throw Exception(IBH_ENOTIMPL, vname + ": matrix type '" + info.att("type").str() + "' is not the Eigen format");
            std::vector<std::string> names = nc::split_names(info.att("dim_names").str());
            if (names.size() != 2) throw Exception(IBH_EINVAL, vname + ".info:dim_names must name two dims");
            for (int k = 0; k < 2; ++k) {
                std::string n = names[(size_t)k];
                if (n.compare(0, vname.size() + 1, vname + ".") == 0) n = n.substr(vname.size() + 1);
                nc::Var const &d = f.var(n);
                std::vector<int64_t> t = d.data.as<int64_t>();
                file_dim_[(size_t)k].assign(t.begin(), t.end());
                file_extent_[(size_t)k] = (long)d.att("sparse_extent").at<int64_t>(0);
            }
            std::vector<int32_t> ind = f.var(vname + ".M.indices").data.as<int32_t>();
            std::vector<double> val = f.var(vname + ".M.values").data.as<double>();
            std::vector<double> wMv = f.var(vname + ".wM").data.as<double>(), Mwv = f.var(vname + ".Mw").data.as<double>();
            std::vector<int32_t> row(val.size()), col(val.size());
            for (size_t e = 0; e < val.size(); ++e) { row[e] = ind[2 * e]; col[e] = ind[2 * e + 1]; }
            const int cons = nc::find(info.attrs, "conservative") ? info.att("conservative").at<int>(0) : 1;
            const int sc = nc::find(info.attrs, "scaled") ? info.att("scaled").at<int>(0) : 1;
            ibh_weighted *h = nullptr;
            check(ibh_weighted_from_coo((int32_t)wMv.size(), (int32_t)Mwv.size(), (int64_t)val.size(), row.data(), col.data(), val.data(),
                                        wMv.data(), Mwv.data(), cons, sc, &h));
            if (h_) ibh_weighted_destroy(h_);
            h_ = h; conservative = cons != 0; scaled = sc != 0; file_dims_ = true; have_wM_ = have_Mw_ = false;
            return;
        }
        if (dim_names[0].empty() || dim_names[1].empty()) throw Exception(IBH_EINVAL, "Weighted::ncio: writing needs the two dim names");
        nc::File &f = ncio.file;
        std::array<std::string, 2> dnames;
        for (int k = 0; k < 2; ++k) {
            std::vector<long> t = dim_to_sparse(k);
            std::vector<int64_t> t64(t.begin(), t.end());
            dnames[(size_t)k] = f.add_dim(dim_names[(size_t)k] + ".dense_extent", (int64_t)t64.size());
            if (!f.has_var(dim_names[(size_t)k])) {
                const int64_t ext = shape()[(size_t)k];
                f.add_var(dim_names[(size_t)k], {dnames[(size_t)k]}, nc::Array::of(t64), {{"sparse_extent", nc::Array::of(&ext, 1)}});
            }
        }
        std::vector<int> row, col; std::vector<double> val;
        M_coo(row, col, val);
        const int32_t zero32 = 0, cons = conservative ? 1 : 0, sc = scaled ? 1 : 0;
        const int64_t zero64 = 0;
        f.add_var(vname + ".info", {}, nc::Array::of(&zero32, 1),
                  {{"type", nc::Array::str("EIGEN")}, {"conservative", nc::Array::of(&cons, 1)}, {"scaled", nc::Array::of(&sc, 1)},
                   {"dim_names", nc::Array::str(vname + "." + dim_names[0] + "," + vname + "." + dim_names[1])}});
        const int64_t shp[2] = {(int64_t)shape_d()[0], (int64_t)shape_d()[1]};
        f.add_var(vname + ".M.info", {}, nc::Array::of(&zero64, 1),
                  {{"shape", nc::Array::of(shp, 2)}, {"conservative", nc::Array::str(conservative ? "t" : "f")}});
        const std::string dn = f.add_dim(vname + ".M.nnz", (int64_t)val.size()), dr = f.add_dim(vname + ".M.rank", 2);
        std::vector<int32_t> ind(2 * val.size());
        for (size_t e = 0; e < val.size(); ++e) { ind[2 * e] = row[e]; ind[2 * e + 1] = col[e]; }
        f.add_var(vname + ".M.indices", {dn, dr}, nc::Array::of(ind));
        f.add_var(vname + ".M.values", {dn}, nc::Array::of(val));
        f.add_var(vname + ".Mw", {dnames[1]}, nc::Array::of(Mw()));
        f.add_var(vname + ".wM", {dnames[0]}, nc::Array::of(wM()));
        ncio.touch();
    }
    /** wM / Mw: host mirrors, downloaded once (the matrix is immutable after its build); wM(jj) as at
        IceCoupler.cpp:456. */
    std::vector<double> const &wM() const {
        if (!have_wM_) { wM_.resize((size_t)shape_d()[0]); check(ibh_weighted_get_wM(h_, wM_.data())); have_wM_ = true; }
        return wM_;
    }
    std::vector<double> const &Mw() const {
        if (!have_Mw_) { Mw_.resize((size_t)shape_d()[1]); check(ibh_weighted_get_Mw(h_, Mw_.data())); have_Mw_ = true; }
        return Mw_;
    }
    double wM(int jj) const { return wM().at((size_t)jj); }
    double Mw(int jj) const { return Mw().at((size_t)jj); }
    /** ->M as row-major COO in dense index space */
    void M_coo(std::vector<int> &row, std::vector<int> &col, std::vector<double> &val) const {
        size_t n = (size_t)nnz();
        row.resize(n); col.resize(n); val.resize(n);
        check(ibh_weighted_get_coo(h_, row.data(), col.data(), val.data()));
    }
    /** The `M` member callers poke (IceCoupler.cpp:408 iterates its entries, :445 multiplies with it): the
        matrix lives in HBM; this view gives its shape and, on demand, a host copy of the triplets. */
    struct MatrixView {
        Weighted const *w;
        int rows() const { return w->shape_d()[0]; }
        int cols() const { return w->shape_d()[1]; }
        long nonZeros() const { return w->nnz(); }
        struct Triplet { int row, col; double value; };
        std::vector<Triplet> triplets() const {
            std::vector<int> r, c; std::vector<double> v;
            w->M_coo(r, c, v);
            std::vector<Triplet> t(v.size());
            for (size_t k = 0; k < v.size(); ++k) t[k] = Triplet{r[k], c[k], v[k]};
            return t;
        }
    };
    MatrixView M{this};
    /** `tmp`: objects that must live as long as the matrix (RegridMatrices_Dynamic::matrix hands the dims it
        allocated to M->tmp, RegridMatrices_Dynamic.cpp:429-436). */
    std::vector<std::shared_ptr<void>> tmp;

    /** apply(A_b, fill, force_conservation, tmp): A_b is (nvar, ncol_d) or a rank-1 vector of
        ncol_d values; returns (nvar, nrow_d) row-major.  The reference's TmpAlloc argument owned
        the result; here the returned vector does. */
    std::vector<double> apply(ArrayView<const double> const &A_b, double fill = std::nan(""),
                              bool force_conservation = true) const {
        auto sh = shape_d();
        long nvar = A_b.rank == 1 ? 1 : A_b.extent0;
        long ncol = A_b.rank == 1 ? A_b.extent0 : A_b.extent1;
        if (ncol != sh[1]) throw Exception(IBH_EINVAL, "apply: input has " + std::to_string(ncol) +
                                                        " columns, matrix has " + std::to_string(sh[1]));
        std::vector<double> B((size_t)(nvar * sh[0]));
        check(ibh_weighted_apply_host(h_, A_b.data, (int32_t)nvar, ncol, B.data(), sh[0], fill, force_conservation ? 1 : 0));
        return B;
    }
    /** The coupler's fused product B = M * (A*T + b) (IceCoupler.cpp:203-252, :445) on HBM-resident
        fields: T is the (sparse) variable transform as a dense row-major nvar_in x nvar_out host array
        (exact zeros are structural), b the offsets. */
    void apply_transformed_device(const double *dA_b, int nvar_in, long lda, std::vector<double> const &T,
                                  std::vector<double> const &b, double *dB_b, long ldb, double fill, void *stream) const {
        check(ibh_weighted_apply_transformed_device(h_, dA_b, nvar_in, lda, T.data(), b.data(), (int32_t)b.size(),
                                                    dB_b, ldb, fill, stream));
    }
    /** Several field batches through one launch (ibh_weighted_apply_many_device): dA_b[q] -> dB_b[q], all nvar x lda / ldb. */
    void apply_many_device(std::vector<const double *> const &dA_b, int nvar, long lda, std::vector<double *> const &dB_b,
                           long ldb, double fill, bool force_conservation, void *stream) const {
        if (dA_b.size() != dB_b.size()) throw Exception(IBH_EINVAL, "apply_many_device: batch lists differ in length");
        check(ibh_weighted_apply_many_device(h_, (int32_t)dA_b.size(), dA_b.data(), nvar, lda, dB_b.data(), ldb, fill,
                                             force_conservation ? 1 : 0, stream));
    }
    /** The legacy COO product with its fill / ignore-NaN contract (coo_matvec, pylib/icebin_cython.cpp:158-192) on device arrays. */
    void matvec_device(const double *dxx, int nvar, long ldx, double *dyy, long ldy, bool ignore_nan, void *stream) const {
        check(ibh_weighted_matvec_device(h_, dxx, nvar, ldx, dyy, ldy, ignore_nan ? 1 : 0, stream));
    }
    /** Build now whatever apply structure (column sweep / bands) applies of up to nvar variables, nbatch batches per
        launch, would build lazily on their second call, and size all per-apply scratch: afterwards the device applies
        only enqueue work (ibh_weighted_prepare).  Once per matrix, e.g. right after matrix_d() in a coupler that
        applies the matrix more than once per step, and before capturing applies into a hipGraph. */
    void prepare(int nvar, int nbatch = 1) const { check(ibh_weighted_prepare(h_, nvar, nbatch)); }
    /** A launch option of THIS matrix (ibh_weighted_set_option): read before the process-wide ibh_set_tuning map. */
    void set_option(std::string const &key, int value) { check(ibh_weighted_set_option(h_, key.c_str(), value)); }
    /** Fused pair B1 = (*this) * A, B2 = second * B1 in one launch (EvI then AvE; ibh_weighted_pair_prepare /
        ibh_weighted_apply_pair_device).  pair_prepare throws Exception(IBH_ENOTIMPL) when the matrices do not pair: make the two
        apply() calls of the reference then. */
    void pair_prepare(Weighted const &second, int nvar = 16) const { check(ibh_weighted_pair_prepare(h_, second.h_, nvar)); }
    void apply_pair_device(Weighted const &second, const double *dA_b, int nvar, long lda, double *dB1_b, long ldb1, double *dB2_b,
                           long ldb2, double fill, void *stream) const {
        check(ibh_weighted_apply_pair_device(h_, second.h_, dA_b, nvar, lda, dB1_b, ldb1, dB2_b, ldb2, fill, stream));
    }
    /** the chain of BASELINE config 3 (EvI, AvE, IvA): the fused pair, then `third` on its result, the third product's launch
        overlapped with the pair kernel (ibh_weighted_apply_chain_device); bitwise the pair apply followed by third.apply_device */
    void apply_chain_device(Weighted const &second, Weighted const &third, const double *dA_b, int nvar, long lda, double *dB1_b, long ldb1,
                            double *dB2_b, long ldb2, double *dB3_b, long ldb3, double fill = std::nan(""), void *stream = nullptr) const {
        check(ibh_weighted_apply_chain_device(h_, second.h_, third.h_, dA_b, nvar, lda, dB1_b, ldb1, dB2_b, ldb2, dB3_b, ldb3, fill, stream));
    }
    /** apply() of world x nvar_local fields sharded by field over the ranks of `comm`: this rank's dA_local (nvar_local x lda)
        -> dB_all (world*nvar_local x ldb, the same on every rank once comm.wait(stream) has been honoured).  The SpMM runs on
        `stream`, the peer-to-peer exchange on the communicator's own stream, block_fields fields at a time (0: by size). */
    void apply_sharded_device(Communicator const &comm, const double *dA_local, int nvar_local, long lda, double *dB_all, long ldb,
                              double fill, void *stream, int block_fields = 0) const {
        check(ibh_weighted_apply_sharded_device(h_, comm.handle(), dA_local, nvar_local, lda, dB_all, ldb, fill, block_fields, stream));
    }
    /** several field batches: ONE SpMM launch, ONE grouped exchange (ibh_weighted_apply_many_sharded_device) */
    void apply_many_sharded_device(Communicator const &comm, std::vector<const double *> const &dA_local, int nvar_local, long lda,
                                   std::vector<double *> const &dB_all, long ldb, double fill, void *stream) const {
        if (dA_local.size() != dB_all.size()) throw Exception(IBH_EINVAL, "apply_many_sharded_device: batch lists differ in length");
        check(ibh_weighted_apply_many_sharded_device(h_, comm.handle(), (int32_t)dA_local.size(), dA_local.data(), nvar_local, lda,
                                                     dB_all.data(), ldb, fill, stream));
    }
    /** Device-resident variant: dA_b (nvar x lda) and dB_b (nvar x ldb) are HBM pointers; enqueues on stream. */
    void apply_device(const double *dA_b, int nvar, long lda, double *dB_b, long ldb, double fill,
                      bool force_conservation, void *stream) const {
        check(ibh_weighted_apply_device(h_, dA_b, nvar, lda, dB_b, ldb, fill, force_conservation ? 1 : 0, stream));
    }
};
typedef Weighted Weighted_Eigen;      // the concrete type callers name (RegridMatrices_Dynamic.hpp:27)
}   // namespace linear

// ---- AbbrGrid.hpp:40-89 ----------------------------------------------------------------------
class ExchangeGrid {
public:
    std::vector<int> indices;       // Length*2: (ixA, ixI)
    std::vector<double> overlaps;
    void reserve(size_t n) { indices.reserve(n * 2); overlaps.reserve(n); }
    void add(std::array<int, 2> const &index, double _area) {
        indices.push_back(index[0]); indices.push_back(index[1]); overlaps.push_back(_area);
    }
    int dense_extent() const { return (int)overlaps.size(); }
    long sparse_extent() const { return (long)overlaps.size(); }
    long to_sparse(int id) const { return id; }
    int ijk(int id, int index) const { return indices[(size_t)id * 2 + (size_t)index]; }
    double native_area(int id) const { return overlaps[(size_t)id]; }
};

/** The parts of AbbrGrid (AbbrGrid.hpp:93-109) the regrid path reads. */
struct AbbrGrid {
    long sparse_extent = 0;             // dim.sparse_extent()
    std::vector<long> dim_to_sparse;    // dim: dense -> sparse
    std::vector<double> native_area;    // dense indexing
    std::string name, sproj;
};

struct InterpStyle { enum { Z_INTERP = 0, ELEV_CLASS_INTERP = 1 }; };    // IceRegridder.hpp:36-39

class GCMRegridder_Standard;
class RegridMatrices_Dynamic;

/** IceRegridder / IceRegridder_L0 (IceRegridder.hpp:46-133): one ice sheet. */
class IceRegridder {
    friend class GCMRegridder_Standard;
    ibh_regridder *h_ = nullptr;
    std::string _name;
    long _nI = 0, _nX = 0;
    // host copies of what ncio() writes (IceRegridder::ncio, IceRegridder.cpp:75-90); drop_host_copy() releases them
    ExchangeGrid aexgrid_;
    std::vector<double> gridA_proj_area_, gridI_centroid_xy_;
public:
    int interp_style = InterpStyle::Z_INTERP;
    void drop_host_copy() { aexgrid_ = ExchangeGrid(); gridA_proj_area_.clear(); gridI_centroid_xy_.clear(); gridA_proj_area_.shrink_to_fit(); }
    ~IceRegridder() { if (h_) ibh_regridder_destroy(h_); }
    std::string const &name() const { return _name; }
    size_t nI() const { return (size_t)_nI; }
    size_t nX() const { return (size_t)_nX; }
    ibh_regridder *handle() const { return h_; }
};

// ---- RegridMatrices.hpp:39-62 ----------------------------------------------------------------
class RegridMatrices {
    RegridParams _params;
public:
    RegridMatrices(RegridParams const &params) : _params(params) {}
    virtual ~RegridMatrices() {}
    RegridParams const &params() const { return _params; }
    virtual std::unique_ptr<linear::Weighted> matrix(std::string const &spec_name) const = 0;
};

// ---- RegridMatrices_Dynamic.hpp:20-59 --------------------------------------------------------
class RegridMatrices_Dynamic : public RegridMatrices {
    ibh_regrid_matrices *h_;
public:
    IceRegridder const *ice_regridder;
    RegridMatrices_Dynamic(IceRegridder const *_ice_regridder, ibh_regrid_matrices *h, RegridParams const &params)
        : RegridMatrices(params), h_(h), ice_regridder(_ice_regridder) {}
    ~RegridMatrices_Dynamic() { if (h_) ibh_regrid_matrices_destroy(h_); }

    /** matrix_d(spec_name, dims, params): ignores this->params() (RegridMatrices_Dynamic.hpp:51-54).
        dims may be pre-populated and are appended to; they must outlive the result. */
    std::unique_ptr<linear::Weighted_Eigen> matrix_d(std::string const &spec_name,
                                                     std::array<SparseSetT *, 2> dims,
                                                     RegridParams const &params) const {
        ibh_weighted *w = nullptr;
        check(ibh_regrid_matrices_matrix_d(h_, spec_name.c_str(), dims[0] ? dims[0]->handle() : nullptr,
                                           dims[1] ? dims[1]->handle() : nullptr, params.scale, params.correctA,
                                           params.sigma.data(), &w));
        return std::unique_ptr<linear::Weighted_Eigen>(new linear::Weighted_Eigen(w));
    }
    /** matrix_d with the ASSEMBLY shared by the ranks of `comm` (ibh_regrid_matrices_matrix_d_sharded): collective -- every rank
        makes the same call -- and every rank receives the whole matrix, bitwise what matrix_d builds.  Smoothing is not shared
        (params.sigma must be zero here). */
    std::unique_ptr<linear::Weighted_Eigen> matrix_d_sharded(Communicator const &comm, std::string const &spec_name,
                                                             std::array<SparseSetT *, 2> dims, RegridParams const &params) const {
        ibh_weighted *w = nullptr;
        check(ibh_regrid_matrices_matrix_d_sharded(h_, comm.handle(), spec_name.c_str(), dims[0] ? dims[0]->handle() : nullptr,
                                                   dims[1] ? dims[1]->handle() : nullptr, params.scale, params.correctA, &w));
        return std::unique_ptr<linear::Weighted_Eigen>(new linear::Weighted_Eigen(w));
    }
    /** The matrices of one coupling step in one call (IceCoupler.cpp:361-468 builds EvI, AvI, IvE, XvE every step): the
        results of matrix_d(specs[k], dims[k], params[k]) in order; independent builds run concurrently in the library. */
    std::vector<std::unique_ptr<linear::Weighted_Eigen>> matrix_batch(std::vector<std::string> const &specs,
                                                                       std::vector<std::array<SparseSetT *, 2>> const &dims,
                                                                       std::vector<RegridParams> const &params) const {
        const size_t n = specs.size();
        if (dims.size() != n || params.size() != n) throw Exception(IBH_EINVAL, "matrix_batch: argument lists differ in length");
        std::vector<const char *> names(n);
        std::vector<ibh_sparse_set *> d0(n), d1(n);
        std::vector<int32_t> sc(n), ca(n);
        const double *sigma = nullptr;
        for (size_t k = 0; k < n; ++k) {
            names[k] = specs[k].c_str();
            d0[k] = dims[k][0] ? dims[k][0]->handle() : nullptr;
            d1[k] = dims[k][1] ? dims[k][1]->handle() : nullptr;
            sc[k] = params[k].scale; ca[k] = params[k].correctA;
            if (params[k].smooth()) sigma = params[k].sigma.data();      // one sigma per batch
        }
        std::vector<ibh_weighted *> w(n, nullptr);
        check(ibh_regrid_matrices_matrix_batch(h_, (int32_t)n, names.data(), d0.data(), d1.data(), sc.data(), ca.data(), sigma, w.data()));
        std::vector<std::unique_ptr<linear::Weighted_Eigen>> out;
        for (size_t k = 0; k < n; ++k) out.emplace_back(new linear::Weighted_Eigen(w[k]));
        return out;
    }
    /** Produces its own dims (RegridMatrices_Dynamic.cpp:425-437). */
    std::unique_ptr<linear::Weighted> matrix(std::string const &spec_name) const override {
        ibh_weighted *w = nullptr;
        check(ibh_regrid_matrices_matrix(h_, spec_name.c_str(), &w));
        return std::unique_ptr<linear::Weighted>(new linear::Weighted(w));
    }
};

// ---- GCMRegridder.hpp:207-399 ----------------------------------------------------------------
class GCMRegridder_Standard {
    AbbrGrid agridA_;
    std::vector<double> _hcdefs;
    long hc_stride_A_ = 1, hc_stride_HC_ = 0;
    std::vector<std::unique_ptr<IceRegridder>> sheets_;
    std::map<std::string, size_t> sheets_index_;
public:
    bool correctA = false;
    AbbrGrid const *agridA = nullptr;

    /** init(agridA, hcdefs, indexingHC, correctA) (GCMRegridder.cpp:65-86).  indexingHC is given by its
        two strides: iE = iA*stride_A + ihc*stride_HC; the Cython-built default is {1, nA}
        (icebin_cython.cpp:69). */
    void init(AbbrGrid &&_agridA, std::vector<double> &&hcdefs, std::array<long, 2> indexingHC_strides, bool _correctA) {
        agridA_ = std::move(_agridA);
        agridA = &agridA_;
        _hcdefs = std::move(hcdefs);
        hc_stride_A_ = indexingHC_strides[0]; hc_stride_HC_ = indexingHC_strides[1];
        correctA = _correctA;
    }
    void init(AbbrGrid &&_agridA, std::vector<double> &&hcdefs, bool _correctA) {
        long nA_ = _agridA.sparse_extent;
        init(std::move(_agridA), std::move(hcdefs), {{1, nA_}}, _correctA);
    }
    std::vector<double> const &hcdefs() const { return _hcdefs; }
    unsigned int nhc() const { return (unsigned int)_hcdefs.size(); }
    unsigned long nA() const { return (unsigned long)agridA_.sparse_extent; }
    unsigned long nE() const { return nA() * nhc(); }                      // GCMRegridder.hpp:273
    size_t nI(int sheet_index) const { return sheets_.at((size_t)sheet_index)->nI(); }

    /** add_sheet(name, regridder) + IceRegridder::init (GCMRegridder.hpp:353-367, IceRegridder.cpp:93-119):
        the ice grid is given by its size, the exchange grid by value; gridA_proj_area (dense, like
        agridA.native_area) may be empty = no projection (IceRegridder.cpp:106-108). */
    size_t add_sheet(std::string const &name, long nI, ExchangeGrid const &aexgrid,
                     std::vector<double> const &gridA_proj_area = {}, int interp_style = InterpStyle::Z_INTERP,
                     std::vector<double> const &gridI_centroid_xy = {} /* [2*nI], only for sigma != 0 */) {
        std::unique_ptr<IceRegridder> sheet(new IceRegridder);
        std::vector<int64_t> a2s(agridA_.dim_to_sparse.begin(), agridA_.dim_to_sparse.end());
        std::vector<double> const &proj = gridA_proj_area.empty() ? agridA_.native_area : gridA_proj_area;
        ibh_regridder_desc d{};
        d.nX = (int64_t)aexgrid.overlaps.size();
        d.ex_indices = aexgrid.indices.data(); d.ex_area = aexgrid.overlaps.data();
        d.nI = nI; d.nA = agridA_.sparse_extent; d.nA_dense = (int32_t)a2s.size();
        d.A_to_sparse = a2s.data(); d.A_native_area = agridA_.native_area.data(); d.A_proj_area = proj.data();
        d.nhc = (int32_t)_hcdefs.size(); d.hcdefs = _hcdefs.data();
        d.hc_stride_A = hc_stride_A_; d.hc_stride_HC = hc_stride_HC_; d.interp_style = interp_style;
        d.I_centroid_xy = gridI_centroid_xy.empty() ? nullptr : gridI_centroid_xy.data();
        check(ibh_regridder_create(&d, &sheet->h_));
        sheet->_name = name; sheet->_nI = nI; sheet->_nX = (long)d.nX; sheet->interp_style = interp_style;
        sheet->aexgrid_ = aexgrid; sheet->gridA_proj_area_ = proj; sheet->gridI_centroid_xy_ = gridI_centroid_xy;
        size_t ix = sheets_.size();
        sheets_index_[name] = ix;
        sheets_.push_back(std::move(sheet));
        return ix;
    }
    /** GCMRegridder_Standard::ncio(ncio, vname) (GCMRegridder.cpp:104-150, AbbrGrid.cpp:23-29,167-194, IceRegridder.cpp:75-90): the
        IceBin input file -- `<v>.info` {correctA, sheets}, `<v>.agridA.*`, `<v>.indexingHC`, `<v>.hcdefs(<v>.nhc)`, and per sheet
        `<v>.<sheet>.info` {name, interp_style}, `.gridA_proj_area`, `.agridI.*`, `.aexgrid.indices / .overlaps`.  Reading builds
        this (empty) regridder from the file, sheets included (their arrays go to HBM); grid specs / polygons are not part of
        the regrid path and are neither written nor read. */
    void ncio(NcIO &ncio, std::string const &vname = "m") {
        static const std::map<std::string, int> interp = {{"Z_INTERP", 0}, {"ELEV_CLASS_INTERP", 1}};
        if (ncio.reading()) {
            nc::File const &f = ncio.file;
            nc::Var const &info = f.var(vname + ".info");
            AbbrGrid a;
            nc::Var const &adim = f.var(vname + ".agridA.dim");
            a.sparse_extent = (long)adim.att("sparse_extent").at<int64_t>(0);
            a.dim_to_sparse = adim.data.as<long>();
            a.native_area = f.var(vname + ".agridA.native_area").data.as<double>();
            a.name = "gridA";
            // indexingHC: dimension ids by descending stride (GCMRegridder.cpp:43): a file may carry either layout
            nc::Var const &ix = f.var(vname + ".indexingHC");
            std::vector<long> extent = ix.att("extent").as<long>(), order = ix.att("indices").as<long>();
            std::array<long, 2> strides = {{0, 0}};
            strides[(size_t)order[1]] = 1;
            strides[(size_t)order[0]] = extent[(size_t)order[1]];
            init(std::move(a), f.var(vname + ".hcdefs").data.as<double>(), strides, info.att("correctA").at<int>(0) != 0);
            for (std::string const &name : nc::split_names(info.att("sheets").str())) {
                const std::string v = vname + "." + name;
                nc::Var const &sinfo = f.var(v + ".info");
                const std::string style = sinfo.att("interp_style").type == nc::CHAR ? sinfo.att("interp_style").str() : std::string();
                const int istyle = style.empty() ? sinfo.att("interp_style").at<int>(0) : interp.at(style);
                nc::Var const &idim = f.var(v + ".agridI.dim");
                const long nI = (long)idim.att("sparse_extent").at<int64_t>(0);
                std::vector<long> i2s = idim.data.as<long>();
                std::vector<double> cen_d = f.var(v + ".agridI.centroid_xy").data.as<double>(), cen;
                bool any = false;
                for (double c : cen_d) any = any || c != 0.0;
                if (any) {                              // dense -> sparse ice index
                    cen.assign((size_t)(2 * nI), 0.0);
                    for (size_t k = 0; k < i2s.size(); ++k) { cen[2 * (size_t)i2s[k]] = cen_d[2 * k]; cen[2 * (size_t)i2s[k] + 1] = cen_d[2 * k + 1]; }
                }
                ExchangeGrid ex;
                ex.indices = f.var(v + ".aexgrid.indices").data.as<int>();
                ex.overlaps = f.var(v + ".aexgrid.overlaps").data.as<double>();
                add_sheet(name, nI, ex, f.var(v + ".gridA_proj_area").data.as<double>(), istyle, cen);
            }
            return;
        }
        nc::File &f = ncio.file;
        const int32_t zero = 0, ca = correctA ? 1 : 0;
        std::vector<std::string> names;
        for (auto const &sh : sheets_) names.push_back(sh->name());
        f.add_var(vname + ".info", {}, nc::Array::of(&zero, 1), {{"correctA", nc::Array::of(&ca, 1)}, {"sheets", nc::Array::str(nc::join_names(names))}});
        auto put_abbr = [&](std::string const &v, std::vector<long> const &to_sparse, long extent, std::vector<double> const &native,
                            std::vector<double> const &centroid, std::string const &gname) {
            f.add_var(v + ".info", {}, nc::Array::of(&zero, 1), {{"coordinates", nc::Array::str("XY")}, {"parameterization", nc::Array::str("L0")},
                                                               {"name", nc::Array::str(gname)}, {"sproj", nc::Array::str("")}});
            std::vector<int64_t> t64(to_sparse.begin(), to_sparse.end());
            const std::string d = f.add_dim(v + ".dim.dense_extent", (int64_t)t64.size());
            const int64_t ext = extent;
            f.add_var(v + ".dim", {d}, nc::Array::of(t64), {{"sparse_extent", nc::Array::of(&ext, 1)}});
            const std::string three = f.add_dim("three", 3), two = f.add_dim("two", 2);
            f.add_var(v + ".ijk", {d, three}, nc::Array::of(std::vector<int32_t>(3 * t64.size(), 0)));
            f.add_var(v + ".native_area", {d}, nc::Array::of(native));
            f.add_var(v + ".centroid_xy", {d, two}, nc::Array::of(centroid.empty() ? std::vector<double>(2 * t64.size(), 0.0) : centroid));
        };
        put_abbr(vname + ".agridA", agridA_.dim_to_sparse, agridA_.sparse_extent, agridA_.native_area, {}, "gridA");
        const int64_t base[2] = {0, 0}, extent[2] = {(int64_t)nA(), (int64_t)nhc()};
        const int32_t order[2] = {hc_stride_HC_ >= hc_stride_A_ ? 1 : 0, hc_stride_HC_ >= hc_stride_A_ ? 0 : 1};
        f.add_var(vname + ".indexingHC", {}, nc::Array::of(&zero, 1),
                  {{"base", nc::Array::of(base, 2)}, {"extent", nc::Array::of(extent, 2)}, {"indices", nc::Array::of(order, 2)}});
        f.add_var(vname + ".hcdefs", {f.add_dim(vname + ".nhc", (int64_t)_hcdefs.size())}, nc::Array::of(_hcdefs));
        f.add_dim("agridA.ndata", (int64_t)agridA_.dim_to_sparse.size());
        for (auto const &sh : sheets_) {
            if (sh->aexgrid_.overlaps.empty() && sh->nX() != 0) throw Exception(IBH_EINVAL, "sheet '" + sh->name() + "': host copy was dropped, cannot be written");
            const std::string v = vname + "." + sh->name();
            f.add_var(v + ".info", {}, nc::Array::of(&zero, 1), {{"name", nc::Array::str(sh->name())},
                      {"interp_style", nc::Array::str(sh->interp_style == InterpStyle::ELEV_CLASS_INTERP ? "ELEV_CLASS_INTERP" : "Z_INTERP")}});
            f.add_var(v + ".gridA_proj_area", {"agridA.ndata"}, nc::Array::of(sh->gridA_proj_area_));
            std::vector<long> iota((size_t)sh->nI());
            for (size_t k = 0; k < iota.size(); ++k) iota[k] = (long)k;
            put_abbr(v + ".agridI", iota, (long)sh->nI(), std::vector<double>(sh->nI(), 0.0), sh->gridI_centroid_xy_, "gridI");
            f.add_var(v + ".aexgrid.indices", {f.add_dim(v + ".aexgrid.nindices", (int64_t)sh->aexgrid_.indices.size())}, nc::Array::of(sh->aexgrid_.indices));
            f.add_var(v + ".aexgrid.overlaps", {f.add_dim(v + ".aexgrid.noverlaps", (int64_t)sh->aexgrid_.overlaps.size())}, nc::Array::of(sh->aexgrid_.overlaps));
        }
        ncio.touch();
    }
    /** ice_regridders().index.at(name) */
    size_t sheet_index(std::string const &name) const {
        auto it = sheets_index_.find(name);
        if (it == sheets_index_.end()) throw Exception(IBH_ENOKEY, "no ice sheet named '" + name + "'");
        return it->second;
    }
    IceRegridder const *ice_regridder(size_t ix) const { return sheets_.at(ix).get(); }

    /** regrid_matrices(sheet_index, elevmaskI, params = RegridParams()) (GCMRegridder.hpp:290-293;
        RegridMatrices_Dynamic.cpp:334-402).  elevmaskI is copied. */
    std::unique_ptr<RegridMatrices_Dynamic> regrid_matrices(int sheet_index, ArrayView<const double> const &elevmaskI,
                                                            RegridParams const &params = RegridParams()) const {
        IceRegridder const *regridder = sheets_.at((size_t)sheet_index).get();
        ibh_regrid_matrices *rm = nullptr;
        check(ibh_regrid_matrices_create(regridder->h_, elevmaskI.data, elevmaskI.size(), params.scale, params.correctA,
                                         params.sigma.data(), &rm));
        return std::unique_ptr<RegridMatrices_Dynamic>(new RegridMatrices_Dynamic(regridder, rm, params));
    }
    /** Same with the elevation mask already in device memory (no counterpart in the reference, whose
        ice models live on the host): copied device-to-device on `stream`. */
    std::unique_ptr<RegridMatrices_Dynamic> regrid_matrices_device(int sheet_index, const double *d_elevmaskI, long n,
                                                                   RegridParams const &params = RegridParams(),
                                                                   void *stream = nullptr) const {
        IceRegridder const *regridder = sheets_.at((size_t)sheet_index).get();
        ibh_regrid_matrices *rm = nullptr;
        check(ibh_regrid_matrices_create_device(regridder->h_, d_elevmaskI, n, params.scale, params.correctA,
                                                params.sigma.data(), stream, &rm));
        return std::unique_ptr<RegridMatrices_Dynamic>(new RegridMatrices_Dynamic(regridder, rm, params));
    }
    /** GCMRegridder::wA (GCMRegridder.hpp:305-315, icebin_cython.cpp:103-117) */
    std::vector<double> wA(std::string const &ice_sheet_name, bool native, double fill = 0.) const {
        std::vector<double> out((size_t)nA());
        check(ibh_regridder_wA(sheets_.at(sheet_index(ice_sheet_name))->h_, native, fill, out.data()));
        return out;
    }
};

// ---- e1ve0.hpp / e1ve0.cpp:55-106 ----------------------------------------------------------------
namespace e1ve0 {
/** compute_E1vE0c(XuE1s, XuE0s, nE, areaX): the correction matrix between last step's and this step's elevation grids,
    one XuE matrix per ice sheet; returned over the sparse E space (rows iE1, columns iE0), where the reference returns
    the same tuples as a spsparse::TupleList (areaX is unused there too). */
inline std::unique_ptr<linear::Weighted_Eigen> compute_E1vE0c(std::vector<linear::Weighted_Eigen const *> const &XuE1s,
                                                               std::vector<linear::Weighted_Eigen const *> const &XuE0s,
                                                               unsigned long nE) {
    if (XuE1s.size() != XuE0s.size() || XuE1s.empty()) throw Exception(IBH_EINVAL, "compute_E1vE0c: need one XuE1 and one XuE0 per ice sheet");
    std::vector<const ibh_weighted *> a, b;
    for (auto *w : XuE1s) a.push_back(w->handle());
    for (auto *w : XuE0s) b.push_back(w->handle());
    ibh_weighted *out = nullptr;
    check(ibh_e1ve0_compute((int32_t)a.size(), a.data(), b.data(), (int64_t)nE, &out));
    return std::unique_ptr<linear::Weighted_Eigen>(new linear::Weighted_Eigen(out));
}
}   // namespace e1ve0

// ---- gridgen/GridGen_Exchange.cpp:175-284 ----------------------------------------------------------
/** make_exchange_grid for a rectilinear XY ice grid under convex projected GCM-cell polygons (ibh_exgrid_generate). */
inline ExchangeGrid make_exchange_grid(std::vector<double> const &xedges, std::vector<double> const &yedges, bool x_fastest,
                                       std::vector<int> const &polyptr, std::vector<double> const &vx, std::vector<double> const &vy,
                                       std::vector<long> const &iA) {
    std::vector<int64_t> ia(iA.begin(), iA.end());
    ibh_exgrid_desc d{};
    d.nx = (int32_t)xedges.size() - 1; d.ny = (int32_t)yedges.size() - 1; d.xedges = xedges.data(); d.yedges = yedges.data();
    d.x_fastest = x_fastest; d.npoly = (int32_t)ia.size(); d.polyptr = polyptr.data(); d.vx = vx.data(); d.vy = vy.data(); d.iA = ia.data();
    ibh_exgrid *h = nullptr;
    check(ibh_exgrid_generate(&d, &h));
    ExchangeGrid ex;
    int64_t n = 0;
    int rc = ibh_exgrid_size(h, &n);
    if (rc == IBH_OK) { ex.indices.resize((size_t)(2 * n)); ex.overlaps.resize((size_t)n); rc = ibh_exgrid_get(h, ex.indices.data(), ex.overlaps.data()); }
    ibh_exgrid_destroy(h);
    check(rc);
    return ex;
}

// ---- pylib/icebin_cython.hpp:70-87 -----------------------------------------------------------
namespace cython {
/** new_regrid_matrices(gcm, sheet_name, elevmaskI, scale, correctA, sigma_x, sigma_y, sigma_z, conserve):
    like the reference (icebin_cython.cpp:215-236) looks the sheet up by name, checks the shape {nI}
    and ignores `conserve`.  The PyObject* becomes a plain (pointer, length). */
inline RegridMatrices *new_regrid_matrices(GCMRegridder_Standard const *gcm, std::string const &sheet_name,
                                           const double *elevmaskI, long elevmaskI_len, bool scale, bool correctA,
                                           double sigma_x, double sigma_y, double sigma_z, bool /*conserve*/) {
    auto sheet_index = gcm->sheet_index(sheet_name);
    return gcm->regrid_matrices((int)sheet_index, ArrayView<const double>(elevmaskI, elevmaskI_len),
                                RegridParams(scale, correctA, {{sigma_x, sigma_y, sigma_z}})).release();
}
/** RegridMatrices_matrix(cself, spec_name) (icebin_cython.cpp:195-198) */
inline linear::Weighted *RegridMatrices_matrix(RegridMatrices *cself, std::string const &spec_name) {
    return cself->matrix(spec_name).release();
}
}   // namespace cython

}   // namespace icebin
