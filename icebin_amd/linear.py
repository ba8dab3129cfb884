"""linear_Weighted: the Python face of ibmisc::linear::Weighted (sphinx/source/matrix_formats.rst:121-195),
and SparseSet (spsparse::SparseSet<long,int>), both thin handles over the C-ABI."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib, ptr


class SparseSet:
    """spsparse::SparseSet<long,int>: dense ids handed out first-seen (eigen_types.hpp:24)."""

    def __init__(self, sparse_extent=-1, to_sparse=None, _handle=None):
        h = C.c_void_p()
        if _handle is not None:
            h = _handle
        elif to_sparse is None:
            check(lib().ibh_sparse_set_create(int(sparse_extent), C.byref(h)))
        else:
            a = np.ascontiguousarray(to_sparse, np.int64)
            check(lib().ibh_sparse_set_from_array(int(sparse_extent), ptr(a), len(a), C.byref(h)))
        self._h = h

    @classmethod
    def identity(cls, n):
        """ibmisc id_sparse_set(n) (modele/merge_topo.cpp:48, IceCoupler.cpp:366)."""
        h = C.c_void_p()
        check(lib().ibh_sparse_set_create_identity(int(n), C.byref(h)))
        return cls(_handle=h)

    def __del__(self):
        try:
            _capi.destroy("ibh_sparse_set_destroy", getattr(self, "_h", None))
        except Exception:      # interpreter shutdown
            pass
        self._h = None

    def sparse_extent(self):
        v = C.c_int64()
        check(lib().ibh_sparse_set_sparse_extent(self._h, C.byref(v)))
        return v.value

    def dense_extent(self):
        v = C.c_int32()
        check(lib().ibh_sparse_set_dense_extent(self._h, C.byref(v)))
        return v.value

    def to_sparse(self):
        out = np.empty(self.dense_extent(), np.int64)
        check(lib().ibh_sparse_set_to_sparse(self._h, ptr(out)))
        return out

    def add_dense(self, sparse):
        """Existing dense id of `sparse`, or the next one (first-seen order)."""
        d = C.c_int32()
        check(lib().ibh_sparse_set_add_dense(self._h, int(sparse), C.byref(d)))
        return d.value

    def to_dense(self, sparse):
        d = C.c_int32()
        check(lib().ibh_sparse_set_to_dense(self._h, int(sparse), C.byref(d)))
        if d.value < 0:
            raise KeyError(sparse)
        return d.value

    def in_sparse(self, sparse):
        d = C.c_int32()
        check(lib().ibh_sparse_set_to_dense(self._h, int(sparse), C.byref(d)))
        return d.value >= 0


class linear_Weighted:
    """A regrid matrix M plus its two weight vectors, resident in HBM.

    Mirrors ibmisc.linear_Weighted (matrix_formats.rst:139-195): apply_M, apply_weight, apply_wM,
    apply_Mw, to_coo, get_weights, shape; plus the C++ members of linear::Weighted_Eigen the IceBin
    callers touch (dims, wM, Mw, conservative, scaled, apply()).
    """

    def __init__(self, handle, keep=()):
        self._h = handle
        self._keep = keep          # objects that must outlive the handle (shared dims, regridder)
        nrow, ncol, nnz = C.c_int32(), C.c_int32(), C.c_int64()
        check(lib().ibh_weighted_shape(handle, C.byref(nrow), C.byref(ncol), C.byref(nnz)))
        self.nrow_d, self.ncol_d, self.nnz = nrow.value, ncol.value, nnz.value
        cons, sc = C.c_int(), C.c_int()
        check(lib().ibh_weighted_flags(handle, C.byref(cons), C.byref(sc)))
        self.conservative, self.scaled = bool(cons.value), bool(sc.value)
        self._dims = [None, None]
        self._sparse_extents = [None, None]

    def __del__(self):
        try:
            _capi.destroy("ibh_weighted_destroy", getattr(self, "_h", None))
        except Exception:      # interpreter shutdown
            pass
        self._h = None

    # ---- loaders -------------------------------------------------------------------------
    @classmethod
    def from_coo(cls, shape, row, col, val, wM, Mw, conservative=True, scaled=True):
        """Dense-indexed triplets -> Weighted (to_eigen_M, eigen_types.cpp:9-34 / nc_read_weighted)."""
        row = np.ascontiguousarray(row, np.int32)
        col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, np.float64)
        wM = np.ascontiguousarray(wM, np.float64)
        Mw = np.ascontiguousarray(Mw, np.float64)
        assert len(row) == len(col) == len(val) and len(wM) == shape[0] and len(Mw) == shape[1]
        h = C.c_void_p()
        check(lib().ibh_weighted_from_coo(shape[0], shape[1], len(val), ptr(row), ptr(col), ptr(val), ptr(wM), ptr(Mw),
                                         int(conservative), int(scaled), C.byref(h)))
        return cls(h)

    @classmethod
    def from_csr(cls, shape, rowptr, colind, val, wM, Mw, conservative=True, scaled=True):
        rowptr = np.ascontiguousarray(rowptr, np.int32)
        colind = np.ascontiguousarray(colind, np.int32)
        val = np.ascontiguousarray(val, np.float64)
        wM = np.ascontiguousarray(wM, np.float64)
        Mw = np.ascontiguousarray(Mw, np.float64)
        assert len(rowptr) == shape[0] + 1 and len(wM) == shape[0] and len(Mw) == shape[1]
        h = C.c_void_p()
        check(lib().ibh_weighted_from_csr(shape[0], shape[1], ptr(rowptr), ptr(colind), ptr(val), ptr(wM), ptr(Mw),
                                         int(conservative), int(scaled), C.byref(h)))
        return cls(h)

    # ---- members -------------------------------------------------------------------------
    def dim(self, k):
        """dims[k]: dense -> sparse index table of this matrix (int64[dense_extent])."""
        if self._dims[k] is None:
            n = self.nrow_d if k == 0 else self.ncol_d
            out = np.empty(n, np.int64)
            check(lib().ibh_weighted_dim_to_sparse(self._h, k, ptr(out)))
            ext = C.c_int64()
            check(lib().ibh_weighted_dim(self._h, k, C.byref(ext), None))
            self._dims[k], self._sparse_extents[k] = out, ext.value
        return self._dims[k]

    def sparse_extent(self, k):
        self.dim(k)
        return self._sparse_extents[k]

    @property
    def shape(self):
        """Sparse shape of the matrix."""
        return (self.sparse_extent(0), self.sparse_extent(1))

    @property
    def wM(self):
        out = np.empty(self.nrow_d, np.float64)
        check(lib().ibh_weighted_get_wM(self._h, ptr(out)))
        return out

    @property
    def Mw(self):
        out = np.empty(self.ncol_d, np.float64)
        check(lib().ibh_weighted_get_Mw(self._h, ptr(out)))
        return out

    def get_weights(self, dim):
        """Weight vector in the sparse space (0 = B/output, 1 = A/input); zero where unset."""
        out = np.zeros(self.sparse_extent(dim), np.float64)
        out[self.dim(dim)] = self.wM if dim == 0 else self.Mw
        return out

    def coo_dense(self):
        """(row_d, col_d, val) in row-major order, dense index spaces."""
        row = np.empty(self.nnz, np.int32)
        col = np.empty(self.nnz, np.int32)
        val = np.empty(self.nnz, np.float64)
        check(lib().ibh_weighted_get_coo(self._h, ptr(row), ptr(col), ptr(val)))
        return row, col, val

    def csr_dense(self):
        rowptr = np.empty(self.nrow_d + 1, np.int32)
        col = np.empty(self.nnz, np.int32)
        val = np.empty(self.nnz, np.float64)
        check(lib().ibh_weighted_get_csr(self._h, ptr(rowptr), ptr(col), ptr(val)))
        return rowptr, col, val

    def to_coo(self):
        """scipy.sparse.coo_matrix in the ORIGINAL sparse index spaces (matrix_formats.rst:188-192)."""
        import scipy.sparse
        row, col, val = self.coo_dense()
        return scipy.sparse.coo_matrix((val, (self.dim(0)[row], self.dim(1)[col])), shape=self.shape)

    # ---- on-disk "Eigen format" (matrix_formats.rst:9-63) -----------------------------------
    def ncio(self, ds, vname, dim_names):
        """Weighted_Eigen::ncio(ncio, vname, {dimB, dimA}) into an icebin_amd.ncio.Dataset (write side;
        modele/global_ec.cpp:571-629, IceCoupler.cpp:473-488)."""
        from . import ncio
        row, col, val = self.coo_dense()
        ncio.put_weighted(ds, vname, dim_names, (self.dim(0), self.dim(1)), (self.sparse_extent(0), self.sparse_extent(1)),
                          row, col, val, self.wM, self.Mw, self.conservative, self.scaled)

    @classmethod
    def nc_read(cls, ds, vname):
        """ibmisc.nc_read_weighted(nc, vname) (matrix_formats.rst:139-147): the matrix, its weights and its
        dims from an Eigen-format file (a path or an icebin_amd.ncio.Dataset)."""
        from . import ncio
        if isinstance(ds, (str, bytes)) or hasattr(ds, "__fspath__"):
            ds = ncio.Dataset.read(ds)
        d = ncio.get_weighted(ds, vname)
        w = cls.from_coo((len(d["dims"][0]), len(d["dims"][1])), d["row"], d["col"], d["val"], d["wM"], d["Mw"],
                         d["conservative"], d["scaled"])
        w._dims = [np.asarray(d["dims"][0], np.int64), np.asarray(d["dims"][1], np.int64)]
        w._sparse_extents = list(d["sparse_extents"])
        return w

    # ---- products ------------------------------------------------------------------------
    def apply(self, A_b, fill=np.nan, force_conservation=True):
        """Weighted_Eigen::apply(A_b, fill, force_conservation, tmp) in DENSE index spaces
        (modele/merge_topo.cpp:65, modele/icebin22m.cpp:153).  A_b: [ncol_d] or [nvar, ncol_d]."""
        A = np.ascontiguousarray(np.atleast_2d(A_b), np.float64)
        if A.shape[1] != self.ncol_d:
            raise ValueError("apply: input has %d columns, matrix has %d dense columns" % (A.shape[1], self.ncol_d))
        B = np.empty((A.shape[0], self.nrow_d), np.float64)
        check(lib().ibh_weighted_apply_host(self._h, ptr(A), A.shape[0], A.shape[1], ptr(B), B.shape[1], float(fill),
                                           int(force_conservation)))
        return B if np.ndim(A_b) == 2 else B[0]

    def apply_device(self, dA, out=None, fill=float("nan"), force_conservation=True, stream=None):
        """Same on fields already resident in HBM: dA is a torch.float64 CUDA tensor [nvar, ncol_d]
        (row stride >= ncol_d); only enqueues work on `stream` (default: torch's current stream)."""
        import torch
        assert dA.is_cuda and dA.dtype == torch.float64 and dA.dim() == 2 and dA.stride(1) == 1
        assert dA.shape[1] == self.ncol_d
        nvar = dA.shape[0]
        if out is None:
            out = _aligned_planes(torch, nvar, self.nrow_d, dA.device)
        assert out.is_cuda and out.dtype == torch.float64 and out.shape == (nvar, self.nrow_d) and out.stride(1) == 1
        s = torch.cuda.current_stream(dA.device).cuda_stream if stream is None else stream
        check(lib().ibh_weighted_apply_device(self._h, C.c_void_p(dA.data_ptr()), nvar, dA.stride(0) if nvar > 1 else max(dA.stride(0), self.ncol_d),
                                             C.c_void_p(out.data_ptr()), out.stride(0) if nvar > 1 else max(out.stride(0), self.nrow_d),
                                             float(fill), int(force_conservation), C.c_void_p(s)))
        return out

    def apply_many_device(self, dAs, outs=None, fill=float("nan"), force_conservation=True, stream=None):
        """Several field batches through one launch (ibh_weighted_apply_many_device): dAs is a list of
        torch.float64 CUDA tensors [nvar, ncol_d] of equal shape and row stride; returns the list of
        results.  Bitwise the results of separate apply_device calls."""
        import torch
        nb = len(dAs)
        if nb == 0:
            return []
        nvar, lda = dAs[0].shape[0], max(dAs[0].stride(0), self.ncol_d)
        for a in dAs:
            assert a.is_cuda and a.dtype == torch.float64 and a.dim() == 2 and a.stride(1) == 1
            assert a.shape == (nvar, self.ncol_d) and max(a.stride(0), self.ncol_d) == lda
        if outs is None:
            outs = [_aligned_planes(torch, nvar, self.nrow_d, dAs[0].device) for _ in range(nb)]
        ldb = max(outs[0].stride(0), self.nrow_d)
        for o in outs:
            assert o.is_cuda and o.dtype == torch.float64 and o.shape == (nvar, self.nrow_d) and o.stride(1) == 1
            assert max(o.stride(0), self.nrow_d) == ldb
        xp = (C.c_void_p * nb)(*[a.data_ptr() for a in dAs])
        yp = (C.c_void_p * nb)(*[o.data_ptr() for o in outs])
        s = torch.cuda.current_stream(dAs[0].device).cuda_stream if stream is None else stream
        check(lib().ibh_weighted_apply_many_device(self._h, nb, xp, nvar, lda, yp, ldb, float(fill), int(force_conservation),
                                                  C.c_void_p(s)))
        return outs

    def reserve(self, nvar):
        """Size the handle's per-apply scratch for up to nvar variables (needed before capturing
        applies into a hipGraph: growing scratch allocates)."""
        check(lib().ibh_weighted_reserve(self._h, int(nvar)))

    def prepare(self, nvar, nbatch=1):
        """Make applies of up to nvar variables, up to nbatch batches per launch, pure enqueues: builds the
        column-sweep / band structure those applies would otherwise build lazily on their second call and
        sizes all scratch (ibh_weighted_prepare).  Call once per matrix before capturing applies into a
        hipGraph or when the latency of the first applies matters."""
        check(lib().ibh_weighted_prepare(self._h, int(nvar), int(nbatch)))

    def pair_prepare(self, second, nvar=16):
        """Prepare the fused pair  B1 = self * A,  B2 = second * B1  (ibh_weighted_pair_prepare): EvI followed by AvE.
        Raises IcebinHipError (IBH_ENOTIMPL) when the two matrices do not pair; then make two applies."""
        check(lib().ibh_weighted_pair_prepare(self._h, second._h, int(nvar)))
        self._pair = second          # (keeps the second handle alive as long as the pairing)

    def apply_pair_device(self, second, dA, out1=None, out2=None, fill=float("nan"), stream=None):
        """One launch for both products (ibh_weighted_apply_pair_device); returns (B1, B2).  B1 is bitwise apply_device's
        result, B2 = second.apply_device(B1, force_conservation=False) to 1e-12."""
        import torch
        assert dA.is_cuda and dA.dtype == torch.float64 and dA.dim() == 2 and dA.stride(1) == 1 and dA.shape[1] == self.ncol_d
        nvar = dA.shape[0]
        if out1 is None:
            out1 = _aligned_planes(torch, nvar, self.nrow_d, dA.device)
        if out2 is None:
            out2 = _aligned_planes(torch, nvar, second.nrow_d, dA.device)
        for o, n in ((out1, self.nrow_d), (out2, second.nrow_d)):
            assert o.is_cuda and o.dtype == torch.float64 and o.shape == (nvar, n) and o.stride(1) == 1
        s = torch.cuda.current_stream(dA.device).cuda_stream if stream is None else stream
        ld = lambda t, n: t.stride(0) if nvar > 1 else max(t.stride(0), n)
        check(lib().ibh_weighted_apply_pair_device(self._h, second._h, C.c_void_p(dA.data_ptr()), nvar, ld(dA, self.ncol_d),
                                                  C.c_void_p(out1.data_ptr()), ld(out1, self.nrow_d), C.c_void_p(out2.data_ptr()),
                                                  ld(out2, second.nrow_d), float(fill), C.c_void_p(s)))
        return out1, out2

    def apply_chain_device(self, second, third, dA, out1=None, out2=None, out3=None, fill=float("nan"), stream=None):
        """B1 = self * A, B2 = second * B1 (the fused pair: pair_prepare(second) first), B3 = third * B2 with the third product's
        launch overlapped with the pair kernel (ibh_weighted_apply_chain_device); returns (B1, B2, B3), bitwise the pair apply
        followed by third.apply_device."""
        import torch
        assert dA.is_cuda and dA.dtype == torch.float64 and dA.dim() == 2 and dA.stride(1) == 1 and dA.shape[1] == self.ncol_d
        nvar = dA.shape[0]
        outs = []
        for o, n in ((out1, self.nrow_d), (out2, second.nrow_d), (out3, third.nrow_d)):
            if o is None:
                o = _aligned_planes(torch, nvar, n, dA.device)
            assert o.is_cuda and o.dtype == torch.float64 and o.shape == (nvar, n) and o.stride(1) == 1
            outs.append(o)
        s = torch.cuda.current_stream(dA.device).cuda_stream if stream is None else stream
        ld = lambda t, n: t.stride(0) if nvar > 1 else max(t.stride(0), n)
        check(lib().ibh_weighted_apply_chain_device(self._h, second._h, third._h, C.c_void_p(dA.data_ptr()), nvar, ld(dA, self.ncol_d),
                                                   C.c_void_p(outs[0].data_ptr()), ld(outs[0], self.nrow_d), C.c_void_p(outs[1].data_ptr()),
                                                   ld(outs[1], second.nrow_d), C.c_void_p(outs[2].data_ptr()), ld(outs[2], third.nrow_d),
                                                   float(fill), C.c_void_p(s)))
        return tuple(outs)

    def apply_transformed_device(self, dV, T, b, out=None, fill=float("nan"), stream=None):
        """The coupler's fused product  M * (V*T + b)  on HBM-resident fields (IceCoupler.cpp:203-252,
        :445): dV torch.float64 CUDA [nvar_in, ncol_d]; T [nvar_in, nvar_out] (the sparse variable
        transform as a dense array, exact zeros are structural), b [nvar_out]; returns [nvar_out, nrow_d]."""
        import torch
        T = np.ascontiguousarray(T, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        assert dV.is_cuda and dV.dtype == torch.float64 and dV.dim() == 2 and dV.stride(1) == 1
        assert dV.shape == (T.shape[0], self.ncol_d) and b.shape == (T.shape[1],)
        nout = T.shape[1]
        if out is None:
            out = _aligned_planes(torch, nout, self.nrow_d, dV.device)
        s = torch.cuda.current_stream(dV.device).cuda_stream if stream is None else stream
        check(lib().ibh_weighted_apply_transformed_device(
            self._h, C.c_void_p(dV.data_ptr()), dV.shape[0], max(dV.stride(0), self.ncol_d), ptr(T), ptr(b), nout,
            C.c_void_p(out.data_ptr()), max(out.stride(0), self.nrow_d), float(fill), C.c_void_p(s)))
        return out

    def apply_transformed(self, V_b, T, b, fill=np.nan):
        """Host-array form of apply_transformed_device (dense index spaces)."""
        import torch
        dV = torch.from_numpy(np.ascontiguousarray(np.atleast_2d(V_b), np.float64)).cuda()
        out = self.apply_transformed_device(dV, T, b, fill=fill)
        torch.cuda.synchronize()
        return out.cpu().numpy()

    def apply_M(self, A_s, fill=np.nan, force_conservation=True):
        """Applies the regrid matrix to A_s (vectors in the ORIGINAL sparse space of the input grid):
        a 1-D vector or a 2-D array of row vectors; un-set output cells get `fill`
        (matrix_formats.rst:156-165)."""
        A = np.asarray(A_s, np.float64)
        one = A.ndim == 1
        A2 = A.reshape(1, -1) if one else A.reshape(A.shape[0], -1)
        if A2.shape[1] != self.sparse_extent(1):
            raise ValueError("apply_M: input has %d cells, the input grid has %d" % (A2.shape[1], self.sparse_extent(1)))
        B_d = self.apply(np.ascontiguousarray(A2[:, self.dim(1)]), fill, force_conservation)
        B_s = np.full((A2.shape[0], self.sparse_extent(0)), fill, np.float64)
        B_s[:, self.dim(0)] = B_d
        return B_s[0] if one else B_s

    def apply_weight(self, dim, A_s):
        """Dot product of a weight vector with A_s (dim 0: wM / output grid, 1: Mw / input grid)."""
        A = np.asarray(A_s, np.float64)
        one = A.ndim == 1
        A2 = A.reshape(1, -1) if one else A.reshape(A.shape[0], -1)
        if A2.shape[1] != self.sparse_extent(dim):
            raise ValueError("apply_weight: input has %d cells, expected %d" % (A2.shape[1], self.sparse_extent(dim)))
        A_d = np.ascontiguousarray(A2[:, self.dim(dim)])
        out = np.empty(A_d.shape[0], np.float64)
        check(lib().ibh_weighted_apply_weight_host(self._h, dim, ptr(A_d), A_d.shape[0], A_d.shape[1], ptr(out)))
        return out[0] if one else out

    def apply_wM(self, A_s):
        return self.apply_weight(0, A_s)

    def apply_Mw(self, A_s):
        return self.apply_weight(1, A_s)

    # ---- tuning / introspection ----------------------------------------------------------
    def set_kernel(self, name):
        check(lib().ibh_weighted_set_kernel(self._h, name.encode()))

    def set_option(self, key, value):
        """ibh_weighted_set_option: a launch option of THIS matrix (the apply-side ibh_set_tuning keys), read before the
        process-wide map; value None removes it."""
        check(lib().ibh_weighted_set_option(self._h, key.encode(), -2 ** 31 if value is None else int(value)))

    def built_fast(self):
        """True when the plan-based fast assembly path built this matrix (sorted exchange grids)."""
        v = C.c_int()
        check(lib().ibh_weighted_built_fast(self._h, C.byref(v)))
        return bool(v.value)

    def built_streamed(self):
        """True when the streamed variant of the plan-based assembly (streamasm.inl: large sorted grids) built this matrix."""
        v = C.c_int()
        check(lib().ibh_weighted_built_fast(self._h, C.byref(v)))
        return v.value in (2, 3)

    def built_sharded(self):
        """True when the ranks of a communicator shared the assembly of this matrix (ibh_regrid_matrices_matrix_d_sharded)."""
        v = C.c_int()
        check(lib().ibh_weighted_built_fast(self._h, C.byref(v)))
        return v.value == 3

    def last_kernel(self):
        buf = C.create_string_buffer(32)
        check(lib().ibh_weighted_last_kernel(self._h, buf, 32))
        return buf.value.decode()

    def last_launch(self):
        """The kernel instantiation of the last apply as rocprofv3 names it ("" when the kernel family does not record it)."""
        buf = C.create_string_buffer(64)
        check(lib().ibh_weighted_last_launch(self._h, buf, 64))
        return buf.value.decode()


def _aligned_planes(torch, nvar, n, device):
    """[nvar, n] float64 view whose rows (field planes) start on 512-byte boundaries: whole-line
    wave stores measure 6.0 instead of 4.0 TB/s on MI355X (shortrow)."""
    ld = (n + 63) // 64 * 64
    return torch.empty((nvar, ld), dtype=torch.float64, device=device)[:, :n]


def coo_multiply(M, xx, fill=np.nan, ignore_nan=False):
    """icebin.coo_multiply(M, x, fill=nan, ignore_nan=False) (coo_matvec, pylib/icebin_cython.cpp:158-192;
    tests/test_conserv/test_conserv.py:139): yy = M xx for a scipy.sparse COO matrix in the original
    (sparse) index spaces; rows with no (surviving) entry hold `fill`."""
    M = M.tocoo()
    xx = np.ascontiguousarray(xx, np.float64).reshape(-1)
    if xx.shape[0] != M.shape[1]:
        raise ValueError("coo_multiply: xx has %d elements, M has %d columns" % (xx.shape[0], M.shape[1]))
    yy = np.full(M.shape[0], fill, np.float64)
    row, col = np.ascontiguousarray(M.row, np.int32), np.ascontiguousarray(M.col, np.int32)
    data = np.ascontiguousarray(M.data, np.float64)
    check(lib().ibh_coo_matvec(ptr(yy), ptr(xx), int(bool(ignore_nan)), M.shape[0], M.shape[1], len(data), ptr(row), ptr(col),
                              ptr(data)))
    return yy


def compute_E1vE0c(XuE1s, XuE0s, nE):
    """e1ve0::compute_E1vE0c (slib/icebin/e1ve0.cpp:55-106): lists of XvE matrices (unscaled, one per ice sheet) for
    the new (E1) and the previous (E0) elevation mask -> the correction matrix as a linear_Weighted over the
    sparse E space; .to_coo() gives the reference's (iE1, iE0, value) tuples, sorted and consolidated."""
    n = len(XuE1s)
    assert n == len(XuE0s) and n >= 1
    a1 = (C.c_void_p * n)(*[w._h.value for w in XuE1s])
    a0 = (C.c_void_p * n)(*[w._h.value for w in XuE0s])
    h = C.c_void_p()
    check(lib().ibh_e1ve0_compute(n, a1, a0, int(nE), C.byref(h)))
    return linear_Weighted(h)


def nc_read_weighted(nc, vname):
    """ibmisc.nc_read_weighted (matrix_formats.rst:139-147)."""
    return linear_Weighted.nc_read(nc, vname)


def set_tuning(key, value):
    check(lib().ibh_set_tuning(key.encode(), int(value)))
