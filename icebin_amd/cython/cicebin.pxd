# cicebin.pxd -- C++ declarations behind the Cython module, the counterpart of the reference's
# pylib/cicebin.pxd:76-118: the same classes and free functions, here provided by the header-only host
# mirror icebin_amd/host/icebin_hip.hpp over the C-ABI of libicebin_hip.so.  Every declaration is
# `except +`: C++ exceptions (icebin::Exception, the stand-in for everytrace::Exception) surface as
# Python RuntimeError exactly as in the reference.
from libcpp cimport bool
from libcpp.string cimport string
from libcpp.vector cimport vector

cdef extern from "<array>" namespace "std" nogil:
    cdef cppclass array2i "std::array<int, 2>":
        int& operator[](size_t)
    cdef cppclass array2l "std::array<long, 2>":
        long& operator[](size_t)

cdef extern from "../host/icebin_hip.hpp" namespace "icebin":
    cdef cppclass ArrayViewCD "icebin::ArrayView<const double>":
        ArrayViewCD(const double *p, long n) except +
        ArrayViewCD(const double *p, long n0, long n1) except +

    cdef cppclass ExchangeGrid:
        vector[int] indices
        vector[double] overlaps
        ExchangeGrid() except +

    cdef cppclass AbbrGrid:
        long sparse_extent
        vector[long] dim_to_sparse
        vector[double] native_area
        AbbrGrid() except +

    cdef cppclass RegridMatrices:
        pass

    cdef cppclass GCMRegridder_Standard:
        bool correctA
        GCMRegridder_Standard() except +
        void init(AbbrGrid &&agridA, vector[double] &&hcdefs, array2l strides, bool correctA) except +
        unsigned long nA() except +
        unsigned long nE() except +
        unsigned int nhc() except +
        size_t add_sheet(const string &name, long nI, const ExchangeGrid &aexgrid, const vector[double] &gridA_proj_area,
                         int interp_style, const vector[double] &gridI_centroid_xy) except +
        vector[double] wA(const string &sheet_name, bool native, double fill) except +

cdef extern from "../host/icebin_hip.hpp" namespace "icebin::linear":
    cdef cppclass Weighted:
        bool conservative
        bool scaled
        array2i shape_d() except +
        array2l shape() except +
        long nnz() except +
        vector[long] dim_to_sparse(int k) except +
        const vector[double] &wM() except +
        const vector[double] &Mw() except +
        void M_coo(vector[int] &row, vector[int] &col, vector[double] &val) except +
        vector[double] apply(const ArrayViewCD &A_b, double fill, bool force_conservation) except +

cdef extern from "../host/icebin_hip.hpp" namespace "icebin::cython":
    # icebin_cython.hpp:70-87
    RegridMatrices *new_regrid_matrices(const GCMRegridder_Standard *gcm, const string &sheet_name,
                                        const double *elevmaskI, long elevmaskI_len, bool scale, bool correctA,
                                        double sigma_x, double sigma_y, double sigma_z, bool conserve) except +
    Weighted *RegridMatrices_matrix(RegridMatrices *cself, const string &spec_name) except +
