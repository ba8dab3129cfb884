// CPU-only check of the C++ NetCDF-classic container (icebin_amd/host/ncio.hpp) against the Python one (icebin_amd/ncio.py):
//   test_ncio write out.nc   writes a file with every construct the regrid path uses;
//   test_ncio dump in.nc     prints a canonical dump (dims, attributes, variables with element counts and sums).
// tests/test_ncio.py has Python read what C++ wrote and C++ dump what Python wrote.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../icebin_amd/host/ncio.hpp"

using namespace icebin;

static void dump_array(nc::Array const &a) {
    if (a.type == nc::CHAR) { std::printf("char[%zu] '%s'", a.size(), a.str().c_str()); return; }
    double s = 0;
    for (size_t k = 0; k < a.size(); ++k) s += a.at<double>(k) * (double)(k % 7 + 1);
    std::printf("type%d[%zu] wsum=%.17g", a.type, a.size(), s);
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string mode = argv[1], path = argv[2];
    try {
        if (mode == "write") {
            NcIO ncio(path, 'w');
            nc::File &f = ncio.file;
            const int64_t ext = 1234567890123ll;
            const int32_t zero = 0;
            f.attrs.emplace_back("title", nc::Array::str("ncio.hpp round trip"));
            f.add_var("m.info", {}, nc::Array::of(&zero, 1), {{"correctA", nc::Array::of(&zero, 1)}, {"sheets", nc::Array::str("greenland,antarctica")}});
            std::vector<int64_t> d64 = {5, 1ll << 40, -3, 7};
            f.add_var("dimB", {f.add_dim("dimB.dense_extent", 4)}, nc::Array::of(d64), {{"sparse_extent", nc::Array::of(&ext, 1)}});
            std::vector<int32_t> ind = {0, 1, 2, 3, 4, 5};
            f.add_var("BvA.M.indices", {f.add_dim("BvA.M.nnz", 3), f.add_dim("BvA.M.rank", 2)}, nc::Array::of(ind));
            std::vector<double> val = {1.5, -2.25e-300, 3.0e300};
            f.add_var("BvA.M.values", {"BvA.M.nnz"}, nc::Array::of(val));
            f.add_var("odd.bytes", {f.add_dim("odd", 5)}, nc::Array::str("abcde"));        // 5 chars: padding inside the data section
            ncio.touch();
        } else {
            NcIO ncio(path, 'r');
            nc::File const &f = ncio.file;
            for (auto const &d : f.dims) std::printf("dim %s %lld\n", d.first.c_str(), (long long)d.second);
            for (auto const &a : f.attrs) { std::printf("gatt %s ", a.first.c_str()); dump_array(a.second); std::printf("\n"); }
            for (auto const &v : f.vars) {
                std::printf("var %s (", v.first.c_str());
                for (size_t k = 0; k < v.second.dims.size(); ++k) std::printf("%s%s", k ? "," : "", v.second.dims[k].c_str());
                std::printf(") ");
                dump_array(v.second.data);
                std::printf("\n");
                for (auto const &a : v.second.attrs) { std::printf("  att %s ", a.first.c_str()); dump_array(a.second); std::printf("\n"); }
            }
        }
    } catch (std::exception const &e) {
        std::printf("ERROR %s\n", e.what());
        return 1;
    }
    return 0;
}
