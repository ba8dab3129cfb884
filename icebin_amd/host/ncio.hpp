// ncio.hpp -- the on-disk side of the C++ host mirror: a self-contained NetCDF *classic* container (CDF-5 writer, CDF-1 / CDF-2 /
// CDF-5 reader, fixed-size variables) and an `NcIO` handle with the constructor shape of ibmisc::NcIO, so that callers written
// like modele/global_ec.cpp:539-605 (`NcIO ncio(ofname, 'w'); mat->ncio(ncio, "AvI", {"dimA", "dimI"});`) and
// GCMRegridder.cpp:104-150 (`gcm.ncio(ncio, "m")`) bind to icebin_hip.hpp.  Same variable / dimension / attribute layout as
// icebin_amd/ncio.py: a file written here is read by the Python side and vice versa (tests/test_ncio.py).
// No NetCDF library is part of this image and the reference's own container is NetCDF-4 / HDF5: a list-of-strings attribute
// (`m.info:sheets`, `BvA.info:dim_names`) is ONE char attribute with the names joined by ',' here.  Files of the reference's
// own build are READ when the translation unit defines ICEBIN_NCIO_HDF5 (pulls in hdf5.hpp, needs -lz); without the macro they
// are refused with a message naming it.  Header-only, C++14, host code only (no HIP, no torch).
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace icebin {
namespace nc {

enum Type { BYTE = 1, CHAR = 2, SHORT = 3, INT = 4, FLOAT = 5, DOUBLE = 6, UBYTE = 7, USHORT = 8, UINT = 9, INT64 = 10, UINT64 = 11 };
inline size_t type_size(int t) {
    switch (t) { case BYTE: case CHAR: case UBYTE: return 1; case SHORT: case USHORT: return 2; case INT: case FLOAT: case UINT: return 4;
                 case DOUBLE: case INT64: case UINT64: return 8; }
    throw std::runtime_error("ncio: unknown NetCDF type code " + std::to_string(t));
}
template <class T> struct type_of;
template <> struct type_of<int32_t> { static const int value = INT; };
template <> struct type_of<int64_t> { static const int value = INT64; };
template <> struct type_of<double> { static const int value = DOUBLE; };
template <> struct type_of<char> { static const int value = CHAR; };

/** typed array in host byte order */
struct Array {
    int type = INT;
    std::vector<char> raw;
    size_t size() const { return raw.size() / type_size(type); }
    template <class T> static Array of(const T *p, size_t n) {
        Array a; a.type = type_of<T>::value; a.raw.resize(n * sizeof(T));
        if (n) std::memcpy(a.raw.data(), p, n * sizeof(T));
        return a;
    }
    template <class T> static Array of(std::vector<T> const &v) { return of(v.data(), v.size()); }
    static Array str(std::string const &s) { return of(s.data(), s.size()); }
    /** element k converted to T (numeric types) */
    template <class T> T at(size_t k) const {
        const char *p = raw.data() + k * type_size(type);
        switch (type) {
            case BYTE: { int8_t v; std::memcpy(&v, p, 1); return (T)v; }
            case CHAR: case UBYTE: { uint8_t v; std::memcpy(&v, p, 1); return (T)v; }
            case SHORT: { int16_t v; std::memcpy(&v, p, 2); return (T)v; }
            case USHORT: { uint16_t v; std::memcpy(&v, p, 2); return (T)v; }
            case INT: { int32_t v; std::memcpy(&v, p, 4); return (T)v; }
            case UINT: { uint32_t v; std::memcpy(&v, p, 4); return (T)v; }
            case FLOAT: { float v; std::memcpy(&v, p, 4); return (T)v; }
            case DOUBLE: { double v; std::memcpy(&v, p, 8); return (T)v; }
            case INT64: { int64_t v; std::memcpy(&v, p, 8); return (T)v; }
            case UINT64: { uint64_t v; std::memcpy(&v, p, 8); return (T)v; }
        }
        throw std::runtime_error("ncio: unknown NetCDF type code");
    }
    template <class T> std::vector<T> as() const { std::vector<T> v(size()); for (size_t k = 0; k < v.size(); ++k) v[k] = at<T>(k); return v; }
    std::string str() const { return std::string(raw.begin(), raw.end()); }
};
typedef std::vector<std::pair<std::string, Array>> Attrs;      // ordered, as written
inline const Array *find(Attrs const &a, std::string const &k) { for (auto const &kv : a) if (kv.first == k) return &kv.second; return nullptr; }

struct Var {
    std::vector<std::string> dims;
    Array data;
    Attrs attrs;
    const Array &att(std::string const &k) const {
        const Array *a = find(attrs, k);
        if (!a) throw std::runtime_error("ncio: attribute '" + k + "' not found");
        return *a;
    }
};

class File;
#ifdef ICEBIN_NCIO_HDF5
File read_hdf5(std::string bytes);                              // hdf5.hpp, included at the end of this header
#endif

class File {
    static void swap_bytes(char *p, size_t n, size_t w) {
        if (w == 1) return;
        for (size_t k = 0; k < n; ++k) for (size_t b = 0; b < w / 2; ++b) std::swap(p[k * w + b], p[k * w + w - 1 - b]);
    }
    static bool little() { const uint16_t one = 1; return *reinterpret_cast<const uint8_t *>(&one) == 1; }
    static size_t pad4(size_t n) { return (4 - n % 4) % 4; }
public:
    std::vector<std::pair<std::string, int64_t>> dims;
    Attrs attrs;
    std::vector<std::pair<std::string, Var>> vars;

    int64_t dim(std::string const &name) const {
        for (auto const &d : dims) if (d.first == name) return d.second;
        throw std::runtime_error("ncio: dimension '" + name + "' not found");
    }
    std::string add_dim(std::string const &name, int64_t n) {
        for (auto const &d : dims) if (d.first == name) {
            if (d.second != n) throw std::runtime_error("ncio: dimension " + name + " redefined");
            return name;
        }
        dims.emplace_back(name, n);
        return name;
    }
    bool has_var(std::string const &name) const { for (auto const &v : vars) if (v.first == name) return true; return false; }
    Var const &var(std::string const &name) const {
        for (auto const &v : vars) if (v.first == name) return v.second;
        throw std::runtime_error("ncio: variable '" + name + "' not found");
    }
    Var &add_var(std::string const &name, std::vector<std::string> const &vdims, Array data, Attrs a = Attrs()) {
        if (has_var(name)) throw std::runtime_error("ncio: variable " + name + " defined twice");
        int64_t n = 1;
        for (auto const &d : vdims) n *= dim(d);
        if ((int64_t)data.size() != n) throw std::runtime_error("ncio: variable " + name + ": data size does not match its dimensions");
        Var v; v.dims = vdims; v.data = std::move(data); v.attrs = std::move(a);
        vars.emplace_back(name, std::move(v));
        return vars.back().second;
    }

    // ---- writer: CDF-5 (64-bit counts and offsets, has int64) -----------------------------------------------------------
    void write(std::string const &path) const {
        std::string out;
        auto put = [&](const void *p, size_t w) {                // one big-endian scalar
            char b[8]; std::memcpy(b, p, w);
            if (little()) swap_bytes(b, 1, w);
            out.append(b, w);
        };
        auto put_i32 = [&](int32_t v) { put(&v, 4); };
        auto put_i64 = [&](int64_t v) { put(&v, 8); };
        auto put_name = [&](std::string const &s) { put_i64((int64_t)s.size()); out += s; out.append(pad4(s.size()), '\0'); };
        auto put_array = [&](Array const &a) {
            std::string raw(a.raw.begin(), a.raw.end());
            if (little() && !raw.empty()) swap_bytes(&raw[0], a.size(), type_size(a.type));
            out += raw; out.append(pad4(raw.size()), '\0');
        };
        auto put_atts = [&](Attrs const &as) {
            if (as.empty()) { put_i32(0); put_i64(0); return; }
            put_i32(12); put_i64((int64_t)as.size());
            for (auto const &kv : as) { put_name(kv.first); put_i32(kv.second.type); put_i64((int64_t)kv.second.size()); put_array(kv.second); }
        };
        auto header = [&](std::vector<int64_t> const &begins) {
            out.clear();
            out += "CDF"; out += '\x05'; put_i64(0);
            if (dims.empty()) { put_i32(0); put_i64(0); }
            else { put_i32(10); put_i64((int64_t)dims.size()); for (auto const &d : dims) { put_name(d.first); put_i64(d.second); } }
            put_atts(attrs);
            if (vars.empty()) { put_i32(0); put_i64(0); return; }
            put_i32(11); put_i64((int64_t)vars.size());
            for (size_t k = 0; k < vars.size(); ++k) {
                Var const &v = vars[k].second;
                put_name(vars[k].first); put_i64((int64_t)v.dims.size());
                for (auto const &d : v.dims) {
                    int64_t id = -1;
                    for (size_t q = 0; q < dims.size(); ++q) if (dims[q].first == d) id = (int64_t)q;
                    put_i64(id);
                }
                put_atts(v.attrs);
                const int64_t nbytes = (int64_t)v.data.raw.size();
                put_i32(v.data.type); put_i64(nbytes + (int64_t)pad4((size_t)nbytes)); put_i64(begins[k]);
            }
        };
        std::vector<int64_t> begins(vars.size(), 0);
        header(begins);
        int64_t off = (int64_t)out.size();
        for (size_t k = 0; k < vars.size(); ++k) { begins[k] = off; const size_t nb = vars[k].second.data.raw.size(); off += (int64_t)(nb + pad4(nb)); }
        header(begins);
        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("ncio: cannot open " + path + " for writing");
        f.write(out.data(), (std::streamsize)out.size());
        for (auto const &kv : vars) { out.clear(); put_array(kv.second.data); f.write(out.data(), (std::streamsize)out.size()); }
        if (!f) throw std::runtime_error("ncio: write to " + path + " failed");
    }

    // ---- reader: CDF-1 / CDF-2 / CDF-5 ---------------------------------------------------------------------------------
    static File read(std::string const &path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("ncio: cannot open " + path);
        std::stringstream ss; ss << f.rdbuf();
        std::string buf = ss.str();
        bool h5sig = false;                                      // the HDF5 signature sits at 0, 512, 1024, ... (a user block may precede it)
        if (buf.compare(0, 3, "CDF") != 0)
            for (size_t at = 0; at + 8 <= buf.size() && !h5sig; at = at ? at * 2 : 512) h5sig = buf.compare(at, 8, "\x89HDF\r\n\x1a\n") == 0;
        if (h5sig) {
#ifdef ICEBIN_NCIO_HDF5
            try { return read_hdf5(std::move(buf)); }
            catch (std::exception const &e) { throw std::runtime_error("ncio: " + path + ": " + e.what()); }
#else
            throw std::runtime_error("ncio: " + path + " is a NetCDF-4/HDF5 file; compile with -DICEBIN_NCIO_HDF5 -lz (hdf5.hpp) or convert it with `nccopy -k cdf5`");
#endif
        }
        if (buf.size() < 8 || buf.compare(0, 3, "CDF") != 0 || (buf[3] != 1 && buf[3] != 2 && buf[3] != 5))
            throw std::runtime_error("ncio: " + path + " is not a NetCDF classic file (CDF-1/2/5)");
        const int ver = buf[3];
        size_t pos = 4;
        auto take = [&](size_t w) -> int64_t {                   // big-endian signed integer of w bytes
            if (pos + w > buf.size()) throw std::runtime_error("ncio: truncated header in " + path);
            int64_t v = 0;
            for (size_t b = 0; b < w; ++b) v = (v << 8) | (uint8_t)buf[pos + b];
            if (w == 4) v = (int32_t)v;
            pos += w;
            return v;
        };
        const size_t cw = ver == 5 ? 8 : 4, ow = ver == 1 ? 4 : 8;
        auto name = [&]() { const size_t n = (size_t)take(cw); std::string s = buf.substr(pos, n); pos += n + pad4(n); return s; };
        auto array = [&](int type, size_t n, size_t at) {
            Array a; a.type = type;
            const size_t nb = n * type_size(type);
            if (at + nb > buf.size()) throw std::runtime_error("ncio: truncated data in " + path);
            a.raw.assign(buf.begin() + (std::ptrdiff_t)at, buf.begin() + (std::ptrdiff_t)(at + nb));
            if (little() && nb) swap_bytes(a.raw.data(), n, type_size(type));
            return a;
        };
        auto atts = [&]() {
            Attrs out;
            const int64_t tag = take(4), n = take(cw);
            if (tag == 0) return out;
            if (tag != 12) throw std::runtime_error("ncio: corrupt header (attribute list) in " + path);
            for (int64_t k = 0; k < n; ++k) {
                std::string key = name();
                const int type = (int)take(4);
                const size_t ne = (size_t)take(cw);
                out.emplace_back(key, array(type, ne, pos));
                pos += ne * type_size(type) + pad4(ne * type_size(type));
            }
            return out;
        };
        File fl;
        const int64_t numrecs = take(cw);
        int64_t tag = take(4), n = take(cw);
        if (tag == 10) for (int64_t k = 0; k < n; ++k) { std::string d = name(); const int64_t ln = take(cw); fl.dims.emplace_back(d, ln ? ln : numrecs); }
        fl.attrs = atts();
        tag = take(4); n = take(cw);
        if (tag == 11) for (int64_t k = 0; k < n; ++k) {
            Var v;
            std::string vn = name();
            const int64_t nd = take(cw);
            int64_t cnt = 1;
            for (int64_t d = 0; d < nd; ++d) { const auto &dd = fl.dims.at((size_t)take(cw)); v.dims.push_back(dd.first); cnt *= dd.second; }
            v.attrs = atts();
            const int type = (int)take(4);
            (void)take(cw);
            const int64_t begin = take(ow);
            v.data = array(type, (size_t)cnt, (size_t)begin);
            fl.vars.emplace_back(vn, std::move(v));
        }
        return fl;
    }
};

inline std::vector<std::string> split_names(std::string const &s) {
    std::vector<std::string> out;
    std::string cur;
    for (char c : s) { if (c == ',') { if (!cur.empty()) out.push_back(cur); cur.clear(); } else cur += c; }
    if (!cur.empty()) out.push_back(cur);
    return out;
}
inline std::string join_names(std::vector<std::string> const &v) { std::string s; for (size_t k = 0; k < v.size(); ++k) s += (k ? "," : "") + v[k]; return s; }

}   // namespace nc
}   // namespace icebin
#ifdef ICEBIN_NCIO_HDF5
#include "hdf5.hpp"
namespace icebin { namespace nc { inline File read_hdf5(std::string bytes) { return h5::read_netcdf4(std::move(bytes)); } } }
#endif
namespace icebin {

/** ibmisc::NcIO as the regrid path uses it: NcIO ncio(fname, 'r' | 'w' | 'a'); objects ->ncio(ncio, vname, ...); the file is
    written when a 'w' / 'a' handle is closed or destroyed ('a' starts from the existing file, as global_ec.cpp:567-605 appends
    one matrix after the other).  The reference's format / compression arguments are accepted and ignored. */
class NcIO {
    bool dirty_ = false, closed_ = false;
public:
    char rw;
    std::string fname;
    nc::File file;
    NcIO(std::string const &_fname, char _rw = 'r', std::string const & /*format*/ = "", bool /*compress*/ = false) : rw(_rw), fname(_fname) {
        if (rw != 'r' && rw != 'w' && rw != 'a') throw std::runtime_error("NcIO: mode must be 'r', 'w' or 'a'");
        if (rw == 'r') file = nc::File::read(fname);
        else if (rw == 'a') { std::ifstream probe(fname, std::ios::binary); if (probe.good()) file = nc::File::read(fname); }
    }
    NcIO(NcIO const &) = delete;
    NcIO &operator=(NcIO const &) = delete;
    bool reading() const { return rw == 'r'; }
    void touch() { dirty_ = true; }
    void close() {
        if (!closed_ && rw != 'r' && dirty_) file.write(fname);
        closed_ = true;
    }
    void flush() { close(); }          // ibmisc::NcIO::flush
    ~NcIO() { try { close(); } catch (...) {} }
};

}   // namespace icebin
