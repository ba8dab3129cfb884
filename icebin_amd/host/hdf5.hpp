// hdf5.hpp -- a self-contained READER for the NetCDF-4 / HDF5 files the reference itself writes (`NcIO(ofname, 'w', "nc4")`,
// modele/global_ec.cpp:539,567; slib/icebin/GCMRegridder.cpp:104-150), so that the C++ host mirror opens a real IceBin file
// without a Python or `nccopy` detour.  No HDF5 / NetCDF library is part of this image: this is a restatement of the published file
// format (the subset netcdf-c produces and icebin_amd/hdf5.py reads): superblocks 0-3, object headers 1 and 2 with continuation
// blocks, compact / dense (fractal heap) / old-style (symbol table) groups, compact / contiguous / chunked layouts (chunk B-tree
// version 1, single chunk), deflate / shuffle / Fletcher-32, fixed and variable-length strings, vlen sequences of references
// (DIMENSION_LIST), compact and dense attributes; every version-2 metadata block is verified against its Jenkins lookup3 checksum.
// On top: the NetCDF-4 conventions (dimension scales ordered by _Netcdf4Dimid, DIMENSION_LIST -> dimension names, hidden
// attributes dropped) -> the same nc::File the classic reader returns; a list-of-strings attribute becomes ONE char attribute
// with the names joined by ',' (what ncio.hpp's readers of `sheets` / `dim_names` split again).  Needs zlib (-lz).
// Read-only; header-only; host code.  tests/test_ncio.py compares it with the Python reader on the reference's example files.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace icebin {
namespace h5 {

struct Error : std::runtime_error { explicit Error(std::string const &m) : std::runtime_error("hdf5: " + m) {} };

// Bob Jenkins' lookup3 hashlittle(), the checksum of version-2 metadata (H5_checksum_lookup3)
inline uint32_t lookup3(const uint8_t *k, size_t length, uint32_t init = 0) {
    auto rot = [](uint32_t x, int r) { return (uint32_t)((x << r) | (x >> (32 - r))); };
    uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + (uint32_t)length + init;
    auto rd = [&](size_t i) { return (uint32_t)k[i] | ((uint32_t)k[i + 1] << 8) | ((uint32_t)k[i + 2] << 16) | ((uint32_t)k[i + 3] << 24); };
    while (length > 12) {
        a += rd(0); b += rd(4); c += rd(8);
        a -= c; a ^= rot(c, 4); c += b; b -= a; b ^= rot(a, 6); a += c; c -= b; c ^= rot(b, 8); b += a;
        a -= c; a ^= rot(c, 16); c += b; b -= a; b ^= rot(a, 19); a += c; c -= b; c ^= rot(b, 4); b += a;
        length -= 12; k += 12;
    }
    switch (length) {
        case 12: c += (uint32_t)k[11] << 24;  // fallthrough
        case 11: c += (uint32_t)k[10] << 16;  // fallthrough
        case 10: c += (uint32_t)k[9] << 8;    // fallthrough
        case 9: c += k[8];                    // fallthrough
        case 8: b += (uint32_t)k[7] << 24;    // fallthrough
        case 7: b += (uint32_t)k[6] << 16;    // fallthrough
        case 6: b += (uint32_t)k[5] << 8;     // fallthrough
        case 5: b += k[4];                    // fallthrough
        case 4: a += (uint32_t)k[3] << 24;    // fallthrough
        case 3: a += (uint32_t)k[2] << 16;    // fallthrough
        case 2: a += (uint32_t)k[1] << 8;     // fallthrough
        case 1: a += k[0]; break;
        case 0: return c;
    }
    c ^= b; c -= rot(b, 14); a ^= c; a -= rot(c, 11); b ^= a; b -= rot(a, 25); c ^= b; c -= rot(b, 16);
    a ^= c; a -= rot(c, 4); b ^= a; b -= rot(a, 14); c ^= b; c -= rot(b, 24);
    return c;
}
// H5_checksum_fletcher32: sums of big-endian 16-bit words, end-around carry
inline uint32_t fletcher32(const uint8_t *d, size_t n) {
    uint32_t s1 = 0, s2 = 0;
    size_t words = n / 2, i = 0;
    while (words) {
        size_t t = std::min<size_t>(words, 360);
        words -= t;
        for (; t; --t) { s1 += ((uint32_t)d[i] << 8) | d[i + 1]; s2 += s1; i += 2; }
        s1 = (s1 & 0xffff) + (s1 >> 16); s2 = (s2 & 0xffff) + (s2 >> 16);
    }
    if (n % 2) { s1 += (uint32_t)d[i] << 8; s2 += s1; s1 = (s1 & 0xffff) + (s1 >> 16); s2 = (s2 & 0xffff) + (s2 >> 16); }
    s1 = (s1 & 0xffff) + (s1 >> 16); s2 = (s2 & 0xffff) + (s2 >> 16);
    return (s2 << 16) | s1;
}

struct DType {
    enum Kind { NUM, STR, VSTR, VLEN, REF, RAW } kind = RAW;
    size_t size = 0;
    bool is_float = false, is_signed = false, big = false;
    std::shared_ptr<DType> base;
};
/** a decoded attribute / dataset: numbers (host byte order), strings, or per element a sequence of object references */
struct Value {
    bool null = false, is_str = false, is_refs = false;
    nc::Array arr;
    std::vector<std::string> strs;
    std::vector<std::vector<uint64_t>> refs;
    std::vector<uint64_t> shape;
};
struct Layout { int kind = -1; uint64_t a = 0, b = 0; std::vector<uint64_t> dims; bool has_fsize = false; uint64_t fsize = 0; uint32_t fmask = 0; };   // 0 compact, 1 contiguous, 2 chunked, 3 single chunk
struct Obj {
    uint64_t addr = 0;
    std::vector<std::pair<std::string, Value>> attrs;
    bool is_group = false;
    std::vector<std::pair<std::string, uint64_t>> links;
    bool has_shape = false, null_space = false, has_dtype = false, has_layout = false, has_fill = false;
    std::vector<uint64_t> shape;
    DType dtype;
    Layout layout;
    std::vector<std::pair<int, std::vector<uint32_t>>> filters;
    std::string fill;
    bool is_dataset() const { return has_dtype && has_layout; }
    const Value *attr(std::string const &k) const { for (auto const &kv : attrs) if (kv.first == k) return &kv.second; return nullptr; }
};

class File {
    std::string buf_;
    const uint8_t *b_;
    size_t n_;
    int O_ = 8, L_ = 8;
    uint64_t base_ = 0, root_addr_ = 0;
    bool has_root_stab_ = false;
    uint64_t root_stab_[2] = {0, 0};
    std::map<uint64_t, std::shared_ptr<Obj>> objs_;
    std::map<uint64_t, std::map<unsigned, std::string>> gcol_;

    void need(size_t pos, size_t n) const { if (pos > n_ || n > n_ - pos) throw Error("truncated or corrupt file (read past the end)"); }
    uint64_t u(size_t pos, int n) const {
        need(pos, (size_t)n);
        uint64_t v = 0;
        for (int i = n - 1; i >= 0; --i) v = (v << 8) | b_[pos + (size_t)i];
        return v;
    }
    uint64_t off(size_t pos) const { return u(pos, O_); }
    uint64_t len(size_t pos) const { return u(pos, L_); }
    uint64_t undef() const { return O_ >= 8 ? ~0ull : ((1ull << (8 * O_)) - 1); }
    bool sig(size_t pos, const char *s) const { need(pos, 4); return std::memcmp(b_ + pos, s, 4) == 0; }
    void check(size_t start, size_t end) {           // bytes [start, end) are followed by their lookup3 checksum
        need(start, end - start + 4);
        const uint32_t want = (uint32_t)u(end, 4), got = lookup3(b_ + start, end - start);
        if (want != got) throw Error("metadata checksum mismatch");
        ++checked_lookup3;
    }
    size_t cstr_end(size_t p) const { while (p < n_ && b_[p]) ++p; if (p >= n_) throw Error("unterminated name"); return p; }
    static int bitlen(uint64_t v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

    void superblock() {
        static const uint8_t SIG[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        size_t pos = 0;
        while (true) {
            if (pos + 8 <= n_ && std::memcmp(b_ + pos, SIG, 8) == 0) break;
            pos = pos == 0 ? 512 : pos * 2;
            if (pos >= n_) throw Error("not an HDF5 file (no superblock signature)");
        }
        const int ver = b_[pos + 8];
        if (ver == 0 || ver == 1) {
            need(pos, 64);
            O_ = b_[pos + 13]; L_ = b_[pos + 14];
            size_t p = pos + 24 + (ver == 1 ? 4 : 0);
            base_ = off(p);
            p += 4 * (size_t)O_;
            root_addr_ = off(p + (size_t)O_);
            const uint32_t cache_type = (uint32_t)u(p + 2 * (size_t)O_, 4);
            if (cache_type == 1) { const size_t sp = p + 2 * (size_t)O_ + 8; has_root_stab_ = true; root_stab_[0] = off(sp); root_stab_[1] = off(sp + (size_t)O_); }
        } else if (ver == 2 || ver == 3) {
            need(pos, 12);
            O_ = b_[pos + 9]; L_ = b_[pos + 10];
            const size_t p = pos + 12;
            base_ = off(p);
            root_addr_ = off(p + 3 * (size_t)O_);
            check(pos, p + 4 * (size_t)O_);
        } else throw Error("superblock version " + std::to_string(ver) + " not supported");
        if (O_ < 2 || O_ > 8 || L_ < 2 || L_ > 8) throw Error("offset / length sizes not supported");
        if (base_ != 0 && base_ != pos) throw Error("base address not supported");
        base_ = base_ == pos ? pos : 0;
    }

    struct Msg { int type; int flags; size_t d; size_t size; };
    std::vector<Msg> messages(uint64_t addr) {
        std::vector<Msg> out;
        const size_t a = (size_t)(addr + base_);
        need(a, 16);
        if (sig(a, "OHDR")) {
            if (b_[a + 4] != 2) throw Error("object header version");
            const int flags = b_[a + 5];
            size_t p = a + 6;
            if (flags & 0x20) p += 16;
            if (flags & 0x10) p += 4;
            const int nsz = 1 << (flags & 3);
            const size_t chunk0 = (size_t)u(p, nsz);
            p += (size_t)nsz;
            check(a, p + chunk0);
            std::vector<std::pair<size_t, size_t>> blocks;
            blocks.emplace_back(p, p + chunk0);
            const size_t hdr = 4 + ((flags & 0x04) ? 2 : 0);
            for (size_t bi = 0; bi < blocks.size(); ++bi) {
                size_t q = blocks[bi].first;
                const size_t end = blocks[bi].second;
                while (q + hdr <= end) {
                    const int mtype = b_[q], mflags = b_[q + 3];
                    const size_t msize = (size_t)u(q + 1, 2), d = q + hdr;
                    if (d + msize > end) break;
                    if (mtype == 0x10) {
                        const size_t coff = (size_t)(off(d) + base_), clen = (size_t)len(d + (size_t)O_);
                        if (!sig(coff, "OCHK")) throw Error("continuation block without OCHK signature");
                        check(coff, coff + clen - 4);
                        blocks.emplace_back(coff + 4, coff + clen - 4);
                    } else if (mtype != 0) out.push_back(Msg{mtype, mflags, d, msize});
                    q = d + msize;
                }
            }
        } else {
            if (b_[a] != 1) throw Error("no object header at " + std::to_string(addr));
            const int nmsg = (int)u(a + 2, 2);
            const size_t size = (size_t)u(a + 8, 4);
            std::vector<std::pair<size_t, size_t>> blocks;
            blocks.emplace_back(a + 16, a + 16 + size);
            int seen = 0;
            for (size_t bi = 0; bi < blocks.size() && seen < nmsg; ++bi) {
                size_t q = blocks[bi].first;
                const size_t end = blocks[bi].second;
                while (q + 8 <= end && seen < nmsg) {
                    const int mtype = (int)u(q, 2), mflags = b_[q + 4];
                    const size_t msize = (size_t)u(q + 2, 2), d = q + 8;
                    ++seen;
                    if (mtype == 0x10) { const size_t coff = (size_t)(off(d) + base_), clen = (size_t)len(d + (size_t)O_); blocks.emplace_back(coff, coff + clen); }
                    else if (mtype != 0) out.push_back(Msg{mtype, mflags, d, msize});
                    q = d + msize;
                }
            }
        }
        return out;
    }

    size_t parse_datatype(size_t pos, DType &t) const {
        need(pos, 8);
        const int cv = b_[pos], b0 = b_[pos + 1], b1 = b_[pos + 2];
        const size_t size = (size_t)u(pos + 4, 4);
        const int cls = cv & 0x0F, ver = cv >> 4;
        size_t p = pos + 8;
        t = DType();
        t.size = size;
        switch (cls) {
            case 0:
                p += 4;
                if (size == 1 || size == 2 || size == 4 || size == 8) { t.kind = DType::NUM; t.big = b0 & 1; t.is_signed = (b0 & 8) != 0; }
                return p - pos;
            case 1:
                p += 12;
                if (size == 4 || size == 8) { t.kind = DType::NUM; t.big = b0 & 1; t.is_float = true; }
                return p - pos;
            case 2: return p + 2 - pos;
            case 3: t.kind = DType::STR; return p - pos;
            case 4: return p + 4 - pos;
            case 5: return p + (size_t)((b0 + 7) & ~7) - pos;
            case 6: {
                const int nmemb = b0 | (b1 << 8);
                for (int m = 0; m < nmemb; ++m) {
                    const size_t e = cstr_end(p), name_len = e - p + 1;
                    if (ver < 3) { p += (name_len + 7) & ~(size_t)7; p += 4; if (ver == 1) p += 1 + 3 + 4 + 4 + 16; }
                    else { p += name_len; int nb = 1; while (nb < 8 && (size >> (8 * nb))) ++nb; p += (size_t)nb; }
                    DType mt;
                    p += parse_datatype(p, mt);
                }
                return p - pos;
            }
            case 7: t.kind = DType::REF; return p - pos;
            case 8: {
                const int nmemb = b0 | (b1 << 8);
                DType bt;
                p += parse_datatype(p, bt);
                for (int m = 0; m < nmemb; ++m) { const size_t e = cstr_end(p), name_len = e - p + 1; p += ver < 3 ? ((name_len + 7) & ~(size_t)7) : name_len; }
                p += (size_t)nmemb * bt.size;
                t = bt; t.size = size;
                return p - pos;
            }
            case 9: {
                auto bt = std::make_shared<DType>();
                p += parse_datatype(p, *bt);
                if ((b0 & 0x0F) == 1) t.kind = DType::VSTR; else { t.kind = DType::VLEN; t.base = bt; }
                return p - pos;
            }
            case 10: {
                need(p, 1);
                const int rank = b_[p];
                p += 1 + (ver < 3 ? 3 : 0) + 4 * (size_t)rank + (ver < 3 ? 4 * (size_t)rank : 0);
                DType bt;
                return p + parse_datatype(p, bt) - pos;
            }
        }
        throw Error("datatype class " + std::to_string(cls) + " not supported");
    }

    std::pair<std::string, std::pair<bool, uint64_t>> parse_link(size_t d) const {
        need(d, 4);
        if (b_[d] != 1) throw Error("link message version");
        const int fl = b_[d + 1];
        size_t p = d + 2;
        int ltype = 0;
        if (fl & 0x08) { ltype = b_[p]; p += 1; }
        if (fl & 0x04) p += 8;
        if (fl & 0x10) p += 1;
        const int nsz = 1 << (fl & 3);
        const size_t n = (size_t)u(p, nsz);
        p += (size_t)nsz;
        need(p, n);
        std::string name(reinterpret_cast<const char *>(b_ + p), n);
        p += n;
        if (ltype == 0) return {name, {true, off(p)}};
        return {name, {false, 0}};                               // soft / external links are not followed
    }
    size_t link_end(size_t q) const {
        const int fl = b_[q + 1];
        size_t p = q + 2 + ((fl & 0x08) ? 1 : 0) + ((fl & 0x04) ? 8 : 0) + ((fl & 0x10) ? 1 : 0);
        const int nsz = 1 << (fl & 3);
        const size_t n = (size_t)u(p, nsz);
        p += (size_t)nsz + n;
        const int ltype = (fl & 0x08) ? b_[q + 2] : 0;
        if (ltype == 0) return p + (size_t)O_;
        return p + 2 + (size_t)u(p, 2);
    }

    void walk_group_btree(uint64_t btree, uint64_t heap, std::vector<std::pair<std::string, uint64_t>> &out) const {
        const size_t h = (size_t)(heap + base_);
        if (!sig(h, "HEAP")) throw Error("local heap signature missing");
        const size_t data = (size_t)(off(h + 8 + 2 * (size_t)L_) + base_);
        walk_group_node(btree, data, out, 0);
    }
    void walk_group_node(uint64_t addr, size_t data, std::vector<std::pair<std::string, uint64_t>> &out, int depth) const {
        if (depth > 32) throw Error("group B-tree too deep");
        const size_t a = (size_t)(addr + base_);
        if (!sig(a, "TREE") || b_[a + 4] != 0) throw Error("group B-tree node expected");
        const int level = b_[a + 5], n = (int)u(a + 6, 2);
        size_t p = a + 8 + 2 * (size_t)O_;
        for (int i = 0; i < n; ++i) {
            const uint64_t child = off(p + (size_t)L_);
            p += (size_t)L_ + (size_t)O_;
            if (level > 0) walk_group_node(child, data, out, depth + 1);
            else {
                const size_t s = (size_t)(child + base_);
                if (!sig(s, "SNOD")) throw Error("symbol table node expected");
                size_t q = s + 8;
                const int ne = (int)u(s + 6, 2);
                for (int k = 0; k < ne; ++k) {
                    const uint64_t noff = off(q), haddr = off(q + (size_t)O_);
                    const size_t e = cstr_end(data + (size_t)noff);
                    out.emplace_back(std::string(reinterpret_cast<const char *>(b_ + data + noff), e - (data + (size_t)noff)), haddr);
                    q += 2 * (size_t)O_ + 24;
                }
            }
        }
    }

    // fractal heaps (dense links / attributes): the start of every managed object, in block order
    struct Heap { int width; uint64_t start_size, max_direct; int off_bytes; int flags; int dhdr; bool links; };
    void heap_direct(Heap const &hp, uint64_t baddr, uint64_t size, std::vector<size_t> &out) {
        const size_t b = (size_t)(baddr + base_);
        need(b, (size_t)size);
        if (!sig(b, "FHDB")) throw Error("fractal heap direct block expected");
        if (hp.flags & 2) {                                      // checksum of the whole block with the checksum field zeroed
            const size_t cpos = b + 5 + (size_t)O_ + (size_t)hp.off_bytes;
            std::string blk(reinterpret_cast<const char *>(b_ + b), (size_t)size);
            if (cpos + 4 > b + (size_t)size) throw Error("corrupt fractal heap direct block");
            std::memset(&blk[cpos - b], 0, 4);
            const uint32_t stored = (uint32_t)u(cpos, 4);
            if (lookup3(reinterpret_cast<const uint8_t *>(blk.data()), blk.size()) != stored) throw Error("fractal heap direct block checksum mismatch");
            ++checked_lookup3;
        }
        size_t q = b + (size_t)hp.dhdr;
        const size_t end = b + (size_t)size;
        while (q < end) {
            const int v = b_[q];
            if (hp.links ? v != 1 : (v < 1 || v > 3)) break;
            out.push_back(q);
            q = hp.links ? link_end(q) : attribute_extent(q);
        }
    }
    void heap_indirect(Heap const &hp, uint64_t iaddr, int nrows, std::vector<size_t> &out, int depth) {
        if (depth > 16) throw Error("fractal heap too deep");
        const size_t b = (size_t)(iaddr + base_);
        if (!sig(b, "FHIB")) throw Error("fractal heap indirect block expected");
        size_t q = b + 5 + (size_t)O_ + (size_t)hp.off_bytes;
        const int max_drows = (bitlen(hp.max_direct) - 1) - (bitlen(hp.start_size) - 1) + 2;
        for (int r = 0; r < nrows; ++r) {
            const uint64_t size = hp.start_size << std::max(0, r - 1);
            for (int c = 0; c < hp.width; ++c) {
                const uint64_t child = off(q);
                q += (size_t)O_;
                if (child == undef()) continue;
                if (r < max_drows) heap_direct(hp, child, size, out);
                else heap_indirect(hp, child, bitlen(size / (uint64_t)hp.width) - 1 - (bitlen(hp.start_size) - 1) + 1, out, depth + 1);
            }
        }
    }
    std::vector<size_t> heap_objects(uint64_t addr, bool links) {
        std::vector<size_t> out;
        const size_t a = (size_t)(addr + base_);
        if (!sig(a, "FRHP")) throw Error("fractal heap header expected");
        size_t p = a + 5;
        p += 2;
        const int filt_len = (int)u(p, 2); p += 2;
        const int flags = b_[p]; p += 1;
        p += 4;
        p += (size_t)L_ + (size_t)O_;
        p += (size_t)L_ + (size_t)O_;
        p += 4 * (size_t)L_;
        p += 4 * (size_t)L_;
        Heap hp;
        hp.links = links; hp.flags = flags;
        hp.width = (int)u(p, 2); p += 2;
        hp.start_size = len(p); p += (size_t)L_;
        hp.max_direct = len(p); p += (size_t)L_;
        const int max_heap_bits = (int)u(p, 2); p += 2;
        p += 2;
        const uint64_t root = off(p); p += (size_t)O_;
        const int cur_rows = (int)u(p, 2); p += 2;
        if (filt_len) throw Error("filtered fractal heaps are not supported");
        check(a, p);
        hp.off_bytes = (max_heap_bits + 7) / 8;
        hp.dhdr = 5 + O_ + hp.off_bytes + ((flags & 2) ? 4 : 0);
        if (hp.width <= 0 || hp.start_size == 0) throw Error("corrupt fractal heap header");
        if (root == undef()) return out;
        if (cur_rows == 0) heap_direct(hp, root, hp.start_size, out);
        else heap_indirect(hp, root, cur_rows, out, 0);
        return out;
    }

    Layout parse_layout(size_t d) const {
        need(d, 8);
        Layout lo;
        const int ver = b_[d];
        int cls = b_[d + 1];
        if (ver == 1 || ver == 2) {
            const int nd = b_[d + 1];
            cls = b_[d + 2];
            size_t p = d + 8;
            uint64_t addr = 0;
            if (cls != 0) { addr = off(p); p += (size_t)O_; }
            std::vector<uint64_t> dims;
            for (int i = 0; i < nd; ++i) dims.push_back(u(p + 4 * (size_t)i, 4));
            p += 4 * (size_t)nd;
            if (cls == 0) { lo.kind = 0; lo.a = p + 4; lo.b = u(p, 4); return lo; }
            if (cls == 1) { lo.kind = 1; lo.a = addr; return lo; }
            if (cls == 2) { lo.kind = 2; lo.a = addr; lo.dims = dims; return lo; }
        } else if (ver == 3 || ver == 4) {
            if (cls == 0) { lo.kind = 0; lo.a = d + 4; lo.b = u(d + 2, 2); return lo; }
            if (cls == 1) { lo.kind = 1; lo.a = off(d + 2); lo.b = len(d + 2 + (size_t)O_); return lo; }
            if (cls == 2 && ver == 3) {
                const int nd = b_[d + 2];
                lo.kind = 2; lo.a = off(d + 3);
                for (int i = 0; i < nd; ++i) lo.dims.push_back(u(d + 3 + (size_t)O_ + 4 * (size_t)i, 4));
                return lo;
            }
            if (cls == 2) {
                const int fl = b_[d + 2], nd = b_[d + 3], enc = b_[d + 4];
                size_t p = d + 5;
                for (int i = 0; i < nd; ++i) lo.dims.push_back(u(p + (size_t)enc * (size_t)i, enc));
                p += (size_t)enc * (size_t)nd;
                const int itype = b_[p]; p += 1;
                if (itype != 1) throw Error("chunk index type " + std::to_string(itype) + " (layout version 4) is not supported");
                if (fl & 2) { lo.has_fsize = true; lo.fsize = len(p); lo.fmask = (uint32_t)u(p + (size_t)L_, 4); p += (size_t)L_ + 4; }
                lo.kind = 3; lo.a = off(p);
                return lo;
            }
        }
        throw Error("data layout version " + std::to_string(ver) + " class " + std::to_string(cls) + " not supported");
    }
    std::vector<std::pair<int, std::vector<uint32_t>>> parse_filters(size_t d) const {
        std::vector<std::pair<int, std::vector<uint32_t>>> out;
        const int ver = b_[d], n = b_[d + 1];
        size_t p = d + (ver == 1 ? 8 : 2);
        for (int k = 0; k < n; ++k) {
            const int fid = (int)u(p, 2); p += 2;
            size_t nlen = 0;
            if (ver == 1 || fid >= 256) { nlen = (size_t)u(p, 2); p += 2; }
            p += 2;
            const int ncd = (int)u(p, 2); p += 2;
            p += ver == 1 ? ((nlen + 7) & ~(size_t)7) : nlen;
            std::vector<uint32_t> cd;
            for (int i = 0; i < ncd; ++i) cd.push_back((uint32_t)u(p + 4 * (size_t)i, 4));
            p += 4 * (size_t)ncd;
            if (ver == 1 && (ncd & 1)) p += 4;
            out.emplace_back(fid, cd);
        }
        return out;
    }
    std::string unfilter(std::string raw, Obj const &o, uint32_t mask) {
        for (int i = (int)o.filters.size() - 1; i >= 0; --i) {
            if (mask & (1u << i)) continue;
            const int fid = o.filters[(size_t)i].first;
            auto const &cd = o.filters[(size_t)i].second;
            if (fid == 1) {
                z_stream zs;
                std::memset(&zs, 0, sizeof(zs));
                if (inflateInit(&zs) != Z_OK) throw Error("zlib init failed");
                std::string out;
                out.resize(std::max<size_t>(raw.size() * 4, 1 << 16));
                zs.next_in = reinterpret_cast<Bytef *>(&raw[0]); zs.avail_in = (uInt)raw.size();
                size_t have = 0;
                int rc;
                do {
                    if (have == out.size()) out.resize(out.size() * 2);
                    zs.next_out = reinterpret_cast<Bytef *>(&out[have]); zs.avail_out = (uInt)(out.size() - have);
                    rc = inflate(&zs, Z_NO_FLUSH);
                    have = out.size() - zs.avail_out;
                } while (rc == Z_OK);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END) throw Error("corrupt deflate stream in a chunk");
                out.resize(have);
                raw.swap(out);
            } else if (fid == 2) {
                const size_t es = cd.empty() ? o.dtype.size : cd[0];
                if (es > 1) {
                    const size_t n = raw.size() / es;
                    std::string out(raw);
                    for (size_t k = 0; k < n; ++k) for (size_t bb = 0; bb < es; ++bb) out[k * es + bb] = raw[bb * n + k];
                    raw.swap(out);
                }
            } else if (fid == 3) {
                if (raw.size() < 4) throw Error("chunk too short for its Fletcher-32 checksum");
                uint32_t want = 0;
                for (int k = 3; k >= 0; --k) want = (want << 8) | (uint8_t)raw[raw.size() - 4 + (size_t)k];
                raw.resize(raw.size() - 4);
                const uint32_t got = fletcher32(reinterpret_cast<const uint8_t *>(raw.data()), raw.size());
                const uint32_t sw = ((got & 0x00FF00FFu) << 8) | ((got & 0xFF00FF00u) >> 8);
                if (got != want && sw != want) throw Error("chunk Fletcher-32 mismatch");
                ++checked_fletcher32;
            } else throw Error("filter " + std::to_string(fid) + " is not supported");
        }
        return raw;
    }
    struct Chunk { std::vector<uint64_t> offs; uint64_t addr, size; uint32_t mask; };
    void chunks(uint64_t btree, int rank, std::vector<Chunk> &out, int depth) const {
        if (btree == undef()) return;
        if (depth > 32) throw Error("chunk B-tree too deep");
        const size_t a = (size_t)(btree + base_);
        if (!sig(a, "TREE") || b_[a + 4] != 1) throw Error("chunk B-tree node expected");
        const int level = b_[a + 5], n = (int)u(a + 6, 2);
        size_t p = a + 8 + 2 * (size_t)O_;
        const size_t key = 8 + 8 * ((size_t)rank + 1);
        for (int k = 0; k < n; ++k) {
            Chunk c;
            c.size = (uint32_t)u(p, 4); c.mask = (uint32_t)u(p + 4, 4);
            for (int i = 0; i < rank; ++i) c.offs.push_back(u(p + 8 + 8 * (size_t)i, 8));
            c.addr = off(p + key);
            p += key + (size_t)O_;
            if (level > 0) chunks(c.addr, rank, out, depth + 1); else out.push_back(c);
        }
    }

    std::string global_heap_object(uint64_t addr, unsigned idx) {
        auto it = gcol_.find(addr);
        if (it == gcol_.end()) {
            const size_t a = (size_t)(addr + base_);
            if (!sig(a, "GCOL")) throw Error("global heap collection expected");
            const size_t size = (size_t)len(a + 8);
            size_t p = a + 8 + (size_t)L_;
            const size_t end = a + size;
            std::map<unsigned, std::string> objs;
            while (p + 8 + (size_t)L_ <= end) {
                const unsigned id = (unsigned)u(p, 2);
                const size_t n = (size_t)len(p + 8);
                if (id == 0) break;
                need(p + 8 + (size_t)L_, n);
                objs[id] = std::string(reinterpret_cast<const char *>(b_ + p + 8 + (size_t)L_), n);
                p += 8 + (size_t)L_ + ((n + 7) & ~(size_t)7);
            }
            it = gcol_.emplace(addr, std::move(objs)).first;
        }
        auto jt = it->second.find(idx);
        if (jt == it->second.end()) throw Error("global heap object missing");
        return jt->second;
    }
    std::string vlen_bytes(const uint8_t *desc) {
        uint64_t n = 0, addr = 0, idx = 0;
        for (int i = 3; i >= 0; --i) n = (n << 8) | desc[i];
        for (int i = O_ - 1; i >= 0; --i) addr = (addr << 8) | desc[4 + i];
        for (int i = 3; i >= 0; --i) idx = (idx << 8) | desc[4 + O_ + i];
        if (n == 0 || addr == 0) return std::string();
        return global_heap_object(addr, (unsigned)idx);
    }

    static int nc_type(DType const &t) {
        if (t.is_float) return t.size == 4 ? nc::FLOAT : nc::DOUBLE;
        switch (t.size) {
            case 1: return t.is_signed ? nc::BYTE : nc::UBYTE;
            case 2: return t.is_signed ? nc::SHORT : nc::USHORT;
            case 4: return t.is_signed ? nc::INT : nc::UINT;
            default: return t.is_signed ? nc::INT64 : nc::UINT64;
        }
    }
    static bool host_little() { const uint16_t one = 1; return *reinterpret_cast<const uint8_t *>(&one) == 1; }
    /** product of extents (times an element size) with overflow turned into an Error: the sizes come from an untrusted file */
    static uint64_t checked_count(std::vector<uint64_t> const &shape, uint64_t es = 1) {
        uint64_t count = es;
        for (auto s : shape) {
            if (s != 0 && count > UINT64_MAX / s) throw Error("dataspace too large (extent product overflows)");
            count *= s;
        }
        return count;
    }
    Value decode(const uint8_t *raw, size_t nbytes, DType const &dt, std::vector<uint64_t> const &shape) {
        Value v;
        v.shape = shape;
        const uint64_t count = checked_count(shape);
        if (count != 0 && dt.size > UINT64_MAX / count) throw Error("dataspace too large (byte size overflows)");
        if (dt.kind != DType::NUM && dt.kind != DType::STR && dt.kind != DType::VSTR && dt.kind != DType::VLEN && dt.kind != DType::REF) {
            v.arr.type = nc::UBYTE;                            // compound / opaque / array: raw bytes
            v.arr.raw.assign(reinterpret_cast<const char *>(raw), reinterpret_cast<const char *>(raw) + std::min<size_t>(nbytes, (size_t)count * dt.size));
            return v;
        }
        if ((uint64_t)nbytes < count * dt.size) throw Error("truncated data");
        if (dt.kind == DType::NUM) {
            v.arr.type = nc_type(dt);
            v.arr.raw.assign(reinterpret_cast<const char *>(raw), reinterpret_cast<const char *>(raw) + count * dt.size);
            if (dt.size > 1 && dt.big == host_little())
                for (uint64_t k = 0; k < count; ++k) std::reverse(v.arr.raw.begin() + (std::ptrdiff_t)(k * dt.size), v.arr.raw.begin() + (std::ptrdiff_t)((k + 1) * dt.size));
            return v;
        }
        if (dt.kind == DType::STR) {
            v.is_str = true;
            for (uint64_t k = 0; k < count; ++k) {
                std::string s(reinterpret_cast<const char *>(raw + k * dt.size), dt.size);
                s = s.substr(0, s.find('\0'));
                v.strs.push_back(s);
            }
            return v;
        }
        if (dt.kind == DType::VSTR) {
            v.is_str = true;
            for (uint64_t k = 0; k < count; ++k) { std::string s = vlen_bytes(raw + 16 * k); v.strs.push_back(s.substr(0, s.find('\0'))); }
            return v;
        }
        if (dt.kind == DType::VLEN) {
            v.is_refs = true;
            for (uint64_t k = 0; k < count; ++k) {
                const std::string s = vlen_bytes(raw + 16 * k);
                std::vector<uint64_t> r;
                const size_t es = dt.base ? dt.base->size : 0;
                if (dt.base && dt.base->kind == DType::REF && es == (size_t)O_)
                    for (size_t q = 0; q + es <= s.size(); q += es) { uint64_t a = 0; for (int i = (int)es - 1; i >= 0; --i) a = (a << 8) | (uint8_t)s[q + (size_t)i]; r.push_back(a); }
                v.refs.push_back(r);
            }
            return v;
        }
        v.is_refs = true;                                        // plain references
        for (uint64_t k = 0; k < count; ++k) { uint64_t a = 0; for (int i = (int)dt.size - 1; i >= 0; --i) a = (a << 8) | raw[k * dt.size + (size_t)i]; v.refs.push_back({a}); }
        return v;
    }

    struct AttrParts { int ver; std::string name; size_t tpos, spos, p; };
    AttrParts attribute_parts(size_t d) const {
        need(d, 9);
        AttrParts ap;
        ap.ver = b_[d];
        const size_t nsz = (size_t)u(d + 2, 2), tsz = (size_t)u(d + 4, 2), ssz = (size_t)u(d + 6, 2);
        size_t p = d + 8 + (ap.ver == 3 ? 1 : 0);
        auto pad = [&](size_t n) { return ap.ver == 1 ? ((n + 7) & ~(size_t)7) : n; };
        need(p, nsz);
        std::string name(reinterpret_cast<const char *>(b_ + p), nsz);
        ap.name = name.substr(0, name.find('\0'));
        p += pad(nsz);
        ap.tpos = p; p += pad(tsz);
        ap.spos = p; p += pad(ssz);
        ap.p = p;
        return ap;
    }
    bool attribute_space(size_t spos, std::vector<uint64_t> &shape) const {      // false: null dataspace
        need(spos, 4);
        const int sver = b_[spos], rank = b_[spos + 1];
        size_t q;
        if (sver == 1) q = spos + 8;
        else { q = spos + 4; if (b_[spos + 3] == 2) return false; }
        shape.clear();
        for (int i = 0; i < rank; ++i) shape.push_back(len(q + (size_t)i * (size_t)L_));
        return true;
    }
    size_t attribute_extent(size_t d) const {
        const AttrParts ap = attribute_parts(d);
        if (ap.ver < 1 || ap.ver > 3) throw Error("attribute message version");
        if (ap.ver > 1 && (b_[d + 1] & 3)) throw Error("shared attribute components are not supported");
        DType dt;
        parse_datatype(ap.tpos, dt);
        std::vector<uint64_t> shape;
        uint64_t count = 0;
        if (attribute_space(ap.spos, shape)) count = checked_count(shape);
        if (count != 0 && dt.size > UINT64_MAX / count) throw Error("attribute too large (byte size overflows)");
        return ap.p + (size_t)(count * dt.size);
    }
    std::pair<std::string, Value> parse_attribute(size_t d) {
        const AttrParts ap = attribute_parts(d);
        if (ap.ver < 1 || ap.ver > 3) throw Error("attribute message version");
        if (ap.ver > 1 && (b_[d + 1] & 3)) throw Error("shared attribute components are not supported");
        DType dt;
        parse_datatype(ap.tpos, dt);
        std::vector<uint64_t> shape;
        if (!attribute_space(ap.spos, shape)) { Value v; v.null = true; return {ap.name, v}; }
        const uint64_t count = checked_count(shape);
        if (count != 0 && dt.size > UINT64_MAX / count) throw Error("attribute too large (byte size overflows)");
        need(ap.p, (size_t)(count * dt.size));
        return {ap.name, decode(b_ + ap.p, (size_t)(count * dt.size), dt, shape)};
    }

    void parse_header(Obj &o) {
        bool have_dense_links = false, have_dense_attrs = false, have_stab = false;
        uint64_t dense_links = 0, dense_attrs = 0, stab[2] = {0, 0};
        for (Msg const &m : messages(o.addr)) {
            const size_t d = m.d;
            if ((m.flags & 0x02) && (m.type == 0x01 || m.type == 0x03 || m.type == 0x05 || m.type == 0x0B)) throw Error("shared header messages are not supported");
            switch (m.type) {
                case 0x01: {
                    const int ver = b_[d], rank = b_[d + 1], fl = b_[d + 2];
                    size_t p = d + (ver == 1 ? 8 : 4);
                    o.shape.clear();
                    for (int i = 0; i < rank; ++i) o.shape.push_back(len(p + (size_t)i * (size_t)L_));
                    (void)fl;
                    o.has_shape = true;
                    if (ver == 2 && b_[d + 3] == 2) o.null_space = true;
                    break;
                }
                case 0x03: parse_datatype(d, o.dtype); o.has_dtype = true; break;
                case 0x04:
                    if (!o.has_fill) { const size_t n = (size_t)u(d, 4); if (n) { need(d + 4, n); o.fill.assign(reinterpret_cast<const char *>(b_ + d + 4), n); o.has_fill = true; } }
                    break;
                case 0x05: {
                    const int ver = b_[d];
                    if (ver == 1 || ver == 2) {
                        if (ver == 1 || b_[d + 3]) { const size_t n = (size_t)u(d + 4, 4); o.has_fill = n > 0; o.fill.clear(); if (n) { need(d + 8, n); o.fill.assign(reinterpret_cast<const char *>(b_ + d + 8), n); } }
                    } else if (ver == 3 && (b_[d + 1] & 0x20)) { const size_t n = (size_t)u(d + 2, 4); o.has_fill = n > 0; o.fill.clear(); if (n) { need(d + 6, n); o.fill.assign(reinterpret_cast<const char *>(b_ + d + 6), n); } }
                    break;
                }
                case 0x06: { auto l = parse_link(d); o.is_group = true; if (l.second.first) o.links.emplace_back(l.first, l.second.second); break; }
                case 0x02: {
                    const int fl = b_[d + 1];
                    const size_t p = d + 2 + ((fl & 1) ? 8 : 0);
                    const uint64_t heap = off(p);
                    o.is_group = true;
                    if (heap != undef()) { have_dense_links = true; dense_links = heap; }
                    break;
                }
                case 0x08: o.layout = parse_layout(d); o.has_layout = true; break;
                case 0x0B: o.filters = parse_filters(d); break;
                case 0x0C: o.attrs.push_back(parse_attribute(d)); break;
                case 0x11: have_stab = true; stab[0] = off(d); stab[1] = off(d + (size_t)O_); break;
                case 0x15: {
                    const int fl = b_[d + 1];
                    const size_t p = d + 2 + ((fl & 1) ? 2 : 0);
                    const uint64_t heap = off(p);
                    if (heap != undef()) { have_dense_attrs = true; dense_attrs = heap; }
                    break;
                }
                default: break;
            }
        }
        if (have_stab) { o.is_group = true; o.links.clear(); walk_group_btree(stab[0], stab[1], o.links); }
        if (have_dense_links)
            for (size_t start : heap_objects(dense_links, true)) { auto l = parse_link(start); if (l.second.first) o.links.emplace_back(l.first, l.second.second); }
        if (have_dense_attrs)
            for (size_t start : heap_objects(dense_attrs, false)) {
                auto kv = parse_attribute(start);
                bool rep = false;
                for (auto &old : o.attrs) if (old.first == kv.first) { old.second = kv.second; rep = true; }
                if (!rep) o.attrs.push_back(kv);
            }
    }

public:
    int checked_lookup3 = 0, checked_fletcher32 = 0;
    explicit File(std::string bytes) : buf_(std::move(bytes)), b_(reinterpret_cast<const uint8_t *>(buf_.data())), n_(buf_.size()) { superblock(); }
    File(File const &) = delete;
    File &operator=(File const &) = delete;
    Obj &obj(uint64_t addr) {
        auto it = objs_.find(addr);
        if (it != objs_.end()) return *it->second;
        if (objs_.size() > 1000000) throw Error("too many objects");
        auto o = std::make_shared<Obj>();
        o->addr = addr;
        objs_[addr] = o;
        parse_header(*o);
        return *o;
    }
    Obj &root() { return obj(root_addr_); }
    /** the dataset's elements as one contiguous buffer in file byte order, then decoded */
    Value read(Obj const &o) {
        if (!o.is_dataset()) throw Error("object is not a dataset");
        DType const &dt = o.dtype;
        std::vector<uint64_t> shape = o.null_space ? std::vector<uint64_t>{0} : o.shape;
        const uint64_t count = checked_count(shape);
        const size_t es = dt.size;
        // deflate expands at most ~1032 : 1, so a dataset larger than that multiple of the file is a corrupt dataspace, not data
        if (es == 0 || count > ((uint64_t)n_ * 1100 + (1u << 20)) / es) throw Error("dataset larger than the file can hold (corrupt dataspace?)");
        std::string raw;
        if (o.layout.kind == 0) { need((size_t)o.layout.a, (size_t)o.layout.b); raw.assign(reinterpret_cast<const char *>(b_ + o.layout.a), (size_t)o.layout.b); }
        else if (o.layout.kind == 1) {
            if (o.layout.a == undef()) {
                raw.assign((size_t)count * es, '\0');
                if (o.has_fill && o.fill.size() == es) for (uint64_t k = 0; k < count; ++k) std::memcpy(&raw[k * es], o.fill.data(), es);
            } else { need((size_t)(o.layout.a + base_), (size_t)count * es); raw.assign(reinterpret_cast<const char *>(b_ + o.layout.a + base_), (size_t)count * es); }
        } else {
            std::vector<uint64_t> cdims;
            std::vector<Chunk> cl;
            const int rank = (int)shape.size();
            if (o.layout.kind == 3) {
                cdims = o.layout.dims;
                if (o.layout.a != undef()) {
                    const uint64_t n = checked_count(cdims, es);
                    const uint64_t csz = o.layout.has_fsize ? o.layout.fsize : n;
                    if (csz > n_) throw Error("single chunk larger than the file");
                    Chunk c; c.offs.assign((size_t)rank, 0); c.addr = o.layout.a; c.size = (decltype(c.size))csz; c.mask = o.layout.fmask;
                    if ((uint64_t)c.size != csz) throw Error("single chunk too large for this reader");
                    cl.push_back(c);
                }
            } else {
                cdims.assign(o.layout.dims.begin(), o.layout.dims.end() - (o.layout.dims.empty() ? 0 : 1));
                chunks(o.layout.a, rank, cl, 0);
            }
            if ((int)cdims.size() != rank) throw Error("chunk rank does not match the dataspace");
            for (auto c : cdims) if (c == 0) throw Error("chunk dimension of zero");
            raw.assign((size_t)count * es, '\0');
            if (o.has_fill && o.fill.size() == es) for (uint64_t k = 0; k < count; ++k) std::memcpy(&raw[k * es], o.fill.data(), es);
            const uint64_t celems = checked_count(cdims);
            if (celems > UINT64_MAX / (es ? es : 1)) throw Error("chunk too large (byte size overflows)");
            std::vector<uint64_t> stride((size_t)rank, 1), cstride((size_t)rank, 1);
            for (int i = rank - 2; i >= 0; --i) { stride[(size_t)i] = stride[(size_t)i + 1] * shape[(size_t)i + 1]; cstride[(size_t)i] = cstride[(size_t)i + 1] * cdims[(size_t)i + 1]; }
            for (Chunk const &c : cl) {
                need((size_t)(c.addr + base_), c.size);
                const std::string ch = unfilter(std::string(reinterpret_cast<const char *>(b_ + c.addr + base_), c.size), o, c.mask);
                if (ch.size() < celems * es) throw Error("chunk shorter than its dimensions");
                // rows of the innermost dimension, clipped to the dataset
                if (rank == 0) { std::memcpy(&raw[0], ch.data(), es); continue; }
                const uint64_t inner = std::min<uint64_t>(cdims[(size_t)rank - 1], shape[(size_t)rank - 1] > c.offs[(size_t)rank - 1] ? shape[(size_t)rank - 1] - c.offs[(size_t)rank - 1] : 0);
                if (!inner) continue;
                const uint64_t nrows = celems / cdims[(size_t)rank - 1];
                for (uint64_t rrow = 0; rrow < nrows; ++rrow) {
                    uint64_t rem = rrow, dst = 0;
                    bool inside = true;
                    for (int i = 0; i < rank - 1; ++i) {
                        const uint64_t ci = rem / (cstride[(size_t)i] / cdims[(size_t)rank - 1]);
                        rem %= cstride[(size_t)i] / cdims[(size_t)rank - 1];
                        const uint64_t gi = c.offs[(size_t)i] + ci;
                        if (gi >= shape[(size_t)i]) { inside = false; break; }
                        dst += gi * stride[(size_t)i];
                    }
                    if (!inside) continue;
                    dst += c.offs[(size_t)rank - 1];
                    std::memcpy(&raw[dst * es], &ch[rrow * cdims[(size_t)rank - 1] * es], (size_t)inner * es);
                }
            }
        }
        return decode(reinterpret_cast<const uint8_t *>(raw.data()), raw.size(), dt, shape);
    }
};

inline bool is_hdf5(std::string const &buf) {
    static const char SIG[8] = {'\x89', 'H', 'D', 'F', '\r', '\n', '\x1a', '\n'};
    for (size_t pos = 0; pos + 8 <= buf.size(); pos = pos == 0 ? 512 : pos * 2)
        if (std::memcmp(buf.data() + pos, SIG, 8) == 0) return true;
    return false;
}

/** root group of a NetCDF-4 file -> the container model of ncio.hpp */
inline nc::File read_netcdf4(std::string bytes) {
    File f(std::move(bytes));
    nc::File out;
    static const char *HIDDEN[] = {"CLASS", "NAME", "DIMENSION_LIST", "REFERENCE_LIST", "_Netcdf4Dimid", "_Netcdf4Coordinates", "_nc3_strict",
                                   "_NCProperties", "DIMENSION_LABELS", "_Netcdf4BeginId"};
    auto hidden = [&](std::string const &k) { for (auto h : HIDDEN) if (k == h) return true; return false; };
    auto to_attr = [&](Value const &v, nc::Array &a) -> bool {
        if (v.null || v.is_refs) return false;
        if (v.is_str) { a = nc::Array::str(nc::join_names(v.strs)); return true; }
        a = v.arr;
        return true;
    };
    Obj &root = f.root();
    for (auto const &kv : root.attrs) { nc::Array a; if (!hidden(kv.first) && to_attr(kv.second, a)) out.attrs.emplace_back(kv.first, a); }
    struct Member { std::string name; Obj *o; };
    std::vector<Member> ds;
    for (auto const &l : root.links) { Obj &o = f.obj(l.second); if (o.is_dataset()) ds.push_back(Member{l.first, &o}); }
    // dimensions: the dimension scales, in _Netcdf4Dimid order (creation order otherwise)
    struct Scale { long id; std::string name; Obj *o; };
    std::vector<Scale> scales;
    for (auto const &m : ds) {
        const Value *cls = m.o->attr("CLASS");
        if (cls && cls->is_str && !cls->strs.empty() && cls->strs[0] == "DIMENSION_SCALE") {
            const Value *id = m.o->attr("_Netcdf4Dimid");
            scales.push_back(Scale{id && !id->is_str && !id->is_refs && id->arr.size() ? (long)id->arr.at<int64_t>(0) : (long)scales.size(), m.name, m.o});
        }
    }
    std::stable_sort(scales.begin(), scales.end(), [](Scale const &a, Scale const &b) { return a.id < b.id; });
    std::map<uint64_t, std::string> by_addr;
    for (auto const &s : scales) { out.dims.emplace_back(s.name, s.o->shape.empty() ? 1 : (int64_t)s.o->shape[0]); by_addr[s.o->addr] = s.name; }
    for (auto const &m : ds) {
        const Value *nm = m.o->attr("NAME");
        if (nm && nm->is_str && !nm->strs.empty() && nm->strs[0].compare(0, 53, "This is a netCDF dimension but not a netCDF variable.") == 0) continue;
        std::vector<uint64_t> shape = m.o->null_space ? std::vector<uint64_t>() : m.o->shape;
        std::vector<std::string> dims;
        bool have = false;
        const Value *dl = m.o->attr("DIMENSION_LIST");
        if (dl && dl->is_refs && !shape.empty() && dl->refs.size() == shape.size()) {
            have = true;
            for (auto const &r : dl->refs) { if (r.empty() || !by_addr.count(r[0])) { have = false; break; } dims.push_back(by_addr[r[0]]); }
            if (!have) dims.clear();
        }
        if (!have && by_addr.count(m.o->addr) && shape.size() == 1) { dims = {by_addr[m.o->addr]}; have = true; }
        if (!have) {
            for (size_t ax = 0; ax < shape.size(); ++ax) {          // anonymous dimensions (plain HDF5 datasets)
                const std::string dn = "phony_dim_" + std::to_string(ax) + "_" + std::to_string(shape[ax]);
                bool known = false;
                for (auto const &d : out.dims) known = known || d.first == dn;
                if (!known) out.dims.emplace_back(dn, (int64_t)shape[ax]);
                dims.push_back(dn);
            }
        }
        nc::Var v;
        v.dims = dims;
        Value val = f.read(*m.o);
        if (val.is_str) {                                        // a string variable: one char row per element
            size_t w = 1;
            for (auto const &s : val.strs) w = std::max(w, s.size());
            const std::string dn = "string_len_" + std::to_string(w);
            bool known = false;
            for (auto const &d : out.dims) known = known || d.first == dn;
            if (!known) out.dims.emplace_back(dn, (int64_t)w);
            v.dims.push_back(dn);
            std::string flat;
            for (auto const &s : val.strs) { flat += s; flat.append(w - s.size(), '\0'); }
            v.data = nc::Array::str(flat);
        } else if (val.is_refs) continue;
        else v.data = val.arr;
        for (auto const &kv : m.o->attrs) { nc::Array a; if (!hidden(kv.first) && to_attr(kv.second, a)) v.attrs.emplace_back(kv.first, a); }
        out.vars.emplace_back(m.name, std::move(v));
    }
    return out;
}

}   // namespace h5
}   // namespace icebin
