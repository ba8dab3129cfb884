// prims.h -- device-wide building blocks of the assembly path (K2): exclusive scan and a
// stable LSD radix sort of (u64 key, u32 payload) pairs, hand-written for wave64.
#pragma once
#include "common.h"

namespace ibh {

// Workspace arena: bump allocator over a few big hipMalloc blocks, reused across matrix builds
// (the coupler rebuilds four matrices per step: no hipMalloc/hipFree on that path).  One per host
// thread; everything that uses it runs on one stream, so reuse after reset() is ordered.
struct Arena {
    struct Block { char *p; size_t cap, used; };
    std::vector<Block> blocks;
    ~Arena();
    void reset() { for (auto &b : blocks) b.used = 0; }
    void *get_bytes(size_t bytes);
    template <class T> T *get(size_t n) { return static_cast<T *>(get_bytes((n ? n : 1) * sizeof(T))); }
};
Arena &arena();
void release_workspace();

// Small device -> host read-back + stream synchronisation through a per-thread PINNED staging buffer
// (a pageable destination makes the runtime stage the copy itself, measurably slower for the two
// synchronisation points of every matrix build).  bytes <= 256.
void readback_sync(void *dst, const void *dsrc, size_t bytes, hipStream_t stream);


// out[i] = sum_{j<i} in[j] (u32, wraps at 2^32); in == out allowed.  If total != nullptr the
// grand total is written there (device pointer).  All work is enqueued on `stream`.
void exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total, hipStream_t stream);
// the same over byte-sized counts (flags: a quarter of the read traffic)
void exclusive_scan_u8(const uint8_t *in, uint32_t *out, size_t n, uint32_t *total, hipStream_t stream);

// Three exclusive scans in one pass over a packed stream: pk[i] bits [0,2) and [2,4) are flag pairs
// (channel value = number of set flags), bits [4,6) a count 0..3.  totals3: device uint32[3].
void exclusive_scan3(const uint32_t *pk, size_t n, uint32_t *o0, uint32_t *o1, uint32_t *o2, uint32_t *totals3,
                     hipStream_t stream);

// Stable sort of n pairs by the key bits listed in `fields` (least significant field first);
// each field is (shift, nbits) with nbits <= 32 and is split into digits of <= 8 bits.
// Ping-pongs between (keys, vals) and (keys_alt, vals_alt); returns true when the sorted
// data ended up in the *_alt buffers.
struct KeyField { int shift, nbits; };
bool radix_sort_pairs(uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t n,
                      const KeyField *fields, int nfields, hipStream_t stream);

// Same result as radix_sort_pairs on key = (hi field at bit 32, lo field at bit 0), but first tries
// to finish in one analysis pass + LDS sorts of the independent pieces the sequence splits into
// (see prims.hip "adaptive ordering").  Synchronises the stream once.  d_info: device scratch.
enum { ORD_FULL_DEC = 1, ORD_HI_DEC = 2, ORD_LO_DEC = 4 };
constexpr int CS_SMALL = 2048, CS_MID = 4096, CS_BIG = 8192;   // size classes of the pieces sorted in LDS; longer pieces -> radix sort
struct OrderInfo { uint32_t flags, nchunks, maxlen, nsmall, nmid, nbig; };
void order_and_chunk_sort(uint64_t *keys, uint32_t *idx, size_t n, OrderInfo *d_info, hipStream_t stream,
                          bool try_pieces = true);
// Pieces of the synchronous form, for callers that fold the read-back of *d_info into a host
// synchronisation they need anyway: order_and_chunk_sort() [async]; read *d_info; if !order_is_final()
// the data is untouched and radix_after_analysis() sorts it (returns true when the result is in *_alt).
bool order_is_final(const OrderInfo &h);
bool radix_after_analysis(const OrderInfo &h, uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt,
                          size_t n, int lo_bits, int hi_bits, hipStream_t stream);
void order_debug(const OrderInfo &h, size_t n, int lo_bits, int hi_bits);
bool adaptive_sort_pairs(uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t n,
                         int lo_bits, int hi_bits, OrderInfo *d_info, hipStream_t stream);

}  // namespace ibh
