// prims.h -- device-wide building blocks of the assembly path (K2): exclusive scan and a
// stable LSD radix sort of (u64 key, u32 payload) pairs, hand-written for wave64.
#pragma once
#include "common.h"

namespace ibh {

// out[i] = sum_{j<i} in[j] (u32, wraps at 2^32); in == out allowed.  If total != nullptr the
// grand total is written there (device pointer).  All work is enqueued on `stream`.
void exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total, hipStream_t stream);

// Stable sort of n pairs by the key bits listed in `fields` (least significant field first);
// each field is (shift, nbits) with nbits <= 32 and is split into digits of <= 8 bits.
// Ping-pongs between (keys, vals) and (keys_alt, vals_alt); returns true when the sorted
// data ended up in the *_alt buffers.
struct KeyField { int shift, nbits; };
bool radix_sort_pairs(uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t n,
                      const KeyField *fields, int nfields, hipStream_t stream);

}  // namespace ibh
