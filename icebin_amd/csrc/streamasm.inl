// streamasm.inl -- the STREAMED build for sorted exchange grids (textually included after fastasm.inl, inside namespace ibh).
//
// fastasm.inl gives every range (the exchange cells of one atmosphere cell) a workgroup that walks it pass by pass: dependent
// load rounds (cell -> mask -> dense id), ranking barriers and the sequential sums all sit in one latency chain per range, and
// both big kernels read every exchange cell in full (measured on the Antarctic sheet: 2.8-3.5 x the algorithmic traffic at
// 0.15 / 0.075 of the HBM peak).  Here the heavy reads are STREAMS -- a thread owns a cell, nothing in a thread's load chain
// depends on another cell -- and everything that is sequential by nature runs on compact per-cell BYTES:
//
//   S1  k_sa_flags   stream: per cell one CODE byte {P key first seen here, head of a group with entries, straddler, unmasked}
//                    and (elevation classes) one CLASS byte {first class, number of classes}; per 64 cells the number of new
//                    P keys (and, I/X-row matrices, of row entries).  Reads 4 + 1 bytes per cell and the mask gather.
//   S2  k_sa_ranges  one wave per range, over the bytes only: first-seen order of the classes, entries per class (new /
//                    straddling), and for every entry its POSITION inside the range's block of entries (16 bits when the
//                    longest range allows).  What k_fa_count's workgroups computed from full cells.
//       scans        new P keys per 64 cells -> dense ids (id = prefix of the wave + popcount of the code bits before: no
//                    per-cell table, an id at any cell f is two small reads); classes / entries per range -> row ids, offsets
//   S3  k_sa_emit    stream: the cells with entries are evaluated ONCE (area, mask value) and every entry goes straight to
//                    ebase[range] + position; the dims table, Mw of the one-cell ice cells (A/E-row matrices) or the whole
//                    row with its weight and scaling (I/X-row matrices: rows come out in x order) from the same pass
//   S4  k_sa_pairs   the ice cells across one GCM-cell edge (two exchange cells: a few %): their Mw / their rows, one thread each
//       k_fa_pelem   the rest of the ice cells with several exchange cells (corners, duplicates), from the static lists
//   S5  k_sa_rows    a 16-lane row of a wave per (range, class): straddlers sorted into place, the sequential sums (spsparse sum()
//                    order) of four segments in lock step through DPP row broadcasts, weights, scaling
// Same sums in the same order as fastasm.inl, the general pipeline and the oracle: bit-identical (tests force either path).
// Not served here (fastasm.inl's kernels take them): plans with underflowing areas (the elevation-class builds count by the sign
// of the area), EvA / AvE (fast_build_eva).

constexpr int SA_T = 256, SA_CPT = 4, SA_TILE = SA_T * SA_CPT;
// code byte: a P key is first seen here | this cell heads a group with entries | those entries straddle in from an earlier range |
// unmasked | static plan bits the emit pass needs (so that it does not read the plan's byte again): the only exchange cell of its
// ice cell, the first-seen cell of its ice cell, a duplicate follows, the first cell of its range
enum { SA_P = 1, SA_ENT = 2, SA_OLD = 4, SA_UNM = 8, SA_ONE = 16, SA_FIRST = 32, SA_DUPNEXT = 64, SA_RHEAD = 128 };
// straddling entries of one (range, class) segment sorted in LDS by k_sa_rows: tables of 128 (many short ranges: four waves of a
// workgroup hold 12 KB, the CU stays full) or 512 (long ranges: few segments, occupancy does not matter); more -> fastasm.inl
constexpr int SA_OLDSEG_S = 128, SA_OLDSEG_L = 512;

struct SaBuf {
    uint8_t *code;          // [nW * 64]
    uint8_t *cls;           // [nW * 64]   elevation classes: first class | number of classes << 6 (heads of groups only)
    uint8_t *rl;            // [nW * 64]   I/X-row matrices: entries of the row owned by this cell
    uint8_t *rli;           // [nI]        I-row matrix on an IDENTITY ice set: the same by ice cell (its rows lie in ice-cell order)
    void *rel;              // [nX * S]    position of every entry inside its range's block (uint16_t / uint32_t)
    uint8_t *cntP;          // [nW]        new P keys per 64 cells
    uint32_t *cntL;         // [nW]        row entries per 64 cells (I/X-row matrices)
    uint32_t *Pw, *Lw;      // their exclusive scans
    unsigned long long *pbits;   // [nW]   the cells of a wave where a P key is first seen (bit = lane): dense id = Pw + popcount below
    // one-class matrices (A rows / A columns): entries first seen in their range / straddling, per 64 cells, their scans, and the
    // scans' values at the start of every range -- the position of an entry inside its range's block follows from these (the
    // elevation-class matrices take it from `rel`, written by k_sa_ranges)
    uint8_t *cntE, *cntO;
    uint32_t *Ew, *Eow, *ewbase, *eobase;
    // the slice of the exchange grid this build works on (a whole sheet, or one rank's share of it -- ibh_regrid_matrices_matrix_d_sharded):
    // ranges [sr0, sr1) = cells [sx0, sx1); the per-cell passes cover whole waves, [gx0, gx1) = the slice rounded out to 64 cells
    long sx0, sx1, gx0, gx1;
    int sr0, sr1;
    int end_row;            // rowptr[end_row] = end_nnz (the closing entry of the row pointer), written by the emit pass
    uint32_t end_nnz;
};

__device__ __forceinline__ uint32_t sa_wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t sa_wave_excl_u32(uint32_t v, int lane) {
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    return inc - v;
}
// OR over the wave with DPP (VALU speed; lanes a step does not write contribute 0, the identity): quad swaps, half-row and row
// mirrors, then the row broadcasts -- lane 63 ends up with the whole wave's value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned sa_dpp_or(unsigned v) {
    return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned sa_wave_or32(unsigned v) {
    v = sa_dpp_or<0xB1, 0xf>(v); v = sa_dpp_or<0x4E, 0xf>(v); v = sa_dpp_or<0x141, 0xf>(v); v = sa_dpp_or<0x140, 0xf>(v);
    v = sa_dpp_or<0x142, 0xa>(v); v = sa_dpp_or<0x143, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long sa_wave_or64(unsigned long long v) {
    const unsigned lo = sa_wave_or32((unsigned)v), hi = sa_wave_or32((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// ---- S1: one byte (two) per cell -------------------------------------------------------------------------------------------------
// The class pattern of an elevation: the classes it lies between and which of the two weights are non-zero
// (IceRegridder_L0.cpp:127-150).  With no area below 1e-290 in the plan, area * weight != 0 <=> area != 0 and weight != 0.
struct SaCls { int c0, ncls; bool err; };
__device__ __forceinline__ SaCls sa_classes(const RgView &rg, double e) {
    SaCls k{0, 0, false};
    const double elevation = e < 0.0 ? 0.0 : e;
    if (rg.interp == 0) {
        int i1 = dev_lower_bound(rg.hc, rg.nhc, elevation);
        if (i1 <= 0) i1 = 1;
        if (i1 >= rg.nhc) { k.err = true; return k; }
        const int i0 = i1 - 1;
        const double ratio = (elevation - rg.hc[i0]) / (rg.hc[i1] - rg.hc[i0]);
        const double w0 = 1.0 - ratio, w1 = ratio;
        const bool h0 = w0 != 0, h1 = w1 != 0;
        k.ncls = (h0 ? 1 : 0) + (h1 ? 1 : 0);
        k.c0 = h0 ? i0 : i1;
    } else {
        const int n = rg.nhc;
        const int i1 = dev_lower_bound(rg.hc, n, elevation);
        int ih;
        if (i1 <= 0) ih = 0;
        else if (i1 >= n) ih = n - 1;
        else {
            const int i0 = i1 - 1;
            const double d0 = fabs(elevation - rg.hc[i0]), d1 = fabs(rg.hc[i1] - elevation);
            ih = d0 <= d1 ? i0 : i1;
        }
        k.ncls = 1; k.c0 = ih;
    }
    return k;
}

// The same per ICE cell, once per elevmask (ibh_regrid_matrices::em_cls): what k_sa_flags needs of an elevation is its mask bit and
// its class pattern, so the per-matrix pass gathers one byte per exchange cell instead of eight (the Antarctic AvI: 558 -> ~230 MB
// fetched by that pass) and does no class search.
__global__ __launch_bounds__(256) void k_em_classes(RgView rg, long nI, uint8_t *__restrict__ out, double *__restrict__ copy_to) {
    __shared__ double s_hc[HC_LDS];
    stage_hc<true>(rg, s_hc);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nI) return;
    const double e = rg.em[i];
    if (copy_to) copy_to[i] = e;                                 // (creation from a device array: the object's own copy in the same pass)
    uint8_t b;
    if (e != e) b = 0xFF;
    else if (rg.nhc < 1) b = 0;
    else {
        const SaCls k = sa_classes(rg, e);
        b = k.err ? (uint8_t)0xFE : (uint8_t)(k.c0 | (k.ncls << 6));
    }
    out[i] = b;
}
// src: the caller's device array when the object's own copy is still to be made (then it is made here)
static const uint8_t *elevmask_classes_impl(const ibh_regrid_matrices *rm, hipStream_t st, const double *src = nullptr) {
    const ibh_regridder *gr = rm->rg;
    std::lock_guard<std::mutex> lk(rm->em_cls_mu);
    if (!src && rm->em_cls.p && rm->em_cls_nhc == gr->nhc && rm->em_cls_interp == gr->interp_style) return rm->em_cls.p;
    IBH_CHECK(gr->nhc <= 64, "more than 64 elevation classes");
    if (!rm->em_cls.p) rm->em_cls.alloc((size_t)std::max<int64_t>(gr->nI, 1));
    RgView rg{};
    rg.em = src ? src : rm->elevmaskI.p; rg.hc = gr->hcdefs.p; rg.nhc = gr->nhc; rg.interp = gr->interp_style;
    if (gr->nI) hipLaunchKernelGGL(k_em_classes, dim3((unsigned)ceil_div(gr->nI, 256l)), dim3(256), 0, st, rg, (long)gr->nI, rm->em_cls.p,
                                   src ? rm->elevmaskI.p : (double *)nullptr);
    IBH_HIP(hipGetLastError());
    IBH_HIP(hipStreamSynchronize(st));                           // (builds on other streams read it next)
    rm->em_cls_nhc = gr->nhc; rm->em_cls_interp = gr->interp_style;
    return rm->em_cls.p;
}

// pkey_x: the P side is keyed by exchange cells (nothing merges); p_list: the list that makes an exchange cell a member of an X
// set (LIST_AP / LIST_EP); prows: I/X-row matrix (row lengths wanted)
template <bool WITH_EP>
__global__ __launch_bounds__(SA_T) void k_sa_flags(RgView rg, PlanView pl, SaBuf sb, int pkey_x, int p_fresh, int p_list, int prows,
                                                   uint32_t *__restrict__ err_x, const uint8_t *__restrict__ emc) {
    const int tid = threadIdx.x, lane = tid & 63;
    const long cb = sb.gx0 + (long)blockIdx.x * SA_TILE;
    int iIv[SA_CPT];
    unsigned fl[SA_CPT];
    unsigned ev[SA_CPT];                                         // the ice cell's byte: mask and class pattern (k_em_classes)
#pragma unroll
    for (int u = 0; u < SA_CPT; ++u) {
        const long x = cb + (long)u * SA_T + tid;
        const long xx = x < sb.gx1 ? x : sb.gx1 - 1;
        iIv[u] = pl.exI[xx]; fl[u] = pl.isdup[xx];
    }
#pragma unroll
    for (int u = 0; u < SA_CPT; ++u) ev[u] = emc[iIv[u]];
#pragma unroll
    for (int u = 0; u < SA_CPT; ++u) {
        const long x = cb + (long)u * SA_T + tid;
        const bool in = x < sb.gx1;
        const bool unm = in && ev[u] != 0xFFu;
        const bool head = pkey_x || !(fl[u] & 1);
        unsigned ga = fl[u] & 24;                                  // sign classes of the areas of the group this cell heads
        if (!pkey_x && head && (fl[u] & 128))
            for (long k = x + 1; k < rg.nX && (pl.isdup[k] & 1); ++k) ga |= pl.isdup[k] & 24;
        const bool gpos = (ga & 8) != 0, gnz = ga != 0;
        int n = 0, c0 = 0, ncls_e = 0;
        if (WITH_EP && unm) {
            if (ev[u] == 0xFEu) atomicMin(err_x, (uint32_t)x);
            else { ncls_e = (int)(ev[u] >> 6); c0 = (int)(ev[u] & 63u); n = (head && gnz) ? ncls_e : 0; }
        }
        const bool ent = head && unm && (WITH_EP ? n > 0 : gpos);
        bool P;
        if (!pkey_x) P = unm && (fl[u] & 4);
        else if (p_list == LIST_AP) P = unm && (fl[u] & 8);
        else P = n > 0;
        const bool old = ent && !pkey_x && p_fresh && (fl[u] & 64);
        if (in) {
            sb.code[x] = (uint8_t)((P ? SA_P : 0) | (ent ? SA_ENT : 0) | (old ? SA_OLD : 0) | (unm ? SA_UNM : 0) | ((fl[u] & 2) ? SA_ONE : 0) |
                                   ((fl[u] & 4) ? SA_FIRST : 0) | ((fl[u] & 128) ? SA_DUPNEXT : 0) | ((fl[u] & 32) ? SA_RHEAD : 0));
            if (WITH_EP) sb.cls[x] = (uint8_t)(c0 | (n << 6));
        }
        const unsigned long long bp = __ballot(P);
        const long xw = cb + (long)u * SA_T + (tid & ~63);
        if (lane == 0 && xw < sb.gx1) { sb.cntP[xw >> 6] = (uint8_t)__popcll(bp); sb.pbits[xw >> 6] = bp; }
        if (!WITH_EP) {
            const unsigned long long bn = __ballot(ent && !old), bo = __ballot(old);
            if (lane == 0 && xw < sb.gx1) { sb.cntE[xw >> 6] = (uint8_t)__popcll(bn); sb.cntO[xw >> 6] = (uint8_t)__popcll(bo); }
        }
        if (prows) {
            // entries of the row this cell owns: a one-cell ice cell / an exchange cell -> its own entries; an ice cell with
            // several exchange cells -> (groups of duplicates with a contributing member: static, plan) x (classes)
            uint32_t len = 0;
            if (P) {
                if (pkey_x || (fl[u] & 2)) len = ent ? (WITH_EP ? (uint32_t)n : 1u) : 0u;
                else len = WITH_EP ? (uint32_t)ncls_e * pl.icnt_nz[iIv[u]] : (uint32_t)pl.icnt_pos[iIv[u]];
            }
            if (in) sb.rl[x] = (uint8_t)len;
            if (sb.rli) {                                        // identity ice set: rows by ice cell (zeroed beforehand: masked cells have empty rows)
                if (P && len) sb.rli[iIv[u]] = (uint8_t)len;
            }
            if (sb.cntL) {                                       // (rows in first-seen order; a SHARED build on an identity ice set: the order its rows travel in)
                const uint32_t tot = sa_wave_sum_u32(len);
                if (lane == 0 && xw < sb.gx1) sb.cntL[xw >> 6] = tot;
            }
        }
    }
}

// entries before cell x: `W` = the scan over whole waves of 64 cells, the rest from the code bytes of x's wave
// (NEW: heads with entries first seen in their range -- ENT set, OLD clear; otherwise the straddlers -- OLD set)
template <bool NEW>
__device__ __forceinline__ uint32_t sa_erank_at(const uint32_t *__restrict__ W, const uint8_t *__restrict__ code, long x) {
    const long w = x >> 6;
    const int k = (int)(x & 63);
    uint32_t rnk = W[w];
    const unsigned long long *__restrict__ q = reinterpret_cast<const unsigned long long *>(code + (w << 6));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int nb = k - 8 * j;
        if (nb <= 0) break;
        const unsigned long long t = q[j];
        const unsigned long long ent = (t >> 1) & 0x0101010101010101ull, old = (t >> 2) & 0x0101010101010101ull;
        const unsigned long long word = NEW ? (ent & ~old) : old;
        rnk += (uint32_t)__popcll(nb >= 8 ? word : (word & ((1ull << (8 * nb)) - 1)));
    }
    return rnk;
}
// ---- S2 of the one-class matrices: a thread per range -- its entries are differences of the scans at its two ends --------------
__global__ void k_sa_rangecounts(PlanView pl, SaBuf sb, FaG g, uint32_t *__restrict__ flags, int oldseg) {
    const int r = sb.sr0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= sb.sr1) return;
    const long x0 = pl.arng[r], x1 = pl.arng[r + 1];
    const uint32_t n0 = sa_erank_at<true>(sb.Ew, sb.code, x0), n1 = sa_erank_at<true>(sb.Ew, sb.code, x1);
    const uint32_t o0 = sa_erank_at<false>(sb.Eow, sb.code, x0), o1 = sa_erank_at<false>(sb.Eow, sb.code, x1);
    const uint32_t cn = n1 - n0, co = o1 - o0;
    const bool member = cn + co > 0;
    g.erank[r] = member ? 0 : -1;
    g.ecntn[r] = cn; g.ecnto[r] = co;
    g.r_ncls[r] = member ? 1u : 0u; g.r_nent[r] = cn + co;
    sb.ewbase[r] = n0; sb.eobase[r] = o0;
    if (co > (uint32_t)oldseg) atomicOr(flags, (uint32_t)FA_ERR_OLDOVER);
}

// ---- S2: over the bytes only; WPR waves per range (1: four ranges per workgroup; 4 / 16: one range per workgroup, every wave
// takes a slice of it in x order, the per-class tables are combined through LDS) -------------------------------------------------
template <bool WITH_EP, typename REL, int WPR>
__global__ __launch_bounds__(WPR == 16 ? 1024 : 256) void k_sa_ranges(RgView rg, PlanView pl, SaBuf sb, FaG g, uint32_t *__restrict__ flags, int oldseg) {
    __shared__ uint32_t s_first[WPR > 1 ? WPR : 1][64], s_cn[WPR > 1 ? WPR : 1][64], s_co[WPR > 1 ? WPR : 1][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = sb.sr0 + (WPR == 1 ? (int)blockIdx.x * 4 + wv : (int)blockIdx.x);
    const int ws = WPR == 1 ? 0 : wv;                              // this wave's slice of the range
    if (r >= sb.sr1) return;                                     // (WPR > 1: the whole workgroup)
    const long r0 = pl.arng[r], r1 = pl.arng[r + 1];
    const long slice = WPR == 1 ? (r1 - r0) : ((((r1 - r0) + WPR - 1) / WPR + 63) & ~63l);
    const long x0 = r0 + ws * slice, x1 = (x0 + slice < r1) ? x0 + slice : r1;
    REL *__restrict__ rel = static_cast<REL *>(sb.rel);
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // pass 1 -- lane c holds the table entry of class c (one class: lane 0): first-seen position, new / straddling entries
    uint32_t first = 0xffffffffu, cn = 0, co = 0;
    for (long cb4 = x0; cb4 < x1; cb4 += 256) {                   // (four chunks of 64 bytes in flight per step)
        unsigned cdv[4], clv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long x = cb4 + 64 * q + lane;
            cdv[q] = x < x1 ? sb.code[x] : 0u;
            clv[q] = (WITH_EP && x < x1) ? sb.cls[x] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long cb = cb4 + 64 * q;
            if (cb >= x1) break;
            const unsigned cd = cdv[q], cb8 = clv[q];
            const unsigned long long men = __ballot((cd & SA_ENT) && !(cd & SA_OLD)), meo = __ballot((cd & SA_OLD) != 0);
            if (!WITH_EP) {
                if (lane == 0) {
                    if (first == 0xffffffffu && (men | meo)) first = 0;
                    cn += (uint32_t)__popcll(men); co += (uint32_t)__popcll(meo);
                }
                continue;
            }
            const int n = (int)(cb8 >> 6), c0 = (int)(cb8 & 63);
            unsigned long long bits = sa_wave_or64(n > 0 ? ((1ull << c0) | (n == 2 ? (2ull << c0) : 0ull)) : 0ull);
            while (bits) {
                const int c = __builtin_ctzll(bits);
                bits &= bits - 1;
                const unsigned long long m0 = __ballot(n > 0 && c0 == c), m1 = __ballot(n == 2 && c0 + 1 == c), mem = m0 | m1;
                const int l = __builtin_ctzll(mem);
                const uint32_t pos = 2u * (uint32_t)(cb + l) + (((m0 >> l) & 1) ? 0u : 1u);
                if (lane == c) {
                    if (first == 0xffffffffu) first = pos;
                    cn += (uint32_t)__popcll(men & mem); co += (uint32_t)__popcll(meo & mem);
                }
            }
        }
    }
    uint32_t runn = 0, runo = 0;                                   // entries of this lane's class in the slices before this wave's
    if (WPR > 1) {
        s_first[ws][lane] = first; s_cn[ws][lane] = cn; s_co[ws][lane] = co;
        __syncthreads();
        first = 0xffffffffu; cn = 0; co = 0;
#pragma unroll
        for (int w = 0; w < WPR; ++w) {
            const uint32_t fw = s_first[w][lane], nw = s_cn[w][lane], ow = s_co[w][lane];
            if (first == 0xffffffffu) first = fw;                 // (slices lie in x order: the first slice that lists the class)
            if (w < ws) { runn += nw; runo += ow; }
            cn += nw; co += ow;
        }
    }
    // first-seen rank of every class of the range, start of its segment inside the range's block of entries
    const unsigned long long present = __ballot(first != 0xffffffffu);
    int rank = 0;
    for (unsigned long long m = present; m; m &= m - 1) {
        const int k = __builtin_ctzll(m);
        const uint32_t fk = (uint32_t)__builtin_amdgcn_readlane((int)first, k);
        rank += fk < first ? 1 : 0;
    }
    const uint32_t seg = cn + co;
    uint32_t start = 0, nent = 0;
    for (unsigned long long m = present; m; m &= m - 1) {
        const int k = __builtin_ctzll(m);
        const int rk = __builtin_amdgcn_readlane(rank, k);
        const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)seg, k);
        start += rk < rank ? sk : 0u;
        nent += sk;
    }
    const bool mine = first != 0xffffffffu;
    if (ws == 0) {
        if (lane < g.NC) {
            g.erank[(size_t)r * g.NC + lane] = (int8_t)(mine ? rank : -1);
            g.ecntn[(size_t)r * g.NC + lane] = cn;
            g.ecnto[(size_t)r * g.NC + lane] = co;
            if (mine && g.tab) {                                   // pre-populated set: every key must already be there
                const long key = (long)pl.riA[r] * rg.sA + (long)lane * rg.sHC;
                if (g.tab[key] < 0) atomicOr(flags, (uint32_t)FA_ERR_MISSING);
            }
            if (!mine && seg) atomicOr(flags, (uint32_t)FA_ERR_MISSING);       // entries of a class that was never listed: cannot happen
            if (co > (uint32_t)oldseg) atomicOr(flags, (uint32_t)FA_ERR_OLDOVER);
        }
        if (lane == 0) { g.r_ncls[r] = (uint32_t)__popcll(present); g.r_nent[r] = nent; }
    }
    // pass 2 -- the position of every entry inside the range's block: segment of its class, straddlers first (in x order here;
    // k_sa_rows sorts them by id), then the entries first seen in this range in x order
    for (long cb4 = x0; cb4 < x1; cb4 += 256) {
        unsigned cdv[4], clv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long x = cb4 + 64 * q + lane;
            cdv[q] = x < x1 ? sb.code[x] : 0u;
            clv[q] = (WITH_EP && x < x1) ? sb.cls[x] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long cb = cb4 + 64 * q;
            if (cb >= x1) break;
            const long x = cb + lane;
            const unsigned cd = cdv[q], cb8 = clv[q];
            const bool en = (cd & SA_ENT) && !(cd & SA_OLD), eo = (cd & SA_OLD) != 0;
            if (!WITH_EP) {
                const unsigned long long mn = __ballot(en), mo = __ballot(eo);
                const uint32_t co0 = (uint32_t)__builtin_amdgcn_readlane((int)co, 0);
                const uint32_t rn = (uint32_t)__builtin_amdgcn_readlane((int)runn, 0), ro = (uint32_t)__builtin_amdgcn_readlane((int)runo, 0);
                if (en) rel[x] = (REL)(co0 + rn + (uint32_t)__popcll(mn & lt));
                if (eo) rel[x] = (REL)(ro + (uint32_t)__popcll(mo & lt));
                if (lane == 0) { runn += (uint32_t)__popcll(mn); runo += (uint32_t)__popcll(mo); }
                continue;
            }
            const int n = (int)(cb8 >> 6), c0 = (int)(cb8 & 63);
            unsigned long long bits = sa_wave_or64((en || eo) ? ((1ull << c0) | (n == 2 ? (2ull << c0) : 0ull)) : 0ull);
            while (bits) {
                const int c = __builtin_ctzll(bits);
                bits &= bits - 1;
                const bool s0 = (en || eo) && c0 == c, s1 = (en || eo) && n == 2 && c0 + 1 == c;
                const unsigned long long mn = __ballot((s0 || s1) && en), mo = __ballot((s0 || s1) && eo);
                const uint32_t st_c = (uint32_t)__builtin_amdgcn_readlane((int)start, c), co_c = (uint32_t)__builtin_amdgcn_readlane((int)co, c);
                const uint32_t rn = (uint32_t)__builtin_amdgcn_readlane((int)runn, c), ro = (uint32_t)__builtin_amdgcn_readlane((int)runo, c);
                if (s0 || s1) {
                    const uint32_t v = en ? st_c + co_c + rn + (uint32_t)__popcll(mn & lt) : st_c + ro + (uint32_t)__popcll(mo & lt);
                    rel[2 * x + (s1 ? 1 : 0)] = (REL)v;
                }
                if (lane == c) { runn += (uint32_t)__popcll(mn); runo += (uint32_t)__popcll(mo); }
            }
        }
    }
}

// ---- S3: the entries, the dims table, Mw / the rows -----------------------------------------------------------------------------
// G_ROWS: A/E-row matrix -- o is the matrix (CSR in place), Mw of the P side.  Otherwise I/X-row matrix -- o is the matrix (rows
// written here, one per P key), sval / sdid the scratch copy of the terms in (range, class) order for the column sums.
template <bool WITH_EP, bool G_ROWS, typename REL, int CPT>
__global__ __launch_bounds__(SA_T) void k_sa_emit(RgView rg, PlanView pl, MatSpec s, FaG g, FaP p, SaBuf sb, int merge, FaOut o,
                                                  double *__restrict__ sval, int32_t *__restrict__ sdid) {
    __shared__ double s_hc[WITH_EP ? HC_LDS : 1];
    stage_hc<WITH_EP>(rg, s_hc);
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    long cb = sb.gx0 + (long)blockIdx.x * (SA_T * CPT);
    const long stride = (long)gridDim.x * (SA_T * CPT);         // (a grid smaller than the slice: every workgroup walks several tiles)
    const REL *__restrict__ rel = static_cast<const REL *>(sb.rel);      // (elevation classes only)
    constexpr int S = WITH_EP ? 2 : 1;
    const bool pkey_x = p.key == KEY_X;
    if (blockIdx.x == 0 && tid == 0 && !sb.rli) o.rowptr[sb.end_row] = (int32_t)sb.end_nnz;      // (identity ice rows: closed by the scan)
    // round 1: the code bytes and (one address per wave) the range of each wave's first cell; round 2: everything the cells with
    // entries / with a new P key need -- the range of a cell is its wave's first range + the range heads up to its lane (code bit
    // 7), so the per-range values are fetched in the same round as the cell itself; round 3: the mask value (elevation classes)
    // (a workgroup that walks several tiles issues round 1 of its next tile before it works on this one)
    unsigned cdn[CPT];
    int rbn[CPT];
    auto round1 = [&](long c0) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const long x = c0 + (long)u * SA_T + tid;
            const long xw = c0 + (long)u * SA_T + (tid & ~63);
            cdn[u] = x < sb.gx1 ? sb.code[x] : 0u;
            rbn[u] = pl.aidx[xw < sb.gx1 ? xw : sb.gx1 - 1];
        }
    };
    round1(cb);
    for (; cb < sb.gx1; cb += stride) {
    unsigned cd[CPT];
    int rb[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) { cd[u] = cdn[u]; rb[u] = rbn[u]; }
    if (cb + stride < sb.gx1) round1(cb + stride);
    const unsigned long long le = lt | (1ull << lane);
    int iIv[CPT], rv[CPT], iAv[CPT], hintv[CPT];
    double av[CPT];
    uint32_t lenv[CPT], ebv[CPT], relv[CPT][S], pwv[CPT], lwv[CPT];
    unsigned long long bpv[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const long x = cb + (long)u * SA_T + tid;
        const long xw = cb + (long)u * SA_T + (tid & ~63);
        // (the wave's ballots -- range heads here, new P keys below -- see every lane; the ROLES only the cells of the slice)
        const unsigned long long heads = __ballot((cd[u] & SA_RHEAD) != 0), pbits = __ballot((cd[u] & SA_P) != 0);
        bpv[u] = pbits;
        // one class per range: an entry's place in its range's block = (entries of its kind before it in x order) - (those before
        // the range) -- scans over whole waves + the wave's ballot; straddlers in front (in x order here, k_sa_rows sorts them)
        const unsigned long long bnew = __ballot((cd[u] & SA_ENT) && !(cd[u] & SA_OLD)), bold = __ballot((cd[u] & SA_OLD) != 0);
        if (x < sb.sx0 || x >= sb.sx1) cd[u] = 0u;
        const bool want = (cd[u] & (SA_ENT | SA_P)) != 0, ent = (cd[u] & SA_ENT) != 0;
        rv[u] = rb[u] + (int)__popcll(heads & le & ~1ull);
        iIv[u] = want ? pl.exI[x] : 0;
        av[u] = ent ? rg.area[x] : 0.0;
        ebv[u] = ent ? g.ebase[rv[u]] : 0u;
        iAv[u] = (ent && (WITH_EP || !G_ROWS)) ? pl.riA[rv[u]] : 0;      // (an A-row matrix without classes never looks at the atmosphere index here)
        if (WITH_EP) {
#pragma unroll
            for (int j = 0; j < S; ++j) relv[u][j] = ent ? (uint32_t)rel[(size_t)S * x + j] : 0u;
        } else if (ent) {
            const bool isold = (cd[u] & SA_OLD) != 0;
            const int rr = rv[u];
            const uint32_t wbase = xw < sb.gx1 ? (isold ? sb.Eow[xw >> 6] : sb.Ew[xw >> 6]) : 0u;
            relv[u][0] = isold ? wbase + (uint32_t)__popcll(bold & lt) - sb.eobase[rr]
                               : g.ecnto[rr] + wbase + (uint32_t)__popcll(bnew & lt) - sb.ewbase[rr];
        } else relv[u][0] = 0u;
        lenv[u] = (!G_ROWS && x < sb.gx1) ? sb.rl[x] : 0u;
        hintv[u] = -1;
        if (WITH_EP && ent) {
            // the class search was done by k_sa_flags: byte = first class | classes << 6.  Z_INTERP: two classes -> (c0, c0 + 1); one
            // class -> the elevation sits exactly on class c0, which lower_bound reports as the UPPER index (the lower one only at 0)
            const unsigned cb8 = sb.cls[x];
            const int c0 = (int)(cb8 & 63), n = (int)(cb8 >> 6);
            hintv[u] = rg.interp == 0 ? (n == 2 ? c0 + 1 : (c0 == 0 ? 1 : c0)) : c0;
        }
        pwv[u] = (p.fresh && xw < sb.gx1) ? sb.Pw[xw >> 6] : 0u;
        lwv[u] = (!G_ROWS && sb.Lw && xw < sb.gx1) ? sb.Lw[xw >> 6] : 0u;
    }
    double ev[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) ev[u] = (WITH_EP && (cd[u] & SA_ENT)) ? rg.em[iIv[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const long x = cb + (long)u * SA_T + tid;
        const bool in = x >= sb.sx0 && x < sb.sx1;
        const bool P = (cd[u] & SA_P) != 0, ent = (cd[u] & SA_ENT) != 0, old = (cd[u] & SA_OLD) != 0;
        const unsigned long long bp = bpv[u];
        const int iI = iIv[u], r = rv[u];
        // dense id of the P key first seen at this cell
        int pown;
        if (p.fresh) pown = (int)pwv[u] + (int)__popcll(bp & lt);
        else pown = pkey_x ? (int)x : iI;
        GEnt ge;
        ge.n = 0; ge.cls0 = ge.cls1 = 0; ge.t0 = ge.t1 = 0.0; ge.gkey0 = ge.gkey1 = 0;
        const long iA = iAv[u];
        if (ent) {
            const XCell c = make_cell<WITH_EP>(rg, iA, iI, av[u], ev[u], hintv[u]);
            if (!(WITH_EP && c.range_error)) fa_group<WITH_EP>(rg, pl, s, G_ROWS, merge != 0 && (cd[u] & SA_DUPNEXT), c, x, ge);
        }
        if (ge.n > 0) {
            // the entries of this group go to their slots of the range's block
            int did;
            if (!p.fresh) did = pkey_x ? (int)x : iI;
            else if (pkey_x || (cd[u] & SA_FIRST)) did = pown;
            else did = (int)sa_prank_at(sb.Pw, sb.pbits, pl.ifirst[iI]);
#pragma unroll
            for (int j = 0; j < S; ++j) {
                if (j >= ge.n) break;
                const uint32_t pos = ebv[u] + relv[u][j];
                if (G_ROWS) { o.colind[pos] = did; o.val[pos] = ge.t(j); }
                else { sval[pos] = ge.t(j); if (old) sdid[pos] = did; }
            }
        }
        // ranks of this group's classes in their range (the order of its entries along a row / a column)
        bool swap01 = false;
        int id0 = 0, id1 = 0;
        if (ge.n > 0 && (!G_ROWS || ge.n > 1)) {
            if (G_ROWS) swap01 = g.erank[(size_t)r * g.NC + ge.cls0] > g.erank[(size_t)r * g.NC + ge.cls1];
            else {
                id0 = fa_gdense(g, r, ge.cls0, ge.gkey0);
                id1 = ge.n > 1 ? fa_gdense(g, r, ge.cls1, ge.gkey1) : 0;
                swap01 = ge.n > 1 && id0 > id1;
            }
        }
        if (G_ROWS) {
            if (P) {
                if (p.fresh) p.to_sparse[pown] = pkey_x ? x : (long)iI;
                // Mw of an exchange cell / of an ice cell with this one exchange cell: its <= 2 entries in ascending row order,
                // from zero (spsparse sum()); a member without entries gets 0.  Ice cells with several cells: k_fa_pelem<SUMS>.
                if (pkey_x || (cd[u] & SA_ONE)) {
                    double sum = 0.0;
                    if (ge.n == 1) sum = sum + ge.t0;
                    else if (ge.n == 2) { sum = sum + (swap01 ? ge.t1 : ge.t0); sum = sum + (swap01 ? ge.t0 : ge.t1); }
                    o.Mw[pown] = sum;
                }
            }
        } else {
            // rows come out in x order of the cells that own them: offset = row entries before this wave + before this lane
            // (an identity ICE set: rows lie in ice-cell order -- the row pointer is the scan over the ice cells, done before this pass,
            // wM of the cells that are no members was cleared with it)
            const bool by_ice = sb.rli != nullptr;
            const uint32_t before = by_ice ? 0u : sa_wave_excl_u32(lenv[u], lane);
            const uint32_t b0 = by_ice ? (P ? (uint32_t)o.rowptr[pown] : 0u) : lwv[u] + before;
            if (!by_ice) {
                if (in && (P || !p.fresh)) o.rowptr[pown] = (int32_t)b0;
                if (in && !P && !p.fresh) o.wM[pown] = 0.0;              // identity set: a cell that is no member has an empty row
            }
            if (P) {
                if (p.fresh) p.to_sparse[pown] = pkey_x ? x : (long)iI;
                if (pkey_x || (cd[u] & SA_ONE)) {
                    // FAM_IVAE rows (RegridMatrices_Dynamic.cpp:201-233): wM = rowsum by ascending column, M = [1/wM] * T [* sApvA]
                    double sum = 0.0;
                    const double ta = swap01 ? ge.t1 : ge.t0, tb = swap01 ? ge.t0 : ge.t1;
                    if (ge.n >= 1) sum = sum + ta;
                    if (ge.n == 2) sum = sum + tb;
                    o.wM[pown] = sum;
                    auto finish = [&](double t) {
                        double v = t;
                        if (o.scale) v = (1.0 / sum) * v;
                        if (o.correctA) v = v * rg.ratioA[iA];
                        return v;
                    };
                    if (ge.n >= 1) { o.colind[b0] = swap01 ? id1 : id0; o.val[b0] = finish(ta); }
                    if (ge.n == 2) { o.colind[b0 + 1] = swap01 ? id0 : id1; o.val[b0 + 1] = finish(tb); }
                }
            }
        }
    }
    }
}

// ---- S4a: the ice cells across ONE GCM-cell edge (two exchange cells in two ranges: nearly all ice cells with several cells) -----
// One thread per pair from three static arrays (first cell, second cell, ice cell): every dynamic load -- the two areas, the mask
// value, the code byte, the id -- is independent of the others, where k_fa_pelem walks mlist -> ilptr -> ilist -> cell (measured on
// the Antarctic AvI: 528 MB fetched for 2.7 M ice cells).  G_ROWS: Mw of the ice cell (its <= 4 entries in ascending row order,
// from zero); otherwise its row (ascending column, wM, scaling) at the offset k_sa_emit recorded.
template <bool WITH_EP, bool G_ROWS>
__global__ __launch_bounds__(256) void k_sa_pairs(RgView rg, PlanView pl, MatSpec s, FaG g, FaP p, SaBuf sb, const int32_t *__restrict__ px1,
                                                   const int32_t *__restrict__ px2, const int32_t *__restrict__ piI, const double *__restrict__ pa1,
                                                   const double *__restrict__ pa2, int npair, FaOut o) {
    __shared__ double s_hc[WITH_EP ? HC_LDS : 1];
    stage_hc<WITH_EP>(rg, s_hc);
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= npair) return;
    const int x1 = px1[k], x2 = px2[k], iI = piI[k];
    if (x1 < sb.sx0 || x1 >= sb.sx1) return;                     // (the rank whose slice holds the first cell serves the ice cell)
    if (!(sb.code[x1] & SA_P)) return;                           // masked (or no area at all): not a member of the I set
    const double e = WITH_EP ? rg.em[iI] : 0.0;
    const double a1 = pa1[k], a2 = pa2[k];                       // (static: stored beside the pair)
    // (an A-row matrix without classes needs neither the ranges nor the atmosphere cells: its terms are functions of the areas)
    constexpr bool NEED_R = WITH_EP || !G_ROWS;
    const int r1 = NEED_R ? pl.aidx[x1] : 0, r2 = NEED_R ? pl.aidx[x2] : 0;
    const int did = p.fresh ? (int)sa_prank_at(sb.Pw, sb.pbits, x1) : iI;
    const long iA1 = NEED_R ? pl.riA[r1] : 0, iA2 = NEED_R ? pl.riA[r2] : 0;
    GEnt g1, g2;
    {
        const XCell c1 = make_cell<WITH_EP>(rg, iA1, iI, a1, e), c2 = make_cell<WITH_EP>(rg, iA2, iI, a2, e);
        g1.n = g2.n = 0; g1.cls0 = g1.cls1 = g2.cls0 = g2.cls1 = 0; g1.t0 = g1.t1 = g2.t0 = g2.t1 = 0.0; g1.gkey0 = g1.gkey1 = g2.gkey0 = g2.gkey1 = 0;
        if (!(WITH_EP && c1.range_error)) {
            fa_group<WITH_EP>(rg, pl, s, G_ROWS, false, c1, x1, g1);
            fa_group<WITH_EP>(rg, pl, s, G_ROWS, false, c2, x2, g2);
        }
    }
    if (G_ROWS) {
        // rows ascend with the range, inside a range with the first-seen rank of the class
        double sum = 0.0;
        auto add = [&](const GEnt &ge, int r) {
            if (ge.n == 1) sum = sum + ge.t0;
            else if (ge.n == 2) {
                const bool sw = g.erank[(size_t)r * g.NC + ge.cls0] > g.erank[(size_t)r * g.NC + ge.cls1];
                sum = sum + (sw ? ge.t1 : ge.t0); sum = sum + (sw ? ge.t0 : ge.t1);
            }
        };
        add(g1, r1); add(g2, r2);
        o.Mw[did] = sum;
        return;
    }
    // the row: <= 4 entries (id, term, atmosphere cell) into ascending column order (a pre-populated column set may list them in
    // any order), wM = their sum in that order, M = [1/wM] * T [* sApvA]   (RegridMatrices_Dynamic.cpp:201-233)
    int id[4] = {0, 0, 0, 0};
    double t[4] = {0, 0, 0, 0};
    long ia[4] = {0, 0, 0, 0};
    int n = 0;
    auto put = [&](const GEnt &ge, int r, long iA) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j >= ge.n) break;
            const int d = fa_gdense(g, r, ge.cls(j), ge.gkey(j));
            // insertion into the sorted prefix (n <= 3 here)
            int pos = n;
#pragma unroll
            for (int q = 3; q >= 1; --q)
                if (q <= n && q == pos && id[q - 1] > d) { id[q] = id[q - 1]; t[q] = t[q - 1]; ia[q] = ia[q - 1]; pos = q - 1; }
            id[pos] = d; t[pos] = ge.t(j); ia[pos] = iA;
            ++n;
        }
    };
    put(g1, r1, iA1); put(g2, r2, iA2);
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (q < n) sum = sum + t[q];
    o.wM[did] = sum;
    const int b0 = o.rowptr[did];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= n) break;
        double v = t[q];
        if (o.scale) v = (1.0 / sum) * v;
        if (o.correctA) v = v * rg.ratioA[ia[q]];
        o.colind[b0 + q] = id[q]; o.val[b0 + q] = v;
    }
}

// ---- S5: FOUR segments per wave -- straddlers into place, sequential sums, weights, scaling -------------------------------------
// A segment = the entries of one (range, class): a row of an A/E-row matrix, or the terms of one column of an I/X-row matrix
// (scratch copy).  The sum must run in sequence (spsparse sum(): ascending column / row).  One wave replaying ONE segment's chain
// from registers (v_readlane) spends three wave-wide instructions per entry: VALU-bound (a1h AvI 139 us, EvI 310).  Here every
// 16-lane row of a wave owns a segment -- the four rows of a range group (one-class matrices: four consecutive ranges; elevation
// classes: four classes of one range) -- holds 16 consecutive values of it per register, and step j of the chain broadcasts lane j
// of EVERY row to its row (DPP row_newbcast: two moves) before one add: three instructions serve four chains.  Shorter segments are
// padded with +0.0 (a sum that starts from +0.0 is never -0.0, so sum + 0.0 == sum bitwise).  Loads stay coalesced (128 bytes per
// row), everything of a segment is issued up front and kept in registers for the scaling.  scratch/sload_test.hip, 17.5 M values
// in 50 000 segments: v_readlane chain 204 us, scalar-load chain 106, a thread per segment 40 (cannot scale in place: its stores
// touch 64 lines per instruction), this form 48.5.
// EMIT: the rows of an A/E-row matrix (CSR in place: o.colind / o.val).  Otherwise the column sums of an I/X-row matrix over the
// scratch copy of the terms (o.val = scratch, sdid = the row ids of the straddling terms).
template <int J> __device__ __forceinline__ double sa_rowbcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + J, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + J, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#define SA_CHAIN16(SUM, V)                                                                                           \
    do {                                                                                                             \
        SUM = SUM + sa_rowbcast<0>(V); SUM = SUM + sa_rowbcast<1>(V); SUM = SUM + sa_rowbcast<2>(V); SUM = SUM + sa_rowbcast<3>(V);     \
        SUM = SUM + sa_rowbcast<4>(V); SUM = SUM + sa_rowbcast<5>(V); SUM = SUM + sa_rowbcast<6>(V); SUM = SUM + sa_rowbcast<7>(V);     \
        SUM = SUM + sa_rowbcast<8>(V); SUM = SUM + sa_rowbcast<9>(V); SUM = SUM + sa_rowbcast<10>(V); SUM = SUM + sa_rowbcast<11>(V);   \
        SUM = SUM + sa_rowbcast<12>(V); SUM = SUM + sa_rowbcast<13>(V); SUM = SUM + sa_rowbcast<14>(V); SUM = SUM + sa_rowbcast<15>(V); \
    } while (0)
__device__ __forceinline__ uint32_t sa_row_sum_u32(uint32_t v) {      // over the 16 lanes of a row
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}
__device__ __forceinline__ int sa_row_max_i32(int v) {
    v = max(v, __shfl_xor(v, 1, 64)); v = max(v, __shfl_xor(v, 2, 64)); v = max(v, __shfl_xor(v, 4, 64)); v = max(v, __shfl_xor(v, 8, 64));
    return v;
}
template <bool EMIT, int OLD, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_sa_rows(RgView rg, PlanView pl, FaG g, FaOut o, int32_t *__restrict__ sdid, int sr0, int sr1) {
    __shared__ int s_did[WPB][4][OLD], s_did2[WPB][4][OLD];
    __shared__ double s_t[WPB][4][OLD], s_t2[WPB][4][OLD];
    constexpr int NV = 8;                                        // chunks of 16 values per segment loaded up front (and kept for the scaling)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane >> 4, li = lane & 15;
    const int wid = (int)blockIdx.x * WPB + wv;
    // one-class matrices: the rows of a wave take four consecutive ranges; elevation classes: four classes of one range, the
    // groups of four dealt to grid dimension y
    const bool one = g.NC == 1;
    const int r = one ? sr0 + wid * 4 + sub : sr0 + wid;
    const int q0 = one ? 0 : (int)blockIdx.y * 4 + sub, qstride = one ? (1 << 28) : (int)gridDim.y * 4;
    const bool rvalid = r < sr1;
    const int ncls = rvalid ? (int)g.r_ncls[r] : 0;
    if (!__ballot(q0 < ncls)) return;                            // (wave-uniform: nothing for any row of this wave)
    const uint32_t eb = rvalid ? g.ebase[r] : 0u;
    const long iA = rvalid ? pl.riA[r] : 0;
    // the class table of the row's range: lane li of the row looks after classes li, li + 16, li + 32, li + 48
    int rk[4];
    uint32_t sg[4], od[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = li + 16 * k;
        rk[k] = -1; sg[k] = 0; od[k] = 0;
        if (rvalid && c < g.NC) {
            rk[k] = g.erank[(size_t)r * g.NC + c];
            od[k] = g.ecnto[(size_t)r * g.NC + c];
            sg[k] = g.ecntn[(size_t)r * g.NC + c] + od[k];
        }
    }
    int32_t *__restrict__ ids = EMIT ? o.colind : sdid;
    for (int q = q0;; q += qstride) {
        const bool act = q < ncls;
        if (!__ballot(act)) break;
        int cls_l = -1;
        uint32_t seg_l = 0, no_l = 0, start_l = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (rk[k] == q) { cls_l = li + 16 * k; seg_l = sg[k]; no_l = od[k]; }
            if (rk[k] >= 0 && rk[k] < q) start_l += sg[k];
        }
        const int cls = sa_row_max_i32(cls_l);
        uint32_t seg = sa_row_sum_u32(seg_l), no = sa_row_sum_u32(no_l);
        const uint32_t start = sa_row_sum_u32(start_l);
        if (!act || cls < 0) { seg = 0; no = 0; }
        if (no > (uint32_t)OLD) no = OLD;                        // (the build is being discarded: FA_ERR_OLDOVER; stay inside the tables)
        const uint32_t b = eb + start, e = b + seg, bn = b + no;
        // every load of the segment is issued up front: the straddling entries (ice cells first seen in an earlier range: smaller
        // ids; k_sa_ranges placed them at the segment's start in x order) and the first NV x 16 entries first seen in this range
        for (uint32_t k = li; k < no; k += 16) { s_did[wv][sub][k] = ids[b + k]; s_t[wv][sub][k] = o.val[b + k]; }
        double v[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const uint32_t k = bn + 16u * c + li;
            v[c] = k < e ? o.val[k] : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the straddlers into ascending id order -- inside a class the ids are distinct: rank by counting
        for (uint32_t k = li; k < no; k += 16) {
            const int did = s_did[wv][sub][k];
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < no; ++j) cnt += s_did[wv][sub][j] < did ? 1u : 0u;
            s_did2[wv][sub][cnt] = did; s_t2[wv][sub][cnt] = s_t[wv][sub][k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double sum = 0.0;
        for (uint32_t k = 0; k < no; ++k) sum = sum + s_t2[wv][sub][k];
        // the chains of the wave's four segments, in lock step; the longest one bounds the loop (wave-uniform)
        const uint32_t len = e - bn;
        uint32_t mx = max(len, (uint32_t)__shfl_xor((int)len, 16, 64));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, 32, 64));
        mx = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            if (16u * c < mx) SA_CHAIN16(sum, v[c]);             // (wave-uniform condition)
        }
        for (uint32_t base = 16u * NV; base < mx; base += 128) {          // longer segments: eight chunks in flight per step
            double u8[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t k = bn + base + 16u * c + li;
                u8[c] = k < e ? o.val[k] : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (base + 16u * c < mx) SA_CHAIN16(sum, u8[c]);
            }
        }
        if (act && cls >= 0) {
            const long gkey = g.key == KEY_E ? iA * rg.sA + (long)cls * rg.sHC : iA;
            const int gd = fa_gdense(g, r, cls, gkey);
            if (gd >= 0) {                                     // (< 0: a key the pre-populated set lacks -- the build is discarded, FA_ERR_MISSING)
                if (EMIT) {                                    // FAM_AEVI rows (RegridMatrices_Dynamic.cpp:100-146)
                    double wM, mul = 1.0;
                    if (o.correctA) {
                        const double rr = ratio_of(rg, g.key, gkey);
                        wM = rr * sum;
                        if (o.scale) mul = (1.0 / rr) * (1.0 / sum);
                    } else {
                        wM = sum;
                        if (o.scale) mul = 1.0 / sum;
                    }
                    if (li == 0) {
                        if (g.to_sparse) g.to_sparse[gd] = gkey;
                        o.wM[gd] = wM;
                        o.rowptr[gd] = (int32_t)b;
                    }
                    for (uint32_t k = li; k < no; k += 16) {
                        o.colind[b + k] = s_did2[wv][sub][k];
                        o.val[b + k] = o.scale ? mul * s_t2[wv][sub][k] : s_t2[wv][sub][k];
                    }
                    if (o.scale) {
#pragma unroll
                        for (int c = 0; c < NV; ++c) {
                            const uint32_t k = bn + 16u * c + li;
                            if (k < e) o.val[k] = mul * v[c];
                        }
                        for (uint32_t k = bn + 16u * NV + li; k < e; k += 16) o.val[k] = mul * o.val[k];
                    }
                } else if (li == 0) {                          // FAM_IVAE columns (:201-233)
                    if (g.to_sparse) g.to_sparse[gd] = gkey;
                    o.Mw[gd] = o.correctA ? ratio_of(rg, g.key, gkey) * sum : sum;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The same with ONE segment per wave (v_readlane chain, 8 x 64 values in registers): grids with long ranges -- few, long segments,
// where four lock-stepped chains per wave only quarter the number of waves in flight (1 km Greenland, 1 143 segments of ~1 700
// entries: 0.184 ms per AvI build with this kernel, 0.250 with the four-segment one).
template <bool EMIT, int OLDSEG>
__global__ __launch_bounds__(256) void k_sa_rows1(RgView rg, PlanView pl, FaG g, FaOut o, int32_t *__restrict__ sdid, int sr0, int sr1) {
    __shared__ int s_did[4][OLDSEG], s_did2[4][OLDSEG];
    __shared__ double s_t[4][OLDSEG], s_t2[4][OLDSEG];
    __shared__ double s_ch[4][64];
    constexpr int NV = 8;                                        // chunks of 64 values of a segment loaded up front and kept in registers
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (walking the ranges from the END of the slice first -- the segments the emit pass wrote last, the likeliest to be still in
    // the Infinity Cache -- changes nothing, measured: this kernel is bound by its chains, not by its reads)
    const int r = sr0 + (int)blockIdx.x * 4 + wv;
    if (r >= sr1) return;
    const int ncls = (int)g.r_ncls[r];
    if ((int)blockIdx.y >= ncls) return;
    const uint32_t eb = g.ebase[r];
    int rank = -1;
    uint32_t seg = 0, nold_c = 0;
    if (lane < g.NC) {
        rank = g.erank[(size_t)r * g.NC + lane];
        nold_c = g.ecnto[(size_t)r * g.NC + lane];
        seg = g.ecntn[(size_t)r * g.NC + lane] + nold_c;
    }
    int32_t *__restrict__ ids = EMIT ? o.colind : sdid;
    const long iA = pl.riA[r];
    // the classes of a range are dealt to the waves of grid dimension y (first-seen rank q, q + gridDim.y, ...): a class's segment
    // starts where the segments of the classes ranked before it end
    for (int q = (int)blockIdx.y; q < ncls; q += (int)gridDim.y) {
        const unsigned long long mq = __ballot(rank == q);
        if (!mq) break;                                          // (cannot happen: ranks 0 .. ncls-1 are all taken)
        const int cls = __builtin_ctzll(mq);
        const uint32_t start = (uint32_t)__builtin_amdgcn_readfirstlane((int)sa_wave_sum_u32((rank >= 0 && rank < q) ? seg : 0u));
        const uint32_t sg = (uint32_t)__builtin_amdgcn_readlane((int)seg, cls);
        uint32_t no = (uint32_t)__builtin_amdgcn_readlane((int)nold_c, cls);
        if (no > (uint32_t)OLDSEG) no = OLDSEG;                  // (the build is being discarded: FA_ERR_OLDOVER; stay inside the tables)
        const uint32_t b = eb + start, e = b + sg, bn = b + no;
        // every load of the segment is issued up front: the straddling entries (ice cells first seen in an earlier range: smaller
        // ids; they lie at the segment's start in x order) and the first NV x 64 of the entries first seen in this range
        for (uint32_t k = lane; k < no; k += 64) { s_did[wv][k] = ids[b + k]; s_t[wv][k] = o.val[b + k]; }
        double v[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const uint32_t k = bn + 64u * c + lane;
            v[c] = k < e ? o.val[k] : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the straddlers into ascending id order -- inside a class the ids are distinct: rank by counting
        for (uint32_t k = lane; k < no; k += 64) {
            const int did = s_did[wv][k];
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < no; ++j) cnt += s_did[wv][j] < did ? 1u : 0u;
            s_did2[wv][cnt] = did; s_t2[wv][cnt] = s_t[wv][k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the sequential sum: the straddlers, then the chain over the entries first seen in the range replayed from registers
        // with v_readlane (every lane computes the identical sum)
        double sum = 0.0;
        for (uint32_t k = 0; k < no; ++k) sum = sum + s_t2[wv][k];
        // The chain (round 5): a chunk's 64 values are parked in the wave's LDS row and ONE lane adds them (ds_read_b128 + two
        // adds per pair of terms where the v_readlane replay took six vector instructions; measured on a dependent chain: 9.3
        // against 20.7 cycles per term; this pass is bound by instruction issue).  Values past the segment's end were loaded as +0.0: a chain that
        // starts from +0.0 never holds -0.0, so adding them changes no bit.
        double *chb = s_ch[wv];
        auto chain = [&](double val, uint32_t) {
            chb[lane] = val;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {                  // (all 64 lanes adding: the same time, 64 x the LDS traffic)
#pragma unroll
                for (int j = 0; j < 64; ++j) sum = sum + chb[j];
            }
            __builtin_amdgcn_wave_barrier();
        };
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const uint32_t cb = bn + 64u * c;
            if (cb >= e) break;
            chain(v[c], cb);
        }
        for (uint32_t base = bn + 64u * NV; base < e; base += 256) {          // longer segments: four chunks in flight per step
            double u4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t k = base + 64u * c + lane;
                u4[c] = k < e ? o.val[k] : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t cb = base + 64u * c;
                if (cb >= e) break;
                chain(u4[c], cb);
            }
        }
        sum = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sum)), __builtin_amdgcn_readfirstlane(__double2loint(sum)));
        const long gkey = g.key == KEY_E ? iA * rg.sA + (long)cls * rg.sHC : iA;
        const int gd = fa_gdense(g, r, cls, gkey);
        if (gd < 0) continue;                                   // a key the pre-populated set lacks: the build is discarded (FA_ERR_MISSING)
        if (EMIT) {                                            // FAM_AEVI rows (RegridMatrices_Dynamic.cpp:100-146)
            double wM, mul = 1.0;
            if (o.correctA) {
                const double rr = ratio_of(rg, g.key, gkey);
                wM = rr * sum;
                if (o.scale) mul = (1.0 / rr) * (1.0 / sum);
            } else {
                wM = sum;
                if (o.scale) mul = 1.0 / sum;
            }
            if (lane == 0) {
                if (g.to_sparse) g.to_sparse[gd] = gkey;
                o.wM[gd] = wM;
                o.rowptr[gd] = (int32_t)b;
            }
            for (uint32_t k = lane; k < no; k += 64) {
                o.colind[b + k] = s_did2[wv][k];
                o.val[b + k] = o.scale ? mul * s_t2[wv][k] : s_t2[wv][k];
            }
            if (o.scale) {
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    const uint32_t k = bn + 64u * c + lane;
                    if (k < e) o.val[k] = mul * v[c];
                }
                for (uint32_t k = bn + 64u * NV + lane; k < e; k += 64) o.val[k] = mul * o.val[k];
            }
        } else if (lane == 0) {                                // FAM_IVAE columns (:201-233)
            if (g.to_sparse) g.to_sparse[gd] = gkey;
            o.Mw[gd] = o.correctA ? ratio_of(rg, g.key, gkey) * sum : sum;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- S5, lane-parallel: SIXTEEN segments per wave, one lane per chain ---------------------------------------------------------------
// The two kernels above spend a whole 64-lane instruction on every term of a chain (v_readlane + v_add_f64: ~3 vector instructions
// per term; the pass is bound by instruction issue, not by its reads -- 138 us for the 17.6 M terms of the Antarctic AvI).  Here a wave
// takes R consecutive ranges and deals their segments to lanes 0..15, sixteen at a time: all 64 lanes fetch -- 128 consecutive values
// (1 KB) of each of the sixteen segments per step -- and park them in LDS, row = segment (129 doubles apart: sixteen lanes reading
// one column hit sixteen banks); lane s then adds row s in sequence.  A step moves 16 KB with 32 loads, 32 LDS writes, 128 LDS
// reads and 128 dependent adds: ~0.15 instructions per term.  Lanes whose segment has ended add +0.0 (the chains start from +0.0, so no
// partial sum is ever -0.0: adding +0.0 changes no bit).  The straddlers are sorted into place beforehand by k_sa_oldsort (a wave per
// range: sixteen segments' sorts one after the other inside this kernel cost more than all its chains); scaling re-reads the segment.
template <int OLDSEG>
__global__ __launch_bounds__(256) void k_sa_oldsort(FaG g, int32_t *__restrict__ ids, double *__restrict__ val, int sr0, int sr1) {
    __shared__ int s_did[4][OLDSEG];
    __shared__ double s_t[4][OLDSEG];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = sr0 + (int)blockIdx.x * 4 + wv;
    if (r >= sr1) return;
    const int ncls = (int)g.r_ncls[r];
    const uint32_t eb = g.ebase[r];
    int rank = -1;
    uint32_t seg = 0, nold_c = 0;
    if (lane < g.NC) {
        rank = g.erank[(size_t)r * g.NC + lane];
        nold_c = g.ecnto[(size_t)r * g.NC + lane];
        seg = g.ecntn[(size_t)r * g.NC + lane] + nold_c;
    }
    if (!__ballot(nold_c != 0)) return;
    for (int q = 0; q < ncls; ++q) {
        const unsigned long long mq = __ballot(rank == q);
        if (!mq) break;
        const int cls = __builtin_ctzll(mq);
        uint32_t no = (uint32_t)__builtin_amdgcn_readlane((int)nold_c, cls);
        if (no == 0) continue;
        if (no > (uint32_t)OLDSEG) no = OLDSEG;                  // (the build is being discarded: FA_ERR_OLDOVER; stay inside the tables)
        const uint32_t b = eb + (uint32_t)__builtin_amdgcn_readfirstlane((int)sa_wave_sum_u32((rank >= 0 && rank < q) ? seg : 0u));
        for (uint32_t k = lane; k < no; k += 64) { s_did[wv][k] = ids[b + k]; s_t[wv][k] = val[b + k]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = lane; k < no; k += 64) {               // inside a class the ids are distinct: rank by counting
            const int did = s_did[wv][k];
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < no; ++j) cnt += s_did[wv][j] < did ? 1u : 0u;
            ids[b + cnt] = did; val[b + cnt] = s_t[wv][k];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool EMIT>
__global__ __launch_bounds__(256) void k_sa_rowsL(RgView rg, PlanView pl, FaG g, FaOut o, int sr0, int sr1, int R) {
    constexpr int K = 128, SEGS = 16;
    __shared__ double s_buf[4][SEGS][K + 1];
    __shared__ double s_mul[4][SEGS];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = sr0 + ((int)blockIdx.x * 4 + wv) * R;
    if (r0 >= sr1) return;
    const int nrange = min(R, sr1 - r0);
    // the segments of these ranges, in (range, first-seen rank of the class) order: `off` = segments of the ranges before lane's
    const uint32_t ncls_l = lane < nrange ? g.r_ncls[r0 + lane] : 0u;
    const uint32_t off_l = sa_wave_excl_u32(ncls_l, lane);
    const int T = (int)__builtin_amdgcn_readfirstlane((int)sa_wave_sum_u32(ncls_l));
    for (int base = 0; base < T; base += SEGS) {
        const int j = base + lane;
        const bool valid = lane < SEGS && j < T;
        int ri = 0, q = 0;
        for (int i = 0; i < nrange; ++i) {
            const int oi = __builtin_amdgcn_readlane((int)off_l, i), ni = __builtin_amdgcn_readlane((int)ncls_l, i);
            if (j >= oi && j < oi + ni) { ri = i; q = j - oi; }
        }
        const int r = r0 + ri;
        int cls = 0;
        uint32_t start = 0, sg = 0, no = 0;
        if (valid) {
            if (g.NC == 1) { no = g.ecnto[r]; sg = g.ecntn[r] + no; }
            else {
                for (int c = 0; c < g.NC; ++c) {
                    const int rk = g.erank[(size_t)r * g.NC + c];
                    if (rk < 0 || rk > q) continue;
                    const uint32_t co = g.ecnto[(size_t)r * g.NC + c], cn = g.ecntn[(size_t)r * g.NC + c];
                    if (rk < q) start += cn + co;
                    else { cls = c; no = co; sg = cn + co; }
                }
            }
        }
        const uint32_t b = valid ? g.ebase[r] + start : 0u, e = b + sg;
        const long iA = valid ? pl.riA[r] : 0;
        const long gkey = g.key == KEY_E ? iA * rg.sA + (long)cls * rg.sHC : iA;
        const int gd = valid ? fa_gdense(g, r, cls, gkey) : -1;
        double sum = 0.0;
        // the chains (the straddlers, sorted into place by k_sa_oldsort, then the entries first seen in the range): a step = the next K
        // values of each of the sixteen segments, fetched
        // by whole-wave loads (64 consecutive values of ONE segment per instruction; position and remaining length of a segment are
        // read from its lane into scalar registers), issued one step ahead of the adds
        uint32_t cur = b, left = e - b;                          // (this lane's own chain)
        double v[2 * SEGS];
        auto fetch = [&]() {
#pragma unroll
            for (int i = 0; i < 2 * SEGS; ++i) {
                const uint32_t c_s = (uint32_t)__builtin_amdgcn_readlane((int)cur, i >> 1), l_s = (uint32_t)__builtin_amdgcn_readlane((int)left, i >> 1);
                const uint32_t k = (uint32_t)((i & 1) * 64 + lane);
                v[i] = k < l_s ? o.val[c_s + k] : 0.0;
            }
            cur += K;
            left = left > (uint32_t)K ? left - K : 0u;
        };
        bool more = __ballot(left != 0) != 0;
        if (more) fetch();
        while (more) {
#pragma unroll
            for (int i = 0; i < 2 * SEGS; ++i) s_buf[wv][i >> 1][(i & 1) * 64 + lane] = v[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            more = __ballot(left != 0) != 0;
            if (more) fetch();                                   // (the next step's values fly while this one's are added)
            if (lane < SEGS) {
#pragma unroll
                for (int h = 0; h < K; h += 32) {
                    double t[32];
#pragma unroll
                    for (int k = 0; k < 32; ++k) t[k] = s_buf[wv][lane][h + k];
#pragma unroll
                    for (int k = 0; k < 32; ++k) sum = sum + t[k];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // weights, scaling
        double mul = 1.0;
        const bool live = valid && gd >= 0;                      // (a key the pre-populated set lacks: the build is discarded, FA_ERR_MISSING)
        if (live) {
            if (g.to_sparse) g.to_sparse[gd] = gkey;
            if (EMIT) {                                          // FAM_AEVI rows (RegridMatrices_Dynamic.cpp:100-146)
                double wM;
                if (o.correctA) {
                    const double rr = ratio_of(rg, g.key, gkey);
                    wM = rr * sum;
                    if (o.scale) mul = (1.0 / rr) * (1.0 / sum);
                } else {
                    wM = sum;
                    if (o.scale) mul = 1.0 / sum;
                }
                o.wM[gd] = wM;
                o.rowptr[gd] = (int32_t)b;
            } else o.Mw[gd] = o.correctA ? ratio_of(rg, g.key, gkey) * sum : sum;      // FAM_IVAE columns (:201-233)
        }
        if (EMIT && o.scale) {
            // the sixteen segments of a round are one contiguous block of entries: the wave streams over it, eight loads in flight per
            // lane, every entry times the factor of the segment it lies in (ends of the segments in scalar registers, factors in LDS)
            const int nv = min(SEGS, T - base);
            if (lane < SEGS) s_mul[wv][lane] = live ? mul : 1.0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t es[SEGS];
#pragma unroll
            for (int q2 = 0; q2 < SEGS; ++q2) es[q2] = q2 < nv ? (uint32_t)__builtin_amdgcn_readlane((int)e, q2) : 0xffffffffu;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)b, 0), hi = (uint32_t)__builtin_amdgcn_readlane((int)e, nv - 1);
            for (uint32_t k0 = lo; k0 < hi; k0 += 512) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t k = k0 + 64u * u + lane;
                    x[u] = k < hi ? o.val[k] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t k = k0 + 64u * u + lane;
                    int sgi = 0;
#pragma unroll
                    for (int q2 = 0; q2 < SEGS - 1; ++q2) sgi += es[q2] <= k ? 1 : 0;
                    if (k < hi) o.val[k] = s_mul[wv][sgi] * x[u];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// ---- all exclusive scans of a build in TWO launches ------------------------------------------------------------------------------
// (new P keys per 64 cells, row entries per 64 cells, classes and entries per range: four channels that the generic scans served
// with three launches each -- every launch of this chain costs the ~4 us launch floor whatever it does.)  Tiles of 2048 elements;
// pass A leaves one sum per tile, pass B has every workgroup add up the sums of the tiles before its own (at most a few hundred
// values) and scan its tile; the last tile of a channel stores the channel's total (and, when asked, out[n] = total).
struct MsCh { const void *in; uint32_t *out; uint32_t *total; long n; int u8; int close; };       // close: also write out[n]
struct MsArgs { MsCh ch[4]; int tile0[5]; int nch; };
constexpr int MS_T = 256, MS_I = 8, MS_TILE = MS_T * MS_I;
__device__ __forceinline__ uint32_t ms_load(const MsCh &c, long i) {
    return c.u8 ? (uint32_t)static_cast<const uint8_t *>(c.in)[i] : static_cast<const uint32_t *>(c.in)[i];
}
__device__ __forceinline__ uint32_t ms_block_sum(uint32_t v, uint32_t *s_w) {
    v = sa_wave_sum_u32(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return t;
}
__global__ __launch_bounds__(MS_T) void k_ms_sums(MsArgs a, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_w[4];
    int c = 0;
    while (c + 1 < a.nch && (int)blockIdx.x >= a.tile0[c + 1]) ++c;
    const MsCh ch = a.ch[c];
    const long base = (long)((int)blockIdx.x - a.tile0[c]) * MS_TILE;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < MS_I; ++i) {
        const long k = base + (long)i * MS_T + threadIdx.x;
        if (k < ch.n) v += ms_load(ch, k);
    }
    const uint32_t t = ms_block_sum(v, s_w);
    if (threadIdx.x == 0) sums[blockIdx.x] = t;
}
__global__ __launch_bounds__(MS_T) void k_ms_apply(MsArgs a, const uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t tile[MS_TILE + MS_T];                    // +1 pad per 8: thread t owns tile[9t .. 9t+7]
    int c = 0;
    while (c + 1 < a.nch && (int)blockIdx.x >= a.tile0[c + 1]) ++c;
    const MsCh ch = a.ch[c];
    const int tl = (int)blockIdx.x - a.tile0[c], ntile = a.tile0[c + 1] - a.tile0[c];
    uint32_t before = 0;
    for (int j = threadIdx.x; j < tl; j += MS_T) before += sums[a.tile0[c] + j];
    before = ms_block_sum(before, s_w);
    const long base = (long)tl * MS_TILE;
#pragma unroll
    for (int i = 0; i < MS_I; ++i) {
        const int e = i * MS_T + threadIdx.x;
        const long k = base + e;
        tile[e + (e >> 3)] = k < ch.n ? ms_load(ch, k) : 0u;
    }
    __syncthreads();
    uint32_t v[MS_I], sum = 0;
#pragma unroll
    for (int i = 0; i < MS_I; ++i) { v[i] = tile[threadIdx.x * 9 + i]; sum += v[i]; }
    // exclusive scan of the per-thread sums over the workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ex = sa_wave_excl_u32(sum, lane);
    if (lane == 63) s_w[wv] = ex + sum;
    __syncthreads();
    uint32_t run = before + ex;
    for (int q = 0; q < wv; ++q) run += s_w[q];
    const uint32_t tot = before + s_w[0] + s_w[1] + s_w[2] + s_w[3];
#pragma unroll
    for (int i = 0; i < MS_I; ++i) { tile[threadIdx.x * 9 + i] = run; run += v[i]; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MS_I; ++i) {
        const int e = i * MS_T + threadIdx.x;
        const long k = base + e;
        if (k < ch.n) ch.out[k] = tile[e + (e >> 3)];
    }
    if (tl == ntile - 1 && threadIdx.x == 0) {
        if (ch.total) *ch.total = tot;
        if (ch.close) ch.out[ch.n] = tot;
    }
}
static void sa_scan_channels(const MsCh *chs, int nch, hipStream_t st) {
    MsArgs a{};
    int t = 0, n = 0;
    for (int i = 0; i < nch; ++i) {
        if (chs[i].n <= 0) {                                     // an empty channel: its total (and closing entry) is zero
            if (chs[i].total) IBH_HIP(hipMemsetAsync(chs[i].total, 0, sizeof(uint32_t), st));
            if (chs[i].close) IBH_HIP(hipMemsetAsync(chs[i].out, 0, sizeof(uint32_t), st));
            continue;
        }
        a.ch[n] = chs[i]; a.tile0[n] = t; t += (int)ceil_div(chs[i].n, (long)MS_TILE); ++n;
    }
    a.tile0[n] = t; a.nch = n;
    if (n == 0) return;
    uint32_t *sums = arena().get<uint32_t>((size_t)t);
    hipLaunchKernelGGL(k_ms_sums, dim3(t), dim3(MS_T), 0, st, a, sums);
    hipLaunchKernelGGL(k_ms_apply, dim3(t), dim3(MS_T), 0, st, a, (const uint32_t *)sums);
}

// sum of the row-length bytes before cell x inside its wave (the P-key counterpart is sa_prank_at)
__device__ __forceinline__ uint32_t sa_lrank_at(const uint32_t *__restrict__ Lw, const uint8_t *__restrict__ rl, long x) {
    const long w = x >> 6;
    const int k = (int)(x & 63);
    uint32_t v = Lw[w];
    for (int j = 0; j < k; ++j) v += rl[(w << 6) + j];
    return v;
}
// a rank's share of a sharded build: keys / row entries of ITS cells [sx0, sx1) from the scans that started at gx0
// out: [2] new P keys, [5] row entries, [6] P keys before sx0 in the local numbering, [7] row entries before sx0
__global__ void k_sa_slice_counts(SaBuf sb, int p_fresh, int prows, uint32_t *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t p0 = 0, p1 = 0, l0 = 0, l1 = 0;
    if (p_fresh) { p0 = sa_prank_at(sb.Pw, sb.pbits, sb.sx0); p1 = sa_prank_at(sb.Pw, sb.pbits, sb.sx1); }
    if (prows) { l0 = sa_lrank_at(sb.Lw, sb.rl, sb.sx0); l1 = sa_lrank_at(sb.Lw, sb.rl, sb.sx1); }
    out[2] = p1 - p0; out[5] = l1 - l0; out[6] = p0; out[7] = l0;
}
// local scans -> global numbering: Pw, Lw over the slice's waves, gbase / ebase over its ranges
__global__ void k_sa_shift(uint32_t *__restrict__ Pw, uint32_t *__restrict__ Lw, long w0, long nw, uint32_t dP, uint32_t dL,
                           uint32_t *__restrict__ gbase, uint32_t *__restrict__ ebase, int r0, int nr, uint32_t dG, uint32_t dE) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nw) { if (Pw) Pw[w0 + i] += dP; if (Lw) Lw[w0 + i] += dL; }
    if (i <= nr) { gbase[r0 + i] += dG; ebase[r0 + i] += dE; }
}
__global__ void k_sa_set_u32(uint32_t *p, uint32_t v) { *p = v; }
// Shared build of an A/E-row matrix on an IDENTITY P set: Mw is indexed by sparse position, a rank's sums are scattered over it --
// but they are exactly the P keys first seen in its slice, which the flag pass counted and the scans numbered.  Every such cell
// puts {sparse position, its sum} at its first-seen rank; the pairs of all ranks are gathered and scattered on every rank.
__global__ void k_sa_pack_mw(SaBuf sb, const int32_t *__restrict__ exI, int pkey_x, const double *__restrict__ Mw, int32_t *__restrict__ cidx,
                             double *__restrict__ cval) {
    const long x = sb.sx0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= sb.sx1 || !(sb.code[x] & SA_P)) return;
    const uint32_t k = sa_prank_at(sb.Pw, sb.pbits, x);
    const int32_t pos = pkey_x ? (int32_t)x : exI[x];
    cidx[k] = pos; cval[k] = Mw[pos];
}
// Shared build of an I-row matrix on the IDENTITY ice set (the coupler's IvE, IceCoupler.cpp:462): its rows lie in ice-cell order, so
// a rank's rows are scattered over the matrix.  (a) Before the row pointer can be scanned every rank needs every ice cell's row
// length: the owners' {ice cell, length} pairs travel at the cells' first-seen ranks.  (b) After the passes a rank's rows -- written
// at their final places -- are copied into first-seen order, gathered, and copied out to their places on every other rank.
__global__ void k_sa_pack_keys(SaBuf sb, const int32_t *__restrict__ exI, int32_t *__restrict__ kidx, uint8_t *__restrict__ klen) {
    const long x = sb.sx0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= sb.sx1 || !(sb.code[x] & SA_P)) return;
    const uint32_t k = sa_prank_at(sb.Pw, sb.pbits, x);
    kidx[k] = exI[x]; klen[k] = sb.rl[x];
}
__global__ void k_sa_unpack_rli(const int32_t *__restrict__ kidx, const uint8_t *__restrict__ klen, long k0, long k1, long n, uint8_t *__restrict__ rli) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || (k >= k0 && k < k1) || !klen[k]) return;
    rli[kidx[k]] = klen[k];
}
__global__ void k_sa_pack_rows(SaBuf sb, const int32_t *__restrict__ exI, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                               const double *__restrict__ val, const double *__restrict__ wM, int32_t *__restrict__ ccol, double *__restrict__ cvl,
                               double *__restrict__ kwM, uint32_t *__restrict__ klo) {
    const long x = sb.sx0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= sb.sx1 || !(sb.code[x] & SA_P)) return;
    const uint32_t k = sa_prank_at(sb.Pw, sb.pbits, x), lo = sa_lrank_at(sb.Lw, sb.rl, x);
    const int iI = exI[x], len = sb.rl[x], b = rowptr[iI];
    kwM[k] = wM[iI]; klo[k] = lo;
    for (int j = 0; j < len; ++j) { ccol[lo + j] = colind[b + j]; cvl[lo + j] = val[b + j]; }
}
__global__ void k_sa_unpack_rows(const int32_t *__restrict__ kidx, const uint8_t *__restrict__ klen, const uint32_t *__restrict__ klo,
                                 const double *__restrict__ kwM, const int32_t *__restrict__ ccol, const double *__restrict__ cvl, long k0, long k1, long n,
                                 const int32_t *__restrict__ rowptr, int32_t *__restrict__ colind, double *__restrict__ val, double *__restrict__ wM) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || (k >= k0 && k < k1)) return;
    const int iI = kidx[k], len = klen[k], b = rowptr[iI];
    const uint32_t lo = klo[k];
    wM[iI] = kwM[k];
    for (int j = 0; j < len; ++j) { colind[b + j] = ccol[lo + j]; val[b + j] = cvl[lo + j]; }
}
// The same for the column sums of an I/X-row matrix on a PRE-POPULATED G set (the coupler's dimE1 in XvE): a class of a range has
// the compact index gbase[r] + erank[r][cls] (the numbering a fresh set would get) and sits at tab[key] of the caller's set.
__global__ void k_sa_pack_mw_g(RgView rg, PlanView pl, FaG g, int sr0, int sr1, const double *__restrict__ Mw, int32_t *__restrict__ cidx,
                               double *__restrict__ cval) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = sr0 + (int)(i / g.NC), cls = (int)(i % g.NC);
    if (r >= sr1) return;
    const int er = g.erank[(size_t)r * g.NC + cls];
    if (er < 0) return;
    const long iA = pl.riA[r];
    const long gkey = g.key == KEY_E ? iA * rg.sA + (long)cls * rg.sHC : iA;
    const uint32_t k = g.gbase[r] + (uint32_t)er;
    const int32_t pos = g.tab[gkey];
    cidx[k] = pos; cval[k] = Mw[pos];
}
__global__ void k_sa_unpack_mw(const int32_t *__restrict__ cidx, const double *__restrict__ cval, long k0, long k1, long n, double *__restrict__ Mw) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || (k >= k0 && k < k1)) return;                   // (this rank's own sums are in place)
    Mw[cidx[k]] = cval[k];
}

// false: not served here (nothing has been touched: fast_build's own kernels run next).
// comm != nullptr: the SHARDED build of ibh_regrid_matrices_matrix_d_sharded -- every rank of the communicator calls this with the
// same arguments; rank k runs the passes over its share of the ranges (contiguous, balanced by exchange cells: the grid is sorted
// by atmosphere cell, AbbrGrid.cpp:10-21, so a block of ranges is a block of rows of an A/E-row matrix), three exchanges make
// every rank's result the whole matrix:
//   1. per rank {first out-of-range cell, fallback flags, new P keys, classes, entries, row entries} (32 bytes) -> offsets of every
//      rank's pieces; an error or a fallback is taken by all ranks together
//   2. the code bytes and the P-key prefixes of every 64 cells (the "first-seen flags": a straddler's id is looked up in the slice
//      of the rank that numbered its ice cell), the class ranks and row offsets of every range   (~1 byte per exchange cell)
//   3. the pieces of the result: CSR, wM, Mw, the dims tables
static bool stream_build(const ibh_regrid_matrices *rm, const MatSpec *sp, ibh_sparse_set *dims[2], int scale, int correctA,
                         const RgView &rg, ibh_weighted *w, hipStream_t st, ibh_comm *comm) {
    const ibh_regridder *gr = rm->rg;
    const ibh_plan &P = gr->plan;
    const long nX = gr->nX;
    const int world = comm ? comm_world(comm) : 1, rank = comm ? comm_rank(comm) : 0;
    if (sp->family == FAM_EVA) return false;
    if (!comm && !get_tuning("assemble_stream", nX >= (1l << 20) ? 1 : 0)) return false;
    const bool g_is_row = sp->family == FAM_AEVI;
    ibh_sparse_set *gset = dims[g_is_row ? 0 : 1], *pset = dims[g_is_row ? 1 : 0];
    const int gkey = g_is_row ? sp->row_key : sp->col_key, glist = g_is_row ? sp->row_list : sp->col_list;
    const int pkey = g_is_row ? sp->col_key : sp->row_key, plist = g_is_row ? sp->col_list : sp->row_list;
    auto extent_of = [&](int key) -> int64_t {
        return key == KEY_A ? gr->nA : key == KEY_E ? gr->nA * (int64_t)gr->nhc : key == KEY_I ? gr->nI : gr->nX;
    };
    const int pmode = fa_pset_mode(pset, extent_of(pkey));
    if (pmode < 0) return false;
    const bool g_fresh = gset->n == 0;
    if (!g_fresh && (gset->identity || gkey != KEY_E || g_is_row)) return false;
    const bool uses_ep = sp->row_list == LIST_EP || sp->col_list == LIST_EP;
    if (uses_ep && (P.tiny || !P.icnt_nz.p)) return false;         // classes are counted by the sign of the area here
    if (!P.icnt_pos.p) return false;
    // an I-row matrix on an identity ice set (the coupler's IvE, IceCoupler.cpp:462): its rows lie in ice-cell order, not in
    // first-seen order -- row lengths are scattered by ice cell and scanned over the ice cells
    const bool by_ice = !g_is_row && pkey == KEY_I && pmode == 0;
    // sharded: the pieces of a rank must be contiguous in every result array -- sets numbered by this build; at most 8 ranks (one
    // 256-byte read-back carries all counters).  An A/E-row matrix on an IDENTITY P set (the coupler's EvI / AvI over the identity
    // dimI, IceCoupler.cpp:366-377) is served too: its CSR pieces are contiguous as they are (an entry's column is its sparse
    // index), only Mw is scattered -- the sums travel as {position, value} pairs at their first-seen ranks (k_sa_pack_mw).
    // An X-row matrix on the identity dimX (the coupler's XvE, :467) has its rows in exchange-cell order: a block of cells is a block
    // of rows.  A pre-populated G set (its dimE1, numbered by EvI) is looked up in the table every rank has; the column sums sit
    // at the caller's dense ids and travel as pairs too, at the ranks a fresh numbering would give the classes (k_sa_pack_mw_g).
    // I rows on the identity dimI (its IvE, :462; by_ice): rows in ice-cell order -- the row lengths are merged before the row pointer
    // is scanned, the rows travel in first-seen order and are copied to their places (k_sa_pack_keys / k_sa_pack_rows).
    const bool ident_p = world > 1 && pmode == 0 && g_is_row;
    const bool ident_x_rows = world > 1 && pmode == 0 && !g_is_row && pkey == KEY_X;
    const bool ident_i_rows = world > 1 && by_ice;
    const bool prepop_g = world > 1 && !g_fresh;                 // (an E column set: checked above)
    if (world > 1 && ((pmode != 1 && !ident_p && !ident_x_rows && !ident_i_rows) || world > 8 || P.nAr < world)) return false;
    const bool pscan = pmode == 1 || ident_p || ident_i_rows;    // the P keys are numbered (first-seen ranks)
    const int merge = (sp->row_key != KEY_X && sp->col_key != KEY_X) ? 1 : 0;
    const int S = uses_ep ? 2 : 1;
    if (gr->nhc > 64) return false;
    const uint8_t *emc = elevmask_classes_impl(rm, st);          // (made when the object was created, or here by the first build)
    const bool rel32 = (int64_t)S * P.maxrange > 65535;

    Arena &A = arena();
    A.reset();
    PlanView pl{P.arng.p, P.aidx.p, P.ilptr.p, P.ilist.p, P.ifirst.p, P.isdup.p, P.mlist.p, P.nAr, P.nmulti, P.icnt_pos.p, P.icnt_nz.p, P.exI.p};
    pl.riA = P.riA.p;
    const int T = FA_T, nAr = P.nAr;
    const long nW = ceil_div(nX, 64l);
    // ---- this rank's share: ranges [sr0, sr1) chosen so that every rank gets about the same number of exchange cells
    std::vector<int> R(world + 1, 0);
    std::vector<long> X(world + 1, 0);
    R[world] = nAr; X[world] = nX;
    if (world > 1) {
        const std::vector<int32_t> &ah = plan_arng_host(gr);
        for (int k = 1; k < world; ++k) {
            const long target = nX * k / world;
            int r = (int)(std::lower_bound(ah.begin(), ah.begin() + nAr, (int32_t)target) - ah.begin());
            r = std::max(r, R[k - 1] + 1);                       // (every rank owns at least one range)
            r = std::min(r, nAr - (world - k));
            R[k] = r; X[k] = ah[r];
        }
    }
    SaBuf sb{};
    sb.sr0 = R[rank]; sb.sr1 = R[rank + 1]; sb.sx0 = X[rank]; sb.sx1 = X[rank + 1];
    sb.gx0 = sb.sx0 & ~63l; sb.gx1 = std::min(nX, (sb.sx1 + 63) & ~63l);
    const int nr = sb.sr1 - sb.sr0;
    const long gW0 = sb.gx0 >> 6, gW1 = ceil_div(sb.gx1, 64l), nwl = gW1 - gW0;
    pl.sx0 = sb.sx0; pl.sx1 = world > 1 ? sb.sx1 : 0;
    PlanView pl3 = pl;                                           // (k_fa_pelem walks pl.mlist: here the ice cells that are no clean pairs)
    pl3.mlist = P.mlist3.p; pl3.nmulti = P.nmulti3;
    FaG g{};
    g.key = gkey; g.list = glist; g.NC = gkey == KEY_E ? gr->nhc : 1;
    const size_t nrc = (size_t)nAr * g.NC;
    g.erank = A.get<int8_t>(nrc); g.ecntn = A.get<uint32_t>(nrc); g.ecnto = A.get<uint32_t>(nrc);
    g.r_ncls = A.get<uint32_t>((size_t)nAr); g.r_nent = A.get<uint32_t>((size_t)nAr);
    g.gbase = A.get<uint32_t>((size_t)nAr + 1); g.ebase = A.get<uint32_t>((size_t)nAr + 1);
    const int64_t gext = extent_of(gkey);
    if (!g_fresh) g.tab = set_inverse_table(gset, gext, st);
    FaP p{};
    p.key = pkey; p.list = plist; p.fresh = pmode;
    sb.code = A.get<uint8_t>((size_t)nW * 64);
    if (uses_ep) sb.cls = A.get<uint8_t>((size_t)nW * 64);
    if (!g_is_row) {
        sb.rl = A.get<uint8_t>((size_t)nW * 64);
        if (by_ice) {
            sb.rli = A.get<uint8_t>((size_t)gr->nI);
            IBH_HIP(hipMemsetAsync(sb.rli, 0, (size_t)gr->nI, st));
        }
        if (!by_ice || world > 1) { sb.cntL = A.get<uint32_t>((size_t)nW + 1); sb.Lw = A.get<uint32_t>((size_t)nW + 1); }
    }
    if (uses_ep) sb.rel = A.get_bytes((size_t)nX * S * (rel32 ? 4 : 2));
    else {
        sb.cntE = A.get<uint8_t>((size_t)nW + 1); sb.cntO = A.get<uint8_t>((size_t)nW + 1);
        sb.Ew = A.get<uint32_t>((size_t)nW + 1); sb.Eow = A.get<uint32_t>((size_t)nW + 1);
        sb.ewbase = A.get<uint32_t>((size_t)nAr); sb.eobase = A.get<uint32_t>((size_t)nAr);
    }
    sb.cntP = A.get<uint8_t>((size_t)nW + 1); sb.Pw = A.get<uint32_t>((size_t)nW + 1);
    sb.pbits = A.get<unsigned long long>((size_t)nW + 1);
    p.Pw = sb.Pw; p.pbits = sb.pbits;
    // counters, 8 per rank, read back with one sync: [0] first out-of-range cell, [1] fallback flags, [2] new P keys, [3] G classes,
    // [4] entries, [5] row entries (I/X-row matrices), [6] / [7] where the slice starts in the local numberings
    uint32_t *d_tot = A.get<uint32_t>((size_t)8 * world);
    uint32_t *d_cnt = d_tot + 8 * rank;
    hipLaunchKernelGGL(k_fa_init, dim3(1), dim3(64), 0, st, d_cnt);
    if (world > 1) {     // (one more wave than the slice has: a zero count behind the last one, so that the scans also yield totals)
        IBH_HIP(hipMemsetAsync(sb.cntP + gW1, 0, 1, st));
        IBH_HIP(hipMemsetAsync(sb.pbits + gW1, 0, 8, st));
        if (!g_is_row) IBH_HIP(hipMemsetAsync(sb.cntL + gW1, 0, sizeof(uint32_t), st));
    }
    if (!uses_ep) {      // (the scans of the entry counts are read AT the end of the last range too)
        IBH_HIP(hipMemsetAsync(sb.cntE + gW1, 0, 1, st));
        IBH_HIP(hipMemsetAsync(sb.cntO + gW1, 0, 1, st));
    }
    const dim3 gs(ceil_div(sb.gx1 - sb.gx0, (long)SA_TILE)), gr4(ceil_div(nr, 4));
    if (uses_ep) hipLaunchKernelGGL((k_sa_flags<true>), gs, dim3(SA_T), 0, st, rg, pl, sb, pkey == KEY_X ? 1 : 0, p.fresh, plist, g_is_row ? 0 : 1, d_cnt, emc);
    else hipLaunchKernelGGL((k_sa_flags<false>), gs, dim3(SA_T), 0, st, rg, pl, sb, pkey == KEY_X ? 1 : 0, p.fresh, plist, g_is_row ? 0 : 1, d_cnt, emc);
    const long mean = nX / std::max(nAr, 1);
    // (straddlers of a range ~ its perimeter: a range of ~10^3 cells has a few dozen, one of 10^4 a few hundred)
    const int forced_os = get_tuning("assemble_stream_oldseg", -1);
    const int oldseg = (forced_os == SA_OLDSEG_S || forced_os == SA_OLDSEG_L) ? forced_os : mean <= 2048 ? SA_OLDSEG_S : SA_OLDSEG_L;
    if (uses_ep) {
        // waves per range by the size of the ranges: one wave walks ~10^3 cells in a few round trips; longer ranges are cut
        const int forced = get_tuning("assemble_stream_wpr", -1);
        const int wpr = (forced == 1 || forced == 4 || forced == 16) ? forced : mean <= 1024 ? 1 : mean <= 4096 ? 4 : 16;
#define SA_LAUNCH_RANGES(RT)                                                                                                    \
        do {                                                                                                                \
            if (wpr == 1) hipLaunchKernelGGL((k_sa_ranges<true, RT, 1>), gr4, dim3(256), 0, st, rg, pl, sb, g, d_cnt + 1, oldseg);    \
            else if (wpr == 4) hipLaunchKernelGGL((k_sa_ranges<true, RT, 4>), dim3(nr), dim3(256), 0, st, rg, pl, sb, g, d_cnt + 1, oldseg); \
            else hipLaunchKernelGGL((k_sa_ranges<true, RT, 16>), dim3(nr), dim3(1024), 0, st, rg, pl, sb, g, d_cnt + 1, oldseg);     \
        } while (0)
        if (rel32) SA_LAUNCH_RANGES(uint32_t); else SA_LAUNCH_RANGES(uint16_t);
#undef SA_LAUNCH_RANGES
    }
    // scans over the slice (numberings local to it until the offsets of the ranks are known), two launches per set of channels.
    // (world > 1: one more wave than the slice has -- the zero count behind the last one makes the scan yield the totals in place)
    {
        const long nws = world == 1 ? nW : nwl + 1;
        const MsCh chP{pscan ? (const void *)(sb.cntP + gW0) : nullptr, sb.Pw + gW0, world == 1 ? d_cnt + 2 : nullptr, pscan ? nws : 0, 1, 0};
        const bool lch = !g_is_row && (!by_ice || world > 1);
        const MsCh chL{lch ? (const void *)(sb.cntL + gW0) : nullptr, lch ? sb.Lw + gW0 : nullptr, world == 1 ? d_cnt + 5 : nullptr, lch ? nws : 0, 0, 0};
        const MsCh chG{g.r_ncls + sb.sr0, g.gbase + sb.sr0, d_cnt + 3, nr, 0, 1};
        const MsCh chN{g.r_nent + sb.sr0, g.ebase + sb.sr0, d_cnt + 4, nr, 0, 1};
        if (uses_ep) {
            const MsCh chs[4] = {chP, chL, chG, chN};
            sa_scan_channels(chs, 4, st);
        } else {
            // one class per range: the entries of a range are differences of two more scans at its ends (no pass over its bytes)
            const MsCh chE{sb.cntE + gW0, sb.Ew + gW0, nullptr, nwl + 1, 1, 0}, chO{sb.cntO + gW0, sb.Eow + gW0, nullptr, nwl + 1, 1, 0};
            const MsCh chs[4] = {chP, chL, chE, chO};
            sa_scan_channels(chs, 4, st);
            hipLaunchKernelGGL(k_sa_rangecounts, dim3(ceil_div(nr, 256)), dim3(256), 0, st, pl, sb, g, d_cnt + 1, oldseg);
            const MsCh chs2[2] = {chG, chN};
            sa_scan_channels(chs2, 2, st);
        }
    }
    if (world > 1) hipLaunchKernelGGL(k_sa_slice_counts, dim3(1), dim3(64), 0, st, sb, pscan ? 1 : 0, g_is_row ? 0 : 1, d_cnt);
    IBH_HIP(hipGetLastError());
    if (world > 1) comm_exchange_blocks(comm, reinterpret_cast<double *>(d_tot), 4, 4, st);          // exchange 1
    uint32_t hh[64];
    readback_sync(hh, d_tot, sizeof(uint32_t) * 8 * (size_t)world, st);
    uint32_t err_x = 0xffffffffu, fl = 0;
    std::vector<uint64_t> P0(world + 1, 0), G0(world + 1, 0), E0(world + 1, 0), L0(world + 1, 0);
    for (int k = 0; k < world; ++k) {
        const uint32_t *hk = hh + 8 * k;
        err_x = std::min(err_x, hk[0]); fl |= hk[1];
        P0[k + 1] = P0[k] + hk[2]; G0[k + 1] = G0[k] + hk[3]; E0[k + 1] = E0[k] + hk[4]; L0[k + 1] = L0[k] + hk[5];
    }
    if (err_x != 0xffffffffu) {                                  // message of linterp_1d_b, IceRegridder_L0.cpp:84-85
        int32_t ij[2];
        IBH_HIP(hipMemcpy(ij, gr->ex_indices.p + 2 * (size_t)err_x, sizeof(ij), hipMemcpyDeviceToHost));
        double e = 0;
        IBH_HIP(hipMemcpy(&e, rm->elevmaskI.p + ij[1], sizeof(double), hipMemcpyDeviceToHost));
        fail(IBH_ERANGE, "Elevation %g out of bounds (%g, %g)", e < 0 ? 0.0 : e, gr->hcdefs_h.front(), gr->hcdefs_h.back());
    }
    if (fl) return false;                                        // a limit of the fast path was hit (on some rank): general pipeline
    IBH_CHECK(E0[world] < (1u << 31) && P0[world] < (1u << 31), "matrix too large for int32 indices");
    const uint32_t *hme = hh + 8 * rank;
    const uint32_t nnz = (uint32_t)E0[world];
    const int np_d = p.fresh ? (int)P0[world] : (int)extent_of(pkey);
    const int ng_d = g_fresh ? (int)G0[world] : gset->n;
    // local -> global numbering of this rank's slice
    if (world > 1) hipLaunchKernelGGL(k_sa_shift, dim3(ceil_div(std::max<long>(nwl + 1, nr + 1), 256l)), dim3(256), 0, st, pscan ? sb.Pw : nullptr, g_is_row ? nullptr : sb.Lw,
                       gW0, nwl + 1, (uint32_t)(P0[rank] - hme[6]), (uint32_t)(L0[rank] - hme[7]), g.gbase, g.ebase, sb.sr0, nr, (uint32_t)G0[rank], (uint32_t)E0[rank]);
    if (world > 1) {                                             // exchange 2: what the ranks look up in each other's slices
        // (the code bytes stay at home: every pass reads them inside its own slice only)
        std::vector<int64_t> op(world + 1), ob(world + 1), oe(world + 1), og(world + 1);
        for (int k = 0; k <= world; ++k) {
            const long wk = k == world ? nW : (X[k] >> 6);
            op[k] = wk * 4; ob[k] = wk * 8; oe[k] = (int64_t)R[k] * g.NC; og[k] = (int64_t)R[k] * 4;
        }
        void *bases[4] = {sb.Pw, sb.pbits, g.erank, g.gbase};
        const int64_t *offs[4] = {op.data(), ob.data(), oe.data(), og.data()};
        comm_gatherv(comm, 4, bases, offs, st);
        hipLaunchKernelGGL(k_sa_set_u32, dim3(1), dim3(1), 0, st, g.gbase + nAr, (uint32_t)G0[world]);
    }
    DevBuf<int64_t> ptable, gtable;
    if (p.fresh) { ptable.alloc((size_t)np_d); p.to_sparse = ptable.p; }
    if (g_fresh) { gtable.alloc((size_t)ng_d); g.to_sparse = gtable.p; }
    const int nrow = g_is_row ? ng_d : np_d, ncol = g_is_row ? np_d : ng_d;
    w->nrow = nrow; w->ncol = ncol; w->nnz = nnz;
    w->rowptr.alloc((size_t)nrow + 1); w->colind.alloc(nnz); w->val.alloc(nnz);
    w->wM.alloc((size_t)nrow); w->Mw.alloc((size_t)ncol);
    FaOut o{w->rowptr.p, w->colind.p, w->val.p, w->wM.p, w->Mw.p, sp->family, scale, correctA, 0};
    uint32_t *flags = d_cnt + 1;
    // row kernel: a 16-lane row of a wave per segment.  One-class matrices: four consecutive ranges per wave; elevation classes: a
    // wave per range and group of four classes, y = the groups a range has on average, + 1 (the others loop)
    const int forced_r4 = get_tuning("assemble_stream_rows4", -1);
    // (measured on the Antarctic sheet, same box, four segments per wave against one: the column sums of the I/X-row matrices win
    // -- IvA 0.902 / 0.939 ms, IvE 1.53 / 1.55, XvE 1.14 / 1.20; the rows of AvI are even, 0.684 / 0.686, but with 128 values of a
    // segment in registers -- more cost occupancy: 146 VGPRs at 256 -- the scaling re-reads the rest: 264 against 176 MB fetched; EvI
    // loses, 1.33 / 1.25: a range has two or three classes there, a quarter to half of the rows idle.  So: column sums only.)
    const bool rows4 = oldseg == SA_OLDSEG_S && (forced_r4 >= 0 ? forced_r4 != 0 : !g_is_row);
    const int rows_wpb = 4;
    const int rows_waves = g.NC == 1 ? ceil_div(nr, 4) : nr;
    const int rows_y = g.NC == 1 ? 1 : std::max(1, std::min((std::min(g.NC, 16) + 3) / 4, ((int)(hme[3] / (uint32_t)std::max(nr, 1)) + 2 + 3) / 4));
    const dim3 grq(ceil_div(rows_waves, rows_wpb), rows_y);
    // (long ranges: one wave per (range, class), y = the classes a range has on average, rounded up, + 1)
    const dim3 grq1(ceil_div(nr, 4), g.NC == 1 ? 1 : std::max(1, std::min(std::min(g.NC, 16), (int)(hme[3] / (uint32_t)std::max(nr, 1)) + 2)));
    // the lane-parallel row kernel (k_sa_oldsort + k_sa_rowsL), R ranges per wave.  Measured on the Antarctic sheet, same box, against
    // the kernels above: the column sums of IvE 42 + 149 us against 231 (IvE 1.55 -> 1.43 ms, IvA 0.937 -> 0.910), the rows of AvI 25 +
    // 143 against 138, of EvI 42 + 340 against 270 -- a wave with sixteen chains waits for its memory round trips with two waves per
    // SIMD (66 KB of LDS per workgroup), and R = 4 beats 16: more waves matter more than full lanes.  1 km Greenland (1 143 long
    // ranges) loses by 2 x.  So: column sums of grids with many ranges only.
    const int forced_rl = get_tuning("assemble_stream_rowsl", -1);
    const bool rowsl = forced_rl >= 0 ? forced_rl != 0 : (!g_is_row && nAr >= 16384);       // (by the whole grid's ranges: every rank of a shared build takes the same kernel)
    const int avg_cls = g.NC == 1 ? 1 : std::max(1, (int)((G0[world] + (uint64_t)std::max(nAr, 1) - 1) / (uint64_t)std::max(nAr, 1)));     // (the whole grid's, as rowsl)
    const int rows_R = std::max(1, std::min(16, get_tuning("assemble_stream_rowsl_r", std::min(4, 16 / avg_cls))));
    const dim3 grl(ceil_div(nr, 4 * rows_R));
    double *sval = nullptr;
    int32_t *sdid = nullptr;
    // cells per thread of the emit pass x workgroups (0 = one per tile).  The pass is bound by instruction issue and latency, not by
    // bandwidth (ablation on the Antarctic AvI: 129 us with every load of cell data, every term and every store removed; stores +95,
    // the first-seen lookup of the straddlers +45 after the bitmask change, +115 before): a workgroup that walks several tiles has the
    // code bytes of its next tile in flight while it works.  Measured, a1h AvI / IvA / EvI / IvE ms, same box: 2 x full grid 0.645 /
    // 0.915 / 1.216 / 1.604; 1 x full 0.704 / 0.919 / 1.228 / 1.555; 1 x 8192 0.636 / 0.878 / 1.135 / 1.470; 2 x 8192 0.642 / 0.926 /
    // 1.183 / 1.574; 4 x full 0.709 / 1.063 / 1.274 / 1.711
    const int ecpt = get_tuning("assemble_stream_emit_cpt", 1);
    const long ncell = sb.gx1 - sb.gx0;
    const long eblocks = get_tuning("assemble_stream_emit_blocks", 8192);
    auto egrid = [&](int cpt) { const long full = ceil_div(ncell, (long)SA_T * cpt); return dim3((unsigned)(eblocks > 0 ? std::min(full, eblocks) : full)); };
#define SA_LAUNCH_EMIT(EP, GR)                                                                                                  \
    do {                                                                                                                        \
        if (rel32) hipLaunchKernelGGL((k_sa_emit<EP, GR, uint32_t, 4>), egrid(4), dim3(SA_T), 0, st, rg, pl, *sp, g, p, sb, merge, o, sval, sdid); \
        else if (ecpt == 2) hipLaunchKernelGGL((k_sa_emit<EP, GR, uint16_t, 2>), egrid(2), dim3(SA_T), 0, st, rg, pl, *sp, g, p, sb, merge, o, sval, sdid); \
        else if (ecpt == 1) hipLaunchKernelGGL((k_sa_emit<EP, GR, uint16_t, 1>), egrid(1), dim3(SA_T), 0, st, rg, pl, *sp, g, p, sb, merge, o, sval, sdid); \
        else hipLaunchKernelGGL((k_sa_emit<EP, GR, uint16_t, 4>), egrid(4), dim3(SA_T), 0, st, rg, pl, *sp, g, p, sb, merge, o, sval, sdid);       \
    } while (0)
    sb.end_row = nrow; sb.end_nnz = nnz;
    int32_t *kidx = nullptr;                                     // shared build, identity ice rows: the ice cell / row length of every P key, in first-seen order
    uint8_t *klen = nullptr;
    const long nkeys = ident_i_rows ? (long)P0[world] : 0;
    if (ident_i_rows) {                                          // exchange 2b: every rank learns every ice cell's row length
        kidx = A.get<int32_t>((size_t)std::max<long>(nkeys, 1));
        klen = A.get<uint8_t>((size_t)std::max<long>(nkeys, 1));
        if (sb.sx1 > sb.sx0) hipLaunchKernelGGL(k_sa_pack_keys, dim3((unsigned)ceil_div(sb.sx1 - sb.sx0, 256l)), dim3(256), 0, st, sb, pl.exI, kidx, klen);
        std::vector<int64_t> o4p(world + 1), o1p(world + 1);
        for (int k = 0; k <= world; ++k) { o4p[k] = (int64_t)P0[k] * 4; o1p[k] = (int64_t)P0[k]; }
        void *bases[2] = {kidx, klen};
        const int64_t *offs[2] = {o4p.data(), o1p.data()};
        comm_gatherv(comm, 2, bases, offs, st);
        if (nkeys) hipLaunchKernelGGL(k_sa_unpack_rli, dim3((unsigned)ceil_div(nkeys, 256l)), dim3(256), 0, st, kidx, klen, (long)P0[rank], (long)P0[rank + 1], nkeys, sb.rli);
    }
    if (by_ice) {                                                // the row pointer over the ice cells (closes itself: rowptr[nI] = nnz); wM of non-members
        const MsCh chI{sb.rli, reinterpret_cast<uint32_t *>(w->rowptr.p), nullptr, (long)gr->nI, 1, 1};
        sa_scan_channels(&chI, 1, st);
        IBH_HIP(hipMemsetAsync(w->wM.p, 0, sizeof(double) * (size_t)nrow, st));
    }
    if (g_is_row) {
        if (ncol && !p.fresh) IBH_HIP(hipMemsetAsync(w->Mw.p, 0, sizeof(double) * (size_t)ncol, st));
        if (uses_ep) SA_LAUNCH_EMIT(true, true); else SA_LAUNCH_EMIT(false, true);
        if (p.key == KEY_I && P.npair) {                      // Mw of the ice cells across one GCM-cell edge
            const dim3 gp2(ceil_div(P.npair, 256));
            if (uses_ep) hipLaunchKernelGGL((k_sa_pairs<true, true>), gp2, dim3(256), 0, st, rg, pl, *sp, g, p, sb, P.px1.p, P.px2.p, P.piI.p, P.pa1.p, P.pa2.p, P.npair, o);
            else hipLaunchKernelGGL((k_sa_pairs<false, true>), gp2, dim3(256), 0, st, rg, pl, *sp, g, p, sb, P.px1.p, P.px2.p, P.piI.p, P.pa1.p, P.pa2.p, P.npair, o);
        }
        if (p.key == KEY_I && P.nmulti3) {                    // ... and of the rest of the ice cells with several exchange cells
            const dim3 gm(ceil_div(P.nmulti3, T));
            if (uses_ep) hipLaunchKernelGGL((k_fa_pelem<true, FA_PSUMS, false>), gm, dim3(T), 0, st, rg, pl3, *sp, g, p, merge, (long)P.nmulti3, o, (uint32_t *)nullptr, flags, 1);
            else hipLaunchKernelGGL((k_fa_pelem<false, FA_PSUMS, false>), gm, dim3(T), 0, st, rg, pl3, *sp, g, p, merge, (long)P.nmulti3, o, (uint32_t *)nullptr, flags, 1);
        }
        if (rowsl) {
            if (oldseg == SA_OLDSEG_S) hipLaunchKernelGGL((k_sa_oldsort<SA_OLDSEG_S>), gr4, dim3(256), 0, st, g, o.colind, o.val, sb.sr0, sb.sr1);
            else hipLaunchKernelGGL((k_sa_oldsort<SA_OLDSEG_L>), gr4, dim3(256), 0, st, g, o.colind, o.val, sb.sr0, sb.sr1);
            hipLaunchKernelGGL((k_sa_rowsL<true>), grl, dim3(256), 0, st, rg, pl, g, o, sb.sr0, sb.sr1, rows_R);
        } else if (rows4) hipLaunchKernelGGL((k_sa_rows<true, SA_OLDSEG_S, 4>), grq, dim3(256), 0, st, rg, pl, g, o, (int32_t *)nullptr, sb.sr0, sb.sr1);
        else if (oldseg == SA_OLDSEG_S) hipLaunchKernelGGL((k_sa_rows1<true, SA_OLDSEG_S>), grq1, dim3(256), 0, st, rg, pl, g, o, (int32_t *)nullptr, sb.sr0, sb.sr1);
        else hipLaunchKernelGGL((k_sa_rows1<true, SA_OLDSEG_L>), grq1, dim3(256), 0, st, rg, pl, g, o, (int32_t *)nullptr, sb.sr0, sb.sr1);
    } else {
        sval = A.get<double>(nnz); sdid = A.get<int32_t>(nnz);
        if (uses_ep) SA_LAUNCH_EMIT(true, false); else SA_LAUNCH_EMIT(false, false);
        if (p.key == KEY_I && P.npair) {                      // the rows of the ice cells across one GCM-cell edge
            const dim3 gp2(ceil_div(P.npair, 256));
            if (uses_ep) hipLaunchKernelGGL((k_sa_pairs<true, false>), gp2, dim3(256), 0, st, rg, pl, *sp, g, p, sb, P.px1.p, P.px2.p, P.piI.p, P.pa1.p, P.pa2.p, P.npair, o);
            else hipLaunchKernelGGL((k_sa_pairs<false, false>), gp2, dim3(256), 0, st, rg, pl, *sp, g, p, sb, P.px1.p, P.px2.p, P.piI.p, P.pa1.p, P.pa2.p, P.npair, o);
        }
        if (p.key == KEY_I && P.nmulti3) {                    // ... and of the rest of the ice cells with several exchange cells
            const dim3 gm(ceil_div(P.nmulti3, T));
            if (!g_fresh) {
                if (uses_ep) hipLaunchKernelGGL((k_fa_pelem<true, FA_PEMIT, true>), gm, dim3(T), 0, st, rg, pl3, *sp, g, p, merge, (long)P.nmulti3, o, (uint32_t *)nullptr, flags, 1);
                else hipLaunchKernelGGL((k_fa_pelem<false, FA_PEMIT, true>), gm, dim3(T), 0, st, rg, pl3, *sp, g, p, merge, (long)P.nmulti3, o, (uint32_t *)nullptr, flags, 1);
            } else {
                if (uses_ep) hipLaunchKernelGGL((k_fa_pelem<true, FA_PEMIT, false>), gm, dim3(T), 0, st, rg, pl3, *sp, g, p, merge, (long)P.nmulti3, o, (uint32_t *)nullptr, flags, 1);
                else hipLaunchKernelGGL((k_fa_pelem<false, FA_PEMIT, false>), gm, dim3(T), 0, st, rg, pl3, *sp, g, p, merge, (long)P.nmulti3, o, (uint32_t *)nullptr, flags, 1);
            }
        }
        if (!g_fresh && ncol) hipLaunchKernelGGL(k_fa_zero_identity, dim3(ceil_div(ncol, T)), dim3(T), 0, st, w->Mw.p, (long)ncol);
        FaOut os = o;
        os.val = sval;
        if (rowsl) {
            if (oldseg == SA_OLDSEG_S) hipLaunchKernelGGL((k_sa_oldsort<SA_OLDSEG_S>), gr4, dim3(256), 0, st, g, sdid, sval, sb.sr0, sb.sr1);
            else hipLaunchKernelGGL((k_sa_oldsort<SA_OLDSEG_L>), gr4, dim3(256), 0, st, g, sdid, sval, sb.sr0, sb.sr1);
            hipLaunchKernelGGL((k_sa_rowsL<false>), grl, dim3(256), 0, st, rg, pl, g, os, sb.sr0, sb.sr1, rows_R);
        } else if (rows4) hipLaunchKernelGGL((k_sa_rows<false, SA_OLDSEG_S, 4>), grq, dim3(256), 0, st, rg, pl, g, os, sdid, sb.sr0, sb.sr1);
        else if (oldseg == SA_OLDSEG_S) hipLaunchKernelGGL((k_sa_rows1<false, SA_OLDSEG_S>), grq1, dim3(256), 0, st, rg, pl, g, os, sdid, sb.sr0, sb.sr1);
        else hipLaunchKernelGGL((k_sa_rows1<false, SA_OLDSEG_L>), grq1, dim3(256), 0, st, rg, pl, g, os, sdid, sb.sr0, sb.sr1);
    }
#undef SA_LAUNCH_EMIT
    IBH_HIP(hipGetLastError());
    if (world > 1) {                                             // exchange 3: the pieces of the result
        // A/E-row matrix: CSR by entries of the ranges, rows by classes; I/X-row matrix: CSR by row entries, rows by P keys (an
        // identity dimX: by exchange cells).  The column-side sums are one range per rank when the column set was numbered by this
        // build; on an identity / pre-populated set they travel as {position, value} pairs and are scattered on every rank.
        std::vector<uint64_t> Xc(world + 1);
        for (int k = 0; k <= world; ++k) Xc[k] = (uint64_t)X[k];
        const std::vector<uint64_t> &En = g_is_row ? E0 : L0, &Rw = g_is_row ? G0 : (ident_x_rows ? Xc : P0), &Cl = g_is_row ? P0 : G0;
        std::vector<int64_t> o4e(world + 1), o8e(world + 1), o4r(world + 1), o8r(world + 1), o4c(world + 1), o8c(world + 1);
        for (int k = 0; k <= world; ++k) {
            o4e[k] = (int64_t)En[k] * 4; o8e[k] = (int64_t)En[k] * 8; o4r[k] = (int64_t)Rw[k] * 4; o8r[k] = (int64_t)Rw[k] * 8;
            o4c[k] = (int64_t)Cl[k] * 4; o8c[k] = (int64_t)Cl[k] * 8;
        }
        const bool pairs = g_is_row ? ident_p : prepop_g;        // the column-side sums as pairs
        const long npk = (long)Cl[world];
        int32_t *cidx = nullptr;
        double *cval = nullptr;
        if (pairs) {
            cidx = A.get<int32_t>((size_t)std::max<long>(npk, 1));
            cval = A.get<double>((size_t)std::max<long>(npk, 1));
            if (g_is_row) {
                if (sb.sx1 > sb.sx0)
                    hipLaunchKernelGGL(k_sa_pack_mw, dim3((unsigned)ceil_div(sb.sx1 - sb.sx0, 256l)), dim3(256), 0, st, sb, pl.exI, pkey == KEY_X ? 1 : 0, w->Mw.p, cidx, cval);
            } else if (nr > 0)
                hipLaunchKernelGGL(k_sa_pack_mw_g, dim3((unsigned)ceil_div((long)nr * g.NC, 256l)), dim3(256), 0, st, rg, pl, g, sb.sr0, sb.sr1, w->Mw.p, cidx, cval);
        }
        std::vector<void *> bases = {w->colind.p, w->val.p, w->rowptr.p, w->wM.p};
        std::vector<const int64_t *> offs = {o4e.data(), o8e.data(), o4r.data(), o8r.data()};
        int32_t *ccol = nullptr;
        double *cvl = nullptr, *kwM = nullptr;
        uint32_t *klo = nullptr;
        std::vector<int64_t> o4p(world + 1), o8p(world + 1);
        if (ident_i_rows) {
            // the rows in first-seen order (every rank scanned the same row pointer: it does not travel), wM and the place of every
            // key's row in that order at the key's rank
            const long nl = (long)L0[world];
            ccol = A.get<int32_t>((size_t)std::max<long>(nl, 1)); cvl = A.get<double>((size_t)std::max<long>(nl, 1));
            kwM = A.get<double>((size_t)std::max<long>(nkeys, 1)); klo = A.get<uint32_t>((size_t)std::max<long>(nkeys, 1));
            if (sb.sx1 > sb.sx0)
                hipLaunchKernelGGL(k_sa_pack_rows, dim3((unsigned)ceil_div(sb.sx1 - sb.sx0, 256l)), dim3(256), 0, st, sb, pl.exI, w->rowptr.p, w->colind.p, w->val.p,
                                   w->wM.p, ccol, cvl, kwM, klo);
            for (int k = 0; k <= world; ++k) { o4p[k] = (int64_t)P0[k] * 4; o8p[k] = (int64_t)P0[k] * 8; }
            bases = {ccol, cvl, kwM, klo};
            offs = {o4e.data(), o8e.data(), o8p.data(), o4p.data()};
        }
        if (pairs) { bases.push_back(cidx); offs.push_back(o4c.data()); bases.push_back(cval); offs.push_back(o8c.data()); }
        else { bases.push_back(w->Mw.p); offs.push_back(o8c.data()); }
        DevBuf<int64_t> &rtable = g_is_row ? gtable : ptable, &ctable = g_is_row ? ptable : gtable;
        if (rtable.p) { bases.push_back(rtable.p); offs.push_back(o8r.data()); }
        if (ctable.p) { bases.push_back(ctable.p); offs.push_back(o8c.data()); }
        comm_gatherv(comm, (int)bases.size(), bases.data(), offs.data(), st);
        if (ident_i_rows && nkeys)
            hipLaunchKernelGGL(k_sa_unpack_rows, dim3((unsigned)ceil_div(nkeys, 256l)), dim3(256), 0, st, kidx, klen, klo, kwM, ccol, cvl, (long)P0[rank],
                               (long)P0[rank + 1], nkeys, w->rowptr.p, w->colind.p, w->val.p, w->wM.p);
        if (pairs && npk) hipLaunchKernelGGL(k_sa_unpack_mw, dim3((unsigned)ceil_div(npk, 256l)), dim3(256), 0, st, cidx, cval, (long)Cl[rank], (long)Cl[rank + 1], npk, w->Mw.p);
    }
    auto commit = [&](ibh_sparse_set *set, int64_t extent, DevBuf<int64_t> &table, int n) {
        set->sparse_extent = extent;
        if (n == 0) return;
        set->host.clear(); set->host_n = 0; set->inv.clear(); set->inv_n = 0;
        set->dev = std::move(table);
        set->dev_n = set->n = n;
        set->identity = false;
    };
    if (p.fresh) commit(pset, extent_of(pkey), ptable, np_d); else if (pset->sparse_extent != extent_of(pkey)) pset->sparse_extent = extent_of(pkey);
    if (g_fresh) commit(gset, gext, gtable, ng_d); else if (gset->sparse_extent != gext) gset->sparse_extent = gext;
    IBH_HIP(hipStreamSynchronize(st));
    w->built_fast = world > 1 ? 3 : 2;
    // (flags raised by the later kernels -- a straddler list that overflowed -- cannot differ from the count pass's; checked there)
    return true;
}
