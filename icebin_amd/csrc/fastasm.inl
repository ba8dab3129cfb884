// fastasm.inl -- the assembly fast path for SORTED exchange grids (textually included at the end of
// assemble.hip, inside namespace ibh: it reuses the generators load_cell / contributions of the
// general pipeline unchanged).
//
// ExchangeGrid's constructor sorts the cells by (iA, iI) (AbbrGrid.cpp:10-21), and what a coupler
// rebuilds every step (IceCoupler.cpp:361-468) depends on the elevation mask only -- the exchange grid
// is static.  So everything that is mask-independent is computed ONCE per ice sheet (ibh_plan):
//   * the cells of one atmosphere cell form a contiguous RANGE of exchange cells;
//   * every ice cell has a (static) list of its <= 8 exchange cells, ascending;
//   * the position at which an ice cell is first seen (spsparse's first-seen numbering) is static, only
//     WHETHER it is seen depends on the mask; atmosphere cells are first seen in range order, elevation
//     classes in first-seen order inside their range (a 64-entry LDS table per range).
// A build is then a handful of streaming kernels instead of three generator passes + scan + ordering +
// segment sums + column slots (~45 launches):
//   numbering  first-occurrence flags (written by k_fa_count) -> scan       (ice / exchange keys; skipped for identity dims)
//              k_fa_count (one workgroup per range; one-class matrices of large grids: k_fa_count_stream, a thread per
//              cell) -> k_fa_rscan / fa_rscan_many                          (A / E keys, entry counts)
//   A/E rows   k_fa_range<EMIT>: every entry is written straight to its CSR slot -- inside a row the
//              columns first seen in this range ascend in x order (rank = a per-class running count),
//              only the ice cells that straddle in from an earlier range (a few %) are ranked in LDS --
//              then sequential row sums, weights and scaling in the same workgroup;
//              k_fa_pelem<SUMS>: Mw per ice cell from its static list (entries already in row order)
//   I/X rows   k_fa_pelem<COUNT> -> scan -> k_fa_pelem<EMIT> (one thread per row: <= 8 entries),
//              k_fa_range<SUMS>: column sums through the same placement, summed sequentially.
// Every sum runs in the order the general pipeline (and the oracle) uses -- duplicates in emission
// order with the first term assigned, row sums by ascending column, column sums by ascending row -- so
// the results are bit-identical; the general pipeline remains the fallback for unsorted grids,
// partially pre-populated dims, EvA / AvE on pre-populated sets, and anything a limit below excludes.
// EvA / AvE on fresh sets: fast_build_eva (count -> range sums -> one thread per range).
// Small grids (<= 2^20 exchange cells, round 5): the scans after k_fa_count ride IN it (FaChain: decoupled look-back over the
// ranges), the counters come from a ring of initialised slots and go back to the host from the build's last kernel
// (fa_publish); independent work rides in extra workgroups of launches that exist anyway (the straddlers' column sums in
// k_fa_range's, the static row lengths of the identity ice set in k_fa_count's) -- a G-row build is k_fa_count -> k_fa_range.

constexpr int FA_NC = 64;           // elevation classes per range (nhc <= 64)
constexpr int FA_ILMAX = 8;         // exchange cells per ice cell
constexpr int FA_DUPMAX = 4;        // consecutive cells with the same (iA, iI)
constexpr int FA_OLDMAX = 512;      // straddling entries of one range ranked in LDS
constexpr int FA_T = 256;          // threads per workgroup of the per-element kernels (the per-range kernels: fa_range_shape)
enum { FA_ERR_OLDOVER = 1, FA_ERR_MISSING = 2 };

struct PlanView {
    const int32_t *arng, *aidx, *ilptr, *ilist, *ifirst;
    const uint8_t *isdup;       // bit 0 duplicate of the previous cell, bit 1 the only exchange cell of its ice cell, bit 2 first-seen
                                // cell of its ice cell, bit 3 area > 0, bit 4 area != 0 and not > 0, bit 5 first cell of its range,
                                // bit 6 its ice cell was first seen in an earlier range, bit 7 the next cell is a duplicate of this one
    const int32_t *mlist;       // ice cells with more than one exchange cell
    int nAr, nmulti;
    const uint8_t *icnt_pos, *icnt_nz;      // static entry counts per ice cell (nullptr: count by visiting), see ibh_plan
    const int32_t *exI;                     // ice-cell index per exchange cell (compact copy of ex_indices[:, 1])
    const int32_t *riA;                     // atmosphere cell of every range (streamasm.inl)
    long sx0, sx1;                          // sharded streamed build: only the ice cells first seen in [sx0, sx1) (sx1 == 0: all)
};

// ---- static plan ---------------------------------------------------------------------------------
__global__ void k_plan_flags(const int32_t *__restrict__ exi, const double *__restrict__ area, long nX, uint32_t *__restrict__ head,
                             uint8_t *__restrict__ isdup, uint32_t *__restrict__ bad, int32_t *__restrict__ exI) {
    const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nX) return;
    const int iA = exi[2 * x], iI = exi[2 * x + 1];
    exI[x] = iI;
    bool h = true, d = false;
    if (x > 0) {
        const int pA = exi[2 * x - 2], pI = exi[2 * x - 1];
        h = pA != iA;
        d = !h && pI == iI;
        if (iA < pA || (iA == pA && iI < pI)) *bad = 1u;           // not sorted by (iA, iI)
    }
    head[x] = h ? 1u : 0u;
    // bits 3 / 4: the sign class of the (static) overlap area -- all a build without elevation classes needs of it to COUNT
    // (GvAp wants area > 0, GvI area != 0: IceRegridder_L0.cpp:186-187,208-209), so k_fa_count<false> does not load the areas
    const double a = area[x];
    isdup[x] = (uint8_t)((d ? 1 : 0) | (a > 0 ? 8 : (a != 0 ? 16 : 0)));
}
__global__ void k_plan_ranges(const uint32_t *__restrict__ head, const uint32_t *__restrict__ hpos, const uint8_t *__restrict__ isdup,
                              long nX, int32_t *__restrict__ aidx, int32_t *__restrict__ arng, uint32_t nAr, uint32_t *__restrict__ bad) {
    const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nX) return;
    const int r = (int)(hpos[x] + head[x]) - 1;
    aidx[x] = r;
    if (head[x]) arng[r] = (int32_t)x;
    if (x == 0) arng[nAr] = (int32_t)nX;
    if (x >= FA_DUPMAX - 1) {                                    // FA_DUPMAX-1 duplicates in a row -> a group of FA_DUPMAX is fine, more is not
        bool all = true;
        for (int j = 0; j < FA_DUPMAX; ++j) all = all && x - j >= 0 && (isdup[x - j] & 1);
        if (all) *bad = 1u;
    }
}
__global__ void k_plan_ikeys(const int32_t *__restrict__ exi, long nX, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < nX) { keys[x] = (uint64_t)(uint32_t)exi[2 * x + 1]; vals[x] = (uint32_t)x; }
}
__global__ void k_plan_ifirst(const int32_t *__restrict__ ilptr, const int32_t *__restrict__ ilist, const double *__restrict__ area,
                              const int32_t *__restrict__ aidx, long nI, int32_t *__restrict__ ifirst, uint8_t *__restrict__ isdup,
                              uint32_t *__restrict__ multi, uint32_t *__restrict__ bad, uint8_t *__restrict__ icnt_pos,
                              uint8_t *__restrict__ icnt_nz, uint32_t *__restrict__ tiny) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nI) return;
    const int b = ilptr[i], e = ilptr[i + 1];
    if (e - b > FA_ILMAX) *bad = 1u;
    int f = -1;
    int npos = 0, nnz = 0;
    bool gpos = false, gnz = false;             // the current group of duplicates has a member with area > 0 / != 0
    for (int k = b; k < e; ++k) {
        const double a = area[ilist[k]];
        if (f < 0 && a != 0) f = ilist[k];
        if (!(isdup[ilist[k]] & 1)) { npos += gpos ? 1 : 0; nnz += gnz ? 1 : 0; gpos = gnz = false; }     // a new group starts
        gpos = gpos || a > 0;
        gnz = gnz || a != 0;
        if (a != 0 && fabs(a) < 1e-290) *tiny = 1u;           // area * class weight could underflow to zero: no static count
    }
    npos += gpos ? 1 : 0; nnz += gnz ? 1 : 0;
    icnt_pos[i] = (uint8_t)npos; icnt_nz[i] = (uint8_t)nnz;
    ifirst[i] = f;
    if (f >= 0) isdup[f] |= 4;                                     // "this exchange cell is the first-seen one of its ice cell": spares the builds the ifirst gather
    (void)aidx;
    if (e - b == 1) isdup[ilist[b]] |= 2;                          // the only exchange cell of its ice cell (own cell only: no two threads share a byte)
    multi[i] = (e - b > 1) ? 1u : 0u;
}
// bits 5..7 and the atmosphere cell of every range (the streamed build, streamasm.inl); after k_plan_ifirst
__global__ void k_plan_bits2(const int32_t *__restrict__ exi, const int32_t *__restrict__ aidx, const int32_t *__restrict__ arng,
                             const int32_t *__restrict__ ifirst, long nX, uint8_t *__restrict__ isdup, int32_t *__restrict__ riA,
                             uint32_t *__restrict__ maxrange) {
    const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nX) return;
    const int r = aidx[x];
    const int x0 = arng[r];
    const int f = ifirst[exi[2 * x + 1]];
    uint8_t b = isdup[x];
    if (x == x0) { b |= 32; riA[r] = exi[2 * x]; atomicMax(maxrange, (uint32_t)(arng[r + 1] - x0)); }
    if (f >= 0 && f < x0) b |= 64;
    if (x + 1 < nX && (isdup[x + 1] & 1)) b |= 128;              // (bit 0 of the neighbour is final since k_plan_flags)
    isdup[x] = b;
}
// ice cells with several exchange cells: a clean pair (two cells, two ranges, the first one first-seen) or not
__global__ void k_plan_pairflags(const int32_t *__restrict__ ilptr, const int32_t *__restrict__ ilist, const int32_t *__restrict__ aidx,
                                 const int32_t *__restrict__ ifirst, long nI, uint32_t *__restrict__ pair, uint32_t *__restrict__ m3) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nI) return;
    const int b = ilptr[i], e = ilptr[i + 1];
    bool pr = false;
    if (e - b == 2) { const int x1 = ilist[b], x2 = ilist[b + 1]; pr = aidx[x1] != aidx[x2] && ifirst[i] == x1; }
    pair[i] = pr ? 1u : 0u;
    m3[i] = (e - b > 1 && !pr) ? 1u : 0u;
}
__global__ void k_plan_pairs(const uint32_t *__restrict__ pair, const uint32_t *__restrict__ ppos, const int32_t *__restrict__ ilptr,
                             const int32_t *__restrict__ ilist, const double *__restrict__ area, long nI, int32_t *__restrict__ px1,
                             int32_t *__restrict__ px2, int32_t *__restrict__ piI, double *__restrict__ pa1, double *__restrict__ pa2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nI && pair[i]) {
        const uint32_t k = ppos[i];
        const int b = ilptr[i], x1 = ilist[b], x2 = ilist[b + 1];
        px1[k] = x1; px2[k] = x2; piI[k] = (int32_t)i; pa1[k] = area[x1]; pa2[k] = area[x2];
    }
}
__global__ void k_plan_mlist(const uint32_t *__restrict__ multi, const uint32_t *__restrict__ mpos, long nI, int32_t *__restrict__ mlist) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nI && multi[i]) mlist[mpos[i]] = (int32_t)i;
}
__global__ void k_u32_to_i32(const uint32_t *__restrict__ in, long n, int32_t *__restrict__ out) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n) out[u] = (int32_t)in[u];
}

static bool ensure_plan(const ibh_regridder *g, hipStream_t st) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    ibh_plan &P = g->plan;
    if (P.tried) return P.ok;
    P.tried = true;
    if (g->nX == 0 || g->nhc > FA_NC) return false;
    Arena &A = arena();
    A.reset();
    const long nX = g->nX, nI = g->nI;
    const int T = 256;
    uint32_t *head = A.get<uint32_t>((size_t)nX), *hpos = A.get<uint32_t>((size_t)nX), *d_cnt = A.get<uint32_t>(2);
    IBH_HIP(hipMemsetAsync(d_cnt, 0, 2 * sizeof(uint32_t), st));
    P.isdup.alloc((size_t)nX);
    P.exI.alloc((size_t)nX);
    hipLaunchKernelGGL(k_plan_flags, dim3(ceil_div(nX, T)), dim3(T), 0, st, g->ex_indices.p, g->ex_area.p, nX, head, P.isdup.p, d_cnt + 1, P.exI.p);
    exclusive_scan_u32(head, hpos, (size_t)nX, d_cnt, st);
    uint32_t h[2];
    readback_sync(h, d_cnt, sizeof(h), st);
    if (h[1]) return false;                                      // unsorted grid: the general pipeline serves it
    P.nAr = (int32_t)h[0];
    P.aidx.alloc((size_t)nX); P.arng.alloc((size_t)P.nAr + 1);
    hipLaunchKernelGGL(k_plan_ranges, dim3(ceil_div(nX, T)), dim3(T), 0, st, head, hpos, P.isdup.p, nX, P.aidx.p, P.arng.p,
                       (uint32_t)P.nAr, d_cnt + 1);
    // exchange cells of every ice cell, ascending: stable sort of x by iI
    uint64_t *k = A.get<uint64_t>((size_t)nX), *k2 = A.get<uint64_t>((size_t)nX);
    uint32_t *v = A.get<uint32_t>((size_t)nX), *v2 = A.get<uint32_t>((size_t)nX);
    hipLaunchKernelGGL(k_plan_ikeys, dim3(ceil_div(nX, T)), dim3(T), 0, st, g->ex_indices.p, nX, k, v);
    KeyField f{0, bits_for((uint64_t)nI)};
    if (f.nbits > 0 && radix_sort_pairs(k, k2, v, v2, (size_t)nX, &f, 1, st)) { std::swap(k, k2); std::swap(v, v2); }
    int32_t *srow = A.get<int32_t>((size_t)nX);
    hipLaunchKernelGGL(k_keys_to_i32, dim3(ceil_div(nX, T)), dim3(T), 0, st, k, nX, srow);
    P.ilptr.alloc((size_t)nI + 1); P.ilist.alloc((size_t)nX); P.ifirst.alloc((size_t)nI);
    rowptr_from_rows(srow, nX, (int)nI, P.ilptr.p, st);
    hipLaunchKernelGGL(k_u32_to_i32, dim3(ceil_div(nX, T)), dim3(T), 0, st, v, nX, P.ilist.p);
    uint32_t *multi = A.get<uint32_t>((size_t)nI), *mpos = A.get<uint32_t>((size_t)nI);
    uint32_t *d_tiny = A.get<uint32_t>(1);
    IBH_HIP(hipMemsetAsync(d_tiny, 0, sizeof(uint32_t), st));
    P.icnt_pos.alloc((size_t)nI); P.icnt_nz.alloc((size_t)nI);
    hipLaunchKernelGGL(k_plan_ifirst, dim3(ceil_div(nI, T)), dim3(T), 0, st, P.ilptr.p, P.ilist.p, g->ex_area.p, P.aidx.p, nI, P.ifirst.p,
                       P.isdup.p, multi, d_cnt + 1, P.icnt_pos.p, P.icnt_nz.p, d_tiny);
    exclusive_scan_u32(multi, mpos, (size_t)nI, d_cnt, st);
    IBH_HIP(hipGetLastError());
    readback_sync(h, d_cnt, sizeof(h), st);
    if (h[1]) return false;
    uint32_t h_tiny = 0;
    IBH_HIP(hipMemcpy(&h_tiny, d_tiny, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (h_tiny) { P.icnt_pos.release(); P.icnt_nz.release(); }
    P.tiny = h_tiny != 0;
    P.nmulti = (int32_t)h[0];
    P.riA.alloc((size_t)P.nAr);
    IBH_HIP(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_plan_bits2, dim3(ceil_div(nX, T)), dim3(T), 0, st, g->ex_indices.p, P.aidx.p, P.arng.p, P.ifirst.p, nX, P.isdup.p,
                       P.riA.p, d_cnt);
    {
        uint32_t mr = 0;
        readback_sync(&mr, d_cnt, sizeof(mr), st);
        P.maxrange = (int32_t)mr;
    }
    P.mlist.alloc((size_t)P.nmulti);
    if (P.nmulti) hipLaunchKernelGGL(k_plan_mlist, dim3(ceil_div(nI, T)), dim3(T), 0, st, multi, mpos, nI, P.mlist.p);
    {   // the same cells split into clean pairs and the rest (streamasm.inl)
        uint32_t *pairf = A.get<uint32_t>((size_t)nI), *ppos = A.get<uint32_t>((size_t)nI), *m3 = A.get<uint32_t>((size_t)nI), *m3pos = A.get<uint32_t>((size_t)nI);
        hipLaunchKernelGGL(k_plan_pairflags, dim3(ceil_div(nI, T)), dim3(T), 0, st, P.ilptr.p, P.ilist.p, P.aidx.p, P.ifirst.p, nI, pairf, m3);
        exclusive_scan_u32(pairf, ppos, (size_t)nI, d_cnt, st);
        exclusive_scan_u32(m3, m3pos, (size_t)nI, d_cnt + 1, st);
        readback_sync(h, d_cnt, sizeof(h), st);
        P.npair = (int32_t)h[0]; P.nmulti3 = (int32_t)h[1];
        P.px1.alloc((size_t)P.npair); P.px2.alloc((size_t)P.npair); P.piI.alloc((size_t)P.npair); P.mlist3.alloc((size_t)P.nmulti3);
        P.pa1.alloc((size_t)P.npair); P.pa2.alloc((size_t)P.npair);
        if (P.npair) hipLaunchKernelGGL(k_plan_pairs, dim3(ceil_div(nI, T)), dim3(T), 0, st, pairf, ppos, P.ilptr.p, P.ilist.p, g->ex_area.p, nI, P.px1.p, P.px2.p,
                                        P.piI.p, P.pa1.p, P.pa2.p);
        if (P.nmulti3) hipLaunchKernelGGL(k_plan_mlist, dim3(ceil_div(nI, T)), dim3(T), 0, st, m3, m3pos, nI, P.mlist3.p);
    }
    IBH_HIP(hipGetLastError());
    IBH_HIP(hipStreamSynchronize(st));
    P.ok = true;
    return true;
}

// ---- per-build pieces ----------------------------------------------------------------------------
// G side: the dims side keyed by atmosphere cells (A) or elevation classes (E) -- numbered per range.
// P side: the side keyed by ice cells (I) or exchange cells (X) -- numbered along x.
struct FaG {
    int key, list;              // KEY_A / KEY_E, LIST_AP / LIST_EP
    int NC;                     // classes per range: 1 (A) or nhc (E)
    const int32_t *tab;         // pre-populated set: sparse -> dense (-1 missing); nullptr: numbered by this build
    int8_t *erank;              // [nAr*NC] first-seen rank of the class inside its range, -1 absent
    uint32_t *ecntn, *ecnto;    // [nAr*NC] entries of the class in the range whose P key is new / old there
    uint32_t *r_ncls, *r_nent;  // [nAr]
    uint32_t *gbase, *ebase;    // [nAr+1] exclusive scans of r_ncls / r_nent
    int64_t *to_sparse;         // numbered by this build: dense -> sparse
    int32_t *tab_out;           // ... and, small grids, sparse -> dense for the builds of the step that will find the set populated
};
struct FaP {
    int key, list;              // KEY_I / KEY_X, LIST_I / LIST_AP / LIST_EP
    int fresh;                  // 1: numbered by this build, 0: identity over the whole extent
    uint8_t *pflag;             // [nX] first-occurrence flags (bytes: written, scanned and read once per build each)
    uint32_t *poff;             // [nX] their exclusive scan: the dense id of a new key is
                                //      poff[its first-seen position] -- for an ice cell poff[ifirst[iI]], no table needed
    int64_t *to_sparse;
    // the streamed build (streamasm.inl) keeps no per-cell table of dense ids: Pw = new keys before every 64 cells, pbits = one bit per
    // cell, "a key is first seen here" (the ballots of k_sa_flags) -- the id at cell f is two independent 4 / 8-byte reads (sa_prank_at)
    const uint32_t *Pw;
    const unsigned long long *pbits;
};
// number of P keys first seen before exchange cell f = dense id of the key first seen AT f
__device__ __forceinline__ uint32_t sa_prank_at(const uint32_t *__restrict__ Pw, const unsigned long long *__restrict__ pbits, long f) {
    const long w = f >> 6;
    const int k = (int)(f & 63);
    return Pw[w] + (uint32_t)__popcll(pbits[w] & ((1ull << k) - 1ull));
}
__device__ __forceinline__ uint32_t fa_poff_at(const uint32_t *__restrict__ poff, const uint32_t *__restrict__ Pw, const unsigned long long *__restrict__ pbits, long f) {
    return Pw ? sa_prank_at(Pw, pbits, f) : poff[f];
}
__device__ __forceinline__ int fa_class(const RgView &rg, int gkey_kind, long key) {
    if (gkey_kind != KEY_E) return 0;
    long a, hc;
    e_decode(rg, key, a, hc);
    return (int)hc;
}
__device__ __forceinline__ int fa_pdense(const FaP &p, const PlanView &pl, long iI, long x) {
    if (p.key == KEY_I) return p.fresh ? (int)fa_poff_at(p.poff, p.Pw, p.pbits, pl.ifirst[iI]) : (int)iI;
    return p.fresh ? (int)fa_poff_at(p.poff, p.Pw, p.pbits, x) : (int)x;
}
// is the P key of this entry numbered inside range [x0, ...)?  (then its dense ids ascend along x)
__device__ __forceinline__ bool fa_pnew(const FaP &p, const PlanView &pl, long iI, long x0) {
    if (p.key != KEY_I || !p.fresh) return true;
    return pl.ifirst[iI] >= x0;
}

// Entries of the duplicate group that starts at exchange cell x (cells with the same (iA, iI) follow each
// other in a sorted grid and contribute to the same matrix entries: summed in x order, first term
// assigned).  <= 2 entries: the two elevation classes of the ice cell.  X-keyed matrices never merge
// (every exchange cell is its own key).
struct GEnt {                     // scalar members on purpose: indexed private arrays live in scratch memory
    int n, cls0, cls1;
    long gkey0, gkey1;
    double t0, t1;
    __device__ __forceinline__ int cls(int j) const { return j ? cls1 : cls0; }
    __device__ __forceinline__ long gkey(int j) const { return j ? gkey1 : gkey0; }
    __device__ __forceinline__ double t(int j) const { return j ? t1 : t0; }
};
template <bool WITH_EP>
__device__ __forceinline__ void fa_group(const RgView &rg, const PlanView &pl, const MatSpec &s, bool g_is_row, bool merge,
                                         const XCell &c0, long x, GEnt &g) {
    g.n = 0; g.cls0 = g.cls1 = 0; g.gkey0 = g.gkey1 = 0; g.t0 = g.t1 = 0.0;
    long xm = x;
    XCell c = c0;
    while (true) {
        if (!(WITH_EP && c.range_error)) {
            Contrib o;
            const int n = contributions(c, xm, s, o);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j >= n) break;
                const long gk = g_is_row ? (j ? o.r1 : o.r0) : (j ? o.c1 : o.c0);
                const double t = j ? o.t1 : o.t0;
                const int cls = fa_class(rg, g_is_row ? s.row_key : s.col_key, gk);
                if (g.n > 0 && g.cls0 == cls) g.t0 = g.t0 + t;
                else if (g.n > 1 && g.cls1 == cls) g.t1 = g.t1 + t;
                else if (g.n == 0) { g.cls0 = cls; g.gkey0 = gk; g.t0 = t; g.n = 1; }
                else if (g.n == 1) { g.cls1 = cls; g.gkey1 = gk; g.t1 = t; g.n = 2; }
            }
        }
        ++xm;
        if (!merge || xm >= rg.nX || !(pl.isdup[xm] & 1)) break;
        c = load_cell<WITH_EP>(rg, xm);
    }
}

// first-occurrence flag of the P key at exchange cell x (the scan of these flags numbers the P set)
template <bool WITH_EP>
__device__ __forceinline__ uint32_t fa_pflag_of(const RgView &rg, const PlanView &pl, const FaP &p, const XCell &c, long x) {
    if (p.key == KEY_I) return (pl.ifirst[c.iI] == (int32_t)x && c.unmasked) ? 1u : 0u;       // GvI: unmasked, area != 0
    if (WITH_EP && c.range_error) return 0u;
    long k0, k1;
    return list_entries(c, x, p.list, KEY_X, k0, k1) > 0 ? 1u : 0u;
}

// ---- G-side numbering and entry counts: one workgroup per range -----------------------------------
// Small grids (round 5): the three scans that used to follow this kernel -- the first-occurrence flags of the P keys over the
// cells, the classes and the entries over the ranges: two launches -- are CHAINED through it.  A range's workgroup publishes its
// three totals in one 64-bit status word {flag 2 | epoch 4 | new P keys 21 | entries 22 | classes 15 bits} and looks back over its
// predecessors' words (decoupled look-back, as prims.hip scan_chained: workgroups are dispatched in index order and publish
// their own totals before they wait, so the wait always ends); the ranks of the new keys inside the range come from ballots.
// status == nullptr: off (large grids scan with kernels of their own; EvA / AvE count other things).
struct FaChain {
    unsigned long long *status;     // [nAr]
    unsigned epoch;                 // 1..15: words of another epoch are "not yet written"
    uint32_t *tot;                  // [3]: new P keys, classes, entries of the whole grid (the counters read back by the host)
    uint32_t *zero;                 // words a later kernel of the build wants cleared (a hipMemsetAsync of an odd size is TWO launches)
    unsigned nzero;
    int32_t *fill_m1;               // words to set to -1 (the sparse -> dense table of the G set, FaG::tab_out)
    unsigned nfill;
    uint32_t *zero2;                // a second array to clear (Mw over a pre-populated column set: columns this mask does not touch)
    unsigned nzero2;
    // I-row builds on the identity ice set: the row lengths (k_fa_pelem<COUNT>'s static form) do not depend on this kernel's results,
    // so the workgroups pc_first .. of ITS launch compute them -- one launch fewer in the coupler's IvE
    // ... and scan them into the row pointer on the way (a second chain of status words, pc_status0 ..: these workgroups are
    // dispatched in index order behind the ranges'), so the scan's launch goes too
    uint32_t *xr_rowptr;            // X-row builds on the identity exchange-cell set: the row pointer of cell x is the entries before
                                    // its range + those of the cells before it in the range -- written by the range's own workgroup
    // I-row builds on an ice set numbered by this build: a row belongs to the cell its ice cell is first seen at, its static length
    // is known there, rows follow each other in first-seen = x order -- a third chain (ir_status0 ..) carries the entries before a range
    uint32_t *ir_rowptr, *ir_tmp;   // (ir_tmp[nX]: the in-range part of ranges longer than one pass)
    int ir_status0;
    uint32_t *pc_rowlen;            // (the lengths themselves, when no row pointer is wanted here)
    uint32_t *pc_rowptr;
    long pc_n;
    int pc_first, pc_status0;
};
// entries of the row of ice cell q, identity ice set, from the plan's static counts (k_fa_pelem<COUNT>, first branch)
template <bool WITH_EP>
__device__ __forceinline__ uint32_t fa_pcount_static(const RgView &rg, const PlanView &pl, long q, long np) {
    if (q >= np) return 0u;
    const double e = rg.em[q];
    uint32_t n = 0;
    if (!(e != e)) {
        if (WITH_EP) {
            const XCell c = make_cell<WITH_EP>(rg, 0, q, 1.0, e);
            n = c.range_error ? 0u : (uint32_t)c.nep * pl.icnt_nz[q];
        } else n = pl.icnt_pos[q];
    }
    return n;
}
template <int T>
__device__ __forceinline__ uint32_t fa_block_excl_scan(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wave) wbase += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return wbase + inc - v;
}
constexpr int FA_CH_NP = 21, FA_CH_NE = 22, FA_CH_NC = 15;
// Exclusive prefix of `mine` (<= 58 bits, fields that cannot overflow) over the workgroups before r; every thread of the workgroup
// calls it and gets the same value.  One round looks at T predecessors at once -- at 5 km every range sees all the others in one.
template <int T>
__device__ __forceinline__ unsigned long long fa_chain_prefix(unsigned long long *status, int r, unsigned epoch, unsigned long long mine,
                                                              unsigned long long *s_part, int *s_flag) {
    constexpr unsigned long long AGG = 1, PRE = 2;
    const unsigned long long tag = (unsigned long long)(epoch & 15u) << 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) __hip_atomic_store(status + r, (mine << 6) | tag | (r == 0 ? PRE : AGG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (r == 0) return 0ull;
    unsigned long long prefix = 0;
    long j = (long)r - 1;
    while (true) {
        const long idx = j - (long)threadIdx.x;
        unsigned long long w = tag | PRE;                       // before the first workgroup: an empty prefix
        if (idx >= 0) {
            do { w = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            while ((w & 0x3cull) != tag || (w & 3ull) == 0);
        }
        const unsigned long long pre = __ballot((w & 3ull) == PRE);
        unsigned long long val = w >> 6;
        if (pre && lane > __builtin_ctzll(pre)) val = 0;        // beyond the nearest predecessor that knows everything before it
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) val += __shfl_xor(val, off, 64);
        if (lane == 0) { s_part[wave] = val; s_flag[wave] = pre != 0; }
        __syncthreads();
        bool done = false;
        for (int q = 0; q < T / 64 && !done; ++q) { prefix += s_part[q]; done = s_flag[q] != 0; }
        __syncthreads();
        if (done) break;
        j -= T;
    }
    if (threadIdx.x == 0) __hip_atomic_store(status + r, ((prefix + mine) << 6) | tag | PRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return prefix;
}
template <bool WITH_EP, int T, int CPT>
__global__ __launch_bounds__(T, T == 128 ? 8 : 1) void k_fa_count(RgView rg, PlanView pl, MatSpec s, FaG g, FaP p, int g_is_row, int merge,
                                                    uint32_t *__restrict__ err_x, uint32_t *__restrict__ flags, int eva_check, FaChain ch) {
    __shared__ double s_hc[WITH_EP ? HC_LDS : 1];
    __shared__ uint32_t s_first[FA_NC], s_cn[FA_NC], s_co[FA_NC];
    __shared__ uint32_t s_pw[CPT][T / 64], s_runp, s_nc, s_ne;
    __shared__ uint32_t s_xw[CPT][T / 64], s_runx;            // (shared by the X-row and the I-row form: a build is one or the other)
    __shared__ unsigned long long s_part[T / 64];
    __shared__ int s_flag[T / 64];
    stage_hc<WITH_EP>(rg, s_hc);
    const bool chained = ch.status != nullptr, prank = chained && p.fresh;
    if (chained && ch.zero)
        for (unsigned i = blockIdx.x * T + threadIdx.x; i < ch.nzero; i += gridDim.x * T) ch.zero[i] = 0u;
    if (chained && ch.fill_m1)
        for (unsigned i = blockIdx.x * T + threadIdx.x; i < ch.nfill; i += gridDim.x * T) ch.fill_m1[i] = -1;
    if (chained && ch.zero2)
        for (unsigned i = blockIdx.x * T + threadIdx.x; i < ch.nzero2; i += gridDim.x * T) ch.zero2[i] = 0u;
    if (chained && ch.pc_n > 0 && (int)blockIdx.x >= ch.pc_first) {           // (uniform: this workgroup counts row lengths)
        const int b = (int)blockIdx.x - ch.pc_first;
        const long q = (long)b * T + threadIdx.x;
        const uint32_t n = fa_pcount_static<WITH_EP>(rg, pl, q, ch.pc_n);
        if (ch.pc_rowptr) {
            uint32_t tot;
            const uint32_t ex = fa_block_excl_scan<T>(n, s_pw[0], tot);
            const unsigned long long before = fa_chain_prefix<T>(ch.status + ch.pc_status0, b, ch.epoch, (unsigned long long)tot, s_part, s_flag);
            if (q < ch.pc_n) ch.pc_rowptr[q] = (uint32_t)before + ex;
            if (b == (int)gridDim.x - ch.pc_first - 1 && threadIdx.x == 0) ch.pc_rowptr[ch.pc_n] = (uint32_t)before + tot;
        } else if (q < ch.pc_n) ch.pc_rowlen[q] = n;
        return;
    }
    const int r = blockIdx.x;
    const long x0 = pl.arng[r], x1 = pl.arng[r + 1];
    const int iA_r = rg.exi[2 * x0];                          // the atmosphere cell of the whole range
    if (threadIdx.x < FA_NC) { s_first[threadIdx.x] = 0xffffffffu; s_cn[threadIdx.x] = 0; s_co[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { s_runp = 0; s_runx = 0; }
    const bool xrows = chained && ch.xr_rowptr != nullptr, irows = chained && ch.ir_rowptr != nullptr;
    uint32_t xv[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) xv[u] = 0;
    const bool one_pass = x1 - x0 <= (long)T * CPT;         // then the ranks wait in registers for the keys before the range
    uint32_t pv[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) pv[u] = 0;
    __syncthreads();
    // CPT cells per thread and pass with their loads staged (exchange cell, then mask / first-seen position of the ice
    // cell) before anything is consumed: a range of ~10^3 cells is one pass, i.e. two dependent round trips in all
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (long cb = x0; cb < x1; cb += (long)T * CPT) {
        int iAv[CPT], iIv[CPT], ifv[CPT];
        double av[CPT], ev[CPT];
        unsigned fl[CPT], prk[CPT], xrk[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const long x = cb + (long)u * T + threadIdx.x;
            const long xx = x < x1 ? x : x1 - 1;
            iAv[u] = iA_r; iIv[u] = pl.exI[xx]; fl[u] = pl.isdup[xx];
            if (WITH_EP) av[u] = rg.area[xx];                   // (the class weights are products with the area)
        }
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            ev[u] = rg.em[iIv[u]];
            if (!WITH_EP) av[u] = (fl[u] & 8) ? 1.0 : (fl[u] & 16) ? -1.0 : 0.0;      // counting needs the sign class only (plan bits 3 / 4)
            const long x = cb + (long)u * T + threadIdx.x;
            // first-seen position of the ice cell: this very cell for ~90 % of them (plan bit 2), a gather for the rest
            ifv[u] = p.key == KEY_I ? ((fl[u] & 4) ? (int)(x < x1 ? x : x1 - 1) : pl.ifirst[iIv[u]]) : 0;
        }
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const long x = cb + (long)u * T + threadIdx.x;
            const bool in = x < x1;
            uint32_t cn0 = 0, co0 = 0;                        // one-class ranges: this lane's new / old entries
            uint32_t pf = 0, xn = 0;                          // (xn: entries of this cell's group -- the row length of an X row)
            if (in) {
                const XCell c = make_cell<WITH_EP>(rg, iAv[u], iIv[u], av[u], ev[u]);
                if (p.fresh) {
                    pf = p.key == KEY_I ? ((ifv[u] == (int32_t)x && c.unmasked) ? 1u : 0u) : fa_pflag_of<WITH_EP>(rg, pl, p, c, x);
                    p.pflag[x] = (uint8_t)pf;
                    if (irows && pf) {                        // the static row length (k_fa_pelem<COUNT>, first branch), at the first-seen cell
                        if (WITH_EP) {
                            const XCell c1 = make_cell<WITH_EP>(rg, 0, iIv[u], 1.0, ev[u]);
                            xn = c1.range_error ? 0u : (uint32_t)c1.nep * pl.icnt_nz[iIv[u]];
                        } else xn = pl.icnt_pos[iIv[u]];
                    }
                }
                if (WITH_EP && c.range_error) atomicMin(err_x, (uint32_t)x);
                else {
                    // EvA / AvE number the A side by "the range has entries": a cell of GvAp without an elevation-class entry
                    // (an area so small that area * weight underflows) would be numbered without one -- general pipeline
                    if (WITH_EP && eva_check && c.inAp && c.nep == 0) atomicOr(flags, (uint32_t)FA_ERR_MISSING);
                    long k0, k1;
                    const int n = list_entries(c, x, g.list, g.key, k0, k1);        // first-seen positions of the G keys: every cell counts
                    if (g.NC == 1) { if (n > 0) atomicMin(&s_first[0], (uint32_t)(2 * x)); }     // (one address: cheap; first hit wins quickly)
                    else {
                        if (n > 0) atomicMin(&s_first[fa_class(rg, g.key, k0)], (uint32_t)(2 * x));
                        if (n > 1) atomicMin(&s_first[fa_class(rg, g.key, k1)], (uint32_t)(2 * x + 1));
                    }
                    if (!(merge && (fl[u] & 1))) {
                        GEnt ge;
                        fa_group<WITH_EP>(rg, pl, s, g_is_row != 0, merge != 0 && (fl[u] & 128), c, x, ge);      // (plan bit 7: a duplicate follows)
                        const bool isnew = p.key != KEY_I || !p.fresh || ifv[u] >= x0;
                        if (!irows) xn = (uint32_t)ge.n;
                        if (g.NC == 1) { cn0 = (ge.n > 0 && isnew) ? 1u : 0u; co0 = (ge.n > 0 && !isnew) ? 1u : 0u; }
                        else {
                            if (ge.n > 0) atomicAdd(isnew ? &s_cn[ge.cls0] : &s_co[ge.cls0], 1u);
                            if (ge.n > 1) atomicAdd(isnew ? &s_cn[ge.cls1] : &s_co[ge.cls1], 1u);
                        }
                    }
                }
            }
            if (g.NC == 1) {                                  // one counter pair: a ballot per wave instead of an atomic per lane
                const unsigned long long bn = __ballot(cn0 != 0), bo = __ballot(co0 != 0);
                if (lane == 0) { if (bn) atomicAdd(&s_cn[0], (uint32_t)__popcll(bn)); if (bo) atomicAdd(&s_co[0], (uint32_t)__popcll(bo)); }
            }
            if (prank) {                                      // rank of a new P key among the new keys of its range, in x order
                const unsigned long long bp = __ballot(pf != 0);
                prk[u] = (uint32_t)__popcll(bp & lt);
                if (lane == 0) s_pw[u][wave] = (uint32_t)__popcll(bp);
            }
            if (irows) {                                      // row entries of the first-seen cells before this one in the range
                uint32_t inc = xn;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
                xrk[u] = inc - xn;
                if (lane == 63) s_xw[u][wave] = inc;
            }
            if (xrows) {                                      // entries of the cells before this one in the range (0, 1 or 2 each)
                const unsigned long long b1 = __ballot(xn >= 1), b2 = __ballot(xn == 2);
                xrk[u] = (uint32_t)(__popcll(b1 & lt) + __popcll(b2 & lt));
                if (lane == 0) s_xw[u][wave] = (uint32_t)(__popcll(b1) + __popcll(b2));
            }
        }
        if (irows) {
            // (xrk / s_xw were filled in the loop above -- see "irows" there)
            __syncthreads();
            uint32_t before = s_runx, all = 0;
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                uint32_t mine = before;
                for (int w = 0; w < T / 64; ++w) { const uint32_t c = s_xw[u][w]; if (w < wave) mine += c; before += c; all += c; }
                const long x = cb + (long)u * T + threadIdx.x;
                if (one_pass) xv[u] = mine + xrk[u];
                else if (x < x1) ch.ir_tmp[x] = mine + xrk[u];
            }
            __syncthreads();
            if (threadIdx.x == 0) s_runx += all;
        }
        if (xrows) {
            __syncthreads();
            uint32_t before = s_runx, all = 0;
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                uint32_t mine = before;
                for (int w = 0; w < T / 64; ++w) { const uint32_t c = s_xw[u][w]; if (w < wave) mine += c; before += c; all += c; }
                const long x = cb + (long)u * T + threadIdx.x;
                if (one_pass) xv[u] = mine + xrk[u];
                else if (x < x1) ch.xr_rowptr[x] = mine + xrk[u];      // (inside the range: the entries before the range are added below)
            }
            __syncthreads();
            if (threadIdx.x == 0) s_runx += all;
        }
        if (prank) {
            __syncthreads();
            uint32_t before = s_runp, all = 0;
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                uint32_t mine = before;
                for (int w = 0; w < T / 64; ++w) { const uint32_t c = s_pw[u][w]; if (w < wave) mine += c; before += c; all += c; }
                const long x = cb + (long)u * T + threadIdx.x;
                if (one_pass) pv[u] = mine + prk[u];
                else if (x < x1) p.poff[x] = mine + prk[u];   // (inside the range: the keys before the range are added below)
            }
            __syncthreads();
            if (threadIdx.x == 0) s_runp += all;
        }
    }
    __syncthreads();
    if (threadIdx.x < FA_NC && (int)threadIdx.x < g.NC) {
        const int k = threadIdx.x;
        const uint32_t f = s_first[k];
        int rank = -1;
        if (f != 0xffffffffu) {
            rank = 0;
            for (int q = 0; q < g.NC; ++q) rank += s_first[q] < f ? 1 : 0;
            if (g.tab) {                                                  // pre-populated set: every key must already be there
                const long iA = rg.exi[2 * x0];
                const long key = g.key == KEY_E ? iA * rg.sA + (long)k * rg.sHC : iA;
                if (g.tab[key] < 0) atomicOr(flags, (uint32_t)FA_ERR_MISSING);
            }
        } else if (s_cn[k] + s_co[k]) atomicOr(flags, (uint32_t)FA_ERR_MISSING);    // entries of a class that was never listed: cannot happen
        g.erank[(size_t)r * g.NC + k] = (int8_t)rank;
        g.ecntn[(size_t)r * g.NC + k] = s_cn[k];
        g.ecnto[(size_t)r * g.NC + k] = s_co[k];
    }
    if (threadIdx.x == 0) {
        uint32_t nc = 0, ne = 0, no = 0;
        for (int k = 0; k < g.NC; ++k) { nc += s_first[k] != 0xffffffffu ? 1u : 0u; ne += s_cn[k] + s_co[k]; no += s_co[k]; }
        g.r_ncls[r] = nc; g.r_nent[r] = ne;
        if (no > (uint32_t)FA_OLDMAX) atomicOr(flags, (uint32_t)FA_ERR_OLDOVER);
        s_nc = nc; s_ne = ne;
    }
    if (!chained) return;
    __syncthreads();
    const unsigned long long mine = (unsigned long long)s_runp | ((unsigned long long)s_ne << FA_CH_NP) | ((unsigned long long)s_nc << (FA_CH_NP + FA_CH_NE));
    const unsigned long long prefix = fa_chain_prefix<T>(ch.status, r, ch.epoch, mine, s_part, s_flag);
    const unsigned long long m21 = (1ull << FA_CH_NP) - 1, m22 = (1ull << FA_CH_NE) - 1;
    if (threadIdx.x == 0) {
        g.gbase[r] = (uint32_t)(prefix >> (FA_CH_NP + FA_CH_NE));
        g.ebase[r] = (uint32_t)((prefix >> FA_CH_NP) & m22);
        if (r == pl.nAr - 1) {
            const unsigned long long t = prefix + mine;
            const uint32_t tc = (uint32_t)(t >> (FA_CH_NP + FA_CH_NE)), te = (uint32_t)((t >> FA_CH_NP) & m22);
            g.gbase[r + 1] = tc; g.ebase[r + 1] = te;
            ch.tot[0] = (uint32_t)(t & m21); ch.tot[1] = tc; ch.tot[2] = te;
        }
    }
    if (xrows) {
        const uint32_t xbase = (uint32_t)((prefix >> FA_CH_NP) & m22);
        if (one_pass) {
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const long x = x0 + (long)u * T + threadIdx.x;
                if (x < x1) ch.xr_rowptr[x] = xv[u] + xbase;
            }
        } else if (xbase) {
            for (long x = x0 + threadIdx.x; x < x1; x += T) ch.xr_rowptr[x] += xbase;
        }
        if (r == pl.nAr - 1 && threadIdx.x == 0) ch.xr_rowptr[x1] = (uint32_t)(((prefix + mine) >> FA_CH_NP) & m22);      // rowptr[nX] = nnz
    }
    const uint32_t pbase = (uint32_t)(prefix & m21);
    if (prank) {
        if (one_pass) {
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const long x = x0 + (long)u * T + threadIdx.x;
                if (x < x1) p.poff[x] = pv[u] + pbase;
            }
        } else if (pbase) {
            for (long x = x0 + threadIdx.x; x < x1; x += T) p.poff[x] += pbase;
        }
    }
    if (irows) {
        // the rows of the ice cells first seen in this range start behind those of the ranges before it (third chain); the dense id of
        // such a cell = the keys before the range + its rank (a first-seen cell's own exclusive count)
        const unsigned long long rb = fa_chain_prefix<T>(ch.status + ch.ir_status0, r, ch.epoch, (unsigned long long)s_runx, s_part, s_flag);
        if (one_pass) {
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const long x = x0 + (long)u * T + threadIdx.x;
                if (x < x1 && p.pflag[x]) ch.ir_rowptr[pv[u] + pbase] = (uint32_t)rb + xv[u];
            }
        } else {
            for (long x = x0 + threadIdx.x; x < x1; x += T)
                if (p.pflag[x]) ch.ir_rowptr[p.poff[x]] = (uint32_t)rb + ch.ir_tmp[x];
        }
        if (r == pl.nAr - 1 && threadIdx.x == 0) ch.ir_rowptr[(uint32_t)((prefix + mine) & m21)] = (uint32_t)rb + s_runx;
    }
}
// ---- the same counts for ONE-class matrices (AvI, AvX, IvA, XvA: the G side is the atmosphere grid, no elevation classes),
// streamed: a thread owns exchange cells, not a range.  k_fa_count gives every range a workgroup that walks it pass by pass --
// two dependent load rounds per pass, a barrier-fenced epilogue -- which is latency, not bytes (2.8 TB/s at best; at 1 km only
// 553 workgroups exist at all).  Here every wave takes 64 consecutive cells, reads them once, and adds what it found to the
// range's counters: lanes are sorted by range (the grid is sorted by (iA, iI)), so a wave holds a few runs and the first lane
// of each run issues the run's integer atomics (sums of integers: the result does not depend on the order).
__global__ void k_fa_zero_counts(uint32_t *__restrict__ a, uint32_t *__restrict__ b, uint32_t *__restrict__ c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0; b[i] = 0; c[i] = 0; }
}
template <int CPT>
__global__ __launch_bounds__(256) void k_fa_count_stream(RgView rg, PlanView pl, MatSpec s, FaG g, FaP p, int g_is_row, int merge) {
    const int lane = threadIdx.x & 63;
    const unsigned long long le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1);
    const long cb = (long)blockIdx.x * (256 * CPT);
    int iAv[CPT], iIv[CPT], rv[CPT];
    unsigned fl[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const long x = cb + (long)u * 256 + threadIdx.x;
        const long xx = x < rg.nX ? x : rg.nX - 1;
        iAv[u] = 0; iIv[u] = pl.exI[xx]; fl[u] = pl.isdup[xx]; rv[u] = pl.aidx[xx];      // (the atmosphere index plays no part in a count)
    }
    double ev[CPT];
    int ifv[CPT], x0v[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const long x = cb + (long)u * 256 + threadIdx.x;
        ev[u] = rg.em[iIv[u]];
        ifv[u] = p.key == KEY_I ? ((fl[u] & 4) ? (int)(x < rg.nX ? x : rg.nX - 1) : pl.ifirst[iIv[u]]) : 0;
        x0v[u] = pl.arng[rv[u]];
    }
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const long x = cb + (long)u * 256 + threadIdx.x;
        const bool in = x < rg.nX;
        bool member = false, cnew = false, cold = false;
        if (in) {
            const XCell c = make_cell<false>(rg, iAv[u], iIv[u], (fl[u] & 8) ? 1.0 : (fl[u] & 16) ? -1.0 : 0.0, ev[u]);
            if (p.fresh) p.pflag[x] = (uint8_t)(p.key == KEY_I ? ((ifv[u] == (int32_t)x && c.unmasked) ? 1u : 0u) : fa_pflag_of<false>(rg, pl, p, c, x));
            long k0, k1;
            member = list_entries(c, x, g.list, g.key, k0, k1) > 0;
            if (!(merge && (fl[u] & 1))) {
                GEnt ge;
                fa_group<false>(rg, pl, s, g_is_row != 0, merge != 0, c, x, ge);
                const bool isnew = p.key != KEY_I || !p.fresh || ifv[u] >= x0v[u];
                cnew = ge.n > 0 && isnew;
                cold = ge.n > 0 && !isnew;
            }
        }
        // runs of equal range inside the wave (lanes past the end form a run of their own: range id -1)
        const int r = in ? rv[u] : -1;
        const int rp = __shfl_up(r, 1, 64);
        const unsigned long long heads = __ballot(lane == 0 || r != rp);
        const unsigned long long bm = __ballot(member), bn = __ballot(cnew), bo = __ballot(cold);
        if (in && ((heads >> lane) & 1)) {
            const unsigned long long rest = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int end = rest ? lane + 1 + __builtin_ctzll(rest) : 64;
            const unsigned long long run = (end == 64 ? ~0ull : ((1ull << end) - 1)) & ~(le_mask >> 1);
            const unsigned nm = (unsigned)__popcll(bm & run), nn = (unsigned)__popcll(bn & run), no = (unsigned)__popcll(bo & run);
            if (nm) atomicOr(&g.r_ncls[r], 1u);
            if (nn) atomicAdd(&g.ecntn[r], nn);
            if (no) atomicAdd(&g.ecnto[r], no);
        }
    }
}
__global__ void k_fa_count_fin(FaG g, int nAr, uint32_t *__restrict__ flags) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nAr) return;
    const uint32_t memb = g.r_ncls[r], cn = g.ecntn[r], co = g.ecnto[r];
    g.r_nent[r] = cn + co;
    g.erank[r] = memb ? 0 : -1;
    if (!memb && (cn + co)) atomicOr(flags, (uint32_t)FA_ERR_MISSING);       // entries of a class that was never listed: cannot happen
    if (co > (uint32_t)FA_OLDMAX) atomicOr(flags, (uint32_t)FA_ERR_OLDOVER);
}

// Tried on top and dropped (measured on MI355X): the same streaming for the matrices WITH elevation classes -- counters per
// (range, class), the lanes of a wave matched by key, atomicMin for the first-seen positions -- is correct (the suite passed)
// but slower than the per-range workgroups: a1h EvI 1.72 -> 1.83 ms, 1 km EvI 0.32 -> 0.42 (the counters of a class take
// thousands of global atomics each); letting the streamed count also add up the tiles of the flag scan that follows (one
// atomic per wave, saving the scan's first pass) lost too (a1h AvI 1.04 -> 1.10); so did one chunked single-workgroup scan
// over the ranges in place of two device-wide ones (a1h AvI 1.22 -> 1.27: 49 strided elements per thread).

// exclusive scans over the ranges (one workgroup: there are 10^2..10^5 ranges); tot[0] = classes, tot[1] = entries
__device__ __forceinline__ uint32_t fa_block_excl_scan_1024(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wave) wbase += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return wbase + inc - v;
}
__global__ __launch_bounds__(1024) void k_fa_rscan(const uint32_t *__restrict__ ncls, const uint32_t *__restrict__ nent, int nAr,
                                                    uint32_t *__restrict__ gbase, uint32_t *__restrict__ ebase, uint32_t *__restrict__ tot) {
    __shared__ uint32_t s_wave[16];
    for (int ch = 0; ch < 2; ++ch) {
        const uint32_t *in = ch ? nent : ncls;
        uint32_t *out = ch ? ebase : gbase;
        uint32_t carry = 0;
        for (int base = 0; base < nAr; base += 1024) {
            const int i = base + threadIdx.x;
            const uint32_t v = i < nAr ? in[i] : 0u;
            uint32_t t;
            const uint32_t ex = fa_block_excl_scan_1024(v, s_wave, t);
            if (i < nAr) out[i] = carry + ex;
            carry += t;
        }
        if (threadIdx.x == 0) { out[nAr] = carry; tot[ch] = carry; }
        __syncthreads();
    }
}
// the same for MANY ranges (the Antarctic sheet: 49 771): three short launches for all channels together -- tile sums (one
// range per thread), the scan of the <= 1024 tile sums, the tile-local scans -- in place of one device-wide scan per channel
// plus a copy of each total (six kernels and three copies, ~45 us, for two channels).  Channel 2 (EvA / AvE: ranges that have
// entries) is optional.
__global__ __launch_bounds__(1024) void k_fa_rscan_tiles(const uint32_t *__restrict__ ncls, const uint32_t *__restrict__ nent, int nAr,
                                                          int nch, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_wave[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    for (int ch = 0; ch < nch; ++ch) {
        const uint32_t v = i < nAr ? (ch == 0 ? ncls[i] : ch == 1 ? nent[i] : (nent[i] > 0 ? 1u : 0u)) : 0u;
        uint32_t t;
        (void)fa_block_excl_scan_1024(v, s_wave, t);
        if (threadIdx.x == 0) sums[ch * gridDim.x + blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_fa_rscan_sums(uint32_t *__restrict__ sums, int nb, int nch, uint32_t *__restrict__ tot, uint32_t *__restrict__ tot_a) {
    __shared__ uint32_t s_wave[16];
    for (int ch = 0; ch < nch; ++ch) {
        const uint32_t v = (int)threadIdx.x < nb ? sums[ch * nb + threadIdx.x] : 0u;
        uint32_t t;
        const uint32_t ex = fa_block_excl_scan_1024(v, s_wave, t);
        if ((int)threadIdx.x < nb) sums[ch * nb + threadIdx.x] = ex;
        if (threadIdx.x == 0) (ch == 2 ? tot_a : tot + ch)[0] = t;
    }
}
__global__ __launch_bounds__(1024) void k_fa_rscan_apply(const uint32_t *__restrict__ ncls, const uint32_t *__restrict__ nent, int nAr, int nch,
                                                          const uint32_t *__restrict__ sums, const uint32_t *__restrict__ tot,
                                                          const uint32_t *__restrict__ tot_a, uint32_t *__restrict__ gbase,
                                                          uint32_t *__restrict__ ebase, uint32_t *__restrict__ abase) {
    __shared__ uint32_t s_wave[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    for (int ch = 0; ch < nch; ++ch) {
        const uint32_t v = i < nAr ? (ch == 0 ? ncls[i] : ch == 1 ? nent[i] : (nent[i] > 0 ? 1u : 0u)) : 0u;
        uint32_t t;
        const uint32_t ex = fa_block_excl_scan_1024(v, s_wave, t) + sums[ch * gridDim.x + blockIdx.x];
        uint32_t *out = ch == 0 ? gbase : ch == 1 ? ebase : abase;
        if (i < nAr) out[i] = ex;
        if (i == nAr - 1) out[nAr] = (ch == 2 ? tot_a : tot + ch)[0];
    }
}
// launches the three; false when there are too many ranges for one workgroup of tile sums (the caller scans channel by channel)
static bool fa_rscan_many(const FaG &g, int nAr, uint32_t *abase, uint32_t *tot, uint32_t *tot_a, hipStream_t st) {
    const int nb = ceil_div(nAr, 1024);
    if (nb > 1024) return false;
    const int nch = abase ? 3 : 2;
    uint32_t *sums = arena().get<uint32_t>((size_t)nch * nb);
    hipLaunchKernelGGL(k_fa_rscan_tiles, dim3(nb), dim3(1024), 0, st, g.r_ncls, g.r_nent, nAr, nch, sums);
    hipLaunchKernelGGL(k_fa_rscan_sums, dim3(1), dim3(1024), 0, st, sums, nb, nch, tot, tot_a);
    hipLaunchKernelGGL(k_fa_rscan_apply, dim3(nb), dim3(1024), 0, st, g.r_ncls, g.r_nent, nAr, nch, (const uint32_t *)sums, (const uint32_t *)tot,
                       (const uint32_t *)tot_a, g.gbase, g.ebase, abase);
    return true;
}
__device__ __forceinline__ int fa_gdense(const FaG &g, int r, int cls, long gkey) {
    if (g.tab) return g.tab[gkey];
    return (int)g.gbase[r] + (int)g.erank[(size_t)r * g.NC + cls];
}

// ---- one workgroup per range: CSR rows (A/E-row matrices) or column sums (I/X-row matrices) ---------
// Every entry of the range goes straight to slot ebase[r] + start(class) + position inside the class
// segment, where the segment lists the entries by ascending P dense id: first the "old" ones (ice cells
// first seen in an earlier range: smaller ids, ranked in LDS), then the "new" ones in x order (their ids
// ascend along x).  Then one thread per segment adds it up sequentially -- the order of spsparse sum().
struct FaOut {
    int32_t *rowptr, *colind;       // EMIT: CSR of the matrix
    double *val;                    // EMIT: CSR values; SUMS: scratch of the same size
    double *wM, *Mw;                // EMIT writes wM (rows = G side), SUMS writes Mw (cols = G side)
    int family, scale, correctA;
    int g_rows;                     // SUMS only: the G side is the ROW side of the spec (EvA through the column-sum machinery)
    // the LAST kernel of a small build hands the eight counters to the host itself (pinned memory) -- the workgroup that finishes
    // last copies them -- instead of a copy kernel behind it (one launch of ~4 us in a chain of four)
    uint32_t *pub_ticket;
    const uint32_t *pub_src;
    uint32_t *pub_dst;
    // k_fa_range<EMIT> of a small build also carries the workgroups of k_fa_psums8 (blocks psums_first ..: the two are independent,
    // neither fills the chip, and a launch of its own costs what the work does)
    int psums_first;
    long psums_n;
#ifdef FA_TIMELINE
    long long *dbg;                 // scratch/r05/range_timeline.py: 8 stamps per range from thread 0
#endif
};
__device__ __forceinline__ void fa_publish(const FaOut &o) {
    if (!o.pub_ticket) return;                                  // (uniform)
    // this thread's counter updates (device-scope atomics, performed at the memory side) are complete before the ticket is taken:
    // waiting for their acknowledgement is enough -- a release fence here would write the whole L2 back once per thread
    // (measured: +45 us on a 5 km IvE build)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(o.pub_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            for (int i = 0; i < 8; ++i) o.pub_dst[i] = __hip_atomic_load(o.pub_src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o.pub_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (for the next build)
        }
    }
}
#ifdef FA_TIMELINE
#define FA_STAMP(i) do { if (tid == 0) { tl[i] = (long long)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define FA_STAMP(i) do { } while (0)
#endif
// Mw of the ice cells with several exchange cells (k_fa_pelem<SUMS>'s job), EIGHT lanes per ice cell: every lane evaluates one of
// the cell's <= 8 exchange cells, lane 0 of the group then adds the terms in the element's order (cells ascending, the two classes
// of a cell by their dense ids) -- the same sequence of additions.  One thread per ice cell walked its cells one after the other,
// ~4 k cycles each; with 13 k such ice cells at 5 km the kernel was a few workgroups waiting for their 8-cell lanes (EvI: 16 us).
template <bool WITH_EP>
__device__ __forceinline__ void fa_psums8_body(const RgView &rg, const PlanView &pl, const MatSpec &s, const FaG &g, const FaP &p, int merge, long np, const FaOut &o,
                                               int first_block) {
    const long t = (long)((int)blockIdx.x - first_block) * blockDim.x + threadIdx.x;
    const long el = t >> 3;
    const int j = (int)(t & 7), lane = threadIdx.x & 63, base = lane & ~7;
    int d = -1, lb = 0, le = 0;
    long q = 0;
    double e = 0.0;
    if (el < np) {
        q = pl.mlist[el];
        e = rg.em[q];
        const bool masked = e != e;
        if (!(masked && p.fresh)) {                          // (a masked cell is no member of a set numbered by this build)
            const int f = pl.ifirst[q];
            if (!p.fresh) d = (int)q;
            else if (f >= 0) d = (int)fa_poff_at(p.poff, p.Pw, p.pbits, f);
            if (!masked) { lb = pl.ilptr[q]; le = pl.ilptr[q + 1]; }
        }
    }
    double ta = 0.0, tb = 0.0;
    int n = 0;
    if (d >= 0 && lb + j < le) {
        const long x = pl.ilist[lb + j];
        const unsigned fx = pl.isdup[x];
        if (!(merge && (fx & 1))) {
            const XCell c = make_cell<WITH_EP>(rg, rg.exi[2 * x], q, rg.area[x], e);
            if (!(WITH_EP && c.range_error)) {
                GEnt ge;
                fa_group<WITH_EP>(rg, pl, s, true, merge != 0 && (fx & 128), c, x, ge);
                if (ge.n > 0) {
                    const int r = pl.aidx[x];
                    const int id0 = fa_gdense(g, r, ge.cls0, ge.gkey0);
                    const int id1 = ge.n < 2 ? 0 : fa_gdense(g, r, ge.cls1, ge.gkey1);
                    n = ge.n;
                    if (ge.n == 1 || id0 <= id1) { ta = ge.t0; tb = ge.t1; } else { ta = ge.t1; tb = ge.t0; }
                }
            }
        }
    }
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {                            // (every lane of the group forms the same sum; lane 0 stores it)
        const int nk = __shfl(n, base + k, 64);
        const double ak = __shfl(ta, base + k, 64), bk = __shfl(tb, base + k, 64);
        if (nk >= 1) sum = sum + ak;
        if (nk == 2) sum = sum + bk;
    }
    if (j == 0 && d >= 0) o.Mw[d] = sum;
}
template <bool WITH_EP>
__global__ __launch_bounds__(FA_T) void k_fa_psums8(RgView rg, PlanView pl, MatSpec s, FaG g, FaP p, int merge, long np, FaOut o) {
    __shared__ double s_hc[WITH_EP ? HC_LDS : 1];
    stage_hc<WITH_EP>(rg, s_hc);
    fa_psums8_body<WITH_EP>(rg, pl, s, g, p, merge, np, o, 0);
    fa_publish(o);
}
template <bool WITH_EP, bool EMIT, int T, int CPT>
__global__ __launch_bounds__(T, (T == 128 && !WITH_EP) ? 6 : 1) void k_fa_range(RgView rg, PlanView pl, MatSpec s, FaG g, FaP p, int merge, FaOut o,
                                                    uint32_t *__restrict__ flags) {
    __shared__ double s_hc[WITH_EP ? HC_LDS : 1];
    __shared__ int s_rank[FA_NC];
    __shared__ uint32_t s_start[FA_NC + 1], s_seg[FA_NC], s_nold[FA_NC], s_run[FA_NC], s_wcnt[T / 64][FA_NC];
    __shared__ double s_mul[FA_NC];
    __shared__ double s_chain[T / 64][64];
    // the values of a range that fits (its entries <= LCAP: every range of a small grid) stay in LDS from their placement to the
    // scaling: the sums and the scaling never read them back from memory
    constexpr int LCAP = T == 1024 ? 4096 : T == 256 ? 2048 : 1024;
    __shared__ __attribute__((aligned(16))) double s_vals[LCAP];
    __shared__ uint32_t s_no, s_w1[2][T / 64];
    __shared__ int o_cls[FA_OLDMAX], o_did[FA_OLDMAX];
    __shared__ double o_t[FA_OLDMAX];
    stage_hc<WITH_EP>(rg, s_hc);
    if (EMIT && o.psums_n > 0 && (int)blockIdx.x >= o.psums_first) {          // (uniform: this workgroup serves the straddling ice cells)
        fa_psums8_body<WITH_EP>(rg, pl, s, g, p, merge, o.psums_n, o, o.psums_first);
        fa_publish(o);
        return;
    }
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef FA_TIMELINE
    long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    FA_STAMP(0);
    const long x0 = pl.arng[r], x1 = pl.arng[r + 1];
    const int iA_r = rg.exi[2 * x0];                          // the atmosphere cell of the whole range
    const uint32_t eb = g.ebase[r];
    if (tid < FA_NC) {
        const bool in = tid < g.NC;
        const int rank = in ? (int)g.erank[(size_t)r * g.NC + tid] : -1;
        s_rank[tid] = rank;
        s_nold[tid] = in ? g.ecnto[(size_t)r * g.NC + tid] : 0;
        s_run[tid] = 0;
        s_seg[tid] = 0;
        if (tid == 0) s_no = 0;
    }
    __syncthreads();
    if (tid < g.NC && s_rank[tid] >= 0) s_seg[s_rank[tid]] = g.ecntn[(size_t)r * g.NC + tid] + g.ecnto[(size_t)r * g.NC + tid];
    __syncthreads();
    const int ncls = (int)g.r_ncls[r];
    if (tid == 0) {
        uint32_t a = 0;
        for (int q = 0; q < ncls; ++q) { s_start[q] = a; a += s_seg[q]; }
        s_start[ncls] = a;
    }
    __syncthreads();
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const bool g_is_row = EMIT || o.g_rows != 0;
    const bool fits = s_start[ncls] <= (uint32_t)LCAP;          // (uniform)
    // values go to memory from here only when nothing later writes them: rows that are scaled are written once, scaled; the
    // terms of the column sums (SUMS: o.val is scratch) are not needed in memory at all when the range is in LDS
    const bool wglobal = !fits || (EMIT && !o.scale);
    FA_STAMP(1);
    // CPT cells per thread and pass: their loads are issued in three staged rounds (exchange cell -> mask and
    // first-seen position of its ice cell -> dense id) before anything is consumed, so a pass pays the dependent
    // round trips once instead of once per 256 cells; the ranking then walks the CPT sub-chunks in x order.
    for (long cb = x0; cb < x1; cb += (long)T * CPT) {
        int iAv[CPT], iIv[CPT], ifv[CPT], didv[CPT];
        double av[CPT], ev[CPT];
        unsigned fl[CPT], pfv[CPT], pov[CPT];
        GEnt gev[CPT];
        bool newv[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const long x = cb + (long)u * T + tid;
            const long xx = x < x1 ? x : x1 - 1;
            iAv[u] = iA_r; iIv[u] = pl.exI[xx]; av[u] = rg.area[xx]; fl[u] = pl.isdup[xx];
            pfv[u] = 0; pov[u] = 0;
            // (ice keys: the first-occurrence flag is "first-seen cell of its ice cell (plan bit 2) and unmasked" -- no need to read
            // it back -- and the dense id is fetched in the third round, for unmasked cells only)
            if (p.fresh && p.key == KEY_X) { pfv[u] = p.pflag[xx]; pov[u] = p.poff[xx]; }
        }
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            ev[u] = rg.em[iIv[u]];
            const long x = cb + (long)u * T + tid;
            ifv[u] = (p.key == KEY_I && p.fresh) ? ((fl[u] & 4) ? (int)(x < x1 ? x : x1 - 1) : pl.ifirst[iIv[u]]) : 0;
        }
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const long x = cb + (long)u * T + tid;
            // dense id of the ice cell = poff[its first-seen position] (that is this cell itself for ~90 % of them)
            if (p.key == KEY_I) {
                const bool unm = !(ev[u] != ev[u]);
                if (p.fresh) {
                    const long at = (fl[u] & 4) ? (x < x1 ? x : x1 - 1) : (long)(ifv[u] < 0 ? 0 : ifv[u]);
                    didv[u] = unm ? (int)p.poff[at] : 0;              // (a masked cell has no entries and is no member: its id is never used)
                    if (EMIT && (fl[u] & 4) && unm && x < x1) { pfv[u] = 1; pov[u] = (unsigned)didv[u]; }
                } else didv[u] = iIv[u];
            } else didv[u] = p.fresh ? (int)pov[u] : (int)(x < x1 ? x : x1 - 1);
        }
#ifdef FA_TIMELINE
        if (cb == x0 && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tl[2] = (long long)__builtin_amdgcn_s_memtime(); }
#endif
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const long x = cb + (long)u * T + tid;
            const bool head = x < x1 && !(merge && (fl[u] & 1));
            GEnt &ge = gev[u];
            ge.n = 0; ge.cls0 = ge.cls1 = 0; ge.t0 = ge.t1 = 0.0; ge.gkey0 = ge.gkey1 = 0;
            bool member = false;                              // EMIT: the P key of this head cell is in the set numbered by this build
            if (head) {
                const XCell c = make_cell<WITH_EP>(rg, iAv[u], iIv[u], av[u], ev[u]);
                if (!(WITH_EP && c.range_error)) fa_group<WITH_EP>(rg, pl, s, g_is_row, merge != 0 && (fl[u] & 128), c, x, ge);
                member = p.fresh && (p.key == KEY_I ? (c.unmasked && ifv[u] >= 0) : pfv[u] != 0);
            }
            newv[u] = head && (p.key != KEY_I || !p.fresh || ifv[u] >= x0);
            if (EMIT) {
                // dims of a P set numbered by this build: the first-seen cell of every key records dense -> sparse
                if (x < x1 && pfv[u]) p.to_sparse[pov[u]] = p.key == KEY_I ? (long)iIv[u] : x;
                // Mw of an ice cell that lives in this range only (and of every exchange cell): its <= 2 entries, in
                // ascending row order, from zero (spsparse sum()).  Ice cells that straddle ranges: k_fa_pelem<SUMS>.
                if (ge.n > 0 && (p.key == KEY_X || (fl[u] & 2))) {
                    double sum = 0.0;
                    if (ge.n == 1) sum = sum + ge.t0;
                    else if (s_rank[ge.cls0] < s_rank[ge.cls1]) { sum = sum + ge.t0; sum = sum + ge.t1; }
                    else { sum = sum + ge.t1; sum = sum + ge.t0; }
                    o.Mw[didv[u]] = sum;
                } else if (ge.n == 0 && member && (p.key == KEY_X || (fl[u] & 2))) {
                    o.Mw[didv[u]] = 0.0;                      // a member without entries (Mw is not cleared beforehand when the set is fresh)
                }
            }
        }
        if (cb == x0) FA_STAMP(3);
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            if (cb + (long)u * T >= x1) break;          // uniform: this sub-chunk lies past the range
            const GEnt &ge = gev[u];
            const bool isnew = newv[u];
            const int did = didv[u];
            if (g.NC == 1) {
                // one class per range (A rows / A columns): the rank is a running count -- one ballot, two barriers
                const int par = u & 1;
                const bool hn = ge.n > 0 && isnew;
                const unsigned long long m = __ballot(hn);
                if (lane == 0) s_w1[par][wave] = (uint32_t)__popcll(m);
                __syncthreads();
                if (ge.n > 0) {
                    if (isnew) {
                        uint32_t before = s_run[0];
                        for (int w = 0; w < wave; ++w) before += s_w1[par][w];
                        const uint32_t pos = eb + s_nold[0] + before + (uint32_t)__popcll(m & lt);
                        if (EMIT) o.colind[pos] = did;
                        if (fits) s_vals[pos - eb] = ge.t0;
                        if (wglobal) o.val[pos] = ge.t0;
                    } else {
                        const uint32_t q = atomicAdd(&s_no, 1u);
                        if (q < (uint32_t)FA_OLDMAX) { o_cls[q] = 0; o_did[q] = did; o_t[q] = ge.t0; }
                    }
                }
                __syncthreads();
                if (tid == 0) { uint32_t a = 0; for (int w = 0; w < T / 64; ++w) a += s_w1[par][w]; s_run[0] += a; }
                continue;
            }
            // Rank of a new entry inside its class segment = entries of the same class at smaller x.  A cell holds a
            // class in at most one of its two slots, so the lanes of a wave are matched on both slots at once
            // (wave ballots, rs_scatter's scheme); every lane takes part in the ballots.
            for (int i = tid; i < (T / 64) * FA_NC; i += T) (&s_wcnt[0][0])[i] = 0;
            __syncthreads();
            const bool has0 = ge.n > 0, has1 = ge.n > 1;
            const int c0 = has0 ? ge.cls0 : 0, c1 = has1 ? ge.cls1 : 0;
            const unsigned long long v0 = __ballot(has0 && isnew), v1 = __ballot(has1 && isnew);
            unsigned long long pa0 = v0, pb0 = v1, pa1 = v0, pb1 = v1;      // pXj: lanes whose slot X holds the class of MY slot j
            for (int b = 0; b < 6; ++b) {
                const unsigned long long m0 = __ballot((c0 >> b) & 1), m1 = __ballot((c1 >> b) & 1);
                const bool b0 = (c0 >> b) & 1, b1 = (c1 >> b) & 1;
                pa0 &= b0 ? m0 : ~m0; pb0 &= b0 ? m1 : ~m1;
                pa1 &= b1 ? m0 : ~m0; pb1 &= b1 ? m1 : ~m1;
            }
            const unsigned long long u0 = pa0 | pb0, u1 = pa1 | pb1;
            const uint32_t in0 = (uint32_t)__popcll(u0 & lt), in1 = (uint32_t)__popcll(u1 & lt);
            if (has0 && isnew && in0 == 0) s_wcnt[wave][c0] = (uint32_t)__popcll(u0);
            if (has1 && isnew && in1 == 0) s_wcnt[wave][c1] = (uint32_t)__popcll(u1);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j >= ge.n) break;
                const int cls = ge.cls(j);
                if (isnew) {
                    uint32_t before = s_run[cls];
                    for (int w = 0; w < wave; ++w) before += s_wcnt[w][cls];
                    const uint32_t pos = eb + s_start[s_rank[cls]] + s_nold[cls] + before + (j ? in1 : in0);
                    if (EMIT) o.colind[pos] = did;
                    if (fits) s_vals[pos - eb] = ge.t(j);
                    if (wglobal) o.val[pos] = ge.t(j);
                } else {                                       // straddler: ranked after the stream
                    const uint32_t q = atomicAdd(&s_no, 1u);
                    if (q < (uint32_t)FA_OLDMAX) { o_cls[q] = cls; o_did[q] = did; o_t[q] = ge.t(j); }
                }
            }
            __syncthreads();
            if (tid < FA_NC) {
                uint32_t a = 0;
                for (int w = 0; w < T / 64; ++w) a += s_wcnt[w][tid];
                s_run[tid] += a;
            }
            __syncthreads();
        }
    }
    FA_STAMP(4);
    const uint32_t no = s_no < (uint32_t)FA_OLDMAX ? s_no : (uint32_t)FA_OLDMAX;
    if (s_no > (uint32_t)FA_OLDMAX && tid == 0) atomicOr(flags, (uint32_t)FA_ERR_OLDOVER);
    for (uint32_t e = tid; e < no; e += T) {            // inside a class the old ids are distinct: rank by counting
        const int cls = o_cls[e], did = o_did[e];
        uint32_t cnt = 0;
        for (uint32_t q = 0; q < no; ++q) cnt += (o_cls[q] == cls && o_did[q] < did) ? 1u : 0u;
        const uint32_t pos = eb + s_start[s_rank[cls]] + cnt;
        if (EMIT) o.colind[pos] = did;
        if (fits) s_vals[pos - eb] = o_t[e];
        if (wglobal) o.val[pos] = o_t[e];
    }
    __threadfence_block();
    __syncthreads();
    FA_STAMP(5);
    // Sequential sums (the order of spsparse sum()), one WAVE per segment: 64 values are loaded coalesced and the
    // chain is replayed from registers with v_readlane (every lane computes the identical sum), as k_seg_sums_wave
    // does; then the weights (k_weights) and, for the rows, the scaling.
    for (int q = wave; q < ncls; q += T / 64) {
        const uint32_t b = eb + s_start[q], e = eb + s_start[q + 1];
        double sum = 0.0;
        // The chain is replayed from LDS: the wave parks 64 values in its own 512 bytes and every lane reads them back one address
        // at a time (a broadcast read) -- one read and one dependent add per value where v_readlane took two scalar moves and an
        // add (25 cycles per value measured; the longest row of a 5 km range, ~1 100 values, was half of this kernel's time).
        // The next chunk's load is in flight while this one is summed.
        if (fits) {                                             // one lane walks the segment in LDS (5.5 cycles per term measured)
            if (lane == 0) {
                const uint32_t s0 = s_start[q], n = e - b;
                uint32_t k = 0;
                if ((s0 & 1u) && n) { sum = sum + s_vals[s0]; k = 1; }      // (then the rest is read 16 bytes at a time)
                const double2 *sv2 = reinterpret_cast<const double2 *>(__builtin_assume_aligned(s_vals + s0 + k, 16));
                for (; k + 64 <= n; k += 64, sv2 += 32) {
#pragma unroll
                    for (int j = 0; j < 32; ++j) { const double2 t2 = sv2[j]; sum = sum + t2.x; sum = sum + t2.y; }
                }
                for (const double *sv = reinterpret_cast<const double *>(sv2); k < n; ++k, ++sv) sum = sum + *sv;
            }
            sum = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sum)), __builtin_amdgcn_readfirstlane(__double2loint(sum)));
        }
        double *buf = s_chain[wave];
        double vnext = (!fits && b + lane < e) ? o.val[b + lane] : 0.0;
        for (uint32_t base = b; base < e && !fits; base += 64) {
            const double v = vnext;
            const uint32_t kn = base + 64 + lane;
            vnext = kn < e ? o.val[kn] : 0.0;
            buf[lane] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int cnt = (int)min(64u, e - base);
            if (cnt == 64) {
#pragma unroll
                for (int j = 0; j < 64; ++j) sum = sum + buf[j];
            } else {
                for (int j = 0; j < cnt; ++j) sum = sum + buf[j];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
            int cls = 0;
            for (int k = 0; k < g.NC; ++k) cls = s_rank[k] == q ? k : cls;
            const long iA = rg.exi[2 * x0];
            const long gkey = g.key == KEY_E ? iA * rg.sA + (long)cls * rg.sHC : iA;
            const int gd = fa_gdense(g, r, cls, gkey);
            if (gd < 0) continue;                               // a key the pre-populated set lacks: the build is discarded (FA_ERR_MISSING)
            if (g.to_sparse) g.to_sparse[gd] = gkey;            // a set numbered by this build: dense -> sparse
            if (g.tab_out) g.tab_out[gkey] = gd;
            if (EMIT) {                                        // FAM_AEVI rows (RegridMatrices_Dynamic.cpp:100-146)
                double wM, mul = 1.0;
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
                s_mul[q] = mul;
            } else {                                           // FAM_IVAE columns (:201-233)
                double Mw = sum;
                if (o.correctA) Mw = ratio_of(rg, g.key, gkey) * sum;
                o.Mw[gd] = Mw;
            }
        }
    }
    FA_STAMP(6);
    if (EMIT && r == pl.nAr - 1 && tid == 0) o.rowptr[g.gbase[pl.nAr]] = (int32_t)g.ebase[pl.nAr];      // rowptr[nrow] = nnz
    if (EMIT && o.scale) {
        __syncthreads();
        for (int q = 0; q < ncls; ++q) {
            const double mul = s_mul[q];
            if (fits) { for (uint32_t k = s_start[q] + tid; k < s_start[q + 1]; k += T) o.val[eb + k] = mul * s_vals[k]; }
            else for (uint32_t k = eb + s_start[q] + tid; k < eb + s_start[q + 1]; k += T) o.val[k] = mul * o.val[k];
        }
    }
    fa_publish(o);
#ifdef FA_TIMELINE
    if (tid == 0 && o.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tl[7] = (long long)__builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) o.dbg[(size_t)r * 8 + i] = tl[i];
    }
#endif
}

// ---- one thread per P element (ice cell or exchange cell): its <= 8 entries --------------------------
enum { FA_PSUMS = 0, FA_PCOUNT = 1, FA_PEMIT = 2 };
// ANYORDER: the G set may number the columns in some other order than this build would (a pre-populated dimE): the rows then
// take the per-row selection branch below.  A set numbered by this build cannot, and the kernel without that branch fits the
// 64 registers of full occupancy (a1h IvE 3.15 -> 3.04 ms, IvA 2.09 -> 1.94).
template <bool WITH_EP, int MODE, bool ANYORDER>
__device__ __forceinline__ void fa_pelem_body(RgView rg, const PlanView &pl, const MatSpec &s, const FaG &g, const FaP &p, int merge, long np, const FaOut &o,
                                              uint32_t *__restrict__ rowlen, uint32_t *__restrict__ flags, int only_multi) {
    __shared__ double s_hc[WITH_EP ? HC_LDS : 1];
#ifdef FA_TIMELINE
    const long wid_tl = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    auto PSTAMP = [&](int i) { if (MODE == FA_PEMIT && o.dbg) { const long long t = (long long)__builtin_amdgcn_s_memtime(); o.dbg[wid_tl * 8 + i] = t; } };
#else
    auto PSTAMP = [&](int) {};
#endif
    PSTAMP(0);
    stage_hc<WITH_EP>(rg, s_hc);
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= np) return;
    if (MODE == FA_PSUMS || only_multi) q = pl.mlist[q];      // only the ice cells with several exchange cells (the others: k_fa_range / k_sa_emit)
    // dense id of this element, -1: not a member of the set
    int d = -1;
    int lb = 0, le = 0;
    long pkey = q;
    double e = 0.0;
    if (p.key == KEY_I) {
        e = rg.em[q];
        const bool masked = e != e;
        if (masked && p.fresh) return;                      // not a member of the set: nothing to write
        const int f = pl.ifirst[q];
        if (only_multi && pl.sx1 > 0 && (f < pl.sx0 || f >= pl.sx1)) return;      // another rank's ice cell
        if (!p.fresh) d = (int)q;
        else if (f >= 0) d = (int)fa_poff_at(p.poff, p.Pw, p.pbits, f);
        if (!masked) { lb = pl.ilptr[q]; le = pl.ilptr[q + 1]; }      // a masked cell has no entries (identity dims: an empty row / column)
    } else {
        d = p.fresh ? (p.pflag[q] ? (int)p.poff[q] : -1) : (int)q;
        le = 1;
    }
    if (d < 0) return;
    (void)flags;
    PSTAMP(1);
    if (MODE == FA_PCOUNT && p.key == KEY_I && pl.icnt_pos) {
        // entries of an ice cell's row: one per group of duplicate exchange cells that contributes, times the elevation
        // classes its elevation lies between -- the groups are static (plan), the classes a function of the mask value
        uint32_t n = 0;
        if (lb < le) {                                        // (unmasked)
            if (WITH_EP) {
                const XCell c = make_cell<WITH_EP>(rg, 0, q, 1.0, e);
                n = c.range_error ? 0u : (uint32_t)c.nep * pl.icnt_nz[q];
            } else n = pl.icnt_pos[q];
        }
        rowlen[d] = n;
        return;
    }
    const bool g_is_row = MODE == FA_PSUMS;                  // SUMS: the matrix has G rows (AEvI); COUNT / EMIT: G columns (IvAE)
    const int hint = (WITH_EP && p.key == KEY_I) ? cell_hint(rg, e) : -1;      // (every cell of an ice cell has its elevation)
    // The element's entries are visited group by group (ascending x = ascending range = ascending G dense id when
    // the G set was numbered by this build; inside a group the two classes go by their rank).  No private arrays:
    // the cells are simply re-evaluated by every pass (they are in cache).
    //   pass(emit_at): emit_at < 0 -> returns count / sum / sortedness; >= 0 -> calls the writer for every entry
    int ne = 0;
    double sum = 0.0;
    bool sorted = true;
    int prev = -1;
    auto visit = [&](auto &&fn) {
        for (int k = lb; k < le; ++k) {
            const long x = p.key == KEY_I ? (long)pl.ilist[k] : q;
            const unsigned fx = pl.isdup[x];
            if (merge && (fx & 1)) continue;
            // COUNTING the entries of an ice cell without elevation classes needs nothing of the exchange cell but the sign
            // class of its area (plan bits 3 / 4): its ice cell is this element, the mask value is in hand -- no gather of
            // the cell's indices and area (IvA: the count pass reads the static lists only)
            // (the column sums of an ice cell -- SUMS, one-class matrices -- need the area itself, still no indices / mask gathers)
            const XCell c = (!WITH_EP && MODE == FA_PCOUNT && p.key == KEY_I)
                                ? make_cell<false>(rg, 0, q, (fx & 8) ? 1.0 : (fx & 16) ? -1.0 : 0.0, e)
                            : (!WITH_EP && MODE == FA_PSUMS && p.key == KEY_I && !g.tab)
                                ? make_cell<false>(rg, 0, q, rg.area[x], e)
                            : p.key == KEY_I                               // (the ice cell is this element, its mask value is in hand:
                                ? make_cell<WITH_EP>(rg, rg.exi[2 * x], q, rg.area[x], e, hint)      //  no gather behind the cell's indices, one class search)
                                : load_cell<WITH_EP>(rg, x);
            if (WITH_EP && c.range_error) continue;
            GEnt ge;
            fa_group<WITH_EP>(rg, pl, s, g_is_row, merge != 0 && (fx & 128), c, x, ge);      // (plan bit 7: a duplicate follows)
            if (ge.n == 0) continue;
            const int r = pl.aidx[x];
            const int id0 = MODE == FA_PCOUNT ? 0 : fa_gdense(g, r, ge.cls0, ge.gkey0);
            const int id1 = (MODE == FA_PCOUNT || ge.n < 2) ? 0 : fa_gdense(g, r, ge.cls1, ge.gkey1);
            if (ge.n == 1) fn(id0, ge.t0, c.iA);
            else if (id0 <= id1) { fn(id0, ge.t0, c.iA); fn(id1, ge.t1, c.iA); }
            else { fn(id1, ge.t1, c.iA); fn(id0, ge.t0, c.iA); }
        }
    };
    // EMIT with columns that ascend (known, or found so by the first visit): the first FA_KEEP entries of the row are parked in LDS (a private slot per thread, no
    // registers), so a row of that many entries -- an ice cell under one or two GCM cells: ~99 % of them -- is written after ONE
    // visit of its exchange cells instead of two (row sum first, scaled values second)
    const int b0 = MODE == FA_PEMIT ? o.rowptr[d] : 0;          // (asked for before the visit: the round trip overlaps it)
    // (elevation classes: two entries per exchange cell -- an ice cell under three GCM cells has six; the waves that held one were
    // the tail of this kernel at 5 km, 71 k cycles against a median of 21 k: they visited their cells twice)
#ifndef FA_KEEP_EP
#define FA_KEEP_EP 8
#endif
    constexpr int FA_KEEP = WITH_EP ? FA_KEEP_EP : 4;
    constexpr bool KEEP = MODE == FA_PEMIT;
    __shared__ double s_kt[KEEP ? FA_KEEP : 1][KEEP ? FA_T : 1];
    __shared__ int s_kid[KEEP ? FA_KEEP : 1][KEEP ? FA_T : 1], s_ka[KEEP ? FA_KEEP : 1][KEEP ? FA_T : 1];
    visit([&](int id, double t, long iA) {
        if (KEEP && ne < FA_KEEP) { s_kt[ne][threadIdx.x] = t; s_kid[ne][threadIdx.x] = id; s_ka[ne][threadIdx.x] = (int)iA; }
        ++ne; sum = sum + t; sorted = sorted && id >= prev; prev = id;
    });
    PSTAMP(2);
    if (MODE == FA_PCOUNT) { rowlen[d] = (uint32_t)ne; return; }
    if (MODE == FA_PSUMS) { o.Mw[d] = sum; return; }          // FAM_AEVI: Mw = colsum (:100); G numbered by this build: in order
    // FAM_IVAE rows: wM = rowsum, M = [1/wM] * T [* sApvA]
    if (p.fresh) p.to_sparse[d] = pkey;
    auto finish = [&](double t, long iA) {
        double v = t;
        if (o.scale) v = (1.0 / sum) * v;
        if (o.correctA) v = v * rg.ratioA[iA];
        return v;
    };
    if (KEEP && ne <= FA_KEEP && (sorted || !ANYORDER)) {
        o.wM[d] = sum;
        for (int a = 0; a < ne; ++a) {
            o.colind[b0 + a] = s_kid[a][threadIdx.x];
            o.val[b0 + a] = finish(s_kt[a][threadIdx.x], (long)s_ka[a][threadIdx.x]);
        }
    } else if (sorted || !ANYORDER) {
        o.wM[d] = sum;
        int a = 0;
        visit([&](int id, double t, long iA) { o.colind[b0 + a] = id; o.val[b0 + a] = finish(t, iA); ++a; });
    } else {
        // a pre-populated G set numbers the columns in some other order: ascending column by repeated selection, the
        // row sum taken in that order (spsparse sum() walks the row by ascending column)
        double rs = 0.0;
        int last = -1;
        for (int a = 0; a < ne; ++a) {
            int best = 0x7fffffff; double bt = 0.0;
            visit([&](int id, double t, long) { if (id > last && id < best) { best = id; bt = t; } });
            rs = rs + bt;
            last = best;
        }
        sum = rs;
        o.wM[d] = sum;
        last = -1;
        for (int a = 0; a < ne; ++a) {
            int best = 0x7fffffff; double bt = 0.0; long ba = 0;
            visit([&](int id, double t, long iA) { if (id > last && id < best) { best = id; bt = t; ba = iA; } });
            o.colind[b0 + a] = best; o.val[b0 + a] = finish(bt, ba);
            last = best;
        }
    }
    PSTAMP(3);
}
template <bool WITH_EP, int MODE, bool ANYORDER>
__global__ __launch_bounds__(FA_T, (WITH_EP && MODE == 2) ? (ANYORDER ? 1 : 6) : 8) void k_fa_pelem(RgView rg, PlanView pl, MatSpec s, FaG g, FaP p, int merge, long np, FaOut o,
                                                    uint32_t *__restrict__ rowlen, uint32_t *__restrict__ flags, int only_multi) {
    fa_pelem_body<WITH_EP, MODE, ANYORDER>(rg, pl, s, g, p, merge, np, o, rowlen, flags, only_multi);
    if (MODE == FA_PSUMS) fa_publish(o);
}
__global__ void k_fa_zero_identity(double *__restrict__ w, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = 0.0;
}

// ---- host side --------------------------------------------------------------------------------------
// Which dims shapes the fast path serves: a G set (A / E) that is fresh or -- E columns only -- pre-populated
// (the coupler's shared dimE); a P set (I / X) that is fresh or the identity over the whole extent.
static int fa_pset_mode(const ibh_sparse_set *set, int64_t extent) {       // 1 fresh, 0 identity-complete, -1 unsupported
    if (set->n == 0) return 1;
    if (set->identity && set->n == extent) return 0;
    return -1;
}

// sparse -> dense table of a pre-populated set on the device, cached in the set (valid while the set has not grown)
static const int32_t *set_inverse_table(ibh_sparse_set *set, int64_t extent, hipStream_t st) {
    if (set->tab_n == set->n && set->tab_extent == extent && set->tab.p) return set->tab.p;
    const int T = 256;
    set->tab.alloc((size_t)extent);
    if (set->dev_n < set->n) {                                  // complete the device copy of the dense -> sparse table first
        set->ensure_host();
        DevBuf<int64_t> grown((size_t)set->n);
        IBH_HIP(hipMemcpyAsync(grown.p, set->host.data(), sizeof(int64_t) * (size_t)set->n, hipMemcpyHostToDevice, st));
        IBH_HIP(hipStreamSynchronize(st));
        set->dev = std::move(grown);
        set->dev_n = set->n;
    }
    hipLaunchKernelGGL(k_fill_i32, dim3(ceil_div(extent, T)), dim3(T), 0, st, set->tab.p, (size_t)extent, -1);
    hipLaunchKernelGGL(k_scatter_existing, dim3(ceil_div(set->n, T)), dim3(T), 0, st, set->tab.p, set->dev.p, set->n);
    set->tab_n = set->n; set->tab_extent = extent;
    return set->tab.p;
}
__global__ void k_fa_init(uint32_t *cnt) {
    if (threadIdx.x < 8) cnt[threadIdx.x] = threadIdx.x == 0 ? 0xffffffffu : 0u;
}

// Workgroup shape of the two per-range kernels, by the size of the ranges.  One workgroup owns one range; what bounds it is
// latency (staged load rounds, ranking barriers, the sequential sums of one wave per class), so the chip wants many small
// workgroups when there are many small ranges and few large ones otherwise.  Measured (ms per build, AvI / EvI):
//   Antarctica 1 km x 1/2 deg (49 771 ranges of ~780 cells): 256 x 4 cells 1.53 / 2.10, 256 x 2 1.40 / 1.92, 128 x 2 1.26 / 1.74,
//                                                            128 x 4 1.33 / 1.87, 64 x 4 1.32 / 2.12, 64 x 2 1.38 / 2.22
//   1 km (553 ranges of ~7 900 cells):                       128 x 2 0.30 / 0.44, 256 x 4 0.24 / 0.32, 1024 x 4 0.34 / 0.37
//   5 km (122 ranges of ~1 500: fewer workgroups than CUs):  128 x 2 0.096 / 0.142, 256 x 4 0.082 / 0.109, 1024 x 4 0.076 / 0.089
// Round 5, the synthetic 5 km grid of the tests (252 ranges of ~720 cells; us per build AvI / EvI / IvE / XvE): 256 x 4 64.8 / 94.3 /
// 115.0 / 89.8, 1024 x 4 58.6 / 75.8 / 97.9 / 74.2, 1024 x 1 54.6 / 68.8 / 92.6 / 69.5; 20 km (60 ranges of ~240): 256 x 4 46.0 / 68.2 /
// 84.1 / 63.4, 1024 x 4 54.5 / 69.5 / 88.0 / 68.0, 1024 x 1 46.2 / 62.5 / 80.3 / 60.1 -- with fewer workgroups than the chip holds, one
// cell per thread: the per-thread loops over the cells of a pass are what a range's workgroup spends its time in.
// shape 0: 128 threads x 2 cells, 1: 256 x 4, 2: 1024 x 4, 3: 1024 x 1 (tuning "assemble_range_shape" overrides; scratch/asm_shapes.py)
static int fa_range_shape(long nX, int nAr) {
    const int forced = get_tuning("assemble_range_shape", -1);
    if (forced >= 0 && forced <= 3) return forced;
    const long mean = nX / (nAr > 0 ? nAr : 1);
    if (mean <= 1024 && nAr <= 512) return 3;           // every workgroup resident at once (two of 1024 threads per CU): a cell per thread
    if (mean <= 1024 && nAr >= 2048) return 0;
    if (mean > 1024 && nAr <= 256) return 2;            // at most one workgroup per CU: make it a big one
    return 1;
}
#define FA_LAUNCH_COUNT(EP, ...)                                                                                        \
    do {                                                                                                                \
        if (fa_shape == 0) hipLaunchKernelGGL((k_fa_count<EP, 128, 2>), dim3(nAr + count_extra), dim3(128), 0, st, __VA_ARGS__);      \
        else if (fa_shape == 2) hipLaunchKernelGGL((k_fa_count<EP, 1024, 4>), dim3(nAr + count_extra), dim3(1024), 0, st, __VA_ARGS__); \
        else if (fa_shape == 3) hipLaunchKernelGGL((k_fa_count<EP, 1024, 1>), dim3(nAr + count_extra), dim3(1024), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((k_fa_count<EP, 256, 4>), dim3(nAr + count_extra), dim3(256), 0, st, __VA_ARGS__);                    \
    } while (0)
#define FA_LAUNCH_RANGE(EP, EM, ...)                                                                                    \
    do {                                                                                                                \
        if (fa_shape == 0) hipLaunchKernelGGL((k_fa_range<EP, EM, 128, 2>), dim3(nAr + range_extra), dim3(128), 0, st, __VA_ARGS__);  \
        else if (fa_shape == 2) hipLaunchKernelGGL((k_fa_range<EP, EM, 1024, 4>), dim3(nAr + range_extra), dim3(1024), 0, st, __VA_ARGS__); \
        else if (fa_shape == 3) hipLaunchKernelGGL((k_fa_range<EP, EM, 1024, 1>), dim3(nAr + range_extra), dim3(1024), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((k_fa_range<EP, EM, 256, 4>), dim3(nAr + range_extra), dim3(256), 0, st, __VA_ARGS__);                \
    } while (0)

// ---- EvA / AvE (compute_EvA, RegridMatrices_Dynamic.cpp:254-332) -----------------------------------------------
// EpvAp = EpvG * diag(1 / rowsum(GvAp)) * GvAp is a product over the exchange cells: entry (e, a) is the sum over the
// cells x of atmosphere cell a that touch class e, in ascending x (Eigen walks column a of GvAp, whose rows are the
// exchange cells in first-seen = x order) -- i.e. per (range, class) exactly the sequential sum the column-sum pass of an
// X-row matrix computes (k_fa_range<SUMS> with every exchange cell its own key: all entries "new", placed in x order).
// So: k_fa_count numbers the classes, k_fa_range<SUMS> leaves T[e] = sum_x t(x, e) per dense class id, and one thread
// per range writes the (at most nhc)-entry CSR piece of its atmosphere cell plus the row / column sums; weights and
// scaling are the general pipeline's own kernels (k_weights, k_scale).  Both sets must be fresh; anything irregular
// (a class listed only by cells that contribute nothing -- negative areas) goes back to the general pipeline.
struct EvaOut {
    int32_t *rowptr, *colind, *row;
    double *val, *rs, *cs;
    int64_t *atable;
};
__global__ __launch_bounds__(1024) void k_eva_ascan(const uint32_t *__restrict__ nent, int nAr, uint32_t *__restrict__ abase, uint32_t *__restrict__ tot) {
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (int base = 0; base < nAr; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < nAr && nent[i] > 0) ? 1u : 0u;
        uint32_t t;
        const uint32_t ex = fa_block_excl_scan_1024(v, s_wave, t);
        if (i < nAr) abase[i] = carry + ex;
        carry += t;
    }
    if (threadIdx.x == 0) { abase[nAr] = carry; tot[0] = carry; }
}
__global__ void k_eva_hasa(const uint32_t *__restrict__ nent, int nAr, uint32_t *__restrict__ has) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nAr) has[i] = nent[i] > 0 ? 1u : 0u;
}
__global__ void k_eva_final(RgView rg, PlanView pl, FaG g, const uint32_t *__restrict__ abase, const double *__restrict__ S,
                            int e_is_row, EvaOut o, uint32_t *__restrict__ flags) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= pl.nAr) return;
    const uint32_t nE = g.gbase[pl.nAr];
    if (r == pl.nAr - 1) o.rowptr[e_is_row ? nE : abase[pl.nAr]] = (int32_t)nE;       // rowptr[nrow] = nnz (one entry per class)
    const uint32_t nc = g.r_ncls[r], ne = g.r_nent[r];
    if (nc == 0 && ne == 0) return;
    // (the tables the weights kernel dereferences are complete even when the build is about to be discarded)
    if (ne > 0) o.atable[abase[r]] = (int64_t)rg.exi[2 * (long)pl.arng[r]];
    bool ok = nc > 0 && ne > 0;
    for (int k = 0; k < g.NC; ++k)
        if (g.erank[(size_t)r * g.NC + k] >= 0 && g.ecntn[(size_t)r * g.NC + k] + g.ecnto[(size_t)r * g.NC + k] == 0) ok = false;
    if (!ok) { atomicOr(flags, (uint32_t)FA_ERR_MISSING); return; }
    const uint32_t a = abase[r], eb = g.gbase[r];
    double sum = 0.0;                                    // over the classes of this cell by ascending dense id (= first-seen rank)
    for (uint32_t q = 0; q < nc; ++q) {
        const uint32_t e = eb + q;
        const double t = S[e];
        sum = sum + t;
        o.val[e] = t;
        if (e_is_row) { o.rowptr[e] = (int32_t)e; o.colind[e] = (int32_t)a; o.row[e] = (int32_t)e; o.rs[e] = 0.0 + t; }
        else { o.colind[e] = (int32_t)e; o.row[e] = (int32_t)a; o.cs[e] = 0.0 + t; }
    }
    if (e_is_row) o.cs[a] = sum;
    else { o.rs[a] = sum; o.rowptr[a] = (int32_t)eb; }
}

static bool fast_build_eva(const ibh_regrid_matrices *rm, const MatSpec *sp, ibh_sparse_set *dims[2], int scale, int correctA,
                           const RgView &rg, ibh_weighted *w, hipStream_t st) {
    const ibh_regridder *gr = rm->rg;
    const bool e_is_row = sp->row_key == KEY_E;
    ibh_sparse_set *eset = dims[e_is_row ? 0 : 1], *aset = dims[e_is_row ? 1 : 0];
    if (eset->n != 0 || aset->n != 0 || eset == aset) return false;
    const int64_t extE = gr->nA * (int64_t)gr->nhc, extA = gr->nA;
    Arena &A = arena();
    A.reset();
    const ibh_plan &P = gr->plan;
    PlanView pl{P.arng.p, P.aidx.p, P.ilptr.p, P.ilist.p, P.ifirst.p, P.isdup.p, P.mlist.p, P.nAr, P.nmulti,
                get_tuning("assemble_static_count", 1) ? P.icnt_pos.p : nullptr, P.icnt_nz.p, P.exI.p};
    const long nX = gr->nX;
    const int T = FA_T, nAr = P.nAr;
    FaG g{};
    g.key = KEY_E; g.list = LIST_EP; g.NC = gr->nhc;
    const size_t nrc = (size_t)nAr * g.NC;
    g.erank = A.get<int8_t>(nrc); g.ecntn = A.get<uint32_t>(nrc); g.ecnto = A.get<uint32_t>(nrc);
    g.r_ncls = A.get<uint32_t>((size_t)nAr); g.r_nent = A.get<uint32_t>((size_t)nAr);
    g.gbase = A.get<uint32_t>((size_t)nAr + 1); g.ebase = A.get<uint32_t>((size_t)nAr + 1);
    uint32_t *abase = A.get<uint32_t>((size_t)nAr + 1);
    FaP p{};
    p.key = KEY_X; p.list = LIST_AP; p.fresh = 0;              // every exchange cell its own key: nothing merges, everything is "new"
    // counters: [0] first out-of-range cell, [1] fallback flags, [3] classes, [4] terms, [5] atmosphere cells
    uint32_t *d_cnt = A.get<uint32_t>(8);
    hipLaunchKernelGGL(k_fa_init, dim3(1), dim3(64), 0, st, d_cnt);
    const int fa_shape = fa_range_shape(nX, nAr);
    const int count_extra = 0;
    FA_LAUNCH_COUNT(true, rg, pl, *sp, g, p, e_is_row ? 1 : 0, 0, d_cnt, d_cnt + 1, 1, FaChain{});
    if (nAr > 4096 && fa_rscan_many(g, nAr, abase, d_cnt + 3, d_cnt + 5, st)) {
        // (three short launches for the three channels)
    } else if (nAr > 4096) {
        exclusive_scan_u32(g.r_ncls, g.gbase, (size_t)nAr, g.gbase + nAr, st);
        exclusive_scan_u32(g.r_nent, g.ebase, (size_t)nAr, g.ebase + nAr, st);
        uint32_t *has = A.get<uint32_t>((size_t)nAr);
        hipLaunchKernelGGL(k_eva_hasa, dim3(ceil_div(nAr, T)), dim3(T), 0, st, g.r_nent, nAr, has);
        exclusive_scan_u32(has, abase, (size_t)nAr, abase + nAr, st);
        IBH_HIP(hipMemcpyAsync(d_cnt + 3, g.gbase + nAr, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        IBH_HIP(hipMemcpyAsync(d_cnt + 4, g.ebase + nAr, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        IBH_HIP(hipMemcpyAsync(d_cnt + 5, abase + nAr, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(k_fa_rscan, dim3(1), dim3(1024), 0, st, g.r_ncls, g.r_nent, nAr, g.gbase, g.ebase, d_cnt + 3);
        hipLaunchKernelGGL(k_eva_ascan, dim3(1), dim3(1024), 0, st, g.r_nent, nAr, abase, d_cnt + 5);
    }
    IBH_HIP(hipGetLastError());
    uint32_t h[8];
    auto check_counters = [&]() -> bool {
        readback_sync(h, d_cnt, sizeof(h), st);
        if (h[0] != 0xffffffffu) {                              // message of linterp_1d_b, IceRegridder_L0.cpp:84-85
            int32_t ij[2];
            IBH_HIP(hipMemcpy(ij, gr->ex_indices.p + 2 * (size_t)h[0], sizeof(ij), hipMemcpyDeviceToHost));
            double e = 0;
            IBH_HIP(hipMemcpy(&e, rm->elevmaskI.p + ij[1], sizeof(double), hipMemcpyDeviceToHost));
            fail(IBH_ERANGE, "Elevation %g out of bounds (%g, %g)", e < 0 ? 0.0 : e, gr->hcdefs_h.front(), gr->hcdefs_h.back());
        }
        return h[1] == 0;
    };
    if (!check_counters()) return false;
    IBH_CHECK(h[4] < (1u << 31), "matrix too large for int32 indices");
    const int nE = (int)h[3], nA = (int)h[5];
    const uint32_t nterm = h[4];
    DevBuf<int64_t> etable, atable;
    etable.alloc((size_t)nE); atable.alloc((size_t)nA);
    g.to_sparse = etable.p;
    const int nrow = e_is_row ? nE : nA, ncol = e_is_row ? nA : nE;
    const long nnz = nE;
    w->nrow = nrow; w->ncol = ncol; w->nnz = nnz;
    w->rowptr.alloc((size_t)nrow + 1); w->colind.alloc((size_t)nnz); w->val.alloc((size_t)nnz);
    w->wM.alloc((size_t)nrow); w->Mw.alloc((size_t)ncol);
    uint32_t *flags = d_cnt + 1;
    double *S = A.get<double>((size_t)nE);
    double *rs = A.get<double>((size_t)nrow), *cs = A.get<double>((size_t)ncol);
    double *rowmul = A.get<double>((size_t)nrow), *colmul = A.get<double>((size_t)ncol);
    int32_t *row = A.get<int32_t>((size_t)nnz);
    if (nE == 0) IBH_HIP(hipMemsetAsync(w->rowptr.p, 0, sizeof(int32_t), st));
    if (nE) {
        // T[e]: the terms go to their (class, x-order) slot of a scratch array, one wave per class adds them up in sequence
        FaOut os{nullptr, nullptr, A.get<double>(nterm), nullptr, S, sp->family, 0, 0, e_is_row ? 1 : 0};
        const int range_extra = 0;
        FA_LAUNCH_RANGE(true, false, rg, pl, *sp, g, p, 0, os, flags);
        EvaOut eo{w->rowptr.p, w->colind.p, row, w->val.p, rs, cs, atable.p};
        hipLaunchKernelGGL(k_eva_final, dim3(ceil_div(nAr, T)), dim3(T), 0, st, rg, pl, g, abase, S, e_is_row ? 1 : 0, eo, flags);
        IBH_HIP(hipGetLastError());
        // the flags these two may raise are read BEFORE the weights and the scaling run: a discarded build leaves rows of
        // the piece-wise CSR unwritten, and those kernels index through it
        if (!check_counters()) return false;
        FinalizeArgs fa{sp->family, scale, correctA, sp->row_key, sp->col_key, nrow, ncol, e_is_row ? etable.p : atable.p,
                        e_is_row ? atable.p : etable.p, rs, cs, w->wM.p, w->Mw.p, rowmul, colmul};
        hipLaunchKernelGGL(k_weights, dim3(ceil_div(std::max(nrow, ncol), T)), dim3(T), 0, st, rg, fa);
        if (scale || correctA)
            hipLaunchKernelGGL(k_scale, dim3(ceil_div(nnz, T)), dim3(T), 0, st, row, w->colind.p, w->val.p, nnz, rowmul, colmul, scale ? 1 : 0, correctA ? 1 : 0);
        IBH_HIP(hipGetLastError());
    }
    auto commit = [&](ibh_sparse_set *set, int64_t extent, DevBuf<int64_t> &table, int n) {
        set->sparse_extent = extent;
        if (n == 0) return;
        set->host.clear(); set->host_n = 0; set->inv.clear(); set->inv_n = 0;
        set->dev = std::move(table);
        set->dev_n = set->n = n;
        set->identity = false;
    };
    commit(eset, extE, etable, nE);
    commit(aset, extA, atable, nA);
    IBH_HIP(hipStreamSynchronize(st));
    return true;
}

// Batch builds (assemble_batch): everything a concurrent fast build would otherwise create lazily in shared objects --
// the sheet's plan, the inverse table of a pre-populated column set -- is created up front on the calling thread.
static void fast_prewarm(const ibh_regrid_matrices *rm, const MatSpec *sp, ibh_sparse_set *dims[2], hipStream_t st) {
    const ibh_regridder *gr = rm->rg;
    if (!get_tuning("assemble_fast", 1) || !ensure_plan(gr, st)) return;
    if (sp->family != FAM_IVAE || !dims[1] || dims[1]->n == 0 || dims[1]->identity || sp->col_key != KEY_E) return;
    (void)set_inverse_table(dims[1], gr->nA * (int64_t)gr->nhc, st);
    IBH_HIP(hipStreamSynchronize(st));
}

static bool stream_build(const ibh_regrid_matrices *rm, const MatSpec *sp, ibh_sparse_set *dims[2], int scale, int correctA,
                         const RgView &rg, ibh_weighted *w, hipStream_t st, ibh_comm *comm);          // streamasm.inl
// the first exchange cell of every range on the host (the sharded build deals ranges to ranks), copied once per plan
// Per host thread (a thread's builds run one after the other on its stream): the status words of the chained scans and a ring of
// counter slots {0xffffffff, 0, ...}.  The words carry a 4-bit epoch, so they are cleared every 15 builds; the ring is
// initialised again when it has gone round (both stream-ordered, both amortised to nothing).
__global__ void k_fa_init_ring(uint32_t *cnt, int nslot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslot * 8) cnt[i] = (i & 7) == 0 ? 0xffffffffu : 0u;
}
struct FaChainState {
    static constexpr int NSLOT = 256;
    unsigned long long *status = nullptr;
    size_t cap = 0;
    unsigned epoch = 0;
    uint32_t *ring = nullptr;
    uint32_t *ticket = nullptr;         // device word, zero between builds (the last workgroup of a build's last kernel resets it)
    uint32_t *h_cnt = nullptr;          // pinned host memory: the counters as the last kernel of the build left them
    int next = NSLOT, device = -1;
    void bind() {
        int dev = 0;
        IBH_HIP(hipGetDevice(&dev));
        if (dev == device) return;
        // (buffers of another device are left to that device's context: a thread that switches devices is rare and they are small)
        status = nullptr; cap = 0; epoch = 0; ring = nullptr; ticket = nullptr; next = NSLOT; device = dev;
    }
    void publish_into(FaOut &o, const uint32_t *d_cnt, hipStream_t st) {
        bind();
        if (!ticket) {
            IBH_HIP(hipMalloc(&ticket, sizeof(uint32_t)));
            IBH_HIP(hipMemsetAsync(ticket, 0, sizeof(uint32_t), st));
        }
        if (!h_cnt) IBH_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_cnt), 8 * sizeof(uint32_t), hipHostMallocDefault));
        o.pub_ticket = ticket; o.pub_src = d_cnt; o.pub_dst = h_cnt;
    }
    uint32_t *counters(hipStream_t st) {
        bind();
        if (!ring) IBH_HIP(hipMalloc(&ring, sizeof(uint32_t) * 8 * NSLOT));
        if (next >= NSLOT) {
            hipLaunchKernelGGL(k_fa_init_ring, dim3(ceil_div(8 * NSLOT, 256)), dim3(256), 0, st, ring, (int)NSLOT);
            next = 0;
        }
        return ring + 8 * (size_t)(next++);
    }
    unsigned long long *words(size_t n, hipStream_t st) {
        bind();
        bool clear = false;
        if (cap < n) {
            if (status) (void)hipFree(status);
            cap = std::max<size_t>(n, 4096) * 2;
            IBH_HIP(hipMalloc(&status, cap * sizeof(unsigned long long)));
            clear = true;
            epoch = 0;
        }
        if (++epoch > 15) { epoch = 1; clear = true; }
        if (clear) IBH_HIP(hipMemsetAsync(status, 0, cap * sizeof(unsigned long long), st));
        return status;
    }
};
static FaChainState &fa_chain_state() { static thread_local FaChainState s; return s; }

static const std::vector<int32_t> &plan_arng_host(const ibh_regridder *g) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    ibh_plan &P = g->plan;
    if (P.arng_h.empty()) {
        P.arng_h.resize((size_t)P.nAr + 1);
        IBH_HIP(hipMemcpy(P.arng_h.data(), P.arng.p, sizeof(int32_t) * ((size_t)P.nAr + 1), hipMemcpyDeviceToHost));
    }
    return P.arng_h;
}
// returns false when the fast path does not apply (nothing has been touched: the caller runs the general pipeline)
static bool fast_build(const ibh_regrid_matrices *rm, const MatSpec *sp, ibh_sparse_set *dims[2], int scale, int correctA,
                       const RgView &rg, ibh_weighted *w, hipStream_t st, ibh_comm *comm = nullptr) {
    const ibh_regridder *gr = rm->rg;
    if (!get_tuning("assemble_fast", 1)) return false;
    if (!ensure_plan(gr, st)) return false;
    if (sp->family == FAM_EVA) return get_tuning("assemble_fast_eva", 1) && fast_build_eva(rm, sp, dims, scale, correctA, rg, w, st);
    // the sharded build; when it does not apply (the same on every rank: static conditions, exchanged flags) the ranks go on
    // to build the matrix redundantly
    if (comm && comm_world(comm) > 1 && stream_build(rm, sp, dims, scale, correctA, rg, w, st, comm)) return true;
    if (stream_build(rm, sp, dims, scale, correctA, rg, w, st, nullptr)) return true;       // large grids: the streamed build
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
    if (!g_fresh && (gset->identity || gkey != KEY_E || g_is_row)) return false;      // pre-populated: E columns only
    const bool uses_ep = sp->row_list == LIST_EP || sp->col_list == LIST_EP;
    const int merge = (sp->row_key != KEY_X && sp->col_key != KEY_X) ? 1 : 0;

    Arena &A = arena();
    A.reset();
    const ibh_plan &P = gr->plan;
    PlanView pl{P.arng.p, P.aidx.p, P.ilptr.p, P.ilist.p, P.ifirst.p, P.isdup.p, P.mlist.p, P.nAr, P.nmulti,
                get_tuning("assemble_static_count", 1) ? P.icnt_pos.p : nullptr, P.icnt_nz.p, P.exI.p};
    const long nX = gr->nX;
    const int T = FA_T, nAr = P.nAr;
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
    // counters read back with one sync: [0] first out-of-range cell, [1] fallback flags, [2] new P keys, [3] G classes, [4] entries
    const bool stream_count = !uses_ep && g.NC == 1 && !g.tab && get_tuning("assemble_stream_count", nX >= (1l << 20) ? 1 : 0);
    // small grids: the scans ride in the counting kernel (FaChain) and the counters come initialised from a ring of slots --
    // three launches fewer in a chain of seven to nine that is all a 5 km build costs
    const bool optimistic = nX <= (1l << 20) && get_tuning("assemble_optimistic", 1);
    const bool chained = optimistic && !stream_count && nrc < (1ul << FA_CH_NC) && get_tuning("assemble_chain", 1);
    FaChain chain{};
    uint32_t *d_cnt;
    uint32_t *rowlen_early = nullptr;
    const int fa_shape = fa_range_shape(nX, nAr);
    int count_extra = 0;                                        // workgroups of k_fa_count that count row lengths (FaChain::pc_*)
    if (chained) {
        FaChainState &cs = fa_chain_state();
        d_cnt = cs.counters(st);
        // (row lengths of an I-row build on the identity ice set: counted and scanned by extra workgroups of the counting launch)
        const bool pc = !g_is_row && pkey == KEY_I && !p.fresh && pl.icnt_pos && get_tuning("assemble_pcount_incount", 1);
        const long pc_rows = pc ? (long)extent_of(pkey) : 0;
        const int tshape_c = fa_shape == 0 ? 128 : fa_shape == 1 ? 256 : 1024;
        count_extra = pc ? ceil_div(pc_rows, tshape_c) : 0;
        const bool ir = !g_is_row && pkey == KEY_I && p.fresh && pl.icnt_pos && get_tuning("assemble_irows_incount", 1);
        chain = FaChain{cs.words((size_t)nAr + (size_t)count_extra + (ir ? (size_t)nAr : 0), st), cs.epoch, d_cnt + 2, nullptr, 0u, nullptr, 0u, nullptr, 0u,
                        nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0l, 0, 0};
        if (ir) {                                               // (I rows on an ice set numbered by this build: the row pointer from the counting launch)
            const size_t np_ub0 = (size_t)std::min<int64_t>(extent_of(pkey), nX);
            w->rowptr.alloc(np_ub0 + 1);
            chain.ir_rowptr = reinterpret_cast<uint32_t *>(w->rowptr.p);
            chain.ir_tmp = A.get<uint32_t>((size_t)nX);
            chain.ir_status0 = nAr + count_extra;
        }
        if (!g_is_row && pkey == KEY_X && !p.fresh && merge == 0 && get_tuning("assemble_xrows_incount", 1)) {
            // (X rows on the identity exchange-cell set: the row pointer falls out of the ranges' own look-back)
            w->rowptr.alloc((size_t)nX + 1);
            chain.xr_rowptr = reinterpret_cast<uint32_t *>(w->rowptr.p);
        }
        if (pc) {
            w->rowptr.alloc((size_t)pc_rows + 1);
            chain.pc_rowptr = reinterpret_cast<uint32_t *>(w->rowptr.p); chain.pc_n = pc_rows; chain.pc_first = nAr; chain.pc_status0 = nAr;
        }
        if (!g_is_row && !g_fresh && gset->n > 0) {             // (what k_fa_zero_identity did in a launch of its own)
            w->Mw.alloc((size_t)gset->n);
            chain.zero2 = reinterpret_cast<uint32_t *>(w->Mw.p); chain.nzero2 = 2u * (unsigned)gset->n;
        }
        // a fresh E set: the step's later builds (IvE, XvE on the dimE this EvI numbers: IceCoupler.cpp:361-377) look keys up in the
        // sparse -> dense table -- written here as the set is numbered instead of by two launches and a synchronisation later
        if (g_fresh && gkey == KEY_E && gext < (1l << 22) && get_tuning("assemble_tab_out", 1)) {
            gset->tab_n = -1;
            gset->tab.alloc((size_t)gext);
            chain.fill_m1 = gset->tab.p; chain.nfill = (unsigned)gext;
            g.tab_out = gset->tab.p;
        }
        // what the later kernels want cleared, by the upper bounds the outputs are allocated with below
        const size_t np_ub = p.fresh ? (size_t)std::min<int64_t>(extent_of(pkey), nX) : (size_t)extent_of(pkey);
        if (g_is_row && !p.fresh && np_ub) {                    // Mw of an identity P set: members without entries stay 0
            w->Mw.alloc(np_ub);
            chain.zero = reinterpret_cast<uint32_t *>(w->Mw.p); chain.nzero = (unsigned)(2 * np_ub);
        } else if (!g_is_row && p.fresh) {                      // row lengths beyond the real number of rows
            rowlen_early = A.get<uint32_t>(np_ub + 1);
            chain.zero = rowlen_early; chain.nzero = (unsigned)(np_ub + 1);
        }
    } else {
        d_cnt = A.get<uint32_t>(8);
        hipLaunchKernelGGL(k_fa_init, dim3(1), dim3(64), 0, st, d_cnt);
    }
    if (p.fresh) { p.pflag = A.get<uint8_t>((size_t)nX); p.poff = A.get<uint32_t>((size_t)nX); }
    const bool pcount_done = count_extra > 0 || chain.xr_rowptr != nullptr || chain.ir_rowptr != nullptr;      // (the row pointer is written by the counting launch)
    if (stream_count) {
        // one-class matrices of large grids: cells streamed, counts by integer atomics (three launches, all short)
        hipLaunchKernelGGL(k_fa_zero_counts, dim3(ceil_div(nAr, 256)), dim3(256), 0, st, g.r_ncls, g.ecntn, g.ecnto, nAr);
        hipLaunchKernelGGL(k_fa_count_stream<4>, dim3(ceil_div(nX, 1024)), dim3(256), 0, st, rg, pl, *sp, g, p, g_is_row ? 1 : 0, merge);
        hipLaunchKernelGGL(k_fa_count_fin, dim3(ceil_div(nAr, 256)), dim3(256), 0, st, g, nAr, d_cnt + 1);
    } else if (uses_ep) FA_LAUNCH_COUNT(true, rg, pl, *sp, g, p, g_is_row ? 1 : 0, merge, d_cnt, d_cnt + 1, 0, chain);
    else FA_LAUNCH_COUNT(false, rg, pl, *sp, g, p, g_is_row ? 1 : 0, merge, d_cnt, d_cnt + 1, 0, chain);
    if (chained) {
        // (poff, gbase / ebase and the three totals were written by k_fa_count)
    } else if (p.fresh) exclusive_scan_u8(p.pflag, p.poff, (size_t)nX, d_cnt + 2, st);
    if (chained) {
    } else if (nAr > 4096 && fa_rscan_many(g, nAr, nullptr, d_cnt + 3, nullptr, st)) {
        // (many ranges: three short launches for both channels)
    } else if (nAr > 4096) {                                    // very many: the device-wide scan; few: one workgroup, one launch
        exclusive_scan_u32(g.r_ncls, g.gbase, (size_t)nAr, g.gbase + nAr, st);
        exclusive_scan_u32(g.r_nent, g.ebase, (size_t)nAr, g.ebase + nAr, st);
        IBH_HIP(hipMemcpyAsync(d_cnt + 3, g.gbase + nAr, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        IBH_HIP(hipMemcpyAsync(d_cnt + 4, g.ebase + nAr, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(k_fa_rscan, dim3(1), dim3(1024), 0, st, g.r_ncls, g.r_nent, nAr, g.gbase, g.ebase, d_cnt + 3);
    }
    IBH_HIP(hipGetLastError());
    uint32_t h[8];
    bool published = false;
    auto check_counters = [&]() -> bool {
        if (published) {
            IBH_HIP(hipStreamSynchronize(st));
            std::memcpy(h, fa_chain_state().h_cnt, sizeof(h));
        } else readback_sync(h, d_cnt, sizeof(h), st);
        if (h[0] != 0xffffffffu) {                              // message of linterp_1d_b, IceRegridder_L0.cpp:84-85
            int32_t ij[2];
            IBH_HIP(hipMemcpy(ij, gr->ex_indices.p + 2 * (size_t)h[0], sizeof(ij), hipMemcpyDeviceToHost));
            double e = 0;
            IBH_HIP(hipMemcpy(&e, rm->elevmaskI.p + ij[1], sizeof(double), hipMemcpyDeviceToHost));
            fail(IBH_ERANGE, "Elevation %g out of bounds (%g, %g)", e < 0 ? 0.0 : e, gr->hcdefs_h.front(), gr->hcdefs_h.back());
        }
        if (h[1]) return false;                                 // a limit of the fast path was hit: general pipeline
        IBH_CHECK(h[4] < (1u << 31) && h[2] < (1u << 31), "matrix too large for int32 indices");
        return true;
    };
    // Small grids are bound by launches and synchronisations, not by bytes: their outputs are allocated by upper bounds
    // (<= 2 entries per exchange cell, <= NC rows per range) so that the counters are read back ONCE, with the final
    // synchronisation; large grids read them here and allocate exactly.
    if (!optimistic && !check_counters()) return false;
    const uint32_t n_pnew = optimistic ? (uint32_t)std::min<int64_t>(extent_of(pkey), nX) : h[2];
    const uint32_t n_g = optimistic ? (uint32_t)nrc : h[3];
    const uint32_t nnz = optimistic ? (uint32_t)((uses_ep ? 2 : 1) * nX) : h[4];
    int np_d = p.fresh ? (int)n_pnew : (int)extent_of(pkey);
    int ng_d = g_fresh ? (int)n_g : gset->n;
    // dims tables of the sets numbered by this build are written straight into the sets' device buffers
    DevBuf<int64_t> ptable, gtable;
    if (p.fresh) { ptable.alloc((size_t)np_d); p.to_sparse = ptable.p; }
    if (g_fresh) { gtable.alloc((size_t)ng_d); g.to_sparse = gtable.p; }
    const int nrow = g_is_row ? ng_d : np_d, ncol = g_is_row ? np_d : ng_d;
    w->nrow = nrow; w->ncol = ncol; w->nnz = nnz;
    w->rowptr.alloc((size_t)nrow + 1); w->colind.alloc(nnz); w->val.alloc(nnz);
    w->wM.alloc((size_t)nrow); w->Mw.alloc((size_t)ncol);
    FaOut o{w->rowptr.p, w->colind.p, w->val.p, w->wM.p, w->Mw.p, sp->family, scale, correctA, 0};
#ifdef FA_TIMELINE
    o.dbg = reinterpret_cast<long long *>(((unsigned long long)(unsigned)get_tuning("gt_dbg_hi", 0) << 32) | (unsigned)get_tuning("gt_dbg_lo", 0));
#endif
    uint32_t *flags = d_cnt + 1;
    const bool can_publish = chained && get_tuning("assemble_publish", 1);     // (chained => optimistic: the counters are read once, at the end)
    int range_extra = 0;                                        // workgroups of k_fa_range that do k_fa_psums8's work
    const long np_s = p.key == KEY_I ? gr->nI : nX;             // P elements by sparse index
    const dim3 gp(ceil_div(np_s, T));
    if (g_is_row) {
        // rows = G: CSR + wM + (most of) Mw from the ranges; members of the P set without entries keep Mw = 0
        // (a set numbered by this build: every member's Mw is written by k_fa_range / k_fa_pelem<SUMS>; an identity set has
        // non-members -- masked cells -- whose Mw stays 0)
        if (ncol && !p.fresh && !chain.zero) IBH_HIP(hipMemsetAsync(w->Mw.p, 0, sizeof(double) * (size_t)ncol, st));
        bool sums_after = p.key == KEY_I && P.nmulti;
        FaOut ol = o;                                         // (the build's last kernel publishes the counters)
        if (can_publish) { fa_chain_state().publish_into(ol, d_cnt, st); published = true; }
        if (sums_after && chained && P.nmulti <= (1l << 20) && get_tuning("assemble_psums8", 1) && get_tuning("assemble_psums_inrange", 1)) {
            // small grids: the straddlers' Mw in extra workgroups of the range kernel's own launch
            const int tshape = fa_shape == 0 ? 128 : fa_shape == 1 ? 256 : 1024;
            range_extra = ceil_div(8 * (long)P.nmulti, tshape);
            ol.psums_first = nAr; ol.psums_n = (long)P.nmulti;
            sums_after = false;
        }
        if (uses_ep) FA_LAUNCH_RANGE(true, true, rg, pl, *sp, g, p, merge, (sums_after ? o : ol), flags);
        else FA_LAUNCH_RANGE(false, true, rg, pl, *sp, g, p, merge, (sums_after ? o : ol), flags);
        range_extra = 0;
        if (sums_after) {                                     // Mw of the ice cells that straddle ranges (a few %)
            const dim3 gm(ceil_div(P.nmulti, T));
            if (P.nmulti <= (1l << 20) && get_tuning("assemble_psums8", 1)) {     // few of them: eight lanes per ice cell
                const dim3 g8(ceil_div(8 * (long)P.nmulti, T));
                if (uses_ep) hipLaunchKernelGGL((k_fa_psums8<true>), g8, dim3(T), 0, st, rg, pl, *sp, g, p, merge, (long)P.nmulti, ol);
                else hipLaunchKernelGGL((k_fa_psums8<false>), g8, dim3(T), 0, st, rg, pl, *sp, g, p, merge, (long)P.nmulti, ol);
            } else if (uses_ep) hipLaunchKernelGGL((k_fa_pelem<true, FA_PSUMS, false>), gm, dim3(T), 0, st, rg, pl, *sp, g, p, merge, (long)P.nmulti, ol, (uint32_t *)nullptr, flags, 0);
            else hipLaunchKernelGGL((k_fa_pelem<false, FA_PSUMS, false>), gm, dim3(T), 0, st, rg, pl, *sp, g, p, merge, (long)P.nmulti, ol, (uint32_t *)nullptr, flags, 0);
        }
    } else {
        // rows = P: count -> scan -> emit per element; Mw (G columns) from the ranges through a scratch copy of the terms
        uint32_t *rowlen = rowlen_early ? rowlen_early : A.get<uint32_t>((size_t)nrow + 1);
        if (optimistic && p.fresh && !rowlen_early) IBH_HIP(hipMemsetAsync(rowlen, 0, sizeof(uint32_t) * ((size_t)nrow + 1), st));   // rows beyond the real count
        if (pcount_done) {
            // (the row lengths were counted by extra workgroups of k_fa_count's launch)
        } else if (uses_ep) hipLaunchKernelGGL((k_fa_pelem<true, FA_PCOUNT, false>), gp, dim3(T), 0, st, rg, pl, *sp, g, p, merge, np_s, o, rowlen, flags, 0);
        else hipLaunchKernelGGL((k_fa_pelem<false, FA_PCOUNT, false>), gp, dim3(T), 0, st, rg, pl, *sp, g, p, merge, np_s, o, rowlen, flags, 0);
        if (!pcount_done) exclusive_scan_u32(rowlen, reinterpret_cast<uint32_t *>(w->rowptr.p), (size_t)nrow, reinterpret_cast<uint32_t *>(w->rowptr.p) + nrow, st);
        // (a pre-populated G set may list the columns in another order: the variant with the per-row selection branch)
        if (!g_fresh) {
            if (uses_ep) hipLaunchKernelGGL((k_fa_pelem<true, FA_PEMIT, true>), gp, dim3(T), 0, st, rg, pl, *sp, g, p, merge, np_s, o, rowlen, flags, 0);
            else hipLaunchKernelGGL((k_fa_pelem<false, FA_PEMIT, true>), gp, dim3(T), 0, st, rg, pl, *sp, g, p, merge, np_s, o, rowlen, flags, 0);
        } else {
            if (uses_ep) hipLaunchKernelGGL((k_fa_pelem<true, FA_PEMIT, false>), gp, dim3(T), 0, st, rg, pl, *sp, g, p, merge, np_s, o, rowlen, flags, 0);
            else hipLaunchKernelGGL((k_fa_pelem<false, FA_PEMIT, false>), gp, dim3(T), 0, st, rg, pl, *sp, g, p, merge, np_s, o, rowlen, flags, 0);
        }
        FaOut os = o;
        os.val = A.get<double>(nnz);
        if (can_publish) { fa_chain_state().publish_into(os, d_cnt, st); published = true; }
        if (!g_fresh && ncol && !chain.zero2) hipLaunchKernelGGL(k_fa_zero_identity, dim3(ceil_div(ncol, T)), dim3(T), 0, st, w->Mw.p, (long)ncol);   // columns of the shared set this mask does not touch
        if (uses_ep) FA_LAUNCH_RANGE(true, false, rg, pl, *sp, g, p, merge, os, flags);
        else FA_LAUNCH_RANGE(false, false, rg, pl, *sp, g, p, merge, os, flags);
    }
    IBH_HIP(hipGetLastError());
    if (optimistic) {                                           // the one read-back of a small build: flags and the real sizes
        if (!check_counters()) return false;
        if (p.fresh) np_d = (int)h[2];
        if (g_fresh) ng_d = (int)h[3];
        w->nrow = g_is_row ? ng_d : np_d; w->ncol = g_is_row ? np_d : ng_d; w->nnz = h[4];
        w->rowptr.n = (size_t)w->nrow + 1; w->colind.n = w->val.n = (size_t)w->nnz;
        w->wM.n = (size_t)w->nrow; w->Mw.n = (size_t)w->ncol;
        ptable.n = p.fresh ? (size_t)np_d : 0; gtable.n = g_fresh ? (size_t)ng_d : 0;
    }
    auto commit = [&](ibh_sparse_set *set, int64_t extent, DevBuf<int64_t> &table, int n) {
        set->sparse_extent = extent;
        if (n == 0) return;
        set->host.clear(); set->host_n = 0; set->inv.clear(); set->inv_n = 0;
        set->dev = std::move(table);
        set->dev_n = set->n = n;
        set->identity = false;
    };
    // (a set that is only read -- identity / pre-populated -- may be shared by builds running concurrently in a batch:
    // its extent is written only when it changes)
    if (p.fresh) commit(pset, extent_of(pkey), ptable, np_d); else if (pset->sparse_extent != extent_of(pkey)) pset->sparse_extent = extent_of(pkey);
    if (g_fresh) commit(gset, gext, gtable, ng_d); else if (gset->sparse_extent != gext) gset->sparse_extent = gext;
    if (g.tab_out) { gset->tab_n = gset->n; gset->tab_extent = gext; }
    IBH_HIP(hipStreamSynchronize(st));
    return true;
}
