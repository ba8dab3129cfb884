// comm.hip -- field-sharded applies across the GPUs of one node, behind the C-ABI.
//
// One process per GPU; rank r owns the fields [r*nvar_local, (r+1)*nvar_local) of a multi-field regrid (contiguous in the
// field-major arrays), the CSR is replicated, the SpMM needs no communication, and the results are reassembled on every
// rank.  The reference has no counterpart: it gathers everything to MPI rank 0 and regrids there
// (modele/GCMCoupler_ModelE.cpp:764-792).
//
// The exchange is DIRECT: every rank sends its block of a field group to each peer and receives the peers' blocks into
// their final places in the caller's result array (grouped ncclSend / ncclRecv) -- on MI355X xGMI is a full mesh of
// point-to-point links (7 x ~153 GB/s per GPU), so seven concurrent peer transfers use all links at once, and no ring
// collective has to walk one link at a time; in-place (no staging copy) for any row range of the result.  It runs on a stream
// the communicator owns: the SpMM of field block b+1 (and of the next apply) overlaps the exchange of block b.
// RCCL is loaded at run time (dlopen): a box without librccl still loads libicebin_hip.so; world 1 needs no library at all.
// A caller may also plug in its own transport (ibh_comm_create_custom: the tests run two ranks over a host-staged one).
#include <dlfcn.h>

#include <algorithm>
#include <memory>

#include <mutex>
#include <vector>

#include "common.h"

namespace ibh {
namespace {
struct nccl_uid { char internal[IBH_UNIQUE_ID_BYTES]; };
typedef void *nccl_comm_t;
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};
constexpr int kNcclFloat64 = 8;      // ncclDataType_t (rccl.h: ncclFloat64 = 8)
constexpr int kNcclInt8 = 0;         //                 (ncclInt8 = ncclChar = 0)

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy already in the process first (a PyTorch process has loaded its bundled librccl: one RCCL per process).
        // ICEBIN_RCCL_LIB, when set, is the ONLY candidate: a caller who names a library gets that one or an error.
        const char *env = getenv("ICEBIN_RCCL_LIB");
        const bool pinned = env && *env;
        const char *names[] = {env, pinned ? nullptr : "librccl.so", pinned ? nullptr : "librccl.so.1", pinned ? nullptr : "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (r.h) break;
        }
        for (const char *n : names) {
            if (r.h) break;
            if (!n || !*n) continue;
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.h) {
            const char *e = dlerror();           // (dlerror() clears the message it returns: read it once)
            r.err = std::string("librccl not found (") + (e ? e : "?") + "); set ICEBIN_RCCL_LIB";
            return;
        }
        auto sym = [&](const char *name) { void *p = dlsym(r.h, name); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + name; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}
Rccl &need_rccl() {
    Rccl &r = rccl();
    if (!r.err.empty()) fail(IBH_ENOTIMPL, "multi-GPU exchange needs RCCL: %s", r.err.c_str());
    return r;
}
#define IBH_NCCL(expr)                                                                                             \
    do {                                                                                                           \
        const int e_ = (expr);                                                                                     \
        if (e_ != 0) ::ibh::fail(IBH_EHIP, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(e_) : "?"); \
    } while (0)
}  // namespace
}  // namespace ibh

struct ibh_comm {
    int world = 1, rank = 0, device = 0;
    ibh::nccl_comm_t nccl = nullptr;
    ibh_exchange_fn custom = nullptr;
    ibh_gatherv_fn custom_v = nullptr;       // pieces of unequal size (the sharded assembly); nullptr: the transport cannot carry them
    void *custom_user = nullptr;
    hipStream_t xs = nullptr;            // the exchange stream
    bool owns_xs = true;                 // false: the caller's (ibh_comm_set_stream) -- never destroyed here
    bool planes_padded = false;          // ibh_comm_set_option("planes_padded"): the gap [nrow, ldb) of a result plane belongs to the call
    hipEvent_t ready = nullptr;
    // exchanges in flight: the byte range of the result array each one writes and the event behind it.  A later apply whose
    // results overlap such a range is ordered behind THAT exchange only -- so with two alternating result arrays the SpMM of
    // apply k+1 overlaps the exchange of apply k.
    static constexpr int RING = 8;
    // A flight stays live until its ring slot is reused: ANY stream that later touches the range is ordered behind it, not
    // only the first one (`waited` remembers the last stream that did, which need not wait twice: stream order).
    struct Flight { const char *lo = nullptr, *hi = nullptr; hipEvent_t ev = nullptr; bool live = false; bool has_waiter = false; hipStream_t waited = nullptr; } ring[RING];
    int next = 0;
};

using namespace ibh;

static void comm_common_init(ibh_comm *c) {
    IBH_HIP(hipGetDevice(&c->device));
    IBH_HIP(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
    IBH_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    for (auto &f : c->ring) IBH_HIP(hipEventCreateWithFlags(&f.ev, hipEventDisableTiming));
}
// the SpMM about to write [lo, hi) on stream st must not overtake an exchange that still reads / writes that range
static void order_behind_flights(ibh_comm *c, const void *lo, const void *hi, hipStream_t st) {
    for (auto &f : c->ring)
        if (f.live && static_cast<const char *>(lo) < f.hi && f.lo < static_cast<const char *>(hi) && !(f.has_waiter && f.waited == st)) {
            IBH_HIP(hipStreamWaitEvent(st, f.ev, 0));
            f.has_waiter = true; f.waited = st;
        }
}
// the exchange just enqueued on c->xs covers [lo, hi)
static void record_flight(ibh_comm *c, const void *lo, const void *hi, hipStream_t st) {
    ibh_comm::Flight &f = c->ring[c->next];
    c->next = (c->next + 1) % ibh_comm::RING;
    // the slot is reused.  If its old exchange has not finished, the new flight takes over its range as well (the hull): the
    // exchange stream runs in order, so whoever waits for the new event has waited for the old one too -- no stream loses a
    // dependency, at the price of a wider range while eight exchanges are still running
    const char *nlo = static_cast<const char *>(lo), *nhi = static_cast<const char *>(hi);
    if (f.live && hipEventQuery(f.ev) != hipSuccess) { nlo = std::min(nlo, f.lo); nhi = std::max(nhi, f.hi); }
    (void)st;
    IBH_HIP(hipEventRecord(f.ev, c->xs));
    lo = nlo; hi = nhi;
    f.lo = static_cast<const char *>(lo); f.hi = static_cast<const char *>(hi); f.live = true; f.has_waiter = false; f.waited = nullptr;
}

// for every base of the list: every rank's block of `nplane` planes (nrow doubles each, ldb apart) at base + rank*stride
// reaches base + rank*stride on every peer; enqueued on the exchange stream as ONE group (one RCCL launch however many
// results travel).  Rows [nrow, ldb) of a plane travel along only when the caller has SAID that they are padding it owns
// (ibh_comm_set_option "planes_padded": the library's own wrappers round planes up to 512 bytes); otherwise ldb is a true
// leading dimension (a column view of a larger array whose other columns are live data) and every plane travels by itself:
// nothing outside [0, nrow) of a plane is touched on any rank.  No guessing from the size of the gap.
static void exchange(ibh_comm *c, double *const *d_bases, int nbase, int nplane, int64_t nrow, int64_t ldb, int64_t stride) {
    if (c->world == 1 || nplane <= 0 || nrow <= 0 || nbase == 0) return;
    const bool whole = ldb == nrow || c->planes_padded || nplane == 1;
    const int npiece = whole ? 1 : nplane;
    const int64_t count = whole ? (int64_t)(nplane - 1) * ldb + nrow : nrow;
    if (c->custom) {
        for (int q = 0; q < nbase; ++q)
            for (int p = 0; p < npiece; ++p) {
                const int rc = c->custom(c->custom_user, d_bases[q] + (int64_t)p * ldb, count, stride, c->world, c->rank, c->xs);
                if (rc != 0) fail(IBH_EHIP, "custom exchange callback failed (%d)", rc);
            }
        return;
    }
    Rccl &r = need_rccl();
    IBH_NCCL(r.GroupStart());
    for (int k = 1; k < c->world; ++k) {             // peers in a rotated order: at step k everybody talks to a different partner
        const int to = (c->rank + k) % c->world, from = (c->rank - k + c->world) % c->world;
        for (int q = 0; q < nbase; ++q)
            for (int p = 0; p < npiece; ++p) {
                double *b = d_bases[q] + (int64_t)p * ldb;
                IBH_NCCL(r.Send(b + (int64_t)c->rank * stride, (size_t)count, kNcclFloat64, to, c->nccl, c->xs));
                IBH_NCCL(r.Recv(b + (int64_t)from * stride, (size_t)count, kNcclFloat64, from, c->nccl, c->xs));
            }
    }
    IBH_NCCL(r.GroupEnd());
}

// ---- pieces of the sharded ASSEMBLY (streamasm.inl): equal small blocks, and all-gathers of pieces of unequal size ------------------
namespace ibh {
int comm_world(const ibh_comm *c) { return c->world; }
int comm_rank(const ibh_comm *c) { return c->rank; }
// every rank's `count` doubles at base + rank*stride -> the same place on every peer, on `st`
void comm_exchange_blocks(ibh_comm *c, double *base, int64_t count, int64_t stride, hipStream_t st) {
    if (c->world == 1 || count == 0) return;
    if (c->custom) {
        const int rc = c->custom(c->custom_user, base, count, stride, c->world, c->rank, st);
        if (rc != 0) fail(IBH_EHIP, "custom exchange callback failed (%d)", rc);
        return;
    }
    Rccl &r = need_rccl();
    IBH_NCCL(r.GroupStart());
    for (int k = 1; k < c->world; ++k) {
        const int to = (c->rank + k) % c->world, from = (c->rank - k + c->world) % c->world;
        IBH_NCCL(r.Send(base + (int64_t)c->rank * stride, (size_t)count, kNcclFloat64, to, c->nccl, st));
        IBH_NCCL(r.Recv(base + (int64_t)from * stride, (size_t)count, kNcclFloat64, from, c->nccl, st));
    }
    IBH_NCCL(r.GroupEnd());
}
// n arrays; of array i rank q owns the bytes [offs[i][q], offs[i][q+1]) and every rank ends up with all of them, in place; one
// group (one RCCL launch) for all arrays, on `st`
void comm_gatherv(ibh_comm *c, int n, void *const *bases, const int64_t *const *offs, hipStream_t st) {
    if (c->world == 1 || n == 0) return;
    if (c->custom) {
        if (!c->custom_v) fail(IBH_ENOTIMPL, "this custom transport carries equal blocks only (ibh_comm_set_custom_gatherv)");
        for (int i = 0; i < n; ++i) {
            const int rc = c->custom_v(c->custom_user, bases[i], offs[i], c->world, c->rank, st);
            if (rc != 0) fail(IBH_EHIP, "custom gatherv callback failed (%d)", rc);
        }
        return;
    }
    Rccl &r = need_rccl();
    IBH_NCCL(r.GroupStart());
    for (int k = 1; k < c->world; ++k) {
        const int to = (c->rank + k) % c->world, from = (c->rank - k + c->world) % c->world;
        for (int i = 0; i < n; ++i) {
            char *b = static_cast<char *>(bases[i]);
            const int64_t mine = offs[i][c->rank + 1] - offs[i][c->rank], theirs = offs[i][from + 1] - offs[i][from];
            if (mine > 0) IBH_NCCL(r.Send(b + offs[i][c->rank], (size_t)mine, kNcclInt8, to, c->nccl, st));
            if (theirs > 0) IBH_NCCL(r.Recv(b + offs[i][from], (size_t)theirs, kNcclInt8, from, c->nccl, st));
        }
    }
    IBH_NCCL(r.GroupEnd());
}
}  // namespace ibh

extern "C" {

int ibh_comm_set_custom_gatherv(ibh_comm *c, ibh_gatherv_fn fn) {
    return guarded([&] {
        IBH_CHECK(c != nullptr && c->custom != nullptr, "not a communicator over a custom transport");
        c->custom_v = fn;
    });
}

int ibh_comm_unique_id(char id[IBH_UNIQUE_ID_BYTES]) {
    return guarded([&] {
        IBH_CHECK(id != nullptr, "null argument");
        nccl_uid u;
        IBH_NCCL(need_rccl().GetUniqueId(&u));
        memcpy(id, u.internal, IBH_UNIQUE_ID_BYTES);
    });
}

int ibh_comm_create(int world, int rank, const char id[IBH_UNIQUE_ID_BYTES], ibh_comm **out) {
    return guarded([&] {
        IBH_CHECK(out != nullptr && world >= 1 && rank >= 0 && rank < world, "bad communicator shape (world %d, rank %d)", world, rank);
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(IBH_ENODEVICE, "no HIP device available; libicebin_hip has no CPU fallback");
        std::unique_ptr<ibh_comm> c(new ibh_comm);
        c->world = world; c->rank = rank;
        comm_common_init(c.get());
        IBH_CHECK(world == 1 || id != nullptr, "a communicator of %d ranks needs the unique id rank 0 obtained from ibh_comm_unique_id", world);
        if (id != nullptr) {                        // (also at world 1 when an id is given: one rank is a valid RCCL communicator)
            nccl_uid u;
            memcpy(u.internal, id, IBH_UNIQUE_ID_BYTES);
            IBH_NCCL(need_rccl().CommInitRank(&c->nccl, world, u, rank));
        }
        *out = c.release();
    });
}

int ibh_comm_create_custom(int world, int rank, ibh_exchange_fn fn, void *user, ibh_comm **out) {
    return guarded([&] {
        IBH_CHECK(out != nullptr && world >= 1 && rank >= 0 && rank < world && fn != nullptr, "bad arguments");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(IBH_ENODEVICE, "no HIP device available; libicebin_hip has no CPU fallback");
        std::unique_ptr<ibh_comm> c(new ibh_comm);
        c->world = world; c->rank = rank; c->custom = fn; c->custom_user = user;
        comm_common_init(c.get());
        *out = c.release();
    });
}

int ibh_comm_destroy(ibh_comm *c) {
    return guarded([&] {
        if (!c) return;
        if (c->xs) (void)hipStreamSynchronize(c->xs);
        if (c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl);
        if (c->ready) (void)hipEventDestroy(c->ready);
        for (auto &f : c->ring) if (f.ev) (void)hipEventDestroy(f.ev);
        if (c->xs && c->owns_xs) (void)hipStreamDestroy(c->xs);      // (a caller's stream, ibh_comm_set_stream, stays the caller's)
        delete c;
    });
}

int ibh_comm_set_stream(ibh_comm *c, void *stream) {
    return guarded([&] {
        IBH_CHECK(c != nullptr && stream != nullptr, "null argument");
        if (c->xs) IBH_HIP(hipStreamSynchronize(c->xs));           // nothing of ours is left on the old stream
        if (c->xs && c->owns_xs) IBH_HIP(hipStreamDestroy(c->xs));
        c->xs = static_cast<hipStream_t>(stream);
        c->owns_xs = false;
    });
}

int ibh_comm_set_option(ibh_comm *c, const char *key, int value) {
    return guarded([&] {
        IBH_CHECK(c != nullptr && key != nullptr, "null argument");
        if (!strcmp(key, "planes_padded")) c->planes_padded = value != 0;
        else fail(IBH_ENOKEY, "unknown communicator option '%s' (expected: planes_padded)", key);
    });
}

int ibh_comm_info(const ibh_comm *c, int *world, int *rank) {
    return guarded([&] {
        IBH_CHECK(c != nullptr, "null argument");
        if (world) *world = c->world;
        if (rank) *rank = c->rank;
    });
}

int ibh_comm_wait(ibh_comm *c, void *stream) {
    return guarded([&] {
        IBH_CHECK(c != nullptr, "null argument");
        for (auto &f : c->ring)
            if (f.live && !(f.has_waiter && f.waited == static_cast<hipStream_t>(stream))) {
                IBH_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), f.ev, 0));
                f.has_waiter = true; f.waited = static_cast<hipStream_t>(stream);
            }
    });
}

int ibh_weighted_apply_sharded_device(const ibh_weighted *w, ibh_comm *c, const double *dA_local, int32_t nvar_local, int64_t lda,
                                      double *dB_all, int64_t ldb, double fill, int32_t block_fields, void *stream) {
    return guarded([&] {
        IBH_CHECK(w && c && nvar_local >= 0 && (nvar_local == 0 || (dA_local && dB_all)), "bad arguments");
        int dev = -1;
        IBH_HIP(hipGetDevice(&dev));
        IBH_CHECK(dev == w->device && dev == c->device, "matrix (device %d), communicator (device %d) and current device (%d) differ",
                  w->device, c->device, dev);
        if (nvar_local == 0) return;
        IBH_CHECK(lda >= w->ncol && ldb >= w->nrow, "apply: leading dimensions (%ld, %ld) smaller than (%d, %d)", (long)lda, (long)ldb, w->ncol, w->nrow);
        hipStream_t st = static_cast<hipStream_t>(stream);
        // field blocks: small results travel as ONE exchange per apply (a [64, 122] AvI result is 62 KB: latency-bound), results
        // of megabytes (I-row matrices) a few fields at a time so that the exchange of block b overlaps the SpMM of block b+1
        int bf = block_fields;
        if (bf <= 0) {
            // (the I-row kernel works on 8-16 fields per thread: blocks of fewer than 8 fields cost it its efficiency -- 64 one-field
            // launches of the 1 km IvA take 1.13 ms against 0.20 ms for one 64-field launch)
            const double plane_mb = 8.0 * (double)ldb / 1048576.0;
            bf = (c->world == 1 || plane_mb * nvar_local <= 4.0) ? nvar_local
                 : std::max(8, std::min(nvar_local, 8 * (int)(128.0 / std::max(8.0 * plane_mb, 1e-9))));
        }
        bf = std::min(bf, (int)nvar_local);
        double *mine = dB_all + (int64_t)c->rank * nvar_local * ldb;
        const int64_t stride = (int64_t)nvar_local * ldb;
        // an earlier exchange may still be writing into this result array: order the new SpMM behind it
        order_behind_flights(c, dB_all, dB_all + (int64_t)c->world * stride, st);
        for (int f0 = 0; f0 < nvar_local; f0 += bf) {
            const int nb = std::min(bf, nvar_local - f0);
            spmm_launch(w, dA_local + (int64_t)f0 * lda, nb, lda, mine + (int64_t)f0 * ldb, ldb, fill, 0, st);
            if (c->world > 1) {
                IBH_HIP(hipEventRecord(c->ready, st));
                IBH_HIP(hipStreamWaitEvent(c->xs, c->ready, 0));
                double *base = dB_all + (int64_t)f0 * ldb;
                exchange(c, &base, 1, nb, w->nrow, ldb, stride);
            }
        }
        if (c->world > 1) record_flight(c, dB_all, dB_all + (int64_t)c->world * stride, st);
    });
}

int ibh_weighted_apply_many_sharded_device(const ibh_weighted *w, ibh_comm *c, int32_t nbatch, const double *const *dA_local,
                                           int32_t nvar_local, int64_t lda, double *const *dB_all, int64_t ldb, double fill, void *stream) {
    return guarded([&] {
        IBH_CHECK(w && c && nbatch >= 0 && nvar_local >= 0 && (nbatch == 0 || nvar_local == 0 || (dA_local && dB_all)), "bad arguments");
        int dev = -1;
        IBH_HIP(hipGetDevice(&dev));
        IBH_CHECK(dev == w->device && dev == c->device, "matrix (device %d), communicator (device %d) and current device (%d) differ",
                  w->device, c->device, dev);
        if (nbatch == 0 || nvar_local == 0) return;
        IBH_CHECK(lda >= w->ncol && ldb >= w->nrow, "apply: leading dimensions (%ld, %ld) smaller than (%d, %d)", (long)lda, (long)ldb, w->ncol, w->nrow);
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int64_t stride = (int64_t)nvar_local * ldb;
        std::vector<double *> mine((size_t)nbatch);
        for (int q = 0; q < nbatch; ++q) {
            IBH_CHECK(dA_local[q] && dB_all[q], "apply: null field pointer in batch %d", q);
            mine[(size_t)q] = dB_all[q] + (int64_t)c->rank * stride;
        }
        const double *lo = dB_all[0], *hi = dB_all[0];
        for (int q = 0; q < nbatch; ++q) { lo = std::min<const double *>(lo, dB_all[q]); hi = std::max<const double *>(hi, dB_all[q] + (int64_t)c->world * stride); }
        order_behind_flights(c, lo, hi, st);
        spmm_launch_many(w, nbatch, dA_local, nvar_local, lda, mine.data(), ldb, fill, 0, st);
        if (c->world > 1) {
            IBH_HIP(hipEventRecord(c->ready, st));
            IBH_HIP(hipStreamWaitEvent(c->xs, c->ready, 0));
            exchange(c, dB_all, nbatch, nvar_local, w->nrow, ldb, stride);
            record_flight(c, lo, hi, st);
        }
    });
}

}  // extern "C"
