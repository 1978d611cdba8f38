// ctag_gather.hip -- implementation of include/ctag_gather.h: frame shards, packed result shards and the one
// collective of the path: all-gather of the detected marker lists over RCCL.
//
// The reference has no counterpart (single process); the record layout that is packed is the
// flattened `vector<MarkerInfo>` of a frame (include/ctag_types.h <- /root/reference/header/corner_detector.h:16-22).
// RCCL is bound with dlopen at first use -- no link-time dependency, and a process that already holds a librccl.so.1
// (PyTorch's) shares it instead of loading a second runtime.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only; every entry point is resolved at run time

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <new>
#include <cstdlib>

#include "../../include/ctag_gather.h"
#include "ctag_internal.h"

namespace {

constexpr int kMaxWorld = 64;
constexpr int kSoloSlot = kMaxWorld;                 // size slot of the stand-alone ctag_pack_results (the gather owns slots 0..world-1)
constexpr int kHeadBytes = 16;                      // shard header, and the per-frame head {status, n_markers, n_features, flags}
constexpr int kRecWords = sizeof(ctag_frame_result) / 4;
constexpr int kMarkerWords = sizeof(ctag_marker_rec) / 4;
constexpr int kFeatureWords = sizeof(ctag_feature_rec) / 4;
constexpr int kMarkersOffWords = offsetof(ctag_frame_result, markers) / 4;
constexpr int kFeaturesOffWords = offsetof(ctag_frame_result, features) / 4;
static_assert(sizeof(ctag_frame_result) % 16 == 0 && kMarkersOffWords == 4, "record head is 16 bytes");

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                       // optional: a bounded wait that expires aborts the communicator
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;  // optional: polled while the host waits
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char err[256] = {0};
};

Rccl* rccl_bind();
Rccl* rccl() {  // bound once per process, also when handles live on several host threads
    static Rccl* const R = rccl_bind();
    return R;
}
Rccl* rccl_bind() {
    static Rccl R;
    const char* env = std::getenv("CTAG_RCCL_LIB");
    if (env && *env) R.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    if (!R.lib) R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);  // the one this process already holds
    if (!R.lib) R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!R.lib) R.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!R.lib) {
        std::snprintf(R.err, sizeof(R.err), "cannot load librccl.so.1: %s", dlerror());
        return &R;
    }
    R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(dlsym(R.lib, "ncclGetUniqueId"));
    R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(dlsym(R.lib, "ncclCommInitRank"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.lib, "ncclCommDestroy"));
    R.AllGather = reinterpret_cast<decltype(R.AllGather)>(dlsym(R.lib, "ncclAllGather"));
    R.CommAbort = reinterpret_cast<decltype(R.CommAbort)>(dlsym(R.lib, "ncclCommAbort"));
    R.CommGetAsyncError = reinterpret_cast<decltype(R.CommGetAsyncError)>(dlsym(R.lib, "ncclCommGetAsyncError"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.lib, "ncclGetErrorString"));
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather) {
        std::snprintf(R.err, sizeof(R.err), "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather");
        R.lib = nullptr;
    }
    return &R;
}

struct GatherState {
    int device = 0;
    hipStream_t gstream = nullptr;   // pack, collectives and unpack run here, beside the detection stream
    hipEvent_t ev_main = nullptr, ev_packed = nullptr, ev_sizes = nullptr, ev_done = nullptr;
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int rank = 0, world = 1;
    unsigned char* d_packed = nullptr;
    size_t packed_cap = 0;
    unsigned char* d_gathered = nullptr;
    size_t gathered_cap = 0;
    uint64_t* d_off = nullptr;       // per-frame payload offsets of the work on the gather stream (pack: n+1, unpack: n_total + world)
    size_t off_cap = 0;
    uint64_t* d_off_main = nullptr;  // the same for the stand-alone ctag_pack_results / ctag_unpack_results (handle's main stream)
    size_t off_main_cap = 0;
    uint64_t* d_sizes = nullptr;     // [world] packed size of every rank
    uint64_t* h_sizes = nullptr;     // pinned copy
    int n_local = 0, n_total = 0;
    const ctag_frame_result* local_dev = nullptr;  // of the gather in flight (a rank with pending frames packs again in ctag_gather_end)
    bool tagged = false;             // the sizes in flight carry the ranks' pending-frame counts in their upper bits
    uint64_t pend_gen = 0;
    bool in_flight = false;
    uint64_t last_local = 0, last_padded = 0;
    int timeout_ms = -1;             // deadline of every host wait of the exchange; -1: CTAG_GATHER_TIMEOUT_MS, else 60 s; 0: none
    char err[256] = {0};
};

bool order_unref(ncclComm_t c, bool wait_for_last);  // below
bool order_dead(ncclComm_t c);
void release_comm(GatherState* g);
int env_timeout_ms();
int timeout_of(const struct GatherState* g);
bool stream_idle_within(hipStream_t s, int timeout_ms);  // hipStreamSynchronize with a deadline (0: none)

void gather_state_free(void* p) {
    GatherState* g = static_cast<GatherState*>(p);
    (void)hipSetDevice(g->device);
    // the handle's own deadline (ctag_gather_set_timeout), taken before the communicator goes; once the communicator is dead the waits
    // ahead have had their deadline already: the final drain gets a second, not another full period
    int drain_ms = g->comm ? timeout_of(g) : env_timeout_ms();
    const bool was_dead = g->comm && order_dead(g->comm);
    release_comm(g);
    if (was_dead || (g->err[0] && drain_ms > 1000)) drain_ms = 1000;
    if (g->gstream && !stream_idle_within(g->gstream, drain_ms)) {
        // the gather stream does not drain (an aborted collective that never left?): what it may still touch is leaked rather than freed under it
        delete g;
        return;
    }
    if (g->d_packed) (void)hipFree(g->d_packed);
    if (g->d_gathered) (void)hipFree(g->d_gathered);
    if (g->d_off) (void)hipFree(g->d_off);
    if (g->d_off_main) (void)hipFree(g->d_off_main);
    if (g->d_sizes) (void)hipFree(g->d_sizes);
    if (g->h_sizes) (void)hipHostFree(g->h_sizes);
    for (hipEvent_t e : {g->ev_main, g->ev_packed, g->ev_sizes, g->ev_done})
        if (e) (void)hipEventDestroy(e);
    if (g->gstream) (void)hipStreamDestroy(g->gstream);
    delete g;
}

GatherState* gather_state(ctag_handle* h) {
    void** slot = ctag::handle_gather_slot(h, gather_state_free);
    if (!*slot) {
        GatherState* g = new (std::nothrow) GatherState();
        if (!g) return nullptr;
        g->device = ctag::handle_device(h);
        bool ok = hipSetDevice(g->device) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&g->gstream, hipStreamNonBlocking) == hipSuccess;
        for (hipEvent_t* e : {&g->ev_main, &g->ev_packed, &g->ev_sizes, &g->ev_done})
            ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipMalloc(reinterpret_cast<void**>(&g->d_sizes), (kMaxWorld + 1) * sizeof(uint64_t)) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void**>(&g->h_sizes), (kMaxWorld + 1) * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
        if (!ok) {
            gather_state_free(g);
            return nullptr;
        }
        *slot = g;
    }
    return static_cast<GatherState*>(*slot);
}

// Collectives of ONE communicator issued from several streams (two handles sharing it: ctag_comm_attach(b, ctag_comm_native(a), ..))
// are ordered here explicitly, not left to the library: every collective waits for the event recorded behind the previous
// collective of the same communicator, whatever stream that one went to.  Per process; a communicator has one entry, counted by the
// handles that use it (its owner + the attached ones).  The ENTRY's lock is held from the wait to the record: two threads driving two
// handles on one communicator cannot both wait for the same predecessor and then issue in either order (round-3 ADVICE); the table's
// lock covers look-ups only, so an RCCL call that blocks on its peers stalls the users of that communicator and nobody else (round-4
// ADVICE).  `dead`: a bounded wait expired (or RCCL reported an asynchronous error) and the communicator was aborted -- by whichever
// handle noticed; every handle that shares it then fails its calls instead of enqueuing behind a collective that will never run.
struct CommOrder {
    ncclComm_t comm = nullptr;
    hipEvent_t last = nullptr;   // created when the entry is taken (on the device that is current then: a communicator lives on one device), destroyed with it
    bool has = false;
    std::atomic<bool> dead{false};  // read without the entry's lock: a collective stuck in its issue (which holds mu) must not stall the pollers or the abort
    int refs = 0;
    std::mutex mu;     // wait -> issue -> record of one collective
    std::thread abort_thread;                         // the ncclCommAbort of this communicator, if one was started: joined (bounded) when the entry is given up
    std::shared_ptr<std::atomic<bool>> abort_done;
};
constexpr int kOrderEntries = 16;
CommOrder* const g_order_tab = new CommOrder[kOrderEntries];  // never destroyed: an entry may own a thread that is still inside ncclCommAbort at exit
struct OrderRange {
    CommOrder* begin() const { return g_order_tab; }
    CommOrder* end() const { return g_order_tab + kOrderEntries; }
} g_order;
std::mutex g_order_mu;  // the table: comm / refs of every entry
CommOrder* order_find(ncclComm_t c) {  // caller holds g_order_mu
    for (CommOrder& o : g_order)
        if (o.comm == c) return &o;
    return nullptr;
}
CommOrder* order_lookup(ncclComm_t c) {  // an entry stays put while a handle holds a reference to it
    std::lock_guard<std::mutex> lk(g_order_mu);
    return order_find(c);
}
// a handle starts / stops using communicator c; false: the table is full (16 communicators per process)
bool order_ref(ncclComm_t c) {
    std::lock_guard<std::mutex> lk(g_order_mu);
    if (CommOrder* o = order_find(c)) {
        o->refs++;
        return true;
    }
    for (CommOrder& o : g_order)
        if (!o.comm) {
            if (hipEventCreateWithFlags(&o.last, hipEventDisableTiming) != hipSuccess) return false;
            o.comm = c;
            o.has = false;
            o.dead.store(false);
            o.refs = 1;
            return true;
        }
    return false;
}
bool order_dead(ncclComm_t c) {
    CommOrder* o = order_lookup(c);
    return o && o->dead.load();
}
// ncclCommAbort, once per communicator, whichever handle asks first: the collectives in flight end (their kernels leave), the
// communicator's memory is released by RCCL; nothing may be issued on it afterwards (dead) and its owner must not destroy it again
void order_abort(ncclComm_t c) {
    CommOrder* o = order_lookup(c);
    if (!o) return;
    // `dead` first and without the entry's lock: another handle's thread may sit in a blocking ncclAllGather issue with the lock held
    // (that is the situation an abort exists for); nothing new is issued from here on, and the first caller alone starts the abort
    if (o->dead.exchange(true)) return;
    if (!rccl()->CommAbort) return;
    // ncclCommAbort returns when the communicator's kernels have left -- at once when they are spinning on a peer that is gone, but it
    // waits as long as whatever holds the stream AHEAD of a collective that has not started.  The caller has a deadline to keep: the abort
    // runs on a thread of its own and is given half a second here; the thread stays with the entry and is joined (bounded again) when the
    // last handle lets go of the communicator.  Should it still be busy then it is detached: such a process must leave through _exit()
    // (include/ctag_gather.h) -- static destructors of HIP / RCCL under a running abort are not safe.
    auto done = std::make_shared<std::atomic<bool>>(false);
    ncclResult_t (*abort_fn)(ncclComm_t) = rccl()->CommAbort;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::thread t([=] {
        (void)hipSetDevice(dev);
        (void)abort_fn(c);
        done->store(true);
    });
    {
        std::lock_guard<std::mutex> lk(g_order_mu);  // the table's lock (never held across an RCCL call) guards the thread object
        o->abort_done = done;
        o->abort_thread = std::move(t);
    }
    for (int i = 0; i < 500 && !done->load(); i++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
}
// the entry's abort thread, if any: joined when it is done within `ms`, detached otherwise.  Caller holds g_order_mu.
bool order_join_abort(CommOrder* o, int ms) {
    if (!o->abort_thread.joinable()) return true;
    for (int i = 0; i < ms && !o->abort_done->load(); i++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    const bool done = o->abort_done->load();
    if (done)
        o->abort_thread.join();
    else
        o->abort_thread.detach();
    o->abort_done.reset();
    return done;
}
bool order_unref(ncclComm_t c, bool wait_for_last);  // below: needs the bounded wait
// one collective on `s`, ordered behind the previous one of this communicator and published for the next, under the entry's lock
template <class F>
hipError_t ordered_collective(ncclComm_t c, hipStream_t s, F issue, ncclResult_t* nr) {
    CommOrder* o = order_lookup(c);
    if (!o) return hipErrorInvalidValue;  // ctag_comm_init / _attach registers every communicator
    std::lock_guard<std::mutex> lk(o->mu);
    if (o->dead.load()) {
        *nr = ncclInvalidUsage;  // aborted earlier
        return hipSuccess;
    }
    if (o->has) {
        const hipError_t e = hipStreamWaitEvent(s, o->last, 0);
        if (e != hipSuccess) return e;
    }
    *nr = issue();
    if (*nr != ncclSuccess) return hipSuccess;
    o->has = true;
    return hipEventRecord(o->last, s);
}

// ---- bounded host waits ---------------------------------------------------------------------------------------
// Every host wait of the exchange polls instead of blocking: a peer that died or never reaches its collective would otherwise
// hold this rank in hipEventSynchronize / hipStreamSynchronize for ever.  While it polls it asks RCCL for
// asynchronous errors of the communicator (a peer's process that ends closes its sockets / IPC handles: RCCL notices); at the
// deadline -- CTAG_GATHER_TIMEOUT_MS, default 60 s, 0 = none; ctag_gather_set_timeout overrides it per handle -- the
// communicator is aborted, the call returns CTAG_ERR_HIP with the reason in ctag_comm_last_error and the caller exits.
int env_timeout_ms() {
    static const int v = [] {
        const char* e = std::getenv("CTAG_GATHER_TIMEOUT_MS");
        if (!e || !*e) return 60000;
        const long t = std::strtol(e, nullptr, 10);
        return t < 0 ? 0 : (t > 86400000 ? 86400000 : (int)t);
    }();
    return v;
}
enum WaitResult { kWaitOk = 0, kWaitHipError, kWaitTimeout, kWaitCommError };
// polls `ready()` (hipEventQuery / hipStreamQuery: hipSuccess, hipErrorNotReady or a failure) until it succeeds, the communicator
// reports an asynchronous error, or the deadline passes; *detail receives the hip / RCCL code
template <class Q>
WaitResult poll_until(Q ready, ncclComm_t comm, int timeout_ms, int* detail) {
    using clock = std::chrono::steady_clock;
    const clock::time_point t0 = clock::now();
    Rccl* R = rccl();
    for (unsigned spin = 0;; spin++) {
        const hipError_t e = ready();
        if (e == hipSuccess) return kWaitOk;
        if (e != hipErrorNotReady) {
            *detail = (int)e;
            return kWaitHipError;
        }
        (void)hipGetLastError();  // hipErrorNotReady is sticky in the last-error slot
        const long long us = std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t0).count();
        if (comm && R->CommGetAsyncError && (spin & 63) == 63 && !order_dead(comm)) {
            ncclResult_t ar = ncclSuccess;
            if (R->CommGetAsyncError(comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                *detail = (int)ar;
                return kWaitCommError;
            }
        }
        if (timeout_ms > 0 && us > (long long)timeout_ms * 1000) return kWaitTimeout;
        if (us > 200) std::this_thread::sleep_for(std::chrono::microseconds(us > 20000 ? 200 : 20));  // the sizes arrive within microseconds normally
    }
}

bool stream_idle_within(hipStream_t s, int timeout_ms) {
    int detail = 0;
    return poll_until([&] { return hipStreamQuery(s); }, nullptr, timeout_ms, &detail) == kWaitOk;
}
// returns whether the communicator is dead (aborted): its owner must not destroy it a second time
bool order_unref(ncclComm_t c, bool wait_for_last) {
    CommOrder* o = order_lookup(c);
    if (!o) return false;
    bool dead = o->dead.load();
    bool has = true;
    if (!dead) {  // (a dead communicator's lock may be held for ever by a collective stuck in its issue)
        std::lock_guard<std::mutex> lk(o->mu);
        has = o->has;
    }
    if (wait_for_last && has && !dead) {  // the owner is about to destroy the communicator: nothing issued on it may still run
        int detail = 0;
        hipEvent_t ev = o->last;
        if (poll_until([&] { return hipEventQuery(ev); }, c, env_timeout_ms(), &detail) != kWaitOk) {
            order_abort(c);
            dead = true;
        }
    }
    std::lock_guard<std::mutex> lk(g_order_mu);
    if (--o->refs <= 0) {
        (void)order_join_abort(o, 2000);
        if (o->last) (void)hipEventDestroy(o->last);
        o->comm = nullptr;
        o->last = nullptr;
        o->has = false;
        o->dead.store(false);
        o->refs = 0;
    }
    return dead;
}

#define G_HIP(expr)                                                                              \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            std::snprintf(g->err, sizeof(g->err), "%s: %s", #expr, hipGetErrorString(e__));      \
            return CTAG_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)
#define G_NCCL(expr)                                                                                                    \
    do {                                                                                                                \
        ncclResult_t r__ = (expr);                                                                                      \
        if (r__ != ncclSuccess) {                                                                                       \
            std::snprintf(g->err, sizeof(g->err), "%s: %s", #expr, R->GetErrorString ? R->GetErrorString(r__) : "RCCL error"); \
            return CTAG_ERR_HIP;                                                                                        \
        }                                                                                                               \
    } while (0)

int timeout_of(const GatherState* g) { return g->comm ? (g->timeout_ms >= 0 ? g->timeout_ms : env_timeout_ms()) : 0; }  // no communicator, no peer to wait for
// the bounded form of hipEventSynchronize / hipStreamSynchronize for the waits of the exchange (poll_until above)
template <class Q>
int bounded_wait(GatherState* g, Q ready, const char* what) {
    int detail = 0;
    const int ms = timeout_of(g);
    const WaitResult w = poll_until(ready, g->comm, ms, &detail);
    if (w == kWaitOk) return CTAG_OK;
    if (w == kWaitHipError) {
        std::snprintf(g->err, sizeof(g->err), "%s: %s", what, hipGetErrorString((hipError_t)detail));
        g->in_flight = false;
        return CTAG_ERR_HIP;
    }
    Rccl* R = rccl();
    if (w == kWaitTimeout)
        std::snprintf(g->err, sizeof(g->err), "%s: not complete after %d ms (CTAG_GATHER_TIMEOUT_MS / ctag_gather_set_timeout): a peer is late or gone; communicator aborted", what, ms);
    else
        std::snprintf(g->err, sizeof(g->err), "%s: RCCL reports an asynchronous error (%s); communicator aborted", what,
                      R->GetErrorString ? R->GetErrorString((ncclResult_t)detail) : "?");
    if (g->comm) order_abort(g->comm);
    g->in_flight = false;
    return CTAG_ERR_HIP;
}
int bounded_event(GatherState* g, hipEvent_t ev, const char* what) {
    return bounded_wait(g, [&] { return hipEventQuery(ev); }, what);
}
int bounded_stream(GatherState* g, hipStream_t s, const char* what) {
    return bounded_wait(g, [&] { return hipStreamQuery(s); }, what);
}
// the handle lets go of its communicator: what it enqueued must have run (bounded: a dead peer ends in an abort, not in a hang); the
// owner then waits for the communicator's last collective, whichever handle issued it, and destroys it -- unless it was aborted
void release_comm(GatherState* g) {
    if (!g->comm) return;
    if (g->gstream && !order_dead(g->comm)) (void)bounded_stream(g, g->gstream, "gather stream at communicator release");  // (dead: that wait has expired once already)
    const bool dead = order_unref(g->comm, g->own_comm);
    if (g->own_comm && !dead && rccl()->CommDestroy) (void)rccl()->CommDestroy(g->comm);
    g->comm = nullptr;
    g->own_comm = false;
}

int grow(GatherState* g, unsigned char** p, size_t* cap, size_t need) {
    if (*cap >= need) return CTAG_OK;
    G_HIP(hipDeviceSynchronize());  // rare: a buffer grows; nothing on either stream may still use the old one
    if (*p) G_HIP(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 2 + 4096;
    G_HIP(hipMalloc(reinterpret_cast<void**>(p), want));
    *cap = want;
    return CTAG_OK;
}
int grow_off(GatherState* g, uint64_t** off, size_t* off_cap, size_t entries) {
    size_t cap = *off_cap * sizeof(uint64_t);
    unsigned char* p = reinterpret_cast<unsigned char*>(*off);
    const int r = grow(g, &p, &cap, entries * sizeof(uint64_t));
    *off = reinterpret_cast<uint64_t*>(p);
    *off_cap = cap / sizeof(uint64_t);
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
struct Segments {  // unpack: one packed shard per rank inside the gathered buffer
    int world;
    int lo[kMaxWorld];       // first frame of the shard
    int n[kMaxWorld];        // frames in the shard
    uint64_t base[kMaxWorld];  // byte offset of the shard in the gathered buffer
    uint64_t off0[kMaxWorld];  // first entry of the shard's offsets in `off`
};

__device__ inline uint32_t payload_bytes(int nm, int nf) {
    nm = min(max(nm, 0), CTAG_MAX_MARKERS);
    nf = min(max(nf, 0), CTAG_MAX_FEATURES);
    return (uint32_t)nm * (uint32_t)sizeof(ctag_marker_rec) + (uint32_t)nf * (uint32_t)sizeof(ctag_feature_rec);
}

// exclusive scan of the per-frame payload sizes by ONE block; heads come either from fixed records (stride = record) or from a
// packed frame table (stride = 16 B).  Optionally copies the heads into a packed frame table and writes the shard header.
__device__ void scan_heads(const unsigned char* heads, size_t stride, int n, uint64_t* off, unsigned char* table_out) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int f0 = 0; f0 < n; f0 += blockDim.x) {
        const int f = f0 + threadIdx.x;
        uint64_t v = 0;
        if (f < n) {
            const int4 hd = *reinterpret_cast<const int4*>(heads + (size_t)f * stride);
            v = payload_bytes(hd.y, hd.z);
            if (table_out) *reinterpret_cast<int4*>(table_out + kHeadBytes + (size_t)f * kHeadBytes) = hd;
        }
        uint64_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint64_t before = carry;
        for (int w = 0; w < wave; w++) before += wave_sum[w];
        if (f < n) off[f] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t t = carry;
            for (int w = 0; w < nwaves; w++) t += wave_sum[w];
            carry = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) off[n] = carry;
}

__global__ __launch_bounds__(1024) void k_pack_scan(const ctag_frame_result* res, int n, uint64_t* off, unsigned char* packed, uint64_t* size_out) {
    scan_heads(reinterpret_cast<const unsigned char*>(res), sizeof(ctag_frame_result), n, off, packed);
    if (threadIdx.x == 0) {
        const uint64_t total = (uint64_t)kHeadBytes + (uint64_t)n * kHeadBytes + off[n];
        int32_t* hd = reinterpret_cast<int32_t*>(packed);
        hd[0] = n;
        hd[1] = 0;
        *reinterpret_cast<uint64_t*>(hd + 2) = total;
        if (size_out) *size_out = total;
    }
}

__global__ __launch_bounds__(128) void k_pack(const ctag_frame_result* res, int n, const uint64_t* off, unsigned char* packed) {
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(res + f);
        const int nm = min(max((int)src[1], 0), CTAG_MAX_MARKERS), nf = min(max((int)src[2], 0), CTAG_MAX_FEATURES);
        uint32_t* dst = reinterpret_cast<uint32_t*>(packed + kHeadBytes + (size_t)n * kHeadBytes + off[f]);
        const int mw = nm * kMarkerWords, fw = nf * kFeatureWords;
        for (int i = threadIdx.x; i < mw; i += blockDim.x) dst[i] = src[kMarkersOffWords + i];
        for (int i = threadIdx.x; i < fw; i += blockDim.x) dst[mw + i] = src[kFeaturesOffWords + i];
    }
}

__global__ __launch_bounds__(1024) void k_unpack_scan(const unsigned char* gathered, Segments S, uint64_t* off) {
    const int r = blockIdx.x;
    scan_heads(gathered + S.base[r] + kHeadBytes, kHeadBytes, S.n[r], off + S.off0[r], nullptr);
}

__global__ __launch_bounds__(256) void k_unpack(const unsigned char* gathered, Segments S, const uint64_t* off, ctag_frame_result* out) {
    const int r = blockIdx.y;
    const int n = S.n[r];
    const unsigned char* shard = gathered + S.base[r];
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        const uint32_t* head = reinterpret_cast<const uint32_t*>(shard + kHeadBytes + (size_t)f * kHeadBytes);
        const int nm = min(max((int)head[1], 0), CTAG_MAX_MARKERS), nf = min(max((int)head[2], 0), CTAG_MAX_FEATURES);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(shard + kHeadBytes + (size_t)n * kHeadBytes + off[S.off0[r] + f]);
        uint32_t* dst = reinterpret_cast<uint32_t*>(out + S.lo[r] + f);
        const int mw = nm * kMarkerWords, fw = nf * kFeatureWords;
        for (int i = threadIdx.x; i < kRecWords; i += blockDim.x) {
            uint32_t v = 0;
            if (i < kMarkersOffWords) v = head[i];
            else if (i < kFeaturesOffWords) v = (i - kMarkersOffWords) < mw ? src[i - kMarkersOffWords] : 0u;
            else v = (i - kFeaturesOffWords) < fw ? src[mw + i - kFeaturesOffWords] : 0u;
            dst[i] = v;
        }
    }
}

// segment table of a `world`-rank job: shard r = ctag_shard_range(n_total, r, world), placed at r * width in the gathered buffer
Segments build_segments(int n_total, int world, uint64_t width) {
    Segments S{};
    S.world = world;
    uint64_t off0 = 0;
    for (int r = 0; r < world; r++) {
        int lo = 0, hi = 0;
        (void)ctag_shard_range(n_total, r, world, &lo, &hi);
        S.lo[r] = lo;
        S.n[r] = hi - lo;
        S.base[r] = (uint64_t)r * width;
        S.off0[r] = off0;
        off0 += (uint64_t)(hi - lo) + 1;
    }
    return S;
}

int enqueue_pack(GatherState* g, bool main_stream, const ctag_frame_result* results_dev, int n, unsigned char* packed, uint64_t* size_dev, hipStream_t s) {
    uint64_t** off = main_stream ? &g->d_off_main : &g->d_off;
    const int r = grow_off(g, off, main_stream ? &g->off_main_cap : &g->off_cap, (size_t)n + 1 + kMaxWorld);
    if (r != CTAG_OK) return r;
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, s, results_dev, n, *off, packed, size_dev);
    if (n > 0) hipLaunchKernelGGL(k_pack, dim3(std::min(n, 65535)), dim3(128), 0, s, results_dev, n, *off, packed);
    G_HIP(hipGetLastError());
    return CTAG_OK;
}

int enqueue_unpack(GatherState* g, bool main_stream, const unsigned char* gathered, const Segments& S, int n_total, ctag_frame_result* out, hipStream_t s) {
    uint64_t** offp = main_stream ? &g->d_off_main : &g->d_off;
    const int r = grow_off(g, offp, main_stream ? &g->off_main_cap : &g->off_cap, (size_t)n_total + 1 + kMaxWorld);
    if (r != CTAG_OK) return r;
    uint64_t* d_off = *offp;
    int nmax = 0;
    for (int k = 0; k < S.world; k++) nmax = std::max(nmax, S.n[k]);
    hipLaunchKernelGGL(k_unpack_scan, dim3(S.world), dim3(1024), 0, s, gathered, S, d_off);
    if (nmax > 0) hipLaunchKernelGGL(k_unpack, dim3(std::min(nmax, 65535), S.world), dim3(256), 0, s, gathered, S, d_off, out);
    G_HIP(hipGetLastError());
    return CTAG_OK;
}

}  // namespace

namespace ctag {
int gather_unpack_gathered(ctag_handle* h, const void* gathered_dev, int n_total, int world, uint64_t width, ctag_frame_result* out_dev) {
    if (!h || !gathered_dev || n_total < 0 || world < 1 || world > kMaxWorld || (n_total > 0 && !out_dev) || (width & 255)) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    if (g->in_flight) return CTAG_ERR_ARG;
    G_HIP(hipSetDevice(g->device));
    const Segments S = build_segments(n_total, world, width);
    const int rc = enqueue_unpack(g, false, static_cast<const unsigned char*>(gathered_dev), S, n_total, out_dev, g->gstream);
    if (rc != CTAG_OK) return rc;
    G_HIP(hipStreamSynchronize(g->gstream));
    return CTAG_OK;
}
}  // namespace ctag

// The size a rank announces carries, above bit 40, how many of its frames still wait for the any-frame pass (CTAG_PENDING records: a cluttered frame
// that exceeded the batch workspace's pools).  Every rank sees every count with the sizes, so all of them take the same path in ctag_gather_end:
// none pending (always, on ordinary content) -> the payload exchange as it is; some -> those ranks complete their frames, all pack and announce again.
// ctag_gather_begin therefore never waits for the detection it is enqueued behind.
constexpr int kPendShift = 40;
constexpr uint64_t kSizeMask = (1ull << kPendShift) - 1ull;
__global__ void k_tag_pending(uint64_t* size, const int32_t* pending_count) {
    const int c = *pending_count;
    *size = (*size & kSizeMask) | ((uint64_t)(c < 0 ? 0 : (c > 0xffffff ? 0xffffff : c)) << kPendShift);
}

extern "C" {

int ctag_shard_range(int n_total, int rank, int world, int* lo, int* hi) {
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return CTAG_ERR_ARG;
    const int base = n_total / world, rem = n_total % world;
    *lo = rank * base + std::min(rank, rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return CTAG_OK;
}

size_t ctag_packed_capacity(int n_frames) {
    if (n_frames < 0) return 0;
    return (size_t)kHeadBytes + (size_t)n_frames * (kHeadBytes + CTAG_MAX_MARKERS * sizeof(ctag_marker_rec) + CTAG_MAX_FEATURES * sizeof(ctag_feature_rec));
}

int ctag_pack_results(ctag_handle* h, const ctag_frame_result* results_dev, int n, void* packed_dev, size_t capacity, uint64_t* packed_bytes_host) {
    if (!h || n < 0 || (n > 0 && !results_dev) || !packed_dev || capacity < ctag_packed_capacity(n)) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    G_HIP(hipSetDevice(g->device));
    {   // records of frames that wait for the any-frame pass (CTAG_PENDING) are completed before they are packed
        const int fr = ctag::handle_finish_pending(h);
        if (fr != CTAG_OK) return fr;
    }
    hipStream_t s = static_cast<hipStream_t>(ctag_stream(h));
    const int r = enqueue_pack(g, true, results_dev, n, static_cast<unsigned char*>(packed_dev), packed_bytes_host ? g->d_sizes + kSoloSlot : nullptr, s);
    if (r != CTAG_OK) return r;
    if (packed_bytes_host) {  // a slot of its own: a gather in flight on the gather stream owns slots 0..world-1
        G_HIP(hipMemcpyAsync(g->h_sizes + kSoloSlot, g->d_sizes + kSoloSlot, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        G_HIP(hipStreamSynchronize(s));
        *packed_bytes_host = g->h_sizes[kSoloSlot];
    }
    return CTAG_OK;
}

int ctag_unpack_results(ctag_handle* h, const void* packed_dev, int n, ctag_frame_result* out_dev) {
    if (!h || n < 0 || !packed_dev || (n > 0 && !out_dev)) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    G_HIP(hipSetDevice(g->device));
    Segments S{};
    S.world = 1;
    S.n[0] = n;
    return enqueue_unpack(g, true, static_cast<const unsigned char*>(packed_dev), S, n, out_dev, static_cast<hipStream_t>(ctag_stream(h)));
}

int ctag_comm_unique_id(void* id_bytes) {
    if (!id_bytes) return CTAG_ERR_ARG;
    Rccl* R = rccl();
    if (!R->lib) return CTAG_ERR_HIP;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return CTAG_ERR_HIP;
    static_assert(sizeof(id) == CTAG_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id_bytes, &id, sizeof(id));
    return CTAG_OK;
}

int ctag_comm_destroy(ctag_handle* h) {
    if (!h) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    if (g->comm) {
        (void)hipSetDevice(g->device);
        release_comm(g);
    }
    g->comm = nullptr;
    g->own_comm = false;
    g->rank = 0;
    g->world = 1;
    g->in_flight = false;
    return CTAG_OK;
}

int ctag_gather_set_timeout(ctag_handle* h, int timeout_ms) {
    if (!h) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    g->timeout_ms = timeout_ms < 0 ? -1 : timeout_ms;
    return CTAG_OK;
}

int ctag_comm_init(ctag_handle* h, const void* id_bytes, int rank, int world) {
    if (!h || !id_bytes || world < 1 || world > kMaxWorld || rank < 0 || rank >= world) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    Rccl* R = rccl();
    if (!R->lib) {
        std::snprintf(g->err, sizeof(g->err), "%s", R->err);
        return CTAG_ERR_HIP;
    }
    (void)ctag_comm_destroy(h);
    G_HIP(hipSetDevice(g->device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    G_NCCL(R->CommInitRank(&g->comm, world, id, rank));
    if (!order_ref(g->comm)) {
        (void)R->CommDestroy(g->comm);
        g->comm = nullptr;
        std::snprintf(g->err, sizeof(g->err), "more than 16 communicators in this process");
        return CTAG_ERR_LIMIT;
    }
    g->own_comm = true;
    g->rank = rank;
    g->world = world;
    return CTAG_OK;
}

int ctag_comm_attach(ctag_handle* h, void* nccl_comm, int rank, int world) {
    if (!h || !nccl_comm || world < 1 || world > kMaxWorld || rank < 0 || rank >= world) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    Rccl* R = rccl();
    if (!R->lib) {
        std::snprintf(g->err, sizeof(g->err), "%s", R->err);
        return CTAG_ERR_HIP;
    }
    (void)ctag_comm_destroy(h);
    G_HIP(hipSetDevice(g->device));
    if (!order_ref(static_cast<ncclComm_t>(nccl_comm))) {
        std::snprintf(g->err, sizeof(g->err), "more than 16 communicators in this process");
        return CTAG_ERR_LIMIT;
    }
    // the communicator stays its owner's: the owner must outlive every collective issued through an attached handle (its ctag_comm_destroy /
    // ctag_destroy waits for the last one issued, on any handle, before it destroys the communicator; nothing may be issued afterwards)
    g->comm = static_cast<ncclComm_t>(nccl_comm);
    g->own_comm = false;
    g->rank = rank;
    g->world = world;
    return CTAG_OK;
}

void* ctag_comm_native(ctag_handle* h) {
    if (!h) return nullptr;
    GatherState* g = gather_state(h);
    return g ? static_cast<void*>(g->comm) : nullptr;
}

const char* ctag_comm_last_error(ctag_handle* h) {
    if (!h) return "";
    GatherState* g = gather_state(h);
    return g ? g->err : "";
}

int ctag_gather_begin(ctag_handle* h, const ctag_frame_result* local_dev, int n_local, int n_total) {
    if (!h || n_local < 0 || n_total < n_local || (n_local > 0 && !local_dev)) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    if (g->in_flight) return CTAG_ERR_ARG;
    int lo = 0, hi = 0;
    (void)ctag_shard_range(n_total, g->rank, g->world, &lo, &hi);
    if (hi - lo != n_local) {
        std::snprintf(g->err, sizeof(g->err), "rank %d of %d owns %d of %d frames, got n_local = %d", g->rank, g->world, hi - lo, n_total, n_local);
        return CTAG_ERR_ARG;
    }
    Rccl* R = rccl();
    if (g->world > 1 && (!g->comm || !R->lib)) {
        std::snprintf(g->err, sizeof(g->err), "no communicator: call ctag_comm_init / ctag_comm_attach first");
        return CTAG_ERR_ARG;
    }
    if (g->comm && order_dead(g->comm)) {
        std::snprintf(g->err, sizeof(g->err), "the communicator was aborted by an earlier gather (deadline or RCCL error): nothing can be gathered through it");
        return CTAG_ERR_HIP;
    }
    G_HIP(hipSetDevice(g->device));
    // records of frames that wait for the any-frame pass (CTAG_PENDING) must be completed before they travel; whether there are any is
    // known only when the detection ahead has run, so the question rides with the sizes (k_tag_pending) and is answered in ctag_gather_end
    const int32_t* pend_count = nullptr;
    g->tagged = ctag::handle_pending_state(h, &pend_count, &g->pend_gen);
    g->local_dev = local_dev;
    hipStream_t main_s = static_cast<hipStream_t>(ctag_stream(h));
    const int n_max = (n_total + g->world - 1) / g->world;
    // the payload all-gather sends the largest packed size ROUNDED UP to 256 bytes from this buffer
    const size_t packed_need = (ctag_packed_capacity(n_max) + 255) & ~(size_t)255;
    const bool fresh = g->packed_cap < packed_need;
    int r = grow(g, &g->d_packed, &g->packed_cap, packed_need);
    if (r != CTAG_OK) return r;
    if (fresh) G_HIP(hipMemsetAsync(g->d_packed, 0, g->packed_cap, g->gstream));  // padding bytes that travel are defined
    // the gather stream picks up behind the detection already enqueued on the main stream
    G_HIP(hipEventRecord(g->ev_main, main_s));
    G_HIP(hipStreamWaitEvent(g->gstream, g->ev_main, 0));
    r = enqueue_pack(g, false, local_dev, n_local, g->d_packed, g->d_sizes + g->rank, g->gstream);
    if (r != CTAG_OK) return r;
    if (g->tagged) hipLaunchKernelGGL(k_tag_pending, dim3(1), dim3(1), 0, g->gstream, g->d_sizes + g->rank, pend_count);
    // local_dev may be overwritten by whatever the caller enqueues next on the main stream: order it behind the pack
    G_HIP(hipEventRecord(g->ev_packed, g->gstream));
    G_HIP(hipStreamWaitEvent(main_s, g->ev_packed, 0));
    if (g->comm) {
        ncclResult_t nr = ncclSuccess;
        G_HIP(ordered_collective(g->comm, g->gstream, [&] { return R->AllGather(g->d_sizes + g->rank, g->d_sizes, 1, ncclUint64, g->comm, g->gstream); }, &nr));
        G_NCCL(nr);
    }
    G_HIP(hipMemcpyAsync(g->h_sizes, g->d_sizes, sizeof(uint64_t) * g->world, hipMemcpyDeviceToHost, g->gstream));
    G_HIP(hipEventRecord(g->ev_sizes, g->gstream));
    g->n_local = n_local;
    g->n_total = n_total;
    g->in_flight = true;
    return CTAG_OK;
}

int ctag_gather_end(ctag_handle* h, ctag_frame_result* out_dev) {
    if (!h || !out_dev) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    if (!g->in_flight) return CTAG_ERR_ARG;
    g->in_flight = false;
    Rccl* R = rccl();
    G_HIP(hipSetDevice(g->device));
    {   // the only host wait of the exchange: world x 8 bytes -- bounded (a dead or late peer must not hold this rank for ever)
        const int wr = bounded_event(g, g->ev_sizes, "all-gather of the packed sizes");
        if (wr != CTAG_OK) return wr;
    }
    {
        uint64_t pending_any = 0;
        for (int r = 0; r < g->world; r++) {
            pending_any |= g->h_sizes[r] >> kPendShift;
            g->h_sizes[r] &= kSizeMask;
        }
        if (g->tagged && !pending_any) ctag::handle_pending_clean(h, g->pend_gen);
        if (pending_any) {
            // the rare path, taken by EVERY rank (they all read the same counts): complete the pending frames where there are any (a host wait for
            // this rank's detection and the any-frame passes), then pack and announce again.  local_dev must still hold the records it held at
            // ctag_gather_begin (include/ctag_gather.h).
            const int fr = ctag::handle_finish_pending(h);
            if (fr != CTAG_OK) return fr;
            hipStream_t main_s = static_cast<hipStream_t>(ctag_stream(h));
            G_HIP(hipEventRecord(g->ev_main, main_s));
            G_HIP(hipStreamWaitEvent(g->gstream, g->ev_main, 0));
            const int pr = enqueue_pack(g, false, g->local_dev, g->n_local, g->d_packed, g->d_sizes + g->rank, g->gstream);
            if (pr != CTAG_OK) return pr;
            G_HIP(hipEventRecord(g->ev_packed, g->gstream));
            G_HIP(hipStreamWaitEvent(main_s, g->ev_packed, 0));
            if (g->comm) {
                ncclResult_t nr = ncclSuccess;
                G_HIP(ordered_collective(g->comm, g->gstream, [&] { return R->AllGather(g->d_sizes + g->rank, g->d_sizes, 1, ncclUint64, g->comm, g->gstream); }, &nr));
                G_NCCL(nr);
            }
            G_HIP(hipMemcpyAsync(g->h_sizes, g->d_sizes, sizeof(uint64_t) * g->world, hipMemcpyDeviceToHost, g->gstream));
            const int wr = bounded_stream(g, g->gstream, "second all-gather of the packed sizes");
            if (wr != CTAG_OK) return wr;
            for (int r = 0; r < g->world; r++) g->h_sizes[r] &= kSizeMask;
        }
    }
    uint64_t width = 0;
    for (int r = 0; r < g->world; r++) {
        if (g->h_sizes[r] > ctag_packed_capacity((g->n_total + g->world - 1) / g->world)) {
            std::snprintf(g->err, sizeof(g->err), "rank %d reports a packed shard of %llu bytes", r, (unsigned long long)g->h_sizes[r]);
            return CTAG_ERR_ARG;
        }
        width = std::max(width, g->h_sizes[r]);
    }
    width = (width + 255) & ~(uint64_t)255;
    g->last_local = g->h_sizes[g->rank];
    g->last_padded = width;
    const Segments S = build_segments(g->n_total, g->world, width);
    const unsigned char* gathered = g->d_packed;
    if (g->comm) {
        const int rc = grow(g, &g->d_gathered, &g->gathered_cap, (size_t)width * g->world);
        if (rc != CTAG_OK) return rc;
        ncclResult_t nr = ncclSuccess;
        G_HIP(ordered_collective(g->comm, g->gstream, [&] { return R->AllGather(g->d_packed, g->d_gathered, (size_t)width, ncclUint8, g->comm, g->gstream); }, &nr));
        G_NCCL(nr);
        gathered = g->d_gathered;
    }
    const int rc = enqueue_unpack(g, false, gathered, S, g->n_total, out_dev, g->gstream);
    if (rc != CTAG_OK) return rc;
    G_HIP(hipEventRecord(g->ev_done, g->gstream));
    return CTAG_OK;
}

int ctag_gather_wait(ctag_handle* h) {
    if (!h) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    G_HIP(hipSetDevice(g->device));
    return bounded_stream(g, g->gstream, "payload all-gather + unpack");
}

int ctag_gather(ctag_handle* h, const ctag_frame_result* local_dev, int n_local, int n_total, ctag_frame_result* out_dev) {
    int r = ctag_gather_begin(h, local_dev, n_local, n_total);
    if (r != CTAG_OK) return r;
    r = ctag_gather_end(h, out_dev);
    if (r != CTAG_OK) return r;
    return ctag_gather_wait(h);
}

int ctag_gather_last_bytes(ctag_handle* h, uint64_t* local_bytes, uint64_t* padded_bytes) {
    if (!h) return CTAG_ERR_ARG;
    GatherState* g = gather_state(h);
    if (!g) return CTAG_ERR_HIP;
    g->err[0] = 0;
    if (local_bytes) *local_bytes = g->last_local;
    if (padded_bytes) *padded_bytes = g->last_padded;
    return CTAG_OK;
}

}  // extern "C"
