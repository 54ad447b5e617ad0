// b2x_capi.cpp — implementation of the C ABI declared in include/b2x.h (HIP runtime glue).
// No computation happens on the host here: plan_create compiles + uploads metadata, plan_execute
// launches the gfx950 kernels.  Without a usable device every compute entry point fails loudly.
#include "../../include/b2x.h"
#include "b2x_kernels.h"
#include "b2x_plan.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <hip/hip_runtime.h>
#include <string>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

using namespace b2x;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int b2x_set_error(int code, const std::string &msg) { return fail(code, msg); } // for b2x_comm.cpp
#define HIPCHK(expr)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return fail(B2X_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                           \
    } while (0)

static const uint64_t kSlackElems = 8; // zeroed elements behind every buffer this library allocates itself

// B2X_DEBUG_POISON=1 (test knob): memory the kernels must never consume — the slack behind owned buffers, the whole W
// scratch before its first use — is filled with NaN instead of zeros, so a stray read shows up in the result.
static bool poison() {
    const char *e = getenv("B2X_DEBUG_POISON");
    return e != nullptr && e[0] == '1';
}
static hipError_t fill_tail(double *p, uint64_t n) { // n elements: zeros (or NaN under the test knob)
    return hipMemset(p, poison() ? 0xFF : 0, n * sizeof(double));
}

// The big per-plan buffers (W scratch, partial slabs) are recycled between plans: a DMRG sweep creates one plan per site
// and destroys it before the next, and a hipMalloc of the 16 GiB scratch + 16 GB of slabs of an M=4000 plan costs ~0.7 s
// (b2x_plan_create 1.05 s against 0.30 s for the plan compiler itself; tools/compile_time.py).  A freed buffer is kept
// (at most kPoolMax buffers, B2X_POOL_MB MiB in total, default a quarter of the card; 0 disables) and handed to the next plan that asks
// for at most its size and at least half of it.  When an allocation fails the pool is emptied and the allocation retried.
namespace {
struct PoolBuf {
    void *p;
    size_t bytes;
};
std::mutex g_pool_mu;
std::vector<PoolBuf> g_pool;
const size_t kPoolMax = 6;
size_t pool_cap_bytes() {
    // B2X_POOL_MB, default: a quarter of the card's memory (72 GB on an MI355X) — the W scratch + slabs of the largest plans
    static const size_t cap = []() -> size_t {
        if (const char *e = getenv("B2X_POOL_MB"))
            return (size_t)atoll(e) << 20;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess)
            (void)hipGetLastError(), tot = (size_t)64 << 30;
        return tot / 4;
    }();
    return cap;
}
void pool_trim_locked(size_t keep_bytes) {
    size_t tot = 0;
    for (const PoolBuf &b : g_pool)
        tot += b.bytes;
    while (!g_pool.empty() && (tot > keep_bytes || g_pool.size() > kPoolMax)) { // oldest first
        tot -= g_pool.front().bytes;
        (void)hipFree(g_pool.front().p);
        g_pool.erase(g_pool.begin());
    }
}
} // namespace
// Every hipMalloc of this library goes through dev_malloc: when the device is out of memory the idle buffers the library
// itself holds are given back first — the buffer pool, then the parked plans of the compiled-plan cache (their work lists) —
// and the allocation is retried.  (b2x_trim does the same on request, for other allocators of the process.)
static size_t reclaim_cached_plans(); // defined behind the plan cache
static size_t vec_cache_flush();      // defined with cached_alloc below
static hipError_t dev_malloc(void **out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess)
        return e;
    (void)hipGetLastError();
    vec_cache_flush();
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool_trim_locked(0);
    }
    if ((e = hipMalloc(out, bytes)) == hipSuccess)
        return e;
    (void)hipGetLastError();
    if (reclaim_cached_plans() == 0)
        return e;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu); // (the scratch of the evicted plans went back to the pool)
        pool_trim_locked(0);
    }
    e = hipMalloc(out, bytes);
    if (e != hipSuccess)
        (void)hipGetLastError();
    return e;
}
// cached_alloc / cached_free (b2x_device_alloc / b2x_device_free and the library's own work lists) keep freed vectors for the next request of the same size class.  A sweep allocates and
// frees the same handful of sizes at every site (enlarged and rotated blocks, the operator arena, psi-sized vectors), and a
// hipMalloc of tens of MB costs about a millisecond, a hipFree a device-wide wait: of the 35 ms an H10 M=500 site took on the
// host, 8 were these calls (tools/sweep_profile.py).  Size classes are 1/8 octave wide (<= 12.5 % slack behind a vector); a
// buffer is parked only after the device is idle (what hipFree waits for as well), so a parked buffer has no reader left.
// B2X_VEC_CACHE_MB caps the parked bytes (default 1/16 of the card, 0 = off); the oldest parked buffers go first; an
// allocation that fails anywhere in the library flushes the cache before it gives up (dev_malloc), b2x_trim flushes it too.
namespace {
struct ParkedVec {
    void *p;
    uint64_t seq;
};
std::mutex g_vc_mu;
std::multimap<size_t, ParkedVec> g_vc_free;      // parked buffers by capacity
std::unordered_map<void *, size_t> g_vc_live;   // capacity of every vector handed out
size_t g_vc_bytes = 0;
uint64_t g_vc_seq = 0;
size_t vec_cache_cap() {
    static const size_t cap = []() -> size_t {
        if (const char *e = getenv("B2X_VEC_CACHE_MB"))
            return (size_t)atoll(e) << 20;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess)
            (void)hipGetLastError(), tot = (size_t)64 << 30;
        return tot / 16;
    }();
    return cap;
}
size_t vec_class(size_t bytes) {
    if (bytes <= 4096)
        return 4096;
    const int lg = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (lg - 3);
    return (bytes + step - 1) & ~(step - 1);
}
} // namespace
static size_t vec_cache_flush() {
    std::vector<void *> out;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_vc_mu);
        for (auto &kv : g_vc_free)
            out.push_back(kv.second.p), bytes += kv.first;
        g_vc_free.clear();
        g_vc_bytes = 0;
    }
    for (void *p : out)
        (void)hipFree(p);
    return bytes;
}
static hipError_t cached_alloc(void **out, size_t bytes) {
    const size_t cap = vec_class(bytes ? bytes : 8);
    if (vec_cache_cap() == 0 || cap > vec_cache_cap() / 4) // never parked: exactly the bytes asked for (a 90 GB arena gets no 6 GB of slack)
        return dev_malloc(out, bytes ? bytes : 8);
    {
        std::lock_guard<std::mutex> lk(g_vc_mu);
        auto it = g_vc_free.find(cap);
        if (it != g_vc_free.end()) {
            *out = it->second.p;
            g_vc_free.erase(it);
            g_vc_bytes -= cap;
            g_vc_live[*out] = cap;
            return hipSuccess;
        }
    }
    hipError_t e = dev_malloc(out, cap);
    if (e != hipSuccess)
        return e;
    std::lock_guard<std::mutex> lk(g_vc_mu);
    g_vc_live[*out] = cap;
    return hipSuccess;
}
static hipError_t cached_free(void *dptr) {
    if (!dptr)
        return hipSuccess;
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> lk(g_vc_mu);
        auto it = g_vc_live.find(dptr);
        if (it != g_vc_live.end())
            cap = it->second, g_vc_live.erase(it);
    }
    if (cap == 0 || vec_cache_cap() == 0 || cap > vec_cache_cap() / 4) // not from cached_alloc, or too big to park
        return hipFree(dptr);
    hipError_t e = hipDeviceSynchronize(); // (hipFree's own guarantee: nothing on the device still reads the vector)
    if (e != hipSuccess)
        return e;
    std::vector<void *> evict;
    {
        std::lock_guard<std::mutex> lk(g_vc_mu);
        g_vc_free.emplace(cap, ParkedVec{dptr, g_vc_seq++});
        g_vc_bytes += cap;
        while (g_vc_bytes > vec_cache_cap()) { // oldest first
            auto old = g_vc_free.begin();
            for (auto it = g_vc_free.begin(); it != g_vc_free.end(); ++it)
                if (it->second.seq < old->second.seq)
                    old = it;
            evict.push_back(old->second.p);
            g_vc_bytes -= old->first;
            g_vc_free.erase(old);
        }
    }
    for (void *p : evict)
        (void)hipFree(p);
    return hipSuccess;
}
namespace {
hipError_t pool_alloc(void **out, size_t bytes, size_t *got) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        int best = -1;
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i].bytes >= bytes && g_pool[i].bytes / 2 <= bytes && (best < 0 || g_pool[i].bytes < g_pool[best].bytes))
                best = (int)i;
        if (best >= 0) {
            *out = g_pool[best].p, *got = g_pool[best].bytes;
            g_pool.erase(g_pool.begin() + best);
            return hipSuccess;
        }
    }
    *got = bytes;
    return bytes < ((size_t)64 << 20) ? cached_alloc(out, bytes) : dev_malloc(out, bytes);
}
void pool_free(void *p, size_t bytes) {
    if (!p)
        return;
    if (pool_cap_bytes() == 0 || bytes < ((size_t)64 << 20) || bytes > pool_cap_bytes()) { // small buffers: the vector cache's business
        (void)cached_free(p);
        return;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(PoolBuf{p, bytes});
    pool_trim_locked(pool_cap_bytes());
}
} // namespace

struct b2x_arena {
    double *dev = nullptr;
    uint64_t len = 0;
    uint64_t cap = 0; // elements that may be read: len, + the slack of an owned arena
    bool owned = false;
    std::vector<const double *> host_bases; // sorted
    std::vector<uint64_t> host_lens, offs;
};

struct b2x_plan {
    const b2x_arena *arena = nullptr;
    b2x_plan_stats stats{};
    bool fallback = false;
    int kernel = 0;
    DPart *d_parts[kNumClasses] = {nullptr, nullptr, nullptr};
    DItem *d_items[kNumClasses] = {nullptr, nullptr, nullptr};
    uint32_t n_items[kNumClasses] = {0, 0, 0};
    DTile *d_tiles = nullptr;
    uint32_t n_tiles = 0;
    double *d_slabs = nullptr;
    b2x_pair *d_pairs = nullptr; // generic kernel only
    uint32_t n_pairs = 0;
    // two-stage path
    GSeg *d_gsegs = nullptr;
    GItem *d_gitems = nullptr;
    DTile *d_gtiles = nullptr;
    double *d_scratch = nullptr, *d_gslabs = nullptr;
    size_t scratch_bytes = 0, gslabs_bytes = 0, slabs_bytes = 0; // (pool_alloc / pool_free)
    OWork *d_sum_work = nullptr;   // sum pass between the stages (distributive law), scratch -> scratch
    OEntry *d_sum_entries = nullptr;
    std::vector<SuperStep> steps;
    std::vector<uint32_t> step_max_elems; // per super-step: elements of its largest psi' tile
    std::vector<uint32_t> step_max_items; // ... and the largest number of partial slabs of one of its tiles
    std::vector<StageCopy> stage_in; // input-vector operands copied into the scratch at the start of every execute
    // the short-tile class of a stage runs beside the tall one on a stream of its own (fork / join with events)
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    double *d_psi = nullptr, *d_sigma = nullptr; // staging for host-pointer execute
    size_t psi_len = 0, sigma_len = 0;
    int dominant_cls = 0;
    bool seg_scaled = false; // single-GEMM list plan (gg_kernel SCALED variant)
    bool short_narrow = false; // the short tile class runs as 1-wave workgroups
    int short_frags = kGGShortFrags;
    int gg_tile_n = 128;
    // The launches of one H.psi as a HIP graph (device-pointer execute only): captured on first use, replayed with the
    // psi / sigma / scale arguments of its kernel nodes patched per call.  An H.psi of a small plan is a handful of
    // short kernels on two streams; launched one by one, the gaps between them (launch latency, the event fork / join of
    // the two tile classes) are a tenth of its time.
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    hipStream_t cap_stream = nullptr;
    struct GNode {
        hipGraphNode_t node;
        hipKernelNodeParams params; // kernelParams points into the graph's own copy of the arguments
        int psi_slot, sigma_slot, scale_slot;
    };
    std::vector<GNode> gnodes;
    const double *g_psi = nullptr;
    double *g_sigma = nullptr;
    double g_scale = 0;
    bool graph_failed = false;
    uint32_t n_device_exec = 0; // device-pointer executes of this binding so far (the graph is captured at the second)
    // what re-binding a cached plan to the next site's arena needs (plan_bind) and the cache key (see g_plan_cache)
    uint64_t scratch_elems = 0, gslab_elems = 0, slab_elems = 0;
    std::vector<StageCopy> stage_arena; // staged operands with an arena source: copied at every bind
    std::vector<uint64_t> scratch_pads;
    bool cacheable = false;
    size_t meta_bytes = 0; // device bytes of the work lists (what a cached plan keeps resident)
    struct Key {
        uint64_t h1 = 0, h2 = 0, n = 0, in_len = 0, out_len = 0, arena_len = 0, arena_cap = 0;
        int kind = 0; // 0 = pair plan, 1 = single-GEMM list
        b2x_plan_options opt{};
        std::vector<unsigned char> blob; // the records themselves: a hit is confirmed byte by byte
    } key;
};

// give back what belongs to ONE binding of the plan: scratch and slabs (to the pool), host-pointer staging, the graph
static void plan_unbind(b2x_plan *p) {
    // (hipFree waits for the device; buffers that go back to the pool must be idle as well before another plan gets them)
    if (p->d_scratch || p->d_gslabs || p->d_slabs)
        (void)hipDeviceSynchronize();
    pool_free(p->d_slabs, p->slabs_bytes), p->d_slabs = nullptr;
    pool_free(p->d_scratch, p->scratch_bytes), p->d_scratch = nullptr;
    pool_free(p->d_gslabs, p->gslabs_bytes), p->d_gslabs = nullptr;
    if (p->d_psi)
        (void)hipFree(p->d_psi), p->d_psi = nullptr;
    if (p->d_sigma)
        (void)hipFree(p->d_sigma), p->d_sigma = nullptr;
    if (p->gexec)
        (void)hipGraphExecDestroy(p->gexec), p->gexec = nullptr;
    if (p->graph)
        (void)hipGraphDestroy(p->graph), p->graph = nullptr;
    p->gnodes.clear();
    p->g_psi = nullptr, p->g_sigma = nullptr, p->graph_failed = false, p->n_device_exec = 0;
    p->arena = nullptr;
}

static void plan_free(b2x_plan *p) {
    plan_unbind(p);
    for (int k = 0; k < kNumClasses; k++) {
        if (p->d_parts[k])
            (void)cached_free(p->d_parts[k]);
        if (p->d_items[k])
            (void)cached_free(p->d_items[k]);
    }
    if (p->d_tiles)
        (void)cached_free(p->d_tiles);
    if (p->d_pairs)
        (void)cached_free(p->d_pairs);
    if (p->d_gsegs)
        (void)cached_free(p->d_gsegs);
    if (p->d_gitems)
        (void)cached_free(p->d_gitems);
    if (p->d_gtiles)
        (void)cached_free(p->d_gtiles);
    if (p->d_sum_work)
        (void)cached_free(p->d_sum_work);
    if (p->d_sum_entries)
        (void)cached_free(p->d_sum_entries);
    if (p->aux_stream)
        (void)hipStreamDestroy(p->aux_stream);
    if (p->ev_fork)
        (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join)
        (void)hipEventDestroy(p->ev_join);
    if (p->cap_stream)
        (void)hipStreamDestroy(p->cap_stream);
    delete p;
}

static thread_local size_t t_upload_bytes = 0; // bytes uploaded by upload() since the caller cleared it
template <typename T> static int upload(T **dst, const std::vector<T> &src) {
    if (src.empty())
        return B2X_OK;
    t_upload_bytes += src.size() * sizeof(T);
    if (cached_alloc((void **)dst, src.size() * sizeof(T)) != hipSuccess) // (freed with cached_free: the lists of the next plan reuse them)
        return fail(B2X_ERR_NOMEM, "hipMalloc(work lists): out of device memory");
    HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return B2X_OK;
}

// bind a plan (fresh, or taken from the plan cache) to an arena: scratch and slab buffers from the pool, scratch zeroed,
// arena-sourced staged operands copied
static int plan_bind(b2x_plan *p, const b2x_arena *arena) {
    p->arena = arena;
    if (p->fallback)
        return B2X_OK;
    if (p->scratch_elems) {
        hipError_t e = pool_alloc((void **)&p->d_scratch, (p->scratch_elems + kSlackElems) * sizeof(double), &p->scratch_bytes);
        if (e != hipSuccess)
            return fail(B2X_ERR_NOMEM, std::string("hipMalloc(W scratch): ") + hipGetErrorString(e));
        // the scratch only ever holds finite values: zeroed here, written by the kernels with products of the inputs
        // (the padding element behind an odd-sized W slot is never written and stays zero)
        if ((e = fill_tail(p->d_scratch, p->scratch_elems + kSlackElems)) != hipSuccess)
            return fail(B2X_ERR_DEVICE, std::string("hipMemset(W scratch): ") + hipGetErrorString(e));
        if (poison()) // the padding elements are zero in production; keep them so under the knob
            for (uint64_t pad : p->scratch_pads)
                if ((e = hipMemset(p->d_scratch + pad, 0, sizeof(double))) != hipSuccess)
                    return fail(B2X_ERR_DEVICE, std::string("hipMemset(pad): ") + hipGetErrorString(e));
        for (const StageCopy &sc : p->stage_arena) // staged operands with an arena source are copied now
            if ((e = hipMemcpy(p->d_scratch + sc.dst_off, arena->dev + sc.src_off, sc.len * sizeof(double),
                               hipMemcpyDeviceToDevice)) != hipSuccess)
                return fail(B2X_ERR_DEVICE, std::string("staging an operand: ") + hipGetErrorString(e));
    }
    if (p->gslab_elems) {
        hipError_t e = pool_alloc((void **)&p->d_gslabs, p->gslab_elems * sizeof(double), &p->gslabs_bytes);
        if (e != hipSuccess)
            return fail(B2X_ERR_NOMEM, std::string("hipMalloc(slabs): ") + hipGetErrorString(e));
    }
    if (p->slab_elems) {
        hipError_t e = pool_alloc((void **)&p->d_slabs, p->slab_elems * sizeof(double), &p->slabs_bytes);
        if (e != hipSuccess)
            return fail(B2X_ERR_NOMEM, std::string("hipMalloc(slabs): ") + hipGetErrorString(e));
    }
    return B2X_OK;
}

// ---- cache of compiled plans --------------------------------------------------------------------------------------
// A DMRG calculation visits the same sites sweep after sweep, and once the bond dimensions have settled the plan of a site
// is THE SAME record list every time (same dimensions, same offsets; only the operator data differ).  The host plan
// compiler is the largest single cost of a site below M ~ 1000 (88 ms against 10 x 0.62 ms of H.psi at M=250, DESIGN.md
// 4d), so a destroyed plan is kept — its work lists stay in HBM, its scratch and slabs go back to the buffer pool — and
// a later b2x_plan_create / b2x_gemm_plan_create with byte-identical records, lengths, options and arena extent takes it
// back and only binds it to the new arena.  LRU, B2X_PLAN_CACHE_MB of work lists in total (default 8192; 0 disables).
namespace {
std::mutex g_cache_mu;
std::vector<b2x_plan *> g_plan_cache; // most recently destroyed last
size_t g_cache_bytes = 0;
uint64_t g_cache_hits = 0, g_cache_misses = 0;
size_t cache_cap_bytes() {
    static const size_t cap = (size_t)(getenv("B2X_PLAN_CACHE_MB") ? atoll(getenv("B2X_PLAN_CACHE_MB")) : 8192) << 20;
    return cap;
}
void make_key(b2x_plan::Key &k, int kind, const void *recs, size_t n, size_t rec_bytes, size_t in_len, size_t out_len,
              const b2x_arena *arena, const b2x_plan_options *opt, bool with_blob) {
    k.kind = kind, k.n = n, k.in_len = in_len, k.out_len = out_len, k.arena_len = arena->len, k.arena_cap = arena->cap;
    k.opt = b2x_plan_options{};
    if (opt)
        k.opt = *opt;
    uint64_t h1 = 0x9E3779B97F4A7C15ull, h2 = 0xC2B2AE3D27D4EB4Full;
    const uint64_t *w = (const uint64_t *)recs; // (records are 8-byte aligned structs whose size is a multiple of 8)
    const size_t nw = n * rec_bytes / 8;
    for (size_t i = 0; i < nw; i++) {
        h1 = (h1 ^ w[i]) * 0xBF58476D1CE4E5B9ull, h1 ^= h1 >> 31;
        h2 = (h2 + w[i]) * 0x94D049BB133111EBull, h2 ^= h2 >> 29;
    }
    k.h1 = h1, k.h2 = h2;
    if (with_blob)
        k.blob.assign((const unsigned char *)recs, (const unsigned char *)recs + n * rec_bytes);
}
bool key_match(const b2x_plan::Key &a, const b2x_plan::Key &b, const void *recs) {
    return a.h1 == b.h1 && a.h2 == b.h2 && a.kind == b.kind && a.n == b.n && a.in_len == b.in_len && a.out_len == b.out_len &&
           a.arena_len == b.arena_len && a.arena_cap == b.arena_cap && memcmp(&a.opt, &b.opt, sizeof(b2x_plan_options)) == 0 &&
           memcmp(a.blob.data(), recs, a.blob.size()) == 0;
}
b2x_plan *cache_take(const b2x_plan::Key &k, const void *recs) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (size_t i = g_plan_cache.size(); i-- > 0;)
        if (key_match(g_plan_cache[i]->key, k, recs)) {
            b2x_plan *p = g_plan_cache[i];
            g_plan_cache.erase(g_plan_cache.begin() + i);
            g_cache_bytes -= p->meta_bytes;
            g_cache_hits++;
            return p;
        }
    g_cache_misses++;
    return nullptr;
}
} // namespace
static size_t reclaim_cached_plans() {
    std::vector<b2x_plan *> all;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        all.swap(g_plan_cache);
        g_cache_bytes = 0;
    }
    for (b2x_plan *q : all)
        plan_free(q);
    return all.size();
}

// upload a compiled plan (work lists + scratch / slab buffers) and hand it out
static int plan_upload(b2x_plan **out, const b2x_arena *arena, const CompiledPlan &cp, size_t n_pairs, const b2x_pair *pairs,
                       size_t psi_len, size_t sigma_len, const b2x_plan_options *opt) {
    int rc = B2X_OK;
    t_upload_bytes = 0;
    b2x_plan *p = new b2x_plan();
    p->arena = arena, p->stats = cp.stats, p->psi_len = psi_len, p->sigma_len = sigma_len;
    p->kernel = opt ? opt->kernel : 0;
    p->fallback = pairs != nullptr && (cp.fallback || p->kernel == 1);
    p->n_pairs = (uint32_t)n_pairs;
    p->seg_scaled = cp.seg_scaled;
    p->short_narrow = cp.short_narrow;
    p->short_frags = cp.short_frags;
    p->gg_tile_n = cp.gg_tile_n;
    if (p->fallback) {
        std::vector<b2x_pair> pv(pairs, pairs + n_pairs);
        rc = upload(&p->d_pairs, pv);
    } else {
        for (int k = 0; k < kNumClasses && rc == B2X_OK; k++) {
            rc = upload(&p->d_parts[k], cp.cls[k].parts);
            if (rc == B2X_OK)
                rc = upload(&p->d_items[k], cp.cls[k].items);
            p->n_items[k] = (uint32_t)cp.cls[k].items.size();
        }
        p->dominant_cls = (int)cp.stats.dominant_class;
        if (rc == B2X_OK)
            rc = upload(&p->d_tiles, cp.tiles);
        p->n_tiles = (uint32_t)cp.tiles.size();
        if (rc == B2X_OK)
            rc = upload(&p->d_gsegs, cp.gsegs);
        if (rc == B2X_OK)
            rc = upload(&p->d_gitems, cp.gitems);
        if (rc == B2X_OK)
            rc = upload(&p->d_gtiles, cp.gtiles);
        if (rc == B2X_OK)
            rc = upload(&p->d_sum_work, cp.sum_work);
        if (rc == B2X_OK)
            rc = upload(&p->d_sum_entries, cp.sum_entries);
        p->steps = cp.steps;
        for (const SuperStep &ss : cp.steps) { // largest tile of every step's reduce (launch_reduce)
            uint32_t mx = 0, mi = 0;
            for (uint32_t ti = ss.tile_begin; ti < ss.tile_end; ti++) {
                mx = std::max(mx, (uint32_t)cp.gtiles[ti].rows * (uint32_t)cp.gtiles[ti].cols);
                mi = std::max(mi, (uint32_t)cp.gtiles[ti].n_items);
            }
            p->step_max_elems.push_back(mx);
            p->step_max_items.push_back(mi);
        }
        p->scratch_elems = cp.scratch_elems, p->gslab_elems = cp.gslab_elems, p->slab_elems = cp.slab_elems;
        p->scratch_pads = cp.scratch_pads;
        for (const StageCopy &sc : cp.stage)
            (sc.src == 0 ? p->stage_arena : p->stage_in).push_back(sc);
        p->meta_bytes = t_upload_bytes;
        p->cacheable = cp.aux_work.empty(); // (operator pre-sums live in the scratch and depend on the operator data)
        if (rc == B2X_OK)
            rc = plan_bind(p, arena);
        if (rc == B2X_OK && !cp.aux_work.empty()) { // operator pre-sums: formed once, into the persistent head of the scratch
            OWork *dw = nullptr;
            OEntry *de = nullptr;
            rc = upload(&dw, cp.aux_work);
            if (rc == B2X_OK)
                rc = upload(&de, cp.aux_entries);
            if (rc == B2X_OK) {
                hipError_t e = launch_outer(dw, (uint32_t)cp.aux_work.size(), de, arena->dev, p->d_scratch, p->d_scratch, 16, nullptr);
                if (e == hipSuccess)
                    e = hipDeviceSynchronize();
                if (e != hipSuccess)
                    rc = fail(B2X_ERR_DEVICE, std::string("operator pre-sums: ") + hipGetErrorString(e));
            }
            if (dw)
                (void)cached_free(dw);
            if (de)
                (void)cached_free(de);
        }
    }
    if (rc != B2X_OK) {
        plan_free(p);
        return rc;
    }
    *out = p;
    return B2X_OK;
}


extern "C" {

const char *b2x_last_error(void) { return g_err.c_str(); }
const char *b2x_version(void) { return "b2x 0.1 (gfx950)"; }

int b2x_device_count(int *n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess)
        c = 0;
    *n = c;
    return B2X_OK;
}

int b2x_device_init(int ordinal) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c == 0)
        return fail(B2X_ERR_DEVICE, "no HIP device visible: the H.psi path has no CPU fallback");
    if (ordinal < 0 || ordinal >= c)
        return fail(B2X_ERR_INVALID, "device ordinal out of range");
    HIPCHK(hipSetDevice(ordinal));
    // Another thread of the process may be capturing a plan into a HIP graph (thread-local mode) while this one copies or
    // allocates — the sweep loop's helper threads do.  A thread in the default (global) capture-interaction mode is refused
    // such calls while ANY capture is open ("would make the legacy stream depend on a capturing blocking stream"): every
    // thread that initialises the library looks at its own captures only.
    {
        hipStreamCaptureMode mode = hipStreamCaptureModeThreadLocal;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        (void)hipGetLastError();
    }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, ordinal));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(B2X_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    return B2X_OK;
}

int b2x_device_sync(void) {
    HIPCHK(hipDeviceSynchronize());
    return B2X_OK;
}
int b2x_device_alloc(void **dptr, size_t bytes) {
    if (!dptr)
        return fail(B2X_ERR_INVALID, "b2x_device_alloc: null argument");
    if (cached_alloc(dptr, bytes) != hipSuccess)
        return fail(B2X_ERR_NOMEM, "b2x_device_alloc: out of device memory");
    return B2X_OK;
}
int b2x_trim(uint64_t *bytes_released) {
    size_t f0 = 0, f1 = 0, tot = 0;
    (void)hipMemGetInfo(&f0, &tot);
    reclaim_cached_plans();
    vec_cache_flush();
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool_trim_locked(0);
    }
    (void)hipMemGetInfo(&f1, &tot);
    (void)hipGetLastError();
    if (bytes_released)
        *bytes_released = f1 > f0 ? (uint64_t)(f1 - f0) : 0;
    return B2X_OK;
}
int b2x_device_free(void *dptr) {
    HIPCHK(cached_free(dptr));
    return B2X_OK;
}
int b2x_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return B2X_OK;
}
int b2x_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return B2X_OK;
}

// ---------------------------------------------------------------------------------- arena
int b2x_arena_create(b2x_arena **out, size_t n_ranges, const double *const *host_bases, const size_t *lens) {
    if (!out || (n_ranges && (!host_bases || !lens)))
        return fail(B2X_ERR_INVALID, "b2x_arena_create: null argument");
    std::vector<size_t> order(n_ranges);
    for (size_t i = 0; i < n_ranges; i++)
        order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return host_bases[a] < host_bases[b]; });
    b2x_arena *a = new b2x_arena();
    uint64_t tot = 0;
    for (size_t k = 0; k < n_ranges; k++) {
        size_t i = order[k];
        if (k > 0 && host_bases[i] < a->host_bases.back() + a->host_lens.back()) {
            delete a;
            return fail(B2X_ERR_INVALID, "b2x_arena_create: host ranges overlap");
        }
        a->host_bases.push_back(host_bases[i]);
        a->host_lens.push_back(lens[i]);
        a->offs.push_back(tot);
        tot += lens[i];
    }
    a->len = tot, a->cap = tot + kSlackElems, a->owned = true;
    hipError_t e = cached_alloc((void **)&a->dev, a->cap * sizeof(double));
    if (e == hipSuccess)
        e = fill_tail(a->dev + tot, kSlackElems);
    if (e != hipSuccess) {
        delete a;
        return fail(B2X_ERR_NOMEM, std::string("hipMalloc(arena): ") + hipGetErrorString(e));
    }
    for (size_t k = 0; k < n_ranges; k++) {
        e = hipMemcpy(a->dev + a->offs[k], a->host_bases[k], a->host_lens[k] * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)cached_free(a->dev);
            delete a;
            return fail(B2X_ERR_DEVICE, std::string("hipMemcpy(arena): ") + hipGetErrorString(e));
        }
    }
    *out = a;
    return B2X_OK;
}

int b2x_arena_adopt_device(b2x_arena **out, double *dev_base, size_t len) {
    if (!out || (!dev_base && len))
        return fail(B2X_ERR_INVALID, "b2x_arena_adopt_device: null argument");
    b2x_arena *a = new b2x_arena();
    a->dev = dev_base, a->len = len, a->cap = len, a->owned = false; // exactly len elements are ever read
    *out = a;
    return B2X_OK;
}

int b2x_arena_resolve(const b2x_arena *a, const double *host_ptr, uint64_t *off) {
    if (!a || !off)
        return fail(B2X_ERR_INVALID, "b2x_arena_resolve: null argument");
    auto it = std::upper_bound(a->host_bases.begin(), a->host_bases.end(), host_ptr);
    if (it == a->host_bases.begin())
        return fail(B2X_ERR_INVALID, "b2x_arena_resolve: pointer precedes every registered range");
    size_t k = (size_t)(it - a->host_bases.begin()) - 1;
    uint64_t d = (uint64_t)(host_ptr - a->host_bases[k]);
    if (d >= a->host_lens[k])
        return fail(B2X_ERR_INVALID, "b2x_arena_resolve: pointer is not inside a registered range");
    *off = a->offs[k] + d;
    return B2X_OK;
}

int b2x_arena_len(const b2x_arena *a, uint64_t *len) {
    if (!a || !len)
        return fail(B2X_ERR_INVALID, "b2x_arena_len: null argument");
    *len = a->len;
    return B2X_OK;
}
int b2x_arena_device_ptr(const b2x_arena *a, double **dev_base) {
    if (!a || !dev_base)
        return fail(B2X_ERR_INVALID, "b2x_arena_device_ptr: null argument");
    *dev_base = a->dev;
    return B2X_OK;
}
int b2x_arena_destroy(b2x_arena *a) {
    if (!a)
        return B2X_OK;
    if (a->owned && a->dev)
        (void)cached_free(a->dev);
    delete a;
    return B2X_OK;
}

// ---------------------------------------------------------------------------------- plan
int b2x_plan_create(b2x_plan **out, const b2x_arena *arena, size_t n_pairs, const b2x_pair *pairs, size_t psi_len,
                    size_t sigma_len, const b2x_plan_options *opt) {
    if (!out || !arena || (n_pairs && !pairs))
        return fail(B2X_ERR_INVALID, "b2x_plan_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(B2X_ERR_DEVICE, "b2x_plan_create: no HIP device (the H.psi path has no CPU fallback)");
    const bool use_cache = cache_cap_bytes() > 0 && n_pairs > 0;
    b2x_plan::Key key;
    if (use_cache) {
        make_key(key, 0, pairs, n_pairs, sizeof(b2x_pair), psi_len, sigma_len, arena, opt, false);
        if (b2x_plan *hit = cache_take(key, pairs)) {
            int rcb = plan_bind(hit, arena);
            if (rcb != B2X_OK) {
                plan_free(hit);
                return rcb;
            }
            *out = hit;
            return B2X_OK;
        }
    }
    const bool dbg = getenv("B2X_PLAN_DEBUG") != nullptr; // (development aid: host time of the steps, on stderr)
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    CompiledPlan cp;
    std::string err;
    int rc = compile_plan(n_pairs, pairs, psi_len, sigma_len, arena->len, arena->cap, opt, cp, err);
    if (rc != B2X_OK)
        return fail(rc, "b2x_plan_create: " + err);
    const auto t1 = now();
    rc = plan_upload(out, arena, cp, n_pairs, pairs, psi_len, sigma_len, opt);
    const auto t2 = now();
    if (rc == B2X_OK && use_cache && (*out)->cacheable && !(*out)->fallback) { // (the hashes of the lookup above; the records are kept for the byte comparison of a later hit)
        (*out)->key = key;
        (*out)->key.blob.assign((const unsigned char *)pairs, (const unsigned char *)pairs + n_pairs * sizeof(b2x_pair));
    }
    if (dbg)
        fprintf(stderr, "[b2x plan] create: compile %.2f ms, upload + bind %.2f ms, cache key %.2f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, now()));
    return rc;
}

int b2x_gemm_plan_create(b2x_plan **out, const b2x_arena *arena, size_t n_gemms, const b2x_gemm *gemms, size_t in_len,
                         size_t out_len, const b2x_plan_options *opt) {
    if (!out || !arena || (n_gemms && !gemms))
        return fail(B2X_ERR_INVALID, "b2x_gemm_plan_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(B2X_ERR_DEVICE, "b2x_gemm_plan_create: no HIP device (this path has no CPU fallback)");
    const bool use_cache = cache_cap_bytes() > 0 && n_gemms > 0;
    b2x_plan::Key key;
    if (use_cache) {
        make_key(key, 1, gemms, n_gemms, sizeof(b2x_gemm), in_len, out_len, arena, opt, false);
        if (b2x_plan *hit = cache_take(key, gemms)) {
            int rcb = plan_bind(hit, arena);
            if (rcb != B2X_OK) {
                plan_free(hit);
                return rcb;
            }
            *out = hit;
            return B2X_OK;
        }
    }
    CompiledPlan cp;
    std::string err;
    int rc = compile_gemm_list(n_gemms, gemms, in_len, out_len, arena->len, arena->cap, opt, cp, err);
    if (rc != B2X_OK)
        return fail(rc, "b2x_gemm_plan_create: " + err);
    rc = plan_upload(out, arena, cp, 0, nullptr, in_len, out_len, nullptr);
    if (rc == B2X_OK && use_cache && (*out)->cacheable) {
        (*out)->key = key;
        (*out)->key.blob.assign((const unsigned char *)gemms, (const unsigned char *)gemms + n_gemms * sizeof(b2x_gemm));
    }
    return rc;
}

// one stage of a super-step: both tile classes, the short one on the plan's own stream beside the tall one
static int launch_stage(b2x_plan *p, const uint32_t *v, const double *psi, hipStream_t st) {
    const bool both = v[1] > v[0] && v[kGGVariants] > v[1];
    if (!both) {
        HIPCHK(launch_gg(p->d_gsegs, p->d_gitems, v, p->arena->dev, psi, p->d_scratch, p->d_gslabs, p->seg_scaled, p->gg_tile_n, st, 0, p->short_narrow, p->short_frags));
        return B2X_OK;
    }
    if (!p->aux_stream) {
        HIPCHK(hipStreamCreateWithFlags(&p->aux_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(p->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(p->aux_stream, p->ev_fork, 0));
    HIPCHK(launch_gg(p->d_gsegs, p->d_gitems, v, p->arena->dev, psi, p->d_scratch, p->d_gslabs, p->seg_scaled, p->gg_tile_n, p->aux_stream, 2, p->short_narrow, p->short_frags));
    HIPCHK(hipEventRecord(p->ev_join, p->aux_stream));
    HIPCHK(launch_gg(p->d_gsegs, p->d_gitems, v, p->arena->dev, psi, p->d_scratch, p->d_gslabs, p->seg_scaled, p->gg_tile_n, st, 1, p->short_narrow, p->short_frags));
    HIPCHK(hipStreamWaitEvent(st, p->ev_join, 0));
    return B2X_OK;
}

static int run_plan(b2x_plan *p, const double *psi, double *sigma, double scale, hipStream_t st) {
    if (p->fallback) {
        HIPCHK(launch_generic(p->d_pairs, p->n_pairs, p->arena->dev, psi, sigma, scale, st));
        return B2X_OK;
    }
    for (int k = 0; k < kNumClasses; k++)
        HIPCHK(launch_main(k, p->d_parts[k], p->d_items[k], p->n_items[k], p->arena->dev, psi, p->d_slabs, st));
    HIPCHK(launch_reduce(p->d_tiles, p->n_tiles, p->d_slabs, sigma, scale, st));
    for (const StageCopy &sc : p->stage_in) // (degenerate operands at the very end of the input vector; normally none)
        HIPCHK(hipMemcpyAsync(p->d_scratch + sc.dst_off, psi + sc.src_off, sc.len * sizeof(double), hipMemcpyDeviceToDevice, st));
    for (size_t si = 0; si < p->steps.size(); si++) {
        const SuperStep &ss = p->steps[si];
        int rc = launch_stage(p, ss.s0_v, psi, st);
        if (rc != B2X_OK)
            return rc;
        if (ss.sum_end > ss.sum_begin)
            HIPCHK(launch_outer(p->d_sum_work + ss.sum_begin, ss.sum_end - ss.sum_begin, p->d_sum_entries, p->arena->dev,
                                p->d_scratch, p->d_scratch, 16, st));
        if ((rc = launch_stage(p, ss.s1_v, psi, st)) != B2X_OK)
            return rc;
        HIPCHK(launch_reduce(p->d_gtiles + ss.tile_begin, ss.tile_end - ss.tile_begin, p->d_gslabs, sigma, scale, st,
                             si < p->step_max_elems.size() ? p->step_max_elems[si] : 0,
                             si < p->step_max_items.size() ? p->step_max_items[si] : 0));
    }
    return B2X_OK;
}

// run_plan through a HIP graph (see b2x_plan::graph).  Returns B2X_OK after a graph launch, a negative value when the
// graph path is not available for this plan (the caller launches directly).
static const int kGraphUnavailable = -1000;
static int run_plan_graph(b2x_plan *p, const double *psi, double *sigma, double scale, hipStream_t st) {
    const char *genv = getenv("B2X_GRAPH"); // (read per call: the tests switch it inside one process)
    const int enabled = genv ? atoi(genv) : 1;
    if (!enabled || p->graph_failed || p->fallback || !p->stage_in.empty())
        return kGraphUnavailable;
    // The first device-pointer execute of a plan launches directly; the graph is captured at the second: a plan that runs
    // once (the rotation of a block, 16 ms of capture + instantiation against 1 ms of work at M=250) never pays for it.
    if (!p->gexec && p->n_device_exec++ == 0)
        return kGraphUnavailable;
    auto give_up = [&]() {
        p->graph_failed = true;
        (void)hipGetLastError();
        return kGraphUnavailable;
    };
    if (!p->gexec) {
        if (hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking) != hipSuccess)
            return give_up();
        if (!p->aux_stream) { // (created outside the capture: launch_stage would create them inside it)
            HIPCHK(hipStreamCreateWithFlags(&p->aux_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
        }
        if (hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess)
            return give_up();
        int rc = run_plan(p, psi, sigma, scale, p->cap_stream);
        hipError_t e = hipStreamEndCapture(p->cap_stream, &p->graph);
        if (rc != B2X_OK || e != hipSuccess || !p->graph)
            return give_up();
        if (hipGraphInstantiate(&p->gexec, p->graph, nullptr, nullptr, 0) != hipSuccess) {
            p->gexec = nullptr;
            return give_up();
        }
        size_t nn = 0;
        if (hipGraphGetNodes(p->graph, nullptr, &nn) != hipSuccess)
            return give_up();
        std::vector<hipGraphNode_t> nodes(nn);
        if (nn && hipGraphGetNodes(p->graph, nodes.data(), &nn) != hipSuccess)
            return give_up();
        for (hipGraphNode_t nd : nodes) {
            hipGraphNodeType ty;
            if (hipGraphNodeGetType(nd, &ty) != hipSuccess)
                return give_up();
            if (ty != hipGraphNodeTypeKernel)
                continue;
            b2x_plan::GNode g{};
            g.node = nd, g.psi_slot = g.sigma_slot = g.scale_slot = -1;
            if (hipGraphKernelNodeGetParams(nd, &g.params) != hipSuccess || !g.params.kernelParams)
                return give_up();
            // which arguments are psi / sigma / scale is recorded by the launcher of each kernel (b2x_kernels.hip,
            // note_slots): psi is argument 3 of gg_kernel and argument 4 of hpsi_wave, (sigma, scale) are arguments 2
            // and 3 of hpsi_reduce, the sum pass has neither.  A kernel the launchers do not know, or a slot that does
            // not hold the captured pointer, switches the graph off for this plan: a replay must never run on stale
            // arguments.
            KernelArgSlots sl;
            if (!kernel_arg_slots(g.params.func, &sl))
                return give_up();
            void **kp = g.params.kernelParams;
            if (sl.psi >= 0) {
                if (*(const double **)kp[sl.psi] != psi)
                    return give_up();
                g.psi_slot = sl.psi;
            }
            if (sl.sigma >= 0) {
                if (*(double **)kp[sl.sigma] != sigma || *(double *)kp[sl.scale] != scale)
                    return give_up();
                g.sigma_slot = sl.sigma, g.scale_slot = sl.scale;
            }
            p->gnodes.push_back(g);
        }
        p->g_psi = psi, p->g_sigma = sigma, p->g_scale = scale;
    }
    if (psi != p->g_psi || sigma != p->g_sigma || scale != p->g_scale) {
        for (b2x_plan::GNode &g : p->gnodes) {
            if (g.psi_slot < 0 && g.sigma_slot < 0)
                continue;
            void **kp = g.params.kernelParams; // the graph's own argument copies: patched in place, then re-submitted
            if (g.psi_slot >= 0)
                *(const double **)kp[g.psi_slot] = psi;
            if (g.sigma_slot >= 0)
                *(double **)kp[g.sigma_slot] = sigma, *(double *)kp[g.scale_slot] = scale;
            if (hipGraphExecKernelNodeSetParams(p->gexec, g.node, &g.params) != hipSuccess)
                return give_up();
        }
        p->g_psi = psi, p->g_sigma = sigma, p->g_scale = scale;
    }
    HIPCHK(hipGraphLaunch(p->gexec, st));
    return B2X_OK;
}

int b2x_plan_execute(b2x_plan *p, const double *psi, double *sigma, double scale, int on_device, void *stream) {
    if (!p || !psi || !sigma)
        return fail(B2X_ERR_INVALID, "b2x_plan_execute: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (on_device) {
        const int rc = run_plan_graph(p, psi, sigma, scale, st);
        return rc == kGraphUnavailable ? run_plan(p, psi, sigma, scale, st) : rc;
    }
    if (!p->d_psi)
        if (dev_malloc((void **)&p->d_psi, (p->psi_len ? p->psi_len : 1) * sizeof(double)) != hipSuccess)
            return fail(B2X_ERR_NOMEM, "b2x_plan_execute: out of device memory (psi staging)");
    if (!p->d_sigma)
        if (dev_malloc((void **)&p->d_sigma, (p->sigma_len ? p->sigma_len : 1) * sizeof(double)) != hipSuccess)
            return fail(B2X_ERR_NOMEM, "b2x_plan_execute: out of device memory (sigma staging)");
    HIPCHK(hipMemcpyAsync(p->d_psi, psi, p->psi_len * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->d_sigma, sigma, p->sigma_len * sizeof(double), hipMemcpyHostToDevice, st));
    int rc = run_plan(p, p->d_psi, p->d_sigma, scale, st);
    if (rc != B2X_OK)
        return rc;
    HIPCHK(hipMemcpyAsync(sigma, p->d_sigma, p->sigma_len * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return B2X_OK;
}

int b2x_plan_get_stats(const b2x_plan *p, b2x_plan_stats *st) {
    if (!p || !st)
        return fail(B2X_ERR_INVALID, "b2x_plan_get_stats: null argument");
    *st = p->stats;
    return B2X_OK;
}

int b2x_plan_time_kernel(b2x_plan *p, const double *psi_dev, double *sigma_dev, int n, void *stream,
                         double *avg_ms_main, double *avg_ms_total) {
    if (!p || !psi_dev || !sigma_dev || n <= 0)
        return fail(B2X_ERR_INVALID, "b2x_plan_time_kernel: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1, e2;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventCreate(&e2));
    double tm = 0, tt = 0;
    for (int i = 0; i < n; i++) {
        float a = 0, b = 0;
        if (p->fallback) {
            HIPCHK(hipEventRecord(e0, st));
            HIPCHK(launch_generic(p->d_pairs, p->n_pairs, p->arena->dev, psi_dev, sigma_dev, 1.0, st));
            HIPCHK(hipEventRecord(e1, st));
            HIPCHK(hipEventSynchronize(e1));
            HIPCHK(hipEventElapsedTime(&a, e0, e1));
            b = a;
        } else if (p->dominant_cls >= kNumClasses) {
            // two-stage plan: the grouped-GEMM launches (gg_kernel, both stages of every super-step) are
            // bracketed together; fused classes + reduces make up the rest of the total
            HIPCHK(hipEventRecord(e0, st));
            for (const SuperStep &ss : p->steps) {
                {
                    int rc0 = launch_stage(p, ss.s0_v, psi_dev, st);
                    if (rc0 != B2X_OK)
                        return rc0;
                }
                if (ss.sum_end > ss.sum_begin)
                    HIPCHK(launch_outer(p->d_sum_work + ss.sum_begin, ss.sum_end - ss.sum_begin, p->d_sum_entries,
                                        p->arena->dev, p->d_scratch, p->d_scratch, 16, st));
                {
                    int rc1 = launch_stage(p, ss.s1_v, psi_dev, st);
                    if (rc1 != B2X_OK)
                        return rc1;
                }
            }
            HIPCHK(hipEventRecord(e1, st));
            HIPCHK(hipEventSynchronize(e1));
            HIPCHK(hipEventElapsedTime(&a, e0, e1));
            HIPCHK(hipEventRecord(e0, st));
            int rc2 = run_plan(p, psi_dev, sigma_dev, 1.0, st);
            if (rc2 != B2X_OK)
                return rc2;
            HIPCHK(hipEventRecord(e2, st));
            HIPCHK(hipEventSynchronize(e2));
            HIPCHK(hipEventElapsedTime(&b, e0, e2));
        } else {
            // the dominant class is bracketed on its own; the rest + reduce make up the total
            int d = p->dominant_cls;
            HIPCHK(hipEventRecord(e0, st));
            HIPCHK(launch_main(d, p->d_parts[d], p->d_items[d], p->n_items[d], p->arena->dev, psi_dev, p->d_slabs, st));
            HIPCHK(hipEventRecord(e1, st));
            for (int k = 0; k < kNumClasses; k++)
                if (k != d)
                    HIPCHK(launch_main(k, p->d_parts[k], p->d_items[k], p->n_items[k], p->arena->dev, psi_dev,
                                       p->d_slabs, st));
            HIPCHK(launch_reduce(p->d_tiles, p->n_tiles, p->d_slabs, sigma_dev, 1.0, st));
            HIPCHK(hipEventRecord(e2, st));
            HIPCHK(hipEventSynchronize(e2));
            HIPCHK(hipEventElapsedTime(&a, e0, e1));
            HIPCHK(hipEventElapsedTime(&b, e0, e2));
        }
        tm += a, tt += b;
    }
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1), (void)hipEventDestroy(e2);
    if (avg_ms_main)
        *avg_ms_main = tm / n;
    if (avg_ms_total)
        *avg_ms_total = tt / n;
    return B2X_OK;
}

int b2x_plan_destroy(b2x_plan *p) {
    if (!p)
        return B2X_OK;
    if (p->key.blob.empty() || p->meta_bytes > cache_cap_bytes()) { // not cacheable (or the cache is off)
        plan_free(p);
        return B2X_OK;
    }
    plan_unbind(p);
    std::vector<b2x_plan *> evict;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g_plan_cache.push_back(p);
        g_cache_bytes += p->meta_bytes;
        while (g_cache_bytes > cache_cap_bytes() && g_plan_cache.size() > 1) { // least recently destroyed first
            g_cache_bytes -= g_plan_cache.front()->meta_bytes;
            evict.push_back(g_plan_cache.front());
            g_plan_cache.erase(g_plan_cache.begin());
        }
    }
    for (b2x_plan *q : evict)
        plan_free(q);
    return B2X_OK;
}

int b2x_plan_cache_stats(uint64_t *hits, uint64_t *misses, uint64_t *plans, uint64_t *bytes) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (hits)
        *hits = g_cache_hits;
    if (misses)
        *misses = g_cache_misses;
    if (plans)
        *plans = g_plan_cache.size();
    if (bytes)
        *bytes = g_cache_bytes;
    return B2X_OK;
}

int b2x_plan_cache_clear(void) {
    std::vector<b2x_plan *> all;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        all.swap(g_plan_cache);
        g_cache_bytes = 0;
    }
    for (b2x_plan *q : all)
        plan_free(q);
    return B2X_OK;
}

// ---------------------------------------------------------------------------------- diagonal
int b2x_diag_build(const b2x_arena *arena, size_t n_terms, const b2x_diag_term *terms, size_t diag_len, double *diag,
                   int on_device, void *stream) {
    if (!arena || !diag || (n_terms && !terms))
        return fail(B2X_ERR_INVALID, "b2x_diag_build: null argument");
    std::vector<DiagComp> comps;
    std::vector<DiagTermD> dterms;
    std::string err;
    int rc = compile_diag(n_terms, terms, diag_len, arena->len, comps, dterms, err);
    if (rc != B2X_OK)
        return fail(rc, "b2x_diag_build: " + err);
    if (comps.empty())
        return B2X_OK;
    hipStream_t st = (hipStream_t)stream;
    DiagComp *dc = nullptr;
    DiagTermD *dt = nullptr;
    double *dd = diag;
    rc = upload(&dc, comps);
    if (rc == B2X_OK)
        rc = upload(&dt, dterms);
    if (rc == B2X_OK && !on_device) {
        hipError_t e = dev_malloc((void **)&dd, diag_len * sizeof(double));
        if (e == hipSuccess)
            e = hipMemcpy(dd, diag, diag_len * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess)
            rc = fail(B2X_ERR_DEVICE, std::string("b2x_diag_build: ") + hipGetErrorString(e));
    }
    if (rc == B2X_OK) {
        hipError_t e = launch_diag(dc, (uint32_t)comps.size(), dt, arena->dev, dd, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st); // metadata is freed below
        if (e == hipSuccess && !on_device)
            e = hipMemcpy(diag, dd, diag_len * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            rc = fail(B2X_ERR_DEVICE, std::string("b2x_diag_build: ") + hipGetErrorString(e));
    }
    if (dc)
        (void)cached_free(dc);
    if (dt)
        (void)cached_free(dt);
    if (!on_device && dd != diag && dd)
        (void)hipFree(dd);
    return rc;
}

// A term list compiled and uploaded once, executed later (and possibly on another thread's time: the sweep loop compiles the
// blocking of the NEXT site while the device iterates Davidson).  b2x_outer_build is create + execute + destroy.
struct b2x_outer_plan {
    OWork *dw = nullptr;
    OEntry *de = nullptr;
    uint32_t n_work = 0;
    uint64_t arena_len = 0;
    size_t in_len = 0, out_len = 0;
};
int b2x_outer_plan_create(b2x_outer_plan **out, uint64_t arena_len, size_t n_terms, const b2x_outer_term *terms, size_t in_len,
                          size_t out_len) {
    if (!out || (n_terms && !terms))
        return fail(B2X_ERR_INVALID, "b2x_outer_plan_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(B2X_ERR_DEVICE, "b2x_outer_plan_create: no HIP device (this path has no CPU fallback)");
    static thread_local std::vector<OWork> work; // (kept between calls: the lists of a blocking are tens of MB)
    static thread_local std::vector<OEntry> entries;
    std::string err;
    int rc = compile_outer(n_terms, terms, in_len, out_len, arena_len, work, entries, err);
    if (rc != B2X_OK)
        return fail(rc, "b2x_outer_plan_create: " + err);
    b2x_outer_plan *p = new b2x_outer_plan();
    p->arena_len = arena_len, p->in_len = in_len, p->out_len = out_len, p->n_work = (uint32_t)work.size();
    if (!work.empty()) {
        rc = upload(&p->dw, work);
        if (rc == B2X_OK)
            rc = upload(&p->de, entries);
        if (rc != B2X_OK) {
            (void)cached_free(p->dw), (void)cached_free(p->de);
            delete p;
            return rc;
        }
    }
    *out = p;
    return B2X_OK;
}
int b2x_outer_plan_execute(const b2x_outer_plan *p, const b2x_arena *arena, const double *in_dev, double *out_dev, void *stream) {
    if (!p || !arena || !out_dev || (p->in_len && !in_dev))
        return fail(B2X_ERR_INVALID, "b2x_outer_plan_execute: null argument");
    if (arena->len != p->arena_len)
        return fail(B2X_ERR_INVALID, "b2x_outer_plan_execute: the arena is not of the extent the list was compiled for");
    if (p->n_work == 0)
        return B2X_OK;
    HIPCHK(launch_outer(p->dw, p->n_work, p->de, arena->dev, in_dev, out_dev, 4, (hipStream_t)stream));
    return B2X_OK;
}
int b2x_outer_plan_destroy(b2x_outer_plan *p) {
    if (!p)
        return B2X_OK;
    (void)cached_free(p->dw), (void)cached_free(p->de); // (waits for the device: a running execute finishes first)
    delete p;
    return B2X_OK;
}

int b2x_outer_build(const b2x_arena *arena, size_t n_terms, const b2x_outer_term *terms, const double *in, size_t in_len,
                    size_t out_len, double *out, int on_device, void *stream) {
    if (!arena || !out || (n_terms && !terms) || (in_len && !in))
        return fail(B2X_ERR_INVALID, "b2x_outer_build: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(B2X_ERR_DEVICE, "b2x_outer_build: no HIP device (this path has no CPU fallback)");
    std::vector<OWork> work;
    std::vector<OEntry> entries;
    std::string err;
    static const bool dbg = getenv("B2X_PLAN_DEBUG") != nullptr; // (development aid: host clock of the three phases, on stderr)
    const auto t_0 = std::chrono::steady_clock::now();
    int rc = compile_outer(n_terms, terms, in_len, out_len, arena->len, work, entries, err);
    if (rc != B2X_OK)
        return fail(rc, "b2x_outer_build: " + err);
    if (work.empty())
        return B2X_OK;
    const auto t_1 = std::chrono::steady_clock::now();
    hipStream_t st = (hipStream_t)stream;
    OWork *dw = nullptr;
    OEntry *de = nullptr;
    double *d_in = const_cast<double *>(in), *d_out = out;
    rc = upload(&dw, work);
    if (rc == B2X_OK)
        rc = upload(&de, entries);
    if (rc == B2X_OK && !on_device) {
        hipError_t e = dev_malloc((void **)&d_out, out_len * sizeof(double));
        if (e == hipSuccess)
            e = hipMemcpy(d_out, out, out_len * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess && in_len) {
            e = dev_malloc((void **)&d_in, in_len * sizeof(double));
            if (e == hipSuccess)
                e = hipMemcpy(d_in, in, in_len * sizeof(double), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess)
            rc = fail(B2X_ERR_DEVICE, std::string("b2x_outer_build: ") + hipGetErrorString(e));
    }
    if (rc == B2X_OK) {
        hipError_t e = launch_outer(dw, (uint32_t)work.size(), de, arena->dev, d_in, d_out, 4, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st); // metadata is freed below
        if (e == hipSuccess && !on_device)
            e = hipMemcpy(out, d_out, out_len * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            rc = fail(B2X_ERR_DEVICE, std::string("b2x_outer_build: ") + hipGetErrorString(e));
    }
    if (dbg) {
        const auto t_2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[b2x outer] terms %zu -> work %zu entries %zu: compile %.3f ms, upload + kernel + wait %.3f ms\n", n_terms,
                work.size(), entries.size(), std::chrono::duration<double, std::milli>(t_1 - t_0).count(),
                std::chrono::duration<double, std::milli>(t_2 - t_1).count());
    }
    if (dw)
        (void)cached_free(dw);
    if (de)
        (void)cached_free(de);
    if (!on_device) {
        if (d_out && d_out != out)
            (void)hipFree(d_out);
        if (d_in && d_in != in)
            (void)hipFree(d_in);
    }
    return rc;
}

// ---------------------------------------------------------------------------------- vectors
static double *g_dot_partial = nullptr, *g_dot_out = nullptr;
static double *g_dot_host = nullptr; // pinned: the few result doubles of every dot product cross PCIe without a staging copy
static int dot_scratch() {
    if (!g_dot_partial) {
        HIPCHK(hipMalloc((void **)&g_dot_partial, 128 * 256 * sizeof(double)));
        HIPCHK(hipMalloc((void **)&g_dot_out, 128 * sizeof(double)));
        HIPCHK(hipHostMalloc((void **)&g_dot_host, 128 * sizeof(double), hipHostMallocDefault));
    }
    return B2X_OK;
}

int b2x_vec_multi_dot(const double *const *vs, int nv, const double *x, size_t n, double *host_result, void *stream) {
    if (nv < 1 || nv > 64 || !vs || !x || !host_result)
        return fail(B2X_ERR_INVALID, "b2x_vec_multi_dot: need 1 <= nv <= 64");
    int rc = dot_scratch();
    if (rc != B2X_OK)
        return rc;
    hipStream_t st = (hipStream_t)stream;
    // (the second stage writes its few results straight into the pinned, device-visible host buffer: no copy operation)
    HIPCHK(launch_multidot(vs, nv, x, n, g_dot_partial, g_dot_host, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(host_result, g_dot_host, nv * sizeof(double));
    return B2X_OK;
}
int b2x_vec_pair_dots(const double *const *us, const double *const *vs, int n_pairs, size_t n, double *host_result,
                      void *stream) {
    if (n_pairs < 1 || n_pairs > 128 || !us || !vs || !host_result)
        return fail(B2X_ERR_INVALID, "b2x_vec_pair_dots: need 1 <= n_pairs <= 128");
    int rc = dot_scratch();
    if (rc != B2X_OK)
        return rc;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(launch_pairdot(us, vs, n_pairs, n, g_dot_partial, g_dot_host, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(host_result, g_dot_host, n_pairs * sizeof(double));
    return B2X_OK;
}
int b2x_vec_ritz_olsen(const double *const *bs, const double *const *ss, int m, const double *alpha, double theta, const double *diag,
                       double *x, double *q, double *q2, double *t, size_t n, void *stream) {
    if (m < 1 || m > 64 || !bs || !ss || !alpha || !diag || !x || !q || !q2 || !t)
        return fail(B2X_ERR_INVALID, "b2x_vec_ritz_olsen: need 1 <= m <= 64 and non-null vectors");
    HIPCHK(launch_ritz_olsen(bs, ss, m, alpha, theta, diag, x, q, q2, t, n, (hipStream_t)stream));
    return B2X_OK;
}
// out = (v - sum_j <b_j, v> b_j) / |.|, everything on the device and asynchronous on `stream`; *status (pinned host memory owned by
// the library, see b2x_vec_gs_status) becomes 1 when the norm was not safely positive.
static int *g_gs_flag_host = nullptr; // pinned, device-visible: the kernel raises it in place (no copy operation)
int b2x_vec_gs_finish(const double *const *bs, int m, const double *v, double *out, size_t n, void *stream) {
    if (m < 0 || m > 63 || (m && !bs) || !v || !out)
        return fail(B2X_ERR_INVALID, "b2x_vec_gs_finish: need 0 <= m <= 63");
    int rc = dot_scratch();
    if (rc != B2X_OK)
        return rc;
    if (!g_gs_flag_host) {
        HIPCHK(hipHostMalloc((void **)&g_gs_flag_host, sizeof(int), hipHostMallocDefault));
        *g_gs_flag_host = 0;
    }
    HIPCHK(launch_gs_finish(bs, m, v, g_dot_partial, g_dot_out, out, n, g_gs_flag_host, (hipStream_t)stream));
    return B2X_OK;
}
int b2x_vec_gs_status(int *degenerate, int reset) {
    if (!degenerate)
        return fail(B2X_ERR_INVALID, "b2x_vec_gs_status: null argument");
    *degenerate = g_gs_flag_host ? *(volatile int *)g_gs_flag_host : 0;
    if (reset && g_gs_flag_host)
        *(volatile int *)g_gs_flag_host = 0; // (call after a wait on the stream: no finish kernel is in flight then)
    return B2X_OK;
}
int b2x_vec_dot(const double *x, const double *y, size_t n, double *host_result, void *stream) {
    const double *vs[1] = {x};
    return b2x_vec_multi_dot(vs, 1, y, n, host_result, stream);
}
int b2x_vec_axpy(double a, const double *x, double *y, size_t n, void *stream) {
    HIPCHK(launch_axpy(a, x, y, n, (hipStream_t)stream));
    return B2X_OK;
}
int b2x_vec_scal(double a, double *x, size_t n, void *stream) {
    HIPCHK(launch_scal(a, x, n, (hipStream_t)stream));
    return B2X_OK;
}
int b2x_vec_copy(const double *x, double *y, size_t n, void *stream) {
    HIPCHK(hipMemcpyAsync(y, x, n * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return B2X_OK;
}
int b2x_vec_gather(double *dst, const double *src, size_t n_ranges, const uint64_t *dst_off, const uint64_t *src_off,
                   const uint64_t *len, void *stream) {
    if (!n_ranges)
        return B2X_OK;
    if (!dst || !src || !dst_off || !src_off || !len)
        return fail(B2X_ERR_INVALID, "b2x_vec_gather: null argument");
    static const uint64_t kCopyPiece = 32768; // elements per workgroup
    struct D {
        uint64_t dst, src, len;
    };
    static thread_local std::vector<D> ds;
    ds.clear();
    for (size_t i = 0; i < n_ranges; i++)
        for (uint64_t o = 0; o < len[i]; o += kCopyPiece)
            ds.push_back(D{dst_off[i] + o, src_off[i] + o, std::min<uint64_t>(kCopyPiece, len[i] - o)});
    if (ds.empty())
        return B2X_OK;
    if (ds.size() > 0x7FFFFFFFull)
        return fail(B2X_ERR_INVALID, "b2x_vec_gather: too many pieces");
    void *dd = nullptr;
    if (cached_alloc(&dd, ds.size() * sizeof(D)) != hipSuccess)
        return fail(B2X_ERR_NOMEM, "b2x_vec_gather: out of device memory");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(dd, ds.data(), ds.size() * sizeof(D), hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = launch_gather(dd, (uint32_t)ds.size(), dst, src, st);
    if (e == hipSuccess)
        e = hipStreamSynchronize(st); // (the table is host memory of this call and a parked device vector afterwards)
    (void)cached_free(dd);
    if (e != hipSuccess)
        return fail(B2X_ERR_DEVICE, std::string("b2x_vec_gather: ") + hipGetErrorString(e));
    return B2X_OK;
}
int b2x_vec_zero(double *x, size_t n, void *stream) {
    HIPCHK(hipMemsetAsync(x, 0, n * sizeof(double), (hipStream_t)stream));
    return B2X_OK;
}
int b2x_vec_precondition(double *q, const double *diag, double shift, size_t n, void *stream) {
    HIPCHK(launch_precond(q, diag, shift, n, (hipStream_t)stream));
    return B2X_OK;
}
int b2x_vec_olsen_prepare(double *q, double *t, const double *c, const double *diag, double ld, size_t n, void *stream) {
    HIPCHK(launch_olsen(q, q, t, c, diag, ld, n, (hipStream_t)stream));
    return B2X_OK;
}
int b2x_vec_olsen_prepare_to(const double *q, double *q_out, double *t, const double *c, const double *diag, double ld, size_t n,
                             void *stream) {
    if (!q || !q_out || !t || !c || !diag)
        return fail(B2X_ERR_INVALID, "b2x_vec_olsen_prepare_to: null argument");
    HIPCHK(launch_olsen(q, q_out, t, c, diag, ld, n, (hipStream_t)stream));
    return B2X_OK;
}
int b2x_vec_lincomb(const double *const *vs, int nv, const double *coef, double *y, size_t n, void *stream) {
    if (nv < 1 || nv > 64)
        return fail(B2X_ERR_INVALID, "b2x_vec_lincomb: need 1 <= nv <= 64");
    HIPCHK(launch_lincomb(vs, coef, nv, y, n, (hipStream_t)stream));
    return B2X_OK;
}

} // extern "C"
