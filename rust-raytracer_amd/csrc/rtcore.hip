// C ABI of the MI355X ray-tracing core (see include/rtcore.h).
//
// Host side only: argument checks, scene upload, workspace, pass planning and
// kernel sequencing.  All device work is in kernels.hip.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <exception>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtcore.h"
#include "bvh_build.hpp"
#include "bvh_build_gpu.hpp"
#include "wide_bvh.hpp"
#include "device_layout.h"
#include "kernels.h"

using namespace rtc;

static_assert(RT_STAT_DEPTHS == kStatDepths, "rt_depth_stats mirrors the device counters");
static_assert(RT_LDS_STACK_ENTRIES == RTC_LDS_STACK_DEPTH, "rtcore.h documents the LDS part of the traversal stack");

// ------------------------------------------------------------------------
// errors

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                        __FILE__, __LINE__);                                                     \
    } while (0)

extern "C" const char *rt_last_error(void) { return g_err.c_str(); }
extern "C" const char *rt_version(void) { return "rtcore-gfx950 0.1"; }

// ------------------------------------------------------------------------
// context

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return RT_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) return fail(RT_ERR_NOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        bytes = need;
        return RT_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct TimedLaunch {
    hipEvent_t start, stop;
    int kind; // 0 extend, 1 connect, 2 other
};

struct Workspace {
    DevBuf rays[2], infos[2], hit_ids, hit_t, shadow_rays, shadow_infos, rad, counters;
    DevBuf keys[2], keys_sorted, perm[2], sort_temp; // [0] bounce queue, [1] shadow queue
    uint64_t paths = 0;
    void release_all()
    {
        for (DevBuf *b : {&rays[0], &rays[1], &infos[0], &infos[1], &hit_ids, &hit_t, &shadow_rays, &shadow_infos, &rad,
                          &counters, &keys[0], &keys[1], &keys_sorted, &perm[0], &perm[1], &sort_temp})
            b->release();
        paths = 0;
    }
};

} // namespace

constexpr uint32_t kCounterStride = 32;      // queue-length words sit in 128-byte lines of their own (the blocks of a
                                             // shade launch add to two of them at once)
constexpr uint32_t kMaxQueueCounters = 2048; // words (stride included) of one pass kept for the next pass's sort sizes

struct rt_ctx {
    int device = 0;
    int num_cus = 0;
    unsigned max_blocks = 0; // resident trace blocks
    hipStream_t stream = nullptr;
    DevBuf ovf;             // traversal stack overflow area: two halves, one per stream that runs trace kernels
    unsigned ovf_depth = 0; // entries per thread available in each half of ovf
    // 1: rt_render runs the any-hit launch of depth d on a stream of its own beside the closest-hit launch of
    // depth d + 1 (they share nothing but the scene), so that the tail of one persistent launch -- its last
    // rays finishing at low occupancy -- could be filled by the start of the other.  Measured on one GPU it is
    // 0.7 % slower on the full San-Miguel-class frame and 1 % slower on an eighth of it (the two kernels end up
    // sharing the CUs for most of their run instead of meeting at the tail), so it is off by default.
    unsigned overlap_connect = 0;
    hipStream_t conn_stream = nullptr;
    std::vector<hipEvent_t> ev_shade, ev_conn; // per depth: shade + sorts done (launch stream), connect done (conn_stream)
    DevBuf dstats;
    DevBuf fetch; // ray-fetch counters of the stream entry points
    DevBuf coop;  // wave-private work stacks of the cooperative straggler path
    unsigned coop_cap = 2048;    // entries per wave
    unsigned coop_steps = 1024;  // a ray is handed to its whole wave after this many loop iterations (4-wide forms)
    unsigned coop_steps_binary = 512; // the same for the forms on the binary tree, whose stragglers cost more
    // refill threshold [depth 0 of rt_render: 0, everything else: 1][closest-hit: 0, any-hit: 1].  The
    // depth-0 launches run the while-while forms on coherent rays, where a refill (for the fused
    // primary launch: making the camera rays) costs more than a few idle lanes do: 20 is 4 % faster than
    // 40 there; the stepwise launches on incoherent rays lose 1-2 % below 40.
    unsigned min_active[2][2] = {{20, 20}, {40, 40}};
    // traversal loop form (TraceLaunch::mode): [0] for depth-0 launches (coherent rays: the
    // while-while form wins), [1] for deeper launches and the stream entry points
    unsigned fast_tree_device_build = 1; // the fast forms' tree is built on the GPU: 0 never, 1 from 2^18 triangles, 2 always
    unsigned fast_tree_buckets = 32; // >= 2: the fast forms get a tree from the search builder with that many buckets; 0: the reference's tree
    // most triangles in a leaf of the fast forms' tree (1..3): 2 against 3 is +2 % on San-Miguel-class, +1 % on
    // dragon-class, nothing on Sponza-class; 1 gains on the deep launches and loses more on the depth-0 ones
    unsigned fast_tree_max_leaf = 2;
    unsigned fast_tree_align = 4;         // placement of the wide tree's items against 128-byte lines, see layout_wide4()
    unsigned trav_mode[2] = {3, 2};
    unsigned trav_deep_connect = 0xFFFFFFFFu; // form of the deeper any-hit launches; 0xFFFFFFFF: trav_mode[1]
    // the stream entry points are the reference's Bvh::intersect_stream / occluded_stream
    // boundary and walk the reference's binary tree unless told otherwise
    unsigned trav_stream = 1;
    // depth 0 of rt_render without ray and PathInfo queues: extend makes the camera rays itself and
    // shade rebuilds them from the slot (off when visit counters are requested)
    unsigned fuse_primary = 1;
    // queues are reordered by the Morton code of the ray origin from these depths
    // on (depth of the launch that consumes the queue); 0xFFFFFFFF = never
    unsigned sort_from_depth[2] = {0xFFFFFFFFu, 1}; // [0] extend (never), [1] connect
    unsigned sort_bits = 21;              // high bits of the 30-bit key that are sorted
    unsigned sort_octant = 0;             // bounce rays: the direction's octant in the three lowest sorted bits
    Workspace ws;
    // Queue lengths of the last finished render pass, read back asynchronously: the next pass
    // sorts only that many entries (+ slack) of each queue instead of the whole allocation.
    uint32_t *h_counts = nullptr; // pinned
    hipEvent_t counts_ready = nullptr;
    bool counts_pending = false;
    std::vector<uint32_t> est_counts; // [0..nb] ray queues, then [0..nb-1] shadow queues
    uint64_t est_key[3] = {0, 0, 0};  // plan the estimate belongs to: scene, slots, depths
    uint64_t pending_key[3] = {0, 0, 0};
    // Paths per render pass.  2^28 holds a 1920x1080 frame at the reference's default 128 samples per pixel
    // in one pass (48 GB of workspace on a 288 GB GPU); the persistent trace launches pay their ramp and tail
    // once per launch, so one pass of 2^28 paths is 5-8 % faster than four of 2^26.
    uint64_t max_paths = 1ull << 28;
    // test hook: a render workspace beyond this many bytes counts as an allocation failure (0 = no limit)
    uint64_t workspace_limit = 0;
    uint64_t last_pass_paths = 0; // path slots per pass the last render call planned with
    DevBuf io_a, io_b; // staging for the host-buffer entry points
    // rt_intersect_stream / rt_occluded_stream work through their rays in chunks: while chunk k is
    // traced, chunk k+1 is uploaded on `copy_stream` and the results of chunk k-1 (hit ids and far
    // distances, 8 bytes per ray instead of the 48-byte ray) are unpacked on the host
    hipStream_t copy_stream = nullptr;
    hipEvent_t h2d_done[2] = {nullptr, nullptr}, kernel_done[2] = {nullptr, nullptr}, out_done[2] = {nullptr, nullptr};
    DevBuf chunk_rays[2], chunk_out[2];
    void *h_out[2] = {nullptr, nullptr}; // pinned, kHostChunk * 8 bytes each
    std::vector<TimedLaunch> pool;
    size_t used = 0;
    bool timing = true;
    rt_trace_stats last_split[2]{}; // [0] closest-hit launches, [1] any-hit launches of the last call with stats
    uint64_t last_max_steps[2] = {0, 0};
    uint64_t last_max_stack[2] = {0, 0};
    rt_depth_stats last_depth[2]{}; // per path depth, render calls with stats
    unsigned char *na_scratch = nullptr; // this device's copy of g_noalloc_scratch (rt_intersect_stream_noalloc)
};

struct rt_scene {
    rt_ctx *ctx = nullptr;
    HostBvh host;
    uint32_t num_triangles = 0;
    DevBuf nodes, witems, tris, big, normals;
    uint32_t num_wide_nodes = 0;
    uint32_t num_wide_units = 0; // 16-byte units of the wide tree's item array
    bool fast_tree_on_device = false; // the search tree came from the device-side builder
    bool fast_items_on_device = false; // and so did the traversal items made from it
    uint32_t stack_bound = 0; // entries a lane's traversal stack can need on this scene
    SceneDev dev{};
    uint64_t device_bytes = 0;
    float bounds_lo[3] = {0, 0, 0};
    float bounds_hi[3] = {0, 0, 0};
    float key_scale[3] = {0, 0, 0}; // 1024 / extent, for the Morton sort keys
};

// ------------------------------------------------------------------------
// Scratch of rt_intersect_stream_noalloc: ray-fetch counters and the part of the traversal stacks
// that does not fit in LDS, for a grid of kNaBlocks blocks.  A file-scope device array: it exists
// once per GPU the library runs on, from module load to process end, and needs no allocation call.
// Host code never uses the array's name as an address; rt_ctx_create asks the runtime for this
// device's address of it (hipGetSymbolAddress) and keeps that in the context.
namespace {
constexpr unsigned kNaBlocks = 512;                                                   // 2 per CU
constexpr unsigned kNaOvfEntries = RT_NOALLOC_MAX_STACK - RTC_LDS_STACK_DEPTH;        // per thread
constexpr size_t kNaFetchBytes = (size_t)kFetchUints * sizeof(unsigned);
constexpr size_t kNaOvfBytes = (size_t)kNaBlocks * RTC_TRACE_BLOCK * kNaOvfEntries * sizeof(unsigned);
constexpr size_t kNaBytes = kNaFetchBytes + kNaOvfBytes;
static_assert(kNaFetchBytes % 256 == 0, "the overflow area starts on a 256-byte boundary");
static_assert(kNaBytes <= (size_t)RT_NOALLOC_SCRATCH_BYTES, "rtcore.h states the size of the scratch array");
} // namespace
static __device__ __align__(256) unsigned char g_noalloc_scratch[kNaBytes];

static int ctx_set_device(rt_ctx *ctx) { HIP_TRY(hipSetDevice(ctx->device)); return RT_OK; }

static int ensure_overflow(rt_ctx *ctx, unsigned max_depth)
{
    // entries a lane can need beyond the LDS part of its stack, for the kernel with the smallest LDS part
    constexpr unsigned in_lds = (unsigned)(RTC_MIN_LDS_STACK_DEPTH);
    unsigned need = max_depth + 1 > in_lds ? max_depth + 1 - in_lds : 1;
    // the two-rays-per-lane form keeps two stacks per lane with fewer LDS entries each
    constexpr unsigned dual_lds = (unsigned)(RTC_DUAL_LDS_STACK_DEPTH);
    need = std::max(need, 2u * (max_depth + 1 > dual_lds ? max_depth + 1 - dual_lds : 1u));
    if (need <= ctx->ovf_depth) return RT_OK;
    size_t threads = (size_t)ctx->max_blocks * RTC_TRACE_BLOCK;
    int rc = ctx->ovf.ensure(2 * threads * need * sizeof(unsigned));
    if (rc != RT_OK) return rc;
    ctx->ovf_depth = need;
    return RT_OK;
}

// cooperative-finish settings of a launch on `sc`: off when the tree is too deep
// for the fixed work-stack capacity
static void fill_coop(const rt_ctx *ctx, const rt_scene *sc, TraceLaunch &tl, bool second_set = false)
{
    // the second set of work stacks belongs to launches on the context's connect stream
    tl.coop = (char *)ctx->coop.p +
              (second_set ? (size_t)ctx->max_blocks * (RTC_TRACE_BLOCK / 64) * ctx->coop_cap * 8 : 0); // 8 bytes per work-stack entry
    tl.coop_cap = ctx->coop_cap;
    // the work stack takes over a lane's whole traversal stack and then needs the
    // headroom coop_finish documents
    // the binary finisher keeps max_depth + 2 entries in reserve, the wide one 3 * wide depth + 8, on top
    // of a lane's whole stack and one batch of 64
    const unsigned reserve = std::max(sc->dev.max_depth + 2u, 3u * sc->dev.wmax_depth + 8u);
    const bool binary = tl.mode == 0 || tl.mode == 1 || tl.mode == 4 || !sc->dev.witems;
    tl.coop_steps = (reserve + 4u * 64u + sc->stack_bound < ctx->coop_cap)
                        ? (binary ? ctx->coop_steps_binary : ctx->coop_steps)
                        : 0xFFFFFFFFu;
    // a triangle job of the queued form holds a unit index in 26 bits (1 GB of wide-tree items)
    tl.job_units_ok = sc->num_wide_units < (1u << 26) ? 1u : 0u;
}

static int timed_begin(rt_ctx *ctx, int kind, hipStream_t s, size_t *slot)
{
    *slot = (size_t)-1;
    if (!ctx->timing) return RT_OK;
    if (ctx->used == ctx->pool.size()) {
        TimedLaunch t{};
        HIP_TRY(hipEventCreate(&t.start));
        HIP_TRY(hipEventCreate(&t.stop));
        ctx->pool.push_back(t);
    }
    *slot = ctx->used++;
    ctx->pool[*slot].kind = kind;
    HIP_TRY(hipEventRecord(ctx->pool[*slot].start, s));
    return RT_OK;
}

static int timed_end(rt_ctx *ctx, size_t slot, hipStream_t s)
{
    if (slot == (size_t)-1) return RT_OK;
    HIP_TRY(hipEventRecord(ctx->pool[slot].stop, s));
    return RT_OK;
}

extern "C" int rt_ctx_create(int device_id, rt_ctx **out)
{
    if (!out) return fail(RT_ERR_INVALID, "rt_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev)
        return fail(RT_ERR_INVALID, "rt_ctx_create: device %d out of range (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return fail(RT_ERR_NOMEM, "rt_ctx_create: out of host memory");
    ctx->device = device_id;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
        delete ctx;
        return fail(RT_ERR_HIP, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
    }
    ctx->num_cus = prop.multiProcessorCount;
    int per_cu = trace_blocks_per_cu();
    if (const char *env = getenv("RTCORE_BLOCKS_PER_CU")) {
        int v = atoi(env);
        if (v >= 1 && v <= 16) per_cu = v;
    }
    ctx->max_blocks = (unsigned)(ctx->num_cus * per_cu);
    if (const char *env = getenv("RTCORE_NO_TIMING")) ctx->timing = !(env[0] == '1');
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->conn_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(RT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    if (const char *env = getenv("RTCORE_OVERLAP_CONNECT")) ctx->overlap_connect = atoi(env) ? 1u : 0u;
    // RTCORE_MIN_ACTIVE, like the option "min_active": the deeper launches and the stream entry points;
    // RTCORE_MIN_ACTIVE_EXTEND / _CONNECT: one kind of them; RTCORE_MIN_ACTIVE_DEPTH0, like
    // "min_active_depth0": the depth-0 launches of rt_render
    if (const char *env = getenv("RTCORE_MIN_ACTIVE")) {
        int v = atoi(env);
        if (v >= 1 && v <= 64) ctx->min_active[1][0] = ctx->min_active[1][1] = (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_TRAVERSAL")) {
        int v = atoi(env);
        if (v >= 0 && v <= 3) ctx->trav_mode[0] = ctx->trav_mode[1] = (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_FUSE_PRIMARY")) ctx->fuse_primary = atoi(env) ? 1u : 0u;
    if (const char *env = getenv("RTCORE_TRAVERSAL_STREAM")) {
        int v = atoi(env);
        if (v >= 0 && v <= 8) ctx->trav_stream = (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_TRAVERSAL_DEPTH0")) {
        int v = atoi(env);
        if (v >= 0 && v <= 3) ctx->trav_mode[0] = (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_TRAVERSAL_DEEP")) {
        int v = atoi(env);
        if ((v >= 0 && v <= 3) || (v >= 5 && v <= 8)) ctx->trav_mode[1] = (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_TRAVERSAL_DEEP_CONNECT")) {
        int v = atoi(env);
        if ((v >= 0 && v <= 3) || (v >= 5 && v <= 8)) ctx->trav_deep_connect = (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_SORT_EXTEND_FROM")) ctx->sort_from_depth[0] = (unsigned)atoi(env);
    if (const char *env = getenv("RTCORE_SORT_CONNECT_FROM")) ctx->sort_from_depth[1] = (unsigned)atoi(env);
    if (const char *env = getenv("RTCORE_SORT_BITS")) {
        int v = atoi(env);
        if (v >= 3 && v <= 30) ctx->sort_bits = (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_SORT_OCTANT")) ctx->sort_octant = atoi(env) != 0;
    if (const char *env = getenv("RTCORE_COOP_STEPS")) {
        long long v = atoll(env);
        ctx->coop_steps = ctx->coop_steps_binary = (v <= 0 || v >= 0xFFFFFFFFll) ? 0xFFFFFFFFu : (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_COOP_STEPS_BINARY")) {
        long long v = atoll(env);
        ctx->coop_steps_binary = (v <= 0 || v >= 0xFFFFFFFFll) ? 0xFFFFFFFFu : (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_MIN_ACTIVE_EXTEND")) {
        int v = atoi(env);
        if (v >= 1 && v <= 64) ctx->min_active[1][0] = (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_MIN_ACTIVE_CONNECT")) {
        int v = atoi(env);
        if (v >= 1 && v <= 64) ctx->min_active[1][1] = (unsigned)v;
    }
    // the depth-0 launches have their own threshold: parsed last, and the variables above leave it alone
    if (const char *env = getenv("RTCORE_MIN_ACTIVE_DEPTH0")) {
        int v = atoi(env);
        if (v >= 1 && v <= 64) ctx->min_active[0][0] = ctx->min_active[0][1] = (unsigned)v;
    }
    if (const char *env = getenv("RTCORE_COOP_CAP")) {
        int v = atoi(env);
        if (v >= 256 && v <= (1 << 20)) ctx->coop_cap = (unsigned)v;
    }
    int rc = ctx->dstats.ensure(2 * sizeof(StatsDev));
    if (rc == RT_OK) rc = ctx->fetch.ensure(kFetchUints * sizeof(unsigned));
    if (rc == RT_OK)
        rc = ctx->coop.ensure(2 * (size_t)ctx->max_blocks * (RTC_TRACE_BLOCK / 64) * ctx->coop_cap * 8); // one set per stream
    if (rc == RT_OK) rc = ensure_overflow(ctx, 0);
    if (rc == RT_OK && hipHostMalloc((void **)&ctx->h_counts, kMaxQueueCounters * sizeof(uint32_t)) != hipSuccess)
        rc = fail(RT_ERR_NOMEM, "rt_ctx_create: pinned host allocation failed");
    if (rc == RT_OK && hipEventCreateWithFlags(&ctx->counts_ready, hipEventDisableTiming) != hipSuccess)
        rc = fail(RT_ERR_HIP, "rt_ctx_create: hipEventCreate failed");
    if (rc == RT_OK) { // device_id is current: the address is that of this GPU's copy of the array
        void *p = nullptr;
        const hipError_t se = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch));
        if (se != hipSuccess || !p)
            rc = fail(RT_ERR_HIP, "rt_ctx_create: hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(se));
        ctx->na_scratch = (unsigned char *)p;
    }
    if (rc != RT_OK) {
        rt_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return RT_OK;
}

extern "C" void rt_ctx_destroy(rt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &t : ctx->pool) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    if (ctx->counts_ready) (void)hipEventDestroy(ctx->counts_ready);
    for (hipEvent_t ev : ctx->ev_shade) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->ev_conn) (void)hipEventDestroy(ev);
    if (ctx->conn_stream) (void)hipStreamDestroy(ctx->conn_stream);
    ctx->ovf.release();
    ctx->dstats.release();
    ctx->fetch.release();
    ctx->coop.release();
    ctx->io_a.release();
    ctx->io_b.release();
    for (int k = 0; k < 2; k++) {
        ctx->chunk_rays[k].release();
        ctx->chunk_out[k].release();
        if (ctx->h_out[k]) (void)hipHostFree(ctx->h_out[k]);
        if (ctx->h2d_done[k]) (void)hipEventDestroy(ctx->h2d_done[k]);
        if (ctx->kernel_done[k]) (void)hipEventDestroy(ctx->kernel_done[k]);
        if (ctx->out_done[k]) (void)hipEventDestroy(ctx->out_done[k]);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    ctx->ws.release_all();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int rt_ctx_set_max_paths(rt_ctx *ctx, uint64_t max_paths)
{
    if (!ctx) return fail(RT_ERR_INVALID, "rt_ctx_set_max_paths: ctx is NULL");
    if (max_paths < 4096 || max_paths > (1ull << 30))
        return fail(RT_ERR_LIMIT, "rt_ctx_set_max_paths: %llu outside [4096, 2^30]", (unsigned long long)max_paths);
    ctx->max_paths = max_paths;
    return RT_OK;
}

extern "C" int rt_ctx_set_option(rt_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return fail(RT_ERR_INVALID, "rt_ctx_set_option: NULL argument");
    const std::string n(name);
    auto in = [&](int64_t lo, int64_t hi) { return value >= lo && value <= hi; };
    if (n == "coop_steps") {
        ctx->coop_steps = ctx->coop_steps_binary = (value <= 0 || value >= 0xFFFFFFFFll) ? 0xFFFFFFFFu : (unsigned)value;
    } else if (n == "coop_steps_binary") {
        ctx->coop_steps_binary = (value <= 0 || value >= 0xFFFFFFFFll) ? 0xFFFFFFFFu : (unsigned)value;
    } else if (n == "min_active" && in(1, 64)) {
        ctx->min_active[1][0] = ctx->min_active[1][1] = (unsigned)value;
    } else if (n == "min_active_depth0" && in(1, 64)) {
        ctx->min_active[0][0] = ctx->min_active[0][1] = (unsigned)value;
    } else if (n == "workspace_limit_mb" && in(0, 1 << 20)) {
        ctx->workspace_limit = (uint64_t)value << 20;
    } else if (n == "overlap_connect" && in(0, 1)) {
        ctx->overlap_connect = (unsigned)value;
    } else if (n == "fuse_primary" && in(0, 1)) {
        ctx->fuse_primary = (unsigned)value;
    } else if (n == "fast_tree_buckets" && (value == 0 || in(2, 64))) {
        ctx->fast_tree_buckets = (unsigned)value;
    } else if (n == "fast_tree_device_build" && in(0, 3)) {
        ctx->fast_tree_device_build = (unsigned)value;
    } else if (n == "fast_tree_max_leaf" && in(1, kWideLeafTris)) {
        ctx->fast_tree_max_leaf = (unsigned)value;
    } else if (n == "fast_tree_align" && (in(0, 1) || in(3, 4))) {
        ctx->fast_tree_align = (unsigned)value;
    } else if (n == "traversal_stream" && in(0, 8)) {
        ctx->trav_stream = (unsigned)value;
    } else if (n == "traversal_depth0" && in(0, 3)) {
        ctx->trav_mode[0] = (unsigned)value;
    } else if (n == "traversal_deep" && (in(0, 3) || in(5, 8))) {
        ctx->trav_mode[1] = (unsigned)value;
    } else if (n == "traversal_deep_connect" && (value == -1 || in(0, 3) || in(5, 8))) {
        ctx->trav_deep_connect = value < 0 ? 0xFFFFFFFFu : (unsigned)value;
    } else if (n == "sort_extend_from" && in(0, 0xFFFFFFFFll)) {
        ctx->sort_from_depth[0] = (unsigned)value;
    } else if (n == "sort_connect_from" && in(0, 0xFFFFFFFFll)) {
        ctx->sort_from_depth[1] = (unsigned)value;
    } else if (n == "sort_bits" && in(3, 30)) {
        ctx->sort_bits = (unsigned)value;
    } else if (n == "sort_octant" && in(0, 1)) {
        ctx->sort_octant = (unsigned)value;
    } else if (n == "timing" && in(0, 1)) {
        ctx->timing = value != 0;
    } else {
        return fail(RT_ERR_INVALID, "rt_ctx_set_option: unknown option or value out of range: %s=%lld", name,
                    (long long)value);
    }
    return RT_OK;
}

extern "C" int rt_ctx_last_kernel_ms(rt_ctx *ctx, float ms[3], uint32_t launches[3])
{
    if (!ctx || !ms || !launches) return fail(RT_ERR_INVALID, "rt_ctx_last_kernel_ms: NULL argument");
    int rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;
    for (int k = 0; k < 3; k++) {
        ms[k] = 0.f;
        launches[k] = 0;
    }
    for (size_t i = 0; i < ctx->used; i++) {
        TimedLaunch &t = ctx->pool[i];
        HIP_TRY(hipEventSynchronize(t.stop));
        float v = 0.f;
        HIP_TRY(hipEventElapsedTime(&v, t.start, t.stop));
        ms[t.kind] += v;
        launches[t.kind] += 1;
    }
    return RT_OK;
}

extern "C" int rt_debug_sort_keys(rt_ctx *ctx, const uint32_t *keys, uint32_t n, uint32_t bits, uint32_t *perm)
{
    if (!ctx || (n && (!keys || !perm))) return fail(RT_ERR_INVALID, "rt_debug_sort_keys: NULL argument");
    if (bits < 1 || bits > 32) return fail(RT_ERR_INVALID, "rt_debug_sort_keys: bits %u outside [1, 32]", bits);
    if (n == 0) return RT_OK;
    int rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;
    // the order is defined by the top `bits` bits alone: the sort may run over the whole word
    // (sort.hip), so the other bits are cleared in the copy that is sorted
    std::vector<uint32_t> masked;
    if (bits < 32) {
        try {
            masked.assign(keys, keys + n);
        } catch (const std::exception &) {
            return fail(RT_ERR_NOMEM, "rt_debug_sort_keys: out of host memory");
        }
        const uint32_t mask = 0xFFFFFFFFu << (32u - bits);
        for (uint32_t &k : masked) k &= mask;
        keys = masked.data();
    }
    DevBuf k_in, k_out, p_out, temp;
    size_t tb = sort_temp_bytes(n, 32u - bits, 32u);
    if (tb == 0) return fail(RT_ERR_HIP, "radix sort temporary-storage query failed");
    rc = k_in.ensure((size_t)n * 4);
    if (rc == RT_OK) rc = k_out.ensure((size_t)n * 4);
    if (rc == RT_OK) rc = p_out.ensure((size_t)n * 4);
    if (rc == RT_OK) rc = temp.ensure(tb);
    hipError_t e = hipSuccess;
    if (rc == RT_OK) {
        hipStream_t s = ctx->stream;
        e = hipMemcpyAsync(k_in.p, keys, (size_t)n * 4, hipMemcpyHostToDevice, s);
        if (e == hipSuccess)
            e = sort_indices_by_key(temp.p, temp.bytes, (const unsigned *)k_in.p, (unsigned *)k_out.p,
                                    (unsigned *)p_out.p, n, 32u - bits, 32u, s);
        if (e == hipSuccess) e = hipMemcpyAsync(perm, p_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    k_in.release();
    k_out.release();
    p_out.release();
    temp.release();
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_debug_sort_keys: %s", hipGetErrorString(e));
    return RT_OK;
}

extern "C" int rt_ctx_last_launch_ms(rt_ctx *ctx, float *ms, uint32_t *kind, uint32_t cap, uint32_t *count)
{
    if (!ctx || !count || (cap && (!ms || !kind)))
        return fail(RT_ERR_INVALID, "rt_ctx_last_launch_ms: NULL argument");
    int rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;
    *count = (uint32_t)ctx->used;
    for (size_t i = 0; i < ctx->used && i < cap; i++) {
        TimedLaunch &t = ctx->pool[i];
        HIP_TRY(hipEventSynchronize(t.stop));
        float v = 0.f;
        HIP_TRY(hipEventElapsedTime(&v, t.start, t.stop));
        ms[i] = v;
        kind[i] = (uint32_t)t.kind;
    }
    return RT_OK;
}

extern "C" int rt_ctx_last_pass_paths(rt_ctx *ctx, uint64_t *out)
{
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_ctx_last_pass_paths: NULL argument");
    *out = ctx->last_pass_paths;
    return RT_OK;
}

extern "C" int rt_ctx_last_stats_split(rt_ctx *ctx, rt_trace_stats out[2])
{
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_ctx_last_stats_split: NULL argument");
    out[0] = ctx->last_split[0];
    out[1] = ctx->last_split[1];
    return RT_OK;
}

extern "C" int rt_ctx_last_max_ray_steps(rt_ctx *ctx, uint64_t out[2])
{
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_ctx_last_max_ray_steps: NULL argument");
    out[0] = ctx->last_max_steps[0];
    out[1] = ctx->last_max_steps[1];
    return RT_OK;
}

extern "C" int rt_ctx_last_depth_stats(rt_ctx *ctx, rt_depth_stats out[2])
{
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_ctx_last_depth_stats: NULL argument");
    out[0] = ctx->last_depth[0];
    out[1] = ctx->last_depth[1];
    return RT_OK;
}

extern "C" int rt_ctx_last_max_stack_entries(rt_ctx *ctx, uint64_t out[2])
{
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_ctx_last_max_stack_entries: NULL argument");
    out[0] = ctx->last_max_stack[0];
    out[1] = ctx->last_max_stack[1];
    return RT_OK;
}

// ------------------------------------------------------------------------
// scene

// A NaN or infinite coordinate makes every binning decision of the builder meaningless (the
// reference would index outside its bucket array and panic, src/bvh/build.rs:118-120); meshes are
// checked once here.  Returns the index of the first such float, or -1.
// fn(begin, end) over [0, n) cut into slices for up to 16 threads (scans of large meshes); threads that were
// started are joined whatever happens
template <class F>
static void scan_slices(size_t n, F fn)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned T = n < (1u << 20) ? 1u : std::min(16u, std::max(1u, hw));
    if (T == 1) {
        fn((size_t)0, n, 0u);
        return;
    }
    struct Pool {
        std::vector<std::thread> threads;
        ~Pool()
        {
            for (auto &th : threads)
                if (th.joinable()) th.join();
        }
    } pool;
    pool.threads.reserve(T);
    for (unsigned t = 1; t < T; t++) pool.threads.emplace_back([&fn, n, t, T] { fn(n * t / T, n * (t + 1) / T, t); });
    fn((size_t)0, n / T, 0u);
}

static long long first_non_finite(const float *positions, uint32_t num_vertices)
{
    long long first[16];
    for (long long &f : first) f = -1;
    scan_slices((size_t)num_vertices * 3, [&](size_t lo, size_t hi, unsigned t) {
        for (size_t i = lo; i < hi; i++)
            if (!std::isfinite(positions[i])) {
                first[t] = (long long)i;
                return;
            }
    });
    for (long long f : first)
        if (f >= 0) return f; // slices are in index order: the first hit of the first slice that has one
    return -1;
}

// index of the first entry of `indices` that is not below num_vertices, or -1
static long long first_bad_index(const uint32_t *indices, size_t count, uint32_t num_vertices)
{
    long long first[16];
    for (long long &f : first) f = -1;
    scan_slices(count, [&](size_t lo, size_t hi, unsigned t) {
        for (size_t i = lo; i < hi; i++)
            if (indices[i] >= num_vertices) {
                first[t] = (long long)i;
                return;
            }
    });
    for (long long f : first)
        if (f >= 0) return f;
    return -1;
}

// Device form of the canonical flat tree: same node order, leaf references
// packed as device_layout.h describes.  Returns the root reference.
static uint32_t make_dev_nodes(const rt_flat_node *nodes, size_t nn, uint32_t root_ref, uint32_t root_count,
                               std::vector<DevNode> &dn, std::vector<BigLeaf> &big)
{
    auto pack_ref = [&](uint32_t ref, uint32_t count) -> uint32_t {
        if (!(ref & kLeafBit)) return ref;
        uint32_t first = ref & 0x7FFFFFFFu;
        if (count >= 1 && count <= 8) return kLeafBit | (first << 3) | (count - 1);
        big.push_back(BigLeaf{first, count});
        return kLeafBit | kBigLeafBit | (uint32_t)(big.size() - 1);
    };
    dn.resize(nn);
    // leaves of more than 8 triangles (coincident centroids) are rare and numbered in node order: the
    // nodes are converted in parallel with those references left open, then the few are filled in
    auto is_big = [](uint32_t ref, uint32_t count) { return (ref & kLeafBit) && !(count >= 1 && count <= 8); };
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned T = nn < (1u << 18) ? 1u : std::min(16u, std::max(1u, hw));
    std::vector<std::vector<uint32_t>> open(T); // nodes with a big-leaf child, per thread in index order
    {
        struct Pool {
            std::vector<std::thread> threads;
            ~Pool()
            {
                for (auto &th : threads)
                    if (th.joinable()) th.join();
            }
        } pool;
        auto work = [&](unsigned t) {
            for (size_t i = nn * t / T; i < nn * (t + 1) / T; i++) {
                const rt_flat_node &s = nodes[i];
                DevNode &d = dn[i];
                memcpy(d.bx, s.bx, sizeof d.bx);
                memcpy(d.by, s.by, sizeof d.by);
                memcpy(d.bz, s.bz, sizeof d.bz);
                const bool bl = is_big(s.left, s.left_count), br = is_big(s.right, s.right_count);
                d.left = bl ? 0u : (s.left & kLeafBit) ? kLeafBit | ((s.left & 0x7FFFFFFFu) << 3) | (s.left_count - 1) : s.left;
                d.right = br ? 0u : (s.right & kLeafBit) ? kLeafBit | ((s.right & 0x7FFFFFFFu) << 3) | (s.right_count - 1) : s.right;
                d.pad0 = d.pad1 = 0;
                if (bl || br) open[t].push_back((uint32_t)i);
            }
        };
        pool.threads.reserve(T);
        for (unsigned t = 1; t < T; t++) pool.threads.emplace_back(work, t);
        work(0);
    }
    for (unsigned t = 0; t < T; t++)
        for (uint32_t i : open[t]) {
            const rt_flat_node &s = nodes[i];
            if (is_big(s.left, s.left_count)) dn[i].left = pack_ref(s.left, s.left_count);
            if (is_big(s.right, s.right_count)) dn[i].right = pack_ref(s.right, s.right_count);
        }
    return pack_ref(root_ref, root_count);
}

static int scene_prepare(rt_ctx *ctx, rt_scene *sc, const float *positions, uint32_t num_vertices,
                         const uint32_t *indices, uint32_t num_triangles, const float *geom_normals);

extern "C" int rt_scene_create(rt_ctx *ctx, const float *positions, uint32_t num_vertices,
                               const uint32_t *indices, uint32_t num_triangles, const float *geom_normals,
                               rt_scene **out)
{
    if (!out) return fail(RT_ERR_INVALID, "rt_scene_create: out is NULL");
    *out = nullptr;
    if (!ctx || !positions || !indices || !geom_normals)
        return fail(RT_ERR_INVALID, "rt_scene_create: NULL argument");
    if (num_triangles == 0 || num_vertices == 0)
        return fail(RT_ERR_INVALID, "rt_scene_create: empty mesh");
    if (num_triangles >= kMaxPackedFirst)
        return fail(RT_ERR_LIMIT, "rt_scene_create: %u triangles, limit is %u", num_triangles, kMaxPackedFirst - 1);
    if (const long long i = first_bad_index(indices, (size_t)num_triangles * 3, num_vertices); i >= 0)
        return fail(RT_ERR_INVALID, "rt_scene_create: index %u at %lld out of range (%u vertices)", indices[i], i,
                    num_vertices);
    if (const long long bad = first_non_finite(positions, num_vertices); bad >= 0)
        return fail(RT_ERR_INVALID, "rt_scene_create: vertex %lld has a NaN or infinite coordinate", bad / 3);
    int rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;

    rt_scene *sc = new (std::nothrow) rt_scene();
    if (!sc) return fail(RT_ERR_NOMEM, "rt_scene_create: out of host memory");
    sc->ctx = ctx;
    sc->num_triangles = num_triangles;
    // Everything below allocates and starts threads on the host; nothing of it may leave through
    // the C ABI as an exception.
    try {
        rc = scene_prepare(ctx, sc, positions, num_vertices, indices, num_triangles, geom_normals);
    } catch (const std::invalid_argument &ex) {
        rc = fail(RT_ERR_INVALID, "rt_scene_create: %s", ex.what());
    } catch (const std::exception &ex) {
        rc = fail(RT_ERR_NOMEM, "rt_scene_create: scene preparation failed: %s", ex.what());
    }
    if (rc != RT_OK) {
        rt_scene_destroy(sc);
        return rc;
    }
    *out = sc;
    return RT_OK;
}

// Host-side half of rt_scene_create: both trees, their device images, the upload.  May throw
// (std::bad_alloc, std::system_error from the builder's threads, std::invalid_argument for a mesh
// the binning cannot split); returns an RT_ERR_* code for everything else.
static int scene_prepare(rt_ctx *ctx, rt_scene *sc, const float *positions, uint32_t num_vertices,
                         const uint32_t *indices, uint32_t num_triangles, const float *geom_normals)
{
    int rc = RT_OK;
    // RTCORE_VERBOSE=1: seconds per preparation phase on stderr
    const bool verbose = getenv("RTCORE_VERBOSE") && getenv("RTCORE_VERBOSE")[0] == '1';
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char *what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "rt_scene_create: %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_phase).count());
        t_phase = now;
    };
    unsigned hw = std::thread::hardware_concurrency();
    unsigned threads = hw ? std::min(hw, 16u) : 4u;
    if (const char *env = getenv("RTCORE_BUILD_THREADS")) {
        int v = atoi(env);
        if (v >= 1 && v <= 256) threads = (unsigned)v;
    }
    // the wide tree is built over a copy of a tree whose larger leaves are split further
    unsigned leaf_tris = ctx->fast_tree_max_leaf;
    if (const char *env = getenv("RTCORE_WIDE_LEAF_TRIS")) {
        int v = atoi(env);
        if (v >= 1 && v <= (int)kWideLeafTris) leaf_tris = (unsigned)v;
    }
    unsigned search_buckets = ctx->fast_tree_buckets;
    if (const char *env = getenv("RTCORE_FAST_TREE_BUCKETS")) {
        const int v = atoi(env);
        if (v == 0 || (v >= 2 && v <= 64)) search_buckets = (unsigned)v;
    }
    // fn(begin, end) over [0, n) on the build threads (plain loops over millions of triangles)
    auto parallel_ranges = [&](size_t n, auto fn) {
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / 65536 + 1));
        if (T == 1) {
            fn((size_t)0, n);
            return;
        }
        // threads already started are joined whatever happens (a later one may fail to start): a
        // joinable std::thread that is destroyed terminates the process
        struct Pool {
            std::vector<std::thread> threads;
            ~Pool()
            {
                for (auto &th : threads)
                    if (th.joinable()) th.join();
            }
        } pool;
        pool.threads.reserve(T);
        for (unsigned t = 0; t < T; t++) pool.threads.emplace_back([&, t] { fn(n * t / T, n * (t + 1) / T); });
    };
    // the large arrays below are written in full by the build threads: allocated without the
    // single-threaded zero fill a std::vector would do first
    auto make_tris = [&](const std::vector<uint32_t> &order, std::unique_ptr<DevTri[]> &dt) {
        dt.reset(new DevTri[num_triangles]);
        parallel_ranges(num_triangles, [&](size_t lo, size_t hi) {
            for (size_t slot = lo; slot < hi; slot++) {
                uint32_t t = order[slot];
                const float *a = positions + 3 * (size_t)indices[3 * (size_t)t + 0];
                const float *b = positions + 3 * (size_t)indices[3 * (size_t)t + 1];
                const float *c = positions + 3 * (size_t)indices[3 * (size_t)t + 2];
                DevTri &d = dt[slot];
                for (int k = 0; k < 3; k++) { // src/triangle_opt.rs:72-81
                    d.v0[k] = a[k];
                    d.e1[k] = b[k] - a[k];
                    d.e2[k] = c[k] - a[k];
                }
                d.id = t;
                d.pad[0] = d.pad[1] = 0;
            }
        });
    };

    // Two pipelines that share nothing but the mesh.  The reference's tree (what the exact forms walk):
    // build, device nodes, triangles in its leaf order, normals.  The fast forms' tree: the search
    // builder's tree (bvh_build.hpp: same builder family, wider search) -- made on the GPU, breadth first
    // (bvh_build_gpu.hip), for meshes where that matters, else on the host -- then leaves refined, 4-wide
    // collapse, layout, quantization, packing.  They run side by side on the host threads; with
    // fast_tree_buckets = 0 the second starts from the first one's tree and follows it.
    const HostBvh &hb = sc->host;
    std::vector<BigLeaf> big;
    std::vector<DevNode> dn;
    std::unique_ptr<DevTri[]> dt;
    std::unique_ptr<float[]> n4;
    uint32_t root = 0;
    auto reference_pipeline = [&] {
        build_bvh(positions, num_vertices, indices, num_triangles, threads, sc->host);
        root = make_dev_nodes(hb.nodes.data(), hb.nodes.size(), hb.root_ref, hb.root_count, dn, big);
        make_tris(hb.order, dt);
        n4.reset(new float[(size_t)num_triangles * 4]);
        parallel_ranges(num_triangles, [&](size_t lo, size_t hi) {
            for (size_t t = lo; t < hi; t++) {
                n4[4 * t + 0] = geom_normals[3 * t + 0];
                n4[4 * t + 1] = geom_normals[3 * t + 1];
                n4[4 * t + 2] = geom_normals[3 * t + 2];
                n4[4 * t + 3] = 0.f;
            }
        });
    };
    HostBvh hs;
    std::unique_ptr<Unit[]> items;
    size_t num_items = 0;
    WideLayout layout;
    uint32_t wide_depth = 0, extra_depth = 0, fast_depth = 0;
    int fast_error = 0; // 1: a leaf too large for the wide tree, 2: too many units (reported on the calling thread)
    unsigned align_mode = ctx->fast_tree_align;
    if (const char *env = getenv("RTCORE_WIDE_ALIGN")) {
        const int v = atoi(env);
        if (v == 0 || v == 1 || v == 3 || v == 4) align_mode = (unsigned)v;
    }
    // from a binary tree (search tree, or the reference's with fast_tree_buckets = 0) to the packed items
    auto fast_pipeline = [&](const rt_flat_node *nodes, size_t num_nodes, uint32_t root_ref, uint32_t root_count,
                             const std::vector<uint32_t> &order, uint32_t depth, uint32_t max_objs) {
        std::vector<DevNode> sdn, xdn;
        std::vector<BigLeaf> sbig;
        std::vector<uint32_t> xorder;
        uint32_t xroot = 0;
        const uint32_t sroot = make_dev_nodes(nodes, num_nodes, root_ref, root_count, sdn, sbig);
        phase("  device nodes");
        if (max_objs <= leaf_tris) { // no leaf to split: the tree and its order are used as they are
            xdn.swap(sdn);
            xroot = sroot;
            extra_depth = 0;
        } else {
            extra_depth = refine_leaves(sdn, sroot, positions, indices, order, leaf_tris, sbig, xdn, xroot, xorder);
        }
        const std::vector<uint32_t> &final_order = max_objs <= leaf_tris ? order : xorder;
        fast_depth = depth;
        phase("  refined leaves");
        std::unique_ptr<DevTri[]> xdt;
        make_tris(final_order, xdt);
        phase("  wide-order triangles");
        std::vector<DevWideNode> wn;
        const uint32_t wroot = collapse_wide4(xdn.data(), xdn.size(), xroot, wn, &wide_depth);
        phase("  4-wide collapse");
        sc->num_wide_nodes = (uint32_t)wn.size();
        // a children block spread beyond its six address bits by alignment gaps: the packed placement always fits
        if (!layout_wide4(wn, wroot, num_triangles, layout, align_mode) &&
            (align_mode == 0 || !layout_wide4(wn, wroot, num_triangles, layout, 0))) {
            fast_error = 1;
            return;
        }
        phase("  layout");
        if (layout.num_units >= (1u << 30)) {
            fast_error = 2;
            return;
        }
        num_items = (size_t)layout.num_units + 1;
        items.reset(new Unit[num_items]);
        pack_wide4(wn, layout, xdt.get(), num_triangles, items.get(), false);
        sc->num_wide_units = (uint32_t)num_items;
        phase("  quantize + pack");
    };
    unsigned device_build = ctx->fast_tree_device_build;
    if (const char *env = getenv("RTCORE_FAST_TREE_DEVICE_BUILD")) {
        const int v = atoi(env);
        if (v >= 0 && v <= 3) device_build = (unsigned)v;
    }
    const bool on_device = search_buckets >= 2 && (device_build >= 2 || (device_build == 1 && num_triangles >= (1u << 18)));
    sc->fast_tree_on_device = false;
    sc->fast_items_on_device = false;
    if (search_buckets >= 2) {
        // the reference pipeline on a thread of its own, the fast one here (the GPU build needs this
        // thread's device)
        std::exception_ptr ref_error;
        std::thread ref_thread([&] {
            try {
                reference_pipeline();
            } catch (...) {
                ref_error = std::current_exception();
            }
        });
        struct Joiner {
            std::thread &t;
            ~Joiner() { if (t.joinable()) t.join(); }
        } ref_joiner{ref_thread};
        const BuildTuning tune{search_buckets, leaf_tris};
        DeviceWideRequest wide_rq;
        if (on_device) {
            std::string why;
            // the items too, unless the host is asked to make them (fast_tree_device_build 3: the device builds
            // the binary tree only; for comparisons)
            wide_rq.align_mode = align_mode;
            wide_rq.leaf_tris = leaf_tris;
            wide_rq.alloc = [&](size_t units) -> Unit * {
                return sc->witems.ensure(units * sizeof(Unit)) == RT_OK ? (Unit *)sc->witems.p : nullptr;
            };
            sc->fast_tree_on_device = build_bvh_search_device(positions, num_vertices, indices, num_triangles, tune, hs, why,
                                                             device_build == 3 ? nullptr : &wide_rq);
            phase(wide_rq.done ? "search tree and items (device)" : "search tree (device)");
            if (!sc->fast_tree_on_device && verbose)
                fprintf(stderr, "rt_scene_create: device build not used: %s\n", why.c_str());
            if (sc->fast_tree_on_device && !wide_rq.done && !wide_rq.why.empty() && verbose)
                fprintf(stderr, "rt_scene_create: items made on the host: %s\n", wide_rq.why.c_str());
        }
        if (!sc->fast_tree_on_device) {
            build_bvh(positions, num_vertices, indices, num_triangles, threads, hs, &tune);
            phase("search tree (host)");
        }
        if (wide_rq.done) {
            sc->fast_items_on_device = true;
            sc->num_wide_nodes = wide_rq.num_wide_nodes;
            sc->num_wide_units = wide_rq.num_units;
            num_items = wide_rq.num_units;
            layout.root_ref = wide_rq.root_ref;
            layout.num_units = wide_rq.num_units - 1u;
            wide_depth = wide_rq.wide_depth;
            fast_depth = hs.stats.max_depth;
            extra_depth = 0;
            sc->device_bytes += (size_t)wide_rq.num_units * sizeof(Unit);
        } else {
            fast_pipeline(hs.nodes.data(), hs.nodes.size(), hs.root_ref, hs.root_count, hs.order, hs.stats.max_depth,
                          hs.stats.max_objs);
            phase("refine, collapse, layout, pack");
        }
        ref_thread.join();
        if (ref_error) std::rethrow_exception(ref_error);
        phase("reference pipeline (rest)");
    } else {
        reference_pipeline();
        phase("reference pipeline");
        fast_pipeline(hb.nodes.data(), hb.nodes.size(), hb.root_ref, hb.root_count, hb.order, hb.stats.max_depth,
                      hb.stats.max_objs);
        phase("refine, collapse, layout, pack");
    }
    if (hb.stats.max_depth > 1024)
        return fail(RT_ERR_LIMIT, "rt_scene_create: tree depth %u beyond the supported 1024", hb.stats.max_depth);
    if (fast_error == 1)
        return fail(RT_ERR_LIMIT, "rt_scene_create: the wide tree has a leaf of more than %u triangles", kWideLeafTris);
    if (fast_error == 2)
        return fail(RT_ERR_LIMIT, "rt_scene_create: %u units in the wide tree, limit is 2^30", layout.num_units);

    auto upload = [&](DevBuf &buf, const void *src, size_t bytes) -> int {
        int r = buf.ensure(bytes ? bytes : 16);
        if (r != RT_OK) return r;
        if (bytes) HIP_TRY(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice));
        sc->device_bytes += bytes;
        return RT_OK;
    };
    rc = upload(sc->nodes, dn.data(), dn.size() * sizeof(DevNode));
    if (rc == RT_OK && !sc->fast_items_on_device) rc = upload(sc->witems, items.get(), num_items * sizeof(Unit));
    if (rc == RT_OK) rc = upload(sc->tris, dt.get(), (size_t)num_triangles * sizeof(DevTri));
    if (rc == RT_OK) rc = upload(sc->big, big.data(), big.size() * sizeof(BigLeaf));
    if (rc == RT_OK) rc = upload(sc->normals, n4.get(), (size_t)num_triangles * 4 * sizeof(float));
    phase("uploads");
    // Deepest traversal stack of a lane.  Binary walk: one deferred sibling per level.
    // Wide walk: up to three deferred children per wide level.
    sc->stack_bound = std::max(std::max(hb.stats.max_depth, fast_depth + extra_depth), 3u * wide_depth) + 1u;
    if (rc == RT_OK) rc = ensure_overflow(ctx, sc->stack_bound);
    if (rc != RT_OK) return rc;
    {
        float lo[3] = {positions[0], positions[1], positions[2]}, hi[3] = {positions[0], positions[1], positions[2]};
        float part[16][6];
        for (auto &p : part)
            for (int k = 0; k < 3; k++) {
                p[k] = positions[k];
                p[3 + k] = positions[k];
            }
        scan_slices(num_vertices, [&](size_t v0, size_t v1, unsigned t) {
            float l[3] = {positions[0], positions[1], positions[2]}, h[3] = {positions[0], positions[1], positions[2]};
            for (size_t v = v0; v < v1; v++)
                for (int k = 0; k < 3; k++) {
                    const float x = positions[3 * v + k];
                    l[k] = std::min(l[k], x);
                    h[k] = std::max(h[k], x);
                }
            for (int k = 0; k < 3; k++) {
                part[t][k] = l[k];
                part[t][3 + k] = h[k];
            }
        });
        for (auto &p : part)
            for (int k = 0; k < 3; k++) {
                lo[k] = std::min(lo[k], p[k]);
                hi[k] = std::max(hi[k], p[3 + k]);
            }
        for (int k = 0; k < 3; k++) {
            float ext = hi[k] - lo[k];
            sc->bounds_lo[k] = lo[k];
            sc->bounds_hi[k] = hi[k];
            sc->key_scale[k] = ext > 0.f ? 1024.f / ext : 0.f;
        }
        if (verbose)
            fprintf(stderr, "rt_scene_create: bounds (%g %g %g) .. (%g %g %g)\n", lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
    }
    sc->dev.nodes = (const DevNode *)sc->nodes.p;
    sc->dev.tris = (const DevTri *)sc->tris.p;
    sc->dev.big_leaves = (const BigLeaf *)sc->big.p;
    sc->dev.normals = (const float *)sc->normals.p;
    sc->dev.root_ref = root;
    sc->dev.witems = (const Unit *)sc->witems.p;
    sc->dev.wroot_ref = layout.root_ref;
    sc->dev.wmax_depth = wide_depth;
    sc->dev.max_depth = hb.stats.max_depth;
    return RT_OK;
}

static int bvh_build_host_impl(const char *who, const float *positions, uint32_t num_vertices, const uint32_t *indices,
                               uint32_t num_triangles, uint32_t threads, const BuildTuning *tuning, rt_flat_node *nodes,
                               uint32_t *order, rt_scene_info *info, rt_bvh_stats *stats)
{
    if (!positions || !indices || !nodes || !order)
        return fail(RT_ERR_INVALID, "%s: NULL argument", who);
    if (num_triangles == 0 || num_vertices == 0) return fail(RT_ERR_INVALID, "%s: empty mesh", who);
    for (size_t i = 0; i < (size_t)num_triangles * 3; i++)
        if (indices[i] >= num_vertices)
            return fail(RT_ERR_INVALID, "%s: index %u at %zu out of range", who, indices[i], i);
    if (const long long bad = first_non_finite(positions, num_vertices); bad >= 0)
        return fail(RT_ERR_INVALID, "%s: vertex %lld has a NaN or infinite coordinate", who, bad / 3);
    if (threads == 0) {
        unsigned hw = std::thread::hardware_concurrency();
        threads = hw ? std::min(hw, 16u) : 4u;
    }
    HostBvh hb;
    try {
        build_bvh(positions, num_vertices, indices, num_triangles, threads, hb, tuning);
    } catch (const std::invalid_argument &ex) {
        return fail(RT_ERR_INVALID, "%s: %s", who, ex.what());
    } catch (const std::exception &ex) {
        return fail(RT_ERR_NOMEM, "%s: %s", who, ex.what());
    }
    if (!hb.nodes.empty()) memcpy(nodes, hb.nodes.data(), hb.nodes.size() * sizeof(rt_flat_node));
    memcpy(order, hb.order.data(), hb.order.size() * sizeof(uint32_t));
    if (info) {
        info->num_triangles = num_triangles;
        info->num_inner_nodes = (uint32_t)hb.nodes.size();
        info->root_ref = hb.root_ref;
        info->root_count = hb.root_count;
        info->device_bytes = 0;
    }
    if (stats) *stats = hb.stats;
    return RT_OK;
}

extern "C" int rt_bvh_build_host(const float *positions, uint32_t num_vertices, const uint32_t *indices,
                                 uint32_t num_triangles, uint32_t threads, rt_flat_node *nodes, uint32_t *order,
                                 rt_scene_info *info, rt_bvh_stats *stats)
{
    return bvh_build_host_impl("rt_bvh_build_host", positions, num_vertices, indices, num_triangles, threads, nullptr,
                               nodes, order, info, stats);
}

extern "C" int rt_scene_bounds(const rt_scene *sc, float lo[3], float hi[3])
{
    if (!sc || !lo || !hi) return fail(RT_ERR_INVALID, "rt_scene_bounds: NULL argument");
    for (int k = 0; k < 3; k++) {
        lo[k] = sc->bounds_lo[k];
        hi[k] = sc->bounds_hi[k];
    }
    return RT_OK;
}

extern "C" int rt_scene_get_wide_items(const rt_scene *sc, void *units, uint64_t capacity_units, uint64_t *num_units)
{
    if (!sc || !num_units) return fail(RT_ERR_INVALID, "rt_scene_get_wide_items: NULL argument");
    *num_units = sc->dev.witems ? sc->num_wide_units : 0;
    if (!units) return RT_OK;
    if (capacity_units < *num_units)
        return fail(RT_ERR_INVALID, "rt_scene_get_wide_items: room for %llu units, the tree has %llu",
                    (unsigned long long)capacity_units, (unsigned long long)*num_units);
    if (*num_units) HIP_TRY(hipMemcpy(units, sc->dev.witems, (size_t)*num_units * sizeof(Unit), hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_scene_fast_tree_on_device(const rt_scene *sc, int *out)
{
    if (!sc || !out) return fail(RT_ERR_INVALID, "rt_scene_fast_tree_on_device: NULL argument");
    *out = sc->fast_items_on_device ? 2 : sc->fast_tree_on_device ? 1 : 0;
    return RT_OK;
}

extern "C" int rt_bvh_build_search_device(rt_ctx *ctx, const float *positions, uint32_t num_vertices,
                                          const uint32_t *indices, uint32_t num_triangles, uint32_t buckets,
                                          uint32_t max_leaf, rt_flat_node *nodes, uint32_t *order, rt_scene_info *info,
                                          rt_bvh_stats *stats)
{
    if (!ctx || !positions || !indices || !nodes || !order)
        return fail(RT_ERR_INVALID, "rt_bvh_build_search_device: NULL argument");
    if (num_triangles == 0 || num_vertices == 0) return fail(RT_ERR_INVALID, "rt_bvh_build_search_device: empty mesh");
    if (buckets < 2 || buckets > 64 || max_leaf < 1 || max_leaf > 8)
        return fail(RT_ERR_INVALID, "rt_bvh_build_search_device: buckets 2..64, max_leaf 1..8");
    for (size_t i = 0; i < (size_t)num_triangles * 3; i++)
        if (indices[i] >= num_vertices)
            return fail(RT_ERR_INVALID, "rt_bvh_build_search_device: index %u at %zu out of range", indices[i], i);
    if (const long long bad = first_non_finite(positions, num_vertices); bad >= 0)
        return fail(RT_ERR_INVALID, "rt_bvh_build_search_device: vertex %lld has a NaN or infinite coordinate", bad / 3);
    int rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;
    HostBvh hb;
    std::string why;
    try {
        if (!build_bvh_search_device(positions, num_vertices, indices, num_triangles, BuildTuning{buckets, max_leaf}, hb, why))
            return fail(RT_ERR_HIP, "rt_bvh_build_search_device: %s", why.c_str());
    } catch (const std::exception &ex) {
        return fail(RT_ERR_NOMEM, "rt_bvh_build_search_device: %s", ex.what());
    }
    if (!hb.nodes.empty()) memcpy(nodes, hb.nodes.data(), hb.nodes.size() * sizeof(rt_flat_node));
    memcpy(order, hb.order.data(), hb.order.size() * sizeof(uint32_t));
    if (info) {
        info->num_triangles = num_triangles;
        info->num_inner_nodes = (uint32_t)hb.nodes.size();
        info->root_ref = hb.root_ref;
        info->root_count = hb.root_count;
        info->device_bytes = 0;
    }
    if (stats) *stats = hb.stats;
    return RT_OK;
}

extern "C" int rt_bvh_build_search_host(const float *positions, uint32_t num_vertices, const uint32_t *indices,
                                        uint32_t num_triangles, uint32_t threads, uint32_t buckets, uint32_t max_leaf,
                                        rt_flat_node *nodes, uint32_t *order, rt_scene_info *info, rt_bvh_stats *stats)
{
    if (buckets < 2 || buckets > 64 || max_leaf < 1 || max_leaf > 8)
        return fail(RT_ERR_INVALID, "rt_bvh_build_search_host: buckets 2..64, max_leaf 1..8");
    const BuildTuning tune{buckets, max_leaf};
    return bvh_build_host_impl("rt_bvh_build_search_host", positions, num_vertices, indices, num_triangles, threads,
                               &tune, nodes, order, info, stats);
}

extern "C" int rt_bvh_wide_host(const rt_flat_node *nodes, uint32_t num_nodes, uint32_t root_ref,
                                uint32_t root_count, uint32_t num_triangles, rt_wide_node *out, rt_qwide_node *qout,
                                uint32_t *node_unit, uint32_t *slot_unit, uint32_t *num_wide, uint32_t *wide_root_ref,
                                uint32_t *wide_depth, uint32_t *num_units)
{
    if ((num_nodes && !nodes) || !num_wide || !wide_root_ref)
        return fail(RT_ERR_INVALID, "rt_bvh_wide_host: NULL argument");
    for (uint32_t i = 0; i < num_nodes; i++) {
        const uint32_t refs[2] = {nodes[i].left, nodes[i].right};
        for (uint32_t ref : refs)
            if (!(ref & kLeafBit) && ref >= num_nodes)
                return fail(RT_ERR_INVALID, "rt_bvh_wide_host: node %u refers to node %u of %u", i, ref, num_nodes);
    }
    if (!(root_ref & kLeafBit) && root_ref >= num_nodes)
        return fail(RT_ERR_INVALID, "rt_bvh_wide_host: root %u of %u nodes", root_ref, num_nodes);
    try {
        std::vector<DevNode> dn;
        std::vector<BigLeaf> big;
        const uint32_t root = make_dev_nodes(nodes, num_nodes, root_ref, root_count, dn, big);
        std::vector<DevWideNode> wn;
        uint32_t depth = 0;
        *wide_root_ref = collapse_wide4(dn.data(), dn.size(), root, wn, &depth);
        *num_wide = (uint32_t)wn.size();
        if (wide_depth) *wide_depth = depth;
        static_assert(sizeof(rt_wide_node) == sizeof(DevWideNode), "rt_wide_node is the device layout");
        if (out && !wn.empty()) memcpy(out, wn.data(), wn.size() * sizeof(DevWideNode));
        if (qout || node_unit || slot_unit || num_units) {
            WideLayout layout;
            if (!layout_wide4(wn, *wide_root_ref, num_triangles, layout))
                return fail(RT_ERR_INVALID,
                            "rt_bvh_wide_host: the device form needs leaves of at most %u triangles inside [0, %u) "
                            "(refine the tree first)", kWideLeafTris, num_triangles);
            static_assert(sizeof(rt_qwide_node) == sizeof(DevQNode), "rt_qwide_node is the device layout");
            if (qout)
                for (size_t w = 0; w < wn.size(); w++) {
                    const DevQNode q = quantize_node(wn[w], layout);
                    memcpy(&qout[w], &q, sizeof q);
                }
            if (node_unit && !wn.empty()) memcpy(node_unit, layout.node_unit.data(), wn.size() * sizeof(uint32_t));
            if (slot_unit && num_triangles) memcpy(slot_unit, layout.slot_unit.data(), (size_t)num_triangles * sizeof(uint32_t));
            if (num_units) *num_units = layout.num_units;
        }
    } catch (const std::exception &ex) {
        return fail(RT_ERR_NOMEM, "rt_bvh_wide_host: %s", ex.what());
    }
    return RT_OK;
}

extern "C" int rt_bvh_refine_host(const float *positions, uint32_t num_vertices, const uint32_t *indices,
                                  uint32_t num_triangles, const rt_flat_node *nodes, uint32_t num_nodes,
                                  uint32_t root_ref, uint32_t root_count, const uint32_t *order, uint32_t max_tris,
                                  rt_flat_node *xnodes, uint32_t *xorder, uint32_t *num_xnodes, uint32_t *xroot_ref,
                                  uint32_t *xroot_count, uint32_t *extra_depth)
{
    if (!positions || !indices || (num_nodes && !nodes) || !order || !xnodes || !xorder || !num_xnodes || !xroot_ref ||
        !xroot_count)
        return fail(RT_ERR_INVALID, "rt_bvh_refine_host: NULL argument");
    if (max_tris < 1 || max_tris > 8) return fail(RT_ERR_INVALID, "rt_bvh_refine_host: max_tris %u outside [1, 8]", max_tris);
    for (size_t i = 0; i < (size_t)num_triangles * 3; i++)
        if (indices[i] >= num_vertices) return fail(RT_ERR_INVALID, "rt_bvh_refine_host: index out of range");
    for (uint32_t i = 0; i < num_triangles; i++)
        if (order[i] >= num_triangles) return fail(RT_ERR_INVALID, "rt_bvh_refine_host: order[%u] out of range", i);
    if (const long long bad = first_non_finite(positions, num_vertices); bad >= 0)
        return fail(RT_ERR_INVALID, "rt_bvh_refine_host: vertex %lld has a NaN or infinite coordinate", bad / 3);
    try {
        std::vector<DevNode> dn, xdn;
        std::vector<BigLeaf> big;
        const uint32_t root = make_dev_nodes(nodes, num_nodes, root_ref, root_count, dn, big);
        std::vector<uint32_t> ord(order, order + num_triangles), xord;
        uint32_t xroot = root;
        const uint32_t extra = refine_leaves(dn, root, positions, indices, ord, max_tris, big, xdn, xroot, xord);
        // back to the canonical encoding: leaf = bit 31 | first slot, with a separate count
        auto unpack = [&](uint32_t ref, uint32_t &out_ref, uint32_t &out_count) {
            if (!(ref & kLeafBit)) {
                out_ref = ref;
                out_count = 0;
            } else if (ref & kBigLeafBit) {
                const BigLeaf &bl = big[ref & 0x3FFFFFFFu];
                out_ref = kLeafBit | bl.first;
                out_count = bl.count;
            } else {
                out_ref = kLeafBit | ((ref >> 3) & (kMaxPackedFirst - 1u));
                out_count = (ref & 7u) + 1u;
            }
        };
        for (size_t i = 0; i < xdn.size(); i++) {
            rt_flat_node &o = xnodes[i];
            memcpy(o.bx, xdn[i].bx, sizeof o.bx);
            memcpy(o.by, xdn[i].by, sizeof o.by);
            memcpy(o.bz, xdn[i].bz, sizeof o.bz);
            unpack(xdn[i].left, o.left, o.left_count);
            unpack(xdn[i].right, o.right, o.right_count);
        }
        memcpy(xorder, xord.data(), xord.size() * sizeof(uint32_t));
        *num_xnodes = (uint32_t)xdn.size();
        unpack(xroot, *xroot_ref, *xroot_count);
        if (extra_depth) *extra_depth = extra;
    } catch (const std::exception &ex) {
        return fail(RT_ERR_NOMEM, "rt_bvh_refine_host: %s", ex.what());
    }
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene *sc)
{
    if (!sc) return;
    (void)hipSetDevice(sc->ctx->device);
    (void)hipDeviceSynchronize();
    sc->nodes.release();
    sc->witems.release();
    sc->tris.release();
    sc->big.release();
    sc->normals.release();
    delete sc;
}

extern "C" int rt_scene_stats(const rt_scene *sc, rt_bvh_stats *out)
{
    if (!sc || !out) return fail(RT_ERR_INVALID, "rt_scene_stats: NULL argument");
    *out = sc->host.stats;
    return RT_OK;
}

extern "C" int rt_scene_get_info(const rt_scene *sc, rt_scene_info *out)
{
    if (!sc || !out) return fail(RT_ERR_INVALID, "rt_scene_get_info: NULL argument");
    out->num_triangles = sc->num_triangles;
    out->num_inner_nodes = (uint32_t)sc->host.nodes.size();
    out->root_ref = sc->host.root_ref;
    out->root_count = sc->host.root_count;
    out->device_bytes = sc->device_bytes;
    return RT_OK;
}

extern "C" int rt_scene_get_flat(const rt_scene *sc, rt_flat_node *nodes, rt_flat_tri *tris)
{
    if (!sc) return fail(RT_ERR_INVALID, "rt_scene_get_flat: scene is NULL");
    if (nodes && !sc->host.nodes.empty())
        memcpy(nodes, sc->host.nodes.data(), sc->host.nodes.size() * sizeof(rt_flat_node));
    if (tris) {
        int rc = ctx_set_device(sc->ctx);
        if (rc != RT_OK) return rc;
        std::vector<DevTri> dt(sc->num_triangles);
        HIP_TRY(hipMemcpy(dt.data(), sc->tris.p, dt.size() * sizeof(DevTri), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < sc->num_triangles; i++) {
            memcpy(tris[i].v0, dt[i].v0, sizeof dt[i].v0);
            memcpy(tris[i].v0v1, dt[i].e1, sizeof dt[i].e1);
            memcpy(tris[i].v0v2, dt[i].e2, sizeof dt[i].e2);
            tris[i].id = dt[i].id;
            tris[i].pad[0] = tris[i].pad[1] = 0;
        }
    }
    return RT_OK;
}

// ------------------------------------------------------------------------
// trace on StreamRay arrays

static void stats_from_dev(const StatsDev &d, rt_trace_stats *s)
{
    s->total_rays += d.total_rays;
    s->inner_visit += d.inner_rays;
    s->inner_rays += d.inner_rays;
    s->leaf_visit += d.leaf_rays;
    s->leaf_rays += d.leaf_rays;
    s->leaf_objs += d.obj_intersect;
    s->obj_intersect += d.obj_intersect;
}

constexpr unsigned kModeOfContext = 0xFFFFFFFFu; // trace_stream_device: use the context's "traversal_stream"

static int trace_stream_device(rt_scene *sc, bool anyhit, rt_stream_ray_dev *d_rays, uint32_t *d_hits,
                               uint8_t *d_occ, uint32_t n, rt_trace_stats *stats, hipStream_t s, bool sync_stats,
                               unsigned mode = kModeOfContext)
{
    rt_ctx *ctx = sc->ctx;
    if (mode == kModeOfContext) mode = ctx->trav_stream;
    bool want = stats != nullptr;
    if (want) HIP_TRY(hipMemsetAsync(ctx->dstats.p, 0, 2 * sizeof(StatsDev), s));
    size_t slot;
    int rc = timed_begin(ctx, anyhit ? 1 : 0, s, &slot);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemsetAsync(ctx->fetch.p, 0, kFetchUints * sizeof(unsigned), s));
    TraceLaunch tl{(unsigned *)ctx->fetch.p, (unsigned *)ctx->ovf.p, (StatsDev *)ctx->dstats.p + (anyhit ? 1 : 0),
                   ctx->min_active[1][anyhit ? 1 : 0], mode, nullptr, 0, nullptr, 0, 0, 0};
    fill_coop(ctx, sc, tl);
    launch_trace_stream(anyhit, want, sc->dev, d_rays, d_hits, d_occ, n, tl, ctx->max_blocks, s);
    HIP_TRY(hipGetLastError());
    rc = timed_end(ctx, slot, s);
    if (rc != RT_OK) return rc;
    if (want && sync_stats) {
        StatsDev h[2]{};
        HIP_TRY(hipMemcpyAsync(h, ctx->dstats.p, sizeof h, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int k = 0; k < 2; k++) {
            ctx->last_split[k] = rt_trace_stats{};
            ctx->last_max_steps[k] = h[k].max_ray_steps;
            ctx->last_max_stack[k] = h[k].max_stack_entries;
            stats_from_dev(h[k], &ctx->last_split[k]);
            stats_from_dev(h[k], stats);
        }
    }
    return RT_OK;
}

extern "C" int rt_intersect_stream_device(rt_scene *sc, rt_stream_ray *d_rays, uint32_t *d_hits, uint32_t n,
                                          rt_trace_stats *stats, void *hip_stream)
{
    if (!sc) return fail(RT_ERR_INVALID, "rt_intersect_stream_device: scene is NULL");
    if (n && (!d_rays || !d_hits)) return fail(RT_ERR_INVALID, "rt_intersect_stream_device: NULL buffer");
    int rc = ctx_set_device(sc->ctx);
    if (rc != RT_OK) return rc;
    sc->ctx->used = 0;
    return trace_stream_device(sc, false, (rt_stream_ray_dev *)d_rays, d_hits, nullptr, n, stats,
                               (hipStream_t)hip_stream, true);
}

extern "C" int rt_occluded_stream_device(rt_scene *sc, const rt_stream_ray *d_rays, uint8_t *d_occluded,
                                         uint32_t n, rt_trace_stats *stats, void *hip_stream)
{
    if (!sc) return fail(RT_ERR_INVALID, "rt_occluded_stream_device: scene is NULL");
    if (n && (!d_rays || !d_occluded)) return fail(RT_ERR_INVALID, "rt_occluded_stream_device: NULL buffer");
    int rc = ctx_set_device(sc->ctx);
    if (rc != RT_OK) return rc;
    sc->ctx->used = 0;
    return trace_stream_device(sc, true, (rt_stream_ray_dev *)d_rays, nullptr, d_occluded, n, stats,
                               (hipStream_t)hip_stream, true);
}

extern "C" int rt_intersect_stream_noalloc(rt_scene *sc, rt_stream_ray *d_rays, uint32_t *d_hits, uint32_t n,
                                           void *hip_stream)
{
    if (!sc) return fail(RT_ERR_INVALID, "rt_intersect_stream_noalloc: scene is NULL");
    if (n && (!d_rays || !d_hits)) return fail(RT_ERR_INVALID, "rt_intersect_stream_noalloc: NULL buffer");
    if (n > RT_NOALLOC_MAX_RAYS)
        return fail(RT_ERR_LIMIT, "rt_intersect_stream_noalloc: %u rays, limit is %u per call", n, RT_NOALLOC_MAX_RAYS);
    rt_ctx *ctx = sc->ctx;
    {
        // deferred nodes a ray of the selected form can hold: one sibling per level of the binary tree, up to
        // three children per level of the wide one
        const unsigned form = ctx->trav_stream;
        const bool wide = (form == 2 || form == 3 || form >= 5) && sc->dev.witems;
        const unsigned bound = wide ? 3u * sc->dev.wmax_depth + 1u : sc->dev.max_depth + 1u;
        if (bound > RT_NOALLOC_MAX_STACK)
            return fail(RT_ERR_LIMIT,
                        "rt_intersect_stream_noalloc: rays of traversal form %u can defer %u nodes on this scene, the "
                        "fixed scratch holds %u",
                        form, bound, (unsigned)RT_NOALLOC_MAX_STACK);
    }
    int rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    unsigned *fetch = reinterpret_cast<unsigned *>(ctx->na_scratch);
    unsigned *ovf = reinterpret_cast<unsigned *>(ctx->na_scratch + kNaFetchBytes);
    // the fetch counters must start at zero; the stack area needs no clearing (entries are written
    // before they are read)
    HIP_TRY(hipMemsetAsync(fetch, 0, kNaFetchBytes, s));
    // the two-ray form keeps two stacks per lane and the queued form fewer LDS entries than the scratch is sized for
    const unsigned mode = (ctx->trav_stream == 5 || ctx->trav_stream == 7) ? 2u : ctx->trav_stream;
    // no statistics, no launch timing (events are created on demand), no straggler hand-off (its
    // work stacks are far larger than this scratch)
    TraceLaunch tl{fetch, ovf, nullptr, ctx->min_active[1][0], mode, nullptr, 0, nullptr, 0, 0xFFFFFFFFu, 0};
    launch_trace_stream(false, false, sc->dev, (rt_stream_ray_dev *)d_rays, d_hits, nullptr, n, tl, kNaBlocks, s);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

namespace {

constexpr uint32_t kHostChunk = 1u << 20; // rays per chunk of the host-buffer entry points

void add_stats(rt_trace_stats &a, const rt_trace_stats &b)
{
    a.total_rays += b.total_rays;
    a.inner_visit += b.inner_visit;
    a.inner_rays += b.inner_rays;
    a.leaf_visit += b.leaf_visit;
    a.leaf_rays += b.leaf_rays;
    a.leaf_objs += b.leaf_objs;
    a.obj_intersect += b.obj_intersect;
}

int ensure_host_pipeline(rt_ctx *ctx, uint32_t n)
{
    if (!ctx->copy_stream) {
        if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess)
            return fail(RT_ERR_HIP, "hipStreamCreate failed");
        for (int k = 0; k < 2; k++)
            if (hipEventCreateWithFlags(&ctx->h2d_done[k], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->kernel_done[k], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->out_done[k], hipEventDisableTiming) != hipSuccess)
                return fail(RT_ERR_HIP, "hipEventCreate failed");
    }
    const size_t chunk = std::min<size_t>(n, kHostChunk);
    const int bufs = n > kHostChunk ? 2 : 1;
    for (int k = 0; k < bufs; k++) {
        int rc = ctx->chunk_rays[k].ensure(chunk * sizeof(rt_stream_ray));
        if (rc == RT_OK) rc = ctx->chunk_out[k].ensure(chunk * 8);
        if (rc != RT_OK) return rc;
        if (!ctx->h_out[k] && hipHostMalloc(&ctx->h_out[k], (size_t)kHostChunk * 8) != hipSuccess)
            return fail(RT_ERR_NOMEM, "pinned host allocation failed");
    }
    return RT_OK;
}

// Closest-hit (hits != NULL) or any-hit (occluded != NULL) over host rays, chunk by chunk.
int host_stream_trace(rt_scene *sc, rt_stream_ray *rays, const rt_stream_ray *rays_in, uint32_t *hits, uint8_t *occluded,
                      uint32_t n, rt_trace_stats *stats, unsigned mode = kModeOfContext)
{
    rt_ctx *ctx = sc->ctx;
    int rc = ctx_set_device(ctx);
    if (rc == RT_OK) rc = ensure_host_pipeline(ctx, n);
    if (rc != RT_OK) return rc;
    const bool anyhit = occluded != nullptr;
    hipStream_t s = ctx->stream, cs = ctx->copy_stream;
    const uint32_t chunks = (n + kHostChunk - 1) / kHostChunk;
    auto count = [&](uint32_t k) { return std::min(kHostChunk, n - k * kHostChunk); };
    const bool single = chunks == 1; // one chunk (a tile of the reference is 32,768 rays): one stream, no events
    // Results of chunk k: pinned buffer -> the caller's arrays.  (Doing this on a helper thread beside
    // the next upload was measured slower: both are bound by host memory bandwidth.)
    auto scatter = [&](uint32_t k) {
        const int b = (int)(k & 1u);
        const uint32_t m = count(k);
        const size_t at = (size_t)k * kHostChunk;
        if (anyhit) {
            memcpy(occluded + at, ctx->h_out[b], m);
        } else {
            const uint32_t *h = (const uint32_t *)ctx->h_out[b];
            const float *f = (const float *)ctx->h_out[b] + m;
            memcpy(hits + at, h, (size_t)m * sizeof(uint32_t));
            for (uint32_t i = 0; i < m; i++) rays[at + i].origin_far[3] = f[i];
        }
    };
    ctx->used = 0;
    rt_trace_stats total{}, split[2]{};
    uint64_t max_steps[2] = {0, 0}, max_stack[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(ctx->chunk_rays[0].p, rays_in, (size_t)count(0) * sizeof(rt_stream_ray), hipMemcpyHostToDevice,
                           single ? s : cs));
    if (!single) HIP_TRY(hipEventRecord(ctx->h2d_done[0], cs));
    for (uint32_t k = 0; k < chunks; k++) {
        const int b = (int)(k & 1u);
        const uint32_t m = count(k);
        rt_stream_ray_dev *d_rays = (rt_stream_ray_dev *)ctx->chunk_rays[b].p;
        uint32_t *d_hits = (uint32_t *)ctx->chunk_out[b].p;
        float *d_far = (float *)ctx->chunk_out[b].p + m;
        if (!single) HIP_TRY(hipStreamWaitEvent(s, ctx->h2d_done[b], 0));
        rt_trace_stats st{};
        rc = trace_stream_device(sc, anyhit, d_rays, anyhit ? nullptr : d_hits, anyhit ? (uint8_t *)d_hits : nullptr, m,
                                 stats ? &st : nullptr, s, true, mode);
        if (rc != RT_OK) return rc;
        if (stats) {
            add_stats(total, st);
            for (int q = 0; q < 2; q++) { // the per-kind view of the call is the sum over its chunks
                add_stats(split[q], ctx->last_split[q]);
                max_steps[q] = std::max(max_steps[q], ctx->last_max_steps[q]);
                max_stack[q] = std::max(max_stack[q], ctx->last_max_stack[q]);
            }
        }
        if (!anyhit) launch_gather_far(d_rays, d_far, m, s);
        if (!single) HIP_TRY(hipEventRecord(ctx->kernel_done[b], s));
        HIP_TRY(hipMemcpyAsync(ctx->h_out[b], ctx->chunk_out[b].p, anyhit ? (size_t)m : (size_t)m * 8, hipMemcpyDeviceToHost, s));
        if (!single) HIP_TRY(hipEventRecord(ctx->out_done[b], s));
        if (k + 1 < chunks) { // upload the next chunk while this one is traced
            if (k >= 1) HIP_TRY(hipStreamWaitEvent(cs, ctx->kernel_done[b ^ 1], 0)); // its buffer was chunk k-1's
            HIP_TRY(hipMemcpyAsync(ctx->chunk_rays[b ^ 1].p, rays_in + (size_t)(k + 1) * kHostChunk,
                                   (size_t)count(k + 1) * sizeof(rt_stream_ray), hipMemcpyHostToDevice, cs));
            HIP_TRY(hipEventRecord(ctx->h2d_done[b ^ 1], cs));
        }
        if (k >= 1) { // chunk k-1 is back by now: unpack it while chunk k is traced
            HIP_TRY(hipEventSynchronize(ctx->out_done[b ^ 1]));
            scatter(k - 1);
        }
    }
    if (single)
        HIP_TRY(hipStreamSynchronize(s));
    else
        HIP_TRY(hipEventSynchronize(ctx->out_done[(chunks - 1) & 1u]));
    scatter(chunks - 1);
    if (stats) {
        add_stats(*stats, total);
        for (int q = 0; q < 2; q++) {
            ctx->last_split[q] = split[q];
            ctx->last_max_steps[q] = max_steps[q];
            ctx->last_max_stack[q] = max_stack[q];
        }
    }
    return RT_OK;
}

} // namespace

extern "C" int rt_intersect_stream(rt_scene *sc, rt_stream_ray *rays, uint32_t *hits, uint32_t n,
                                   rt_trace_stats *stats)
{
    if (!sc) return fail(RT_ERR_INVALID, "rt_intersect_stream: scene is NULL");
    if (n && (!rays || !hits)) return fail(RT_ERR_INVALID, "rt_intersect_stream: NULL buffer");
    if (n == 0) return RT_OK;
    return host_stream_trace(sc, rays, rays, hits, nullptr, n, stats);
}

extern "C" int rt_occluded_stream(rt_scene *sc, const rt_stream_ray *rays, uint8_t *occluded, uint32_t n,
                                  rt_trace_stats *stats)
{
    if (!sc) return fail(RT_ERR_INVALID, "rt_occluded_stream: scene is NULL");
    if (n && (!rays || !occluded)) return fail(RT_ERR_INVALID, "rt_occluded_stream: NULL buffer");
    if (n == 0) return RT_OK;
    return host_stream_trace(sc, nullptr, rays, nullptr, occluded, n, stats);
}

extern "C" int rt_intersect1(rt_scene *sc, rt_stream_ray *ray, uint32_t *hit, rt_trace_stats *stats)
{
    if (!sc || !ray || !hit) return fail(RT_ERR_INVALID, "rt_intersect1: NULL argument");
    return host_stream_trace(sc, ray, ray, hit, nullptr, 1, stats, 4u);
}

extern "C" int rt_occluded1(rt_scene *sc, const rt_stream_ray *ray, uint8_t *occluded, rt_trace_stats *stats)
{
    if (!sc || !ray || !occluded) return fail(RT_ERR_INVALID, "rt_occluded1: NULL argument");
    return host_stream_trace(sc, nullptr, ray, nullptr, occluded, 1, stats, 1u);
}

// ------------------------------------------------------------------------
// camera frame, src/camera.rs:25-37 and 56-88, on the host in f32

namespace {

struct H3 {
    float x, y, z;
};
inline H3 h3(const float *p) { return H3{p[0], p[1], p[2]}; }
inline H3 hadd(H3 a, H3 b) { return H3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline H3 hsub(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 hscale(H3 a, float s) { return H3{a.x * s, a.y * s, a.z * s}; }
inline float hdot(H3 a, H3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline H3 hcross(H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline H3 hnorm(H3 a)
{
    float n = sqrtf(hdot(a, a));
    return H3{a.x / n, a.y / n, a.z / n};
}
inline void put(float *d, H3 v)
{
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}

CameraDev make_camera(const rt_camera &c, uint32_t w, uint32_t h)
{
    CameraDev o{};
    H3 eye = h3(c.eye);
    H3 fwd = hnorm(hsub(h3(c.target), eye));
    H3 right = hnorm(hcross(fwd, h3(c.up)));
    H3 up = hcross(right, fwd);
    const float pi = 3.14159274f;
    float half_w = tanf(c.fov_deg * (pi / 180.f) / 2.f);
    float half_h = half_w * (((float)h - 1.f) / ((float)w - 1.f));
    H3 step_x = hscale(right, 2.f * half_w / ((float)w - 1.f));
    H3 step_y = hscale(H3{-up.x, -up.y, -up.z}, 2.f * half_h / ((float)h - 1.f));
    H3 tl = hadd(hsub(fwd, hscale(right, half_w)), hscale(up, half_h));
    put(o.eye, eye);
    put(o.right, right);
    put(o.up, up);
    put(o.top_left, tl);
    put(o.step_x, step_x);
    put(o.step_y, step_y);
    o.pixel_size = half_w * 2.f / ((float)w - 1.f);
    return o;
}

inline uint32_t pcg_hash_host(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
inline uint32_t seed_key_of(uint64_t seed)
{
    uint32_t h = pcg_hash_host((uint32_t)seed ^ 0x9E3779B9u);
    return pcg_hash_host(h ^ (uint32_t)(seed >> 32));
}

int check_params(const rt_render_params *p, const char *who)
{
    if (p->width < 2 || p->height < 2 || p->width > 65536 || p->height > 65536)
        return fail(RT_ERR_INVALID, "%s: image size %ux%u outside [2, 65536]", who, p->width, p->height);
    if (p->tile_size < 1 || p->tile_size > 256)
        return fail(RT_ERR_INVALID, "%s: tile_size %u outside [1, 256]", who, p->tile_size);
    if (p->max_bounces > 255) return fail(RT_ERR_INVALID, "%s: max_bounces %u > 255", who, p->max_bounces);
    uint32_t sc = p->shard_count ? p->shard_count : 1;
    if (p->shard_index >= sc)
        return fail(RT_ERR_INVALID, "%s: shard_index %u >= shard_count %u", who, p->shard_index, sc);
    if ((uint64_t)p->sample_offset + p->samples > 0xFFFFFFFFull)
        return fail(RT_ERR_INVALID, "%s: sample range overflows u32", who);
    return RT_OK;
}

} // namespace

// ------------------------------------------------------------------------
// render

static int ensure_workspace(rt_ctx *ctx, uint64_t paths, uint32_t ncounters, bool sorting, unsigned sort_bits)
{
    Workspace &w = ctx->ws;
    int rc = RT_OK;
    if (ctx->workspace_limit) {
        const uint64_t per_path = 2 * sizeof(QRay) + 2 * sizeof(PathInfo) + 2 * sizeof(uint32_t) + sizeof(QRay) +
                                  sizeof(ShadowInfo) + 3 * sizeof(float) + (sorting ? 5 * sizeof(uint32_t) : 0);
        if (paths * per_path > ctx->workspace_limit)
            return fail(RT_ERR_NOMEM, "render workspace of %llu paths exceeds the workspace_limit_mb option",
                        (unsigned long long)paths);
    }
    for (int i = 0; i < 2 && rc == RT_OK; i++) rc = w.rays[i].ensure(paths * sizeof(QRay));
    for (int i = 0; i < 2 && rc == RT_OK; i++) rc = w.infos[i].ensure(paths * sizeof(PathInfo));
    if (rc == RT_OK) rc = w.hit_ids.ensure(paths * sizeof(uint32_t));
    if (rc == RT_OK) rc = w.hit_t.ensure(paths * sizeof(float));
    if (rc == RT_OK) rc = w.shadow_rays.ensure(paths * sizeof(QRay));
    if (rc == RT_OK) rc = w.shadow_infos.ensure(paths * sizeof(ShadowInfo));
    if (rc == RT_OK) rc = w.rad.ensure(paths * 3 * sizeof(float));
    if (rc == RT_OK) rc = w.counters.ensure((size_t)ncounters * sizeof(uint32_t));
    if (rc == RT_OK && sorting) {
        for (int i = 0; i < 2 && rc == RT_OK; i++) rc = w.keys[i].ensure(paths * sizeof(uint32_t));
        for (int i = 0; i < 2 && rc == RT_OK; i++) rc = w.perm[i].ensure(paths * sizeof(uint32_t));
        if (rc == RT_OK) rc = w.keys_sorted.ensure(paths * sizeof(uint32_t));
        if (rc == RT_OK) {
            size_t tb = sort_temp_bytes((unsigned)paths, 30u - sort_bits, 32u);
            if (tb == 0) return fail(RT_ERR_HIP, "radix sort temporary-storage query failed");
            rc = w.sort_temp.ensure(tb);
        }
    }
    if (rc == RT_OK) w.paths = std::max(w.paths, paths);
    return rc;
}

extern "C" int rt_render_device(rt_scene *sc, const rt_camera *camera, const rt_light *light,
                                const rt_render_params *params, float *d_accum, rt_trace_stats *stats,
                                void *hip_stream)
{
    if (!sc || !camera || !light || !params || !d_accum)
        return fail(RT_ERR_INVALID, "rt_render_device: NULL argument");
    int rc = check_params(params, "rt_render_device");
    if (rc != RT_OK) return rc;
    rt_ctx *ctx = sc->ctx;
    rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    ctx->used = 0;
    if (params->samples == 0) return RT_OK;

    const uint32_t ts = params->tile_size, ts2 = ts * ts;
    const uint32_t tiles_x = (params->width + ts - 1) / ts, tiles_y = (params->height + ts - 1) / ts;
    const uint64_t ntiles = (uint64_t)tiles_x * tiles_y;
    const uint32_t shard_count = params->shard_count ? params->shard_count : 1;
    const uint32_t my_tiles =
        (uint32_t)((ntiles + shard_count - 1 - params->shard_index) / shard_count); // i % sc == si
    if (my_tiles == 0) return RT_OK;

    // pass plan: tiles_per_pass tiles x k samples <= cap slots.  The cap starts at max_paths; when the
    // workspace for it cannot be allocated (a smaller or shared GPU, a large scene resident) what was
    // grown so far is released and the frame is planned again with half the pass, down to one tile
    // or 2^16 paths.
    const uint32_t nb = params->max_bounces + 1; // extend launches per pass
    // per pass: ray count per depth (nb+1), shadow count per depth (nb), then the
    // ray-fetch counters of the 2*nb trace launches; zeroed by one memset
    const uint32_t nqueue = (2 * nb + 2) * kCounterStride;
    const uint32_t ncounters = nqueue + 2 * nb * kFetchUints;
    // a queue consumed at depth d is sorted when d >= sort_from_depth
    const bool sort_any = (nb > ctx->sort_from_depth[0]) || (nb > ctx->sort_from_depth[1]);
    uint64_t cap = ctx->max_paths;
    if (cap < ts2) return fail(RT_ERR_LIMIT, "rt_render_device: max_paths %llu below one tile (%u)",
                               (unsigned long long)cap, ts2);
    uint32_t k = 0, tiles_per_pass = 0;
    uint64_t slots = 0;
    for (;;) {
        k = params->samples_per_pass ? params->samples_per_pass : 0;
        if ((uint64_t)my_tiles * ts2 <= cap) {
            tiles_per_pass = my_tiles;
            uint64_t kmax = cap / ((uint64_t)my_tiles * ts2);
            if (k == 0 || k > kmax) k = (uint32_t)std::min<uint64_t>(kmax, 0xFFFFFFFFu);
        } else {
            k = 1;
            tiles_per_pass = (uint32_t)(cap / ts2);
        }
        k = std::min(k, params->samples);
        if (k < 1) k = 1;
        slots = (uint64_t)tiles_per_pass * ts2 * k;
        if (slots > 0xFFFFFFFFull) return fail(RT_ERR_LIMIT, "rt_render_device: pass of %llu paths exceeds 2^32",
                                               (unsigned long long)slots);
        rc = ensure_workspace(ctx, slots, ncounters, sort_any, ctx->sort_bits);
        if (rc != RT_ERR_NOMEM) break;
        const uint64_t floor_paths = std::max<uint64_t>(ts2, 1u << 16);
        if (slots <= floor_paths) break; // nothing smaller to try: the error stands
        HIP_TRY(hipStreamSynchronize(s)); // nothing of this context may still be using the buffers
        ctx->ws.release_all();
        cap = std::max<uint64_t>(floor_paths, slots / 2);
    }
    if (rc != RT_OK) return rc;
    ctx->last_pass_paths = slots;
    Workspace &w = ctx->ws;

    const CameraDev cam = make_camera(*camera, params->width, params->height);
    const uint32_t seed_key = seed_key_of(params->seed);
    const bool want = stats != nullptr;
    const bool visit = want && (params->flags & RT_RENDER_VISIT_STATS);
    StatsDev *dstats = (StatsDev *)ctx->dstats.p; // [0] extend, [1] connect
    if (want) HIP_TRY(hipMemsetAsync(dstats, 0, 2 * sizeof(StatsDev), s));

    float *rad_r = (float *)w.rad.p, *rad_g = rad_r + slots, *rad_b = rad_g + slots;
    uint32_t *counters = (uint32_t *)w.counters.p;
    unsigned *ovf = (unsigned *)ctx->ovf.p;
    // any-hit launches on their own stream, with their own overflow area and straggler work stacks
    const bool overlap = ctx->overlap_connect && !visit;
    hipStream_t sconn = overlap ? ctx->conn_stream : s;
    unsigned *ovf_conn = overlap ? ovf + (size_t)ctx->max_blocks * RTC_TRACE_BLOCK * ctx->ovf_depth : ovf;
    if (overlap)
        while (ctx->ev_shade.size() < nb) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            ctx->ev_shade.push_back(a);
            HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            ctx->ev_conn.push_back(b);
        }

    const uint64_t plan_key[3] = {(uint64_t)(uintptr_t)sc, slots, nb};
    for (uint32_t tile0 = 0; tile0 < my_tiles; tile0 += tiles_per_pass) {
        const uint32_t tcount = std::min(tiles_per_pass, my_tiles - tile0);
        for (uint32_t s0 = 0; s0 < params->samples; s0 += k) {
            PassDev ps{};
            ps.width = params->width;
            ps.height = params->height;
            ps.tile_size = ts;
            ps.tiles_x = tiles_x;
            ps.shard_index = params->shard_index;
            ps.shard_count = shard_count;
            ps.tile_begin = tile0;
            ps.tile_count = tcount;
            ps.samples_per_pass = k;
            ps.samples_this_pass = std::min(k, params->samples - s0);
            ps.sample_base = params->sample_offset + s0;
            ps.num_slots = tcount * ts2 * k;
            fill_pass_shifts(ps);
            ps.seed_key = seed_key;

            // queue lengths of the previous pass, if its read-back has landed (never waited for)
            if (ctx->counts_pending && hipEventQuery(ctx->counts_ready) == hipSuccess) {
                const size_t words = 2 * (size_t)ctx->pending_key[2] + 1;
                ctx->est_counts.resize(words);
                for (size_t i = 0; i < words; i++) ctx->est_counts[i] = ctx->h_counts[i * kCounterStride];
                memcpy(ctx->est_key, ctx->pending_key, sizeof ctx->est_key);
                ctx->counts_pending = false;
            }
            const bool have_est = !ctx->est_counts.empty() && ctx->est_counts.size() == 2 * nb + 1 &&
                                  memcmp(ctx->est_key, plan_key, sizeof plan_key) == 0;
            // entries of a queue worth sorting: last pass's length plus 2 % and a block, all if unknown
            auto sort_len = [&](uint32_t index) -> uint32_t {
                if (!have_est) return ps.num_slots;
                uint64_t e = (uint64_t)ctx->est_counts[index] + ctx->est_counts[index] / 50u + 4096u;
                return (uint32_t)std::min<uint64_t>(e, ps.num_slots);
            };
            HIP_TRY(hipMemsetAsync(counters, 0, (size_t)ncounters * sizeof(uint32_t), s));
            auto n_ray = [&](uint32_t i) { return counters + (size_t)i * kCounterStride; };            // [0..nb]
            auto n_sh = [&](uint32_t i) { return counters + (size_t)(nb + 1 + i) * kCounterStride; }; // [0..nb-1]
            uint32_t *fetch = counters + nqueue; // [2*nb][kFetchUints]
            int cur = 0;
            size_t slot;
            const unsigned *perm_extend = nullptr; // order of the ray queue about to be extended
            unsigned perm_extend_n = 0;
            const bool fused = ctx->fuse_primary && !visit;
            rc = timed_begin(ctx, 2, s, &slot);
            if (rc != RT_OK) return rc;
            if (fused) {
                // no generate pass: the radiance slots are cleared and the number of primary rays
                // (slots inside the image) is known on the host
                HIP_TRY(hipMemsetAsync(rad_r, 0, (size_t)slots * 3 * sizeof(float), s));
                uint64_t valid_px = 0;
                for (uint32_t t = 0; t < tcount; t++) {
                    const uint64_t tile = params->shard_index + (uint64_t)(tile0 + t) * shard_count;
                    const uint32_t tx = (uint32_t)(tile % tiles_x), ty = (uint32_t)(tile / tiles_x);
                    valid_px += (uint64_t)std::min(ts, params->width - tx * ts) * std::min(ts, params->height - ty * ts);
                }
                launch_set_count(n_ray(0), (unsigned)(valid_px * ps.samples_this_pass), s);
            } else {
                launch_generate(cam, ps, (QRay *)w.rays[cur].p, (PathInfo *)w.infos[cur].p, n_ray(0), rad_r, rad_g,
                                rad_b, s);
            }
            rc = timed_end(ctx, slot, s);
            if (rc != RT_OK) return rc;
            for (uint32_t b = 0; b < nb; b++) {
                rc = timed_begin(ctx, 0, s, &slot);
                if (rc != RT_OK) return rc;
                TraceLaunch tle{fetch + (size_t)(2 * b) * kFetchUints, ovf, dstats, ctx->min_active[b == 0 ? 0 : 1][0],
                                ctx->trav_mode[b == 0 ? 0 : 1], perm_extend, perm_extend_n, nullptr, 0, 0, b};
                fill_coop(ctx, sc, tle);
                if (fused && b == 0)
                    launch_extend_primary(sc->dev, cam, ps, (uint32_t *)w.hit_ids.p, (float *)w.hit_t.p, tle,
                                          ctx->max_blocks, s);
                else
                    launch_extend(visit, sc->dev, (QRay *)w.rays[cur].p, (uint32_t *)w.hit_ids.p, n_ray(b),
                                  ps.num_slots, tle, ctx->max_blocks, s);
                rc = timed_end(ctx, slot, s);
                if (rc != RT_OK) return rc;
                if (want) launch_count_rays(n_ray(b), dstats, b, !visit, s);
                // shade(b) rewrites the shadow queue and its sort buffers: the any-hit launch of depth b - 1,
                // which ran beside this extend launch, must be through with them
                if (overlap && b > 0) HIP_TRY(hipStreamWaitEvent(s, ctx->ev_conn[b - 1], 0));

                ShadeArgs a{};
                a.rays_in = (const QRay *)w.rays[cur].p;
                a.infos_in = (const PathInfo *)w.infos[cur].p;
                a.hit_ids = (const uint32_t *)w.hit_ids.p;
                a.n_in = n_ray(b);
                a.depth = b;
                a.ps = ps; // the pass geometry: pixel and sample of a path follow from its slot
                if (fused && b == 0) {
                    a.primary = 1;
                    a.n_in = nullptr;
                    a.n_in_val = ps.num_slots;
                    a.hit_t = (const float *)w.hit_t.p;
                    a.cam = cam;
                }
                a.normals = sc->dev.normals;
                a.rays_out = (QRay *)w.rays[cur ^ 1].p;
                a.infos_out = (PathInfo *)w.infos[cur ^ 1].p;
                a.n_out = n_ray(b + 1);
                a.shadow_rays = (QRay *)w.shadow_rays.p;
                a.shadow_infos = (ShadowInfo *)w.shadow_infos.p;
                a.n_shadow = n_sh(b);
                memcpy(a.light_pos, light->pos, sizeof a.light_pos);
                memcpy(a.light_intensity, light->intensity, sizeof a.light_intensity);
                a.max_bounces = params->max_bounces;
                a.seed_key = seed_key;
                // the shadow queue is consumed at depth b, the bounce queue at depth b+1
                const bool sort_shadow = b >= ctx->sort_from_depth[1];
                const bool sort_bounce = (b + 1 < nb) && (b + 1 >= ctx->sort_from_depth[0]);
                for (int k = 0; k < 3; k++) {
                    a.key_lo[k] = sc->bounds_lo[k];
                    a.key_scale[k] = sc->key_scale[k];
                }
                a.key_mask = (0x3FFFFFFFu >> (30u - ctx->sort_bits)) << (30u - ctx->sort_bits);
                a.key_dir_shift = ctx->sort_octant ? 30u - ctx->sort_bits : 32u;
                a.shadow_keys = sort_shadow ? (uint32_t *)w.keys[1].p : nullptr;
                a.bounce_keys = sort_bounce ? (uint32_t *)w.keys[0].p : nullptr;
                rc = timed_begin(ctx, 2, s, &slot);
                if (rc != RT_OK) return rc;
                // the shadow queue of depth b has counter index nb+1+b, the bounce queue b+1
                const uint32_t m_shadow = sort_shadow ? sort_len(nb + 1 + b) : 0;
                const uint32_t m_bounce = sort_bounce ? sort_len(b + 1) : 0;
                // unused entries keep the largest key so that they sort behind every ray
                if (sort_shadow) HIP_TRY(hipMemsetAsync(w.keys[1].p, 0xFF, (size_t)m_shadow * sizeof(uint32_t), s));
                if (sort_bounce) HIP_TRY(hipMemsetAsync(w.keys[0].p, 0xFF, (size_t)m_bounce * sizeof(uint32_t), s));
                launch_shade(a, ps.num_slots, s);
                const unsigned *perm_connect = nullptr;
                unsigned perm_connect_n = 0;
                if (sort_shadow || sort_bounce) {
                    size_t tb = sort_temp_bytes(std::max(m_shadow, m_bounce), 30u - ctx->sort_bits, 32u);
                    if (tb == 0) return fail(RT_ERR_HIP, "radix sort temporary-storage query failed");
                    tb = std::max(tb, sort_temp_bytes(std::min(m_shadow ? m_shadow : m_bounce, m_bounce ? m_bounce : m_shadow),
                                                      30u - ctx->sort_bits, 32u));
                    rc = w.sort_temp.ensure(tb);
                    if (rc != RT_OK) return rc;
                }
                if (sort_shadow) {
                    HIP_TRY(sort_indices_by_key(w.sort_temp.p, w.sort_temp.bytes, (const unsigned *)w.keys[1].p,
                                                (unsigned *)w.keys_sorted.p, (unsigned *)w.perm[1].p, m_shadow,
                                                30u - ctx->sort_bits, 32u, s));
                    perm_connect = (const unsigned *)w.perm[1].p;
                    perm_connect_n = m_shadow;
                }
                perm_extend = nullptr;
                perm_extend_n = 0;
                if (sort_bounce) {
                    HIP_TRY(sort_indices_by_key(w.sort_temp.p, w.sort_temp.bytes, (const unsigned *)w.keys[0].p,
                                                (unsigned *)w.keys_sorted.p, (unsigned *)w.perm[0].p, m_bounce,
                                                30u - ctx->sort_bits, 32u, s));
                    perm_extend = (const unsigned *)w.perm[0].p;
                    perm_extend_n = m_bounce;
                }
                rc = timed_end(ctx, slot, s);
                if (rc != RT_OK) return rc;

                if (overlap) {
                    HIP_TRY(hipEventRecord(ctx->ev_shade[b], s));
                    HIP_TRY(hipStreamWaitEvent(sconn, ctx->ev_shade[b], 0));
                }
                rc = timed_begin(ctx, 1, sconn, &slot);
                if (rc != RT_OK) return rc;
                TraceLaunch tlc{fetch + (size_t)(2 * b + 1) * kFetchUints, ovf_conn, dstats + 1, ctx->min_active[b == 0 ? 0 : 1][1],
                                (b != 0 && ctx->trav_deep_connect != 0xFFFFFFFFu) ? ctx->trav_deep_connect
                                                                                   : ctx->trav_mode[b == 0 ? 0 : 1],
                                perm_connect, perm_connect_n, nullptr, 0, 0, b};
                fill_coop(ctx, sc, tlc, overlap);
                launch_connect(visit, sc->dev, (const QRay *)w.shadow_rays.p, (const ShadowInfo *)w.shadow_infos.p,
                               n_sh(b), ps.num_slots, rad_r, rad_g, rad_b, tlc, ctx->max_blocks, sconn);
                rc = timed_end(ctx, slot, sconn);
                if (rc != RT_OK) return rc;
                if (want) launch_count_rays(n_sh(b), dstats + 1, b, !visit, sconn);
                if (overlap) HIP_TRY(hipEventRecord(ctx->ev_conn[b], sconn));
                cur ^= 1;
            }
            if (overlap) HIP_TRY(hipStreamWaitEvent(s, ctx->ev_conn[nb - 1], 0)); // the radiance slots are complete
            rc = timed_begin(ctx, 2, s, &slot);
            if (rc != RT_OK) return rc;
            launch_accumulate(ps, rad_r, rad_g, rad_b, d_accum, s);
            rc = timed_end(ctx, slot, s);
            if (rc != RT_OK) return rc;
            if (sort_any && !ctx->counts_pending && (2 * nb + 1) * kCounterStride <= kMaxQueueCounters) {
                HIP_TRY(hipMemcpyAsync(ctx->h_counts, counters, (size_t)(2 * nb + 1) * kCounterStride * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, s));
                HIP_TRY(hipEventRecord(ctx->counts_ready, s));
                ctx->counts_pending = true;
                memcpy(ctx->pending_key, plan_key, sizeof plan_key);
            }
            HIP_TRY(hipGetLastError());
        }
    }
    if (want) {
        StatsDev h[2]{};
        HIP_TRY(hipMemcpyAsync(h, dstats, sizeof h, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int q = 0; q < 2; q++) {
            ctx->last_split[q] = rt_trace_stats{};
            ctx->last_max_steps[q] = h[q].max_ray_steps;
            ctx->last_max_stack[q] = h[q].max_stack_entries;
            for (unsigned d = 0; d < kStatDepths; d++) {
                ctx->last_depth[q].rays[d] = h[q].depth_rays[d];
                ctx->last_depth[q].inner_rays[d] = h[q].depth_inner[d];
                ctx->last_depth[q].obj_intersect[d] = h[q].depth_tris[d];
            }
            if (visit) {
                stats_from_dev(h[q], &ctx->last_split[q]);
                stats_from_dev(h[q], stats);
            } else {
                ctx->last_split[q].total_rays = h[q].total_rays;
                stats->total_rays += h[q].total_rays;
            }
        }
    }
    return RT_OK;
}

extern "C" int rt_render(rt_scene *sc, const rt_camera *camera, const rt_light *light,
                         const rt_render_params *params, float *accum, rt_trace_stats *stats)
{
    if (!sc || !camera || !light || !params || !accum) return fail(RT_ERR_INVALID, "rt_render: NULL argument");
    int rc = check_params(params, "rt_render");
    if (rc != RT_OK) return rc;
    rt_ctx *ctx = sc->ctx;
    rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;
    size_t bytes = (size_t)params->width * params->height * 4 * sizeof(float);
    rc = ctx->io_a.ensure(bytes);
    if (rc != RT_OK) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, accum, bytes, hipMemcpyHostToDevice, s));
    rc = rt_render_device(sc, camera, light, params, (float *)ctx->io_a.p, stats, s);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(accum, ctx->io_a.p, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RT_OK;
}

// ------------------------------------------------------------------------
// one-node multi-GPU

extern "C" int rt_shard_plan_make(uint32_t rank, uint32_t world, int mode, uint32_t samples, uint32_t sample_offset,
                                  rt_shard_plan *out)
{
    if (!out) return fail(RT_ERR_INVALID, "rt_shard_plan_make: out is NULL");
    if (world < 1 || rank >= world) return fail(RT_ERR_INVALID, "rt_shard_plan_make: rank %u of world %u", rank, world);
    if (mode == RT_SHARD_SAMPLE) {
        const uint32_t base = samples / world, rem = samples % world;
        out->samples = base + (rank < rem ? 1u : 0u);
        out->sample_offset = sample_offset + rank * base + std::min(rank, rem);
        out->shard_index = 0;
        out->shard_count = 1;
    } else if (mode == RT_SHARD_TILE) {
        out->samples = samples;
        out->sample_offset = sample_offset;
        out->shard_index = rank;
        out->shard_count = world;
    } else {
        return fail(RT_ERR_INVALID, "rt_shard_plan_make: mode %d is neither RT_SHARD_SAMPLE nor RT_SHARD_TILE", mode);
    }
    return RT_OK;
}

namespace {
// ncclReduce(sendbuff, recvbuff, count, datatype, op, root, comm, stream) of librccl.so, bound at first use
typedef int (*nccl_reduce_fn)(const void *, void *, size_t, int, int, int, void *, hipStream_t);
typedef const char *(*nccl_error_fn)(int);
nccl_reduce_fn g_nccl_reduce = nullptr;
nccl_error_fn g_nccl_error = nullptr;
std::once_flag g_nccl_once;
constexpr int kNcclFloat32 = 7, kNcclSum = 0; // ncclDataType_t / ncclRedOp_t values of rccl.h
} // namespace

extern "C" int rt_node_render(rt_scene *sc, const rt_camera *camera, const rt_light *light,
                              const rt_render_params *frame, uint32_t rank, uint32_t world, int mode, void *nccl_comm,
                              float *d_accum, rt_trace_stats *stats, void *hip_stream)
{
    if (!sc || !camera || !light || !frame || !d_accum) return fail(RT_ERR_INVALID, "rt_node_render: NULL argument");
    rt_shard_plan plan{};
    int rc = rt_shard_plan_make(rank, world, mode, frame->samples, frame->sample_offset, &plan);
    if (rc != RT_OK) return rc;
    rt_render_params p = *frame;
    p.samples = plan.samples;
    p.sample_offset = plan.sample_offset;
    p.shard_index = plan.shard_index;
    p.shard_count = plan.shard_count;
    if (p.samples > 0) {
        rc = rt_render_device(sc, camera, light, &p, d_accum, stats, hip_stream);
        if (rc != RT_OK) return rc;
    }
    if (!nccl_comm) return RT_OK;
    std::call_once(g_nccl_once, [] {
        void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (lib) {
            g_nccl_reduce = (nccl_reduce_fn)dlsym(lib, "ncclReduce");
            g_nccl_error = (nccl_error_fn)dlsym(lib, "ncclGetErrorString");
        }
    });
    if (!g_nccl_reduce) return fail(RT_ERR_HIP, "rt_node_render: librccl.so (ncclReduce) is not available: %s", dlerror());
    rc = ctx_set_device(sc->ctx);
    if (rc != RT_OK) return rc;
    hipStream_t s = (hipStream_t)hip_stream; // the stream the render was enqueued on
    const size_t count = (size_t)frame->width * frame->height * 4;
    const int nr = g_nccl_reduce(d_accum, d_accum, count, kNcclFloat32, kNcclSum, 0, nccl_comm, s);
    if (nr != 0)
        return fail(RT_ERR_HIP, "rt_node_render: ncclReduce failed: %s", g_nccl_error ? g_nccl_error(nr) : "unknown");
    return RT_OK;
}

extern "C" int rt_generate_primary(rt_ctx *ctx, const rt_camera *camera, const rt_render_params *params,
                                   rt_stream_ray *rays)
{
    if (!ctx || !camera || !params || !rays) return fail(RT_ERR_INVALID, "rt_generate_primary: NULL argument");
    int rc = check_params(params, "rt_generate_primary");
    if (rc != RT_OK) return rc;
    rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;
    unsigned long long total = (unsigned long long)params->width * params->height * params->samples;
    if (total == 0) return RT_OK;
    if (total > (1ull << 31)) return fail(RT_ERR_LIMIT, "rt_generate_primary: %llu rays exceed 2^31", total);
    rc = ctx->io_a.ensure(total * sizeof(rt_stream_ray));
    if (rc != RT_OK) return rc;
    PassDev ps{};
    ps.width = params->width;
    ps.height = params->height;
    ps.tile_size = params->tile_size;
    ps.samples_this_pass = params->samples;
    ps.sample_base = params->sample_offset;
    ps.seed_key = seed_key_of(params->seed);
    CameraDev cam = make_camera(*camera, params->width, params->height);
    hipStream_t s = ctx->stream;
    launch_generate_stream(cam, ps, (rt_stream_ray_dev *)ctx->io_a.p, total, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rays, ctx->io_a.p, total * sizeof(rt_stream_ray), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RT_OK;
}

extern "C" int rt_resolve(rt_ctx *ctx, const float *accum, uint32_t width, uint32_t height, uint8_t *rgb)
{
    if (!ctx || !accum || !rgb) return fail(RT_ERR_INVALID, "rt_resolve: NULL argument");
    int rc = ctx_set_device(ctx);
    if (rc != RT_OK) return rc;
    unsigned long long npix = (unsigned long long)width * height;
    if (npix == 0) return RT_OK;
    rc = ctx->io_a.ensure(npix * 4 * sizeof(float));
    if (rc == RT_OK) rc = ctx->io_b.ensure(npix * 3);
    if (rc != RT_OK) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->io_a.p, accum, npix * 4 * sizeof(float), hipMemcpyHostToDevice, s));
    launch_resolve((const float *)ctx->io_a.p, npix, (unsigned char *)ctx->io_b.p, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgb, ctx->io_b.p, npix * 3, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RT_OK;
}
