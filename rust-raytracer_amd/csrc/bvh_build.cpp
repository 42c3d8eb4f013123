// Host-side binned-SAH BVH builder (see bvh_build.hpp).
//
// Follows src/bvh/build.rs:65-207 decision for decision so the resulting tree
// (and therefore every hit id, tie and visit counter) is the reference's:
//   - leaf iff 1..8 triangles, or zero centroid extent         (80-81, 93-101)
//   - axis: x if sx>sy && sx>sz, else y if sy>sz, else z        (103-109)
//   - bucket = (k1*(c-k0)) as usize, k1 = 8*(1-0.01)/extent     (114-118)
//   - 7 candidate splits, cost=(SA_l*n_l+SA_r*n_r)/SA, strict < (132-150)
//   - Hoare partition on bucket <= min_bucket                   (152-168)
//   - child boxes are the unions of the bucket boxes            (147-148)
// All arithmetic is f32 and un-fused (this file is built with
// -ffp-contract=off).  The recursion of the reference is replaced by an explicit
// work stack that emits nodes in pre-order; large inputs are cut into
// independent subtrees that are built on several threads and spliced.
#include "bvh_build.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <thread>

namespace rtc {
namespace {

constexpr int kBuckets = 8;     // the reference's builder
constexpr int kMaxBuckets = 64; // the search builder
constexpr uint32_t kLeaf = 0x80000000u;
constexpr uint32_t kTaskTag = 0x40000000u; // placeholder ref: deferred subtree (top phase only)

struct Box {
    float lo[3], hi[3];
    void clear()
    {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::numeric_limits<float>::infinity();
            hi[a] = -std::numeric_limits<float>::infinity();
        }
    }
    void add_point(const float *p)
    {
        for (int a = 0; a < 3; a++) {
            lo[a] = lo[a] < p[a] ? lo[a] : p[a];
            hi[a] = hi[a] > p[a] ? hi[a] : p[a];
        }
    }
    void add_box(const Box &b)
    {
        for (int a = 0; a < 3; a++) {
            lo[a] = lo[a] < b.lo[a] ? lo[a] : b.lo[a];
            hi[a] = hi[a] > b.hi[a] ? hi[a] : b.hi[a];
        }
    }
    float area() const // src/aabb.rs:37-40
    {
        float sx = hi[0] - lo[0], sy = hi[1] - lo[1], sz = hi[2] - lo[2];
        return 2.f * ((sx * sy + sx * sz) + sy * sz);
    }
};

struct Prim {
    float c[3]; // centroid, src/mesh.rs:107-110
    Box box;    // src/mesh.rs:97-104
};

inline uint32_t bucket_of(float k0, float k1, float c)
{
    float f = k1 * (c - k0);
    if (!(f > 0.f)) return 0; // Rust `as usize`: negative and NaN -> 0
    if (f >= 4294967296.f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// Worker threads that stay alive for the top phase of one build: a node of the top levels makes up to six
// parallel passes over its range, and a thread started for each slice of each pass costs more than the pass.
class SlicePool {
  public:
    explicit SlicePool(unsigned workers)
    {
        for (unsigned t = 0; t < workers; t++) threads_.emplace_back([this, t] { loop(t + 1); });
    }
    ~SlicePool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            generation_++;
        }
        start_.notify_all();
        for (auto &t : threads_) t.join();
    }
    unsigned workers() const { return (unsigned)threads_.size(); }
    // job(t) for t = 1..workers on the pool's threads and job(0) on the caller's; returns when all are done
    void run(const std::function<void(unsigned)> &job)
    {
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &job;
            pending_ = (unsigned)threads_.size();
            generation_++;
        }
        start_.notify_all();
        job(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop(unsigned t)
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *job;
            {
                std::unique_lock<std::mutex> g(m_);
                start_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                job = job_;
            }
            (*job)(t);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable start_, done_;
    const std::function<void(unsigned)> *job_ = nullptr;
    unsigned pending_ = 0;
    unsigned long generation_ = 0;
    bool stop_ = false;
};
// the pool of the build running on this thread, if it has one (builds of two trees run side by side)
thread_local SlicePool *t_pool = nullptr;

// Runs fn(t, lo, hi) on `threads` threads over [begin, end) cut into equal slices.
template <class F>
void parallel_slices(unsigned threads, uint32_t begin, uint32_t end, F fn)
{
    const uint32_t n = end - begin;
    auto slice = [&](unsigned t) {
        fn(t, begin + (uint32_t)((uint64_t)n * t / threads), begin + (uint32_t)((uint64_t)n * (t + 1) / threads));
    };
    if (t_pool && t_pool->workers() + 1 == threads) {
        t_pool->run(slice);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) pool.emplace_back([&slice, t] { slice(t); });
    slice(0u);
    for (auto &th : pool) th.join();
}

// ranges of at least this many triangles are scanned by all threads of the top phase (the unions and
// counts are exact whatever the order, so the tree does not depend on it)
constexpr uint32_t kParallelScan = 1u << 15;

// The reference's in-place partition (src/bvh/build.rs:152-168: scan up to the first element that belongs
// right, scan down to the first that belongs left, swap, repeat) on several threads, with the same result
// element for element.  That loop ends with the m elements that belong left in front, and what it swaps is
// fixed by the input alone: the i-th misplaced element of [0, m) counted from the front with the i-th
// misplaced element of [m, n) counted from the back; everything else stays where it is.  So: flag every
// element (one gather of its centroid each), count, list the misplaced positions of both sides by slice
// with prefix offsets, swap the pairs.  Returns m.
template <class Pred>
uint32_t parallel_hoare_partition(unsigned threads, uint32_t *idx, uint32_t n, Pred goes_left)
{
    std::vector<unsigned char> flag(n);
    std::vector<uint32_t> count(threads + 1, 0);
    parallel_slices(threads, 0, n, [&](unsigned t, uint32_t lo, uint32_t hi) {
        uint32_t c = 0;
        for (uint32_t i = lo; i < hi; i++) {
            const unsigned char f = goes_left(idx[i]) ? 1 : 0;
            flag[i] = f;
            c += f;
        }
        count[t + 1] = c;
    });
    for (unsigned t = 0; t < threads; t++) count[t + 1] += count[t];
    const uint32_t m = count[threads];
    if (m == 0 || m == n) return m;
    // misplaced positions: in [0, m) the elements that belong right, in [m, n) those that belong left
    std::vector<uint32_t> wrong_front(threads + 1, 0), wrong_back(threads + 1, 0);
    parallel_slices(threads, 0, n, [&](unsigned t, uint32_t lo, uint32_t hi) {
        uint32_t a = 0, b = 0;
        for (uint32_t i = lo; i < hi; i++) {
            a += (i < m && !flag[i]);
            b += (i >= m && flag[i]);
        }
        wrong_front[t + 1] = a;
        wrong_back[t + 1] = b;
    });
    for (unsigned t = 0; t < threads; t++) {
        wrong_front[t + 1] += wrong_front[t];
        wrong_back[t + 1] += wrong_back[t];
    }
    const uint32_t k = wrong_front[threads]; // == wrong_back[threads]
    std::vector<uint32_t> front(k), back(k);
    parallel_slices(threads, 0, n, [&](unsigned t, uint32_t lo, uint32_t hi) {
        uint32_t a = wrong_front[t], b = wrong_back[t];
        for (uint32_t i = lo; i < hi; i++) {
            if (i < m) {
                if (!flag[i]) front[a++] = i;
            } else if (flag[i]) {
                back[b++] = i;
            }
        }
    });
    parallel_slices(threads, 0, k, [&](unsigned, uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; i++) std::swap(idx[front[i]], idx[back[k - 1 - i]]);
    });
    return m;
}

struct Work {
    uint32_t begin, end; // range in the order array
    uint32_t depth;
    uint32_t parent;     // node index in `nodes` receiving the child ref, or ~0u for the root
    uint32_t side;       // 0 = left, 1 = right
};

struct Deferred {
    uint32_t begin, end, depth;
};

struct Emit {
    std::vector<rt_flat_node> nodes;
    rt_bvh_stats stats{};
    uint32_t root_ref = 0, root_count = 0;
    // top phase only: pre-order sequence of events; value = node index, or
    // kTaskTag | task index.
    std::vector<uint32_t> events;
    // Set where the reference would panic (a bucket index beyond its bucket array,
    // src/bvh/build.rs:118-120) or a partition leaves one side empty.  The builder carries on
    // with a clamped bucket or a leaf, so that worker threads never throw, and build_bvh reports it.
    bool failed = false;
};

inline void set_child(Emit &e, const Work &w, uint32_t ref, uint32_t count)
{
    if (w.parent == ~0u) {
        e.root_ref = ref;
        e.root_count = count;
    } else if (w.side == 0) {
        e.nodes[w.parent].left = ref;
        e.nodes[w.parent].left_count = count;
    } else {
        e.nodes[w.parent].right = ref;
        e.nodes[w.parent].right_count = count;
    }
}

// Builds the subtree over order[begin,end).  With `defer` non-null, ranges of
// at most `grain` triangles (but more than a leaf) are not expanded; they are
// recorded as tasks.
template <bool SEARCH>
void build_range(const Prim *prims, uint32_t *order, uint32_t begin, uint32_t end, uint32_t depth,
                 Emit &e, std::vector<Deferred> *defer, uint32_t grain, bool track_events, BuildTuning tune,
                 unsigned scan_threads)
{
    const uint32_t max_leaf = SEARCH ? tune.max_leaf : 8u;
    const int nb = SEARCH ? (int)tune.buckets : kBuckets;
    std::vector<Work> stack;
    stack.push_back({begin, end, depth, ~0u, 0});
    while (!stack.empty()) {
        Work w = stack.back();
        stack.pop_back();
        uint32_t n = w.end - w.begin;

        bool leaf = (n >= 1 && n <= max_leaf);
        Box bounds, cb;
        float size[3] = {0, 0, 0};
        if (!leaf) {
            if (defer && n <= grain) {
                uint32_t t = (uint32_t)defer->size();
                defer->push_back({w.begin, w.end, w.depth});
                set_child(e, w, kTaskTag | t, 0);
                if (track_events) e.events.push_back(kTaskTag | t);
                continue;
            }
            bounds.clear();
            cb.clear();
            if (scan_threads > 1 && n >= kParallelScan) {
                std::vector<Box> part(2 * (size_t)scan_threads);
                parallel_slices(scan_threads, w.begin, w.end, [&](unsigned t, uint32_t lo, uint32_t hi) {
                    Box b, c;
                    b.clear();
                    c.clear();
                    for (uint32_t i = lo; i < hi; i++) {
                        const Prim &p = prims[order[i]];
                        b.add_box(p.box);
                        c.add_point(p.c);
                    }
                    part[2 * t] = b;
                    part[2 * t + 1] = c;
                });
                for (unsigned t = 0; t < scan_threads; t++) {
                    bounds.add_box(part[2 * t]);
                    cb.add_box(part[2 * t + 1]);
                }
            } else {
                for (uint32_t i = w.begin; i < w.end; i++) {
                    const Prim &p = prims[order[i]];
                    bounds.add_box(p.box);
                    cb.add_point(p.c);
                }
            }
            for (int a = 0; a < 3; a++) size[a] = cb.hi[a] - cb.lo[a];
            float nsq = (size[0] * size[0] + size[1] * size[1]) + size[2] * size[2];
            if (nsq == 0.f) leaf = true;
        }
        if (leaf) {
            set_child(e, w, kLeaf | w.begin, n);
            e.stats.num_leaves += 1;
            e.stats.num_objs += n;
            e.stats.max_depth = std::max(e.stats.max_depth, w.depth);
            e.stats.max_objs = std::max(e.stats.max_objs, n);
            continue;
        }

        int axis = (size[0] > size[1] && size[0] > size[2]) ? 0 : (size[1] > size[2] ? 1 : 2);
        float k0 = cb.lo[axis];
        float k1 = (float)nb * (1.f - 0.01f) / (cb.hi[axis] - cb.lo[axis]);
        uint32_t min_bucket = 0;
        Box bl, br;
        bl.clear();
        br.clear();
        {
            float min_cost = std::numeric_limits<float>::infinity();
            const float parent_area = bounds.area();
            // the reference looks along the axis of largest centroid extent only; the search
            // builder tries every axis that has one
            for (int ax = 0; ax < 3; ax++) {
                if (SEARCH ? !(size[ax] > 0.f) : ax != axis) continue;
                const float a0 = cb.lo[ax];
                const float a1 = (float)nb * (1.f - 0.01f) / (cb.hi[ax] - cb.lo[ax]);
                Box bbox[kMaxBuckets];
                uint32_t bcnt[kMaxBuckets];
                for (int b = 0; b < nb; b++) {
                    bbox[b].clear();
                    bcnt[b] = 0;
                }
                if (scan_threads > 1 && n >= kParallelScan) {
                    struct Bins {
                        Box box[kMaxBuckets];
                        uint32_t cnt[kMaxBuckets];
                        bool failed = false;
                    };
                    std::vector<Bins> part(scan_threads);
                    parallel_slices(scan_threads, w.begin, w.end, [&](unsigned t, uint32_t lo, uint32_t hi) {
                        Bins &m = part[t];
                        for (int b = 0; b < nb; b++) {
                            m.box[b].clear();
                            m.cnt[b] = 0;
                        }
                        for (uint32_t i = lo; i < hi; i++) {
                            const Prim &p = prims[order[i]];
                            uint32_t b = bucket_of(a0, a1, p.c[ax]);
                            if (b >= (uint32_t)nb) {
                                b = (uint32_t)nb - 1u;
                                if (!SEARCH) m.failed = true;
                            }
                            m.box[b].add_box(p.box);
                            m.cnt[b] += 1;
                        }
                    });
                    for (unsigned t = 0; t < scan_threads; t++) {
                        if (part[t].failed) e.failed = true;
                        for (int b = 0; b < nb; b++) {
                            bbox[b].add_box(part[t].box[b]);
                            bcnt[b] += part[t].cnt[b];
                        }
                    }
                } else {
                    for (uint32_t i = w.begin; i < w.end; i++) {
                        const Prim &p = prims[order[i]];
                        uint32_t b = bucket_of(a0, a1, p.c[ax]);
                        if (b >= (uint32_t)nb) {
                            b = (uint32_t)nb - 1u;
                            if (!SEARCH) e.failed = true;
                        }
                        bbox[b].add_box(p.box);
                        bcnt[b] += 1;
                    }
                }
                // prefix / suffix unions (min/max are exact, so the fold order of the
                // reference's per-candidate folds does not matter)
                Box pre[kMaxBuckets], suf[kMaxBuckets];
                uint32_t pcnt[kMaxBuckets], scnt[kMaxBuckets];
                pre[0] = bbox[0];
                pcnt[0] = bcnt[0];
                for (int b = 1; b < nb; b++) {
                    pre[b] = pre[b - 1];
                    pre[b].add_box(bbox[b]);
                    pcnt[b] = pcnt[b - 1] + bcnt[b];
                }
                suf[nb - 1] = bbox[nb - 1];
                scnt[nb - 1] = bcnt[nb - 1];
                for (int b = nb - 2; b >= 0; b--) {
                    suf[b] = suf[b + 1];
                    suf[b].add_box(bbox[b]);
                    scnt[b] = scnt[b + 1] + bcnt[b];
                }
                for (int i = 0; i < nb - 1; i++) {
                    if (SEARCH && (pcnt[i] == 0 || scnt[i + 1] == 0)) continue;
                    float cost =
                        (pre[i].area() * (float)pcnt[i] + suf[i + 1].area() * (float)scnt[i + 1]) / parent_area;
                    if (cost < min_cost) {
                        min_bucket = (uint32_t)i;
                        min_cost = cost;
                        bl = pre[i];
                        br = suf[i + 1];
                        axis = ax;
                        k0 = a0;
                        k1 = a1;
                    }
                }
            }
        }

        uint32_t *idx = order + w.begin;
        uint32_t left = 0, right = n - 1;
        // The reference's loops (src/bvh/build.rs:152-168) stop on a checked index; here the bounds
        // are explicit, and a partition with an empty side (impossible for finite centroids: the
        // smallest one is in bucket 0, the largest in the last) ends the subtree with a leaf.
        if (scan_threads > 1 && n >= kParallelScan) {
            left = parallel_hoare_partition(scan_threads, idx, n, [&](uint32_t prim) {
                return bucket_of(k0, k1, prims[prim].c[axis]) <= min_bucket;
            });
        } else {
            for (;;) {
                while (left < n && bucket_of(k0, k1, prims[idx[left]].c[axis]) <= min_bucket) left++;
                while (right > 0 && bucket_of(k0, k1, prims[idx[right]].c[axis]) > min_bucket) right--;
                if (left >= right) break;
                std::swap(idx[left], idx[right]);
            }
        }
        uint32_t split = left;
        if (split == 0 || split >= n) {
            e.failed = true;
            set_child(e, w, kLeaf | w.begin, n);
            e.stats.num_leaves += 1;
            e.stats.num_objs += n;
            e.stats.max_depth = std::max(e.stats.max_depth, w.depth);
            e.stats.max_objs = std::max(e.stats.max_objs, n);
            continue;
        }

        uint32_t me = (uint32_t)e.nodes.size();
        rt_flat_node nd{};
        for (int a = 0; a < 3; a++) {
            float *dst = a == 0 ? nd.bx : (a == 1 ? nd.by : nd.bz);
            dst[0] = bl.lo[a];
            dst[1] = br.lo[a];
            dst[2] = bl.hi[a];
            dst[3] = br.hi[a];
        }
        e.nodes.push_back(nd);
        if (track_events) e.events.push_back(me);
        set_child(e, w, me, 0);
        // right first so that the left subtree is emitted next (pre-order)
        stack.push_back({w.begin + split, w.end, w.depth + 1, me, 1});
        stack.push_back({w.begin, w.begin + split, w.depth + 1, me, 0});
    }
}

void run_range(bool search, const Prim *prims, uint32_t *order, uint32_t begin, uint32_t end, uint32_t depth,
               Emit &e, std::vector<Deferred> *defer, uint32_t grain, bool track_events, BuildTuning tune,
               unsigned scan_threads = 1)
{
    if (search)
        build_range<true>(prims, order, begin, end, depth, e, defer, grain, track_events, tune, scan_threads);
    else
        build_range<false>(prims, order, begin, end, depth, e, defer, grain, track_events, tune, scan_threads);
}

const char *const kDegenerate =
    "centroid range the binning cannot resolve (bucket index out of range; the reference panics here)";

void merge_stats(rt_bvh_stats &a, const rt_bvh_stats &b) // src/bvh/build.rs:35-42
{
    a.num_leaves += b.num_leaves;
    a.num_objs += b.num_objs;
    a.max_depth = std::max(a.max_depth, b.max_depth);
    a.max_objs = std::max(a.max_objs, b.max_objs);
}

} // namespace

void build_bvh(const float *positions, uint32_t nv, const uint32_t *indices, uint32_t nt,
               unsigned threads, HostBvh &out, const BuildTuning *tuning)
{
    (void)nv;
    const bool search = tuning != nullptr;
    BuildTuning tune = search ? *tuning : BuildTuning{8, 8};
    tune.buckets = std::min<uint32_t>(std::max<uint32_t>(tune.buckets, 2), kMaxBuckets);
    tune.max_leaf = std::min<uint32_t>(std::max<uint32_t>(tune.max_leaf, 1), 8);
    std::vector<Prim> prims(nt);
    out.order.resize(nt);
    auto fill = [&](uint32_t t0, uint32_t t1) {
        for (uint32_t t = t0; t < t1; t++) {
            const float *a = positions + 3 * (size_t)indices[3 * (size_t)t + 0];
            const float *b = positions + 3 * (size_t)indices[3 * (size_t)t + 1];
            const float *c = positions + 3 * (size_t)indices[3 * (size_t)t + 2];
            Prim &p = prims[t];
            for (int k = 0; k < 3; k++) p.c[k] = ((a[k] + b[k]) + c[k]) / 3.f;
            p.box.clear();
            p.box.add_point(a);
            p.box.add_point(b);
            p.box.add_point(c);
            out.order[t] = t;
        }
    };
    if (threads < 1) threads = 1;
    const bool verbose = getenv("RTCORE_VERBOSE") != nullptr && nt >= 200000;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  host build (%s): %-22s %.3f s\n", search ? "search" : "reference", what,
                std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    if (threads == 1 || nt < 200000) {
        fill(0, nt);
        Emit e;
        run_range(search, prims.data(), out.order.data(), 0, nt, 0, e, nullptr, 0, false, tune);
        if (e.failed) throw std::invalid_argument(kDegenerate);
        out.nodes.swap(e.nodes);
        out.root_ref = e.root_ref;
        out.root_count = e.root_count;
        out.stats = e.stats;
        return;
    }

    {
        std::vector<std::thread> pool;
        uint32_t per = (nt + threads - 1) / threads;
        for (unsigned i = 0; i < threads; i++) {
            uint32_t t0 = std::min(nt, i * per), t1 = std::min(nt, (i + 1) * per);
            pool.emplace_back(fill, t0, t1);
        }
        for (auto &t : pool) t.join();
    }

    mark("centroids and boxes");
    // top phase: expand until ranges are small, remember the rest as tasks
    Emit top;
    std::vector<Deferred> tasks;
    uint32_t grain = std::max<uint32_t>(4096, nt / (threads * 8));
    {
        SlicePool pool(threads - 1);
        t_pool = &pool;
        try {
            run_range(search, prims.data(), out.order.data(), 0, nt, 0, top, &tasks, grain, true, tune, threads);
        } catch (...) {
            t_pool = nullptr;
            throw;
        }
        t_pool = nullptr;
    }

    mark("top levels");
    // parallel phase: every task builds its subtree with local node indices
    std::vector<Emit> sub(tasks.size());
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= tasks.size()) break;
            run_range(search, prims.data(), out.order.data(), tasks[i].begin, tasks[i].end, tasks[i].depth,
                      sub[i], nullptr, 0, false, tune);
        }
    };
    {
        std::vector<std::thread> pool;
        for (unsigned i = 0; i < threads; i++) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    }

    mark("subtrees");
    bool failed = top.failed;
    for (const Emit &se : sub) failed = failed || se.failed;
    if (failed) throw std::invalid_argument(kDegenerate);

    // splice in pre-order
    std::vector<uint32_t> top_final(top.nodes.size()), task_base(tasks.size());
    uint32_t counter = 0;
    for (uint32_t ev : top.events) {
        if (ev & kTaskTag) {
            uint32_t t = ev & ~kTaskTag;
            task_base[t] = counter;
            counter += (uint32_t)sub[t].nodes.size();
        } else {
            top_final[ev] = counter++;
        }
    }
    out.nodes.assign(counter, rt_flat_node{});
    auto map_top_ref = [&](uint32_t ref, uint32_t &count) -> uint32_t {
        if (ref & kLeaf) return ref;
        if (ref & kTaskTag) {
            uint32_t t = ref & ~kTaskTag;
            if (sub[t].root_ref & kLeaf) { // cannot happen (tasks hold > 8 triangles) unless degenerate
                count = sub[t].root_count;
                return sub[t].root_ref;
            }
            return task_base[t] + sub[t].root_ref;
        }
        return top_final[ref];
    };
    for (size_t i = 0; i < top.nodes.size(); i++) {
        rt_flat_node nd = top.nodes[i];
        nd.left = map_top_ref(nd.left, nd.left_count);
        nd.right = map_top_ref(nd.right, nd.right_count);
        out.nodes[top_final[i]] = nd;
    }
    {
        std::atomic<size_t> next_task{0};
        auto copy_tasks = [&]() {
            for (;;) {
                const size_t t = next_task.fetch_add(1);
                if (t >= tasks.size()) break;
                const uint32_t base = task_base[t];
                for (size_t i = 0; i < sub[t].nodes.size(); i++) {
                    rt_flat_node nd = sub[t].nodes[i];
                    if (!(nd.left & kLeaf)) nd.left += base;
                    if (!(nd.right & kLeaf)) nd.right += base;
                    out.nodes[base + i] = nd;
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned i = 1; i < threads; i++) pool.emplace_back(copy_tasks);
        copy_tasks();
        for (auto &t : pool) t.join();
    }
    for (size_t t = 0; t < tasks.size(); t++) merge_stats(top.stats, sub[t].stats);
    out.stats = top.stats;
    uint32_t rc = top.root_count;
    out.root_ref = map_top_ref(top.root_ref, rc);
    out.root_count = rc;
    mark("splice");
}

} // namespace rtc
