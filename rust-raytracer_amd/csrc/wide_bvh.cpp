#include "wide_bvh.hpp"
#include "wide_node_ops.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace rtc {

namespace {

using Slot = WideSlot;

// child boxes of a binary node: lanes {minL, minR, maxL, maxR}
struct DevNodeTree {
    const DevNode *bn;
    void children(uint32_t bin, Slot &l, Slot &r) const
    {
        const DevNode &n = bn[bin];
        const float *b[3] = {n.bx, n.by, n.bz};
        for (int k = 0; k < 3; k++) {
            l.lo[k] = b[k][0];
            r.lo[k] = b[k][1];
            l.hi[k] = b[k][2];
            r.hi[k] = b[k][3];
        }
        l.ref = n.left;
        r.ref = n.right;
    }
};

struct Pending {
    uint32_t bin;    // binary node to collapse
    uint32_t parent; // wide node whose slot refers to it, or ~0u for the root
    uint32_t slot;
    uint32_t depth;
};

} // namespace

namespace {

unsigned worker_threads()
{
    const unsigned hw = std::thread::hardware_concurrency();
    return std::min(16u, std::max(1u, hw));
}

template <class F>
void run_parallel(unsigned n, F fn)
{
    // threads already started are joined whatever happens (a later one may fail to start)
    struct Pool {
        std::vector<std::thread> threads;
        ~Pool()
        {
            for (auto &th : threads)
                if (th.joinable()) th.join();
        }
    } pool;
    pool.threads.reserve(n);
    for (unsigned t = 1; t < n; t++) pool.threads.emplace_back([&fn, t] { fn(t); });
    fn(0);
}

// The wide node over binary node `bin` (collapse_slots() in wide_node_ops.hpp); returns how many slots are used.
int collapse_node(const DevNode *bn, uint32_t bin, DevWideNode &w, Slot (&c)[4])
{
    const int n = collapse_slots(DevNodeTree{bn}, bin, c);
    fill_wide_node(c, n, bin, w);
    return n;
}

// Collapses the subtree under `first` into `out` (indices local to `out`, the subtree's root at 0
// when out is empty); returns the deepest level reached (the first node's level being first.depth).
uint32_t collapse_subtree(const DevNode *bn, Pending first, std::vector<DevWideNode> &out)
{
    std::vector<Pending> todo{first};
    uint32_t deepest = 0;
    while (!todo.empty()) {
        const Pending job = todo.back();
        todo.pop_back();
        const uint32_t self = (uint32_t)out.size();
        out.emplace_back();
        if (job.parent != 0xFFFFFFFFu) out[job.parent].child[job.slot] = self;
        if (job.depth > deepest) deepest = job.depth;
        Slot c[4];
        const int n = collapse_node(bn, job.bin, out[self], c);
        // inner children become wide nodes; the first one is numbered next (pre-order)
        for (int i = n - 1; i >= 0; i--)
            if (!(c[i].ref & kLeafBit)) todo.push_back(Pending{c[i].ref, self, (uint32_t)i, job.depth + 1});
    }
    return deepest;
}

} // namespace

// The upper levels are collapsed breadth first on the calling thread until there are enough open
// subtrees for every worker; the subtrees are collapsed in parallel into vectors of their own and
// appended.  Wide-node indices therefore follow no particular order (layout_wide4 follows the
// child references from the root).
uint32_t collapse_wide4(const DevNode *bn, size_t num_nodes, uint32_t root_ref, std::vector<DevWideNode> &out,
                        uint32_t *max_wide_depth)
{
    if (max_wide_depth) *max_wide_depth = 0;
    if ((root_ref & kLeafBit) || num_nodes == 0) return root_ref;
    out.reserve(out.size() + num_nodes / 2 + 1);
    const uint32_t root_index = (uint32_t)out.size();
    const unsigned T = num_nodes < 65536 ? 1u : worker_threads();
    uint32_t deepest = 0;
    std::vector<Pending> open{Pending{root_ref, 0xFFFFFFFFu, 0, 1}};
    if (T > 1) {
        size_t head = 0;
        while (head < open.size() && open.size() - head < 16u * T) {
            const Pending job = open[head++];
            const uint32_t self = (uint32_t)out.size();
            out.emplace_back();
            if (job.parent != 0xFFFFFFFFu) out[job.parent].child[job.slot] = self;
            if (job.depth > deepest) deepest = job.depth;
            Slot c[4];
            const int n = collapse_node(bn, job.bin, out[self], c);
            for (int i = 0; i < n; i++)
                if (!(c[i].ref & kLeafBit)) open.push_back(Pending{c[i].ref, self, (uint32_t)i, job.depth + 1});
        }
        open.erase(open.begin(), open.begin() + (std::ptrdiff_t)head);
    }
    if (T == 1 || open.size() < 2) {
        for (const Pending &job : open) { // parents are in `out` already: indices are global
            const uint32_t self = (uint32_t)out.size();
            const Pending local{job.bin, 0xFFFFFFFFu, 0, job.depth};
            std::vector<DevWideNode> sub;
            deepest = std::max(deepest, collapse_subtree(bn, local, sub));
            for (DevWideNode w : sub) {
                for (int i = 0; i < 4; i++)
                    if (!(w.child[i] & kLeafBit)) w.child[i] += self;
                out.push_back(w);
            }
            if (job.parent != 0xFFFFFFFFu) out[job.parent].child[job.slot] = self;
        }
        if (max_wide_depth) *max_wide_depth = deepest;
        return root_index;
    }
    std::vector<std::vector<DevWideNode>> sub(open.size());
    std::vector<uint32_t> sub_deepest(open.size(), 0);
    std::atomic<size_t> next{0};
    run_parallel(T, [&](unsigned) {
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= open.size()) break;
            sub_deepest[j] = collapse_subtree(bn, Pending{open[j].bin, 0xFFFFFFFFu, 0, open[j].depth}, sub[j]);
        }
    });
    std::vector<size_t> base(open.size() + 1, out.size());
    for (size_t j = 0; j < open.size(); j++) base[j + 1] = base[j] + sub[j].size();
    out.resize(base[open.size()]);
    next = 0;
    run_parallel(T, [&](unsigned) {
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= open.size()) break;
            const uint32_t b = (uint32_t)base[j];
            for (size_t i = 0; i < sub[j].size(); i++) {
                DevWideNode w = sub[j][i];
                for (int k = 0; k < 4; k++)
                    if (!(w.child[k] & kLeafBit)) w.child[k] += b;
                out[b + i] = w;
            }
        }
    });
    for (size_t j = 0; j < open.size(); j++) {
        out[open[j].parent].child[open[j].slot] = (uint32_t)base[j];
        deepest = std::max(deepest, sub_deepest[j]);
    }
    if (max_wide_depth) *max_wide_depth = deepest;
    return root_index;
}

namespace {

struct Box3 {
    float lo[3], hi[3];
    void reset()
    {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::numeric_limits<float>::infinity();
            hi[k] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const Box3 &o)
    {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], o.lo[k]);
            hi[k] = std::max(hi[k], o.hi[k]);
        }
    }
    float area() const
    {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return 2.f * (x * y + x * z + y * z);
    }
};

struct Refiner {
    const float *positions;
    const uint32_t *indices;
    unsigned max_tris;
    std::vector<DevNode> *xn;
    std::vector<uint32_t> *xorder;
    const std::vector<BigLeaf> *big;
    uint32_t deepest = 0;

    Box3 tri_box(uint32_t t) const
    {
        Box3 b;
        b.reset();
        for (int v = 0; v < 3; v++) {
            const float *p = positions + 3 * (size_t)indices[3 * (size_t)t + v];
            for (int k = 0; k < 3; k++) {
                b.lo[k] = std::min(b.lo[k], p[k]);
                b.hi[k] = std::max(b.hi[k], p[k]);
            }
        }
        return b;
    }
    float centroid(uint32_t t, int k) const
    {
        const float *a = positions + 3 * (size_t)indices[3 * (size_t)t + 0];
        const float *b = positions + 3 * (size_t)indices[3 * (size_t)t + 1];
        const float *c = positions + 3 * (size_t)indices[3 * (size_t)t + 2];
        return (a[k] + b[k] + c[k]) / 3.f;
    }
    static uint32_t leaf_ref(uint32_t first, uint32_t count) { return kLeafBit | (first << 3) | (count - 1u); }

    // a node over slots [first, first + k) and [first + k, first + n) with the exact bounds of both
    uint32_t emit(uint32_t first, uint32_t k, uint32_t n, const Box3 &box_a, const Box3 &box_b, uint32_t depth)
    {
        const uint32_t self = (uint32_t)xn->size();
        xn->emplace_back();
        deepest = std::max(deepest, depth);
        const uint32_t l = split(first, k, depth + 1);
        const uint32_t r = split(first + k, n - k, depth + 1);
        DevNode &d = (*xn)[self];
        float *b[3] = {d.bx, d.by, d.bz};
        for (int a = 0; a < 3; a++) {
            b[a][0] = box_a.lo[a];
            b[a][1] = box_b.lo[a];
            b[a][2] = box_a.hi[a];
            b[a][3] = box_b.hi[a];
        }
        d.left = l;
        d.right = r;
        d.pad0 = d.pad1 = 0;
        return self;
    }
    // nothing to order the triangles by (or too many for the exhaustive cut): halves by slot
    uint32_t halve(uint32_t first, uint32_t n, uint32_t depth)
    {
        const uint32_t *slot = xorder->data() + first;
        const uint32_t k = n / 2;
        Box3 a, b;
        a.reset();
        b.reset();
        for (uint32_t i = 0; i < k; i++) a.grow(tri_box(slot[i]));
        for (uint32_t i = k; i < n; i++) b.grow(tri_box(slot[i]));
        return emit(first, k, n, a, b, depth);
    }

    // slots [first, first + n) of xorder; returns the reference that replaces the leaf
    uint32_t split(uint32_t first, uint32_t n, uint32_t depth)
    {
        if (n <= max_tris) return leaf_ref(first, n);
        if (n > 8) return halve(first, n, depth);
        uint32_t *slot = xorder->data() + first;
        float clo[3], chi[3];
        for (int k = 0; k < 3; k++) {
            clo[k] = std::numeric_limits<float>::infinity();
            chi[k] = -clo[k];
        }
        for (uint32_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                const float c = centroid(slot[i], k);
                clo[k] = std::min(clo[k], c);
                chi[k] = std::max(chi[k], c);
            }
        int axis = 0;
        for (int k = 1; k < 3; k++)
            if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
        if (!(chi[axis] - clo[axis] > 0.f)) return halve(first, n, depth);
        std::stable_sort(slot, slot + n, [&](uint32_t a, uint32_t b) {
            const float ca = centroid(a, axis), cb = centroid(b, axis);
            return ca < cb || (ca == cb && a < b);
        });
        Box3 suffix[9];
        suffix[n].reset();
        for (uint32_t i = n; i-- > 0;) {
            suffix[i] = suffix[i + 1];
            suffix[i].grow(tri_box(slot[i]));
        }
        Box3 prefix;
        prefix.reset();
        uint32_t best_k = 1;
        float best_cost = std::numeric_limits<float>::infinity();
        Box3 best_a = prefix;
        for (uint32_t k = 1; k < n; k++) {
            prefix.grow(tri_box(slot[k - 1]));
            const float cost = prefix.area() * (float)k + suffix[k].area() * (float)(n - k);
            if (cost < best_cost) {
                best_cost = cost;
                best_k = k;
                best_a = prefix;
            }
        }
        return emit(first, best_k, n, best_a, suffix[best_k], depth);
    }
    uint32_t refine_ref(uint32_t ref)
    {
        if (!(ref & kLeafBit)) return ref;
        if (ref & kBigLeafBit) {
            const BigLeaf &bl = (*big)[ref & 0x3FFFFFFFu];
            return split(bl.first, bl.count, 1);
        }
        const uint32_t first = (ref >> 3) & (kMaxPackedFirst - 1u), count = (ref & 7u) + 1u;
        return split(first, count, 1);
    }
};

} // namespace

uint32_t refine_leaves(const std::vector<DevNode> &bn, uint32_t root_ref, const float *positions,
                       const uint32_t *indices, const std::vector<uint32_t> &order, unsigned max_tris,
                       const std::vector<BigLeaf> &big, std::vector<DevNode> &xn, uint32_t &xroot,
                       std::vector<uint32_t> &xorder)
{
    xn = bn;
    xorder = order;
    xroot = root_ref;
    if (max_tris < 1 || max_tris >= 8) return 0;
    const size_t n = bn.size();
    // Leaves are independent (each reorders only its own slots of xorder).  Every thread takes a
    // contiguous range of the original nodes and collects the nodes it creates in a vector of its
    // own, numbered from 0; joining the vectors in range order gives the numbering a single
    // sequential pass would give.
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(worker_threads(), n / 4096 + 1));
    std::vector<std::vector<DevNode>> made(T);
    std::vector<uint32_t> deepest(T, 0);
    const uint32_t kLocal = 0x40000000u; // marks a reference to a thread-local new node (inner refs stay below)
    run_parallel(T, [&](unsigned t) {
        Refiner rf{positions, indices, max_tris, &made[t], &xorder, &big};
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        for (size_t i = lo; i < hi; i++) {
            uint32_t l = rf.refine_ref(bn[i].left);
            uint32_t r = rf.refine_ref(bn[i].right);
            if (l != bn[i].left && !(l & kLeafBit)) l |= kLocal;
            if (r != bn[i].right && !(r & kLeafBit)) r |= kLocal;
            xn[i].left = l;
            xn[i].right = r;
        }
        deepest[t] = rf.deepest;
    });
    std::vector<size_t> base(T + 1, n);
    for (unsigned t = 0; t < T; t++) base[t + 1] = base[t] + made[t].size();
    xn.resize(base[T]);
    run_parallel(T, [&](unsigned t) {
        const uint32_t b = (uint32_t)base[t];
        for (size_t j = 0; j < made[t].size(); j++) { // references between new nodes are local indices
            DevNode d = made[t][j];
            if (!(d.left & kLeafBit)) d.left += b;
            if (!(d.right & kLeafBit)) d.right += b;
            xn[b + j] = d;
        }
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        for (size_t i = lo; i < hi; i++) {
            if (!(xn[i].left & kLeafBit) && (xn[i].left & kLocal)) xn[i].left = (xn[i].left & ~kLocal) + b;
            if (!(xn[i].right & kLeafBit) && (xn[i].right & kLocal)) xn[i].right = (xn[i].right & ~kLocal) + b;
        }
    });
    uint32_t deep = 0;
    for (uint32_t d : deepest) deep = std::max(deep, d);
    // the root is a leaf only in a tree without inner nodes
    {
        std::vector<DevNode> extra;
        Refiner rf{positions, indices, max_tris, &extra, &xorder, &big};
        const uint32_t r = rf.refine_ref(root_ref);
        if (r != root_ref && !(r & kLeafBit)) {
            const uint32_t b = (uint32_t)xn.size();
            for (DevNode d : extra) {
                if (!(d.left & kLeafBit)) d.left += b;
                if (!(d.right & kLeafBit)) d.right += b;
                xn.push_back(d);
            }
            xroot = r + b;
        } else {
            xroot = r;
        }
        deep = std::max(deep, rf.deepest);
    }
    return deep;
}

namespace {
// children(w, refs) gives the used child references of wide node w and returns how many there are
template <class Children>
bool layout_wide4_of(size_t num_wide, Children children, uint32_t wroot, uint32_t num_triangles, WideLayout &out,
                     unsigned align_mode)
{
    // align_mode 0: items back to back; 1: nodes start on 64-byte lines; 2: every item does; 3: no item lies
    // across a 128-byte line (8 units), gaps only where one would; 4: as 3, and a leaf of one or two triangles
    // lies within one line
    auto crosses = [](uint32_t unit) { return (unit & 7u) > 8u - kUnitsPerItem; };
    auto place = [&](uint32_t &next, bool node) {
        if (align_mode == 2 || (align_mode == 1 && node)) next = (next + 3u) & ~3u;
        if (align_mode >= 3 && node)
            while (crosses(next)) next++;
    };
    auto place_leaf = [&](uint32_t &next, uint32_t count) {
        if (align_mode < 3) return;
        for (;; next++) {
            bool ok = true;
            for (uint32_t t = 0; t < count; t++) ok = ok && !crosses(next + kUnitsPerItem * t);
            if (align_mode == 4 && kUnitsPerItem * count <= 8u) ok = ok && (next & 7u) + kUnitsPerItem * count <= 8u;
            if (ok) return;
        }
    };
    out.node_unit.assign(num_wide, 0xFFFFFFFFu);
    out.slot_unit.assign(num_triangles, 0xFFFFFFFFu);
    out.num_units = 0;
    auto leaf_range = [](uint32_t ref, uint32_t &first, uint32_t &count) {
        first = (ref >> 3) & (kMaxPackedFirst - 1u);
        count = (ref & 7u) + 1u;
        return !(ref & kBigLeafBit) && count <= kWideLeafTris;
    };
    uint32_t next = 0;
    if (wroot & kLeafBit) { // a mesh of at most kWideLeafTris triangles: the root is its leaf
        uint32_t first, count;
        if (!leaf_range(wroot, first, count) || first + count > num_triangles) return false;
        for (uint32_t t = 0; t < count; t++) out.slot_unit[first + t] = kUnitsPerItem * t;
        out.num_units = kUnitsPerItem * count;
        out.root_ref = count;
        return true;
    }
    if (wroot >= num_wide) return false;
    out.node_unit[wroot] = 0;
    next = kUnitsPerItem;
    out.root_ref = 0;
    std::vector<uint32_t> todo{wroot};
    while (!todo.empty()) {
        const uint32_t w = todo.back();
        todo.pop_back();
        uint32_t refs[4];
        const int used = children(w, refs);
        uint32_t inner[4];
        int ninner = 0;
        uint32_t block_first = 0xFFFFFFFFu; // a child is addressed by six bits of units past the block's first item
        auto in_block = [&](uint32_t unit) {
            if (block_first == 0xFFFFFFFFu) block_first = unit;
            return unit - block_first < 64u;
        };
        for (int i = 0; i < used; i++) {
            const uint32_t ref = refs[i];
            if (ref & kLeafBit) {
                uint32_t first, count;
                if (!leaf_range(ref, first, count) || first + count > num_triangles) return false;
                if (align_mode == 2) {
                    return false; // leaves must stay contiguous: not supported with per-item alignment
                }
                place_leaf(next, count);
                if (!in_block(next)) return false;
                for (uint32_t t = 0; t < count; t++) out.slot_unit[first + t] = next + kUnitsPerItem * t;
                next += kUnitsPerItem * count;
            } else {
                if (ref >= num_wide || out.node_unit[ref] != 0xFFFFFFFFu) return false;
                place(next, true);
                if (!in_block(next)) return false;
                out.node_unit[ref] = next;
                next += kUnitsPerItem;
                inner[ninner++] = ref;
            }
        }
        for (int i = ninner - 1; i >= 0; i--) todo.push_back(inner[i]); // first child's block comes next
    }
    out.num_units = next;
    return true;
}
} // namespace

bool layout_wide4(const std::vector<DevWideNode> &wide, uint32_t wroot, uint32_t num_triangles, WideLayout &out,
                  unsigned align_mode)
{
    return layout_wide4_of(
        wide.size(),
        [&](uint32_t w, uint32_t (&refs)[4]) {
            const int used = wide_children_used(wide[w]);
            for (int i = 0; i < used; i++) refs[i] = wide[w].child[i];
            return used;
        },
        wroot, num_triangles, out, align_mode);
}

bool layout_wide4(const uint32_t *children, size_t num_wide, uint32_t wroot, uint32_t num_triangles, WideLayout &out,
                  unsigned align_mode)
{
    return layout_wide4_of(
        num_wide,
        [&](uint32_t w, uint32_t (&refs)[4]) {
            int used = 0;
            while (used < 4 && children[4 * (size_t)w + used] != kNoWideChild) {
                refs[used] = children[4 * (size_t)w + used];
                used++;
            }
            return used;
        },
        wroot, num_triangles, out, align_mode);
}

DevQNode quantize_node(const DevWideNode &src, const WideLayout &layout)
{
    DevQNode q{};
    const int n = wide_children_used(src);
    quantize_planes(src, n, q);
    // children block: in slot order, the way layout_wide4 numbered it
    quantize_children(
        n,
        [&](int i) {
            const uint32_t ref = src.child[i];
            return (ref & kLeafBit) ? layout.slot_unit[(ref >> 3) & (kMaxPackedFirst - 1u)] : layout.node_unit[ref];
        },
        [&](int i) { return (src.child[i] & kLeafBit) ? (src.child[i] & 7u) + 1u : 0u; }, q);
    return q;
}

void pack_wide4(const std::vector<DevWideNode> &wide, const WideLayout &layout, const DevTri *xtris,
                uint32_t num_triangles, std::vector<Unit> &items)
{
    items.resize((size_t)layout.num_units + 1);
    pack_wide4(wide, layout, xtris, num_triangles, items.data(), true);
}

void pack_wide4(const std::vector<DevWideNode> &wide, const WideLayout &layout, const DevTri *xtris,
                uint32_t num_triangles, Unit *items, bool gaps_cleared)
{
    static_assert(sizeof(DevQNode) == kUnitsPerItem * sizeof(Unit) && sizeof(DevTri) == kUnitsPerItem * sizeof(Unit),
                  "items are three units");
    const size_t nodes = wide.size(), units = (size_t)layout.num_units + 1;
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(worker_threads(), (nodes + num_triangles) / 65536 + 1));
    // units no item covers (alignment gaps, the spare unit at the end) read as zeros
    const bool dense = (size_t)kUnitsPerItem * (nodes + num_triangles) == (size_t)layout.num_units;
    if (!gaps_cleared) {
        if (dense)
            items[units - 1] = Unit{};
        else
            run_parallel(T, [&](unsigned t) {
                std::memset((void *)(items + units * t / T), 0, (units * (t + 1) / T - units * t / T) * sizeof(Unit));
            });
    }
    run_parallel(T, [&](unsigned t) {
        for (size_t w = nodes * t / T; w < nodes * (t + 1) / T; w++) {
            const DevQNode q = quantize_node(wide[w], layout);
            std::memcpy(&items[layout.node_unit[w]], &q, sizeof q);
        }
        for (size_t s = (size_t)num_triangles * t / T; s < (size_t)num_triangles * (t + 1) / T; s++)
            std::memcpy(&items[layout.slot_unit[s]], &xtris[s], sizeof(DevTri));
    });
}

} // namespace rtc
