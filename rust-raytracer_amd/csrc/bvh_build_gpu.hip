// Device-side build of the tree the fast traversal forms walk (see bvh_build_gpu.hpp).
//
// The search builder of bvh_build.cpp -- binned SAH in the manner of src/bvh/build.rs:65-207 with
// `buckets` buckets, every axis with a centroid extent tried, candidates with an empty side
// skipped, leaves of at most `max_leaf` triangles -- run breadth first on the GPU: one pass per
// tree level over all triangles instead of one recursion per node.  Every decision is the host
// builder's, in the host builder's f32 arithmetic (this file is compiled with -ffp-contract=off
// like the rest): bounds and bins are unions (min/max, exact in any order) and counts, the cost
// of a candidate is (area_l * n_l + area_r * n_r) / area_parent, the first minimum in (axis,
// bucket) order wins.  So the tree has the host tree's nodes, boxes and leaf ranges; what differs
// is the numbering of the nodes (breadth first here, pre-order there) and the order of the
// triangles inside a leaf (the partition here is stable, the host's is Hoare's).
//
// Per level, with A nodes still open:
//   bounds   over the triangles: box and centroid box of every open node (atomics on ordered keys;
//            a block whose whole chunk lies in one node reduces in LDS first)
//   setup    per node: leaf when all centroids coincide, else the bucket scale of every axis
//   bin      over the triangles: box and count of every (node, axis, bucket)
//   split    per node: prefix/suffix unions, the cheapest candidate, the node record
//   children per node: open children get their slot for the next level (scan), small ones become leaves
//   flag + scan + scatter over the triangles: stable partition of every node's range
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "bvh_build_gpu.hpp"
#include "wide_bvh.hpp"
#include "wide_node_ops.hpp"

namespace rtc {
namespace {

constexpr uint32_t kLeaf = 0x80000000u;
constexpr uint32_t kNoParent = 0xFFFFFFFFu;
constexpr int kMaxB = 64;          // most buckets (bvh_build.cpp: kMaxBuckets)
constexpr int kChunk = 2048;       // triangles per block of the passes over the triangles
constexpr int kThreads = 256;
constexpr int kItems = kChunk / kThreads;

// f32 <-> unsigned keys that order like the floats (for atomicMin / atomicMax)
__host__ __device__ inline uint32_t fkey(float f)
{
    uint32_t b;
    memcpy(&b, &f, 4);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ inline float funkey(uint32_t k)
{
    uint32_t b = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    float f;
    memcpy(&f, &b, 4);
    return f;
}
constexpr uint32_t kKeyPosInf = 0xFF800000u; // fkey(+inf): the empty lower bound
constexpr uint32_t kKeyNegInf = 0x007FFFFFu; // fkey(-inf): the empty upper bound

struct Prims { // triangles in their current order (structure of arrays)
    float *c[3], *lo[3], *hi[3];
    uint32_t *order;
};

struct Open { // a node still to be split: order[begin, end)
    uint32_t begin, end;
    uint32_t parent; // node index that receives the reference, kNoParent for the root
    uint32_t side;   // 0 = left, 1 = right
};

struct Info { // what a level works out about an open node
    uint32_t blo[3], bhi[3]; // box (keys)
    uint32_t clo[3], chi[3]; // centroid box (keys)
    uint32_t inner;          // 1: gets a node, 0: becomes a leaf (all centroids coincide)
    uint32_t node;           // index of its node record
    float a0[3], a1[3];      // bucket = a1 * (c - a0) per axis (a1 = 0: axis without extent)
    float parent_area;
    uint32_t axis, min_bucket, nl; // the split: triangles with bucket <= min_bucket on `axis` go left
    uint32_t child_slot[2];        // slot of the child among the next level's open nodes, or ~0u (leaf)
};

struct Bin {
    uint32_t lo[3], hi[3];
    uint32_t cnt, pad;
};

struct Totals {
    uint32_t num_nodes;   // node records written so far
    uint32_t next_open;   // open nodes of the next level
    uint32_t failed;      // a node had no valid candidate (the host builder reports the mesh as degenerate)
    uint32_t root_ref, root_count;
    uint32_t num_leaves, num_objs, max_depth, max_objs; // BvhStats, src/bvh/build.rs:5-11
};

__device__ inline uint32_t bucket_of(float k0, float k1, float c, uint32_t nb)
{
    const float f = k1 * (c - k0);
    uint32_t b;
    if (!(f > 0.f))
        b = 0; // Rust `as usize`: negative and NaN -> 0
    else if (f >= 4294967296.f)
        b = 0xFFFFFFFFu;
    else
        b = (uint32_t)f;
    return b >= nb ? nb - 1u : b; // the search builder clamps
}

// centroid (src/mesh.rs:107-110) and box (97-104) of every triangle, identity order
__global__ __launch_bounds__(256) void prims_kernel(const float *pos, const uint32_t *idx, uint32_t n, Prims p,
                                                    int *node_of)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const float *a = pos + 3 * (size_t)idx[3 * (size_t)t + 0];
    const float *b = pos + 3 * (size_t)idx[3 * (size_t)t + 1];
    const float *c = pos + 3 * (size_t)idx[3 * (size_t)t + 2];
    for (int k = 0; k < 3; k++) {
        p.c[k][t] = ((a[k] + b[k]) + c[k]) / 3.f;
        p.lo[k][t] = fminf(fminf(a[k], b[k]), c[k]);
        p.hi[k][t] = fmaxf(fmaxf(a[k], b[k]), c[k]);
    }
    p.order[t] = t;
    node_of[t] = 0;
}

__global__ __launch_bounds__(256) void info_init_kernel(Info *info, uint32_t count)
{
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= count) return;
    for (int k = 0; k < 3; k++) {
        info[a].blo[k] = info[a].clo[k] = kKeyPosInf;
        info[a].bhi[k] = info[a].chi[k] = kKeyNegInf;
    }
}

// box and centroid box of every open node
__global__ __launch_bounds__(kThreads) void bounds_kernel(Prims p, const int *node_of, uint32_t n, Info *info)
{
    __shared__ uint32_t red[12];
    const uint32_t base = blockIdx.x * (uint32_t)kChunk;
    const uint32_t last = min(base + (uint32_t)kChunk, n) - 1u;
    const int first_node = node_of[base], last_node = node_of[last];
    // ranges are contiguous: a chunk that starts and ends in one open node lies in it as a whole (a chunk
    // that starts and ends outside every open node may still hold some)
    const bool uniform = first_node == last_node && first_node >= 0;
    if (uniform) {
        if (threadIdx.x < 12) red[threadIdx.x] = (threadIdx.x % 6) < 3 ? kKeyPosInf : kKeyNegInf;
        __syncthreads();
        float v[12] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY,
                       INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int it = 0; it < kItems; it++) {
            const uint32_t i = base + it * kThreads + threadIdx.x;
            if (i > last) break;
            for (int k = 0; k < 3; k++) {
                v[k] = fminf(v[k], p.lo[k][i]);
                v[3 + k] = fmaxf(v[3 + k], p.hi[k][i]);
                const float c = p.c[k][i];
                v[6 + k] = fminf(v[6 + k], c);
                v[9 + k] = fmaxf(v[9 + k], c);
            }
        }
        for (int k = 0; k < 12; k++) {
            if ((k % 6) < 3)
                atomicMin(&red[k], fkey(v[k]));
            else
                atomicMax(&red[k], fkey(v[k]));
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            Info &f = info[first_node];
            uint32_t *dst = threadIdx.x < 3 ? &f.blo[threadIdx.x] : threadIdx.x < 6 ? &f.bhi[threadIdx.x - 3]
                          : threadIdx.x < 9 ? &f.clo[threadIdx.x - 6] : &f.chi[threadIdx.x - 9];
            if ((threadIdx.x % 6) < 3)
                atomicMin(dst, red[threadIdx.x]);
            else
                atomicMax(dst, red[threadIdx.x]);
        }
        return;
    }
    for (int it = 0; it < kItems; it++) {
        const uint32_t i = base + it * kThreads + threadIdx.x;
        if (i > last) break;
        const int a = node_of[i];
        if (a < 0) continue;
        Info &f = info[a];
        for (int k = 0; k < 3; k++) {
            atomicMin(&f.blo[k], fkey(p.lo[k][i]));
            atomicMax(&f.bhi[k], fkey(p.hi[k][i]));
            const uint32_t c = fkey(p.c[k][i]);
            atomicMin(&f.clo[k], c);
            atomicMax(&f.chi[k], c);
        }
    }
}

// leaf or inner, bucket scales (bvh_build.cpp: build_range, the part before the candidate loop)
__global__ __launch_bounds__(256) void setup_kernel(const Open *open, Info *info, uint32_t count, uint32_t nb,
                                                    uint32_t *inner_flag)
{
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= count) return;
    Info &f = info[a];
    float size[3], blo[3], bhi[3];
    for (int k = 0; k < 3; k++) {
        const float clo = funkey(f.clo[k]), chi = funkey(f.chi[k]);
        size[k] = chi - clo;
        blo[k] = funkey(f.blo[k]);
        bhi[k] = funkey(f.bhi[k]);
        f.a0[k] = clo;
        f.a1[k] = size[k] > 0.f ? (float)nb * (1.f - 0.01f) / (chi - clo) : 0.f;
    }
    const float nsq = (size[0] * size[0] + size[1] * size[1]) + size[2] * size[2];
    const float sx = bhi[0] - blo[0], sy = bhi[1] - blo[1], sz = bhi[2] - blo[2];
    f.parent_area = 2.f * ((sx * sy + sx * sz) + sy * sz);
    f.inner = nsq == 0.f ? 0u : 1u;
    f.child_slot[0] = f.child_slot[1] = 0xFFFFFFFFu;
    inner_flag[a] = f.inner;
    (void)open;
}

__global__ __launch_bounds__(256) void bins_init_kernel(Bin *bins, size_t count)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    Bin b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = kKeyPosInf;
        b.hi[k] = kKeyNegInf;
    }
    b.cnt = 0;
    b.pad = 0;
    bins[i] = b;
}

// node index of every inner open node; the reference to it (or to the leaf it became) goes to its parent
__global__ __launch_bounds__(256) void assign_kernel(const Open *open, Info *info, const uint32_t *inner_scan,
                                                     uint32_t count, uint32_t depth, rt_flat_node *nodes, Totals *tot)
{
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= count) return;
    Info &f = info[a];
    const Open o = open[a];
    uint32_t ref, cnt;
    if (f.inner) {
        f.node = tot->num_nodes + inner_scan[a];
        ref = f.node;
        cnt = 0;
    } else {
        ref = kLeaf | o.begin;
        cnt = o.end - o.begin;
        atomicAdd(&tot->num_leaves, 1u);
        atomicAdd(&tot->num_objs, cnt);
        atomicMax(&tot->max_depth, depth);
        atomicMax(&tot->max_objs, cnt);
    }
    if (o.parent == kNoParent) {
        tot->root_ref = ref;
        tot->root_count = cnt;
    } else if (o.side == 0) {
        nodes[o.parent].left = ref;
        nodes[o.parent].left_count = cnt;
    } else {
        nodes[o.parent].right = ref;
        nodes[o.parent].right_count = cnt;
    }
}

// box and count of every (node, axis, bucket)
__global__ __launch_bounds__(kThreads) void bin_kernel(Prims p, const int *node_of, uint32_t n, const Info *info,
                                                       Bin *bins, uint32_t nb)
{
    __shared__ uint32_t sb[3 * kMaxB * 7];
    const uint32_t base = blockIdx.x * (uint32_t)kChunk;
    const uint32_t last = min(base + (uint32_t)kChunk, n) - 1u;
    const int first_node = node_of[base], last_node = node_of[last];
    const bool uniform = first_node == last_node && first_node >= 0;
    if (uniform && !info[first_node].inner) return;
    if (uniform) {
        for (uint32_t w = threadIdx.x; w < 3u * nb * 7u; w += kThreads) {
            const uint32_t field = w % 7u;
            sb[w] = field < 3 ? kKeyPosInf : field < 6 ? kKeyNegInf : 0u;
        }
        __syncthreads();
        const Info &f = info[first_node];
        for (int it = 0; it < kItems; it++) {
            const uint32_t i = base + it * kThreads + threadIdx.x;
            if (i > last) break;
            uint32_t lo[3], hi[3];
            for (int k = 0; k < 3; k++) {
                lo[k] = fkey(p.lo[k][i]);
                hi[k] = fkey(p.hi[k][i]);
            }
            for (int ax = 0; ax < 3; ax++) {
                if (!(f.a1[ax] > 0.f)) continue;
                const uint32_t b = bucket_of(f.a0[ax], f.a1[ax], p.c[ax][i], nb);
                uint32_t *s = sb + ((uint32_t)ax * nb + b) * 7u;
                for (int k = 0; k < 3; k++) {
                    atomicMin(&s[k], lo[k]);
                    atomicMax(&s[3 + k], hi[k]);
                }
                atomicAdd(&s[6], 1u);
            }
        }
        __syncthreads();
        Bin *dst = bins + (size_t)first_node * 3u * nb;
        for (uint32_t w = threadIdx.x; w < 3u * nb * 7u; w += kThreads) {
            const uint32_t bin = w / 7u, field = w % 7u;
            const uint32_t v = sb[w];
            if (field < 3) {
                if (v != kKeyPosInf) atomicMin(&dst[bin].lo[field], v);
            } else if (field < 6) {
                if (v != kKeyNegInf) atomicMax(&dst[bin].hi[field - 3], v);
            } else if (v) {
                atomicAdd(&dst[bin].cnt, v);
            }
        }
        return;
    }
    for (int it = 0; it < kItems; it++) {
        const uint32_t i = base + it * kThreads + threadIdx.x;
        if (i > last) break;
        const int a = node_of[i];
        if (a < 0) continue;
        const Info &f = info[a];
        if (!f.inner) continue;
        for (int ax = 0; ax < 3; ax++) {
            if (!(f.a1[ax] > 0.f)) continue;
            const uint32_t b = bucket_of(f.a0[ax], f.a1[ax], p.c[ax][i], nb);
            Bin &s = bins[((size_t)a * 3u + (uint32_t)ax) * nb + b];
            for (int k = 0; k < 3; k++) {
                atomicMin(&s.lo[k], fkey(p.lo[k][i]));
                atomicMax(&s.hi[k], fkey(p.hi[k][i]));
            }
            atomicAdd(&s.cnt, 1u);
        }
    }
}

struct DBox {
    float lo[3], hi[3];
};
__device__ inline void dbox_clear(DBox &b)
{
    for (int k = 0; k < 3; k++) {
        b.lo[k] = INFINITY;
        b.hi[k] = -INFINITY;
    }
}
__device__ inline void dbox_add(DBox &b, const DBox &o)
{
    for (int k = 0; k < 3; k++) {
        b.lo[k] = b.lo[k] < o.lo[k] ? b.lo[k] : o.lo[k];
        b.hi[k] = b.hi[k] > o.hi[k] ? b.hi[k] : o.hi[k];
    }
}
__device__ inline float dbox_area(const DBox &b) // src/aabb.rs:37-40
{
    const float sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
    return 2.f * ((sx * sy + sx * sz) + sy * sz);
}
__device__ inline DBox dbox_of(const Bin &s)
{
    DBox b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = funkey(s.lo[k]);
        b.hi[k] = funkey(s.hi[k]);
    }
    return b;
}

// the cheapest candidate of every inner open node (bvh_build.cpp: the candidate loop of build_range<true>),
// its node record, and which of its two children stay open
__global__ __launch_bounds__(128) void split_kernel(const Open *open, Info *info, const Bin *bins, uint32_t count,
                                                    uint32_t nb, uint32_t max_leaf, uint32_t depth,
                                                    rt_flat_node *nodes, uint32_t *child_flag, Totals *tot)
{
    const uint32_t a = blockIdx.x * 128u + threadIdx.x;
    if (a >= count) return;
    Info &f = info[a];
    child_flag[2 * a] = child_flag[2 * a + 1] = 0;
    if (!f.inner) return;
    const Open o = open[a];
    const uint32_t n = o.end - o.begin;
    float min_cost = INFINITY;
    uint32_t min_bucket = 0, axis = 0, nl = 0;
    DBox bl, br;
    dbox_clear(bl);
    dbox_clear(br);
    DBox suf[kMaxB];
    uint32_t scnt[kMaxB];
    for (uint32_t ax = 0; ax < 3; ax++) {
        if (!(f.a1[ax] > 0.f)) continue;
        const Bin *bb = bins + ((size_t)a * 3u + ax) * nb;
        suf[nb - 1] = dbox_of(bb[nb - 1]);
        scnt[nb - 1] = bb[nb - 1].cnt;
        for (int b = (int)nb - 2; b >= 0; b--) {
            suf[b] = suf[b + 1];
            dbox_add(suf[b], dbox_of(bb[b]));
            scnt[b] = scnt[b + 1] + bb[b].cnt;
        }
        DBox pre;
        dbox_clear(pre);
        uint32_t pcnt = 0;
        for (uint32_t i = 0; i + 1 < nb; i++) {
            dbox_add(pre, dbox_of(bb[i]));
            pcnt += bb[i].cnt;
            if (pcnt == 0 || scnt[i + 1] == 0) continue;
            const float cost = (dbox_area(pre) * (float)pcnt + dbox_area(suf[i + 1]) * (float)scnt[i + 1]) / f.parent_area;
            if (cost < min_cost) {
                min_cost = cost;
                min_bucket = i;
                axis = ax;
                nl = pcnt;
                bl = pre;
                br = suf[i + 1];
            }
        }
    }
    if (!(min_cost < INFINITY) || nl == 0 || nl >= n) { // the host builder gives up on such a mesh
        atomicExch(&tot->failed, 1u);
        f.inner = 0;
        return;
    }
    f.axis = axis;
    f.min_bucket = min_bucket;
    f.nl = nl;
    rt_flat_node nd;
    nd.bx[0] = bl.lo[0]; nd.bx[1] = br.lo[0]; nd.bx[2] = bl.hi[0]; nd.bx[3] = br.hi[0];
    nd.by[0] = bl.lo[1]; nd.by[1] = br.lo[1]; nd.by[2] = bl.hi[1]; nd.by[3] = br.hi[1];
    nd.bz[0] = bl.lo[2]; nd.bz[1] = br.lo[2]; nd.bz[2] = bl.hi[2]; nd.bz[3] = br.hi[2];
    // children of at most max_leaf triangles are leaves now; the others report to this node next level
    const uint32_t cn[2] = {nl, n - nl}, cb[2] = {o.begin, o.begin + nl};
    uint32_t ref[2], cnt[2];
    for (int s = 0; s < 2; s++) {
        if (cn[s] <= max_leaf) {
            ref[s] = kLeaf | cb[s];
            cnt[s] = cn[s];
            atomicAdd(&tot->num_leaves, 1u);
            atomicAdd(&tot->num_objs, cn[s]);
            atomicMax(&tot->max_depth, depth + 1u);
            atomicMax(&tot->max_objs, cn[s]);
        } else {
            ref[s] = 0;
            cnt[s] = 0;
            child_flag[2 * a + s] = 1;
        }
    }
    nd.left = ref[0];
    nd.right = ref[1];
    nd.left_count = cnt[0];
    nd.right_count = cnt[1];
    nodes[f.node] = nd;
}

__global__ __launch_bounds__(256) void children_kernel(const Open *open, Info *info, const uint32_t *child_flag,
                                                       const uint32_t *child_scan, uint32_t count, Open *next)
{
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= count) return;
    Info &f = info[a];
    if (!f.inner) return;
    const Open o = open[a];
    for (uint32_t s = 0; s < 2; s++)
        if (child_flag[2 * a + s]) {
            const uint32_t slot = child_scan[2 * a + s];
            f.child_slot[s] = slot;
            next[slot] = Open{s == 0 ? o.begin : o.begin + f.nl, s == 0 ? o.begin + f.nl : o.end, f.node, s};
        }
}

__global__ __launch_bounds__(256) void flag_kernel(Prims p, const int *node_of, uint32_t n, const Info *info,
                                                   uint32_t nb, uint32_t *is_left)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int a = node_of[i];
    uint32_t v = 0;
    if (a >= 0 && info[a].inner) {
        const Info &f = info[a];
        v = bucket_of(f.a0[f.axis], f.a1[f.axis], p.c[f.axis][i], nb) <= f.min_bucket ? 1u : 0u;
    }
    is_left[i] = v;
}

// stable partition of every split node's range; everything else stays where it is
__global__ __launch_bounds__(256) void scatter_kernel(Prims src, Prims dst, const int *node_of, int *node_of_next,
                                                      uint32_t n, const Open *open, const Info *info,
                                                      const uint32_t *is_left, const uint32_t *left_scan)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int a = node_of[i];
    uint32_t to = i;
    int next = -1;
    if (a >= 0 && info[a].inner) {
        const Info &f = info[a];
        const uint32_t begin = open[a].begin;
        const uint32_t lefts_before = left_scan[i] - left_scan[begin];
        const uint32_t side = is_left[i] ? 0u : 1u;
        to = side == 0 ? begin + lefts_before : begin + f.nl + ((i - begin) - lefts_before);
        next = (int)f.child_slot[side]; // ~0u = -1: the child is a leaf
    }
    for (int k = 0; k < 3; k++) {
        dst.c[k][to] = src.c[k][i];
        dst.lo[k][to] = src.lo[k][i];
        dst.hi[k][to] = src.hi[k][i];
    }
    dst.order[to] = src.order[i];
    node_of_next[to] = next;
}

// ------------------------------------------------------------------------
// From the binary tree to the traversal items without leaving the device: 4-wide collapse breadth
// first (the wide node of a binary node depends on that node's descendants alone, so a level of
// wide nodes is one launch), quantization and packing per node and per triangle.  Only the placement
// of the items -- a depth-first walk with a running position, wide_bvh.cpp: layout_wide4 -- stays
// on the host, fed with the 16 bytes of child references per wide node.

struct WidePending { // a wide node still to be made
    uint32_t bin;    // the binary node it is collapsed from
    uint32_t parent; // wide node whose slot refers to it, kNoParent for the root
    uint32_t slot;
};

struct FlatTree { // the builder's node records as the collapse reads them
    const rt_flat_node *nodes;
    __device__ static uint32_t packed(uint32_t ref, uint32_t count)
    {
        return (ref & kLeaf) ? kLeafBit | ((ref & 0x7FFFFFFFu) << 3) | (count - 1u) : ref;
    }
    __device__ void children(uint32_t bin, WideSlot &l, WideSlot &r) const
    {
        const rt_flat_node &n = nodes[bin];
        const float *b[3] = {n.bx, n.by, n.bz};
        for (int k = 0; k < 3; k++) {
            l.lo[k] = b[k][0];
            r.lo[k] = b[k][1];
            l.hi[k] = b[k][2];
            r.hi[k] = b[k][3];
        }
        l.ref = packed(n.left, n.left_count);
        r.ref = packed(n.right, n.right_count);
    }
};

__global__ __launch_bounds__(256) void wide_collapse_kernel(const rt_flat_node *nodes, const WidePending *open,
                                                            uint32_t count, uint32_t wide_base, DevWideNode *wide,
                                                            uint32_t *inner_count)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const WidePending job = open[i];
    WideSlot c[4];
    const int n = collapse_slots(FlatTree{nodes}, job.bin, c);
    fill_wide_node(c, n, job.bin, wide[wide_base + i]);
    uint32_t inner = 0;
    for (int k = 0; k < n; k++) inner += (c[k].ref & kLeafBit) ? 0u : 1u;
    inner_count[i] = inner;
    if (job.parent != kNoParent) wide[job.parent].child[job.slot] = wide_base + i;
}

// the inner children of this level's wide nodes are the next level's jobs, in node and slot order
__global__ __launch_bounds__(256) void wide_children_kernel(const DevWideNode *wide, uint32_t count, uint32_t wide_base,
                                                            const uint32_t *inner_scan, WidePending *next)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const DevWideNode &w = wide[wide_base + i];
    const int n = wide_children_used(w);
    uint32_t to = inner_scan[i];
    for (int k = 0; k < n; k++)
        if (!(w.child[k] & kLeafBit)) next[to++] = WidePending{w.child[k], wide_base + i, (uint32_t)k};
}

__global__ __launch_bounds__(256) void wide_child_refs_kernel(const DevWideNode *wide, uint32_t count, uint32_t *refs)
{
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= count) return;
    const int n = wide_children_used(wide[w]);
    uint4 r;
    r.x = wide[w].child[0];
    r.y = n > 1 ? wide[w].child[1] : kNoWideChild;
    r.z = n > 2 ? wide[w].child[2] : kNoWideChild;
    r.w = n > 3 ? wide[w].child[3] : kNoWideChild;
    reinterpret_cast<uint4 *>(refs)[w] = r;
}

__global__ __launch_bounds__(256) void wide_pack_nodes_kernel(const DevWideNode *wide, uint32_t count,
                                                              const uint32_t *node_unit, const uint32_t *slot_unit,
                                                              Unit *items)
{
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= count) return;
    const DevWideNode &src = wide[w];
    DevQNode q;
    const int n = wide_children_used(src);
    quantize_planes(src, n, q);
    quantize_children(
        n,
        [&](int i) {
            const uint32_t ref = src.child[i];
            return (ref & kLeafBit) ? slot_unit[(ref >> 3) & (kMaxPackedFirst - 1u)] : node_unit[ref];
        },
        [&](int i) { return (src.child[i] & kLeafBit) ? (src.child[i] & 7u) + 1u : 0u; }, q);
    const uint4 *from = reinterpret_cast<const uint4 *>(&q);
    uint4 *to = reinterpret_cast<uint4 *>(items + node_unit[w]);
    to[0] = from[0];
    to[1] = from[1];
    to[2] = from[2];
}

// src/triangle_opt.rs:72-81 for the triangle in slot s of the leaf order, at its units
__global__ __launch_bounds__(256) void wide_pack_tris_kernel(const float *pos, const uint32_t *idx, const uint32_t *order,
                                                             uint32_t n, const uint32_t *slot_unit, Unit *items)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n) return;
    const uint32_t t = order[s];
    const float *a = pos + 3 * (size_t)idx[3 * (size_t)t + 0];
    const float *b = pos + 3 * (size_t)idx[3 * (size_t)t + 1];
    const float *c = pos + 3 * (size_t)idx[3 * (size_t)t + 2];
    DevTri d;
    for (int k = 0; k < 3; k++) {
        d.v0[k] = a[k];
        d.e1[k] = b[k] - a[k];
        d.e2[k] = c[k] - a[k];
    }
    d.id = t;
    d.pad[0] = d.pad[1] = 0;
    const uint4 *from = reinterpret_cast<const uint4 *>(&d);
    uint4 *to = reinterpret_cast<uint4 *>(items + slot_unit[s]);
    to[0] = from[0];
    to[1] = from[1];
    to[2] = from[2];
}

struct DevMem { // frees what it holds when the build ends, however it ends
    std::vector<void *> blocks;
    ~DevMem()
    {
        for (void *p : blocks) (void)hipFree(p);
    }
    template <class T>
    bool get(T *&p, size_t count)
    {
        void *v = nullptr;
        if (hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        blocks.push_back(v);
        p = (T *)v;
        return true;
    }
};

} // namespace

bool build_bvh_search_device(const float *positions, uint32_t nv, const uint32_t *indices, uint32_t nt,
                             BuildTuning tune, HostBvh &out, std::string &why, DeviceWideRequest *wide_request)
{
    if (wide_request) wide_request->done = false;
    const uint32_t nb = std::min<uint32_t>(std::max<uint32_t>(tune.buckets, 2), kMaxB);
    const uint32_t max_leaf = std::min<uint32_t>(std::max<uint32_t>(tune.max_leaf, 1), 8);
    if (nt <= max_leaf) {
        why = "the whole mesh is one leaf";
        return false;
    }
#define GPU_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            why = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
            (void)hipGetLastError();                                                   \
            return false;                                                              \
        }                                                                              \
    } while (0)
    hipStream_t s = nullptr; // the legacy default stream: ordered with the calls below
    // RTCORE_VERBOSE=1: seconds per part of the build on stderr
    const bool verbose = getenv("RTCORE_VERBOSE") && getenv("RTCORE_VERBOSE")[0] == '1';
    auto t_mark = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!verbose) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  device build: %-24s %.3f s\n", what, std::chrono::duration<double>(now - t_mark).count());
        t_mark = now;
    };
    DevMem mem;
    const size_t n = nt;
    // open nodes of a level hold more than max_leaf triangles each
    const size_t max_open = n / (max_leaf + 1u) + 2u;
    float *d_pos = nullptr, *prim_block = nullptr;
    uint32_t *d_idx = nullptr, *order_block = nullptr, *flags = nullptr, *scan = nullptr, *inner_flag = nullptr,
             *inner_scan = nullptr, *child_flag = nullptr, *child_scan = nullptr;
    int *node_of[2] = {nullptr, nullptr};
    Open *open[2] = {nullptr, nullptr};
    Info *info = nullptr;
    rt_flat_node *nodes = nullptr;
    Totals *tot = nullptr;
    bool ok = mem.get(d_pos, 3 * (size_t)nv) && mem.get(d_idx, 3 * n) && mem.get(prim_block, 2 * 9 * n) &&
              mem.get(order_block, 2 * n) && mem.get(node_of[0], n) && mem.get(node_of[1], n) && mem.get(flags, n) &&
              mem.get(scan, n) && mem.get(open[0], max_open) && mem.get(open[1], max_open) && mem.get(info, max_open) &&
              mem.get(inner_flag, max_open) && mem.get(inner_scan, max_open) && mem.get(child_flag, 2 * max_open) &&
              mem.get(child_scan, 2 * max_open) && mem.get(nodes, n) && mem.get(tot, 1);
    if (!ok) {
        why = "device memory for the build could not be allocated";
        return false;
    }
    mark("allocations");
    size_t scan_bytes = 0;
    GPU_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flags, scan, 0u, std::max(n, 2 * max_open),
                                    rocprim::plus<uint32_t>(), s));
    char *scan_temp = nullptr;
    if (!mem.get(scan_temp, scan_bytes + 256)) {
        why = "device memory for the build could not be allocated";
        return false;
    }
    Prims prims[2];
    for (int b = 0; b < 2; b++) {
        float *base = prim_block + (size_t)b * 9 * n;
        for (int k = 0; k < 3; k++) {
            prims[b].c[k] = base + (size_t)k * n;
            prims[b].lo[k] = base + (size_t)(3 + k) * n;
            prims[b].hi[k] = base + (size_t)(6 + k) * n;
        }
        prims[b].order = order_block + (size_t)b * n;
    }
    GPU_TRY(hipMemcpyAsync(d_pos, positions, 3 * (size_t)nv * sizeof(float), hipMemcpyHostToDevice, s));
    GPU_TRY(hipMemcpyAsync(d_idx, indices, 3 * n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    GPU_TRY(hipMemsetAsync(tot, 0, sizeof(Totals), s));
    const unsigned grid_n = (unsigned)((n + 255) / 256), grid_chunks = (unsigned)((n + kChunk - 1) / kChunk);
    hipLaunchKernelGGL(prims_kernel, dim3(grid_n), dim3(256), 0, s, d_pos, d_idx, nt, prims[0], node_of[0]);
    const Open root{0u, nt, kNoParent, 0u};
    GPU_TRY(hipMemcpyAsync(open[0], &root, sizeof root, hipMemcpyHostToDevice, s));

    mark("upload + triangles");
    Bin *bins = nullptr;
    size_t bins_cap = 0;
    uint32_t count = 1; // open nodes of this level
    int cur = 0;
    Totals h{};
    for (uint32_t depth = 0; count != 0; depth++) {
        if (depth > 4096) {
            why = "more than 4096 levels";
            return false;
        }
        const unsigned grid_a = (count + 255u) / 256u;
        const size_t nbins = (size_t)count * 3u * nb;
        if (nbins > bins_cap) { // grown by doubling, at most to what max_open nodes need
            if (bins) {
                GPU_TRY(hipStreamSynchronize(s));
                (void)hipFree(bins);
                mem.blocks.erase(std::find(mem.blocks.begin(), mem.blocks.end(), (void *)bins));
                bins = nullptr;
            }
            bins_cap = std::min(std::max(nbins, 2 * bins_cap), max_open * 3u * nb);
            if (!mem.get(bins, bins_cap)) {
                why = "device memory for the bins of one level could not be allocated";
                return false;
            }
        }
        hipLaunchKernelGGL(info_init_kernel, dim3(grid_a), dim3(256), 0, s, info, count);
        hipLaunchKernelGGL(bounds_kernel, dim3(grid_chunks), dim3(kThreads), 0, s, prims[cur], node_of[cur], nt, info);
        hipLaunchKernelGGL(setup_kernel, dim3(grid_a), dim3(256), 0, s, open[cur], info, count, nb, inner_flag);
        GPU_TRY(rocprim::exclusive_scan(scan_temp, scan_bytes, inner_flag, inner_scan, 0u, (size_t)count,
                                        rocprim::plus<uint32_t>(), s));
        hipLaunchKernelGGL(assign_kernel, dim3(grid_a), dim3(256), 0, s, open[cur], info, inner_scan, count, depth,
                           nodes, tot);
        hipLaunchKernelGGL(bins_init_kernel, dim3((unsigned)((nbins + 255) / 256)), dim3(256), 0, s, bins, nbins);
        hipLaunchKernelGGL(bin_kernel, dim3(grid_chunks), dim3(kThreads), 0, s, prims[cur], node_of[cur], nt, info, bins,
                           nb);
        hipLaunchKernelGGL(split_kernel, dim3((count + 127u) / 128u), dim3(128), 0, s, open[cur], info, bins, count, nb,
                           max_leaf, depth, nodes, child_flag, tot);
        GPU_TRY(rocprim::exclusive_scan(scan_temp, scan_bytes, child_flag, child_scan, 0u, 2 * (size_t)count,
                                        rocprim::plus<uint32_t>(), s));
        hipLaunchKernelGGL(children_kernel, dim3(grid_a), dim3(256), 0, s, open[cur], info, child_flag, child_scan, count,
                           open[cur ^ 1]);
        hipLaunchKernelGGL(flag_kernel, dim3(grid_n), dim3(256), 0, s, prims[cur], node_of[cur], nt, info, nb, flags);
        GPU_TRY(rocprim::exclusive_scan(scan_temp, scan_bytes, flags, scan, 0u, n, rocprim::plus<uint32_t>(), s));
        hipLaunchKernelGGL(scatter_kernel, dim3(grid_n), dim3(256), 0, s, prims[cur], prims[cur ^ 1], node_of[cur],
                           node_of[cur ^ 1], nt, open[cur], info, flags, scan);
        // this level's tallies: inner nodes made, children left open
        uint32_t last_flags[2] = {0, 0}, last_scans[2] = {0, 0};
        GPU_TRY(hipMemcpyAsync(&last_flags[0], inner_flag + (count - 1), 4, hipMemcpyDeviceToHost, s));
        GPU_TRY(hipMemcpyAsync(&last_scans[0], inner_scan + (count - 1), 4, hipMemcpyDeviceToHost, s));
        GPU_TRY(hipMemcpyAsync(&last_flags[1], child_flag + (2 * (size_t)count - 1), 4, hipMemcpyDeviceToHost, s));
        GPU_TRY(hipMemcpyAsync(&last_scans[1], child_scan + (2 * (size_t)count - 1), 4, hipMemcpyDeviceToHost, s));
        GPU_TRY(hipMemcpyAsync(&h, tot, sizeof h, hipMemcpyDeviceToHost, s));
        GPU_TRY(hipStreamSynchronize(s));
        GPU_TRY(hipGetLastError());
        if (h.failed) {
            why = "a node without a valid split candidate (degenerate mesh)";
            return false;
        }
        h.num_nodes += last_flags[0] + last_scans[0];
        GPU_TRY(hipMemcpyAsync(&tot->num_nodes, &h.num_nodes, 4, hipMemcpyHostToDevice, s));
        count = last_flags[1] + last_scans[1];
        if (count > max_open) {
            why = "more open nodes than the build allows for";
            return false;
        }
        cur ^= 1;
    }
    GPU_TRY(hipStreamSynchronize(s));
    mark("levels");
    GPU_TRY(hipMemcpy(&h, tot, sizeof h, hipMemcpyDeviceToHost));
    out.root_ref = h.root_ref;
    out.root_count = h.root_count;
    out.stats.num_leaves = h.num_leaves;
    out.stats.num_objs = h.num_objs;
    out.stats.max_depth = h.max_depth;
    out.stats.max_objs = h.max_objs;

    // ---- the traversal items, if asked for and the tree needs no leaf refinement on the host
    if (wide_request && h.num_nodes != 0 && !(h.root_ref & kLeaf) && h.max_objs <= wide_request->leaf_tris &&
        h.max_objs <= kWideLeafTris && wide_request->alloc) {
        DeviceWideRequest &rq = *wide_request;
        std::string wide_why;
        const bool made = [&]() -> bool {
#define WIDE_TRY(expr)                                                       \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            wide_why = std::string(#expr) + ": " + hipGetErrorString(e_);    \
            (void)hipGetLastError();                                         \
            return false;                                                    \
        }                                                                    \
    } while (0)
            const size_t max_wide = h.num_nodes; // a wide node per binary node at most
            DevWideNode *wide = nullptr;
            WidePending *wopen[2] = {nullptr, nullptr};
            uint32_t *wcount = nullptr, *wscan = nullptr, *refs = nullptr, *d_node_unit = nullptr, *d_slot_unit = nullptr;
            if (!(mem.get(wide, max_wide) && mem.get(wopen[0], max_wide) && mem.get(wopen[1], max_wide) &&
                  mem.get(wcount, max_wide) && mem.get(wscan, max_wide) && mem.get(refs, 4 * max_wide) &&
                  mem.get(d_node_unit, max_wide) && mem.get(d_slot_unit, n))) {
                wide_why = "device memory for the wide tree could not be allocated";
                return false;
            }
            const WidePending first{h.root_ref, kNoParent, 0u};
            WIDE_TRY(hipMemcpyAsync(wopen[0], &first, sizeof first, hipMemcpyHostToDevice, s));
            uint32_t level = 0, wcur = 0, num_wide = 0, wlevel_count = 1;
            while (wlevel_count != 0) {
                if (num_wide + (size_t)wlevel_count > max_wide) {
                    wide_why = "more wide nodes than binary nodes";
                    return false;
                }
                const unsigned grid_w = (wlevel_count + 255u) / 256u;
                hipLaunchKernelGGL(wide_collapse_kernel, dim3(grid_w), dim3(256), 0, s, nodes, wopen[wcur], wlevel_count,
                                   num_wide, wide, wcount);
                WIDE_TRY(rocprim::exclusive_scan(scan_temp, scan_bytes, wcount, wscan, 0u, (size_t)wlevel_count,
                                                 rocprim::plus<uint32_t>(), s));
                hipLaunchKernelGGL(wide_children_kernel, dim3(grid_w), dim3(256), 0, s, wide, wlevel_count, num_wide, wscan,
                                   wopen[wcur ^ 1]);
                uint32_t last[2] = {0, 0};
                WIDE_TRY(hipMemcpyAsync(&last[0], wcount + (wlevel_count - 1), 4, hipMemcpyDeviceToHost, s));
                WIDE_TRY(hipMemcpyAsync(&last[1], wscan + (wlevel_count - 1), 4, hipMemcpyDeviceToHost, s));
                WIDE_TRY(hipStreamSynchronize(s));
                WIDE_TRY(hipGetLastError());
                num_wide += wlevel_count;
                wlevel_count = last[0] + last[1];
                wcur ^= 1;
                level++;
            }
            mark("4-wide collapse");
            // placement on the host, from the child references
            std::vector<uint32_t> host_refs(4 * (size_t)num_wide);
            hipLaunchKernelGGL(wide_child_refs_kernel, dim3((num_wide + 255u) / 256u), dim3(256), 0, s, wide, num_wide, refs);
            WIDE_TRY(hipMemcpy(host_refs.data(), refs, host_refs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            WideLayout layout;
            // a children block spread beyond its six address bits by alignment gaps: the packed placement always fits
            if (!layout_wide4(host_refs.data(), num_wide, 0u, nt, layout, rq.align_mode) &&
                (rq.align_mode == 0 || !layout_wide4(host_refs.data(), num_wide, 0u, nt, layout, 0))) {
                wide_why = "the wide tree cannot be placed";
                return false;
            }
            if (layout.num_units >= (1u << 30)) {
                wide_why = "more than 2^30 units in the wide tree";
                return false;
            }
            mark("layout (host)");
            const size_t units = (size_t)layout.num_units + 1; // one spare unit: a wave-uniform item is read as 64 bytes
            Unit *items = rq.alloc(units);
            if (!items) {
                wide_why = "device memory for the traversal items could not be allocated";
                return false;
            }
            WIDE_TRY(hipMemcpyAsync(d_node_unit, layout.node_unit.data(), (size_t)num_wide * 4, hipMemcpyHostToDevice, s));
            WIDE_TRY(hipMemcpyAsync(d_slot_unit, layout.slot_unit.data(), n * 4, hipMemcpyHostToDevice, s));
            WIDE_TRY(hipMemsetAsync(items, 0, units * sizeof(Unit), s)); // gaps and the spare unit read as zeros
            hipLaunchKernelGGL(wide_pack_nodes_kernel, dim3((num_wide + 255u) / 256u), dim3(256), 0, s, wide, num_wide,
                               d_node_unit, d_slot_unit, items);
            hipLaunchKernelGGL(wide_pack_tris_kernel, dim3(grid_n), dim3(256), 0, s, d_pos, d_idx, prims[cur].order, nt,
                               d_slot_unit, items);
            WIDE_TRY(hipStreamSynchronize(s));
            WIDE_TRY(hipGetLastError());
            mark("quantize + pack");
            rq.num_units = (uint32_t)units;
            rq.root_ref = layout.root_ref;
            rq.wide_depth = level;
            rq.num_wide_nodes = num_wide;
            return true;
#undef WIDE_TRY
        }();
        if (made) {
            rq.done = true;
            out.nodes.clear();
            out.order.clear();
            return true;
        }
        rq.why = wide_why;
    }
    out.nodes.resize(h.num_nodes);
    out.order.resize(nt);
    if (h.num_nodes) GPU_TRY(hipMemcpy(out.nodes.data(), nodes, (size_t)h.num_nodes * sizeof(rt_flat_node), hipMemcpyDeviceToHost));
    GPU_TRY(hipMemcpy(out.order.data(), prims[cur].order, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    mark("download");
#undef GPU_TRY
    return true;
}

} // namespace rtc
