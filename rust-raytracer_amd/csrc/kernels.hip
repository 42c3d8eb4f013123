// HIP kernels of the MI355X ray-tracing core (gfx950, wave64).
//
//   extend   : closest-hit BVH traversal          <- Bvh::intersect_stream
//   connect  : any-hit traversal + radiance add   <- Bvh::occluded1 / occluded_stream
//   generate : camera rays                        <- generate_rays
//   shade    : NEE shadow ray + diffuse bounce    <- integrate_tile_stream loop body
//   accumulate / resolve                          <- pixel accumulation, tonemap
//
// Arithmetic rules (this file is compiled with -ffp-contract=off): every f32
// operation is un-fused and in the reference's order, except the node slab
// test, which uses one fused multiply-subtract per plane exactly where the
// reference calls fused_mul_sub (src/bvh/intersect1.rs:159-164,
// src/bvh/intersect_stream.rs:123-128).  Division and sqrt are the correctly
// rounded ones (hipcc default).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "kernels.h"

namespace rtc {

// ------------------------------------------------------------------------
// small helpers

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 divs(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// nalgebra: dot = (ax*bx + ay*by) + az*bz, cross as below, all un-fused
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 normalize(V3 a) { return divs(a, sqrtf(dot(a, a))); }

__device__ __forceinline__ unsigned lane_id()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` below the calling lane
__device__ __forceinline__ unsigned prefix_count(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Wave-aggregated queue append: one atomic per wave.  Returns the slot of the
// calling lane (valid only where pred is true).
__device__ __forceinline__ unsigned queue_reserve(bool pred, unsigned *counter)
{
    unsigned long long mask = __ballot(pred);
    unsigned total = (unsigned)__popcll(mask);
    unsigned base = 0;
    if (total != 0) {
        unsigned leader = (unsigned)__ffsll((long long)mask) - 1u;
        if (lane_id() == leader) base = atomicAdd(counter, total);
        base = (unsigned)__shfl((int)base, (int)leader);
    }
    return base + prefix_count(mask);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// Block-aggregated queue append.  A global counter takes about 88 M atomics/s
// on one word, so one atomic per wave (260 K per 16 M-ray launch) costs
// milliseconds; here a block of NW waves with ITEMS items per thread and NQ
// output queues makes NQ atomics in total.  Entry e = (q*ITEMS + it)*NW + wave
// of `lds` (NQ*ITEMS*NW <= 64 entries) first holds the wave's count for item
// `it` of queue q, afterwards the queue position of that wave's first element.
template <int NW, int ITEMS, int NQ>
__device__ __forceinline__ void block_reserve(const bool (&pred)[NQ][ITEMS], unsigned *const (&counters)[NQ],
                                              unsigned (&pos)[NQ][ITEMS], unsigned *lds)
{
    static_assert(NQ * ITEMS * NW <= 64, "one wave scans all entries");
    const unsigned wave = threadIdx.x >> 6;
    const unsigned lane = lane_id();
    unsigned pre[NQ][ITEMS];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const unsigned long long mask = __ballot(pred[q][it]);
            pre[q][it] = prefix_count(mask);
            if (lane == 0) lds[(q * ITEMS + it) * NW + wave] = (unsigned)__popcll(mask);
        }
    __syncthreads();
    if (wave == 0) {
        constexpr int E = NQ * ITEMS * NW;
        const unsigned mine = lane < E ? lds[lane] : 0u;
        const unsigned q_of = lane / (ITEMS * NW); // queue of this entry
        // inclusive scan within each queue's run of ITEMS*NW entries
        unsigned incl = mine;
#pragma unroll
        for (int off = 1; off < ITEMS * NW; off <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, off);
            if ((lane % (ITEMS * NW)) >= (unsigned)off) incl += up;
        }
        // last entry of each queue holds that queue's total; lane q reserves for queue q, so the
        // atomics of all queues are in flight together (one round trip, not NQ)
        unsigned my_total = 0;
        unsigned *my_counter = counters[0];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const unsigned total = (unsigned)__shfl((int)incl, (q + 1) * ITEMS * NW - 1);
            if (lane == (unsigned)q) {
                my_total = total;
                my_counter = counters[q];
            }
        }
        unsigned reserved = 0;
        if (lane < (unsigned)NQ && my_total != 0) reserved = atomicAdd(my_counter, my_total);
        unsigned base = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const unsigned b = (unsigned)__shfl((int)reserved, q);
            if (q_of == (unsigned)q) base = b;
        }
        if (lane < E) lds[lane] = base + incl - mine;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int it = 0; it < ITEMS; it++) pos[q][it] = lds[(q * ITEMS + it) * NW + wave] + pre[q][it];
}

// ------------------------------------------------------------------------
// traversal core

struct RayRegs {
    V3 o, d, rd;   // origin, direction, 1/direction
    float near_t;  // dir_recip_near.w
    float far_t;   // origin_far.w, shrinks on closest-hit
};

struct Counters {
    unsigned inner, leaf, tris;
    unsigned ray_start; // inner + tris when the lane's current ray began
    unsigned max_steps; // longest ray this lane has finished
    unsigned max_sp;    // most stack entries this lane has held at once
};

// Per-lane traversal stack: the first kLdsDepth entries live in LDS in
// [entry][thread] order (lane l and lane l+32 hit the same bank in different
// halves of a ds_read_b32, so accesses are conflict-free); deeper entries go to
// a global overflow area laid out the same way.
//
// DIST: the LDS entries also carry the entry distance of their box, as the upper 16 bits of the f32
// (rounded towards zero, i.e. down: conservative), so that a closest-hit walk can drop a deferred
// node that a hit found since has put out of reach without loading it.  The distances of entries
// 2k and 2k+1 share the lane's word k of a second set of planes right behind the entry planes
// (same [plane][thread] layout, addressed from the same base: no second per-lane pointer).  Entries
// in the overflow area carry none (they read as distance 0: never dropped).
template <int BLOCK, int LDS_DEPTH, bool DIST = false>
struct Stack {
    static constexpr bool kDist = DIST;
    unsigned *lds;       // &shared[threadIdx.x]
    static constexpr int kLdsWords = LDS_DEPTH + (DIST ? (LDS_DEPTH + 1) / 2 : 0); // per lane
    __device__ __forceinline__ unsigned short *dist_at(int entry) const
    {
        return reinterpret_cast<unsigned short *>(lds + (LDS_DEPTH + (entry >> 1)) * BLOCK) + (entry & 1);
    }
    unsigned *ovf_base;  // overflow area (the same for every lane: no per-lane 64-bit pointer is kept)
    unsigned ovf_lane;   // this stack's column in it
    unsigned ovf_stride; // columns (stacks in the grid)
    int sp;
    __device__ __forceinline__ unsigned *ovf_at(int entry) const
    {
        return ovf_base + ((size_t)(entry - LDS_DEPTH) * ovf_stride + ovf_lane);
    }
    __device__ __forceinline__ void push(unsigned v)
    {
        if (sp < LDS_DEPTH)
            lds[sp * BLOCK] = v;
        else
            *ovf_at(sp) = v;
        sp++;
    }
    // writes entry `at` without moving the stack pointer
    __device__ __forceinline__ void store(int at, unsigned v)
    {
        if (at < LDS_DEPTH)
            lds[at * BLOCK] = v;
        else
            *ovf_at(at) = v;
    }
    // entry `at` with the sort key of its box (entry-distance bits, slot in the two lowest)
    __device__ __forceinline__ void store_keyed(int at, unsigned v, unsigned key)
    {
        if (at < LDS_DEPTH) {
            lds[at * BLOCK] = v;
            if (DIST) *dist_at(at) = (unsigned short)(key >> 16);
        } else {
            *ovf_at(at) = v;
        }
    }
    // pop() that also gives a lower bound of the entry distance the entry was pushed with
    __device__ __forceinline__ unsigned pop_keyed(float &tnear)
    {
        sp--;
        if (sp < LDS_DEPTH) {
            tnear = DIST ? __uint_as_float((unsigned)*dist_at(sp) << 16) : 0.f;
            return lds[sp * BLOCK];
        }
        tnear = 0.f;
        return *ovf_at(sp);
    }
    __device__ __forceinline__ unsigned load(int at) const
    {
        if (at < LDS_DEPTH) return lds[at * BLOCK];
        return *ovf_at(at);
    }
    __device__ __forceinline__ unsigned pop()
    {
        sp--;
        if (sp < LDS_DEPTH) return lds[sp * BLOCK];
        return *ovf_at(sp);
    }
    // pop() whose (rare) read of the overflow area waits for itself.  A load the compiler can see would
    // make it wait for every vector memory operation at the point where the paths join -- also for
    // gathers that are meant to stay in flight across this call.
    __device__ __forceinline__ unsigned pop_isolated()
    {
        sp--;
        if (sp < LDS_DEPTH) return lds[sp * BLOCK];
        unsigned v;
        const unsigned *p = ovf_at(sp);
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
};

// Moeller-Trumbore of src/triangle_opt.rs:19-42: no back-face cull, |det| <
// f32::EPSILON rejected, u in [0,1], v >= 0, u+v <= 1; returns t or +inf.
__device__ __forceinline__ float tri_test(const float4 t0, const float4 t1, const float2 t2, V3 o, V3 d)
{
    V3 v0 = v3(t0.x, t0.y, t0.z);
    V3 e1 = v3(t0.w, t1.x, t1.y);
    V3 e2 = v3(t1.z, t1.w, t2.x);
    V3 pvec = cross(d, e1);
    float det = dot(e2, pvec);
    float inv_det = 1.f / det;
    V3 tvec = o - v0;
    float u = dot(tvec, pvec) * inv_det;
    V3 qvec = cross(tvec, e2);
    float v = dot(d, qvec) * inv_det;
    float t = dot(e1, qvec) * inv_det;
    bool ok = (fabsf(det) >= 1.1920929e-7f) && (u >= 0.f) && (u <= 1.f) && (v >= 0.f) && (u + v <= 1.f);
    return ok ? t : __builtin_inff();
}

// Traversal of one ray per lane, written as a resumable per-lane state machine
// so that a wave can replace finished rays while the others keep going.
//
// ANYHIT = false: Bvh::intersect_stream seen from one ray
// (src/bvh/intersect_stream.rs:67-181): at an inner node both children are
// tested against [near, current far]; the child with the smaller entry
// distance (left on ties, nearL <= nearR) is descended first and the other is
// deferred without a re-test; triangles are accepted for near <= t < far and
// shrink far (418-422).
//
// ANYHIT = true: Bvh::occluded1 (src/bvh/intersect1.rs:7-106): left child
// first, far fixed, done at the first accepted triangle.
//
// NaN: bb*inf - o*inf can be NaN.  v_max_f32 / v_min_f32 ignore a NaN operand,
// which drops that slab constraint (more visits, never a missed hit), like the
// reference's operand order does on x86.
struct LaneState {
    RayRegs r;
    float ordx, ordy, ordz; // o * (1/d), rounded once per ray (intersect_stream.rs:112-114)
    bool negx, negy, negz;  // sign bit of 1/d (blendv, intersect_stream.rs:116-121)
    unsigned cur;           // node reference being visited, kDone when idle
    unsigned tri, tri_end;  // triangle cursor inside the current leaf (stepwise traversal)
    unsigned hit;           // closest: mesh triangle id or ~0u; any-hit: 1 when occluded
    unsigned index;         // position of the ray in its queue
    unsigned steps;         // loop iterations spent on this ray (straggler detection)
    float far0;             // far distance the ray arrived with (single-ray order only)
    bool have_ray;
};

__device__ __forceinline__ void lane_begin(LaneState &s, unsigned root_ref, const RayRegs &r, unsigned index)
{
    s.r = r;
    s.negx = (__float_as_uint(r.rd.x) >> 31) != 0;
    s.negy = (__float_as_uint(r.rd.y) >> 31) != 0;
    s.negz = (__float_as_uint(r.rd.z) >> 31) != 0;
    s.ordx = r.o.x * r.rd.x;
    s.ordy = r.o.y * r.rd.y;
    s.ordz = r.o.z * r.rd.z;
    s.cur = root_ref;
    s.tri = s.tri_end = 0;
    s.hit = 0xFFFFFFFFu;
    s.index = index;
    s.steps = 0;
    s.far0 = r.far_t;
    s.have_ray = true;
}

// v_max/v_min as single instructions: fmaxf/fminf on values the compiler cannot
// prove quiet are preceded by a canonicalising v_max x,x each.  The hardware
// instructions return the other operand when one is NaN, which is the rule the
// slab tests rely on.
__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Runs the lane's ray until it is done (s.cur == kDone) or until fewer than
// `min_active` lanes of the wave are still traversing (the wave then goes back
// to fetch rays for its idle lanes).
// Which child an any-hit walk of the binary tree takes first when both are hit.  The reference goes left
// (src/bvh/intersect1.rs:7-106), and with visit counters asked for so does this walk.  Without them the
// order is free: the far distance of an any-hit ray never changes, so which leaves the walk can reach --
// and with that the answer -- does not depend on the order, only how soon an occluder is found does.
// RTC_ANYHIT_BINARY_ORDER 0: left first; 1: the child entered first; 2: the child the ray stays in longer.
#ifndef RTC_ANYHIT_BINARY_ORDER
#define RTC_ANYHIT_BINARY_ORDER 2
#endif
template <bool STATS>
__device__ __forceinline__ bool anyhit_left_first(float nearl, float farl, float nearr, float farr)
{
    if (STATS || RTC_ANYHIT_BINARY_ORDER == 0) return true;
    if (RTC_ANYHIT_BINARY_ORDER == 1) return nearl <= nearr;
    return farl - nearl >= farr - nearr;
}

template <bool ANYHIT, bool STATS, class STACK>
__device__ __forceinline__ void traverse(const SceneDev &sc, LaneState &s, STACK &st,
                                         Counters &cnt, unsigned min_active, unsigned max_steps)
{
    const float4 *__restrict__ nodes = reinterpret_cast<const float4 *>(sc.nodes);
    const float4 *__restrict__ tris = reinterpret_cast<const float4 *>(sc.tris);
    RayRegs &r = s.r;
    unsigned cur = s.cur;
    unsigned steps = s.steps;

    // a ray that has used max_steps iterations stops here at a node boundary and
    // is finished by the whole wave (coop_finish)
    while (cur != kDone && steps < max_steps) {
        // ---- inner nodes
        while (!(cur & kLeafBit) && steps < max_steps) {
            steps++;
            float4 bx, by, bz;
            float2 lk; // child refs only (8 of the last 16 bytes)
            {
                const float4 *q = nodes + (size_t)cur * 4;
                bx = q[0];
                by = q[1];
                bz = q[2];
                lk = *reinterpret_cast<const float2 *>(q + 3);
            }
            if (STATS) cnt.inner++;
            // near/far plane per child, picked by the sign bit of 1/d
            const float nxl = s.negx ? bx.z : bx.x, nxr = s.negx ? bx.w : bx.y;
            const float fxl = s.negx ? bx.x : bx.z, fxr = s.negx ? bx.y : bx.w;
            const float nyl = s.negy ? by.z : by.x, nyr = s.negy ? by.w : by.y;
            const float fyl = s.negy ? by.x : by.z, fyr = s.negy ? by.y : by.w;
            const float nzl = s.negz ? bz.z : bz.x, nzr = s.negz ? bz.w : bz.y;
            const float fzl = s.negz ? bz.x : bz.z, fzr = s.negz ? bz.y : bz.w;
            const float tnxl = __fmaf_rn(nxl, r.rd.x, -s.ordx), tnxr = __fmaf_rn(nxr, r.rd.x, -s.ordx);
            const float tfxl = __fmaf_rn(fxl, r.rd.x, -s.ordx), tfxr = __fmaf_rn(fxr, r.rd.x, -s.ordx);
            const float tnyl = __fmaf_rn(nyl, r.rd.y, -s.ordy), tnyr = __fmaf_rn(nyr, r.rd.y, -s.ordy);
            const float tfyl = __fmaf_rn(fyl, r.rd.y, -s.ordy), tfyr = __fmaf_rn(fyr, r.rd.y, -s.ordy);
            const float tnzl = __fmaf_rn(nzl, r.rd.z, -s.ordz), tnzr = __fmaf_rn(nzr, r.rd.z, -s.ordz);
            const float tfzl = __fmaf_rn(fzl, r.rd.z, -s.ordz), tfzr = __fmaf_rn(fzr, r.rd.z, -s.ordz);
            const float nearl = vmax3(tnxl, tnyl, vmax(tnzl, r.near_t));
            const float nearr = vmax3(tnxr, tnyr, vmax(tnzr, r.near_t));
            const float farl = vmin3(tfxl, tfyl, vmin(tfzl, r.far_t));
            const float farr = vmin3(tfxr, tfyr, vmin(tfzr, r.far_t));
            const bool hl = nearl <= farl;
            const bool hr = nearr <= farr;
            const unsigned left = __float_as_uint(lk.x), right = __float_as_uint(lk.y);
            const bool left_first = ANYHIT ? anyhit_left_first<STATS>(nearl, farl, nearr, farr) : (nearl <= nearr);
            const unsigned first = left_first ? left : right;
            const unsigned second = left_first ? right : left;
            const bool hf = left_first ? hl : hr;
            const bool hs = left_first ? hr : hl;
            if (hf) {
                cur = first;
                if (hs) {
                    st.push(second);
                    if (STATS) cnt.max_sp = (unsigned)st.sp > cnt.max_sp ? (unsigned)st.sp : cnt.max_sp;
                }
            } else if (hs) {
                cur = second;
            } else {
                cur = st.sp ? st.pop() : kDone;
            }
        }

        // ---- leaf
        if ((cur & kLeafBit) && cur != kDone) {
            steps++;
            unsigned first, count;
            if (cur & kBigLeafBit) {
                const BigLeaf bl = sc.big_leaves[cur & 0x3FFFFFFFu];
                first = bl.first;
                count = bl.count;
            } else {
                first = (cur >> 3) & (kMaxPackedFirst - 1u);
                count = (cur & 7u) + 1u;
            }
            if (STATS) {
                cnt.leaf++;
                cnt.tris += count;
            }
            bool occluded = false;
            for (unsigned i = 0; i < count; i++) {
                const size_t base = (size_t)(first + i) * 3;
                const float4 t0 = tris[base + 0];
                const float4 t1 = tris[base + 1];
                const float2 t2 = *reinterpret_cast<const float2 *>(tris + base + 2); // v0v2.z, id (8 of 16 B)
                const float t = tri_test(t0, t1, t2, r.o, r.d);
                if (t >= r.near_t && t < r.far_t) {
                    if (ANYHIT) {
                        occluded = true;
                        break;
                    }
                    r.far_t = t;
                    s.hit = __float_as_uint(t2.y);
                }
            }
            if (ANYHIT && occluded) {
                s.hit = 1u;
                cur = kDone;
            } else {
                cur = st.sp ? st.pop() : kDone;
            }
        }

        // ---- too few lanes still traversing: let the wave refill
        if ((unsigned)__popcll(__ballot(true)) < min_active) break;
    }
    s.cur = cur;
    s.steps = steps;
}

// Stepwise form of the same traversal: every loop iteration a lane handles ONE
// item -- an inner node (both child boxes) or one triangle of its current leaf
// -- and all lanes issue their loads together, whichever kind they are at.
// Lanes at nodes never wait for lanes inside leaves and vice versa, which is
// what starves the while-while form on incoherent rays (a third of the lanes
// active on San-Miguel-class bounce rays).  The per-ray order of node visits
// and triangle tests is unchanged, so results and counters are identical.
//
// SINGLE = true walks in the order of Bvh::intersect1 (src/bvh/intersect1.rs:109-210) instead:
// left child first whatever the entry distances (181-191), and boxes culled against the far
// distance the ray ARRIVED with -- the reference loads its near/far register once before the loop
// (128-129) -- while triangles are accepted against the shrinking far (198).  Same closest
// distance; other visit counters, and on an exact-distance tie the triangle found first in that order.
template <bool ANYHIT, bool STATS, bool SINGLE, class STACK>
__device__ __forceinline__ void traverse_steps(const SceneDev &sc, LaneState &s, STACK &st,
                                               Counters &cnt, unsigned min_active, unsigned max_steps)
{
    const float4 *__restrict__ nodes = reinterpret_cast<const float4 *>(sc.nodes);
    const float4 *__restrict__ tris = reinterpret_cast<const float4 *>(sc.tris);
    RayRegs &r = s.r;
    unsigned cur = s.cur, tri = s.tri, tri_end = s.tri_end;
    unsigned steps = s.steps;

    // stop at a node boundary (never inside a leaf) once the ray has used max_steps
    while (cur != kDone && (steps < max_steps || tri < tri_end)) {
        steps++;
        // entering a leaf (from a node step, a pop, or a leaf root)
        if ((cur & kLeafBit) && tri >= tri_end) {
            unsigned first, count;
            if (cur & kBigLeafBit) {
                const BigLeaf bl = sc.big_leaves[cur & 0x3FFFFFFFu];
                first = bl.first;
                count = bl.count;
            } else {
                first = (cur >> 3) & (kMaxPackedFirst - 1u);
                count = (cur & 7u) + 1u;
            }
            tri = first;
            tri_end = first + count;
            if (STATS) {
                cnt.leaf++;
                cnt.tris += count;
            }
        }
        const bool in_leaf = tri < tri_end;
        float4 a0, a1, a2;
        float2 lk = make_float2(0.f, 0.f); // child refs (nodes only)
        {
            const float4 *p = in_leaf ? tris + (size_t)tri * 3 : nodes + (size_t)cur * 4;
            a0 = p[0];
            a1 = p[1];
            a2 = p[2];
            if (!in_leaf) lk = *reinterpret_cast<const float2 *>(p + 3);
        }
        bool need_pop = false;
        if (!in_leaf) {
            if (STATS) cnt.inner++;
            const float nxl = s.negx ? a0.z : a0.x, nxr = s.negx ? a0.w : a0.y;
            const float fxl = s.negx ? a0.x : a0.z, fxr = s.negx ? a0.y : a0.w;
            const float nyl = s.negy ? a1.z : a1.x, nyr = s.negy ? a1.w : a1.y;
            const float fyl = s.negy ? a1.x : a1.z, fyr = s.negy ? a1.y : a1.w;
            const float nzl = s.negz ? a2.z : a2.x, nzr = s.negz ? a2.w : a2.y;
            const float fzl = s.negz ? a2.x : a2.z, fzr = s.negz ? a2.y : a2.w;
            const float tnxl = __fmaf_rn(nxl, r.rd.x, -s.ordx), tnxr = __fmaf_rn(nxr, r.rd.x, -s.ordx);
            const float tfxl = __fmaf_rn(fxl, r.rd.x, -s.ordx), tfxr = __fmaf_rn(fxr, r.rd.x, -s.ordx);
            const float tnyl = __fmaf_rn(nyl, r.rd.y, -s.ordy), tnyr = __fmaf_rn(nyr, r.rd.y, -s.ordy);
            const float tfyl = __fmaf_rn(fyl, r.rd.y, -s.ordy), tfyr = __fmaf_rn(fyr, r.rd.y, -s.ordy);
            const float tnzl = __fmaf_rn(nzl, r.rd.z, -s.ordz), tnzr = __fmaf_rn(nzr, r.rd.z, -s.ordz);
            const float tfzl = __fmaf_rn(fzl, r.rd.z, -s.ordz), tfzr = __fmaf_rn(fzr, r.rd.z, -s.ordz);
            const float nearl = vmax3(tnxl, tnyl, vmax(tnzl, r.near_t));
            const float nearr = vmax3(tnxr, tnyr, vmax(tnzr, r.near_t));
            const float cull_far = SINGLE ? s.far0 : r.far_t;
            const float farl = vmin3(tfxl, tfyl, vmin(tfzl, cull_far));
            const float farr = vmin3(tfxr, tfyr, vmin(tfzr, cull_far));
            const bool hl = nearl <= farl;
            const bool hr = nearr <= farr;
            const unsigned left = __float_as_uint(lk.x), right = __float_as_uint(lk.y);
            const bool left_first = SINGLE   ? true
                                    : ANYHIT ? anyhit_left_first<STATS>(nearl, farl, nearr, farr)
                                             : (nearl <= nearr);
            const unsigned first = left_first ? left : right;
            const unsigned second = left_first ? right : left;
            const bool hf = left_first ? hl : hr;
            const bool hs = left_first ? hr : hl;
            if (hf) {
                cur = first;
                if (hs) {
                    st.push(second);
                    if (STATS) cnt.max_sp = (unsigned)st.sp > cnt.max_sp ? (unsigned)st.sp : cnt.max_sp;
                }
            } else if (hs) {
                cur = second;
            } else {
                need_pop = true;
            }
        } else {
            const float t = tri_test(a0, a1, make_float2(a2.x, a2.y), r.o, r.d);
            tri++;
            if (t >= r.near_t && t < r.far_t) {
                if (ANYHIT) {
                    s.hit = 1u;
                    cur = kDone;
                    tri = tri_end;
                } else {
                    r.far_t = t;
                    s.hit = __float_as_uint(a2.y);
                }
            }
            need_pop = (tri >= tri_end) && (cur != kDone);
        }
        if (need_pop) cur = st.sp ? st.pop() : kDone;
        if ((unsigned)__popcll(__ballot(true)) < min_active) break;
    }
    s.cur = cur;
    s.tri = tri;
    s.tri_end = tri_end;
    s.steps = steps;
}

// The three 16-byte loads of one traversal item (a node or a triangle of the wide tree, 48 bytes)
// as exactly three vector memory instructions, written out so that the compiler neither trims the
// third load to the bytes one kind of item needs nor splits it: every instruction is a pass of its
// lanes through the CU's address unit and L1 tag pipeline, the busiest units of these kernels.
// RTC_PROBE_EXTRA_LOADS (measurement builds only): 1 or 2 more loads of the item's first 16 bytes, which hit
// the line the first load brought in: more work for the address unit and the L1 tag pipeline, no more
// misses.  (San-Miguel-class deep extend launch: +33 % L1 accesses, L1 pipeline 0.80 -> 0.985 accesses
// per clock and CU, launch time +7 %.)
#ifndef RTC_PROBE_EXTRA_LOADS
#define RTC_PROBE_EXTRA_LOADS 0
#endif
__device__ __forceinline__ void load_item(const float4 *p, float4 &a0, float4 &a1, float4 &a2)
{
#if RTC_PROBE_EXTRA_LOADS
    float4 x0, x1;
    asm volatile("global_load_dwordx4 %0, %5, off\n\t"
                 "global_load_dwordx4 %1, %5, off offset:16\n\t"
                 "global_load_dwordx4 %2, %5, off offset:32\n\t"
                 "global_load_dwordx4 %3, %5, off\n\t"
#if RTC_PROBE_EXTRA_LOADS > 1
                 "global_load_dwordx4 %4, %5, off offset:16\n\t"
#endif
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(x0), "=&v"(x1)
                 : "v"(p)
                 : "memory");
    return;
#endif
    asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2)
                 : "v"(p)
                 : "memory");
}

// The same three loads with one cache policy for the lanes in `tmask` (inside a leaf: triangles, read once
// per ray and rarely shared) and another for the rest (nodes, shared by the rays of a wave and of the CU):
// six instructions, each for part of the lanes, one wait.  RTC_TRI_LOAD_POLICY / RTC_NODE_LOAD_POLICY are the
// policies by number, see RTC_POLICY_TEXT_n.
#ifndef RTC_SPLIT_LOADS
#define RTC_SPLIT_LOADS 0
#endif
#define RTC_POLICY_TEXT_0 ""
#define RTC_POLICY_TEXT_1 "nt"
#define RTC_POLICY_TEXT_2 "sc1"
#define RTC_POLICY_TEXT_3 "sc1 nt"
#define RTC_POLICY_TEXT_4 "sc0 sc1"
#define RTC_POLICY_TEXT_5 "sc0"
#define RTC_POLICY_TEXT_(id) RTC_POLICY_TEXT_##id
#define RTC_POLICY_TEXT(id) RTC_POLICY_TEXT_(id)
#ifndef RTC_TRI_POLICY
#define RTC_TRI_POLICY 1
#endif
#ifndef RTC_NODE_POLICY
#define RTC_NODE_POLICY 0
#endif
#define RTC_TRI_LOAD_POLICY RTC_POLICY_TEXT(RTC_TRI_POLICY)
#define RTC_NODE_LOAD_POLICY RTC_POLICY_TEXT(RTC_NODE_POLICY)
__device__ __forceinline__ void load_item_split(const float4 *p, unsigned long long tmask, float4 &a0, float4 &a1,
                                                float4 &a2)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %3, exec\n\t"
                 "s_and_b64 exec, %3, %5\n\t"
                 "global_load_dwordx4 %0, %4, off " RTC_TRI_LOAD_POLICY "\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16 " RTC_TRI_LOAD_POLICY "\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 " RTC_TRI_LOAD_POLICY "\n\t"
                 "s_andn2_b64 exec, %3, %5\n\t"
                 "global_load_dwordx4 %0, %4, off " RTC_NODE_LOAD_POLICY "\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16 " RTC_NODE_LOAD_POLICY "\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 " RTC_NODE_LOAD_POLICY "\n\t"
                 "s_mov_b64 exec, %3\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&s"(saved)
                 : "v"(p), "s"(tmask)
                 : "memory", "scc");
}

// RTC_PROBE_MISS_LOADS (measurement builds only): one more 16-byte load per lane and iteration from a
// pseudo-random place of the item array, i.e. one more miss per lane.
#ifndef RTC_PROBE_VALU
#define RTC_PROBE_VALU 0
#endif
#ifndef RTC_PROBE_MISS_LOADS
#define RTC_PROBE_MISS_LOADS 0
#endif
__device__ __forceinline__ void load_item_and_miss(const float4 *p, const float4 *far_p, float4 &a0, float4 &a1,
                                                   float4 &a2)
{
    float4 x0;
    asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\t"
                 "global_load_dwordx4 %3, %5, off\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(x0)
                 : "v"(p), "v"(far_p)
                 : "memory");
}

// 64 bytes through the scalar cache into SGPRs: used when every active lane of
// the wave is at the same node or triangle (coherent rays near the top of the
// tree), which takes that item off the vector memory path altogether.
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ u32x16 scalar_load64(const void *p)
{
    u32x16 n;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n) : "s"(p) : "memory");
    return n;
}
__device__ __forceinline__ float4 f4_of(const u32x16 &n, int i)
{
    return make_float4(__uint_as_float(n[i]), __uint_as_float(n[i + 1]), __uint_as_float(n[i + 2]),
                       __uint_as_float(n[i + 3]));
}

// compare-exchange of two packed (entry distance, slot) keys
#define RTC_CE(ka, kb)                          \
    {                                           \
        const unsigned lo_ = ka < kb ? ka : kb; \
        kb = ka < kb ? kb : ka;                 \
        ka = lo_;                               \
    }

constexpr unsigned kMissKey = 0xFFFFFFFFu;

// traversal reference of the child in slot `slot` (0..3) of a node: base4 + meta byte
__device__ __forceinline__ unsigned child_ref(unsigned base4, unsigned meta, unsigned slot)
{
    // v_perm_b32: selector bytes 0..3 pick a byte of `meta`, 0x0c gives 0x00
    return base4 + __builtin_amdgcn_perm(0u, meta, slot | 0x0C0C0C00u);
}

// Entry and exit distance of the four child boxes of a quantized node (DevQNode as three 16-byte
// pieces: a0 = origin.xyz, exps; a1 = base4, meta, lo.x, lo.y; a2 = lo.z, hi.xyz).
// Child planes are decoded in ray space: t = byte * (step/d) + (origin - o)/d.
struct SlabRay { // what the box test needs of a ray
    V3 rd;                  // 1/d
    float ordx, ordy, ordz; // o * (1/d)
    bool negx, negy, negz;  // sign bit of 1/d
};
__device__ __forceinline__ SlabRay slab_ray(const LaneState &s)
{
    return SlabRay{s.r.rd, s.ordx, s.ordy, s.ordz, s.negx, s.negy, s.negz};
}
// the wide forms do not keep o * (1/d) per ray: three multiplications per node instead of three registers
__device__ __forceinline__ SlabRay slab_ray_lean(const LaneState &s)
{
    return SlabRay{s.r.rd, s.r.o.x * s.r.rd.x, s.r.o.y * s.r.rd.y, s.r.o.z * s.r.rd.z, s.negx, s.negy, s.negz};
}
__device__ __forceinline__ SlabRay slab_ray(const RayRegs &r)
{
    return SlabRay{r.rd, r.o.x * r.rd.x, r.o.y * r.rd.y, r.o.z * r.rd.z, (__float_as_uint(r.rd.x) >> 31) != 0,
                   (__float_as_uint(r.rd.y) >> 31) != 0, (__float_as_uint(r.rd.z) >> 31) != 0};
}

__device__ __forceinline__ void qnode_slabs(const float4 a0, const float4 a1, const float4 a2, const SlabRay &s,
                                            float near_t, float far_t, float (&n)[4], float (&f)[4])
{
    const SlabRay &r = s;
    const unsigned lox = __float_as_uint(a1.z), loy = __float_as_uint(a1.w), loz = __float_as_uint(a2.x);
    const unsigned hix = __float_as_uint(a2.y), hiy = __float_as_uint(a2.z), hiz = __float_as_uint(a2.w);
    const unsigned nxw = s.negx ? hix : lox, fxw = s.negx ? lox : hix;
    const unsigned nyw = s.negy ? hiy : loy, fyw = s.negy ? loy : hiy;
    const unsigned nzw = s.negz ? hiz : loz, fzw = s.negz ? loz : hiz;
    // grid steps: the top ten bits of a positive float each (exponent and two mantissa bits)
    const unsigned e = __float_as_uint(a0.w);
    const float gx = __uint_as_float((e << 21) & 0x7FE00000u), gy = __uint_as_float((e << 11) & 0x7FE00000u),
                gz = __uint_as_float((e << 1) & 0x7FE00000u);
    const float sx = gx * r.rd.x, sy = gy * r.rd.y, sz = gz * r.rd.z;
    const float bx = __fmaf_rn(a0.x, r.rd.x, -s.ordx), by = __fmaf_rn(a0.y, r.rd.y, -s.ordy),
                bz = __fmaf_rn(a0.z, r.rd.z, -s.ordz);
#define RTC_BYTE(w, i) ((float)(((w) >> (8 * (i))) & 0xFFu))
#pragma unroll
    for (int i = 0; i < 4; i++) {
        n[i] = vmax3(__fmaf_rn(RTC_BYTE(nxw, i), sx, bx), __fmaf_rn(RTC_BYTE(nyw, i), sy, by),
                     vmax(__fmaf_rn(RTC_BYTE(nzw, i), sz, bz), near_t));
        f[i] = vmin3(__fmaf_rn(RTC_BYTE(fxw, i), sx, bx), __fmaf_rn(RTC_BYTE(fyw, i), sy, by),
                     vmin(__fmaf_rn(RTC_BYTE(fzw, i), sz, bz), far_t));
    }
#undef RTC_BYTE
}

// One quantized wide node: tests the four child boxes, moves `cur` to the first
// child to visit and defers the others; returns true when no child is hit (the
// caller pops).
template <bool ANYHIT, class STACK>
__device__ __forceinline__ bool qwide_step(const float4 a0, const float4 a1, const float4 a2, const SlabRay &sr,
                                           float near_t, float far_t, STACK &st, unsigned &cur)
{
    float n[4], f[4];
    qnode_slabs(a0, a1, a2, sr, near_t, far_t, n, f);
    const unsigned base4 = __float_as_uint(a1.x), meta = __float_as_uint(a1.y);
    if (!ANYHIT || RTC_ANYHIT_SORTED) {
        // Hit children nearest first: the entry distance (never negative, so its bits order like an
        // unsigned) with the slot number in the two lowest mantissa bits, sorted by a 4-input network
        // of unsigned min/max; ties go to the lower slot.
        // (any-hit with RTC_ANYHIT_SORTED 2: by the length of the ray's stay in the box instead, longest first:
        // the bits of a non-negative length, inverted)
        const bool by_stay = ANYHIT && RTC_ANYHIT_SORTED == 2;
#define RTC_CHILD_KEY(i) \
    (n[i] <= f[i] ? ((by_stay ? ~__float_as_uint(f[i] - n[i]) & 0x7FFFFFFFu : __float_as_uint(n[i])) & ~3u) | (unsigned)(i) : kMissKey)
        unsigned k0 = RTC_CHILD_KEY(0);
        unsigned k1 = RTC_CHILD_KEY(1);
        unsigned k2 = RTC_CHILD_KEY(2);
        unsigned k3 = RTC_CHILD_KEY(3);
#undef RTC_CHILD_KEY
        RTC_CE(k0, k1);
        RTC_CE(k2, k3);
        RTC_CE(k0, k2);
        RTC_CE(k1, k3);
        RTC_CE(k1, k2);
        if (k0 == kMissKey) return true;
        cur = child_ref(base4, meta, k0 & 3u);
        const int m = (k1 != kMissKey ? 1 : 0) + (k2 != kMissKey ? 1 : 0) + (k3 != kMissKey ? 1 : 0);
        if (m == 3) st.store_keyed(st.sp, child_ref(base4, meta, k3 & 3u), k3);
        if (m >= 2) st.store_keyed(st.sp + m - 2, child_ref(base4, meta, k2 & 3u), k2);
        if (m >= 1) st.store_keyed(st.sp + m - 1, child_ref(base4, meta, k1 & 3u), k1);
        st.sp += m;
        return false;
    } else {
        // any-hit: the answer does not depend on the order; take the hit children in slot
        // order (src/bvh/intersect1.rs:7-106 goes left first as well)
        unsigned next = kDone;
#pragma unroll
        for (int i = 3; i >= 0; i--)
            if (n[i] <= f[i]) {
                if (next != kDone) st.push(next);
                next = base4 + ((meta >> (8 * i)) & 0xFFu);
            }
        if (next == kDone) return true;
        cur = next;
        return false;
    }
}

// Next deferred item of a closest-hit walk on the wide tree (kDone when the stack is empty).  With
// entry distances on the stack, up to two entries that start at or beyond the current far distance
// are dropped on the way: a third candidate is taken as it is (its box test will reject it), so the
// number of dependent LDS reads per iteration stays bounded.
template <bool ANYHIT, class STACK>
__device__ __forceinline__ unsigned pop_next(STACK &st, float far_t)
{
    if (ANYHIT || !STACK::kDist) return st.sp ? st.pop() : kDone;
    unsigned cur = kDone;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (cur == kDone && st.sp) {
            float tn;
            const unsigned c = st.pop_keyed(tn);
            if (k == 2 || !(tn >= far_t)) cur = c;
        }
    }
    return cur;
}

// Stepwise traversal of the quantized 4-wide tree: an iteration handles one
// item -- a wide node (four child boxes) or one triangle, 48 bytes either way --
// and its three 16-byte loads are the same instructions for both kinds, so an
// iteration costs three vector memory instructions whatever the lanes are
// doing, and a wide node stands for about 2.2 binary nodes.  The trace kernels
// are bound by the CU's vector memory path (L1 line lookups, misses in flight)
// as much as by arithmetic, which is why trading bytes for decode arithmetic
// pays.
//
// Child planes are decoded in ray space: t = byte * (step/d) + (origin - o)/d.
// The boxes are conservative (see DevQNode), the hit children are visited
// nearest first (ties in slot order = left to right) and the others deferred
// nearest on top.  The triangle arithmetic is that of the binary forms, so hit
// distances agree with them except where the visit order matters to rounding
// (two surfaces within an ulp, boxes culled at a face); ids may differ on exact
// ties.  No visit counters in this form.
template <bool ANYHIT, class STACK>
__device__ __forceinline__ void traverse_wide_steps(const SceneDev &sc, LaneState &s, STACK &st, unsigned min_active,
                                                    unsigned max_steps)
{
    const float4 *__restrict__ items = reinterpret_cast<const float4 *>(sc.witems);
    RayRegs &r = s.r;
    // cur: reference (unit << 2 | triangles) of the item being visited; leaf (kept in s.tri): unit << 2 |
    // triangles still to test of the leaf the ray is in, the encoding of a reference, so that entering
    // a leaf is one move
    unsigned cur = s.cur, leaf = s.tri;
    unsigned steps = s.steps;

    while (cur != kDone && (steps < max_steps || (leaf & 3u))) {
        steps++;
#if RTC_PROBE_VALU
        {   // measurement builds only: RTC_PROBE_VALU more vector instructions per iteration, no memory access
            float x = r.near_t;
#pragma unroll
            for (int k = 0; k < RTC_PROBE_VALU; k++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
            asm volatile("" ::"v"(x));
        }
#endif
        if ((cur & 3u) && !(leaf & 3u)) leaf = cur; // entering a leaf
        const bool in_leaf = (leaf & 3u) != 0;
        const float4 *p = items + ((in_leaf ? leaf : cur) >> 2);
        float4 a0, a1, a2;
#if RTC_SPLIT_LOADS
        load_item_split(p, __ballot(in_leaf), a0, a1, a2);
#elif RTC_PROBE_MISS_LOADS
        load_item_and_miss(p, items + ((((cur >> 2) + steps) * 2654435761u) % 30000000u), a0, a1, a2); // scenes of 3e7 units and more only
#else
        load_item(p, a0, a1, a2);
#endif
        bool need_pop = false;
        if (!in_leaf) {
            need_pop = qwide_step<ANYHIT>(a0, a1, a2, slab_ray_lean(s), r.near_t, r.far_t, st, cur);
        } else {
            const float t = tri_test(a0, a1, make_float2(a2.x, a2.y), r.o, r.d);
            leaf += (kUnitsPerItem << 2) - 1u; // the next triangle, one fewer to go
            if (t >= r.near_t && t < r.far_t) {
                if (ANYHIT) {
                    s.hit = 1u;
                    cur = kDone;
                    leaf = 0;
                } else {
                    r.far_t = t;
                    s.hit = __float_as_uint(a2.y);
                }
            }
            need_pop = !(leaf & 3u) && (cur != kDone);
        }
        if (need_pop) cur = pop_next<ANYHIT>(st, r.far_t);
        if ((unsigned)__popcll(__ballot(true)) < min_active) break;
    }
    s.cur = cur;
    s.tri = leaf;
    s.steps = steps;
}
// ------------------------------------------------------------------------
// Stepwise traversal that takes a leaf and the item that follows it in one iteration (form 8).
//
// A wave-iteration of traverse_wide_steps() ends in a wait for the slowest gather of its lanes, and a ray
// needs about as many iterations for its triangles, one each, as for its nodes.  After a leaf the walk
// always goes on with the next deferred item, and
// which one that is does not depend on the leaf's triangles.  So an iteration that stands at a leaf
// pops that item first and loads it together with the leaf's first two triangles (nine loads in
// flight, one wait); then the triangles are tested one after the other, and the node, against the far
// distance they leave behind.  The visits, their order and the arithmetic are those of
// traverse_wide_steps(): same distances, same ids.  A third triangle of a leaf (trees built with
// leaves of three) is fetched in a pass of its own.  Measured on the San-Miguel-class bounce rays:
// 28 % fewer wave-iterations, 6 % MORE vector instructions (the triangle code runs twice per
// iteration, each time for fewer lanes), 93 registers (5 waves per SIMD): the launch is 3 % slower
// than with form 2.  Kept as a measured alternative; not selected by default.
__device__ __forceinline__ void load_leaf_and_node(const float4 *pt, unsigned long long amask, unsigned long long bmask,
                                                   const float4 *pn, unsigned long long nmask, float4 &t0, float4 &t1,
                                                   float2 &t2, float4 &u0, float4 &u1, float2 &u2, float4 &n0,
                                                   float4 &n1, float4 &n2)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %9, exec\n\t"
                 "s_mov_b64 exec, %12\n\t"
                 "global_load_dwordx4 %0, %10, off\n\t"
                 "global_load_dwordx4 %1, %10, off offset:16\n\t"
                 "global_load_dwordx2 %2, %10, off offset:32\n\t"
                 "s_mov_b64 exec, %13\n\t"
                 "global_load_dwordx4 %3, %10, off offset:48\n\t"
                 "global_load_dwordx4 %4, %10, off offset:64\n\t"
                 "global_load_dwordx2 %5, %10, off offset:80\n\t"
                 "s_mov_b64 exec, %14\n\t"
                 "global_load_dwordx4 %6, %11, off\n\t"
                 "global_load_dwordx4 %7, %11, off offset:16\n\t"
                 "global_load_dwordx4 %8, %11, off offset:32\n\t"
                 "s_mov_b64 exec, %9\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(n0), "=&v"(n1), "=&v"(n2),
                   "=&s"(saved)
                 : "v"(pt), "v"(pn), "s"(amask), "s"(bmask), "s"(nmask)
                 : "memory");
}

template <bool ANYHIT, class STACK>
__device__ __forceinline__ void traverse_wide_pairs(const SceneDev &sc, LaneState &s, STACK &st, unsigned min_active,
                                                    unsigned max_steps)
{
    static_assert(kUnitsPerItem == 3, "the second triangle of a leaf is read 48 bytes behind the first");
    const float4 *__restrict__ items = reinterpret_cast<const float4 *>(sc.witems);
    RayRegs &r = s.r;
    unsigned cur = s.cur; // reference (unit << 2 | triangles) of the item the ray stands at
    unsigned steps = s.steps;

    while (cur != kDone && steps < max_steps) {
        steps++;
        const unsigned tris = cur & 3u;
        const float4 *pt = items + (cur >> 2);
        unsigned next = cur; // the item this iteration ends at: `cur` itself, or what follows the leaf
        if (tris) next = pop_next<ANYHIT>(st, r.far_t);
        const bool at_node = next != kDone && !(next & 3u);
        float4 t0, t1, u0, u1, n0, n1, n2;
        float2 t2, u2;
        load_leaf_and_node(pt, __ballot(tris != 0), __ballot(tris >= 2), items + (next >> 2), __ballot(at_node), t0, t1,
                           t2, u0, u1, u2, n0, n1, n2);
        if (tris) {
            cur = next;
            float t = tri_test(t0, t1, t2, r.o, r.d);
            if (t >= r.near_t && t < r.far_t) {
                r.far_t = t;
                s.hit = ANYHIT ? 1u : __float_as_uint(t2.y);
            }
            if (tris >= 2) {
                t = tri_test(u0, u1, u2, r.o, r.d);
                if (t >= r.near_t && t < r.far_t) {
                    r.far_t = t;
                    s.hit = ANYHIT ? 1u : __float_as_uint(u2.y);
                }
            }
            if (kWideLeafTris > 2 && tris == 3) {
                float4 a0, a1, a2;
                load_item(pt + 2 * kUnitsPerItem, a0, a1, a2);
                t = tri_test(a0, a1, make_float2(a2.x, a2.y), r.o, r.d);
                if (t >= r.near_t && t < r.far_t) {
                    r.far_t = t;
                    s.hit = ANYHIT ? 1u : __float_as_uint(a2.y);
                }
            }
            if (ANYHIT && s.hit == 1u) cur = kDone;
        }
        if (at_node && cur != kDone) {
            if (qwide_step<ANYHIT>(n0, n1, n2, slab_ray_lean(s), r.near_t, r.far_t, st, cur))
                cur = pop_next<ANYHIT>(st, r.far_t);
        }
        if ((unsigned)__popcll(__ballot(true)) < min_active) break;
    }
    s.cur = cur;
    s.steps = steps;
}

// ------------------------------------------------------------------------
// Stepwise traversal with a wave-level triangle-job queue (form 7, incoherent rays).
//
// In traverse_wide_steps() every iteration runs the node code for the lanes that are at a node and
// the triangle code for the lanes inside a leaf, and with incoherent rays a wave always has both
// kinds: 28 of 64 lanes are active per instruction.  Here a lane only ever walks nodes.  A leaf it
// reaches is not tested by that lane: one job per triangle -- owner lane and triangle -- goes
// into a ring the wave keeps in LDS, and the lane goes straight on with its next deferred node.
// When 64 jobs are pending the whole wave tests 64 triangles at once, whichever rays they belong
// to: the lane that runs a job fetches the owner's ray from the owner's registers (ds_bpermute)
// and merges an accepted distance into the owner's result slot in LDS with one 64-bit minimum on
// (distance, triangle id).  This is what the reference's stream traversal does on the CPU when it
// gathers the rays that reached a leaf and tests them four at a time
// (src/bvh/intersect_stream.rs:384-426), with the wave as the batch.
//
// What changes for a ray: between reaching a leaf and the next triangle phase it walks on against
// a far distance that the pending jobs may still shorten (a few extra visits, as in
// traverse_wide_split()), and of several triangles at exactly the same distance the one with the
// lowest id wins whatever the visit order.  Closest distances are those of the other wide forms:
// every triangle the ordered walk would test is still tested, with the same arithmetic.
#ifndef RTC_Q_NODE_MIN
#define RTC_Q_NODE_MIN 24 // with fewer lanes than this at a node, pending jobs and leaves are dealt with first
#endif
// 1: a node step and a triangle phase share one wait for memory (their loads are in flight together).
// Needs 127 VGPRs (4 waves per SIMD) to stay clear of spills: measured 20.2 ms against 22.7 ms for the
// plain form at 6 waves and 16.7 ms for traverse_wide_steps() on the San-Miguel-class deep extend launch.
#ifndef RTC_Q_FUSE
#define RTC_Q_FUSE 0
#endif
#ifndef RTC_Q_FUSE_MIN
#define RTC_Q_FUSE_MIN 24 // pending jobs that are worth a triangle phase beside a node step
#endif
#ifndef RTC_Q_LEAF_MIN
#define RTC_Q_LEAF_MIN 12 // leaves at hand in the wave that are worth an enqueue round of their own
#endif
constexpr unsigned kJobRing = 256;     // entries per wave: fewer than 64 pending before a round adds at most 3 * 64
constexpr unsigned kJobUnitBits = 26;  // a job is owner lane << 26 | first unit of the triangle
static_assert(kJobRing >= 64 + 64 * kWideLeafTris && (kJobRing & (kJobRing - 1)) == 0, "ring size");

// Distances as unsigned keys that order like the floats.  NONNEG: every accepted distance is >= +0
// (the ray's near is not negative), so the bits themselves do.
template <bool NONNEG>
__device__ __forceinline__ unsigned dist_key(float t)
{
    const unsigned b = __float_as_uint(t);
    return NONNEG ? b : b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
}
template <bool NONNEG>
__device__ __forceinline__ float key_dist(unsigned k)
{
    return __uint_as_float(NONNEG ? k : k ^ ((unsigned)((int)~k >> 31) | 0x80000000u));
}

struct JobQueue {
    unsigned *ring;  // this wave's kJobRing job words
    unsigned *slots; // this wave's result slots: closest-hit {triangle id, distance key} per lane, any-hit one flag per lane
    unsigned head, tail; // free-running, wave-uniform: jobs [head, tail) are pending
};

template <bool ANYHIT, bool NONNEG>
__device__ __forceinline__ void job_slot_init(const JobQueue &jq, unsigned lane, float far_t)
{
    if (ANYHIT) {
        jq.slots[lane] = 0u;
    } else {
        jq.slots[2 * lane] = 0xFFFFFFFFu;
        jq.slots[2 * lane + 1] = dist_key<NONNEG>(far_t);
    }
}

// One triangle phase: the first min(pending, 64) jobs, one per lane.  job_issue() starts the
// triangle loads and job_finish() tests, so that a node step's loads can be in flight with them.
struct JobLoads {
    float4 t0, t1;
    float2 t2;
    unsigned owner, n;
};
__device__ __forceinline__ JobLoads job_issue(const float4 *__restrict__ items, const JobQueue &jq, unsigned lane)
{
    JobLoads j;
    const unsigned pending = jq.tail - jq.head;
    j.n = pending < 64u ? pending : 64u;
    const bool mine = lane < j.n;
    const unsigned job = mine ? jq.ring[(jq.head + lane) & (kJobRing - 1u)] : lane << kJobUnitBits;
    j.owner = job >> kJobUnitBits;
    const float4 *q = items + (job & ((1u << kJobUnitBits) - 1u));
    j.t0 = make_float4(0.f, 0.f, 0.f, 0.f);
    j.t1 = j.t0;
    j.t2 = make_float2(0.f, 0.f);
    if (mine) {
        j.t0 = q[0];
        j.t1 = q[1];
        j.t2 = *reinterpret_cast<const float2 *>(q + 2);
    }
    return j;
}
template <bool ANYHIT, bool NONNEG>
__device__ __forceinline__ void job_finish(const JobLoads &j, const LaneState &s, JobQueue &jq, unsigned lane)
{
    const unsigned n = j.n, owner = j.owner;
    const bool mine = lane < n;
    const float4 t0 = j.t0, t1 = j.t1;
    const float2 t2 = j.t2;
    // the owner's ray, from the owner's registers (every lane takes part: a lane without a job reads its own)
    const int a = (int)(owner << 2);
    const V3 o = v3(__int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(s.r.o.x))),
                    __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(s.r.o.y))),
                    __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(s.r.o.z))));
    const V3 d = v3(__int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(s.r.d.x))),
                    __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(s.r.d.y))),
                    __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(s.r.d.z))));
    const float near_t = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(s.r.near_t)));
    float far_t;
    bool skip = !mine;
    if (ANYHIT) {
        far_t = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(s.r.far_t)));
        skip = skip || jq.slots[owner] != 0u; // the owner's ray is already occluded
    } else {
        far_t = key_dist<NONNEG>(jq.slots[2 * owner + 1]);
    }
    if (!skip) {
        const float t = tri_test(t0, t1, t2, o, d);
        if (t >= near_t && t < far_t) {
            if (ANYHIT) {
                jq.slots[owner] = 1u;
            } else {
                // t + 0 turns an accepted -0 into +0, whose key orders
                const unsigned long long v =
                    ((unsigned long long)dist_key<NONNEG>(t + 0.f) << 32) | (unsigned long long)__float_as_uint(t2.y);
                atomicMin(reinterpret_cast<unsigned long long *>(jq.slots) + owner, v);
            }
        }
    }
    jq.head += n;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <bool ANYHIT, bool NONNEG>
__device__ __forceinline__ void job_phase(const float4 *__restrict__ items, const LaneState &s, JobQueue &jq, unsigned lane)
{
    const JobLoads j = job_issue(items, jq, lane);
    job_finish<ANYHIT, NONNEG>(j, s, jq, lane);
}

// s.tri holds the ray's job mark: the value of jq.tail after its last job was written; its jobs are
// done when jq.head has reached the mark.  On return s.r.far_t and s.hit of every lane without
// outstanding jobs are up to date.
template <bool ANYHIT, bool NONNEG, class STACK>
__device__ __forceinline__ void traverse_wide_queue(const SceneDev &sc, LaneState &s, STACK &st, JobQueue &jq,
                                                    unsigned min_active, unsigned max_steps)
{
    const float4 *__restrict__ items = reinterpret_cast<const float4 *>(sc.witems);
    RayRegs &r = s.r;
    const unsigned lane = lane_id();
    unsigned cur = s.cur, mark = s.tri, steps = s.steps;

    for (;;) {
        const bool has_leaf = cur != kDone && (cur & 3u) != 0; // kDone has its low bits set
        const bool walking = !(cur & 3u) && steps < max_steps;
        const unsigned nl = (unsigned)__popcll(__ballot(has_leaf));
        const unsigned nw = (unsigned)__popcll(__ballot(walking));
        const unsigned pending = jq.tail - jq.head;

        // what a triangle phase changes for the lane's own ray
        auto refresh = [&]() {
            if (ANYHIT) {
                if (s.have_ray && jq.slots[lane] != 0u) { // occluded: the walk is over, jobs still pending are skipped
                    s.hit = 1u;
                    cur = kDone;
                    st.sp = 0;
                }
            } else {
                r.far_t = key_dist<NONNEG>(jq.slots[2 * lane + 1]);
            }
        };
        if ((!RTC_Q_FUSE && pending >= 64u) || (pending != 0u && nl == 0u && nw < (unsigned)RTC_Q_NODE_MIN)) {
            job_phase<ANYHIT, NONNEG>(items, s, jq, lane);
            refresh();
            continue;
        }
        if (nl != 0u && (nl >= (unsigned)RTC_Q_LEAF_MIN || nw < (unsigned)RTC_Q_NODE_MIN)) {
            // one job per triangle of the leaves at hand, then on with the next deferred item
            const unsigned count = has_leaf ? (cur & 3u) : 0u;
#pragma unroll
            for (unsigned k = 0; k < kWideLeafTris; k++) {
                const bool p = count > k;
                const unsigned long long m = __ballot(p);
                if (p)
                    jq.ring[(jq.tail + prefix_count(m)) & (kJobRing - 1u)] =
                        (lane << kJobUnitBits) | ((cur >> 2) + kUnitsPerItem * k);
                jq.tail += (unsigned)__popcll(m);
            }
            if (has_leaf) {
                mark = jq.tail;
                cur = pop_next<ANYHIT>(st, r.far_t);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (nw == 0u) break; // nothing pending, no leaf at hand, nobody walking
        const bool busy = has_leaf || walking || (s.have_ray && (int)(jq.head - mark) < 0);
        if ((unsigned)__popcll(__ballot(busy)) < min_active) break;
        // a node step; with enough jobs pending their triangle loads are in flight beside the node loads
        const bool fuse = RTC_Q_FUSE && pending >= (unsigned)RTC_Q_FUSE_MIN;
        JobLoads jl;
        if (fuse) jl = job_issue(items, jq, lane);
        float4 a0, a1, a2;
        if (walking) {
            steps++;
            load_item(items + (cur >> 2), a0, a1, a2);
        }
        if (fuse) {
            job_finish<ANYHIT, NONNEG>(jl, s, jq, lane);
            refresh(); // the step below already sees what the jobs found
        }
        if (walking && cur != kDone) {
            if (qwide_step<ANYHIT>(a0, a1, a2, slab_ray_lean(s), r.near_t, r.far_t, st, cur))
                cur = pop_next<ANYHIT>(st, r.far_t);
        }
    }
    if (!ANYHIT) s.hit = jq.slots[2 * lane];
    s.cur = cur;
    s.tri = mark;
    s.steps = steps;
}

// The items of one iteration of traverse_wide_split(): a triangle (40 bytes used) for the lanes in
// tmask and a node for the lanes in nmask.  The six loads are always issued -- lanes without an
// item of a kind are masked off through exec, not by a branch -- and one wait covers them all.
__device__ __forceinline__ void load_tri_and_node(const float4 *pt, unsigned long long tmask, const float4 *pn,
                                                  unsigned long long nmask, float4 &t0, float4 &t1, float2 &t2,
                                                  float4 &n0, float4 &n1, float4 &n2)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %6, exec\n\t"
                 "s_mov_b64 exec, %9\n\t"
                 "global_load_dwordx4 %0, %7, off\n\t"
                 "global_load_dwordx4 %1, %7, off offset:16\n\t"
                 "global_load_dwordx2 %2, %7, off offset:32\n\t"
                 "s_mov_b64 exec, %10\n\t"
                 "global_load_dwordx4 %3, %8, off\n\t"
                 "global_load_dwordx4 %4, %8, off offset:16\n\t"
                 "global_load_dwordx4 %5, %8, off offset:32\n\t"
                 "s_mov_b64 exec, %6\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&s"(saved)
                 : "v"(pt), "v"(pn), "s"(tmask), "s"(nmask)
                 : "memory");
}

// Stepwise traversal with the node walk and the triangle tests of a ray side by side (form 6).
//
// In traverse_wide_steps() a lane is either at a node or inside a leaf, and since the lanes of a
// wave of incoherent rays are never all at the same kind of item every iteration runs the node
// code and the triangle code, each for the lanes that need it: on the San-Miguel-class bounce
// rays (about 30 nodes and 32 triangles per ray) 28 of 64 lanes are active per instruction.  Here
// a lane keeps BOTH going: when the walk reaches a leaf the leaf is handed to the lane's triangle
// cursor and the walk goes on with the next deferred node at once; an iteration loads and tests one
// triangle of the pending leaf and one node.  The instructions per iteration are the same, the
// iterations per ray drop from nodes + triangles towards the larger of the two.  The price: a node
// visited while a leaf is pending is tested against a far that the leaf's triangles may still shorten
// (a few wasted visits after the leaf that holds the hit), and the results of a node and a triangle
// live in registers at the same time.  The walk waits when it reaches another leaf before the
// triangle cursor is free.  Closest hits do not depend on the order (ids on exact-distance ties
// may, as between any two forms).
template <bool ANYHIT, class STACK>
__device__ __forceinline__ void traverse_wide_split(const SceneDev &sc, LaneState &s, STACK &st, unsigned min_active,
                                                    unsigned max_steps)
{
    const float4 *__restrict__ items = reinterpret_cast<const float4 *>(sc.witems);
    RayRegs &r = s.r;
    // cur: the walk's item (a node, or a leaf waiting for the triangle cursor, or kDone);
    // leaf: unit << 2 | triangles still to test of the pending leaf (0 = none)
    unsigned cur = s.cur, leaf = s.tri;
    unsigned steps = s.steps;

    // past the step budget the walk stands still and only the pending leaf is finished, so that a
    // ray is handed to the wave-cooperative finisher with nothing but `cur` and its stack
    while ((leaf & 3u) || (cur != kDone && steps < max_steps)) {
        const bool budget = steps < max_steps;
        steps++;
        const bool do_tri = (leaf & 3u) != 0;
        const bool do_node = budget && !(cur & 3u); // kDone has its low bits set
        float4 t0, t1, n0, n1, n2;
        float2 t2;
        load_tri_and_node(items + (leaf >> 2), __ballot(do_tri), items + (cur >> 2), __ballot(do_node), t0, t1, t2, n0,
                          n1, n2);
        if (do_tri) {
            const float t = tri_test(t0, t1, t2, r.o, r.d);
            leaf += (kUnitsPerItem << 2) - 1u; // the next triangle, one fewer to go
            if (t >= r.near_t && t < r.far_t) {
                if (ANYHIT) {
                    s.hit = 1u;
                    cur = kDone;
                    leaf = 0;
                } else {
                    r.far_t = t;
                    s.hit = __float_as_uint(t2.y);
                }
            }
        }
        bool need_pop = false;
        if (do_node && (!ANYHIT || cur != kDone))
            need_pop = qwide_step<ANYHIT>(n0, n1, n2, slab_ray_lean(s), r.near_t, r.far_t, st, cur);
        // a leaf reached by the walk goes to the triangle cursor as soon as that is free
        if (budget && !need_pop && cur != kDone && (cur & 3u) && !(leaf & 3u)) {
            leaf = cur;
            need_pop = true;
        }
        if (need_pop) cur = pop_next<ANYHIT>(st, r.far_t);
        if ((unsigned)__popcll(__ballot(true)) < min_active) break;
    }
    s.cur = cur;
    s.tri = leaf;
    s.steps = steps;
}

// While-while form on the quantized 4-wide tree (coherent rays): all lanes walk
// wide nodes until each has reached a leaf or finished, then the leaves are
// tested; see traverse() for the loop and traverse_wide_steps() for the node.
template <bool ANYHIT, class STACK>
__device__ __forceinline__ void traverse_wide(const SceneDev &sc, LaneState &s, STACK &st, unsigned min_active,
                                              unsigned max_steps)
{
    const float4 *__restrict__ items = reinterpret_cast<const float4 *>(sc.witems);
    RayRegs &r = s.r;
    unsigned cur = s.cur;
    unsigned steps = s.steps;

    while (cur != kDone && steps < max_steps) {
        // the early exit below is relative to the lanes that enter the node loop (RTC_WW_NODE_MIN of 64): with
        // an absolute count a wave that enters with fewer lanes (launch tail, small frames, just before a
        // refill) would leave the loop after every single node
#if RTC_WW_NODE_RELATIVE
        const unsigned node_min = ((unsigned)__popcll(__ballot(!(cur & 3u))) * (unsigned)RTC_WW_NODE_MIN) >> 6;
#else
        const unsigned node_min = (unsigned)RTC_WW_NODE_MIN;
#endif
        while (!(cur & 3u) && steps < max_steps) { // kDone has its low bits set
            steps++;
#if RTC_PROBE_VALU
            {   // measurement builds only: RTC_PROBE_VALU more vector instructions per node iteration
                float x = r.near_t;
#pragma unroll
                for (int k = 0; k < RTC_PROBE_VALU; k++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
                asm volatile("" ::"v"(x));
            }
#endif
            const unsigned ucur = (unsigned)__builtin_amdgcn_readfirstlane((int)cur);
            bool pop;
            if (__ballot(cur != ucur) == 0) { // every lane still in this loop is at node ucur
                const u32x16 n = scalar_load64(items + (ucur >> 2));
                pop = qwide_step<ANYHIT>(f4_of(n, 0), f4_of(n, 4), f4_of(n, 8), slab_ray_lean(s), r.near_t, r.far_t, st, cur);
            } else {
                const float4 *q = items + (cur >> 2);
                const float4 a0 = q[0];
                const float4 a1 = q[1];
                const float4 a2 = q[2];
                pop = qwide_step<ANYHIT>(a0, a1, a2, slab_ray_lean(s), r.near_t, r.far_t, st, cur);
            }
            if (pop) cur = pop_next<ANYHIT>(st, r.far_t);
            // the last few lanes still descending do not hold up the lanes that wait at their leaves
            if (RTC_WW_NODE_MIN > 1 && (unsigned)__popcll(__ballot(!(cur & 3u))) < node_min) break;
        }
        if (cur != kDone && (cur & 3u)) {
            steps++;
            const unsigned first = cur >> 2, count = cur & 3u;
            bool occluded = false;
            const unsigned ucur = (unsigned)__builtin_amdgcn_readfirstlane((int)cur);
            if (__ballot(cur != ucur) == 0) { // one leaf for the whole wave: triangles through the scalar cache
                const unsigned ufirst = ucur >> 2, ucount = ucur & 3u;
                for (unsigned i = 0; i < ucount; i++) {
                    const u32x16 n = scalar_load64(items + ufirst + kUnitsPerItem * i); // 48 bytes used
                    const float t = tri_test(f4_of(n, 0), f4_of(n, 4),
                                             make_float2(__uint_as_float(n[8]), __uint_as_float(n[9])), r.o, r.d);
                    if (!occluded && t >= r.near_t && t < r.far_t) {
                        if (ANYHIT) {
                            occluded = true;
                        } else {
                            r.far_t = t;
                            s.hit = n[9];
                        }
                    }
                    if (ANYHIT && __ballot(!occluded) == 0) break;
                }
            } else {
                for (unsigned i = 0; i < count; i++) {
                    const float4 *q = items + first + kUnitsPerItem * i;
                    const float4 t0 = q[0];
                    const float4 t1 = q[1];
                    const float2 t2 = *reinterpret_cast<const float2 *>(q + 2);
                    const float t = tri_test(t0, t1, t2, r.o, r.d);
                    if (t >= r.near_t && t < r.far_t) {
                        if (ANYHIT) {
                            occluded = true;
                            break;
                        }
                        r.far_t = t;
                        s.hit = __float_as_uint(t2.y);
                    }
                }
            }
            if (ANYHIT && occluded) {
                s.hit = 1u;
                cur = kDone;
            } else {
                cur = pop_next<ANYHIT>(st, r.far_t);
            }
        }
        if ((unsigned)__popcll(__ballot(true)) < min_active) break;
    }
    s.cur = cur;
    s.steps = steps;
}

// Cooperative finish of ONE ray of a wide form by the whole wave: the same scheme as coop_finish()
// below on the wide tree itself.  An entry of the work stack is a traversal reference with the
// entry distance of its box; a node entry tests its four children and pushes the hit ones,
// farthest first; a leaf entry tests its triangles.
struct CoopItem {
    unsigned ref;
    float tnear;
};

template <bool ANYHIT, class STATE, int BLOCK, int LDS_DEPTH, bool DIST>
__device__ __forceinline__ void coop_finish_wide(const SceneDev &sc, STATE &s, Stack<BLOCK, LDS_DEPTH, DIST> &st,
                                                 unsigned p, CoopItem *work, unsigned cap)
{
    const float4 *__restrict__ items = reinterpret_cast<const float4 *>(sc.witems);
    const unsigned lane = lane_id();
    // the ray of lane p, in every lane
    struct {
        RayRegs r;
    } ray;
    ray.r.o = v3(__shfl(s.r.o.x, p), __shfl(s.r.o.y, p), __shfl(s.r.o.z, p));
    ray.r.d = v3(__shfl(s.r.d.x, p), __shfl(s.r.d.y, p), __shfl(s.r.d.z, p));
    ray.r.rd = v3(__shfl(s.r.rd.x, p), __shfl(s.r.rd.y, p), __shfl(s.r.rd.z, p));
    ray.r.near_t = __shfl(s.r.near_t, p);
    const SlabRay sray = slab_ray(ray.r); // o * (1/d) rounded as lane_begin rounds it
    float far_t = __shfl(s.r.far_t, p);
    unsigned hit = (unsigned)__shfl((int)s.hit, p);

    // seed: lane p's stack entries (lane e reads entry e) and its current item
    const int sp_p = __shfl(st.sp, p);
    const unsigned cur_p = (unsigned)__shfl((int)s.cur, p);
    unsigned count = 0; // wave-uniform
    {
        unsigned *lds_p = st.lds - (threadIdx.x & 63) + p; // column of lane p
        const unsigned col_p = st.ovf_lane - (threadIdx.x & 63) + p;
        for (int e = (int)lane; e < sp_p; e += 64) {
            const unsigned ref =
                e < LDS_DEPTH ? lds_p[e * BLOCK] : st.ovf_base[(size_t)(e - LDS_DEPTH) * st.ovf_stride + col_p];
            work[e] = CoopItem{ref, -__builtin_inff()};
        }
        count = (unsigned)sp_p;
        if (lane == 0) work[count] = CoopItem{cur_p, -__builtin_inff()};
        count += 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // Stack bound.  Popping m entries pushes at most 4m, so a batch is sized to keep
    // count <= lim = cap - reserve; at the limit one entry is popped per iteration, a plain
    // depth-first walk that grows the stack by at most three entries per level.
    const unsigned lim = cap - (3u * sc.wmax_depth + 8u);
    bool found = false; // any-hit result (wave-uniform)
    while (count != 0 && !(ANYHIT && found)) {
        unsigned m = count < 64u ? count : 64u;
        if (count + 3u * m > lim) {
            m = lim > count ? (lim - count) / 3u : 0u;
            if (m == 0) m = 1;
        }
        CoopItem it{kDone, 0.f};
        if (lane < m) it = work[count - 1u - lane];
        count -= m;
        const bool active = lane < m && it.tnear <= far_t;

        float best_t = __builtin_inff();
        unsigned best_id = 0xFFFFFFFFu;
        unsigned key[4] = {kMissKey, kMissKey, kMissKey, kMissKey}; // children to push, nearest in key[0]
        float kn[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned base4 = 0, meta = 0;
        if (active && !(it.ref & 3u)) {
            const float4 *q = items + (it.ref >> 2);
            const float4 a0 = q[0], a1 = q[1], a2 = q[2];
            float f[4];
            qnode_slabs(a0, a1, a2, sray, ray.r.near_t, far_t, kn, f);
            base4 = __float_as_uint(a1.x);
            meta = __float_as_uint(a1.y);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (kn[i] <= f[i]) key[i] = (__float_as_uint(kn[i]) & ~3u) | (unsigned)i;
            RTC_CE(key[0], key[1]);
            RTC_CE(key[2], key[3]);
            RTC_CE(key[0], key[2]);
            RTC_CE(key[1], key[3]);
            RTC_CE(key[1], key[2]);
        } else if (active) {
            const unsigned first = it.ref >> 2, n = it.ref & 3u;
            float lane_far = far_t;
            for (unsigned i = 0; i < n; i++) {
                const float4 *q = items + first + kUnitsPerItem * i;
                const float4 t0 = q[0];
                const float4 t1 = q[1];
                const float2 t2 = *reinterpret_cast<const float2 *>(q + 2);
                const float t = tri_test(t0, t1, t2, ray.r.o, ray.r.d);
                if (t >= ray.r.near_t && t < lane_far) {
                    lane_far = t;
                    best_t = t;
                    best_id = __float_as_uint(t2.y);
                    if (ANYHIT) break;
                }
            }
        }

        // combine the lanes' hits
        if (ANYHIT) {
            found = __ballot(best_t < __builtin_inff()) != 0;
        } else {
            float wt = best_t;
            unsigned wi = best_id;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ot = __shfl_xor(wt, off);
                const unsigned oi = (unsigned)__shfl_xor((int)wi, off);
                if (ot < wt || (ot == wt && oi < wi)) {
                    wt = ot;
                    wi = oi;
                }
            }
            if (wt < far_t) {
                far_t = wt;
                hit = wi;
            }
        }

        // push the children: every lane's farthest first, its nearest last (on top)
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const bool push = key[j] != kMissKey;
            const unsigned long long mask = __ballot(push);
            if (push) {
                // the key holds the entry distance rounded down to a multiple of 4 ulp: conservative for
                // the cull at pop time
                work[count + prefix_count(mask)] =
                    CoopItem{child_ref(base4, meta, key[j] & 3u), __uint_as_float(key[j] & ~3u)};
            }
            count += (unsigned)__popcll(mask);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    if (lane == p) {
        if (ANYHIT) {
            s.hit = found ? 1u : s.hit;
        } else {
            s.hit = hit;
            s.r.far_t = far_t;
        }
        s.cur = kDone;
        s.tri = s.tri_end = 0;
        st.sp = 0;
    }
}
#undef RTC_CE

// Cooperative finish of ONE ray by the whole wave.
//
// A few rays need 10^4..10^5 traversal steps (grazing rays through dense
// foliage); one lane walking them at one dependent memory access per step takes
// tens of milliseconds and becomes the tail of the launch.  Such a ray (lane
// `p`, parked at a node boundary with `steps >= max_steps`) is finished here
// with 64 lanes: the pending work -- the lane's current node and its stack -- is
// moved to a wave-private stack in global memory, and every iteration up to 64
// entries are popped and processed in parallel (an inner node tests both
// children and pushes the hit ones with their entry distance; a leaf tests its
// triangles), with the closest distance reduced across the wave after every
// iteration and entries beyond it discarded at pop time.
//
// The closest distance found is exactly the single-lane result (the same
// triangle arithmetic on a superset of candidates); on an exact-distance tie
// the id may be another triangle of the tie, and visit counters of such a ray
// differ (other visit order, pop-time culling).
template <bool ANYHIT, bool STATS, int BLOCK, int LDS_DEPTH, bool DIST>
__device__ __forceinline__ void coop_finish(const SceneDev &sc, LaneState &s, Stack<BLOCK, LDS_DEPTH, DIST> &st, Counters &cnt,
                                         unsigned p, CoopItem *work, unsigned cap)
{
    const float4 *__restrict__ nodes = reinterpret_cast<const float4 *>(sc.nodes);
    const float4 *__restrict__ tris = reinterpret_cast<const float4 *>(sc.tris);
    const unsigned lane = lane_id();

    // the ray of lane p, in every lane
    const V3 o = v3(__shfl(s.r.o.x, p), __shfl(s.r.o.y, p), __shfl(s.r.o.z, p));
    const V3 d = v3(__shfl(s.r.d.x, p), __shfl(s.r.d.y, p), __shfl(s.r.d.z, p));
    const V3 rd = v3(__shfl(s.r.rd.x, p), __shfl(s.r.rd.y, p), __shfl(s.r.rd.z, p));
    const float near_t = __shfl(s.r.near_t, p);
    float far_t = __shfl(s.r.far_t, p);
    unsigned hit = (unsigned)__shfl((int)s.hit, p);
    const float ordx = __shfl(s.ordx, p), ordy = __shfl(s.ordy, p), ordz = __shfl(s.ordz, p);
    const bool negx = (__float_as_uint(rd.x) >> 31) != 0;
    const bool negy = (__float_as_uint(rd.y) >> 31) != 0;
    const bool negz = (__float_as_uint(rd.z) >> 31) != 0;

    // seed: lane p's stack entries (lane e reads entry e) and its current node
    const int sp_p = __shfl(st.sp, p);
    const unsigned cur_p = (unsigned)__shfl((int)s.cur, p);
    unsigned count = 0; // wave-uniform
    {
        unsigned *lds_p = st.lds - (threadIdx.x & 63) + p;            // column of lane p
        const unsigned col_p = st.ovf_lane - (threadIdx.x & 63) + p;
        for (int e = (int)lane; e < sp_p; e += 64) {
            unsigned ref;
            if (e < LDS_DEPTH)
                ref = lds_p[e * BLOCK];
            else
                ref = st.ovf_base[(size_t)(e - LDS_DEPTH) * st.ovf_stride + col_p];
            work[e] = CoopItem{ref, -__builtin_inff()};
        }
        count = (unsigned)sp_p;
        if (lane == 0) {
            work[count] = CoopItem{cur_p, -__builtin_inff()};
        }
        count += 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // Stack bound.  Popping m entries pushes at most 2m, so a batch is sized to keep
    // count <= lim = cap - reserve; at the limit one entry is popped per iteration,
    // which is a plain depth-first walk and grows the stack by at most the tree depth
    // (reserve = max_depth + 2).  The host guarantees cap > reserve + 64 + max_depth.
    const unsigned lim = cap - (sc.max_depth + 2u);
    bool found = false; // any-hit result (wave-uniform)
    while (count != 0 && !(ANYHIT && found)) {
        unsigned m = count < 64u ? count : 64u;
        if (count + m > lim) m = lim > count ? lim - count : 1u;
        CoopItem it{kDone, 0.f};
        if (lane < m) it = work[count - 1u - lane];
        count -= m;
        bool active = lane < m && it.tnear <= far_t;

        // results of this lane for this iteration
        float best_t = __builtin_inff();
        unsigned best_id = 0xFFFFFFFFu;
        bool push_a = false, push_b = false; // a: pushed first (deeper in the stack), b: on top
        CoopItem ca{kDone, 0.f}, cb{kDone, 0.f};

        if (active && !(it.ref & kLeafBit)) {
            const float4 bx = nodes[(size_t)it.ref * 4 + 0];
            const float4 by = nodes[(size_t)it.ref * 4 + 1];
            const float4 bz = nodes[(size_t)it.ref * 4 + 2];
            const float2 lk = *reinterpret_cast<const float2 *>(nodes + (size_t)it.ref * 4 + 3);
            if (STATS) cnt.inner++;
            const float nxl = negx ? bx.z : bx.x, nxr = negx ? bx.w : bx.y;
            const float fxl = negx ? bx.x : bx.z, fxr = negx ? bx.y : bx.w;
            const float nyl = negy ? by.z : by.x, nyr = negy ? by.w : by.y;
            const float fyl = negy ? by.x : by.z, fyr = negy ? by.y : by.w;
            const float nzl = negz ? bz.z : bz.x, nzr = negz ? bz.w : bz.y;
            const float fzl = negz ? bz.x : bz.z, fzr = negz ? bz.y : bz.w;
            const float tnxl = __fmaf_rn(nxl, rd.x, -ordx), tnxr = __fmaf_rn(nxr, rd.x, -ordx);
            const float tfxl = __fmaf_rn(fxl, rd.x, -ordx), tfxr = __fmaf_rn(fxr, rd.x, -ordx);
            const float tnyl = __fmaf_rn(nyl, rd.y, -ordy), tnyr = __fmaf_rn(nyr, rd.y, -ordy);
            const float tfyl = __fmaf_rn(fyl, rd.y, -ordy), tfyr = __fmaf_rn(fyr, rd.y, -ordy);
            const float tnzl = __fmaf_rn(nzl, rd.z, -ordz), tnzr = __fmaf_rn(nzr, rd.z, -ordz);
            const float tfzl = __fmaf_rn(fzl, rd.z, -ordz), tfzr = __fmaf_rn(fzr, rd.z, -ordz);
            const float nearl = fmaxf(fmaxf(fmaxf(tnxl, tnyl), tnzl), near_t);
            const float nearr = fmaxf(fmaxf(fmaxf(tnxr, tnyr), tnzr), near_t);
            const float farl = fminf(fminf(fminf(tfxl, tfyl), tfzl), far_t);
            const float farr = fminf(fminf(fminf(tfxr, tfyr), tfzr), far_t);
            const bool hl = nearl <= farl;
            const bool hr = nearr <= farr;
            const unsigned left = __float_as_uint(lk.x), right = __float_as_uint(lk.y);
            const bool left_first = nearl <= nearr;
            // the nearer child goes on top (b), the farther one below (a)
            const CoopItem cl{left, nearl}, cr{right, nearr};
            if (hl && hr) {
                push_a = push_b = true;
                ca = left_first ? cr : cl;
                cb = left_first ? cl : cr;
            } else if (hl || hr) {
                push_b = true;
                cb = hl ? cl : cr;
            }
        } else if (active) {
            unsigned first, n;
            if (it.ref & kBigLeafBit) {
                const BigLeaf bl = sc.big_leaves[it.ref & 0x3FFFFFFFu];
                first = bl.first;
                n = bl.count;
            } else {
                first = (it.ref >> 3) & (kMaxPackedFirst - 1u);
                n = (it.ref & 7u) + 1u;
            }
            if (STATS) {
                cnt.leaf++;
                cnt.tris += n;
            }
            float lane_far = far_t;
            for (unsigned i = 0; i < n; i++) {
                const size_t base = (size_t)(first + i) * 3;
                const float4 t0 = tris[base + 0];
                const float4 t1 = tris[base + 1];
                const float2 t2 = *reinterpret_cast<const float2 *>(tris + base + 2); // v0v2.z, id (8 of 16 B)
                const float t = tri_test(t0, t1, t2, o, d);
                if (t >= near_t && t < lane_far) {
                    lane_far = t;
                    best_t = t;
                    best_id = __float_as_uint(t2.y);
                    if (ANYHIT) break;
                }
            }
        }

        // combine the lanes' hits
        if (ANYHIT) {
            found = __ballot(best_t < __builtin_inff()) != 0;
        } else {
            float wt = best_t;
            unsigned wi = best_id;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ot = __shfl_xor(wt, off);
                const unsigned oi = (unsigned)__shfl_xor((int)wi, off);
                if (ot < wt || (ot == wt && oi < wi)) {
                    wt = ot;
                    wi = oi;
                }
            }
            if (wt < far_t) {
                far_t = wt;
                hit = wi;
            }
        }

        // push the children: all "a" entries first, then all "b" entries
        const unsigned long long ma = __ballot(push_a), mb = __ballot(push_b);
        const unsigned na = (unsigned)__popcll(ma), nb = (unsigned)__popcll(mb);
        if (push_a) work[count + prefix_count(ma)] = ca;
        if (push_b) work[count + na + prefix_count(mb)] = cb;
        count += na + nb;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    if (lane == p) {
        if (ANYHIT) {
            s.hit = found ? 1u : s.hit;
        } else {
            s.hit = hit;
            s.r.far_t = far_t;
        }
        s.cur = kDone;
        s.tri = s.tri_end = 0;
        st.sp = 0;
    }
}

template <bool STATS>
__device__ __forceinline__ void flush_stats(const Counters &cnt, unsigned long long rays, StatsDev *stats, unsigned depth)
{
    if (!STATS) return;
    unsigned long long a = wave_sum(cnt.inner), b = wave_sum(cnt.leaf), c = wave_sum(cnt.tris),
                       d = wave_sum(rays);
    unsigned long long m = cnt.max_steps, ms = cnt.max_sp;
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(m, off);
        m = o > m ? o : m;
        const unsigned long long os = __shfl_down(ms, off);
        ms = os > ms ? os : ms;
    }
    if (lane_id() == 0) {
        atomicMax(&stats->max_ray_steps, m);
        atomicMax(&stats->max_stack_entries, ms);
        atomicAdd(&stats->inner_rays, a);
        atomicAdd(&stats->leaf_rays, b);
        atomicAdd(&stats->obj_intersect, c);
        atomicAdd(&stats->total_rays, d);
        const unsigned slot = depth < kStatDepths ? depth : kStatDepths - 1u;
        atomicAdd(&stats->depth_inner[slot], a);
        atomicAdd(&stats->depth_tris[slot], c);
    }
}

constexpr int kBlock = RTC_TRACE_BLOCK;
constexpr int kLdsDepth = RTC_LDS_STACK_DEPTH;

// ------------------------------------------------------------------------
// persistent trace kernel
//
// The grid is sized to the chip (resident blocks only).  Each wave owns 64
// lane states and repeatedly (1) fetches rays for its idle lanes with ONE
// atomic per refill (ballot + popcount + prefix rank), (2) traverses until fewer
// than `min_active` lanes are busy.  To keep the fetch counter off the critical
// path the queue is cut into kFetchRanges ranges with one counter each, 128
// bytes apart; a wave starts on range (wave id mod kFetchRanges) and moves to
// the next when its range is exhausted, so all ranges drain together and the
// tail is at most one ray per lane.


// IO policies: how rays are read and results written.

struct StreamIO { // rt_stream_ray arrays, the reference's StreamRay boundary
    static constexpr bool kNearNonNegative = false; // callers' rays: any near
    rt_stream_ray_dev *rays;
    unsigned *hits;
    unsigned char *occluded;
    __device__ __forceinline__ RayRegs load(unsigned i) const
    {
        const float4 of = rays[i].origin_far;
        const float4 rn = rays[i].dir_recip_near;
        const float4 dd = rays[i].direction;
        RayRegs r;
        r.o = v3(of.x, of.y, of.z);
        r.far_t = of.w;
        r.rd = v3(rn.x, rn.y, rn.z);
        r.near_t = rn.w;
        r.d = v3(dd.x, dd.y, dd.z);
        return r;
    }
    template <bool ANYHIT>
    __device__ __forceinline__ void store(const LaneState &s) const
    {
        if (ANYHIT) {
            occluded[s.index] = (unsigned char)(s.hit == 1u);
        } else {
            hits[s.index] = s.hit;
            rays[s.index].origin_far.w = s.r.far_t;
        }
    }
};

struct QueueIO { // render pipeline queues
    static constexpr bool kNearNonNegative = true; // the shade kernel's rays start at SHADOW_RAY_NEAR
    QRay *rays;
    unsigned *hit_ids;         // extend
    const ShadowInfo *infos;   // connect
    float *rad_r, *rad_g, *rad_b;
    __device__ __forceinline__ RayRegs load(unsigned i) const
    {
        const float4 *rq = reinterpret_cast<const float4 *>(rays);
        const float4 a = rq[(size_t)i * 2 + 0];
        const float4 b = rq[(size_t)i * 2 + 1];
        RayRegs r;
        r.o = v3(a.x, a.y, a.z);
        r.near_t = a.w;
        r.d = v3(b.x, b.y, b.z);
        r.far_t = b.w;
        r.rd = v3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z); // src/ray.rs:18
        return r;
    }
    template <bool ANYHIT>
    __device__ __forceinline__ void store(const LaneState &s) const
    {
        if (ANYHIT) {
            // an un-occluded shadow ray adds its weight to the radiance slot of its
            // path (src/integrators/integrate_stream.rs:152-161).  A path has at
            // most one shadow ray per bounce and launches are stream ordered, so
            // the read-modify-write needs no atomic.
            if (s.hit != 1u) {
                const float4 w = reinterpret_cast<const float4 *>(infos)[s.index];
                const unsigned slot = __float_as_uint(w.w);
                rad_r[slot] += w.x;
                rad_g[slot] += w.y;
                rad_b[slot] += w.z;
            }
        } else {
            hit_ids[s.index] = s.hit;
            rays[s.index].far_t = s.r.far_t;
        }
    }
};

struct PrimaryIO; // depth-0 closest-hit launches that make their own camera rays (defined below the camera code)

// Resident waves per SIMD the compiler allocates registers for, and LDS stack entries per lane, by
// kernel (measured side by side on one GPU).  Every form is partly latency bound -- an iteration ends in a wait for
// the slowest gather of the wave -- so each kernel runs at the highest occupancy its loop reaches
// without spilling inside the loop: depth-0 extend 6 waves, depth-0 connect (while-while any-hit) 7
// with 16 LDS entries, stepwise extend on the wide tree 6 (5 before the ray state was trimmed: +2 %
// on the San-Miguel-class frame; at 7 its spills cost 25 %), stepwise any-hit 7 with 16 LDS entries.
template <bool ANYHIT, bool STATS, int MODE, class IO>
struct TraceTune {
    static constexpr int waves = RTC_TRACE_WAVES_PER_SIMD;
    static constexpr int lds_depth = kLdsDepth;
    static constexpr bool stack_dist = false; // entry distances beside the LDS stack entries (closest-hit, wide tree)
};
#if RTC_TRACE_TUNED
template <bool ANYHIT, class IO> // exact forms without visit counters
struct TraceTune<ANYHIT, false, 0, IO> {
    static constexpr int waves = RTC_TUNE_EXACT0_WAVES;
    static constexpr int lds_depth = kLdsDepth;
    static constexpr bool stack_dist = false;
};
template <bool ANYHIT, class IO>
struct TraceTune<ANYHIT, false, 1, IO> {
    static constexpr int waves = RTC_TUNE_EXACT1_WAVES;
    static constexpr int lds_depth = kLdsDepth;
    static constexpr bool stack_dist = false;
};
template <>
struct TraceTune<false, false, 3, PrimaryIO> {
    static constexpr int waves = RTC_TUNE_PRIMARY_WAVES;
    static constexpr int lds_depth = RTC_TUNE_PRIMARY_LDS;
    static constexpr bool stack_dist = RTC_TUNE_PRIMARY_DIST != 0;
};
template <>
struct TraceTune<true, false, 3, QueueIO> {
    static constexpr int waves = RTC_TUNE_ANY3_WAVES;
    static constexpr int lds_depth = RTC_TUNE_ANY3_LDS;
    static constexpr bool stack_dist = false;
};
template <>
struct TraceTune<true, false, 2, QueueIO> {
    static constexpr int waves = RTC_TUNE_ANY2_WAVES;
    static constexpr int lds_depth = RTC_TUNE_ANY2_LDS;
    static constexpr bool stack_dist = false;
};
template <>
struct TraceTune<false, false, 2, QueueIO> {
    static constexpr int waves = RTC_TUNE_EXT2_WAVES;
    static constexpr int lds_depth = RTC_TUNE_EXT2_LDS;
    static constexpr bool stack_dist = RTC_TUNE_EXT2_DIST != 0;
};
template <>
struct TraceTune<false, false, 6, QueueIO> {
    static constexpr int waves = RTC_TUNE_EXT6_WAVES;
    static constexpr int lds_depth = RTC_TUNE_EXT6_LDS;
    static constexpr bool stack_dist = false;
};
template <>
struct TraceTune<true, false, 6, QueueIO> {
    static constexpr int waves = RTC_TUNE_ANY6_WAVES;
    static constexpr int lds_depth = RTC_TUNE_ANY6_LDS;
    static constexpr bool stack_dist = false;
};
template <>
struct TraceTune<false, false, 8, QueueIO> {
    static constexpr int waves = RTC_TUNE_EXT8_WAVES;
    static constexpr int lds_depth = RTC_TUNE_EXT8_LDS;
    static constexpr bool stack_dist = false;
};
template <>
struct TraceTune<true, false, 8, QueueIO> {
    static constexpr int waves = RTC_TUNE_ANY8_WAVES;
    static constexpr int lds_depth = RTC_TUNE_ANY8_LDS;
    static constexpr bool stack_dist = false;
};
// form 7 keeps a job ring (4 KB per block) and result slots (2 KB closest-hit, 1 KB any-hit) in LDS
// beside the stacks: 19 + 1 + 4 + 2 = 26 KB per block for six blocks per CU, 16 + 1 + 4 + 1 = 22 KB for seven
template <class IO>
struct TraceTune<false, false, 7, IO> {
    static constexpr int waves = RTC_TUNE_EXT7_WAVES;
    static constexpr int lds_depth = RTC_TUNE_EXT7_LDS;
    static constexpr bool stack_dist = false;
};
template <class IO>
struct TraceTune<true, false, 7, IO> {
    static constexpr int waves = RTC_TUNE_ANY7_WAVES;
    static constexpr int lds_depth = RTC_TUNE_ANY7_LDS;
    static constexpr bool stack_dist = false;
};
#endif

// The allocation-free stream entry point sizes its fixed overflow area from RTC_LDS_STACK_DEPTH entries
// in LDS: every closest-hit kernel it can launch (forms 0..4, 6 and 8 on StreamIO) must keep that many.
static_assert(TraceTune<false, false, 0, StreamIO>::lds_depth == kLdsDepth &&
                  TraceTune<false, false, 1, StreamIO>::lds_depth == kLdsDepth &&
                  TraceTune<false, false, 2, StreamIO>::lds_depth == kLdsDepth &&
                  TraceTune<false, false, 3, StreamIO>::lds_depth == kLdsDepth &&
                  TraceTune<false, false, 4, StreamIO>::lds_depth == kLdsDepth &&
                  TraceTune<false, false, 6, StreamIO>::lds_depth == kLdsDepth &&
                  TraceTune<false, false, 8, StreamIO>::lds_depth == kLdsDepth,
              "a StreamIO closest-hit kernel keeps fewer LDS stack entries than the fixed scratch assumes");

template <bool ANYHIT, bool STATS, int MODE, class IO>
__global__ __launch_bounds__(kBlock, (TraceTune<ANYHIT, STATS, MODE, IO>::waves)) void trace_kernel(SceneDev sc, IO io, const unsigned *n_ptr, unsigned n_val,
                                                      TraceLaunch tl)
{
    constexpr int kDepth = TraceTune<ANYHIT, STATS, MODE, IO>::lds_depth;
    constexpr bool kWide = MODE == 2 || MODE == 3 || MODE == 6 || MODE == 7 || MODE == 8; // forms on the quantized 4-wide tree
    constexpr bool kQueue = MODE == 7;                                       // triangle jobs through a wave-level ring
    constexpr bool kNonNeg = IO::kNearNonNegative;
    constexpr bool kDist = TraceTune<ANYHIT, STATS, MODE, IO>::stack_dist && kWide && !ANYHIT;
    __shared__ unsigned lds_stack[Stack<kBlock, kDepth, kDist>::kLdsWords * kBlock];
    __shared__ unsigned lds_index[kBlock]; // queue position of the lane's ray: needed again only when it retires
    __shared__ unsigned lds_jobs[kQueue ? kJobRing * (kBlock / 64) : 1];
    __shared__ unsigned long long lds_slots[kQueue ? (ANYHIT ? kBlock / 2 : kBlock) : 1];
    const unsigned n = n_ptr ? *n_ptr : n_val;
    const unsigned gtid = blockIdx.x * kBlock + threadIdx.x;
    Stack<kBlock, kDepth, kDist> st;
    st.lds = lds_stack + threadIdx.x;
    st.ovf_base = tl.ovf;
    st.ovf_lane = gtid;
    st.ovf_stride = gridDim.x * kBlock;
    st.sp = 0;
    Counters cnt{0, 0, 0, 0, 0, 0};
    unsigned long long nrays = 0;

    const unsigned range_len = (n + kFetchRanges - 1u) / kFetchRanges;
    unsigned range = (gtid >> 6) % kFetchRanges; // wave-uniform
    unsigned ranges_left = kFetchRanges;         // ranges not yet seen empty by this wave
    const unsigned lane = lane_id();

    LaneState s;
    s.cur = kDone;
    s.tri = s.tri_end = 0; // no leaf pending: form 6 walks a lane for as long as it has one
    s.have_ray = false;
    JobQueue jq;
    jq.ring = lds_jobs + (kQueue ? (threadIdx.x >> 6) * kJobRing : 0u);
    jq.slots = reinterpret_cast<unsigned *>(lds_slots) + (kQueue ? (threadIdx.x >> 6) * (ANYHIT ? 64u : 128u) : 0u);
    jq.head = jq.tail = 0;

    while (true) {
        // ---- refill idle lanes
        unsigned long long idle_mask = __ballot(!s.have_ray);
        while (ranges_left != 0 && idle_mask != 0) {
            const unsigned need = (unsigned)__popcll(idle_mask);
            const unsigned rank = prefix_count(idle_mask);
            const unsigned leader = (unsigned)__ffsll((long long)idle_mask) - 1u;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&tl.fetch[range * kFetchStride], need);
            base = (unsigned)__shfl((int)base, (int)leader);
            const unsigned begin = range * range_len;
            const unsigned end = (begin + range_len < n) ? begin + range_len : n; // begin may exceed n: empty
            const unsigned idx = begin + base + rank;
            if (!s.have_ray && begin + base < end && idx < end) {
                unsigned ray = idx;
                if (tl.perm && idx < tl.perm_n) { // the first perm_n queue entries are ordered, the rest are not
                    ray = tl.perm[idx];
                    if (ray >= n) ray = idx; // never index outside the queue, whatever the permutation holds
                }
                RayRegs loaded = io.load(ray);
                if (kWide) {
                    // An exactly axis-parallel ray (cosine sampling returns the surface normal once
                    // in 2^24 draws, and floors and walls are axis aligned) has 1/d = +-inf, and
                    // plane*inf - o*inf is +-inf or NaN by the signs of plane and o: with far still
                    // infinite almost every box passes and one such ray walks millions of nodes (the
                    // reference does the same on a CPU thread, where it goes unnoticed).  The wide
                    // forms bound the reciprocal the way the reference's packet rays do
                    // (src/simd_ray.rs:46-47): |d| < 1e-24 counts as 1e-24, which makes the slab
                    // test of such an axis a proper inside/outside test.  Triangle tests use d itself.
                    const float tiny = 1e-24f;
                    const float dx = fabsf(loaded.d.x) < tiny ? copysignf(tiny, loaded.d.x) : loaded.d.x;
                    const float dy = fabsf(loaded.d.y) < tiny ? copysignf(tiny, loaded.d.y) : loaded.d.y;
                    const float dz = fabsf(loaded.d.z) < tiny ? copysignf(tiny, loaded.d.z) : loaded.d.z;
                    loaded.rd = v3(1.f / dx, 1.f / dy, 1.f / dz);
                }
                lane_begin(s, kWide ? sc.wroot_ref : sc.root_ref, loaded, 0u);
                lds_index[threadIdx.x] = ray;
                st.sp = 0;
                if (kQueue) {
                    s.tri = jq.head; // no job outstanding
                    job_slot_init<ANYHIT, kNonNeg>(jq, lane, loaded.far_t);
                }
                // A ray with a NaN or infinite origin or direction component cannot be accepted by
                // the triangle test (its determinant is not finite), but NaN slab distances drop every
                // box constraint and it would walk the whole tree.  It is retired as the miss it is;
                // with visit counters requested it is walked, so that the counters stay those of the
                // reference's traversal.  (Slots outside the image arrive here as such rays.)
                if (!STATS) {
                    const float probe = ((loaded.o.x * 0.f + loaded.o.y * 0.f) + loaded.o.z * 0.f) +
                                        ((loaded.d.x * 0.f + loaded.d.y * 0.f) + loaded.d.z * 0.f);
                    if (!(probe == 0.f)) s.cur = kDone;
                }
                if (STATS) {
                    nrays++;
                    cnt.ray_start = cnt.inner + cnt.tris;
                }
            }
            if (begin + base + need > end || begin >= end) { // range exhausted by this fetch
                range = (range + 1u) % kFetchRanges;
                ranges_left--;
            }
            idle_mask = __ballot(!s.have_ray);
        }
        if (__ballot(s.have_ray) == 0) break; // nothing in flight and nothing left to fetch

        // ---- traverse
        if (MODE == 0)
            traverse<ANYHIT, STATS>(sc, s, st, cnt, ranges_left ? tl.min_active : 1u, tl.coop_steps);
        else if (MODE == 3)
            traverse_wide<ANYHIT>(sc, s, st, ranges_left ? tl.min_active : 1u, tl.coop_steps);
        else if (MODE == 2)
            traverse_wide_steps<ANYHIT>(sc, s, st, ranges_left ? tl.min_active : 1u, tl.coop_steps);
        else if (MODE == 6)
            traverse_wide_split<ANYHIT>(sc, s, st, ranges_left ? tl.min_active : 1u, tl.coop_steps);
        else if (MODE == 7)
            traverse_wide_queue<ANYHIT, kNonNeg>(sc, s, st, jq, ranges_left ? tl.min_active : 1u, tl.coop_steps);
        else if (MODE == 8)
            traverse_wide_pairs<ANYHIT>(sc, s, st, ranges_left ? tl.min_active : 1u, tl.coop_steps);
        else
            traverse_steps<ANYHIT, STATS, (MODE == 4)>(sc, s, st, cnt, ranges_left ? tl.min_active : 1u, tl.coop_steps);

        // ---- stragglers: rays that used up their step budget are finished by the whole wave
        {
            // form 7: a ray is handed over once its triangle jobs are done (a leaf at hand goes with it)
            const bool inside_leaf = kQueue ? (int)(jq.head - s.tri) < 0
                                            : (MODE == 2 || MODE == 6 || MODE == 8) ? (s.tri & 3u) != 0 : s.tri < s.tri_end;
            unsigned long long parked = __ballot(s.have_ray && s.cur != kDone && s.steps >= tl.coop_steps && !inside_leaf);
            CoopItem *work = reinterpret_cast<CoopItem *>(tl.coop) + (size_t)(gtid >> 6) * tl.coop_cap;
            while (parked != 0) {
                const unsigned p = (unsigned)__ffsll((long long)parked) - 1u;
                if (kWide)
                    coop_finish_wide<ANYHIT>(sc, s, st, p, work, tl.coop_cap);
                else
                    coop_finish<ANYHIT, STATS>(sc, s, st, cnt, p, work, tl.coop_cap);
                parked &= parked - 1ull;
            }
        }

        // ---- retire finished rays (forms 6 and 7: the walk can be done while triangles are still pending)
        if (kQueue && (int)(jq.head - s.tri) >= 0) s.tri = jq.head; // marks of lanes without jobs follow the ring
        if (s.have_ray && s.cur == kDone && !(MODE == 6 && (s.tri & 3u)) && !(kQueue && (int)(jq.head - s.tri) < 0)) {
            s.index = lds_index[threadIdx.x];
            io.template store<ANYHIT>(s);
            s.have_ray = false;
            if (STATS) {
                const unsigned steps = cnt.inner + cnt.tris - cnt.ray_start;
                cnt.max_steps = steps > cnt.max_steps ? steps : cnt.max_steps;
            }
        }
    }
    flush_stats<STATS>(cnt, nrays, tl.stats, tl.depth);
}

// ------------------------------------------------------------------------
// Two rays per lane (traversal form 5, incoherent rays on the wide tree).
//
// One stepwise iteration ends in a wait for the slowest of the wave's gathers, and with incoherent
// rays nearly every iteration has a lane that goes to HBM: a wave spends most of its life in
// s_waitcnt, and the five waves a SIMD holds at this register budget do not cover that.  Here a
// lane walks two rays, A and B, in turn: the loads of B's next item are issued before A's step is
// computed and the other way round, so each wave always has one gather in flight behind its
// arithmetic -- the memory-level parallelism of twice the waves for the registers of one more ray
// state.  Per ray the walk is traverse_wide_steps(): same items, same order, same results.

struct DualState {
    RayRegs r;
    unsigned cur, tri, tri_end; // as in traverse_wide_steps
    unsigned hit, index, steps;
    bool have_ray;
};

constexpr int kDualLdsDepth = RTC_DUAL_LDS_STACK_DEPTH;

template <bool ANYHIT>
__device__ __forceinline__ bool dual_running(const DualState &x, unsigned max_steps)
{
    return x.cur != kDone && (x.steps < max_steps || x.tri < x.tri_end);
}

// Enters a leaf when the ray has just reached one; returns the item the ray's next step needs.  A
// lane whose ray is not walking reads the root instead: the gather is issued by every lane, always,
// so that the number of loads in flight does not depend on the path taken and a wait for one ray's
// data can leave exactly the other ray's three loads outstanding.
__device__ __forceinline__ const float4 *dual_item(const float4 *items, DualState &x, bool running)
{
    if (running && (x.cur & 3u) && x.tri >= x.tri_end) {
        x.tri = x.cur >> 2;
        x.tri_end = x.tri + kUnitsPerItem * (x.cur & 3u);
    }
    const unsigned unit = x.tri < x.tri_end ? x.tri : x.cur >> 2;
    return items + (running ? unit : 0u);
}

template <bool ANYHIT, class STACK>
__device__ __forceinline__ void dual_step(DualState &x, STACK &st, const float4 a0, const float4 a1, const float4 a2)
{
    x.steps++;
    bool need_pop;
    if (!(x.tri < x.tri_end)) {
        need_pop = qwide_step<ANYHIT>(a0, a1, a2, slab_ray(x.r), x.r.near_t, x.r.far_t, st, x.cur);
    } else {
        const float t = tri_test(a0, a1, make_float2(a2.x, a2.y), x.r.o, x.r.d);
        x.tri += kUnitsPerItem;
        if (t >= x.r.near_t && t < x.r.far_t) {
            if (ANYHIT) {
                x.hit = 1u;
                x.cur = kDone;
                x.tri = x.tri_end;
            } else {
                x.r.far_t = t;
                x.hit = __float_as_uint(a2.y);
            }
        }
        need_pop = (x.tri >= x.tri_end) && (x.cur != kDone);
    }
    if (need_pop) x.cur = st.sp ? st.pop_isolated() : kDone;
}

template <bool ANYHIT, class STACK>
__device__ __forceinline__ void traverse_wide_dual(const SceneDev &sc, DualState &A, DualState &B, STACK &stA, STACK &stB,
                                                   unsigned min_running, unsigned max_steps)
{
    const float4 *__restrict__ items = reinterpret_cast<const float4 *>(sc.witems);
    float4 a0, a1, a2, b0, b1, b2;
    bool ra = dual_running<ANYHIT>(A, max_steps);
    {
        const float4 *p = dual_item(items, A, ra);
        a0 = p[0];
        a1 = p[1];
        a2 = p[2];
    }
    for (;;) {
        const bool rb = dual_running<ANYHIT>(B, max_steps);
        { // B's gather goes out before A's arithmetic
            const float4 *p = dual_item(items, B, rb);
            b0 = p[0];
            b1 = p[1];
            b2 = p[2];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ra) dual_step<ANYHIT>(A, stA, a0, a1, a2);
        ra = dual_running<ANYHIT>(A, max_steps);
        {
            const float4 *p = dual_item(items, A, ra);
            a0 = p[0];
            a1 = p[1];
            a2 = p[2];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rb) dual_step<ANYHIT>(B, stB, b0, b1, b2);
        const unsigned running = (unsigned)__popcll(__ballot(ra)) +
                                 (unsigned)__popcll(__ballot(dual_running<ANYHIT>(B, max_steps)));
        if (running < min_running) break; // (the gather of A's issued above is repeated on re-entry)
    }
}

template <bool ANYHIT, class IO>
__global__ __launch_bounds__(kBlock, RTC_DUAL_WAVES_PER_SIMD) void trace_kernel_dual(SceneDev sc, IO io, const unsigned *n_ptr,
                                                                                  unsigned n_val, TraceLaunch tl)
{
    __shared__ unsigned lds_stack[2 * kDualLdsDepth * kBlock];
    const unsigned n = n_ptr ? *n_ptr : n_val;
    const unsigned gtid = blockIdx.x * kBlock + threadIdx.x;
    const unsigned threads = gridDim.x * kBlock;
    Stack<kBlock, kDualLdsDepth> st[2];
#pragma unroll
    for (int k = 0; k < 2; k++) { // the two stacks of a lane: separate LDS planes, interleaved overflow columns
        st[k].lds = lds_stack + k * kDualLdsDepth * kBlock + threadIdx.x;
        st[k].ovf_base = tl.ovf;
        st[k].ovf_lane = k * threads + gtid;
        st[k].ovf_stride = 2u * threads;
        st[k].sp = 0;
    }
    const unsigned range_len = (n + kFetchRanges - 1u) / kFetchRanges;
    unsigned range = (gtid >> 6) % kFetchRanges; // wave-uniform
    unsigned ranges_left = kFetchRanges;         // ranges not yet seen empty by this wave
    const unsigned lane = lane_id();

    DualState s[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        s[k].cur = kDone;
        s[k].tri = s[k].tri_end = 0;
        s[k].steps = 0;
        s[k].have_ray = false;
    }

    while (true) {
        // ---- refill the idle slots, A's then B's (trace_kernel's fetch, once per slot)
#pragma unroll
        for (int k = 0; k < 2; k++) {
            unsigned long long idle_mask = __ballot(!s[k].have_ray);
            while (ranges_left != 0 && idle_mask != 0) {
                const unsigned need = (unsigned)__popcll(idle_mask);
                const unsigned rank = prefix_count(idle_mask);
                const unsigned leader = (unsigned)__ffsll((long long)idle_mask) - 1u;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&tl.fetch[range * kFetchStride], need);
                base = (unsigned)__shfl((int)base, (int)leader);
                const unsigned begin = range * range_len;
                const unsigned end = (begin + range_len < n) ? begin + range_len : n;
                const unsigned idx = begin + base + rank;
                if (!s[k].have_ray && begin + base < end && idx < end) {
                    unsigned ray = idx;
                    if (tl.perm && idx < tl.perm_n) {
                        ray = tl.perm[idx];
                        if (ray >= n) ray = idx;
                    }
                    RayRegs loaded = io.load(ray);
                    // bounded reciprocal and retirement of non-finite rays: see trace_kernel
                    const float tiny = 1e-24f;
                    const float dx = fabsf(loaded.d.x) < tiny ? copysignf(tiny, loaded.d.x) : loaded.d.x;
                    const float dy = fabsf(loaded.d.y) < tiny ? copysignf(tiny, loaded.d.y) : loaded.d.y;
                    const float dz = fabsf(loaded.d.z) < tiny ? copysignf(tiny, loaded.d.z) : loaded.d.z;
                    loaded.rd = v3(1.f / dx, 1.f / dy, 1.f / dz);
                    s[k].r = loaded;
                    s[k].cur = sc.wroot_ref;
                    s[k].tri = s[k].tri_end = 0;
                    s[k].hit = 0xFFFFFFFFu;
                    s[k].index = ray;
                    s[k].steps = 0;
                    s[k].have_ray = true;
                    st[k].sp = 0;
                    const float probe = ((loaded.o.x * 0.f + loaded.o.y * 0.f) + loaded.o.z * 0.f) +
                                        ((loaded.d.x * 0.f + loaded.d.y * 0.f) + loaded.d.z * 0.f);
                    if (!(probe == 0.f)) s[k].cur = kDone;
                }
                if (begin + base + need > end || begin >= end) {
                    range = (range + 1u) % kFetchRanges;
                    ranges_left--;
                }
                idle_mask = __ballot(!s[k].have_ray);
            }
        }
        if (__ballot(s[0].have_ray || s[1].have_ray) == 0) break;

        // ---- traverse: until fewer than min_active of the wave's 128 slots are walking
        traverse_wide_dual<ANYHIT>(sc, s[0], s[1], st[0], st[1], ranges_left ? 2u * tl.min_active : 1u, tl.coop_steps);

        // ---- stragglers and retirement, per slot
#pragma unroll
        for (int k = 0; k < 2; k++) {
            unsigned long long parked = __ballot(s[k].have_ray && s[k].cur != kDone && s[k].steps >= tl.coop_steps &&
                                                 s[k].tri >= s[k].tri_end);
            CoopItem *work = reinterpret_cast<CoopItem *>(tl.coop) + (size_t)(gtid >> 6) * tl.coop_cap;
            while (parked != 0) {
                const unsigned p = (unsigned)__ffsll((long long)parked) - 1u;
                coop_finish_wide<ANYHIT>(sc, s[k], st[k], p, work, tl.coop_cap);
                parked &= parked - 1ull;
            }
            if (s[k].have_ray && s[k].cur == kDone) {
                LaneState done; // the IO policies read index, hit and far distance
                done.index = s[k].index;
                done.hit = s[k].hit;
                done.r.far_t = s[k].r.far_t;
                io.template store<ANYHIT>(done);
                s[k].have_ray = false;
            }
        }
    }
}

// rays of one launch by depth; also into total_rays when per-visit statistics are off (with them on
// the trace kernel counts the rays it fetched)
__global__ void count_rays_kernel(const unsigned *n_ptr, StatsDev *stats, unsigned depth, bool add_total)
{
    const unsigned long long n = *n_ptr;
    if (add_total) atomicAdd(&stats->total_rays, n);
    atomicAdd(&stats->depth_rays[depth < kStatDepths ? depth : kStatDepths - 1u], n);
}

// ------------------------------------------------------------------------
// random numbers and sampling

// Counter-based generator keyed by (seed, global pixel, global sample,
// dimension); u32 -> f32 as rand 0.8.5's Standard: (x >> 8) * 2^-24.
__device__ __forceinline__ unsigned pcg_hash(unsigned v)
{
    unsigned s = v * 747796405u + 2891336453u;
    unsigned w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
__device__ __forceinline__ unsigned rng_key(unsigned seed_key, unsigned pixel, unsigned sample)
{
    return pcg_hash(pcg_hash(seed_key ^ pixel) ^ sample);
}
__device__ __forceinline__ float rng_f32(unsigned key, unsigned dim)
{
    return (float)(pcg_hash(key ^ dim) >> 8) * (1.f / 16777216.f);
}

// sin and cos of 2*pi*u, u in [0,1): Cody-Waite reduction by pi/2 and Cephes
// minimax polynomials, a fixed sequence of un-fused f32 operations (stands in
// for the libm / `wide` calls of src/brdf.rs:13-16, 82-83, which are not
// bit-reproducible across targets).
__device__ __forceinline__ void sincos_2pi(float u, float &s_out, float &c_out)
{
    const float b = (2.f * 3.14159274f) * u;
    const float kf = floorf(b * 0.636619772f + 0.5f);
    const int k = (int)kf;
    float r = b - kf * 1.5703125f;
    r = r - kf * 4.83751297e-4f;
    r = r - kf * 7.54978995e-8f;
    const float z = r * r;
    const float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z -
                     0.5f * z + 1.f;
    switch (k & 3) {
    case 0: s_out = sp; c_out = cp; break;
    case 1: s_out = cp; c_out = -sp; break;
    case 2: s_out = -sp; c_out = -cp; break;
    default: s_out = -cp; c_out = sp; break;
    }
}

struct Quat {
    float w;
    V3 v;
};

// fast_rotation_between(from, +Z), src/math.rs:48-58; nalgebra's
// new_normalize divides by sqrt(((i*i+j*j)+k*k)+w*w).  dot == -1 takes the
// rotation by pi about Y (w = cos(pi/2) in f32).
__device__ __forceinline__ Quat rotation_to_z(V3 from)
{
    const V3 to = v3(0.f, 0.f, 1.f);
    Quat q;
    const float dt = dot(from, to);
    if (dt == -1.f) {
        q.w = -4.37113883e-8f;
        q.v = v3(0.f, 1.f, 0.f);
        return q;
    }
    const float w = 1.f + dt;
    const V3 c = cross(from, to);
    const float nrm = sqrtf(((c.x * c.x + c.y * c.y) + c.z * c.z) + w * w);
    q.w = w / nrm;
    q.v = divs(c, nrm);
    return q;
}

// nalgebra UnitQuaternion * Vector3: t = 2*(q.v x x); (t*q.w + q.v x t) + x
__device__ __forceinline__ V3 rotate(Quat q, V3 x)
{
    const V3 t = scale(cross(q.v, x), 2.f);
    const V3 c = cross(q.v, t);
    return (scale(t, q.w) + c) + x;
}

// ------------------------------------------------------------------------
// generate: camera rays (src/camera.rs:56-135, integrate_stream.rs:14-54)

// (rx, ry) is the top-left pixel of the tile that holds (x, y)
__device__ __forceinline__ V3 camera_pixel_ray(const CameraDev &cam, unsigned rx, unsigned ry, unsigned x, unsigned y)
{
    // region_top_left of the tile, then the row start, then x advanced by
    // repeated += (src/camera.rs:72-75, 98-113) so that rounding matches the
    // reference's per-tile walk.
    const V3 tl = v3(cam.top_left[0], cam.top_left[1], cam.top_left[2]);
    const V3 sx = v3(cam.step_x[0], cam.step_x[1], cam.step_x[2]);
    const V3 sy = v3(cam.step_y[0], cam.step_y[1], cam.step_y[2]);
    const V3 region_tl = (tl + scale(sx, (float)rx)) + scale(sy, (float)ry);
    const unsigned cy = y - ry, cx = x - rx;
    V3 cur = region_tl;
    if (cy > 0) cur = region_tl + scale(sy, (float)cy);
    for (unsigned i = 0; i < cx; i++) cur = cur + sx;
    return cur;
}

__device__ __forceinline__ V3 camera_sample(const CameraDev &cam, V3 pixel_ray, float u0, float u1)
{
    const V3 right = v3(cam.right[0], cam.right[1], cam.right[2]);
    const V3 up = v3(cam.up[0], cam.up[1], cam.up[2]);
    const V3 off = scale(scale(right, u0 - 0.5f) + scale(up, u1 - 0.5f), cam.pixel_size);
    return normalize(pixel_ray + off);
}

// Path slots of the pass: slot = ((tile_local*ts*ts + py*ts + px) * k + s);
// tiles are the shard's tiles [tile_begin, tile_begin+tile_count).  A block
// covers kGenItems*256 consecutive slots; slots outside the image (partial edge
// tiles) or beyond the pass's sample count produce no ray.
constexpr int kGenItems = 8;
constexpr int kGenBlock = 512;

// Division by a pass constant: a shift when the constant is a power of two (tile size 16, tile area
// 256 and the usual samples per pass are), the general division otherwise.  `shift` is 0xFFFFFFFF
// when the divisor is not a power of two.
__device__ __forceinline__ void divmod(unsigned n, unsigned d, unsigned shift, unsigned &q, unsigned &r)
{
    if (shift != 0xFFFFFFFFu) { // uniform over the launch
        q = n >> shift;
        r = n & (d - 1u);
    } else {
        q = n / d;
        r = n - q * d;
    }
}

// Pixel (x, y), the top-left pixel (rx, ry) of its tile, sample index s and pixel slot p of a pass slot.
__device__ __forceinline__ bool slot_to_pixel(const PassDev &ps, unsigned slot, unsigned &x, unsigned &y,
                                              unsigned &s, unsigned &rx, unsigned &ry, unsigned &p)
{
    if (slot >= ps.num_slots) return false;
    unsigned pin, tl, px, py;
    divmod(slot, ps.samples_per_pass, ps.k_shift, p, s);
    divmod(p, ps.tile_size * ps.tile_size, ps.ts2_shift, tl, pin);
    const unsigned tile = ps.shard_index + (ps.tile_begin + tl) * ps.shard_count;
    const unsigned ty = tile / ps.tiles_x, tx = tile - ty * ps.tiles_x;
    divmod(pin, ps.tile_size, ps.ts_shift, py, px);
    rx = tx * ps.tile_size;
    ry = ty * ps.tile_size;
    x = rx + px;
    y = ry + py;
    return (x < ps.width) && (y < ps.height) && (s < ps.samples_this_pass);
}
__device__ __forceinline__ bool slot_to_pixel(const PassDev &ps, unsigned slot, unsigned &x, unsigned &y,
                                              unsigned &s)
{
    unsigned rx, ry, p;
    return slot_to_pixel(ps, slot, x, y, s, rx, ry, p);
}

// Primary path of pass slot `slot`: the camera ray of generate_rays
// (src/integrators/integrate_stream.rs:14-54) and its PathInfo.  False for slots
// outside the image or beyond the pass's samples.
__device__ __forceinline__ bool primary_path(const CameraDev &cam, const PassDev &ps, unsigned slot, QRay &q,
                                             PathInfo &pi)
{
    unsigned x, y, sidx, rx, ry, p;
    if (!slot_to_pixel(ps, slot, x, y, sidx, rx, ry, p)) return false;
    const unsigned pixel = y * ps.width + x;
    const unsigned gs = ps.sample_base + sidx;
    const unsigned key = rng_key(ps.seed_key, pixel, gs);
    const V3 pr = camera_pixel_ray(cam, rx, ry, x, y);
    const V3 d = camera_sample(cam, pr, rng_f32(key, 0u), rng_f32(key, 1u));
    q.o[0] = cam.eye[0];
    q.o[1] = cam.eye[1];
    q.o[2] = cam.eye[2];
    q.near_t = 0.f;
    q.d[0] = d.x;
    q.d[1] = d.y;
    q.d[2] = d.z;
    q.far_t = __builtin_inff();
    pi.w[0] = pi.w[1] = pi.w[2] = 1.f;
    // radiance index: sample-major over the pass's pixel slots (see accumulate_kernel)
    pi.slot = sidx * ps.pixel_slots + p;
    return true;
}

// Pixel (y * width + x) and global sample index of the path that owns radiance slot `slot`: the
// inverse of the slot numbering above, for the random-number keys of the deeper bounces.
__device__ __forceinline__ void path_keys(const PassDev &ps, unsigned slot, unsigned &pixel, unsigned &gs)
{
    const unsigned sidx = slot / ps.pixel_slots, p = slot - sidx * ps.pixel_slots;
    unsigned tl, pin, py, px;
    divmod(p, ps.tile_size * ps.tile_size, ps.ts2_shift, tl, pin);
    const unsigned tile = ps.shard_index + (ps.tile_begin + tl) * ps.shard_count;
    const unsigned ty = tile / ps.tiles_x, tx = tile - ty * ps.tiles_x;
    divmod(pin, ps.tile_size, ps.ts_shift, py, px);
    pixel = (ty * ps.tile_size + py) * ps.width + (tx * ps.tile_size + px);
    gs = ps.sample_base + sidx;
}

__global__ __launch_bounds__(kGenBlock) void generate_kernel(CameraDev cam, PassDev ps, QRay *rays,
                                                            PathInfo *infos, unsigned *n_out, float *rad_r,
                                                            float *rad_g, float *rad_b)
{
    __shared__ unsigned lds[64];
    const unsigned block_base = blockIdx.x * (unsigned)(kGenBlock * kGenItems);
    bool pred[1][kGenItems];
    unsigned pos[1][kGenItems];
#pragma unroll
    for (int it = 0; it < kGenItems; it++) {
        unsigned x, y, sidx;
        pred[0][it] = slot_to_pixel(ps, block_base + it * kGenBlock + threadIdx.x, x, y, sidx);
    }
    unsigned *const counters[1] = {n_out};
    block_reserve<kGenBlock / 64, kGenItems, 1>(pred, counters, pos, lds);
#pragma unroll
    for (int it = 0; it < kGenItems; it++) {
        const unsigned slot = block_base + it * kGenBlock + threadIdx.x;
        if (slot < ps.num_slots) {
            rad_r[slot] = 0.f;
            rad_g[slot] = 0.f;
            rad_b[slot] = 0.f;
        }
        QRay q;
        PathInfo pi;
        if (!primary_path(cam, ps, slot, q, pi)) continue;
        const unsigned idx = pos[0][it];
        rays[idx] = q;
        infos[idx] = pi;
    }
}

// Primary rays of the whole frame as StreamRays in (pixel, sample) order.
__global__ __launch_bounds__(256) void generate_stream_kernel(CameraDev cam, PassDev ps, rt_stream_ray_dev *rays,
                                                             unsigned long long total)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const unsigned s = (unsigned)(i % ps.samples_this_pass);
    const unsigned pixel = (unsigned)(i / ps.samples_this_pass);
    const unsigned x = pixel % ps.width, y = pixel / ps.width;
    const unsigned gs = ps.sample_base + s;
    const unsigned key = rng_key(ps.seed_key, pixel, gs);
    const V3 pr = camera_pixel_ray(cam, x / ps.tile_size * ps.tile_size, y / ps.tile_size * ps.tile_size, x, y);
    const V3 d = camera_sample(cam, pr, rng_f32(key, 0u), rng_f32(key, 1u));
    rt_stream_ray_dev r;
    r.origin_far = make_float4(cam.eye[0], cam.eye[1], cam.eye[2], __builtin_inff());
    r.dir_recip_near = make_float4(1.f / d.x, 1.f / d.y, 1.f / d.z, 0.f);
    r.direction = make_float4(d.x, d.y, d.z, 0.f);
    rays[i] = r;
}

// ------------------------------------------------------------------------
// shade: loop body of integrate_tile_stream
// (src/integrators/integrate_stream.rs:104-138) for every ray of the queue.

// 30-bit Morton code of a point inside the scene bounds (10 bits per axis)
__device__ __forceinline__ unsigned spread10(unsigned v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ unsigned morton_key(const ShadeArgs &a, const float *p)
{
    unsigned q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float f = (p[k] - a.key_lo[k]) * a.key_scale[k];
        f = fminf(fmaxf(f, 0.f), 1023.f);
        q[k] = (unsigned)f;
    }
    return ((spread10(q[2]) << 2) | (spread10(q[1]) << 1) | spread10(q[0])) & a.key_mask;
}

struct ShadeOut {
    QRay sh_ray, bn_ray;
    ShadowInfo sh_info;
    PathInfo bn_info;
};

__device__ __forceinline__ void shade_one(const ShadeArgs &a, unsigned i, unsigned n, bool &has_shadow,
                                          bool &has_bounce, ShadeOut &o)
{
    has_shadow = false;
    has_bounce = false;
    if (i >= n) return;
    const unsigned id = a.hit_ids[i];
    if (id == 0xFFFFFFFFu) return; // a miss contributes nothing and ends the path (integrate_stream.rs:105)
    QRay ray;
    PathInfo info;
    if (a.primary) { // depth 0 without queues: slot i is its own path
        if (!primary_path(a.cam, a.ps, i, ray, info)) return;
        ray.far_t = a.hit_t[i];
    } else {
        ray = a.rays_in[i];
        info = a.infos_in[i];
    }
    const V3 org = v3(ray.o[0], ray.o[1], ray.o[2]);
    const V3 d = v3(ray.d[0], ray.d[1], ray.d[2]);
    const V3 hp = org + scale(d, ray.far_t); // src/ray.rs:40-42
    const V3 wo = -d;
    const float4 n4 = reinterpret_cast<const float4 *>(a.normals)[id];
    const V3 nrm = v3(n4.x, n4.y, n4.z);
    const V3 pw = v3(info.w[0], info.w[1], info.w[2]);

    // sample_light, src/integrators/common.rs:37-60
    {
        const V3 lv = v3(a.light_pos[0], a.light_pos[1], a.light_pos[2]) - hp;
        float ndotl = dot(nrm, lv);
        if (!(ndotl * dot(nrm, wo) < 0.f)) {
            const float d2 = dot(lv, lv);
            const float dist = sqrtf(d2);
            const V3 wi = divs(lv, dist);
            ndotl = fabsf(ndotl / dist);
            const float f = 0.8f / 3.14159274f; // src/brdf.rs:24-26
            const V3 li = v3(a.light_intensity[0], a.light_intensity[1], a.light_intensity[2]);
            const V3 w = divs(scale(mul(li, v3(f, f, f)), ndotl), d2);
            if (!(w.x == 0.f && w.y == 0.f && w.z == 0.f)) {
                has_shadow = true;
                o.sh_ray.o[0] = hp.x; o.sh_ray.o[1] = hp.y; o.sh_ray.o[2] = hp.z;
                o.sh_ray.near_t = 1e-5f; // SHADOW_RAY_NEAR, common.rs:12
                o.sh_ray.d[0] = wi.x; o.sh_ray.d[1] = wi.y; o.sh_ray.d[2] = wi.z;
                o.sh_ray.far_t = dist;
                const V3 c = mul(pw, w);
                o.sh_info.w[0] = c.x; o.sh_info.w[1] = c.y; o.sh_info.w[2] = c.z;
                o.sh_info.slot = info.slot;
            }
        }
    }

    // sample_diffuse_ray, src/integrators/common.rs:64-78 + src/brdf.rs:10-17,36-44
    if (a.depth < a.max_bounces) {
        unsigned pixel, gs;
        path_keys(a.ps, info.slot, pixel, gs);
        const unsigned key = rng_key(a.seed_key, pixel, gs);
        const float u0 = rng_f32(key, 2u + 2u * a.depth);
        const float u1 = rng_f32(key, 3u + 2u * a.depth);
        const Quat q = rotation_to_z(nrm);
        const V3 wo_l = rotate(q, wo);
        const float aa = sqrtf(u0);
        float sn, cs;
        sincos_2pi(u1, sn, cs);
        V3 wi_l = v3(aa * cs, aa * sn, sqrtf(1.f - u0));
        const float pdf = wi_l.z / 3.14159274f;
        const float f = 0.8f / 3.14159274f;
        wi_l.z = copysignf(wi_l.z, wo_l.z);
        Quat qi = q;
        qi.v = -q.v;
        const V3 wi = rotate(qi, wi_l);
        const float ndotl = fabsf(dot(nrm, wi));
        const float w = f * ndotl / pdf;
        has_bounce = true;
        o.bn_ray.o[0] = hp.x; o.bn_ray.o[1] = hp.y; o.bn_ray.o[2] = hp.z;
        o.bn_ray.near_t = 1e-5f;
        o.bn_ray.d[0] = wi.x; o.bn_ray.d[1] = wi.y; o.bn_ray.d[2] = wi.z;
        o.bn_ray.far_t = __builtin_inff();
        o.bn_info.w[0] = pw.x * w; o.bn_info.w[1] = pw.y * w; o.bn_info.w[2] = pw.z * w;
        o.bn_info.slot = info.slot;
    }
}

// One path per thread in blocks of 1024: 2 per thread is 28 % slower (fewer loads in flight per
// register), blocks of 512 / 256 are 23 % / 110 % slower (one queue reservation per block and queue:
// the two counters take about 10^8 atomics per second each).
#ifndef RTC_SHADE_BLOCK
#define RTC_SHADE_BLOCK 1024
#endif
#ifndef RTC_SHADE_ITEMS
#define RTC_SHADE_ITEMS 1
#endif
constexpr int kShadeBlock = RTC_SHADE_BLOCK;
constexpr int kShadeItems = RTC_SHADE_ITEMS;

__global__ __launch_bounds__(kShadeBlock) void shade_kernel(ShadeArgs a)
{
    __shared__ unsigned lds[64];
    const unsigned n = a.n_in ? *a.n_in : a.n_in_val;
    const unsigned block_base = blockIdx.x * (unsigned)(kShadeBlock * kShadeItems);
    if (block_base >= n) return; // whole block past the queue (uniform)
    bool pred[2][kShadeItems];
    unsigned pos[2][kShadeItems];
    ShadeOut out[kShadeItems];
#pragma unroll
    for (int it = 0; it < kShadeItems; it++)
        shade_one(a, block_base + it * kShadeBlock + threadIdx.x, n, pred[0][it], pred[1][it], out[it]);
    unsigned *const counters[2] = {a.n_shadow, a.n_out};
    block_reserve<kShadeBlock / 64, kShadeItems, 2>(pred, counters, pos, lds);
#pragma unroll
    for (int it = 0; it < kShadeItems; it++) {
        if (pred[0][it]) {
            a.shadow_rays[pos[0][it]] = out[it].sh_ray;
            a.shadow_infos[pos[0][it]] = out[it].sh_info;
            if (a.shadow_keys) a.shadow_keys[pos[0][it]] = morton_key(a, out[it].sh_ray.o);
        }
        if (pred[1][it]) {
            a.rays_out[pos[1][it]] = out[it].bn_ray;
            a.infos_out[pos[1][it]] = out[it].bn_info;
            if (a.bounce_keys) {
                unsigned key = morton_key(a, out[it].bn_ray.o);
                if (a.key_dir_shift < 32u) {
                    const unsigned oct = (__float_as_uint(out[it].bn_ray.d[0]) >> 31) |
                                         ((__float_as_uint(out[it].bn_ray.d[1]) >> 31) << 1) |
                                         ((__float_as_uint(out[it].bn_ray.d[2]) >> 31) << 2);
                    key = (key & ~(7u << a.key_dir_shift)) | (oct << a.key_dir_shift);
                }
                a.bounce_keys[pos[1][it]] = key;
            }
        }
    }
}

// Depth-0 extend without a ray queue: the lane makes the camera ray of its slot itself and
// leaves hit id and distance per slot; shade_kernel rebuilds ray and PathInfo from the slot in
// the same way.  A slot outside the image gets a NaN direction, which the fetch retires as a miss.
struct PrimaryIO {
    static constexpr bool kNearNonNegative = true; // camera rays start at 0
    CameraDev cam;
    PassDev ps;
    unsigned *hit_ids;
    float *hit_t;
    __device__ __forceinline__ RayRegs load(unsigned slot) const
    {
        QRay q;
        PathInfo pi;
        RayRegs r;
        if (!primary_path(cam, ps, slot, q, pi)) {
            const float nan = __builtin_nanf("");
            r.o = v3(0.f, 0.f, 0.f);
            r.d = v3(nan, nan, nan);
            r.rd = v3(nan, nan, nan);
            r.near_t = 0.f;
            r.far_t = __builtin_inff();
            return r;
        }
        r.o = v3(q.o[0], q.o[1], q.o[2]);
        r.near_t = q.near_t;
        r.d = v3(q.d[0], q.d[1], q.d[2]);
        r.far_t = q.far_t;
        r.rd = v3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z); // src/ray.rs:18
        return r;
    }
    template <bool ANYHIT>
    __device__ __forceinline__ void store(const LaneState &s) const
    {
        hit_ids[s.index] = s.hit;
        hit_t[s.index] = s.r.far_t;
    }
};

// ------------------------------------------------------------------------
// accumulate: per pixel, sum the pass's k sample slots in a fixed order and
// add (sum, number of samples) to the float4 accumulator (src/main.rs:108,
// integrate_stream.rs:85-87).

__global__ __launch_bounds__(256) void accumulate_kernel(PassDev ps, const float *rad_r, const float *rad_g,
                                                        const float *rad_b, float *accum)
{
    const unsigned p = blockIdx.x * 256u + threadIdx.x; // pixel slot of the pass
    const unsigned ts2 = ps.tile_size * ps.tile_size;
    if (p >= ps.tile_count * ts2) return;
    const unsigned pin = p % ts2;
    const unsigned tl = p / ts2;
    const unsigned tile = ps.shard_index + (ps.tile_begin + tl) * ps.shard_count;
    const unsigned tx = tile % ps.tiles_x, ty = tile / ps.tiles_x;
    const unsigned x = tx * ps.tile_size + pin % ps.tile_size;
    const unsigned y = ty * ps.tile_size + pin / ps.tile_size;
    if (x >= ps.width || y >= ps.height) return;
    float r = 0.f, g = 0.f, b = 0.f;
    // radiance slots are sample-major (sample s of pass pixel p at s * pixels + p): neighbouring
    // threads read neighbouring words whatever the samples per pass
    const size_t pixels = (size_t)ps.tile_count * ts2;
    for (unsigned s = 0; s < ps.samples_this_pass; s++) {
        r += rad_r[s * pixels + p];
        g += rad_g[s * pixels + p];
        b += rad_b[s * pixels + p];
    }
    float4 *dst = reinterpret_cast<float4 *>(accum) + ((size_t)y * ps.width + x);
    float4 v = *dst;
    v.x += r;
    v.y += g;
    v.z += b;
    v.w += (float)ps.samples_this_pass;
    *dst = v;
}

// color_vec_to_rgb_norm_gamma, src/main.rs:34-41
__global__ __launch_bounds__(256) void resolve_kernel(const float *accum, unsigned long long npix,
                                                     unsigned char *rgb)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    const float4 v = reinterpret_cast<const float4 *>(accum)[i];
    const float c[3] = {v.x, v.y, v.z};
    for (int k = 0; k < 3; k++) {
        const float f = sqrtf(c[k] / v.w) * 255.f;
        unsigned char b;
        if (!(f > 0.f)) b = 0;
        else if (f >= 255.f) b = 255;
        else b = (unsigned char)f;
        rgb[i * 3 + k] = b;
    }
}

// ------------------------------------------------------------------------
// launchers

static unsigned trace_grid(unsigned n, unsigned max_blocks)
{
    unsigned need = (n + kBlock - 1) / kBlock;
    if (need < 1) need = 1;
    return need < max_blocks ? need : max_blocks;
}

template <bool ANYHIT, bool STATS, int MODE, class IO>
static void launch_one(const SceneDev &sc, const IO &io, const unsigned *n_ptr, unsigned n_val, unsigned grid,
                       const TraceLaunch &tl, hipStream_t stream)
{
    // The grid is the same for every trace kernel (the most blocks any of them can keep resident);
    // blocks of a kernel with a lower occupancy start when the first ones end and find the queue empty.
    hipLaunchKernelGGL((trace_kernel<ANYHIT, STATS, MODE, IO>), dim3(grid), dim3(kBlock), 0, stream, sc, io, n_ptr,
                       n_val, tl);
}

template <int MODE, class IO>
static void launch_trace_mode(bool anyhit, bool stats, const SceneDev &sc, const IO &io, const unsigned *n_ptr,
                              unsigned n_val, unsigned grid, const TraceLaunch &tl, hipStream_t stream)
{
    if (anyhit) {
        if (stats)
            launch_one<true, true, MODE, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
        else
            launch_one<true, false, MODE, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
    } else {
        if (stats)
            launch_one<false, true, MODE, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
        else
            launch_one<false, false, MODE, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
    }
}

template <class IO>
static void launch_trace(bool anyhit, bool stats, const SceneDev &sc, const IO &io, const unsigned *n_ptr,
                         unsigned n_val, unsigned n_upper, const TraceLaunch &tl, unsigned max_blocks,
                         hipStream_t stream)
{
    const unsigned grid = trace_grid(n_upper, max_blocks);
    if (tl.mode == 5 && !stats && sc.witems) {
        // two rays per lane: half the threads for the same rays in flight
        const unsigned dual_grid = trace_grid((n_upper + 1u) / 2u, max_blocks);
        if (anyhit)
            hipLaunchKernelGGL((trace_kernel_dual<true, IO>), dim3(dual_grid), dim3(kBlock), 0, stream, sc, io, n_ptr,
                               n_val, tl);
        else
            hipLaunchKernelGGL((trace_kernel_dual<false, IO>), dim3(dual_grid), dim3(kBlock), 0, stream, sc, io, n_ptr,
                               n_val, tl);
        return;
    }
    if ((tl.mode == 2 || tl.mode == 3 || tl.mode == 6 || tl.mode == 7 || tl.mode == 8) && !stats && sc.witems) {
        // 4-wide tree, no visit counters in these forms
        if (tl.mode == 8) {
            if (anyhit)
                launch_one<true, false, 8, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
            else
                launch_one<false, false, 8, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
        } else if (tl.mode == 7 && tl.job_units_ok) {
            if (anyhit)
                launch_one<true, false, 7, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
            else
                launch_one<false, false, 7, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
        } else if (tl.mode == 6) {
            if (anyhit)
                launch_one<true, false, 6, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
            else
                launch_one<false, false, 6, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
        } else if (tl.mode == 2 || tl.mode == 7) {
            if (anyhit)
                launch_one<true, false, 2, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
            else
                launch_one<false, false, 2, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
        } else {
            if (anyhit)
                launch_one<true, false, 3, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
            else
                launch_one<false, false, 3, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
        }
    }
    else if (tl.mode == 0)
        launch_trace_mode<0>(anyhit, stats, sc, io, n_ptr, n_val, grid, tl, stream);
    else if (tl.mode == 4 && !anyhit) { // the any-hit walk of form 1 already is Bvh::occluded1
        if (stats)
            launch_one<false, true, 4, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
        else
            launch_one<false, false, 4, IO>(sc, io, n_ptr, n_val, grid, tl, stream);
    } else
        launch_trace_mode<1>(anyhit, stats, sc, io, n_ptr, n_val, grid, tl, stream);
}

void launch_trace_stream(bool anyhit, bool stats, const SceneDev &sc, rt_stream_ray_dev *rays, unsigned *hits,
                         unsigned char *occluded, unsigned n, const TraceLaunch &tl, unsigned max_blocks,
                         hipStream_t stream)
{
    if (n == 0) return;
    StreamIO io{rays, hits, occluded};
    launch_trace(anyhit, stats, sc, io, nullptr, n, n, tl, max_blocks, stream);
}

void launch_extend(bool stats, const SceneDev &sc, QRay *rays, unsigned *hit_ids, const unsigned *n_ptr,
                   unsigned n_upper, const TraceLaunch &tl, unsigned max_blocks, hipStream_t stream)
{
    QueueIO io{rays, hit_ids, nullptr, nullptr, nullptr, nullptr};
    launch_trace(false, stats, sc, io, n_ptr, 0u, n_upper, tl, max_blocks, stream);
}

void launch_connect(bool stats, const SceneDev &sc, const QRay *rays, const ShadowInfo *infos,
                    const unsigned *n_ptr, unsigned n_upper, float *rad_r, float *rad_g, float *rad_b,
                    const TraceLaunch &tl, unsigned max_blocks, hipStream_t stream)
{
    QueueIO io{const_cast<QRay *>(rays), nullptr, infos, rad_r, rad_g, rad_b};
    launch_trace(true, stats, sc, io, n_ptr, 0u, n_upper, tl, max_blocks, stream);
}

void launch_extend_primary(const SceneDev &sc, const CameraDev &cam, const PassDev &ps, unsigned *hit_ids,
                           float *hit_t, const TraceLaunch &tl, unsigned max_blocks, hipStream_t stream)
{
    const PrimaryIO io{cam, ps, hit_ids, hit_t};
    const unsigned n = ps.num_slots;
    const unsigned grid = trace_grid(n, max_blocks);
    const unsigned mode = ((tl.mode >= 2 && !sc.witems) || tl.mode > 3) ? 1u : tl.mode;
    switch (mode) {
    case 0:
        launch_one<false, false, 0, PrimaryIO>(sc, io, nullptr, n, grid, tl, stream);
        break;
    case 1:
        launch_one<false, false, 1, PrimaryIO>(sc, io, nullptr, n, grid, tl, stream);
        break;
    case 2:
        launch_one<false, false, 2, PrimaryIO>(sc, io, nullptr, n, grid, tl, stream);
        break;
    default:
        launch_one<false, false, 3, PrimaryIO>(sc, io, nullptr, n, grid, tl, stream);
        break;
    }
}

// far distances of a StreamRay array as a dense array (what a closest-hit call changes in its rays)
__global__ __launch_bounds__(256) void gather_far_kernel(const rt_stream_ray_dev *rays, float *far_t, unsigned n)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) far_t[i] = rays[i].origin_far.w;
}

void launch_gather_far(const rt_stream_ray_dev *rays, float *far_t, unsigned n, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(gather_far_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, rays, far_t, n);
}

__global__ void set_count_kernel(unsigned *dst, unsigned value) { *dst = value; }

void launch_set_count(unsigned *dst, unsigned value, hipStream_t stream)
{
    hipLaunchKernelGGL(set_count_kernel, dim3(1), dim3(1), 0, stream, dst, value);
}

void launch_count_rays(const unsigned *n_ptr, StatsDev *dstats, unsigned depth, bool add_total, hipStream_t stream)
{
    hipLaunchKernelGGL(count_rays_kernel, dim3(1), dim3(1), 0, stream, n_ptr, dstats, depth, add_total);
}

void launch_generate(const CameraDev &cam, const PassDev &ps, QRay *rays, PathInfo *infos, unsigned *n_out,
                     float *rad_r, float *rad_g, float *rad_b, hipStream_t stream)
{
    const unsigned per_block = (unsigned)(kGenBlock * kGenItems);
    const unsigned grid = (ps.num_slots + per_block - 1u) / per_block;
    if (grid == 0) return;
    hipLaunchKernelGGL(generate_kernel, dim3(grid), dim3(kGenBlock), 0, stream, cam, ps, rays, infos, n_out,
                       rad_r, rad_g, rad_b);
}

void launch_generate_stream(const CameraDev &cam, const PassDev &ps, rt_stream_ray_dev *rays,
                            unsigned long long total, hipStream_t stream)
{
    const unsigned grid = (unsigned)((total + 255ull) / 256ull);
    if (grid == 0) return;
    hipLaunchKernelGGL(generate_stream_kernel, dim3(grid), dim3(256), 0, stream, cam, ps, rays, total);
}

void launch_shade(const ShadeArgs &a, unsigned n_upper, hipStream_t stream)
{
    const unsigned per_block = (unsigned)(kShadeBlock * kShadeItems);
    const unsigned grid = (n_upper + per_block - 1u) / per_block;
    if (grid == 0) return;
    hipLaunchKernelGGL(shade_kernel, dim3(grid), dim3(kShadeBlock), 0, stream, a);
}

void launch_accumulate(const PassDev &ps, const float *rad_r, const float *rad_g, const float *rad_b,
                       float *accum, hipStream_t stream)
{
    const unsigned npx = ps.tile_count * ps.tile_size * ps.tile_size;
    const unsigned grid = (npx + 255u) / 256u;
    if (grid == 0) return;
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid), dim3(256), 0, stream, ps, rad_r, rad_g, rad_b, accum);
}

void launch_resolve(const float *accum, unsigned long long npix, unsigned char *rgb, hipStream_t stream)
{
    const unsigned grid = (unsigned)((npix + 255ull) / 256ull);
    if (grid == 0) return;
    hipLaunchKernelGGL(resolve_kernel, dim3(grid), dim3(256), 0, stream, accum, npix, rgb);
}

int trace_blocks_per_cu()
{
    int best = 1;
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, trace_kernel<false, false, 1, QueueIO>, kBlock, 0) == hipSuccess && v > best) best = v;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, trace_kernel<true, false, 0, QueueIO>, kBlock, 0) == hipSuccess && v > best) best = v;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, trace_kernel<false, true, 1, StreamIO>, kBlock, 0) == hipSuccess && v > best) best = v;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, trace_kernel<true, false, 3, QueueIO>, kBlock, 0) == hipSuccess && v > best) best = v;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, trace_kernel<false, false, 3, PrimaryIO>, kBlock, 0) == hipSuccess && v > best) best = v;
    if (best > 8) best = 8;
    return best;
}

} // namespace rtc
