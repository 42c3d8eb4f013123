// The two per-node computations of the wide tree, written once for the host (wide_bvh.cpp) and the
// GPU (bvh_build_gpu.hip): which descendants of a binary node become the children of its wide node,
// and the node's 8-bit grid.  Both sides run the same operations in the same order (no fused
// multiply-add on either: -ffp-contract=off), so a tree collapsed and quantized on the device is
// the host's bit for bit.
#pragma once

#include <cmath>
#include <cstdint>

#include "device_layout.h"

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define RTC_HD __host__ __device__
#else
#define RTC_HD
#endif

namespace rtc {

struct WideSlot {
    uint32_t ref; // packed reference (DevNode encoding): a leaf, or a binary node
    float lo[3], hi[3];
    RTC_HD float area() const
    {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return 2.f * (x * y + x * z + y * z);
    }
};

// The children of the wide node over binary node `bin`: children of inner children are pulled up,
// largest surface area first, until there are four; then ordered by surface area, largest first
// (equal areas keep their order).  tree.children(bin, left, right) gives the two child slots of a
// binary node.  Returns how many slots are used.
template <class Tree>
RTC_HD int collapse_slots(const Tree &tree, uint32_t bin, WideSlot (&c)[4])
{
    int n = 2;
    tree.children(bin, c[0], c[1]);
    while (n < 4) {
        int pick = -1;
        float best = -1.f;
        for (int i = 0; i < n; i++) {
            if (c[i].ref & kLeafBit) continue;
            const float a = c[i].area();
            if (pick < 0 || a > best) {
                pick = i;
                best = a;
            }
        }
        if (pick < 0) break;
        // the two children take the place of the opened child
        for (int i = n - 1; i > pick; i--) c[i + 1] = c[i];
        const uint32_t opened = c[pick].ref;
        tree.children(opened, c[pick], c[pick + 1]);
        n++;
    }
    // Slot order is the order the any-hit walk visits hit children in.  Left to right (the
    // binary tree's order, which follows the split axis whatever the ray's direction) measured
    // 3-20 % slower there than any order by box size; largest first is the one kept.  The
    // closest-hit walk orders by entry distance and only falls back to slots on ties.
    // (A stable insertion sort: what std::stable_sort gives.)
    for (int i = 1; i < n; i++) {
        const WideSlot x = c[i];
        const float ax = x.area();
        int j = i;
        while (j > 0 && ax > c[j - 1].area()) {
            c[j] = c[j - 1];
            j--;
        }
        c[j] = x;
    }
    return n;
}

// The slots as a DevWideNode (unused slots: the empty box and the reference of slot 0).
RTC_HD inline void fill_wide_node(const WideSlot (&c)[4], int n, uint32_t bin, DevWideNode &w)
{
    const float inf = __builtin_inff();
    for (int i = 0; i < 4; i++) {
        const bool used = i < n;
        for (int k = 0; k < 3; k++) {
            w.lo[k][i] = used ? c[i].lo[k] : inf;
            w.hi[k][i] = used ? c[i].hi[k] : -inf;
        }
        w.child[i] = used ? c[i].ref : c[0].ref; // n < 4 only when every child is a leaf
    }
    w.bin = bin;
    w.pad[0] = w.pad[1] = w.pad[2] = 0;
}

RTC_HD inline int wide_children_used(const DevWideNode &w)
{
    int n = 0;
    while (n < 4 && w.lo[0][n] < __builtin_inff() && w.lo[0][n] > -__builtin_inff()) n++;
    return n;
}

RTC_HD inline double quant_step_of(uint32_t c) { return ldexp(1.0 + 0.25 * (double)(c & 3u), (int)(c >> 2) - 127); }

// One node on the 8-bit grid: origin, step codes and the lo/hi plane bytes of DevQNode (base4 and
// meta are the caller's: they depend on where the children are placed).  See quantize_node() in
// wide_bvh.hpp for the rounding rules.
RTC_HD inline void quantize_planes(const DevWideNode &src, int n, DevQNode &q)
{
    uint32_t exps = 0;
    for (int k = 0; k < 3; k++) {
        double nlo = src.lo[k][0], nhi = src.hi[k][0];
        for (int i = 1; i < n; i++) {
            const double l = src.lo[k][i], h = src.hi[k][i];
            nlo = l < nlo ? l : nlo;
            nhi = h > nhi ? h : nhi;
        }
        const double ext = nhi - nlo;
        double big = fabs(nlo) > fabs(nhi) ? fabs(nlo) : fabs(nhi);
        big = big > ext ? big : ext;
        double margin = ldexp(big, -18);
        if (!(margin > 0.0)) margin = ldexp(1.0, -100); // a node at the coordinate origin with no extent
        // origin: a float at or below nlo - margin
        float origin = (float)(nlo - margin);
        if ((double)origin > nlo - margin) origin = nextafterf(origin, -__builtin_inff());
        // step: the smallest float with a 2-bit mantissa (2^e * {1, 1.25, 1.5, 1.75}, ten bits in the
        // node) with origin + 255 * step >= nhi + margin
        const double want = (nhi + margin - (double)origin) / 255.0;
        int e = 0;
        (void)frexp(want, &e); // want = m * 2^e, m in [0.5, 1)
        e = e - 1;
        e = e < -126 ? -126 : (e > 127 ? 127 : e);
        uint32_t code = (uint32_t)(e + 127) << 2; // exponent << 2 | mantissa
        while (code < ((254u << 2) | 3u) && (double)origin + 255.0 * quant_step_of(code) < nhi + margin) code++;
        const double step = quant_step_of(code);
        q.origin[k] = origin;
        exps |= code << (10 * k);
        uint32_t lo_word = 0, hi_word = 0;
        for (int i = 0; i < 4; i++) {
            uint32_t bl = 255, bh = 0; // unused slot: near plane beyond far plane on every axis
            if (i < n) {
                const double lo = (double)src.lo[k][i] - margin, hi = (double)src.hi[k][i] + margin;
                double fl = floor((lo - (double)origin) / step);
                fl = fl < 0.0 ? 0.0 : (fl > 255.0 ? 255.0 : fl);
                while (fl > 0.0 && (double)origin + fl * step > lo + 0.5 * margin) fl -= 1.0;
                double ch = ceil((hi - (double)origin) / step);
                ch = ch < 0.0 ? 0.0 : (ch > 255.0 ? 255.0 : ch);
                while (ch < 255.0 && (double)origin + ch * step < hi - 0.5 * margin) ch += 1.0;
                bl = (uint32_t)fl;
                bh = (uint32_t)ch;
            }
            lo_word |= bl << (8 * i);
            hi_word |= bh << (8 * i);
        }
        q.lo[k] = lo_word;
        q.hi[k] = hi_word;
    }
    q.exps = exps;
}

// base4 and meta of a node from the units of its children (unit_of_child(i) for i < n; tris_of_child(i) = 0
// for an inner child).
template <class UnitOf, class TrisOf>
RTC_HD void quantize_children(int n, UnitOf unit_of_child, TrisOf tris_of_child, DevQNode &q)
{
    uint32_t base = 0xFFFFFFFFu, meta = 0;
    for (int i = 0; i < n; i++) {
        const uint32_t unit = unit_of_child(i), tris = tris_of_child(i);
        if (i == 0) base = unit;
        meta |= (((unit - base) << 2) | tris) << (8 * i);
    }
    for (int i = n; i < 4; i++) meta |= (meta & 0xFFu) << (8 * i);
    q.base4 = base << 2;
    q.meta = meta;
}

} // namespace rtc
