#include "wide_bvh.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace rtc {

namespace {

struct Slot {
    uint32_t ref;
    float lo[3], hi[3];
    float area() const
    {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return 2.f * (x * y + x * z + y * z);
    }
};

// child boxes of a binary node: lanes {minL, minR, maxL, maxR}
void children_of(const DevNode &n, Slot &l, Slot &r)
{
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

struct Pending {
    uint32_t bin;    // binary node to collapse
    uint32_t parent; // wide node whose slot refers to it, or ~0u for the root
    uint32_t slot;
    uint32_t depth;
};

} // namespace

uint32_t collapse_wide4(const DevNode *bn, size_t num_nodes, uint32_t root_ref, std::vector<DevWideNode> &out,
                        uint32_t *max_wide_depth)
{
    if (max_wide_depth) *max_wide_depth = 0;
    if ((root_ref & kLeafBit) || num_nodes == 0) return root_ref;
    const float inf = std::numeric_limits<float>::infinity();
    out.reserve(out.size() + num_nodes / 2 + 1);
    const uint32_t root_index = (uint32_t)out.size();
    std::vector<Pending> todo;
    todo.push_back(Pending{root_ref, 0xFFFFFFFFu, 0, 1});
    uint32_t deepest = 0;
    while (!todo.empty()) {
        const Pending job = todo.back();
        todo.pop_back();
        const uint32_t self = (uint32_t)out.size();
        out.emplace_back();
        if (job.parent != 0xFFFFFFFFu) out[job.parent].child[job.slot] = self;
        if (job.depth > deepest) deepest = job.depth;

        Slot c[4];
        int n = 2;
        children_of(bn[job.bin], c[0], c[1]);
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
            // the two children take the place of the opened child, keeping the left-to-right order
            for (int i = n - 1; i > pick; i--) c[i + 1] = c[i];
            const uint32_t opened = c[pick].ref;
            children_of(bn[opened], c[pick], c[pick + 1]);
            n++;
        }

        DevWideNode &w = out[self];
        for (int i = 0; i < 4; i++) {
            const bool used = i < n;
            for (int k = 0; k < 3; k++) {
                w.lo[k][i] = used ? c[i].lo[k] : inf;
                w.hi[k][i] = used ? c[i].hi[k] : -inf;
            }
            w.child[i] = used ? c[i].ref : c[0].ref; // n < 4 only when every child is a leaf
        }
        w.bin = job.bin;
        w.pad[0] = w.pad[1] = w.pad[2] = 0;
        // inner children become wide nodes; the first one is numbered next (pre-order)
        for (int i = n - 1; i >= 0; i--)
            if (!(c[i].ref & kLeafBit)) todo.push_back(Pending{c[i].ref, self, (uint32_t)i, job.depth + 1});
    }
    if (max_wide_depth) *max_wide_depth = deepest;
    return root_index;
}

void quantize_wide4(const std::vector<DevWideNode> &wide, std::vector<DevQWideNode> &out,
                    std::vector<uint32_t> &bin)
{
    out.resize(wide.size());
    bin.resize(wide.size());
    for (size_t w = 0; w < wide.size(); w++) {
        const DevWideNode &src = wide[w];
        DevQWideNode &q = out[w];
        bin[w] = src.bin;
        int n = 0;
        while (n < 4 && std::isfinite(src.lo[0][n])) n++;
        for (int k = 0; k < 3; k++) {
            double nlo = src.lo[k][0], nhi = src.hi[k][0];
            for (int i = 1; i < n; i++) {
                nlo = std::min<double>(nlo, src.lo[k][i]);
                nhi = std::max<double>(nhi, src.hi[k][i]);
            }
            const double ext = nhi - nlo;
            double margin = std::ldexp(std::max({std::fabs(nlo), std::fabs(nhi), ext}), -18);
            if (!(margin > 0.0)) margin = std::ldexp(1.0, -100); // a node at the coordinate origin with no extent
            // origin: a float at or below nlo - margin
            float origin = (float)(nlo - margin);
            if ((double)origin > nlo - margin) origin = std::nextafterf(origin, -std::numeric_limits<float>::infinity());
            // scale: a float with origin + 255 * scale >= nhi + margin
            float scale = (float)((nhi + margin - (double)origin) / 255.0);
            while ((double)origin + 255.0 * (double)scale < nhi + margin || !(scale > 0.f))
                scale = std::nextafterf(scale, std::numeric_limits<float>::infinity());
            q.origin[k] = origin;
            q.scale[k] = scale;
            uint32_t lo_word = 0, hi_word = 0;
            for (int i = 0; i < 4; i++) {
                uint32_t bl = 255, bh = 0; // unused slot: near plane beyond far plane on every axis
                if (i < n) {
                    const double lo = (double)src.lo[k][i] - margin, hi = (double)src.hi[k][i] + margin;
                    double fl = std::floor((lo - (double)origin) / (double)scale);
                    fl = std::min(255.0, std::max(0.0, fl));
                    while (fl > 0.0 && (double)origin + fl * (double)scale > lo + 0.5 * margin) fl -= 1.0;
                    double ch = std::ceil((hi - (double)origin) / (double)scale);
                    ch = std::min(255.0, std::max(0.0, ch));
                    while (ch < 255.0 && (double)origin + ch * (double)scale < hi - 0.5 * margin) ch += 1.0;
                    bl = (uint32_t)fl;
                    bh = (uint32_t)ch;
                }
                lo_word |= bl << (8 * i);
                hi_word |= bh << (8 * i);
            }
            q.lo[k] = lo_word;
            q.hi[k] = hi_word;
        }
        for (int i = 0; i < 4; i++) q.child[i] = src.child[i];
    }
}

} // namespace rtc
