#include "wide_bvh.hpp"

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

} // namespace rtc
