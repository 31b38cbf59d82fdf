// Host BVH builder: replaces Embree's rtcCommitScene (scene.cpp:20-27, build quality HIGH) with a binned-SAH
// binary tree collapsed into a BVH4 (largest-area child opened first) whose nodes are emitted breadth-first — a
// prefix of the node array is the top of the tree, which is what the extend kernel stages into LDS.  Each node record
// carries the boxes of its four children (one 128-byte fetch per step).
#include "flatten.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace lj {

namespace {

struct Box {
    float lo[3], hi[3];
    Box() { for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -std::numeric_limits<float>::infinity(); } }
    void grow(const float *l, const float *h) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); } }
    void grow(const Box &b) { grow(b.lo, b.hi); }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct TmpNode {
    Box box;
    int left = -1, right = -1;   // TmpNode indices (inner)
    int first = 0, count = 0;    // leaf range into the order array
    int depth = 0;
};

struct Builder {
    const std::vector<BuildPrim> &prims;
    std::vector<int> order;
    std::vector<float> centroid;  // 3 per prim
    std::vector<TmpNode> tmp;
    int max_leaf, max_depth;

    int build(int first, int count, int depth) {
        int id = (int)tmp.size();
        tmp.emplace_back();
        Box box, cbox;
        for (int i = 0; i < count; i++) {
            int p = order[first + i];
            box.grow(prims[p].lo, prims[p].hi);
            cbox.grow(&centroid[3 * p], &centroid[3 * p]);
        }
        tmp[id].box = box; tmp[id].depth = depth;
        auto make_leaf = [&]() { tmp[id].first = first; tmp[id].count = count; return id; };
        if (count <= 1 || (depth >= max_depth && count <= 8)) return make_leaf();
        // binned SAH over the three axes
        constexpr int kBins = 16;
        float best_cost = std::numeric_limits<float>::infinity(); int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; axis++) {
            float c0 = cbox.lo[axis], c1 = cbox.hi[axis];
            if (!(c1 > c0)) continue;
            Box bins[kBins]; int counts[kBins] = {0};
            float scale = kBins / (c1 - c0);
            for (int i = 0; i < count; i++) {
                int p = order[first + i];
                int b = std::min(kBins - 1, std::max(0, (int)((centroid[3 * p + axis] - c0) * scale)));
                bins[b].grow(prims[p].lo, prims[p].hi); counts[b]++;
            }
            float right_area[kBins]; int right_count[kBins];
            Box acc; int n = 0;
            for (int b = kBins - 1; b > 0; b--) { acc.grow(bins[b]); n += counts[b]; right_area[b] = acc.half_area(); right_count[b] = n; }
            acc = Box(); n = 0;
            for (int b = 0; b < kBins - 1; b++) {
                acc.grow(bins[b]); n += counts[b];
                if (n == 0 || right_count[b + 1] == 0) continue;
                float cost = acc.half_area() * n + right_area[b + 1] * right_count[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        const float leaf_cost = box.half_area() * count;
        int mid;
        if (best_axis < 0) {
            if (count <= max_leaf) return make_leaf();
            mid = first + count / 2;  // all centroids coincide: split the list in half
        } else {
            // 1.0 box-test cost vs 1.2 primitive-test cost, both children boxes are tested in the parent
            const float split_cost = 1.0f * box.half_area() + best_cost * 1.2f;
            if (count <= max_leaf && leaf_cost * 1.2f <= split_cost) return make_leaf();
            float c0 = cbox.lo[best_axis], c1 = cbox.hi[best_axis];
            float scale = kBins / (c1 - c0);
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int p) {
                int b = std::min(kBins - 1, std::max(0, (int)((centroid[3 * p + best_axis] - c0) * scale)));
                return b <= best_bin;
            });
            mid = (int)(it - order.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        int l = build(first, mid - first, depth + 1);
        int r = build(mid, first + count - mid, depth + 1);
        tmp[id].left = l; tmp[id].right = r;
        return id;
    }
};

} // namespace

void build_bvh(const std::vector<BuildPrim> &prims, int max_leaf, int max_depth,
               std::vector<ljd::DNode4> &nodes, std::vector<int> &leaf_order, int &depth_out) {
    nodes.clear(); leaf_order.clear(); depth_out = 0;
    const int n = (int)prims.size();
    const float inf = std::numeric_limits<float>::infinity();
    auto empty_node = [&]() {
        ljd::DNode4 nd{};
        for (int k = 0; k < 4; k++) { nd.lox[k] = nd.loy[k] = nd.loz[k] = inf; nd.hix[k] = nd.hiy[k] = nd.hiz[k] = -inf; nd.child[k] = 0; }
        return nd;
    };
    if (n == 0) { nodes.push_back(empty_node()); depth_out = 1; return; }
    Builder b{prims, {}, {}, {}, max_leaf, max_depth};
    b.order.resize(n); b.centroid.resize(3 * (size_t)n);
    for (int i = 0; i < n; i++) { b.order[i] = i; for (int k = 0; k < 3; k++) b.centroid[3 * i + k] = 0.5f * (prims[i].lo[k] + prims[i].hi[k]); }
    b.tmp.reserve(2 * (size_t)n);
    const int root = b.build(0, n, 0);
    // Leaves hold at most 8 primitives (3-bit count in the child code); beyond the depth cap the builder keeps
    // splitting until that holds, and the caller checks the resulting depth against its traversal stack.
    if ((long long)n * 8 >= (1ll << 30)) throw LjError(LJ_ERR_UNSUPPORTED, "too many primitives for the 30-bit leaf code");
    leaf_order = b.order;
    auto is_leaf = [&](int id) { return b.tmp[id].left < 0; };
    auto set_child = [&](ljd::DNode4 &nd, int k, int id, int code) {
        const Box &bx = b.tmp[id].box;
        nd.lox[k] = bx.lo[0]; nd.loy[k] = bx.lo[1]; nd.loz[k] = bx.lo[2];
        nd.hix[k] = bx.hi[0]; nd.hiy[k] = bx.hi[1]; nd.hiz[k] = bx.hi[2];
        nd.child[k] = code;
    };
    auto leaf_code = [&](int id) { return ~(b.tmp[id].first * 8 + b.tmp[id].count - 1); };
    if (is_leaf(root)) {
        ljd::DNode4 nd = empty_node();
        set_child(nd, 0, root, leaf_code(root));
        nodes.push_back(nd); depth_out = 1;
        return;
    }
    // collapse: a wide node starts from the two children of a binary node and opens its largest inner child until it
    // has four; breadth-first numbering of the wide nodes
    struct Wide { int kids[4]; int n; int depth; };
    std::vector<Wide> wide;
    std::vector<int> wide_of(b.tmp.size(), -1);  // binary node id -> wide node index (for the roots of wide nodes)
    std::vector<int> queue{root};
    wide_of[root] = 0;
    std::vector<int> qdepth{1};
    for (size_t h = 0; h < queue.size(); h++) {
        const TmpNode &t = b.tmp[queue[h]];
        Wide w; w.kids[0] = t.left; w.kids[1] = t.right; w.n = 2; w.depth = qdepth[h];
        while (w.n < 4) {
            int pick = -1; float area = -1.0f;
            for (int k = 0; k < w.n; k++) if (!is_leaf(w.kids[k])) { const float a = b.tmp[w.kids[k]].box.half_area(); if (a > area) { area = a; pick = k; } }
            if (pick < 0) break;
            const TmpNode &c = b.tmp[w.kids[pick]];
            w.kids[pick] = c.left; w.kids[w.n++] = c.right;
        }
        for (int k = 0; k < w.n; k++) if (!is_leaf(w.kids[k])) { wide_of[w.kids[k]] = (int)queue.size(); queue.push_back(w.kids[k]); qdepth.push_back(w.depth + 1); }
        wide.push_back(w);
        depth_out = std::max(depth_out, w.depth);
    }
    nodes.resize(wide.size());
    for (size_t h = 0; h < wide.size(); h++) {
        ljd::DNode4 nd = empty_node();
        for (int k = 0; k < wide[h].n; k++) { const int id = wide[h].kids[k]; set_child(nd, k, id, is_leaf(id) ? leaf_code(id) : wide_of[id]); }
        nodes[h] = nd;
    }
}

} // namespace lj
