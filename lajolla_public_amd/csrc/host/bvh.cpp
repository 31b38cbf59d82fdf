// Host BVH builder: replaces Embree's rtcCommitScene (scene.cpp:20-27, build quality HIGH) with a binned-SAH
// binary tree with spatial splits (SBVH) collapsed into a BVH4 (largest-area child opened first) whose nodes are emitted breadth-first — a
// prefix of the node array is the top of the tree, which is what the extend kernel stages into LDS.  Each node record
// carries the boxes of its four children (one 128-byte fetch per step).
#include "flatten.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <cstdlib>
#include <cstdio>

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

// One reference to a primitive: its (possibly clipped) padded box.  Object splits move references; a spatial split cuts the ones
// that straddle its plane in two (Stich et al., "Spatial Splits in Bounding Volume Hierarchies", HPG 2009 — what Embree's
// RTC_BUILD_QUALITY_HIGH, which the reference asks for at scene.cpp:22, builds).
struct Ref { Box box; int prim; };

struct Builder {
    const std::vector<BuildPrim> &prims;
    std::vector<int> order;       // leaf order: primitive ids, one leaf after the other
    std::vector<TmpNode> tmp;
    int max_leaf, max_depth;
    bool spatial;                 // spatial splits allowed at all
    size_t ref_budget;            // total references the tree may hold (duplication cap)
    size_t refs_total;
    float root_area;
    float alpha = 1e-5f, gain = 1.0f;   // overlap threshold (fraction of the root's area) and the factor a spatial split has to beat the object split by
    float prim_cost = 1.2f;            // SAH: cost of a primitive test against 1.0 for a box test (LJ_TUNE_PRIM_COST)

    static void pad(Box &b) {     // the builder's box padding (flatten.cpp): a box test may accept a box the exact ray misses, never the reverse
        for (int k = 0; k < 3; k++) {
            const float l = b.lo[k], h = b.hi[k];
            const float p = 1e-5f * (std::fabs(l) + std::fabs(h)) + 1e-7f * (h - l) + 1e-30f;
            b.lo[k] = l - p; b.hi[k] = h + p;
        }
    }
    // box of the part of reference r inside the slab [a, b] of `axis`: the triangle clipped (in double), or the box cut for a sphere
    Box clip(const Ref &r, int axis, float a, float b) const {
        Box out;
        const BuildPrim &p = prims[r.prim];
        if (!p.tri) {
            out = r.box; out.lo[axis] = std::max(out.lo[axis], a); out.hi[axis] = std::min(out.hi[axis], b);
            return out;
        }
        double poly[2][10][3]; int n = 3, cur = 0;
        for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) poly[0][i][k] = p.v[i][k];
        for (int side = 0; side < 2 && n > 0; side++) {   // keep x >= a, then x <= b (Sutherland-Hodgman)
            const double plane = side == 0 ? (double)a : (double)b, sgn = side == 0 ? 1.0 : -1.0;
            int m = 0;
            for (int i = 0; i < n; i++) {
                const double *P = poly[cur][i], *Q = poly[cur][(i + 1) % n];
                const double dp = sgn * (P[axis] - plane), dq = sgn * (Q[axis] - plane);
                if (dp >= 0) { for (int k = 0; k < 3; k++) poly[cur ^ 1][m][k] = P[k]; m++; }
                if ((dp > 0 && dq < 0) || (dp < 0 && dq > 0)) {
                    const double t = dp / (dp - dq);
                    for (int k = 0; k < 3; k++) poly[cur ^ 1][m][k] = P[k] + t * (Q[k] - P[k]);
                    poly[cur ^ 1][m][axis] = plane; m++;
                }
            }
            cur ^= 1; n = m;
        }
        if (n == 0) return out;   // (empty)
        for (int i = 0; i < n; i++) { float q[3] = {(float)poly[cur][i][0], (float)poly[cur][i][1], (float)poly[cur][i][2]}; out.grow(q, q); }
        // the builder's padding around the clipped part (it covers the float rounding of the clipped vertices too: a child may reach
        // past the split plane by that hair), never beyond the reference's own box
        pad(out);
        for (int k = 0; k < 3; k++) { out.lo[k] = std::max(out.lo[k], r.box.lo[k]); out.hi[k] = std::min(out.hi[k], r.box.hi[k]); }
        return out;
    }

    int build(std::vector<Ref> &refs, int depth) {
        const int id = (int)tmp.size();
        tmp.emplace_back();
        const int count = (int)refs.size();
        Box box, cbox;
        for (const Ref &r : refs) { box.grow(r.box); float c[3] = {0.5f * (r.box.lo[0] + r.box.hi[0]), 0.5f * (r.box.lo[1] + r.box.hi[1]), 0.5f * (r.box.lo[2] + r.box.hi[2])}; cbox.grow(c, c); }
        tmp[id].box = box; tmp[id].depth = depth;
        auto make_leaf = [&]() { tmp[id].first = (int)order.size(); tmp[id].count = count; for (const Ref &r : refs) order.push_back(r.prim); return id; };
        if (count <= 1 || (depth >= max_depth && count <= 8)) return make_leaf();
        constexpr int kBins = 16;
        // ---- object split: binned SAH over the reference centroids, three axes
        float best_cost = std::numeric_limits<float>::infinity(); int best_axis = -1, best_bin = -1;
        Box best_l, best_r;
        for (int axis = 0; axis < 3; axis++) {
            const float c0 = cbox.lo[axis], c1 = cbox.hi[axis];
            if (!(c1 > c0)) continue;
            Box bins[kBins]; int counts[kBins] = {0};
            const float scale = kBins / (c1 - c0);
            for (const Ref &r : refs) {
                const float c = 0.5f * (r.box.lo[axis] + r.box.hi[axis]);
                const int b = std::min(kBins - 1, std::max(0, (int)((c - c0) * scale)));
                bins[b].grow(r.box); counts[b]++;
            }
            Box rbox[kBins]; int right_count[kBins];
            Box acc; int n = 0;
            for (int b = kBins - 1; b > 0; b--) { acc.grow(bins[b]); n += counts[b]; rbox[b] = acc; right_count[b] = n; }
            acc = Box(); n = 0;
            for (int b = 0; b < kBins - 1; b++) {
                acc.grow(bins[b]); n += counts[b];
                if (n == 0 || right_count[b + 1] == 0) continue;
                const float cost = acc.half_area() * n + rbox[b + 1].half_area() * right_count[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; best_l = acc; best_r = rbox[b + 1]; }
            }
        }
        // ---- spatial split: binned over the node's extent, references clipped to the bins they span.  Tried only where the object
        // split leaves its children overlapping by more than 1e-5 of the root's area (the SBVH restriction), and while the budget lasts.
        float sp_cost = std::numeric_limits<float>::infinity(); int sp_axis = -1; float sp_plane = 0;
        if (spatial && refs_total < ref_budget && count > max_leaf) {
            bool try_it = best_axis < 0;
            if (!try_it) {
                Box ov; for (int k = 0; k < 3; k++) { ov.lo[k] = std::max(best_l.lo[k], best_r.lo[k]); ov.hi[k] = std::min(best_l.hi[k], best_r.hi[k]); }
                try_it = ov.half_area() > alpha * root_area;
            }
            for (int axis = 0; try_it && axis < 3; axis++) {
                const float a0 = box.lo[axis], a1 = box.hi[axis];
                if (!(a1 > a0)) continue;
                Box bins[kBins]; int entry[kBins] = {0}, exit_[kBins] = {0};
                const float width = (a1 - a0) / kBins, scale = kBins / (a1 - a0);
                for (const Ref &r : refs) {
                    int b0 = std::min(kBins - 1, std::max(0, (int)((r.box.lo[axis] - a0) * scale)));
                    int b1 = std::min(kBins - 1, std::max(b0, (int)((r.box.hi[axis] - a0) * scale)));
                    entry[b0]++; exit_[b1]++;
                    if (b0 == b1) { bins[b0].grow(r.box); continue; }
                    for (int b = b0; b <= b1; b++) {
                        const Box c = clip(r, axis, a0 + b * width, b == kBins - 1 ? a1 : a0 + (b + 1) * width);
                        if (c.lo[0] <= c.hi[0]) bins[b].grow(c);
                    }
                }
                Box rbox[kBins]; int right_count[kBins];
                Box acc; int n = 0;
                for (int b = kBins - 1; b > 0; b--) { acc.grow(bins[b]); n += exit_[b]; rbox[b] = acc; right_count[b] = n; }
                acc = Box(); n = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.grow(bins[b]); n += entry[b];
                    if (n == 0 || right_count[b + 1] == 0 || n == count || right_count[b + 1] == count) continue;   // (a split that only duplicates is no split)
                    const float cost = acc.half_area() * n + rbox[b + 1].half_area() * right_count[b + 1];
                    if (cost < sp_cost) { sp_cost = cost; sp_axis = axis; sp_plane = a0 + (b + 1) * width; }
                }
            }
        }
        const float leaf_cost = box.half_area() * count;
        std::vector<Ref> left, right;
        if (sp_axis >= 0 && sp_cost < gain * best_cost) {   // the spatial split has to pay for its duplicates
            for (const Ref &r : refs) {
                if (r.box.hi[sp_axis] <= sp_plane) left.push_back(r);
                else if (r.box.lo[sp_axis] >= sp_plane) right.push_back(r);
                else {
                    Ref a{clip(r, sp_axis, r.box.lo[sp_axis], sp_plane), r.prim}, b{clip(r, sp_axis, sp_plane, r.box.hi[sp_axis]), r.prim};
                    const bool va = a.box.lo[0] <= a.box.hi[0], vb = b.box.lo[0] <= b.box.hi[0];
                    if (va) left.push_back(a);
                    if (vb) right.push_back(b);
                    if (!va && !vb) left.push_back(r);
                }
            }
            if (left.empty() || right.empty() || left.size() == refs.size() || right.size() == refs.size()) { left.clear(); right.clear(); }
            else refs_total += left.size() + right.size() - refs.size();
        }
        if (left.empty()) {
            if (best_axis < 0) {
                if (count <= max_leaf) return make_leaf();
                left.assign(refs.begin(), refs.begin() + count / 2); right.assign(refs.begin() + count / 2, refs.end());   // all centroids coincide: halve the list
            } else {
                // 1.0 box-test cost vs 1.2 primitive-test cost, both children boxes are tested in the parent
                const float split_cost = 1.0f * box.half_area() + best_cost * prim_cost;
                if (count <= max_leaf && leaf_cost * prim_cost <= split_cost) return make_leaf();
                const float c0 = cbox.lo[best_axis], c1 = cbox.hi[best_axis], scale = kBins / (c1 - c0);
                for (const Ref &r : refs) {
                    const float c = 0.5f * (r.box.lo[best_axis] + r.box.hi[best_axis]);
                    const int b = std::min(kBins - 1, std::max(0, (int)((c - c0) * scale)));
                    (b <= best_bin ? left : right).push_back(r);
                }
                if (left.empty() || right.empty()) { left.assign(refs.begin(), refs.begin() + count / 2); right.assign(refs.begin() + count / 2, refs.end()); }
            }
        }
        std::vector<Ref>().swap(refs);   // (release before recursing)
        const int l = build(left, depth + 1);
        const int r = build(right, depth + 1);
        tmp[id].left = l; tmp[id].right = r;
        return id;
    }
};

} // namespace

// ---- the binary tree collapsed eight wide, children quantised on a per-node grid (device/dtypes.h DNode8)
namespace {

// Which binary nodes become the children of a wide node.  `kids_of(n, W)` with W = 8 or 4: the children of the wide node rooted at binary
// inner node n.  Two rules:
//  greedy (default): start from n's two children and keep opening the inner child with the largest surface area;
//  optimal (LJ_TUNE_COLLAPSE=1; Ylitie, Karras, Laine 2017, section 4.1, for fixed leaves): a dynamic programme over the binary tree that minimises the
//  summed surface area of all wide nodes — the expected number of node visits of a random ray — cost[n][i] being the cheapest way to cover
//  n's subtree with at most i roots (wide nodes or leaves) that hang under one parent:
//      cost[leaf][i] = 0                                   (leaves are the same in every collapse)
//      cost[n][1]    = area(n) + split[n][W]               (n is a wide node)
//      cost[n][i]    = min(split[n][i], cost[n][i - 1])    with split[n][j] = min over k of cost[left][k] + cost[right][j - k]
//  Measured (round 3): on the host twin's path rays the optimal collapse saves 6 % of an extension ray's BVH8 node steps and 3 % of its BVH4
//  steps (sponza 12.27 -> 11.56 / 16.98 -> 16.47; 13 % fewer nodes), on the GPU nothing (sponza 256 spp 197.9 vs 195.1 ms, disney_bsdf 83.1 vs
//  82.6, matpreview 60.1 vs 61.2): the extend kernels are not bound by their step count.  So the greedy rule stays.
struct Collapse {
    const std::vector<TmpNode> &tmp;
    int W; bool optimal;
    std::vector<float> cost;            // [node][i - 1], i = 1 .. W
    std::vector<unsigned char> how;     // [node][i - 1]: 0 = as cost[n][i - 1]; k >= 1 = left gets k roots, right i - k; for i == 1: unused
    Collapse(const std::vector<TmpNode> &t, int width, int root) : tmp(t), W(width) {
        optimal = getenv("LJ_TUNE_COLLAPSE") && atoi(getenv("LJ_TUNE_COLLAPSE")) != 0;
        if (!optimal) return;
        cost.assign(tmp.size() * (size_t)W, 0.0f); how.assign(tmp.size() * (size_t)W, 0);
        // post-order without recursion (the tree can be 40 levels deep, but wide scenes have millions of nodes)
        std::vector<int> order, stack{root};
        while (!stack.empty()) { const int n = stack.back(); stack.pop_back(); order.push_back(n); if (tmp[n].left >= 0) { stack.push_back(tmp[n].left); stack.push_back(tmp[n].right); } }
        for (size_t q = order.size(); q-- > 0;) {
            const int n = order[q];
            if (tmp[n].left < 0) continue;   // leaf: all zero
            const float *cl = &cost[(size_t)tmp[n].left * W], *cr = &cost[(size_t)tmp[n].right * W];
            float split[9]; unsigned char arg[9];
            for (int j = 2; j <= W; j++) {
                float best = std::numeric_limits<float>::infinity(); int bk = 1;
                for (int k = 1; k < j; k++) { const float c = cl[k - 1] + cr[j - k - 1]; if (c < best) { best = c; bk = k; } }
                split[j] = best; arg[j] = (unsigned char)bk;
            }
            float *c = &cost[(size_t)n * W]; unsigned char *h = &how[(size_t)n * W];
            c[0] = tmp[n].box.half_area() + split[W]; h[0] = 0;
            for (int i = 2; i <= W; i++) {
                if (split[i] < c[i - 2]) { c[i - 1] = split[i]; h[i - 1] = arg[i]; }
                else { c[i - 1] = c[i - 2]; h[i - 1] = 0; }
            }
        }
    }
    void expand(int n, int i, std::vector<int> &out) const {   // the roots that cover n's subtree within a budget of i
        if (tmp[n].left < 0 || i == 1) { out.push_back(n); return; }
        const unsigned char k = how[(size_t)n * W + i - 1];
        if (k == 0) { expand(n, i - 1, out); return; }
        expand(tmp[n].left, k, out); expand(tmp[n].right, i - k, out);
    }
    void kids_of(int n, int *kids, int &count) const {
        if (optimal) {
            // the wide node rooted at n distributes its W slots over its two subtrees as split[n][W] decided: that is how[n][W - 1] unless
            // cost[n][W] fell back to fewer roots — the split for the node's OWN children is recomputed here from the children's tables
            const float *cl = &cost[(size_t)tmp[n].left * W], *cr = &cost[(size_t)tmp[n].right * W];
            float best = std::numeric_limits<float>::infinity(); int bk = 1;
            for (int k = 1; k < W; k++) { const float c = cl[k - 1] + cr[W - k - 1]; if (c < best) { best = c; bk = k; } }
            std::vector<int> out;
            expand(tmp[n].left, bk, out); expand(tmp[n].right, W - bk, out);
            count = (int)out.size();
            for (int k = 0; k < count; k++) kids[k] = out[k];
            return;
        }
        kids[0] = tmp[n].left; kids[1] = tmp[n].right; count = 2;
        while (count < W) {
            int pick = -1; float area = -1.0f;
            for (int k = 0; k < count; k++) if (tmp[kids[k]].left >= 0) { const float a = tmp[kids[k]].box.half_area(); if (a > area) { area = a; pick = k; } }
            if (pick < 0) break;
            const TmpNode &c = tmp[kids[pick]];
            kids[pick] = c.left; kids[count++] = c.right;
        }
    }
};

// grid step exponent for one axis: the smallest e with 255 * 2^e >= extent (so every plane of the node fits 8 bits)
int grid_exponent(double extent) {
    int e = -126;
    if (extent > 0) { e = (int)std::ceil(std::log2(extent / 255.0)); if (e < -126) e = -126; }
    while (std::ldexp(255.0, e) < extent) e++;
    if (e > 127) throw LjError(LJ_ERR_UNSUPPORTED, "scene extent beyond the float range of the BVH grid");
    return e;
}

} // namespace

void build_bvh(const std::vector<BuildPrim> &prims, int max_leaf, int max_depth,
               std::vector<ljd::DNode4> &nodes, std::vector<ljd::DNode8> &nodes8, std::vector<int> &leaf_order, int &depth_out, int &depth8_out) {
    nodes.clear(); nodes8.clear(); leaf_order.clear(); depth_out = 0; depth8_out = 0;
    const int n = (int)prims.size();
    const float inf = std::numeric_limits<float>::infinity();
    auto empty_node = [&]() {
        ljd::DNode4 nd{};
        for (int k = 0; k < 4; k++) { nd.lox[k] = nd.loy[k] = nd.loz[k] = inf; nd.hix[k] = nd.hiy[k] = nd.hiz[k] = -inf; nd.child[k] = 0; }
        return nd;
    };
    auto empty_node8 = [&]() {
        ljd::DNode8 nd{};
        for (int k = 0; k < 3; k++) nd.e[k] = 1;
        for (int k = 0; k < 8; k++) { nd.qlo_x[k] = nd.qlo_y[k] = nd.qlo_z[k] = 255; nd.qhi_x[k] = nd.qhi_y[k] = nd.qhi_z[k] = 0; }
        return nd;
    };
    if (n == 0) { nodes.push_back(empty_node()); nodes8.push_back(empty_node8()); depth_out = 1; depth8_out = 1; return; }
    // Spatial splits for scenes beyond the tiny ones (those run the flat leaf scan of mega.hip, which indexes primitives with 16 bits
    // and wants no duplicates); at most twice as many references as primitives (sponza ends at 1.36x: -16 % node steps and -30 %
    // primitive tests per extension ray against the object-split tree).  LJ_TUNE_SBVH=0 turns them off.
    const bool spatial = n > 256 && !(getenv("LJ_TUNE_SBVH") && atoi(getenv("LJ_TUNE_SBVH")) == 0);
    double budget = 1.0;
    if (const char *e = getenv("LJ_TUNE_SBVH_BUDGET")) budget = atof(e);
    if (max_leaf > 4) max_leaf = 4;   // (a DNode8 addresses the primitives of its leaf children with 5-bit offsets: 8 leaves x 4)
    Builder b{prims, {}, {}, max_leaf, max_depth, spatial, (size_t)n + (size_t)(budget * n), (size_t)n, 0.0f};
    std::vector<Ref> refs((size_t)n);
    Box rootb;
    for (int i = 0; i < n; i++) { refs[i].prim = i; for (int k = 0; k < 3; k++) { refs[i].box.lo[k] = prims[i].lo[k]; refs[i].box.hi[k] = prims[i].hi[k]; } rootb.grow(refs[i].box); }
    b.root_area = rootb.half_area();
    if (const char *e = getenv("LJ_TUNE_SBVH_ALPHA")) b.alpha = (float)atof(e);
    if (const char *e = getenv("LJ_TUNE_SBVH_GAIN")) b.gain = (float)atof(e);
    if (const char *e = getenv("LJ_TUNE_PRIM_COST")) b.prim_cost = (float)atof(e);
    b.order.reserve((size_t)n + n / 3);
    b.tmp.reserve(2 * (size_t)n);
    const int root = b.build(refs, 0);
    if ((long long)n * 8 >= (1ll << 30)) throw LjError(LJ_ERR_UNSUPPORTED, "too many primitives for the 30-bit leaf code");
    // Beyond the depth cap the builder stops at leaves of up to 8 primitives (coincident primitives cannot be separated); the wide
    // nodes want at most 4 per leaf, so such a leaf becomes an inner node over two leaves with its own box.
    for (size_t id = 0; id < b.tmp.size(); id++) {
        if (b.tmp[id].left >= 0 || b.tmp[id].count <= 4) continue;
        TmpNode l = b.tmp[id], r = b.tmp[id];
        l.count = 4; r.first = l.first + 4; r.count = b.tmp[id].count - 4; l.depth = r.depth = b.tmp[id].depth + 1;
        b.tmp[id].left = (int)b.tmp.size(); b.tmp.push_back(l);
        b.tmp[id].right = (int)b.tmp.size(); b.tmp.push_back(r);
    }
    auto is_leaf = [&](int id) { return b.tmp[id].left < 0; };

    // ---- BVH8 first: it fixes the leaf order (the primitives of a wide node's leaf children are consecutive, in slot order)
    std::vector<int> new_first(b.tmp.size(), -1);   // binary leaf id -> position of its first primitive in leaf_order
    {
        struct Wide8 { int kids[8]; int n; int depth; };
        std::vector<int> queue{root}, qdepth{1};
        const bool root_leaf = is_leaf(root);
        const Collapse collapse8(b.tmp, 8, root);
        for (size_t h = 0; h < queue.size(); h++) {
            Wide8 w; w.depth = qdepth[h];
            if (root_leaf) { w.kids[0] = root; w.n = 1; }
            else {
                collapse8.kids_of(queue[h], w.kids, w.n);
            }
            depth8_out = std::max(depth8_out, w.depth);
            // grid: origin = lower corner of the union, one power-of-two step per axis
            Box u; for (int k = 0; k < w.n; k++) u.grow(b.tmp[w.kids[k]].box);
            ljd::DNode8 nd = empty_node8();
            double step[3];
            for (int a = 0; a < 3; a++) {
                nd.p[a] = u.lo[a];
                const int e = grid_exponent((double)u.hi[a] - (double)u.lo[a]);
                nd.e[a] = (uint8_t)(e + 127); step[a] = std::ldexp(1.0, e);
            }
            // octant-ordered slots: child c goes to the free slot s that maximises (centre_c - centre_node) . corner_s, best pairs first
            int slot_of[8]; bool slot_used[8] = {false}, kid_done[8] = {false};
            double ctr[3]; for (int a = 0; a < 3; a++) ctr[a] = 0.5 * ((double)u.lo[a] + (double)u.hi[a]);
            for (int round = 0; round < w.n; round++) {
                double best = -std::numeric_limits<double>::infinity(); int bc = -1, bs = -1;
                for (int c = 0; c < w.n; c++) if (!kid_done[c]) {
                    const Box &cb = b.tmp[w.kids[c]].box;
                    for (int s = 0; s < 8; s++) if (!slot_used[s]) {
                        double v = 0;
                        for (int a = 0; a < 3; a++) v += (0.5 * ((double)cb.lo[a] + (double)cb.hi[a]) - ctr[a]) * ((s >> a) & 1 ? 1.0 : -1.0);
                        if (v > best) { best = v; bc = c; bs = s; }
                    }
                }
                slot_of[bc] = bs; slot_used[bs] = true; kid_done[bc] = true;
            }
            int kid_at[8]; for (int s = 0; s < 8; s++) kid_at[s] = -1;
            for (int c = 0; c < w.n; c++) kid_at[slot_of[c]] = w.kids[c];
            nd.child_base = (uint32_t)queue.size(); nd.prim_base = (uint32_t)leaf_order.size();
            for (int s = 0; s < 8; s++) {
                const int id = kid_at[s];
                if (id < 0) continue;
                const Box &cb = b.tmp[id].box;
                uint8_t *qlo[3] = {nd.qlo_x, nd.qlo_y, nd.qlo_z}, *qhi[3] = {nd.qhi_x, nd.qhi_y, nd.qhi_z};
                for (int a = 0; a < 3; a++) {   // lower planes down, upper planes up, checked in exact arithmetic (q * step is exact in double)
                    const double p = nd.p[a];
                    long lo = (long)std::floor(((double)cb.lo[a] - p) / step[a]), hi = (long)std::ceil(((double)cb.hi[a] - p) / step[a]);
                    lo = std::min(255l, std::max(0l, lo)); hi = std::min(255l, std::max(0l, hi));
                    while (lo > 0 && p + lo * step[a] > (double)cb.lo[a]) lo--;
                    while (hi < 255 && p + hi * step[a] < (double)cb.hi[a]) hi++;
                    if (p + lo * step[a] > (double)cb.lo[a] || p + hi * step[a] < (double)cb.hi[a]) throw LjError(LJ_ERR_INTERNAL, "BVH8 grid does not contain a child box");
                    qlo[a][s] = (uint8_t)lo; qhi[a][s] = (uint8_t)hi;
                }
                if (is_leaf(id)) {
                    const TmpNode &t = b.tmp[id];
                    const uint32_t off = (uint32_t)leaf_order.size() - nd.prim_base;
                    nd.meta[s] = (uint8_t)(0x80u | ((uint32_t)(t.count - 1) << 5) | off);
                    new_first[id] = (int)leaf_order.size();
                    for (int i = 0; i < t.count; i++) leaf_order.push_back(b.order[t.first + i]);
                } else {
                    nd.imask |= (uint8_t)(1u << s);
                    queue.push_back(id); qdepth.push_back(w.depth + 1);
                }
            }
            nodes8.push_back(nd);
            if (root_leaf) break;
        }
        if (nodes8.size() >= (1u << 24)) throw LjError(LJ_ERR_UNSUPPORTED, "too many BVH8 nodes for the 24-bit node index");
    }

    // ---- BVH4 over the same leaves
    auto set_child = [&](ljd::DNode4 &nd, int k, int id, int code) {
        const Box &bx = b.tmp[id].box;
        nd.lox[k] = bx.lo[0]; nd.loy[k] = bx.lo[1]; nd.loz[k] = bx.lo[2];
        nd.hix[k] = bx.hi[0]; nd.hiy[k] = bx.hi[1]; nd.hiz[k] = bx.hi[2];
        nd.child[k] = code;
    };
    auto leaf_code = [&](int id) { return ~(new_first[id] * 8 + b.tmp[id].count - 1); };
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
    const Collapse collapse4(b.tmp, 4, root);
    for (size_t h = 0; h < queue.size(); h++) {
        Wide w; w.depth = qdepth[h];
        collapse4.kids_of(queue[h], w.kids, w.n);
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
