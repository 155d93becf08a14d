/* bvh_build.cpp -- see bvh_build.h.  Compiled with -ffp-contract=off: the reference-topology
 * builder must reproduce the reference's float comparisons (bvh.cu:58-68, 205-253). */
#include "bvh_build.h"
#include <algorithm>
#include <limits>
#include <cmath>
#include <cstring>
#include <thread>
#include <memory>
#include <functional>
#include <cstdlib>

namespace crt {

namespace {

constexpr float kFmax = std::numeric_limits<float>::max();
constexpr float kFminPos = std::numeric_limits<float>::min();   /* smallest positive: the reference's max-init */

struct Centroid { float c[3]; };

/* sphere centre / per-axis vertex average exactly as bvh.cu:61-66,130 */
inline Centroid centroid_of(const crt_prim& o) {
    Centroid r;
    if (o.type == 0) { r.c[0] = o.position[0]; r.c[1] = o.position[1]; r.c[2] = o.position[2]; }
    else for (int a = 0; a < 3; a++) r.c[a] = (o.v1[a] + o.v2[a] + o.v3[a]) / 3.0f;
    return r;
}

/* strict grow updates of main.cu:45-104 / bvh.cu:70-129 / bvh.cu:301-360 */
inline void grow(RefNode& n, const crt_prim& o) {
    if (o.type == 0) {
        float lo[3] = {o.position[0] - o.radius, o.position[1] - o.radius, o.position[2] - o.radius};
        float hi[3] = {o.position[0] + o.radius, o.position[1] + o.radius, o.position[2] + o.radius};
        if (lo[0] < n.minX) n.minX = lo[0];
        if (lo[1] < n.minY) n.minY = lo[1];
        if (lo[2] < n.minZ) n.minZ = lo[2];
        if (hi[0] > n.maxX) n.maxX = hi[0];
        if (hi[1] > n.maxY) n.maxY = hi[1];
        if (hi[2] > n.maxZ) n.maxZ = hi[2];
    } else if (o.type == 1) {
        const float* vs[3] = {o.v1, o.v2, o.v3};
        for (const float* v : vs) {
            if (v[0] < n.minX) n.minX = v[0];
            if (v[1] < n.minY) n.minY = v[1];
            if (v[2] < n.minZ) n.minZ = v[2];
            if (v[0] > n.maxX) n.maxX = v[0];
            if (v[1] > n.maxY) n.maxY = v[1];
            if (v[2] > n.maxZ) n.maxZ = v[2];
        }
    }
}
inline void box_reset(RefNode& n) {
    n.maxX = n.maxY = n.maxZ = kFminPos;
    n.minX = n.minY = n.minZ = kFmax;
}
/* bvh.cu:205-253: strict comparisons; on a tie nothing is assigned (fields stay 0) */
inline void choose_axis(RefNode& n, const float mid[3]) {
    float ex = n.maxX - n.minX, ey = n.maxY - n.minY, ez = n.maxZ - n.minZ;
    if (ex > ey && ex > ez) { n.longestAxis = 0; n.midpoint = mid[0]; }
    if (ey > ex && ey > ez) { n.longestAxis = 1; n.midpoint = mid[1]; }
    if (ez > ex && ez > ey) { n.longestAxis = 2; n.midpoint = mid[2]; }
}

/* Threads for the host builders: subtrees over disjoint primitive ranges are independent, so a node whose range is
 * larger than `cutoff` hands its left child to a new thread and keeps the right one.  The results do not depend on the
 * schedule: every node's own loop (the order-dependent float sums included) stays sequential, and the node numbering is
 * the serial one.  CRT_BUILD_THREADS=1 builds on the calling thread alone. */
int build_cutoff(int n) {
    if (const char* e = getenv("CRT_BUILD_THREADS")) if (atoi(e) == 1) return n;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw <= 1) return n;
    return std::max(n / 64, 8192);
}

/* The descendants of one node in the reference's creation order (bvh.cu:12-273 appends both children, then recurses into
 * the left and the right one): [left, right, descendants of left..., descendants of right...].  Links are LOCAL indices
 * into `nodes`; parent == -1 marks the two children of the block's owner. */
struct RefSub {
    std::vector<RefNode> nodes;
    int depth = 0;
};

struct RefBuilder {
    const crt_prim* prims;
    const std::vector<Centroid>& cen;
    std::vector<int>& order;
    int cutoff;
    RefBuilder(const crt_prim* p, const std::vector<Centroid>& c, std::vector<int>& o, int cut) : prims(p), cen(c), order(o), cutoff(cut) {}

    /* `node` (box, axis, midpoint set) covers order[first, first+count); its descendants go to `sub` */
    int split(RefNode& node, int first, int count, int depth, RefSub& sub, std::vector<int>& scratch) {
        if (depth > sub.depth) sub.depth = depth;
        if (count <= kMaxLeafPrims) {
            node.isleaf = true; node.first = first; node.count = count;
            return 0;
        }
        if (depth > 256) return CRT_ERR_BUILD;
        RefNode zero; std::memset(&zero, 0, sizeof(zero));
        zero.left = zero.right = -1; zero.parent = -1;
        RefNode L = zero, R = zero;
        box_reset(L); box_reset(R);
        const int axis = (int)node.longestAxis;
        const float mid = node.midpoint;
        /* stable partition; centroid sums accumulate in object order (bvh.cu:85,130,151,196) */
        float sumL[3] = {0, 0, 0}, sumR[3] = {0, 0, 0};
        int nl = 0, nr = 0;
        if ((int)scratch.size() < count) scratch.resize(count);
        int* ord = &order[first];
        for (int i = 0; i < count; i++) {
            int o = ord[i];
            if (cen[o].c[axis] <= mid) {
                grow(L, prims[o]);
                for (int a = 0; a < 3; a++) sumL[a] += cen[o].c[a];
                ord[nl++] = o;
            } else {
                grow(R, prims[o]);
                for (int a = 0; a < 3; a++) sumR[a] += cen[o].c[a];
                scratch[nr++] = o;
            }
        }
        std::copy(scratch.begin(), scratch.begin() + nr, ord + nl);
        float midL[3], midR[3];
        for (int a = 0; a < 3; a++) { midL[a] = sumL[a] / (float)(size_t)nl; midR[a] = sumR[a] / (float)(size_t)nr; }
        choose_axis(L, midL);
        choose_axis(R, midR);
        RefSub subL, subR;
        int rcL = 0, rcR = 0;
        if (count > cutoff) {
            std::thread th([&] { std::vector<int> sc; rcL = split(L, first, nl, depth + 1, subL, sc); });
            rcR = split(R, first + nl, nr, depth + 1, subR, scratch);
            th.join();
        } else {
            rcL = split(L, first, nl, depth + 1, subL, scratch);
            if (!rcL) rcR = split(R, first + nl, nr, depth + 1, subR, scratch);
        }
        if (rcL) return rcL;
        if (rcR) return rcR;
        /* this node's block: [L, R, block of L, block of R] */
        const int base = (int)sub.nodes.size();
        const int offL = base + 2, offR = offL + (int)subL.nodes.size();
        if (!L.isleaf) { L.left += offL; L.right += offL; }
        if (!R.isleaf) { R.left += offR; R.right += offR; }
        sub.nodes.reserve(sub.nodes.size() + 2 + subL.nodes.size() + subR.nodes.size());
        sub.nodes.push_back(L); sub.nodes.push_back(R);
        auto append = [&](const RefSub& c, int off, int owner) {
            for (RefNode nd : c.nodes) {
                nd.parent = nd.parent < 0 ? owner : nd.parent + off;
                if (!nd.isleaf) { nd.left += off; nd.right += off; }
                sub.nodes.push_back(nd);
            }
        };
        append(subL, offL, base); append(subR, offR, base + 1);
        node.left = base; node.right = base + 1;        /* relative to this block; the caller shifts them */
        sub.depth = std::max(sub.depth, std::max(subL.depth, subR.depth));
        return 0;
    }
};

/* box of a primitive in the wide tree: a sphere's holds the ball its hit predicate can reach (layout.h sphere_box_half) */
inline void prim_bounds(const crt_prim& o, float origin_bound, float lo[3], float hi[3]) {
    if (o.type == 0) {
        const float hs = sphere_box_half(o.position[0], o.position[1], o.position[2], o.radius, origin_bound);
        for (int a = 0; a < 3; a++) { lo[a] = o.position[a] - hs; hi[a] = o.position[a] + hs; }
    } else {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(o.v1[a], std::min(o.v2[a], o.v3[a]));
            hi[a] = std::max(o.v1[a], std::max(o.v2[a], o.v3[a]));
        }
    }
}

template <class N>
void node_clear(N& n) {
    std::memset(&n, 0, sizeof n);
    for (int k = 0; k < N::W; k++) {
        n.lox[k] = n.loy[k] = n.loz[k] = kFmax;
        n.hix[k] = n.hiy[k] = n.hiz[k] = -kFmax;
        n.child[k] = kChildEmpty;
    }
    n.parent = kChildEmpty; n.parent_slot = 0; n.n_internal = 0; n.pad = 0;
}
void node4_clear(Node4& n) { node_clear(n); }

}  // namespace

int build_reference_tree(const crt_prim* prims, int n, RefTree& t) {
    t.nodes.clear(); t.order.resize(n); t.leafs.clear(); t.depth = 0;
    for (int i = 0; i < n; i++) t.order[i] = i;
    std::vector<Centroid> cen(n);
    for (int i = 0; i < n; i++) cen[i] = centroid_of(prims[i]);
    /* newBuild, main.cu:33-136 */
    RefNode root; std::memset(&root, 0, sizeof(root));
    root.parent = -1; root.left = root.right = -1;
    box_reset(root);
    for (int i = 0; i < n; i++) grow(root, prims[i]);
    float mid[3] = {(root.maxX + root.minX) / 2, (root.maxY + root.minY) / 2, (root.maxZ + root.minZ) / 2};
    choose_axis(root, mid);
    RefBuilder b(prims, cen, t.order, build_cutoff(n));
    RefSub sub;
    std::vector<int> scratch;
    const int rc = b.split(root, 0, n, 0, sub, scratch);
    if (rc) { t.nodes.push_back(root); return rc; }
    /* the root is node 0; its block follows */
    t.nodes.reserve(sub.nodes.size() + 1);
    if (!root.isleaf) { root.left += 1; root.right += 1; }
    t.nodes.push_back(root);
    for (RefNode nd : sub.nodes) {
        nd.parent = nd.parent < 0 ? 0 : nd.parent + 1;
        if (!nd.isleaf) { nd.left += 1; nd.right += 1; }
        t.nodes.push_back(nd);
    }
    t.depth = sub.depth;
    /* leaves in the order the reference's recursion reaches them (left before right) */
    std::vector<int> stack{0};
    while (!stack.empty()) {
        const int ni = stack.back(); stack.pop_back();
        if (t.nodes[ni].isleaf) { t.leafs.push_back(ni); continue; }
        stack.push_back(t.nodes[ni].right); stack.push_back(t.nodes[ni].left);
    }
    return 0;
}

void refit_reference_tree(const crt_prim* prims, RefTree& t) {
    /* bvh.cu:288-400: for every leaf walk to the root; boxes only grow.  min/max are exact and
     * associative, so one bottom-up pass in reverse creation order gives the same boxes. */
    for (int ni = (int)t.nodes.size() - 1; ni >= 0; ni--) {
        RefNode& p = t.nodes[ni];
        if (p.isleaf) {
            for (int k = 0; k < p.count; k++) grow(p, prims[t.order[p.first + k]]);
        } else {
            const RefNode& l = t.nodes[p.left]; const RefNode& r = t.nodes[p.right];
            const RefNode* cs[2] = {&l, &r};
            for (const RefNode* c : cs) {
                if (c->minX < p.minX) p.minX = c->minX;
                if (c->maxX > p.maxX) p.maxX = c->maxX;
                if (c->minY < p.minY) p.minY = c->minY;
                if (c->maxY > p.maxY) p.maxY = c->maxY;
                if (c->minZ < p.minZ) p.minZ = c->minZ;
                if (c->maxZ > p.maxZ) p.maxZ = c->maxZ;
            }
        }
    }
}

static void ranks_from(const RefTree& t, int n, std::vector<uint32_t>& rank) {
    rank.assign(n, 0);
    for (int i = 0; i < n; i++) rank[t.order[i]] = (uint32_t)i;
}

/* Reference topology in the packed format: every internal reference node becomes a Node4 using
 * two child slots; reference leaves become leaf links (<=64 primitives); empty leaves become
 * kChildEmpty but keep their (inverted) box so that dumps compare with the reference's. */
int pack_reference(const crt_prim* prims, int n, const RefTree& t, HostBVH& out) {
    (void)prims;
    out.nodes.clear(); out.nodes8.clear(); out.mode = CRT_BVH_REFERENCE; out.depth = t.depth + 1; out.leaves = (int)t.leafs.size();
    out.slot_to_scene.resize(n);
    for (int i = 0; i < n; i++) out.slot_to_scene[i] = (uint32_t)t.order[i];   /* slot == reference rank */
    ranks_from(t, n, out.rank_of_scene);
    const RefNode& r = t.nodes[0];
    out.root_box[0] = r.minX; out.root_box[1] = r.maxX; out.root_box[2] = r.minY;
    out.root_box[3] = r.maxY; out.root_box[4] = r.minZ; out.root_box[5] = r.maxZ;
    auto set_box = [](Node4& nd, int k, const RefNode& c) {
        nd.lox[k] = c.minX; nd.hix[k] = c.maxX; nd.loy[k] = c.minY; nd.hiy[k] = c.maxY; nd.loz[k] = c.minZ; nd.hiz[k] = c.maxZ;
    };
    auto leaf_link = [](const RefNode& c) { return c.count > 0 ? make_leaf((uint32_t)c.first, (uint32_t)c.count) : kChildEmpty; };
    if (r.isleaf) {                     /* <= 64 objects: the whole scene is one leaf (Ray.cu:186-188) */
        Node4 nd; node4_clear(nd);
        set_box(nd, 0, r); nd.child[0] = leaf_link(r);
        out.nodes.push_back(nd);
        return 0;
    }
    /* map internal reference nodes to packed indices in creation order */
    std::vector<int> packed(t.nodes.size(), -1);
    int cnt = 0;
    for (size_t i = 0; i < t.nodes.size(); i++) if (!t.nodes[i].isleaf) packed[i] = cnt++;
    out.nodes.resize(cnt);
    for (size_t i = 0; i < t.nodes.size(); i++) {
        if (t.nodes[i].isleaf) continue;
        Node4& nd = out.nodes[packed[i]]; node4_clear(nd);
        const int ch[2] = {t.nodes[i].left, t.nodes[i].right};
        for (int k = 0; k < 2; k++) {
            const RefNode& c = t.nodes[ch[k]];
            set_box(nd, k, c);
            if (c.isleaf) nd.child[k] = leaf_link(c);
            else { nd.child[k] = (uint32_t)packed[ch[k]]; nd.n_internal++; }
        }
    }
    for (size_t i = 0; i < t.nodes.size(); i++) {
        if (t.nodes[i].isleaf) continue;
        const int ch[2] = {t.nodes[i].left, t.nodes[i].right};
        for (int k = 0; k < 2; k++)
            if (!t.nodes[ch[k]].isleaf) { Node4& c = out.nodes[packed[ch[k]]]; c.parent = (uint32_t)packed[i]; c.parent_slot = (uint32_t)k; }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Wide builder: binned SAH (16 bins, 3 axes) binary tree, then collapse to 8 children per node.
 * ------------------------------------------------------------------------------------------ */
namespace {

struct AABB {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = kFmax; hi[a] = -kFmax; } }
    void add(const float l[3], const float h[3]) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); } }
    void add(const AABB& b) { add(b.lo, b.hi); }
    void addp(const float p[3]) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.f;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

struct BNode { AABB box; int left, right; int first, count; };   /* leaf when left < 0 */

struct PrimRef { AABB box; float c[3]; int idx; };

constexpr int kBins = 16;
constexpr int kWideLeaf = 4;

/* SAH termination: a range of n <= max_leaf primitives becomes a leaf when n * 1 (cost of n
 * primitive tests) <= node_cost + (A_L n_L + A_R n_R) / A.  Tunables for experiments:
 * CRT_SAH_NODE_COST (default below), CRT_MAX_LEAF (1..4). */
struct WideTuning { float node_cost; int max_leaf; };
WideTuning wide_tuning() {
    WideTuning t{0.35f, kWideLeaf};
    if (const char* e = getenv("CRT_SAH_NODE_COST")) t.node_cost = (float)atof(e);
    if (const char* e = getenv("CRT_MAX_LEAF")) { int v = atoi(e); if (v >= 1 && v <= kWideLeaf) t.max_leaf = v; }
    return t;
}

/* Binary SAH tree in a preallocated array: the node of a range of `count` primitives sits at index `ni`, its left
 * subtree from ni + 1, its right subtree from ni + 2 * (primitives on the left) -- a subtree over c primitives has at
 * most 2c - 1 nodes, so the blocks never overlap and subtrees can be built by different threads without any exchange.
 * Unused entries are never read (the collapse follows links only). */
struct WideBuilder {
    std::vector<PrimRef>& refs;
    BNode* nodes;
    WideTuning tune;
    int cutoff;
    WideBuilder(std::vector<PrimRef>& r, BNode* storage, int cut) : refs(r), nodes(storage), tune(wide_tuning()), cutoff(cut) {}

    void build(int first, int count, int ni) {
        AABB box, cb; box.reset(); cb.reset();
        for (int i = first; i < first + count; i++) { box.add(refs[i].box); cb.addp(refs[i].c); }
        nodes[ni].box = box; nodes[ni].left = nodes[ni].right = -1; nodes[ni].first = first; nodes[ni].count = count;
        if (count <= 1) return;
        int bestAxis = -1, bestSplit = -1; float bestCost = kFmax;
        for (int a = 0; a < 3; a++) {
            float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.f)) continue;
            AABB bb[kBins]; int bc[kBins];
            for (int b = 0; b < kBins; b++) { bb[b].reset(); bc[b] = 0; }
            float scale = (float)kBins / ext;
            for (int i = first; i < first + count; i++) {
                int b = (int)((refs[i].c[a] - cb.lo[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bb[b].add(refs[i].box); bc[b]++;
            }
            float rightArea[kBins]; int rightCnt[kBins];
            AABB acc; acc.reset(); int c = 0;
            for (int b = kBins - 1; b > 0; b--) { acc.add(bb[b]); c += bc[b]; rightArea[b] = acc.area(); rightCnt[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < kBins - 1; b++) {
                acc.add(bb[b]); c += bc[b];
                if (c == 0 || rightCnt[b + 1] == 0) continue;
                float cost = acc.area() * (float)c + rightArea[b + 1] * (float)rightCnt[b + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = b; }
            }
        }
        if (count <= tune.max_leaf) {
            const float a = box.area();
            const float split_cost = (bestAxis >= 0 && a > 0.f) ? tune.node_cost + bestCost / a : kFmax;
            if ((float)count <= split_cost) return;
        }
        int mid;
        if (bestAxis < 0) {
            mid = first + count / 2;            /* all centroids coincide: split by count */
        } else {
            float ext = cb.hi[bestAxis] - cb.lo[bestAxis];
            float scale = (float)kBins / ext; float lo = cb.lo[bestAxis]; int a = bestAxis, s = bestSplit;
            auto it = std::partition(refs.begin() + first, refs.begin() + first + count, [&](const PrimRef& r) {
                int b = (int)((r.c[a] - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= s;
            });
            mid = (int)(it - refs.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        const int l = ni + 1, r = ni + 2 * (mid - first);
        if (count > cutoff) {
            std::thread th([&] { build(first, mid - first, l); });
            build(mid, first + count - mid, r);
            th.join();
        } else {
            build(first, mid - first, l);
            build(mid, first + count - mid, r);
        }
        nodes[ni].left = l; nodes[ni].right = r;
    }
};

}  // namespace

/* |o| of every ray that starts inside the scene or inside the reference's room (walls at +-200, +-200, +-300, Ray.cu:238-457):
 * sqrt(3) x the largest coordinate of either.  Only sets where the cost of sphere_box_half's cancellation noise goes -- into the
 * sphere boxes (rays from inside) or into the ray's own pad (rays from further out); any value is correct */
float origin_bound_of(const crt_prim* prims, int n) {
    float m = 300.f;
    for (int i = 0; i < n; i++) {
        const crt_prim& o = prims[i];
        if (o.type == 0) { for (int a = 0; a < 3; a++) m = std::max(m, std::fabs(o.position[a]) + o.radius); }
        else for (int a = 0; a < 3; a++) m = std::max(m, std::max(std::fabs(o.v1[a]), std::max(std::fabs(o.v2[a]), std::fabs(o.v3[a]))));
    }
    return 1.7320509f * 1.001f * m;
}

int build_wide(const crt_prim* prims, int n, const RefTree* ref, HostBVH& out) {
    out.nodes.clear(); out.nodes8.clear(); out.mode = CRT_BVH_WIDE; out.depth = 0; out.leaves = 0;
    out.slot_to_scene.clear(); out.slot_to_scene.reserve(n);
    if (ref && (int)ref->order.size() != n) return -1;   /* reference tree of another scene */
    if (ref) ranks_from(*ref, n, out.rank_of_scene);
    else { out.rank_of_scene.resize(n); for (int i = 0; i < n; i++) out.rank_of_scene[i] = (uint32_t)i; }

    out.origin_bound = origin_bound_of(prims, n);
    std::vector<PrimRef> refs(n);
    for (int i = 0; i < n; i++) {
        prim_bounds(prims[i], out.origin_bound, refs[i].box.lo, refs[i].box.hi);
        for (int a = 0; a < 3; a++) refs[i].c[a] = 0.5f * (refs[i].box.lo[a] + refs[i].box.hi[a]);
        refs[i].idx = i;
    }
    std::unique_ptr<BNode[]> storage(new BNode[(size_t)2 * (size_t)std::max(n, 1)]);      /* not initialised: see WideBuilder */
    WideBuilder wb(refs, storage.get(), build_cutoff(n));
    if (n > 0) wb.build(0, n, 0);

    Node8 rootn; node_clear(rootn);
    out.nodes8.push_back(rootn);
    AABB rb; rb.reset();
    if (n > 0) rb = wb.nodes[0].box;
    out.root_box[0] = rb.lo[0]; out.root_box[1] = rb.hi[0]; out.root_box[2] = rb.lo[1];
    out.root_box[3] = rb.hi[1]; out.root_box[4] = rb.lo[2]; out.root_box[5] = rb.hi[2];
    if (n == 0) return 0;

    /* collapse: (binary node, packed index, depth) */
    struct Item { int b; int p; int depth; };
    std::vector<Item> stack;
    stack.push_back({0, 0, 1});
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        if (it.depth > out.depth) out.depth = it.depth;
        int kids[8]; int nk = 0;
        const BNode& bn = wb.nodes[it.b];
        if (bn.left < 0) kids[nk++] = it.b;
        else { kids[nk++] = bn.left; kids[nk++] = bn.right; }
        while (nk < 8) {            /* open the child with the largest surface area */
            int best = -1; float bestA = -1.f;
            for (int k = 0; k < nk; k++) {
                const BNode& c = wb.nodes[kids[k]];
                if (c.left < 0) continue;
                float a = c.box.area();
                if (a > bestA) { bestA = a; best = k; }
            }
            if (best < 0) break;
            int c = kids[best];
            kids[best] = wb.nodes[c].left;
            kids[nk++] = wb.nodes[c].right;
        }
        /* octant slots: slot bit 4/2/1 = child centre on the +x/+y/+z side of the node centre.
         * Greedy assignment of (child, slot) pairs by the projection of the child's offset on the
         * slot's diagonal; the traversal visits hit children in descending (slot ^ ray octant). */
        float ctr[3];
        for (int a = 0; a < 3; a++) ctr[a] = 0.5f * (bn.box.lo[a] + bn.box.hi[a]);
        int slot_of_kid[8]; bool slot_used[8] = {false, false, false, false, false, false, false, false};
        bool kid_done[8] = {false, false, false, false, false, false, false, false};
        for (int round = 0; round < nk; round++) {
            int bi = -1, bs = -1; float bc = -kFmax;
            for (int i = 0; i < nk; i++) {
                if (kid_done[i]) continue;
                const AABB& cb = wb.nodes[kids[i]].box;
                float v[3];
                for (int a = 0; a < 3; a++) v[a] = 0.5f * (cb.lo[a] + cb.hi[a]) - ctr[a];
                for (int s = 0; s < 8; s++) {
                    if (slot_used[s]) continue;
                    const float c = ((s & 4) ? v[0] : -v[0]) + ((s & 2) ? v[1] : -v[1]) + ((s & 1) ? v[2] : -v[2]);
                    if (c > bc || bi < 0) { bc = c; bi = i; bs = s; }      /* bi < 0: a NaN offset still gets a slot (as lbvh.hip) */
                }
            }
            kid_done[bi] = true; slot_used[bs] = true; slot_of_kid[bi] = bs;
        }
        int kid_at[8];
        for (int s = 0; s < 8; s++) kid_at[s] = -1;
        for (int i = 0; i < nk; i++) kid_at[slot_of_kid[i]] = kids[i];

        /* internal children become consecutive nodes, leaf primitives one consecutive slot range, both in slot order */
        int childPacked[8];
        for (int s = 0; s < 8; s++) {
            childPacked[s] = -1;
            if (kid_at[s] < 0) continue;
            const BNode& c = wb.nodes[kid_at[s]];
            Node8& nd = out.nodes8[it.p];
            nd.lox[s] = c.box.lo[0]; nd.loy[s] = c.box.lo[1]; nd.loz[s] = c.box.lo[2];
            nd.hix[s] = c.box.hi[0]; nd.hiy[s] = c.box.hi[1]; nd.hiz[s] = c.box.hi[2];
            if (c.left >= 0) {
                childPacked[s] = (int)out.nodes8.size();
                Node8 nn; node_clear(nn); nn.parent = (uint32_t)it.p; nn.parent_slot = (uint32_t)s;
                out.nodes8.push_back(nn);
                out.nodes8[it.p].child[s] = (uint32_t)childPacked[s];
                out.nodes8[it.p].n_internal++;
            }
        }
        uint32_t node_prims = 0;
        for (int s = 0; s < 8; s++) {
            if (kid_at[s] < 0) continue;
            const BNode& c = wb.nodes[kid_at[s]];
            if (c.left < 0) {
                if (c.count > kWide8LeafMax) return CRT_ERR_BUILD;
                uint32_t start = (uint32_t)out.slot_to_scene.size();
                for (int i = c.first; i < c.first + c.count; i++) out.slot_to_scene.push_back((uint32_t)refs[i].idx);
                out.nodes8[it.p].child[s] = make_leaf(start, (uint32_t)c.count);
                out.leaves++;
                node_prims += (uint32_t)c.count;
            }
        }
        if (node_prims > 32u) return CRT_ERR_BUILD;       /* QNode8::meta holds a 5-bit offset */
        for (int s = 7; s >= 0; s--)
            if (childPacked[s] >= 0) stack.push_back({kid_at[s], childPacked[s], it.depth + 1});
    }
    if (out.nodes8.size() >= (1u << 24)) return CRT_ERR_BUILD;   /* QNode8::base_vmask keeps 24 bits for the index */
    return 0;
}

}  // namespace crt
