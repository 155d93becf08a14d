/* bvh_build.h -- host BVH construction.
 *
 * Two builders feed the packed node formats of layout.h:
 *   - build_reference(): the reference's topology -- top-down split on the node's longest axis at
 *     the MEAN of the child centroids, stable partition `centroid <= midpoint -> left`, leaf at
 *     <= 64 primitives (main.cu:33-136 newBuild, bvh.cu:12-273 constructTree).  It also yields
 *     the reference's leaf order, which defines the tie-break rank of every primitive.
 *   - build_wide(): binned-SAH binary tree over the same primitives, collapsed to 8-wide nodes
 *     (Node8) with <= 4 primitives per leaf and children placed in octant slots.  This is what the
 *     render kernels use by default; nearest hits are independent of tree shape once ties resolve
 *     by reference rank.
 */
#ifndef CRT_BVH_BUILD_H
#define CRT_BVH_BUILD_H
#include <vector>
#include <stdint.h>
#include "../../include/crt.h"
#include "layout.h"

namespace crt {

/* Reference binary tree in index form (replaces the pointer tree of bvh.hpp:14-33). */
struct RefNode {
    float minX, maxX, minY, maxY, minZ, maxZ;
    int parent, left, right;
    bool isleaf;
    int first, count;          /* into RefTree::order (leaf order, DFS left first) */
    float midpoint;
    float longestAxis;
};
struct RefTree {
    std::vector<RefNode> nodes;        /* nodes[0] is the root; creation order = nodeNum order */
    std::vector<int> order;            /* scene indices in leaf order */
    std::vector<int> leafs;            /* node indices in the order constructTree pushes them */
    int depth = 0;
};

struct HostBVH {
    std::vector<Node4> nodes;              /* CRT_BVH_REFERENCE */
    std::vector<Node8> nodes8;             /* CRT_BVH_WIDE */
    size_t node_count() const { return mode == CRT_BVH_WIDE ? nodes8.size() : nodes.size(); }
    std::vector<uint32_t> slot_to_scene;
    std::vector<uint32_t> rank_of_scene;   /* reference leaf-order position of each scene index */
    float root_box[6];
    float origin_bound = 0.f;      /* wide trees: what the sphere boxes were padded for (layout.h sphere_box_half) */
    int depth = 0, leaves = 0, mode = 0;
};

/* returns 0, or CRT_ERR_BUILD when the reference's recursion would not terminate */
int build_reference_tree(const crt_prim* prims, int n, RefTree& out);
void refit_reference_tree(const crt_prim* prims, RefTree& t);          /* bvh.cu:288-400 on the index tree */
int pack_reference(const crt_prim* prims, int n, const RefTree& t, HostBVH& out);
int build_wide(const crt_prim* prims, int n, const RefTree* ref_or_null, HostBVH& out);
float origin_bound_of(const crt_prim* prims, int n);

}  // namespace crt
#endif
