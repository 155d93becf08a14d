/* compat/bvh.hpp -- the reference's BVH API (reference bvh.hpp:14-37) over the C ABI.
 *
 *   constructTree  builds the device BVH (crt_scene_set + crt_bvh_build) and materialises the
 *                  reference-shaped host Node tree (same topology, boxes and leaf membership as
 *                  bvh.cu:12-273) for callers that inspect it.  One call builds the whole tree;
 *                  the caller-side box/axis pre-fill of newBuild (main.cu:33-136) is not needed
 *                  but harmless.
 *   refitTree      re-reads the SceneObjects through the leaves' pointers (crt_scene_update),
 *                  refits on the device (crt_bvh_refit) and grows the host boxes as bvh.cu:288-400.
 *   freeTree       frees every node once (the reference double-frees leaves and skips right
 *                  subtrees, bvh.cu:274-286) and releases the device BVH.
 */
#ifndef CRT_COMPAT_BVH_HPP
#define CRT_COMPAT_BVH_HPP
#include "SceneObjects.hpp"
#include <vector>
#include <deque>

class Node {
public:
    Node* parent = nullptr;
    Node* left = nullptr;
    Node* right = nullptr;
    bool isleaf = false;
    SceneObject* objs[64] = {};
    int numObjs = 0;
    float minX = 0, maxX = 0, minY = 0, maxY = 0, minZ = 0, maxZ = 0;
    int nodeNum = 0;
    float midpoint = 0;
    float longestAxis = 0;
};

Node* constructTree(std::vector<SceneObject*>& objects, Node*& currentNode, std::deque<Node*>& leafs, Node*& parentNode);
void refitTree(std::deque<Node*>& leafs);
void freeTree(Node*& root);
/* main.cu:33-136, kept for applications that call it: resets the root and builds */
void newBuild(Node*& root, std::deque<Node*>& leafs, std::vector<SceneObject*>& scene);
#endif
