/* compat/compat.cpp -- the reference's C++ entry points (same names, argument meaning and
 * call order as Ray.hpp:35, bvh.hpp:35-37, objloader.hpp:4-9) as thin wrappers over the C ABI of
 * include/crt.h.  Nothing is computed here: scene marshalling and bookkeeping only. */
#include "Ray.hpp"
#include "objloader.hpp"
#include "../../../include/crt.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <functional>

int numBounces = 2;

namespace {
crt_mgpu* g_mgpu = nullptr;                  /* every visible GPU (CRT_GPUS=n limits it): rank 0 owns the scene */
crt_ctx* g_ctx = nullptr;                    /* = crt_mgpu_ctx(g_mgpu, 0) */
std::string g_err;
std::vector<SceneObject*> g_scene;          /* objects of the tree built last, in scene order */
std::vector<Node*> g_nodes;                 /* ownership of the host Node tree */

crt_ctx* ctx() {
    if (!g_ctx) {
        int err = 0;
        /* the reference is single-GPU; here the frame's tiles are sharded over all GPUs of the node (RCCL broadcast of the
         * scene, tile gather to GPU 0).  CRT_GPUS=n uses the first n devices; with one GPU no communicator is created */
        int want = 0;
        if (const char* e = getenv("CRT_GPUS")) want = atoi(e);
        std::vector<int> devs;
        for (int d = 0; d < want; d++) devs.push_back(d);
        g_mgpu = crt_mgpu_create(devs.empty() ? nullptr : devs.data(), (int)devs.size(), 0, &err);
        if (!g_mgpu && devs.size() != 1) {
            /* the multi-GPU layer could not come up (librccl missing, ncclCommInitAll failed ...): the reference is a
             * single-GPU program, so one GPU -- which loads no RCCL at all -- is always a valid way to serve it */
            fprintf(stderr, "CudaRayTracer compat: multi-GPU set-up failed (status %d); rendering on GPU 0 only\n", err);
            const int one = 0;
            g_mgpu = crt_mgpu_create(&one, 1, 0, &err);
        }
        if (g_mgpu) g_ctx = crt_mgpu_ctx(g_mgpu, 0);
        if (!g_ctx) g_err = "crt_mgpu_create failed (no HIP device?)";
    }
    return g_ctx;
}
bool ok(int rc, const char* what) {
    if (rc == CRT_OK) { g_err.clear(); return true; }
    g_err = std::string(what) + ": " + (g_ctx ? crt_last_error(g_ctx) : "no context");
    fprintf(stderr, "%s\n", g_err.c_str());
    return false;
}
bool okm(int rc, const char* what) {
    if (rc == CRT_OK) { g_err.clear(); return true; }
    g_err = std::string(what) + ": " + crt_mgpu_last_error(g_mgpu);
    fprintf(stderr, "%s\n", g_err.c_str());
    return false;
}
void put(float* d, const glm::vec3& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

void marshal(const std::vector<SceneObject*>& scene, std::vector<crt_prim>& out) {
    out.resize(scene.size());
    for (size_t i = 0; i < scene.size(); i++) {
        const SceneObject& s = *scene[i];
        crt_prim& p = out[i];
        memset(&p, 0, sizeof p);
        p.type = s.sphere ? 0 : 1;              /* bvhTraverse tests `sphere` first (Ray.cu:192,209) */
        p.objNum = s.objNum; p.radius = s.radius; p.shininess = s.shininess; p.speed = s.speed;
        put(p.position, s.position); put(p.v1, s.v1); put(p.v2, s.v2); put(p.v3, s.v3);
        put(p.v1Norm, s.v1Norm); put(p.v2Norm, s.v2Norm); put(p.v3Norm, s.v3Norm);
        put(p.ambient, s.ambient); put(p.diffuse, s.diffuse); put(p.specular, s.specular);
        put(p.reflective, s.reflective); put(p.direction, s.direction);
    }
}
void free_nodes(Node* keep) {
    for (Node* n : g_nodes) if (n != keep) delete n;
    g_nodes.clear();
}
void grow(Node* n, float lo0, float lo1, float lo2, float hi0, float hi1, float hi2) {
    if (lo0 < n->minX) n->minX = lo0;
    if (lo1 < n->minY) n->minY = lo1;
    if (lo2 < n->minZ) n->minZ = lo2;
    if (hi0 > n->maxX) n->maxX = hi0;
    if (hi1 > n->maxY) n->maxY = hi1;
    if (hi2 > n->maxZ) n->maxZ = hi2;
}
}  // namespace

const char* crtCompatLastError() { return g_err.c_str(); }

bool loadOBJ(const char* path, std::vector<glm::vec3>& out_vertices, std::vector<glm::vec2>& out_uvs,
             std::vector<glm::vec3>& out_normals) {
    (void)out_uvs;                               /* never filled by the reference either */
    printf("Loading OBJ file %s...\n", path);
    int n = crt_load_obj(path, nullptr, nullptr, 0);
    if (n < 0) { printf("Impossible to open or parse the file %s\n", path); return false; }
    std::vector<float> v(3 * (size_t)n), nm(3 * (size_t)n);
    if (crt_load_obj(path, v.data(), nm.data(), n) < 0) return false;
    for (int i = 0; i < n; i++) {
        out_vertices.push_back(glm::vec3(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
        out_normals.push_back(glm::vec3(nm[3 * i], nm[3 * i + 1], nm[3 * i + 2]));
    }
    return true;
}

Node* constructTree(std::vector<SceneObject*>& objects, Node*& currentNode, std::deque<Node*>& leafs, Node*& parentNode) {
    (void)parentNode;
    if (!ctx() || !currentNode) return currentNode;
    std::vector<crt_prim> prims;
    marshal(objects, prims);
    const int n = (int)prims.size();
    if (!ok(crt_scene_set(g_ctx, prims.data(), n), "constructTree/crt_scene_set")) return currentNode;
    if (!ok(crt_bvh_build(g_ctx, CRT_BVH_WIDE), "constructTree/crt_bvh_build")) return currentNode;
    if (!okm(crt_mgpu_sync_scene(g_mgpu), "constructTree/crt_mgpu_sync_scene")) return currentNode;
    g_scene = objects;
    /* reference-shaped host tree for callers that walk it */
    int32_t nn = 0, depth = 0;
    if (crt_host_bvh(prims.data(), n, CRT_BVH_REFERENCE, &nn, &depth, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0) != CRT_OK)
        return currentNode;
    std::vector<float> boxes((size_t)nn * CRT_BVH_DUMP_WIDTH * 6), root(6);
    std::vector<uint32_t> links((size_t)nn * CRT_BVH_DUMP_WIDTH), parents((size_t)nn * 2), s2s(n), rank(n);
    crt_host_bvh(prims.data(), n, CRT_BVH_REFERENCE, &nn, &depth, root.data(), boxes.data(), links.data(), parents.data(),
                 s2s.data(), rank.data(), nn);
    free_nodes(currentNode);
    g_nodes.push_back(currentNode);
    leafs.clear();
    auto set_box = [](Node* nd, const float* b) { nd->minX = b[0]; nd->maxX = b[1]; nd->minY = b[2]; nd->maxY = b[3]; nd->minZ = b[4]; nd->maxZ = b[5]; };
    auto make_leaf = [&](Node* nd, uint32_t link) {
        nd->isleaf = true; nd->left = nd->right = nullptr; nd->numObjs = 0;
        if (link != 0xffffffffu) {
            const uint32_t start = link & ((1u << 25) - 1u), cnt = ((link >> 25) & 63u) + 1u;
            for (uint32_t k = 0; k < cnt; k++) nd->objs[k] = objects[s2s[start + k]];
            nd->numObjs = (int)cnt;
        }
        leafs.push_back(nd);
    };
    Node* rootn = currentNode;
    rootn->parent = nullptr; rootn->nodeNum = 0;
    set_box(rootn, root.data());
    int counter = 1;
    if (nn == 1 && (links[0] & 0x80000000u) && links[1] == 0xffffffffu && n <= 64) {
        make_leaf(rootn, links[0]);             /* <= 64 objects: the root is the only leaf */
        return rootn;
    }
    /* left subtree first, so `leafs` fills in the DFS order of bvh.cu:22 and nodeNum counts as bvh.cu:29,37 */
    std::function<void(Node*, uint32_t)> expand = [&](Node* host, uint32_t packed) {
        host->isleaf = false;
        Node* kids[2];
        for (int k = 0; k < 2; k++) {
            kids[k] = new Node(); g_nodes.push_back(kids[k]);
            kids[k]->parent = host; kids[k]->nodeNum = counter++;
            set_box(kids[k], &boxes[((size_t)packed * CRT_BVH_DUMP_WIDTH + k) * 6]);
        }
        host->left = kids[0]; host->right = kids[1];
        for (int k = 0; k < 2; k++) {
            const uint32_t link = links[(size_t)packed * CRT_BVH_DUMP_WIDTH + k];
            if (link == 0xffffffffu || (link & 0x80000000u)) make_leaf(kids[k], link);
            else expand(kids[k], link);
        }
    };
    expand(rootn, 0);
    return rootn;
}

void newBuild(Node*& root, std::deque<Node*>& leafs, std::vector<SceneObject*>& scene) {
    if (!root) root = new Node();
    constructTree(scene, root, leafs, root);
}

void refitTree(std::deque<Node*>& leafs) {
    if (!ctx() || g_scene.empty()) return;
    std::vector<crt_prim> prims;
    marshal(g_scene, prims);
    if (!ok(crt_scene_update(g_ctx, prims.data(), (int)prims.size()), "refitTree/crt_scene_update")) return;
    if (!ok(crt_bvh_refit(g_ctx, CRT_REFIT_TIGHT), "refitTree/crt_bvh_refit")) return;
    if (!okm(crt_mgpu_sync_scene(g_mgpu), "refitTree/crt_mgpu_sync_scene")) return;
    /* host boxes: bvh.cu:288-400, grow only */
    for (Node* leaf : leafs) {
        for (Node* p = leaf; p; p = p->parent) {
            if (p->isleaf) {
                for (int o = 0; o < p->numObjs; o++) {
                    const SceneObject* s = p->objs[o];
                    if (s->triangle) {
                        const glm::vec3* vs[3] = {&s->v1, &s->v2, &s->v3};
                        for (const glm::vec3* v : vs) grow(p, v->x, v->y, v->z, v->x, v->y, v->z);
                    }
                    if (s->sphere)
                        grow(p, s->position.x - s->radius, s->position.y - s->radius, s->position.z - s->radius,
                             s->position.x + s->radius, s->position.y + s->radius, s->position.z + s->radius);
                }
            } else {
                for (Node* c : {p->left, p->right}) grow(p, c->minX, c->minY, c->minZ, c->maxX, c->maxY, c->maxZ);
            }
        }
    }
}

void freeTree(Node*& root) {
    free_nodes(nullptr);                         /* includes the root: the reference frees it too (bvh.cu:285) */
    root = nullptr;
    g_scene.clear();
    if (g_ctx) ok(crt_bvh_free(g_ctx), "freeTree/crt_bvh_free");
}

void startRayTracing(float width, float height, unsigned char*& pixelcolorBuffer, glm::vec3 cameraPosition,
                     glm::vec3 cameraDirection, std::vector<SceneObject*>& scene, std::vector<Light>& lights, Node* rootnode) {
    (void)scene; (void)rootnode;                 /* the reference does not read `scene` either (Ray.cu:665-742) */
    if (!ctx()) return;
    std::vector<crt_light> ls(lights.size());
    for (size_t i = 0; i < lights.size(); i++) {
        memset(&ls[i], 0, sizeof(crt_light));
        put(ls[i].color, lights[i].color); put(ls[i].direction, lights[i].direction); put(ls[i].position, lights[i].position);
        ls[i].point = lights[i].point; ls[i].area = lights[i].area; ls[i].radius = lights[i].radius;
        ls[i].constantTerm = lights[i].constantTerm; ls[i].linearTerm = lights[i].linearTerm; ls[i].quadraticTerm = lights[i].quadraticTerm;
    }
    if (!okm(crt_mgpu_lights_set(g_mgpu, ls.data(), (int)ls.size()), "startRayTracing/crt_mgpu_lights_set")) return;
    crt_params p; crt_params_default(&p);
    p.width = (int)width; p.height = (int)height; p.bounces = numBounces;
    p.use_graph = 1;        /* the application's camera and frame size rarely change (main.cu:459,508): replay the launches */
    crt_camera cam; put(cam.position, cameraPosition); put(cam.target, cameraDirection);
    okm(crt_mgpu_render(g_mgpu, &cam, &p, pixelcolorBuffer), "startRayTracing/crt_mgpu_render");
}
