/* oracle/crt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C ABI of the CPU restatement of the reference's render path (Ray.cu, bvh.cu, main.cu:33-136,
 * main.cu:545-573, objloader.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (cudaraytracer_amd/) never does.
 *
 * PARITY STATUS: the vec3 arithmetic and the OBJ loader are pinned bit-for-bit against the
 * reference's own sources compiled here (oracle/_ref, see oracle/Makefile).  The render logic
 * itself (Ray.cu / bvh.cu) cannot be compiled in this image without writing stand-ins for the
 * CUDA runtime and the <<<>>> launch syntax, the reference has no tests or golden vectors, and its
 * shipped BMPs come from an older scene, so for those functions this oracle is "parity unpinned":
 * a line-by-line restatement, each function citing the file:line it follows.
 */
#ifndef CRT_ORACLE_H
#define CRT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* POD mirror of SceneObject (SceneObjects.hpp:30-52).  Same field meaning as the product's
 * crt_prim so one numpy dtype serves both; defined independently here. */
typedef struct orc_prim {
    int32_t type;            /* 0 sphere (sphere=true), 1 triangle (triangle=true) */
    int32_t objNum;
    float position[3];
    float radius;
    float v1[3], v2[3], v3[3];
    float v1Norm[3], v2Norm[3], v3Norm[3];
    float ambient[3], diffuse[3], specular[3], reflective[3];
    float shininess;
    float direction[3];
    float speed;
} orc_prim;

/* POD mirror of Light (SceneObjects.hpp:16-28). */
typedef struct orc_light {
    float color[3];
    float direction[3];
    int32_t point;
    float constantTerm, linearTerm, quadraticTerm;
    float position[3];
    int32_t area;
    float radius;
} orc_light;

/* Mirror of Isect (isect.hpp:12-28) for known-answer tests of wallIntersection. */
typedef struct orc_isect {
    float isectPoint[3], incidentDirection[3], reflectionCoef[3];
    float ambient[3], diffuse[3], specular[3], reflective[3];
    float shininess;
    float normal[3], color[3];
} orc_isect;

typedef struct orc_stats {
    uint64_t traversals;     /* bvhTraverse calls from RayIntersection/Shade (Ray.cu:478,621,650) */
    uint64_t primary, shadow, reflect;
    uint64_t box_tests, tri_tests, sph_tests;
    uint64_t live_traversals; /* traversals on paths whose reflection coefficient is non-zero */
} orc_stats;

typedef struct orc_scene orc_scene;

/* ---- per-function known-answer hooks (Ray.cu:54-457) ---- */
int orc_bbox(const float box[6] /*minX,maxX,minY,maxY,minZ,maxZ*/, const float o[3], const float d[3]);
int orc_sphere(const orc_prim* s, const float o[3], const float d[3], float* t, float n[3], float x[3]);
int orc_triangle(const orc_prim* s, const float o[3], const float d[3], float* t, float n[3], float x[3]);
int orc_wall(const float o[3], const float d[3], int raytype, const float coef_in[3],
             const float color_in[3], orc_isect* out, float refl_coef_out[3]);

/* ---- scene + BVH (bvh.cu, main.cu:33-136) ---- */
orc_scene* orc_scene_create(const orc_prim* prims, int n);
void orc_scene_destroy(orc_scene*);
int orc_scene_size(const orc_scene*);
void orc_scene_get(const orc_scene*, orc_prim* out);
void orc_scene_set(orc_scene*, const orc_prim* prims);     /* same count; geometry moved by caller */
int orc_build(orc_scene*);                                 /* newBuild + constructTree; <0 on runaway recursion */
void orc_refit(orc_scene*);                                /* refitTree */
void orc_animate(orc_scene*);                              /* main.cu:545-573 */
int orc_num_nodes(const orc_scene*);
int orc_num_leaves(const orc_scene*);
/* DFS (left-then-right) dump: per node 6 box floats (minX,maxX,minY,maxY,minZ,maxZ), isleaf, parent index,
 * left, right (-1 for leaves), first/count into leaf_prims. Returns number of nodes. */
int orc_tree_dump(const orc_scene*, float* boxes6, int32_t* isleaf, int32_t* parent, int32_t* left,
                  int32_t* right, int32_t* first, int32_t* count, int32_t* leaf_prims);

/* ---- traversal (Ray.cu:184-236) ---- */
/* rays: n x 6 floats (origin, direction). prim_out = scene index or -1; t_out = minT (1e9 on miss).
 * normal/point outputs optional (may be NULL). */
void orc_trace(orc_scene*, const float* rays, int n, int32_t* prim_out, float* t_out,
               float* normal_out, float* point_out, int nthreads);
/* same primitives, no BVH: loop in scene order with the strict-< update */
void orc_trace_brute(orc_scene*, const float* rays, int n, int32_t* prim_out, float* t_out, int nthreads);

/* ---- frame (Ray.cu:31-52, 459-742) ---- */
/* bgr: w*h*3 (BGR, row 0 = bottom). Optional outputs (NULL to skip): color_f32 w*h*3 RGB before A10;
 * ids: bounces*w*h scene index per bounce (-1 none, -2-k wall k); ts: bounces*w*h minT.
 * rect = {x0,y0,x1,y1} renders only that pixel rectangle (others untouched) or NULL for all. */
int orc_render(orc_scene*, const orc_light* lights, int nlights, const float cam[3], const float target[3],
               int width, int height, int bounces, const int32_t* rect, unsigned char* bgr, float* color_f32,
               int32_t* ids, float* ts, int nthreads);
/* the same with area lights (orc_light.area != 0): area_samples shadow rays per light and shading point.
 * Ray.cu:563-603 is commented out in the reference, so there is no reference behaviour to pin this against; the
 * restatement and its departures are described at shade() in crt_oracle.cpp. */
int orc_render_area(orc_scene*, const orc_light* lights, int nlights, const float cam[3], const float target[3],
                    int width, int height, int bounces, const int32_t* rect, unsigned char* bgr, float* color_f32,
                    int32_t* ids, float* ts, int nthreads, int area_samples);
/* the same with spp_n x spp_n stratified sub-pixel samples per pixel (an extension: the reference's antialiasing
 * globals are stubs, Ray.cu:26,28,41-42); spp_n = 1 is orc_render_area */
int orc_render_spp(orc_scene*, const orc_light* lights, int nlights, const float cam[3], const float target[3],
                   int width, int height, int bounces, const int32_t* rect, unsigned char* bgr, float* color_f32,
                   int32_t* ids, float* ts, int nthreads, int area_samples, int spp_n);
void orc_area_table(float* out3, int n);                   /* n <= 1024 entries (U1, cos(theta), sin(theta)) */
void orc_primary_rays(const float cam[3], const float target[3], int width, int height, float* rays6);
void orc_stats_get(orc_scene*, orc_stats* out);
void orc_stats_reset(orc_scene*);
void orc_set_all_boxes_hit(orc_scene*, int on);      /* G3 probe: bvhTraverse without the false negatives of its box test */

/* ---- vec3 probe (pinned against vendored GLM by oracle/ref_driver.cpp) ---- */
#define ORC_PROBE_W 32
void orc_math_probe(const float* in9, int n, float* out32);

/* powf exactly as Shade's specular term calls it (Ray.cu:655), value by value: what the device's restatement is held against */
void orc_powf(const float* x, const float* y, int n, float* out);

/* ---- OBJ loader (objloader.cpp:20-112) ---- */
/* returns number of corners written (3 per face) or -1; pass NULL outputs to query the count. */
int orc_load_obj(const char* path, float* verts3, float* norms3, int max_corners);

#ifdef __cplusplus
}
#endif
#endif
