/* oracle/crt_oracle.cpp -- TEST INFRASTRUCTURE ONLY (never linked or loaded by the product).
 *
 * CPU restatement of boonemiller/CudaRayTracer's render path, written from the reference's
 * algorithm description line by line; every function cites the reference file:line it follows.
 * Arithmetic is IEEE binary32 with the reference's operation order (see ovec.h); where the
 * reference's C++ promotes a comparison to double the promotion is kept.  `pow(float,int)` and
 * `sqrt(float)` are evaluated in float, which is what the CUDA device build of the reference does
 * (the parity target of the kernels).  -DORC_HOST_POW builds the other reading (liboracle_hostpow.so):
 * std::pow(float,int) and sqrt of its sum in double, as a HOST build of Ray.cu:110,592-593,606-607,626-627
 * evaluates them; tests/test_oracle_hostpow.py measures how far the two are apart.
 *
 * PARITY STATUS: "parity unpinned" for the render logic (Ray.cu, bvh.cu cannot be compiled here
 * without stand-ins for CUDA and the reference ships no tests/golden vectors); vec3 math and the
 * OBJ loader are pinned against the reference's own compiled sources (oracle/_ref).
 *
 * Defined behaviour where the reference's is undefined (SURVEY.md 9.1 class G6):
 *   - longest-axis ties leave longestAxis/midpoint at 0 (zero-filled managed memory);
 *   - a ray that hits neither an object nor a wall terminates its path, colour unchanged
 *     (the reference reads an uninitialised shininess and produces NaN);
 *   - more than 64 primitives with one centroid make constructTree recurse forever; orc_build
 *     returns -1 past depth 256.
 * Image width is a parameter (the reference hard-codes 720 at Ray.cu:37-38; identical at 720).
 */
#include "crt_oracle.h"
#include "ovec.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits>
#include <vector>
#include <deque>
#include <thread>
#include <atomic>
#include <mutex>
#include <string>

namespace {

/* the three expressions of the reference that call pow(float, 2) (and sqrt of a sum of them) */
#ifdef ORC_HOST_POW
/* host build: pow promotes to double -- x^2 is exact there -- the sum and the square root are double, the assignment rounds once */
inline float sub_r2(float sum, float r) { return (float)((double)sum - (double)r * (double)r); }                                    /* Ray.cu:110 */
inline float len3(float a, float b, float c) { return (float)sqrt(((double)a * (double)a + (double)b * (double)b) + (double)c * (double)c); }   /* Ray.cu:592-593,606,626-627 */
inline float attenuation(float ct, float lt, float qt, float d) { return (float)(1.0f / ((double)(ct + lt * d) + (double)qt * ((double)d * (double)d))); }   /* Ray.cu:607 */
#else
/* CUDA device build: pow(float, int) and sqrt(float) stay in float */
inline float sub_r2(float sum, float r) { return sum - r * r; }
inline float len3(float a, float b, float c) { return sqrtf(a * a + b * b + c * c); }
inline float attenuation(float ct, float lt, float qt, float d) { return 1.0f / (ct + lt * d + qt * (d * d)); }
#endif

struct Counters {
    uint64_t traversals = 0, primary = 0, shadow = 0, reflect = 0;
    uint64_t box = 0, tri = 0, sph = 0, live = 0;
    void add(const Counters& o) {
        traversals += o.traversals; primary += o.primary; shadow += o.shadow; reflect += o.reflect;
        box += o.box; tri += o.tri; sph += o.sph; live += o.live;
    }
};

/* bvh.hpp:14-33 (object pointers become scene indices) */
struct Node {
    Node* parent; Node* left; Node* right;
    bool isleaf;
    int objs[64];
    int numObjs;
    float minX, maxX, minY, maxY, minZ, maxZ;
    int nodeNum;
    float midpoint;
    float longestAxis;
};

}  // namespace

struct orc_scene {
    std::vector<orc_prim> prims;
    Node* root = nullptr;
    std::deque<Node*> leafs;
    std::vector<Node*> all;     /* ownership; the reference's freeTree is buggy (bvh.cu:274-286) */
    int nNum = 1;               /* bvh.cu:11 */
    Counters stats;
    bool all_boxes_hit = false; /* G3 probe (orc_set_all_boxes_hit): bvhTraverse as if boundingBoxIntersection never said no */
};

namespace {

Node* new_node(orc_scene* s) {
    Node* n = (Node*)calloc(1, sizeof(Node));   /* managed memory arrives zero-filled */
    s->all.push_back(n);
    return n;
}
void free_nodes(orc_scene* s) {
    for (Node* n : s->all) free(n);
    s->all.clear(); s->leafs.clear(); s->root = nullptr; s->nNum = 1;
}

/* ---------------- Ray.cu:54-104 boundingBoxIntersection ---------------- */
bool bbox_hit(const ov3& position, const ov3& direction, float minX, float maxX, float minY, float maxY,
              float minZ, float maxZ) {
    float tmin = (minX - position.x) / direction.x;
    float tmax = (maxX - position.x) / direction.x;
    if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
    float tymin = (minY - position.y) / direction.y;
    float tymax = (maxY - position.y) / direction.y;
    if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (minZ - position.z) / direction.z;
    float tzmax = (maxZ - position.z) / direction.z;
    if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return true;
}

/* ---------------- Ray.cu:105-141 intersectSphere ---------------- */
bool sphere_hit(const orc_prim& s, const ov3& position, const ov3& direction, float& isectT, ov3& normal,
                ov3& intersection) {
    const float RAY_EPSILON = 0.000000001f;
    ov3 spos(s.position);
    float a = odot(direction, direction);
    float b = 2 * odot(direction, position - spos);
    float c = sub_r2(odot(spos, spos) + odot(position, position) + (-2 * odot(spos, position)), s.radius);
    float discriminant = b * b - 4 * a * c;
    if ((double)discriminant > 0.0 + (double)RAY_EPSILON) {
        float t = (-b - sqrtf(discriminant)) / (2 * a);
        float t2 = (-b + sqrtf(discriminant)) / (2 * a);
        if (t2 > RAY_EPSILON) {
            if (t > RAY_EPSILON) isectT = t; else isectT = t2;
            intersection = position + t * direction;          /* uses t even when t2 was returned */
            normal = onormalize((intersection - spos) / s.radius);
            return true;
        }
    }
    return false;
}

/* ---------------- Ray.cu:143-182 intersectTriangle ---------------- */
bool triangle_hit(const orc_prim& s, const ov3& position, const ov3& direction, ov3& n, ov3& intersection,
                  float& time) {
    ov3 sv1(s.v1), sv2(s.v2), sv3(s.v3);
    ov3 normal = onormalize(ocross((sv2 - sv1), (sv3 - sv1)));
    float denom = odot(normal, direction);
    if (fabsf(denom) > .0001f) {
        float t = odot((sv1 - position), normal) / denom;
        if ((double)t >= 0.0 - (double).0001f) {
            ov3 intersect = position + t * direction;
            float test1 = odot(ocross((sv2 - sv1), (intersect - sv1)), normal);
            float test2 = odot(ocross((sv3 - sv2), (intersect - sv2)), normal);
            float test3 = odot(ocross((sv1 - sv3), (intersect - sv3)), normal);
            if (test1 >= 0.0 && test2 >= 0.0 && test3 >= 0.0) {
                ov3 v0 = sv2 - sv1;
                ov3 v1 = sv3 - sv1;
                ov3 v2 = intersect - sv1;
                float d00 = odot(v0, v0);
                float d01 = odot(v0, v1);
                float d11 = odot(v1, v1);
                float d20 = odot(v2, v0);
                float d21 = odot(v2, v1);
                float den = d00 * d11 - d01 * d01;
                float v = (d11 * d20 - d01 * d21) / den;
                float w = (d00 * d21 - d01 * d20) / den;
                float u = 1.0f - v - w;
                n = onormalize(u * ov3(s.v1Norm) + v * ov3(s.v2Norm) + w * ov3(s.v3Norm));
                intersection = intersect;
                time = t;
                return true;
            }
        }
    }
    return false;
}

/* ---------------- Ray.cu:184-236 bvhTraverse ---------------- */
void bvh_traverse(const orc_scene* sc, const ov3& position, const ov3& direction, const Node* cur, bool& intersect,
                  float& minT, int& obj, ov3& minTnormal, ov3& minTintersection, Counters& k) {
    if (cur->isleaf) {
        k.box++;
        if (sc->all_boxes_hit || bbox_hit(position, direction, cur->minX, cur->maxX, cur->minY, cur->maxY, cur->minZ, cur->maxZ)) {
            for (int i = 0; i < cur->numObjs; i++) {
                const orc_prim& p = sc->prims[cur->objs[i]];
                if (p.type == 0) {
                    float iTime; ov3 normal, intersection;
                    k.sph++;
                    if (sphere_hit(p, position, direction, iTime, normal, intersection)) {
                        if (iTime < minT) {
                            minTnormal = normal; minTintersection = intersection; obj = cur->objs[i];
                            minT = iTime; intersect = true;
                        }
                    }
                } else if (p.type == 1) {
                    float iT; ov3 normal, intersection;
                    k.tri++;
                    if (triangle_hit(p, position, direction, normal, intersection, iT)) {
                        if (iT < minT) {
                            minTnormal = normal; minTintersection = intersection; obj = cur->objs[i];
                            minT = iT; intersect = true;
                        }
                    }
                }
            }
        }
    } else {
        const Node* l = cur->left; const Node* r = cur->right;
        k.box++;
        if (sc->all_boxes_hit || bbox_hit(position, direction, l->minX, l->maxX, l->minY, l->maxY, l->minZ, l->maxZ))
            bvh_traverse(sc, position, direction, l, intersect, minT, obj, minTnormal, minTintersection, k);
        k.box++;
        if (sc->all_boxes_hit || bbox_hit(position, direction, r->minX, r->maxX, r->minY, r->maxY, r->minZ, r->maxZ))
            bvh_traverse(sc, position, direction, r, intersect, minT, obj, minTnormal, minTintersection, k);
    }
}

struct RayS {          /* Ray.hpp:16-33, fields the path uses */
    ov3 position, direction, surfaceReflectiveCoef;
    int raytype;
};
struct IsectS {        /* isect.hpp:12-28 */
    ov3 isectPoint, incidentDirection, reflectionCoef, ambient, diffuse, specular, reflective;
    float shininess = 0.f;
    ov3 normal, color;
};

/* ---------------- Ray.cu:238-457 wallIntersection ---------------- */
struct WallDef { float nx, ny, nz; float px, py, pz; int kind; float dr, dg, db; float ar, ag, ab; };
/* order floor, left, front, back, right, ceiling; kind 0 bounds on (z,x), 1 on (z,y), 2 on (x,y) */
const WallDef kWalls[6] = {
    {0, 1, 0,   0, -200.0f, 0,  0,  1.0f, 0.2f, 0.2f,  1.0f, 0.2f, 0.2f},
    {1, 0, 0,  -200.0f, 0, 0,   1,  0.2f, 0.2f, 1.0f,  0.2f, 0.2f, 1.0f},
    {0, 0, 1,   0, 0, -300.0f,  2,  0.0f, 1.0f, 0.0f,  0.0f, 1.0f, 0.0f},
    {0, 0, -1,  0, 0, 300.0f,   2,  1.0f, 1.0f, 0.0f,  1.0f, 1.0f, 0.0f},
    {-1, 0, 0,  200.0f, 0, 0,   1,  0.2f, 0.2f, 1.0f,  0.2f, 0.2f, 1.0f},
    {0, -1, 0,  0, 200.0f, 0,   0,  0.9f, 0.9f, 0.9f,  1.0f, 1.0f, 1.0f},
};

int wall_hit(IsectS& ipoint, const RayS& r, RayS& reflect) {
    const float x = 200.0f, y = 200.0f, z = 300.0f;
    for (int w = 0; w < 6; w++) {
        const WallDef& W = kWalls[w];
        ov3 up(W.nx, W.ny, W.nz);
        float denom = odot(up, r.direction);
        if (fabsf(denom) > .0001f) {
            float t = odot((ov3(W.px, W.py, W.pz) - r.position), up) / denom;
            if ((double)t >= 0.0 - (double).0001f) {
                ov3 intersect = r.position + t * r.direction;
                bool inside;
                if (W.kind == 0) inside = intersect.z > -z && intersect.z < z && intersect.x > -x && intersect.x < x;
                else if (W.kind == 1) inside = intersect.z > -z && intersect.z < z && intersect.y < y && intersect.y > -y;
                else inside = intersect.x > -x && intersect.x < x && intersect.y < y && intersect.y > -y;
                if (inside) {
                    /* ambient colour used for the 0.2 ambient term is the wall colour (== ambient field) */
                    if (r.raytype == 0) {
                        ipoint.color = 0.2f * ov3(W.ar, W.ag, W.ab);
                        reflect.surfaceReflectiveCoef = ov3(0, 0, 0);
                        ipoint.reflectionCoef = ov3(1.0f, 1.0f, 1.0f);
                    } else if (r.raytype == 1) {
                        ipoint.reflectionCoef = r.surfaceReflectiveCoef;
                        reflect.surfaceReflectiveCoef = r.surfaceReflectiveCoef * ov3(0, 0, 0);
                    }
                    ipoint.normal = up;
                    ipoint.isectPoint = intersect;
                    ipoint.incidentDirection = onormalize(oreflect(r.direction, up));
                    ipoint.diffuse = ov3(W.dr, W.dg, W.db);
                    ipoint.ambient = ov3(W.ar, W.ag, W.ab);
                    ipoint.specular = ov3(0, 0, 0);
                    ipoint.shininess = 2;
                    ipoint.reflective = ov3(0, 0, 0);
                    return w;
                }
            }
        }
    }
    return -1;
}

/* ---------------- Ray.cu:459-544 RayIntersection, one ray ---------------- */
/* returns scene index >= 0, -2-k for wall k, -1 for a total miss */
int ray_intersection(const orc_scene* sc, const RayS& ray, RayS& refl, IsectS& is, float& tOut, Counters& k) {
    ov3 direction = ray.direction, position = ray.position;
    refl.raytype = 1;
    float minT = 1000000000;
    int obj = -1; ov3 minTnormal, minTintersection; bool intersect = false;
    k.traversals++;
    if (ray.raytype == 0) k.primary++; else k.reflect++;
    bool live = ray.raytype == 0 || ray.surfaceReflectiveCoef.x != 0.f || ray.surfaceReflectiveCoef.y != 0.f ||
                ray.surfaceReflectiveCoef.z != 0.f;
    if (live) k.live++;
    bvh_traverse(sc, position, direction, sc->root, intersect, minT, obj, minTnormal, minTintersection, k);
    tOut = minT;
    if (intersect) {
        const orc_prim& o = sc->prims[obj];
        if (ray.raytype == 0) {
            refl.surfaceReflectiveCoef = ov3(o.reflective);
            is.reflectionCoef = ov3(1.0f, 1.0f, 1.0f);
            is.color = 0.2f * ov3(o.ambient);
        } else if (ray.raytype == 1) {
            is.reflectionCoef = ray.surfaceReflectiveCoef;
            refl.surfaceReflectiveCoef = ray.surfaceReflectiveCoef * ov3(o.reflective);
        }
        minTintersection = minTintersection + minTnormal * 0.7f;
        refl.position = minTintersection;
        refl.direction = onormalize(oreflect(ray.direction, minTnormal));
        is.isectPoint = minTintersection;
        is.incidentDirection = ray.direction;
        is.normal = minTnormal;
        is.diffuse = ov3(o.diffuse);
        is.ambient = ov3(o.ambient);
        is.shininess = o.shininess;
        is.specular = ov3(o.specular);
        is.reflective = ov3(o.reflective);
        return obj;
    }
    IsectS point;
    point.color = is.color;
    int w = wall_hit(point, ray, refl);
    is = point;
    refl.position = point.isectPoint;
    refl.direction = onormalize(oreflect(ray.direction, point.normal));
    return w >= 0 ? -2 - w : -1;
}

/* ---------------- Ray.cu:546-663 Shade, one intersection ---------------- */
/* Area-light samples.  The reference draws radius and theta from curand (Ray.cu:569-570, commented out); this
 * restatement fixes the draws so that two implementations can agree: a table of 1024 (U1, cos, sin) triples
 * from the LCG x <- 1664525 x + 1013904223 (x0 = 0x2545F491, U = (x >> 8) / 2^24), and sample k of a shading
 * point reads entry (hash(pixel, light, bounce, 0) + k) mod 1024. */
#define ORC_PI 3.14159265359
static const int kAreaTableN = 1024;
static void area_table(float* out3, int n) {
    uint32_t x = 0x2545F491u;
    for (int i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        float u1 = (float)(x >> 8) * (1.0f / 16777216.0f);
        x = x * 1664525u + 1013904223u;
        float u2 = (float)(x >> 8) * (1.0f / 16777216.0f);
        float theta = u2 * (2 * ORC_PI);                       /* Ray.cu:570 */
        out3[3 * i] = u1; out3[3 * i + 1] = cosf(theta); out3[3 * i + 2] = sinf(theta);
    }
}
static const float* area_table_get() {
    static float tab[kAreaTableN * 3];
    static std::once_flag once;
    std::call_once(once, [] { area_table(tab, kAreaTableN); });
    return tab;
}
static uint32_t area_hash(uint32_t pixel, uint32_t light, uint32_t bounce, uint32_t sample) {
    uint32_t h = pixel * 0x9E3779B1u + light * 0x85EBCA77u + bounce * 0xC2B2AE3Du + sample * 0x27D4EB2Fu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}

void shade(const orc_scene* sc, IsectS& is, const orc_light* lights, int numlights, bool live, Counters& k,
           int lightSamples = 30, uint32_t pixel = 0, uint32_t bounce = 0, uint32_t sample = 0) {
    const float RAY_EPSILON = 0.000000001f;
    ov3 color;
    ov3 direction = onormalize(is.incidentDirection);
    ov3 intersection = is.isectPoint;
    ov3 normal = is.normal;
    float distance;
    ov3 toLight, reflectFromLight;
    for (int j = 0; j < numlights; j++) {
        const orc_light& l = lights[j];
        ov3 lpos(l.position), lcol(l.color), ldir(l.direction);
        if (l.area) {
            /* Ray.cu:563-603, commented out in the reference.  Restated with three stated departures: the draws
             * come from the table above; the sample offset is taken from the light's centre (the block adds it to
             * l.position cumulatively, a random walk); the light is skipped by the point/directional branches
             * below (the block falls through into them). */
            const float* tab = area_table_get();
            const uint32_t h = area_hash(pixel, (uint32_t)j, bounce, sample);
            ov3 avgcolor;
            for (int si = 0; si < lightSamples; si++) {
                const float* e = tab + 3 * ((h + (uint32_t)si) & (uint32_t)(kAreaTableN - 1));
                float radius = e[0] * l.radius;
                float x = radius * e[1];
                float z = radius * e[2];
                ov3 spos = lpos;
                spos.x = spos.x + x;
                spos.z = spos.z + z;
                toLight = onormalize(spos - intersection);
                reflectFromLight = -toLight;
                distance = 1.0f;
                float minT = 1000000000; int obj = -1; ov3 mn, mi; bool shadow = false;
                intersection = intersection + .01f * normal;
                k.traversals++; k.shadow++; if (live) k.live++;
                bvh_traverse(sc, intersection, toLight, sc->root, shadow, minT, obj, mn, mi, k);
                if (shadow) {
                    if (minT > RAY_EPSILON) {
                        ov3 ipoint = intersection + minT * toLight;
                        float ax = intersection.x - spos.x, ay = intersection.y - spos.y, az = intersection.z - spos.z;
                        float dtoLight = len3(ax, ay, az);
                        float bx = ipoint.x - intersection.x, by = ipoint.y - intersection.y, bz = ipoint.z - intersection.z;
                        float dtoLightIntersection = len3(bx, by, bz);
                        if (dtoLight > dtoLightIntersection) distance = distance * 0;
                    }
                }
                avgcolor += distance * lcol * (.6f * is.diffuse * omax(odot(toLight, normal), 0.0f) +
                                               .2f * is.specular * powf(odot(oreflect(reflectFromLight, normal), -direction), is.shininess));
            }
            color += avgcolor / (float)lightSamples;
        } else if (l.point) {
            float dx = intersection.x - lpos.x, dy = intersection.y - lpos.y, dz = intersection.z - lpos.z;
            float d = len3(dx, dy, dz);
            distance = attenuation(l.constantTerm, l.linearTerm, l.quadraticTerm, d);
            if (distance > 1.5) distance = .5;
            toLight = onormalize(lpos - intersection);
            reflectFromLight = -toLight;
            float minT = 1000000000; int obj = -1; ov3 mn, mi; bool shadow = false;
            intersection = intersection + .01f * normal;
            k.traversals++; k.shadow++; if (live) k.live++;
            bvh_traverse(sc, intersection, toLight, sc->root, shadow, minT, obj, mn, mi, k);
            if (shadow) {
                ov3 ipoint = intersection + minT * toLight;
                float ax = intersection.x - lpos.x, ay = intersection.y - lpos.y, az = intersection.z - lpos.z;
                float dtoLight = len3(ax, ay, az);
                float bx = ipoint.x - intersection.x, by = ipoint.y - intersection.y, bz = ipoint.z - intersection.z;
                float dtoLightIntersection = len3(bx, by, bz);
                if (dtoLight > dtoLightIntersection) distance = distance * 0;
            }
            color += distance * lcol * (.6f * is.diffuse * omax(odot(toLight, normal), 0.0f) +
                                        .2f * is.specular * powf(odot(oreflect(reflectFromLight, normal), -direction), is.shininess));
        } else {
            distance = 1.0f;
            toLight = -onormalize(ldir);
            reflectFromLight = onormalize(ldir);
            float minT = 1000000000; int obj = -1; ov3 mn, mi; bool shadow = false;
            float dotP = odot(reflectFromLight, normal);
            if ((double)dotP > -0.00001) shadow = true;
            else {
                k.traversals++; k.shadow++; if (live) k.live++;
                bvh_traverse(sc, intersection, toLight, sc->root, shadow, minT, obj, mn, mi, k);
            }
            if (shadow) {
                if (minT > RAY_EPSILON) distance = distance * 0;
            }
            color += distance * lcol * (.6f * is.diffuse * omax(odot(toLight, normal), 0.0f) +
                                        .2f * is.specular * powf(odot(oreflect(reflectFromLight, normal), -direction), is.shininess));
        }
    }
    is.color += is.reflectionCoef * color;
}

/* ---------------- Ray.cu:684-689 camera basis ---------------- */
struct Camera { ov3 L, u, v, pos; };
Camera make_camera(const ov3& cameraPosition, const ov3& cameraDirection, float width, float height) {
    Camera c;
    ov3 n = onormalize(cameraPosition - cameraDirection);
    c.u = onormalize(ocross(ov3(0, 1, 0), n));
    c.v = ocross(n, c.u);
    float fov = 45 / (180.0 / ORC_PI);
    float d = (height / tanf(fov / 2)) / 2;     /* C++ <math.h> resolves tan(float) to the float overload */
    c.L = (cameraPosition - n * d) - c.u * (width / 2) - c.v * (height / 2);
    c.pos = cameraPosition;
    return c;
}
/* Ray.cu:31-52 GeneratePrimaryRays, one pixel (literal 720 generalised to width) */
/* spp_n > 1: sub-sample `sample` of an spp_n x spp_n stratified grid, offsets ((k + .5) / spp_n - .5) along u and v
 * (SURVEY.md 8(d) config 5).  The reference has no supersampling (Ray.cu:26,28,41-42 are stubs): an extension, restated
 * here so that spp > 1 frames have an oracle; spp_n <= 1 is Ray.cu:31-52 untouched. */
void primary_ray(const Camera& c, int i, int width, RayS& r, int spp_n = 1, int smp = 0) {
    int ipix = i % width, jpix = i / width;
    ov3 pix;
    if (spp_n <= 1) pix = (c.L + c.u * float(ipix) + c.v * float(jpix));
    else {
        const float ox = ((float)(smp % spp_n) + .5f) / (float)spp_n - .5f;
        const float oy = ((float)(smp / spp_n) + .5f) / (float)spp_n - .5f;
        pix = (c.L + c.u * (float(ipix) + ox) + c.v * (float(jpix) + oy));
    }
    float xoffset = 0, yoffset = 0;
    ov3 sample = onormalize(ov3(pix.x + xoffset, pix.y + yoffset, pix.z) - c.pos);
    r.raytype = 0; r.position = c.pos; r.direction = sample; r.surfaceReflectiveCoef = ov3(0, 0, 0);
}

void box_init(Node* n) {
    n->maxX = std::numeric_limits<float>::min(); n->minX = std::numeric_limits<float>::max();
    n->maxY = std::numeric_limits<float>::min(); n->minY = std::numeric_limits<float>::max();
    n->maxZ = std::numeric_limits<float>::min(); n->minZ = std::numeric_limits<float>::max();
}
/* the grow-if-strictly-outside updates shared by main.cu:43-106, bvh.cu:70-131 and bvh.cu:298-361 */
void grow_prim(float& minX, float& maxX, float& minY, float& maxY, float& minZ, float& maxZ, const orc_prim& o,
               bool both_flags_refit) {
    (void)both_flags_refit;
    if (o.type == 0) {
        if (o.position[0] - o.radius < minX) minX = o.position[0] - o.radius;
        if (o.position[1] - o.radius < minY) minY = o.position[1] - o.radius;
        if (o.position[2] - o.radius < minZ) minZ = o.position[2] - o.radius;
        if (o.position[0] + o.radius > maxX) maxX = o.position[0] + o.radius;
        if (o.position[1] + o.radius > maxY) maxY = o.position[1] + o.radius;
        if (o.position[2] + o.radius > maxZ) maxZ = o.position[2] + o.radius;
    } else if (o.type == 1) {
        const float* vs[3] = {o.v1, o.v2, o.v3};
        for (int q = 0; q < 3; q++) {
            const float* v = vs[q];
            if (v[0] < minX) minX = v[0];
            if (v[1] < minY) minY = v[1];
            if (v[2] < minZ) minZ = v[2];
            if (v[0] > maxX) maxX = v[0];
            if (v[1] > maxY) maxY = v[1];
            if (v[2] > maxZ) maxZ = v[2];
        }
    }
}

/* ---------------- bvh.cu:12-273 constructTree ---------------- */
int construct_tree(orc_scene* sc, std::vector<int>& objects, Node* currentNode, Node* parentNode, int depth) {
    if (objects.size() <= 64) {
        for (size_t i = 0; i < objects.size(); i++) currentNode->objs[i] = objects[i];
        currentNode->numObjs = (int)objects.size();
        currentNode->isleaf = true;
        sc->leafs.push_back(currentNode);
        return 0;
    }
    if (depth > 256) return -1;
    Node* L = new_node(sc);
    L->parent = parentNode; L->nodeNum = sc->nNum++; L->left = NULL; L->right = NULL; L->isleaf = false;
    Node* R = new_node(sc);
    R->parent = parentNode; R->nodeNum = sc->nNum++; R->left = NULL; R->right = NULL; R->isleaf = false;

    std::vector<int> leftObjects, rightObjects;
    ov3 midLeft, midRight;
    box_init(L); box_init(R);
    int axis = (int)currentNode->longestAxis;
    for (size_t i = 0; i < objects.size(); i++) {
        const orc_prim& o = sc->prims[objects[i]];
        float objmidpoint;
        if (o.type == 0) objmidpoint = o.position[axis];
        else objmidpoint = (o.v1[axis] + o.v2[axis] + o.v3[axis]) / 3.0f;
        ov3 centroid = (o.type == 0) ? ov3(o.position)
                                     : ov3((o.v1[0] + o.v2[0] + o.v3[0]) / 3.0f, (o.v1[1] + o.v2[1] + o.v3[1]) / 3.0f,
                                           (o.v1[2] + o.v2[2] + o.v3[2]) / 3.0f);
        if (objmidpoint <= currentNode->midpoint) {
            grow_prim(L->minX, L->maxX, L->minY, L->maxY, L->minZ, L->maxZ, o, false);
            midLeft += centroid;
            leftObjects.push_back(objects[i]);
        } else {
            grow_prim(R->minX, R->maxX, R->minY, R->maxY, R->minZ, R->maxZ, o, false);
            midRight += centroid;
            rightObjects.push_back(objects[i]);
        }
    }
    /* float / size_t: the divisor converts to float (bvh.cu:202-203) */
    midLeft = ov3(midLeft.x / leftObjects.size(), midLeft.y / leftObjects.size(), midLeft.z / leftObjects.size());
    midRight = ov3(midRight.x / rightObjects.size(), midRight.y / rightObjects.size(), midRight.z / rightObjects.size());

    Node* ch[2] = {L, R}; ov3 mids[2] = {midLeft, midRight};
    for (int c = 0; c < 2; c++) {                   /* bvh.cu:205-253; ties assign nothing */
        Node* n = ch[c];
        if (n->maxX - n->minX > n->maxY - n->minY) {
            if (n->maxX - n->minX > n->maxZ - n->minZ) { n->longestAxis = 0; n->midpoint = mids[c].x; }
        }
        if (n->maxY - n->minY > n->maxX - n->minX) {
            if (n->maxY - n->minY > n->maxZ - n->minZ) { n->longestAxis = 1; n->midpoint = mids[c].y; }
        }
        if (n->maxZ - n->minZ > n->maxX - n->minX) {
            if (n->maxZ - n->minZ > n->maxY - n->minY) { n->longestAxis = 2; n->midpoint = mids[c].z; }
        }
    }
    currentNode->left = L;
    if (construct_tree(sc, leftObjects, L, L, depth + 1) < 0) return -1;
    currentNode->right = R;
    if (construct_tree(sc, rightObjects, R, R, depth + 1) < 0) return -1;
    return 0;
}

/* ---------------- main.cu:33-136 newBuild ---------------- */
int new_build(orc_scene* sc) {
    free_nodes(sc);
    Node* root = new_node(sc);
    sc->root = root;
    root->isleaf = false;
    box_init(root);
    for (const orc_prim& o : sc->prims)
        grow_prim(root->minX, root->maxX, root->minY, root->maxY, root->minZ, root->maxZ, o, false);
    if (root->maxX - root->minX > root->maxY - root->minY) {
        if (root->maxX - root->minX > root->maxZ - root->minZ) { root->longestAxis = 0; root->midpoint = (root->maxX + root->minX) / 2; }
    }
    if (root->maxY - root->minY > root->maxX - root->minX) {
        if (root->maxY - root->minY > root->maxZ - root->minZ) { root->longestAxis = 1; root->midpoint = (root->maxY + root->minY) / 2; }
    }
    if (root->maxZ - root->minZ > root->maxX - root->minX) {
        if (root->maxZ - root->minZ > root->maxY - root->minY) { root->longestAxis = 2; root->midpoint = (root->maxZ + root->minZ) / 2; }
    }
    root->parent = NULL;
    root->nodeNum = 0;
    std::vector<int> all(sc->prims.size());
    for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
    /* note constructTree(scene, cudaRoot, leafs, cudaRoot): children get parentNode == their own
     * node's parent argument, i.e. `parent` of a child is the node it hangs under. */
    return construct_tree(sc, all, root, root, 0);
}

/* ---------------- bvh.cu:288-400 refitTree ---------------- */
void refit_tree(orc_scene* sc) {
    for (size_t i = 0; i < sc->leafs.size(); i++) {
        Node* p = sc->leafs[i];
        while (p != NULL) {
            if (p->isleaf) {
                for (int o = 0; o < p->numObjs; o++)
                    grow_prim(p->minX, p->maxX, p->minY, p->maxY, p->minZ, p->maxZ, sc->prims[p->objs[o]], true);
            } else {
                Node* l = p->left; Node* r = p->right;
                if (l->minX < p->minX) p->minX = l->minX;
                if (r->minX < p->minX) p->minX = r->minX;
                if (l->maxX > p->maxX) p->maxX = l->maxX;
                if (r->maxX > p->maxX) p->maxX = r->maxX;
                if (l->minY < p->minY) p->minY = l->minY;
                if (r->minY < p->minY) p->minY = r->minY;
                if (l->maxY > p->maxY) p->maxY = l->maxY;
                if (r->maxY > p->maxY) p->maxY = r->maxY;
                if (l->minZ < p->minZ) p->minZ = l->minZ;
                if (r->minZ < p->minZ) p->minZ = r->minZ;
                if (l->maxZ > p->maxZ) p->maxZ = l->maxZ;
                if (r->maxZ > p->maxZ) p->maxZ = r->maxZ;
            }
            p = p->parent;
        }
    }
}

/* glm::rotate(mat4(1), 0.01f, (0,1,0)) then mat4 * vec4(v,1): gtc/matrix_transform.inl:52-87,
 * detail/type_mat4x4.inl:640-651 ((m0*v0 + m1*v1) + (m2*v2 + m3*v3)). */
struct Mat4 { float c[4][4]; };
Mat4 rotate_identity(float angle, const ov3& v) {
    float a = angle, c = cosf(a), s = sinf(a);
    ov3 axis = onormalize(v);
    ov3 temp = (1.0f - c) * axis;
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = 0 + temp.x * axis.y + s * axis.z;
    R[0][2] = 0 + temp.x * axis.z - s * axis.y;
    R[1][0] = 0 + temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = 0 + temp.y * axis.z + s * axis.x;
    R[2][0] = 0 + temp.z * axis.x + s * axis.y;
    R[2][1] = 0 + temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    float m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    Mat4 out;
    for (int col = 0; col < 3; col++)
        for (int k = 0; k < 4; k++)
            out.c[col][k] = m[0][k] * R[col][0] + m[1][k] * R[col][1] + m[2][k] * R[col][2];
    for (int k = 0; k < 4; k++) out.c[3][k] = m[3][k];
    return out;
}
ov3 xform_point(const Mat4& m, const float* v) {
    float r[3];
    for (int k = 0; k < 3; k++) {
        float mul0 = m.c[0][k] * v[0], mul1 = m.c[1][k] * v[1];
        float add0 = mul0 + mul1;
        float mul2 = m.c[2][k] * v[2], mul3 = m.c[3][k] * 1.0f;
        float add1 = mul2 + mul3;
        r[k] = add0 + add1;
    }
    return ov3(r[0], r[1], r[2]);
}

/* ---------------- main.cu:545-573 animation step ---------------- */
void animate(orc_scene* sc) {
    for (orc_prim& s : sc->prims) {
        if (s.type == 1) {
            Mat4 trans = rotate_identity(0.01f, ov3(0.0f, 1.0f, 0.0f));
            xform_point(trans, s.v1).store(s.v1);
            xform_point(trans, s.v2).store(s.v2);
            xform_point(trans, s.v3).store(s.v3);
            xform_point(trans, s.v1Norm).store(s.v1Norm);
            xform_point(trans, s.v2Norm).store(s.v2Norm);
            xform_point(trans, s.v3Norm).store(s.v3Norm);
        }
        if (s.type == 0) {
            ov3 dir(s.direction);
            if (s.position[0] + s.radius > 200.0f) dir = oreflect(dir, ov3(-1, 0, 0));
            if (s.position[0] - s.radius < -200.0f) dir = oreflect(dir, ov3(1, 0, 0));
            if (s.position[1] + s.radius > 200.0f) dir = oreflect(dir, ov3(0, -1, 0));
            if (s.position[1] - s.radius < -200.0f) dir = oreflect(dir, ov3(0, 1, 0));
            if (s.position[2] + s.radius > 300.0f) dir = oreflect(dir, ov3(0, 0, -1));
            if (s.position[2] - s.radius < -300.0f) dir = oreflect(dir, ov3(0, 0, 1));
            dir.store(s.direction);
            ov3 p = ov3(s.position) + dir * s.speed * 8.0f;
            p.store(s.position);
        }
    }
}

template <class F>
void parallel_for(int n, int nthreads, F f) {
    if (nthreads <= 1 || n < 2) { f(0, n, 0); return; }
    std::vector<std::thread> th;
    std::atomic<int> next(0);
    const int chunk = 256;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t]() {
            for (;;) {
                int b = next.fetch_add(chunk);
                if (b >= n) break;
                int e = b + chunk < n ? b + chunk : n;
                f(b, e, t);
            }
        });
    for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

int orc_bbox(const float box[6], const float o[3], const float d[3]) {
    return bbox_hit(ov3(o), ov3(d), box[0], box[1], box[2], box[3], box[4], box[5]) ? 1 : 0;
}
int orc_sphere(const orc_prim* s, const float o[3], const float d[3], float* t, float n[3], float x[3]) {
    float tt = 0; ov3 nn, xx;
    bool h = sphere_hit(*s, ov3(o), ov3(d), tt, nn, xx);
    if (h) { *t = tt; nn.store(n); xx.store(x); }
    return h;
}
int orc_triangle(const orc_prim* s, const float o[3], const float d[3], float* t, float n[3], float x[3]) {
    float tt = 0; ov3 nn, xx;
    bool h = triangle_hit(*s, ov3(o), ov3(d), nn, xx, tt);
    if (h) { *t = tt; nn.store(n); xx.store(x); }
    return h;
}
int orc_wall(const float o[3], const float d[3], int raytype, const float coef_in[3], const float color_in[3],
             orc_isect* out, float refl_coef_out[3]) {
    RayS r; r.position = ov3(o); r.direction = ov3(d); r.raytype = raytype; r.surfaceReflectiveCoef = ov3(coef_in);
    RayS refl; refl.surfaceReflectiveCoef = ov3(coef_in);
    IsectS p; p.color = ov3(color_in);
    int w = wall_hit(p, r, refl);
    p.isectPoint.store(out->isectPoint); p.incidentDirection.store(out->incidentDirection);
    p.reflectionCoef.store(out->reflectionCoef); p.ambient.store(out->ambient); p.diffuse.store(out->diffuse);
    p.specular.store(out->specular); p.reflective.store(out->reflective); out->shininess = p.shininess;
    p.normal.store(out->normal); p.color.store(out->color);
    refl.surfaceReflectiveCoef.store(refl_coef_out);
    return w;
}

void orc_powf(const float* x, const float* y, int n, float* out) {
    for (int i = 0; i < n; i++) out[i] = powf(x[i], y[i]);
}

/* Probe of the vec3 restatement, compared bit-for-bit with the vendored GLM by oracle/ref_driver.cpp.
 * in: n rows of 9 floats (a, b, c); out: n rows of ORC_PROBE_W floats. */
void orc_math_probe(const float* in, int n, float* out) {
    for (int i = 0; i < n; i++) {
        ov3 a(in + 9 * i), b(in + 9 * i + 3), c(in + 9 * i + 6);
        float* o = out + 32 * i;
        o[0] = odot(a, b);
        ocross(a, b).store(o + 1);
        onormalize(a).store(o + 4);
        oreflect(a, b).store(o + 7);
        (a / b.x).store(o + 10);
        (a * b.x).store(o + 13);
        (b.x * a).store(o + 16);
        ov3 s = a * 255.0f;
        ov3(oclamp(s.x, 0.0f, 255.0f), oclamp(s.y, 0.0f, 255.0f), oclamp(s.z, 0.0f, 255.0f)).store(o + 19);
        o[22] = omax(a.x, b.x);
        o[23] = omin(a.x, b.x);
        Mat4 m = rotate_identity(0.01f, ov3(0.0f, 1.0f, 0.0f));
        float av[3] = {a.x, a.y, a.z};
        xform_point(m, av).store(o + 24);
        (a * b * c).store(o + 27);
        o[30] = powf(fabsf(a.x), c.x);
        o[31] = (a + b * c.x - c).x;
    }
}

orc_scene* orc_scene_create(const orc_prim* prims, int n) {
    orc_scene* s = new orc_scene();
    s->prims.assign(prims, prims + n);
    return s;
}
void orc_scene_destroy(orc_scene* s) { if (!s) return; free_nodes(s); delete s; }
int orc_scene_size(const orc_scene* s) { return (int)s->prims.size(); }
void orc_scene_get(const orc_scene* s, orc_prim* out) { memcpy(out, s->prims.data(), s->prims.size() * sizeof(orc_prim)); }
void orc_scene_set(orc_scene* s, const orc_prim* prims) { memcpy(s->prims.data(), prims, s->prims.size() * sizeof(orc_prim)); }
int orc_build(orc_scene* s) { return new_build(s); }
void orc_refit(orc_scene* s) { refit_tree(s); }
void orc_animate(orc_scene* s) { animate(s); }
int orc_num_nodes(const orc_scene* s) { return (int)s->all.size(); }
int orc_num_leaves(const orc_scene* s) { return (int)s->leafs.size(); }

static int dump_rec(const Node* n, int parent, int& idx, int& lp, float* boxes6, int32_t* isleaf, int32_t* par,
                    int32_t* left, int32_t* right, int32_t* first, int32_t* count, int32_t* leaf_prims) {
    int me = idx++;
    float* b = boxes6 + 6 * me;
    b[0] = n->minX; b[1] = n->maxX; b[2] = n->minY; b[3] = n->maxY; b[4] = n->minZ; b[5] = n->maxZ;
    isleaf[me] = n->isleaf; par[me] = parent; left[me] = -1; right[me] = -1; first[me] = lp; count[me] = 0;
    if (n->isleaf) {
        count[me] = n->numObjs;
        for (int i = 0; i < n->numObjs; i++) leaf_prims[lp++] = n->objs[i];
    } else {
        left[me] = dump_rec(n->left, me, idx, lp, boxes6, isleaf, par, left, right, first, count, leaf_prims);
        right[me] = dump_rec(n->right, me, idx, lp, boxes6, isleaf, par, left, right, first, count, leaf_prims);
    }
    return me;
}
int orc_tree_dump(const orc_scene* s, float* boxes6, int32_t* isleaf, int32_t* parent, int32_t* left, int32_t* right,
                  int32_t* first, int32_t* count, int32_t* leaf_prims) {
    if (!s->root) return 0;
    int idx = 0, lp = 0;
    dump_rec(s->root, -1, idx, lp, boxes6, isleaf, parent, left, right, first, count, leaf_prims);
    return idx;
}

void orc_trace(orc_scene* s, const float* rays, int n, int32_t* prim_out, float* t_out, float* normal_out,
               float* point_out, int nthreads) {
    std::vector<Counters> ks(nthreads > 1 ? nthreads : 1);
    parallel_for(n, nthreads, [&](int b, int e, int t) {
        for (int i = b; i < e; i++) {
            ov3 o(rays + 6 * i), d(rays + 6 * i + 3);
            float minT = 1000000000; int obj = -1; ov3 mn, mi; bool hit = false;
            ks[t].traversals++;
            bvh_traverse(s, o, d, s->root, hit, minT, obj, mn, mi, ks[t]);
            prim_out[i] = hit ? obj : -1;
            t_out[i] = minT;
            if (normal_out) mn.store(normal_out + 3 * i);
            if (point_out) mi.store(point_out + 3 * i);
        }
    });
    for (auto& k : ks) s->stats.add(k);
}

void orc_trace_brute(orc_scene* s, const float* rays, int n, int32_t* prim_out, float* t_out, int nthreads) {
    parallel_for(n, nthreads, [&](int b, int e, int) {
        for (int i = b; i < e; i++) {
            ov3 o(rays + 6 * i), d(rays + 6 * i + 3);
            float minT = 1000000000; int obj = -1;
            for (size_t p = 0; p < s->prims.size(); p++) {
                float t; ov3 nn, xx; bool h = false;
                if (s->prims[p].type == 0) h = sphere_hit(s->prims[p], o, d, t, nn, xx);
                else if (s->prims[p].type == 1) h = triangle_hit(s->prims[p], o, d, nn, xx, t);
                if (h && t < minT) { minT = t; obj = (int)p; }
            }
            prim_out[i] = obj; t_out[i] = minT;
        }
    });
}

void orc_primary_rays(const float cam[3], const float target[3], int width, int height, float* rays6) {
    Camera c = make_camera(ov3(cam), ov3(target), (float)width, (float)height);
    for (int i = 0; i < width * height; i++) {
        RayS r; primary_ray(c, i, width, r);
        r.position.store(rays6 + 6 * i); r.direction.store(rays6 + 6 * i + 3);
    }
}

/* Ray.cu:665-742 startRayTracing, pixel-major (pixels are independent in the reference) */
int orc_render(orc_scene* s, const orc_light* lights, int nlights, const float cam[3], const float target[3],
               int width, int height, int bounces, const int32_t* rect, unsigned char* bgr, float* color_f32,
               int32_t* ids, float* ts, int nthreads) {
    return orc_render_area(s, lights, nlights, cam, target, width, height, bounces, rect, bgr, color_f32, ids, ts, nthreads, 30);
}

void orc_area_table(float* out3, int n) { area_table(out3, n); }

int orc_render_area(orc_scene* s, const orc_light* lights, int nlights, const float cam[3], const float target[3],
                    int width, int height, int bounces, const int32_t* rect, unsigned char* bgr, float* color_f32,
                    int32_t* ids, float* ts, int nthreads, int area_samples) {
    return orc_render_spp(s, lights, nlights, cam, target, width, height, bounces, rect, bgr, color_f32, ids, ts, nthreads, area_samples, 1);
}

/* spp_n x spp_n stratified samples per pixel: every sample is one full path as below; the float colours are summed in
 * sample order and divided by spp_n^2 before Ray.cu:724-731.  ids / ts report the LAST sample's hits. */
int orc_render_spp(orc_scene* s, const orc_light* lights, int nlights, const float cam[3], const float target[3],
                   int width, int height, int bounces, const int32_t* rect, unsigned char* bgr, float* color_f32,
                   int32_t* ids, float* ts, int nthreads, int area_samples, int spp_n) {
    if (!s->root) return -1;
    if (spp_n < 1) spp_n = 1;
    Camera c = make_camera(ov3(cam), ov3(target), (float)width, (float)height);
    int x0 = 0, y0 = 0, x1 = width, y1 = height;
    if (rect) { x0 = rect[0]; y0 = rect[1]; x1 = rect[2]; y1 = rect[3]; }
    int rw = x1 - x0, rh = y1 - y0;
    int total = width * height;
    std::vector<Counters> ks(nthreads > 1 ? nthreads : 1);
    parallel_for(rw * rh, nthreads, [&](int b, int e, int t) {
        Counters& k = ks[t];
        for (int q = b; q < e; q++) {
            int i = (y0 + q / rw) * width + (x0 + q % rw);
            ov3 sum;
            ov3 col;
            for (int smp = 0; smp < spp_n * spp_n; smp++) {
            RayS ray, refl; IsectS is;
            primary_ray(c, i, width, ray, spp_n, smp);
            bool dead = false;
            for (int bn = 0; bn < bounces; bn++) {
                if (dead) {
                    if (ids) ids[(size_t)bn * total + i] = -1;
                    if (ts) ts[(size_t)bn * total + i] = 1000000000;
                    continue;
                }
                float tHit;
                bool live = ray.raytype == 0 || ray.surfaceReflectiveCoef.x != 0.f ||
                            ray.surfaceReflectiveCoef.y != 0.f || ray.surfaceReflectiveCoef.z != 0.f;
                ov3 keep = is.color;
                int id = ray_intersection(s, ray, refl, is, tHit, k);
                if (ids) ids[(size_t)bn * total + i] = id;
                if (ts) ts[(size_t)bn * total + i] = tHit;
                if (id == -1) { is.color = keep; dead = true; continue; }   /* G6: defined as path end */
                shade(s, is, lights, nlights, live, k, area_samples, (uint32_t)i, (uint32_t)bn, (uint32_t)smp);
                ray = refl;
            }
            col = is.color;
            if (spp_n > 1) { if (smp == 0) sum = ov3(0, 0, 0); sum = ov3(sum.x + col.x, sum.y + col.y, sum.z + col.z); }
            }
            if (spp_n > 1) { const float inv = (float)(spp_n * spp_n); col = ov3(sum.x / inv, sum.y / inv, sum.z / inv); }
            if (color_f32) col.store(color_f32 + 3 * (size_t)i);
            /* Ray.cu:724-731 */
            col = col * 255.0f;
            col = ov3(oclamp(col.x, 0.0f, 255.0f), oclamp(col.y, 0.0f, 255.0f), oclamp(col.z, 0.0f, 255.0f));
            bgr[(size_t)i * 3] = (unsigned char)col.z;
            bgr[(size_t)i * 3 + 1] = (unsigned char)col.y;
            bgr[(size_t)i * 3 + 2] = (unsigned char)col.x;
        }
    });
    for (auto& k : ks) s->stats.add(k);
    return 0;
}

void orc_stats_get(orc_scene* s, orc_stats* o) {
    o->traversals = s->stats.traversals; o->primary = s->stats.primary; o->shadow = s->stats.shadow;
    o->reflect = s->stats.reflect; o->box_tests = s->stats.box; o->tri_tests = s->stats.tri;
    o->sph_tests = s->stats.sph; o->live_traversals = s->stats.live;
}
void orc_stats_reset(orc_scene* s) { s->stats = Counters(); }
/* G3 probe: the reference's traversal with every box test answered "hit" -- its own leaves in its own order, so
 * exact-t ties resolve as in the reference, but none of the false negatives of its slab test (Ray.cu:54-104: 0 * inf for a ray
 * inside the plane of a box of zero thickness).  What a pixel or ray "should" be when the reference disagrees with itself. */
void orc_set_all_boxes_hit(orc_scene* s, int on) { s->all_boxes_hit = on != 0; }

/* objloader.cpp:20-112: "v", "vn", "f a b c" only; the vertex index doubles as normal index;
 * other tokens eat the line.  Missing file returns -1 (the reference blocks on getchar()). */
int orc_load_obj(const char* path, float* verts3, float* norms3, int max_corners) {
    FILE* file = fopen(path, "r");
    if (!file) return -1;
    std::vector<unsigned int> vi;
    std::vector<float> tv, tn;
    for (;;) {
        char lineHeader[128];
        int res = fscanf(file, "%127s", lineHeader);
        if (res == EOF) break;
        if (strcmp(lineHeader, "v") == 0) {
            float a = 0, b = 0, c = 0;
            if (fscanf(file, "%f %f %f\n", &a, &b, &c)) {}
            tv.push_back(a); tv.push_back(b); tv.push_back(c);
        } else if (strcmp(lineHeader, "vt") == 0) {
            float a, b;
            if (fscanf(file, "%f %f\n", &a, &b)) {}
        } else if (strcmp(lineHeader, "vn") == 0) {
            float a = 0, b = 0, c = 0;
            if (fscanf(file, "%f %f %f\n", &a, &b, &c)) {}
            tn.push_back(a); tn.push_back(b); tn.push_back(c);
        } else if (strcmp(lineHeader, "f") == 0) {
            unsigned int i0, i1, i2;
            int m = fscanf(file, "%d %d %d\n", (int*)&i0, (int*)&i1, (int*)&i2);
            if (m == 3) { vi.push_back(i0); vi.push_back(i1); vi.push_back(i2); }
            else { fclose(file); return -1; }
        } else {
            char buf[1000];
            if (!fgets(buf, 1000, file)) {}
        }
    }
    fclose(file);
    int n = (int)vi.size();
    if (!verts3 || !norms3) return n;
    if (n > max_corners) return -1;
    for (int i = 0; i < n; i++) {
        unsigned int k = vi[i];
        if (k < 1 || 3 * (size_t)k > tv.size() || 3 * (size_t)k > tn.size()) return -1;
        for (int q = 0; q < 3; q++) { verts3[3 * i + q] = tv[3 * (k - 1) + q]; norms3[3 * i + q] = tn[3 * (k - 1) + q]; }
    }
    return n;
}

}  // extern "C"
