/* scene.cpp -- host-side scene layer behind the C ABI: OBJ loading, the reference's scene
 * literals restated as scene builders, the per-frame animation step, and headless BMP output.
 *
 *   crt_load_obj        objloader.cpp:20-112
 *   crt_scene_teapot    main.cu:230-260 (teapot triangles) + main.cu:304-321 (sphere s2)
 *   crt_lights_default  main.cu:464-488 (l1 and l3; l2 is never pushed)
 *   crt_camera_default  main.cu:459,508
 *   crt_animate_host    main.cu:545-573
 *   crt_scene_spheres / crt_scene_soup   synthetic scenes of SURVEY.md 8(d) configs 3 and 4
 *   crt_write_bmp       replaces the OpenGL display (main.cu:574-633)
 */
#include "../../include/crt.h"
#include "vec.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <random>

using crt::vec3;

extern "C" {

/* "v", "vn" and "f a b c" only; the vertex index doubles as the normal index
 * (objloader.cpp:71-76); every other token eats its line.  A missing file returns CRT_ERR_IO
 * (the reference blocks on getchar(), objloader.cpp:37). */
int crt_load_obj(const char* path, float* verts3, float* norms3, int max_corners) {
    FILE* f = fopen(path, "r");
    if (!f) return CRT_ERR_IO;
    std::vector<float> pos, nrm;
    std::vector<int> idx;
    char tok[128];
    while (fscanf(f, "%127s", tok) != EOF) {
        if (!strcmp(tok, "v") || !strcmp(tok, "vn")) {
            float a = 0, b = 0, c = 0;
            if (fscanf(f, "%f %f %f\n", &a, &b, &c) < 0) break;
            std::vector<float>& dst = tok[1] ? nrm : pos;
            dst.push_back(a); dst.push_back(b); dst.push_back(c);
        } else if (!strcmp(tok, "vt")) {
            float a, b;
            if (fscanf(f, "%f %f\n", &a, &b) < 0) break;
        } else if (!strcmp(tok, "f")) {
            int a, b, c;
            if (fscanf(f, "%d %d %d\n", &a, &b, &c) != 3) { fclose(f); return CRT_ERR_IO; }
            idx.push_back(a); idx.push_back(b); idx.push_back(c);
        } else {
            char line[1000];
            if (!fgets(line, sizeof line, f)) break;
        }
    }
    fclose(f);
    const int n = (int)idx.size();
    if (!verts3 || !norms3) return n;
    if (n > max_corners) return CRT_ERR_ARG;
    for (int i = 0; i < n; i++) {
        const long k = idx[i];
        if (k < 1 || (size_t)(3 * k) > pos.size() || (size_t)(3 * k) > nrm.size()) return CRT_ERR_IO;
        memcpy(verts3 + 3 * i, &pos[3 * (k - 1)], 3 * sizeof(float));
        memcpy(norms3 + 3 * i, &nrm[3 * (k - 1)], 3 * sizeof(float));
    }
    return n;
}

static void set3(float* d, float a, float b, float c) { d[0] = a; d[1] = b; d[2] = c; }

static void teapot_material(crt_prim& s) {
    set3(s.ambient, 0.f, 0.f, 1.f);
    set3(s.specular, (float)0.9, (float)0.4, 0.f);
    set3(s.diffuse, (float)0.8, (float)0.3, (float)0.1);
    set3(s.reflective, (float).5, (float).5, (float).5);
    s.shininess = 64;
}
static void sphere_material(crt_prim& s) {       /* s2, main.cu:313-319 */
    set3(s.ambient, 1.f, 1.f, 1.f);
    set3(s.specular, .5f, .5f, .5f);
    set3(s.diffuse, 1.f, 1.f, 1.f);
    set3(s.reflective, .5f, .5f, .5f);
    s.shininess = 64;
}

int crt_scene_teapot(const char* obj_path, int with_sphere, crt_prim* out, int cap) {
    int nc = crt_load_obj(obj_path, nullptr, nullptr, 0);
    if (nc < 0) return nc;
    std::vector<float> v(3 * (size_t)nc), n(3 * (size_t)nc);
    int rc = crt_load_obj(obj_path, v.data(), n.data(), nc);
    if (rc < 0) return rc;
    const int ntri = nc / 3;
    const int total = ntri + (with_sphere ? 1 : 0);
    if (!out) return total;
    if (total > cap) return CRT_ERR_ARG;
    for (int t = 0; t < ntri; t++) {
        crt_prim& s = out[t];
        memset(&s, 0, sizeof s);
        s.type = 1; s.objNum = 3 * t; s.radius = 0;
        teapot_material(s);
        for (int c = 0; c < 3; c++) {
            float* pv = c == 0 ? s.v1 : (c == 1 ? s.v2 : s.v3);
            float* pn = c == 0 ? s.v1Norm : (c == 1 ? s.v2Norm : s.v3Norm);
            memcpy(pv, &v[3 * (3 * t + c)], 3 * sizeof(float));
            crt::normalize(vec3(&n[3 * (3 * t + c)])).store(pn);     /* main.cu:249-253 */
        }
    }
    if (with_sphere) {
        crt_prim& s = out[ntri];
        memset(&s, 0, sizeof s);
        /* first four values of glibc rand() with the default seed, as an unseeded process sees them
         * (main.cu:306-310) */
        const int r[4] = {1804289383, 846930886, 1681692777, 1714636915};
        const float rmax = (float)2147483647;
        s.direction[0] = -1 + ((float)r[0] / rmax * 2);
        s.direction[1] = -1 + ((float)r[1] / rmax * 2);
        s.direction[2] = -1 + ((float)r[2] / rmax * 2);
        s.speed = (float)(.01 + ((float)r[3] / rmax * .3));
        s.type = 0; s.objNum = 1; s.radius = 20;
        sphere_material(s);
        set3(s.position, -40, 40, 0);
    }
    return total;
}

int crt_lights_default(crt_light* out) {
    memset(out, 0, 2 * sizeof(crt_light));
    set3(out[0].direction, 2, -1, -1); set3(out[0].color, 1, 1, 1);
    set3(out[1].direction, -2, -1, 1); set3(out[1].color, 1, 1, 1);
    return 2;
}

/* Unit-disk samples of the area lights (Ray.cu:569-573: radius = U * l.radius, theta = U * 2*PI, offset
 * (radius*cos(theta), 0, radius*sin(theta))).  The reference draws U from curand per thread; here the draws are
 * a fixed table from the 32-bit LCG x <- 1664525 x + 1013904223, x0 = 0x2545F491, U = (x >> 8) / 2^24. */
int crt_area_table(float* out3, int n) {
    if (!out3 || n < 0 || n > 1024) return CRT_ERR_ARG;
    uint32_t x = 0x2545F491u;
    for (int i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        const float u1 = (float)(x >> 8) * (1.0f / 16777216.0f);
        x = x * 1664525u + 1013904223u;
        const float u2 = (float)(x >> 8) * (1.0f / 16777216.0f);
        const float theta = u2 * (2 * 3.14159265359);     /* PI of Ray.cu:16; the product is a double, stored as float */
        out3[3 * i] = u1; out3[3 * i + 1] = cosf(theta); out3[3 * i + 2] = sinf(theta);
    }
    return n;
}

void crt_camera_default(crt_camera* c) { set3(c->position, 0, 50, 250); set3(c->target, 0, 0, 0); }

static inline float u01(std::mt19937& g) { return (float)(g() >> 8) * (1.0f / 16777216.0f); }

/* n spheres, centres uniform in the room shrunk by the radius, radius
 * uniform [2,8], direction uniform [-1,1]^3 (not normalised) and speed .01 + U*.3 as main.cu:306-310,
 * material of s2. */
int crt_scene_spheres(int n, uint32_t seed, crt_prim* out) {
    if (n < 0 || !out) return CRT_ERR_ARG;
    std::mt19937 g(seed);
    for (int i = 0; i < n; i++) {
        crt_prim& s = out[i];
        memset(&s, 0, sizeof s);
        s.type = 0; s.objNum = i;
        s.radius = 2.f + 6.f * u01(g);
        const float hx = 200.f - s.radius, hy = 200.f - s.radius, hz = 300.f - s.radius;
        set3(s.position, -hx + 2.f * hx * u01(g), -hy + 2.f * hy * u01(g), -hz + 2.f * hz * u01(g));
        set3(s.direction, -1.f + 2.f * u01(g), -1.f + 2.f * u01(g), -1.f + 2.f * u01(g));
        s.speed = (float)(.01 + u01(g) * .3);
        sphere_material(s);
    }
    return n;
}

/* triangle soup.  Centre uniform in [-180,180]x[-180,180]x[-280,200],
 * vertices = centre + uniform [-1.5,1.5]^3, vertex normals = geometric normal, teapot material. */
int crt_scene_soup(int ntris, uint32_t seed, crt_prim* out) {
    if (ntris < 0 || !out) return CRT_ERR_ARG;
    std::mt19937 g(seed);
    for (int i = 0; i < ntris; i++) {
        crt_prim& s = out[i];
        memset(&s, 0, sizeof s);
        s.type = 1; s.objNum = i;
        teapot_material(s);
        const float cx = -180.f + 360.f * u01(g), cy = -180.f + 360.f * u01(g), cz = -280.f + 480.f * u01(g);
        float* vs[3] = {s.v1, s.v2, s.v3};
        for (float* v : vs) set3(v, cx + (-1.5f + 3.f * u01(g)), cy + (-1.5f + 3.f * u01(g)), cz + (-1.5f + 3.f * u01(g)));
        vec3 ng = crt::normalize(crt::cross(vec3(s.v2) - vec3(s.v1), vec3(s.v3) - vec3(s.v1)));
        ng.store(s.v1Norm); ng.store(s.v2Norm); ng.store(s.v3Norm);
    }
    return ntris;
}

/* glm::rotate(mat4(1), 0.01f, (0,1,0)) followed by mat4 * vec4(v, 1):
 * gtc/matrix_transform.inl:52-87 and detail/type_mat4x4.inl:640-651 -- (m0*x + m1*y) + (m2*z + m3*1). */
namespace {
struct Rot { float m[4][4]; };
Rot make_rot(float angle, vec3 axis_in) {
    const float c = cosf(angle), s = sinf(angle);
    const vec3 axis = crt::normalize(axis_in);
    const vec3 temp = (1.0f - c) * axis;
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
    const float I[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    Rot r;
    for (int col = 0; col < 3; col++)
        for (int k = 0; k < 4; k++) r.m[col][k] = I[0][k] * R[col][0] + I[1][k] * R[col][1] + I[2][k] * R[col][2];
    for (int k = 0; k < 4; k++) r.m[3][k] = I[3][k];
    return r;
}
void apply(const Rot& r, float* v) {
    float o[3];
    for (int k = 0; k < 3; k++) {
        const float a0 = r.m[0][k] * v[0] + r.m[1][k] * v[1];
        const float a1 = r.m[2][k] * v[2] + r.m[3][k] * 1.0f;
        o[k] = a0 + a1;
    }
    v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
}
}  // namespace

}  // extern "C"
namespace crt {
/* the per-frame rotation of main.cu:548 as 16 floats, column-major, for the device animation kernel */
void animation_matrix(float out16[16]) {
    const Rot rot = make_rot(0.01f, vec3(0.0f, 1.0f, 0.0f));
    for (int c = 0; c < 4; c++) for (int k = 0; k < 4; k++) out16[c * 4 + k] = rot.m[c][k];
}
}  // namespace crt
extern "C" {

int crt_animate_host(crt_prim* prims, int n) {
    if (!prims || n < 0) return CRT_ERR_ARG;
    const Rot rot = make_rot(0.01f, vec3(0.0f, 1.0f, 0.0f));
    for (int i = 0; i < n; i++) {
        crt_prim& s = prims[i];
        if (s.type == 1) {
            apply(rot, s.v1); apply(rot, s.v2); apply(rot, s.v3);
            apply(rot, s.v1Norm); apply(rot, s.v2Norm); apply(rot, s.v3Norm);   /* vec4(n,1): no translation in the matrix */
        } else if (s.type == 0) {
            vec3 d(s.direction);
            if (s.position[0] + s.radius > 200.0f) d = crt::reflect(d, vec3(-1, 0, 0));
            if (s.position[0] - s.radius < -200.0f) d = crt::reflect(d, vec3(1, 0, 0));
            if (s.position[1] + s.radius > 200.0f) d = crt::reflect(d, vec3(0, -1, 0));
            if (s.position[1] - s.radius < -200.0f) d = crt::reflect(d, vec3(0, 1, 0));
            if (s.position[2] + s.radius > 300.0f) d = crt::reflect(d, vec3(0, 0, -1));
            if (s.position[2] - s.radius < -300.0f) d = crt::reflect(d, vec3(0, 0, 1));
            d.store(s.direction);
            (vec3(s.position) + d * s.speed * 8.0f).store(s.position);
        }
    }
    return CRT_OK;
}

/* 24-bit BMP is bottom-up BGR: exactly the layout startRayTracing fills (Ray.cu:724-731). */
int crt_write_bmp(const char* path, const unsigned char* bgr, int width, int height) {
    if (!path || !bgr || width <= 0 || height <= 0) return CRT_ERR_ARG;
    FILE* f = fopen(path, "wb");
    if (!f) return CRT_ERR_IO;
    const int rowbytes = (width * 3 + 3) & ~3;
    const uint32_t datasz = (uint32_t)rowbytes * (uint32_t)height, filesz = 54 + datasz;
    unsigned char h[54] = {0};
    h[0] = 'B'; h[1] = 'M';
    memcpy(h + 2, &filesz, 4);
    uint32_t off = 54, hs = 40, planes_bpp = 1 | (24 << 16);
    int32_t w = width, ht = height;
    memcpy(h + 10, &off, 4); memcpy(h + 14, &hs, 4); memcpy(h + 18, &w, 4); memcpy(h + 22, &ht, 4);
    memcpy(h + 26, &planes_bpp, 4); memcpy(h + 34, &datasz, 4);
    fwrite(h, 1, 54, f);
    const unsigned char pad[3] = {0, 0, 0};
    for (int y = 0; y < height; y++) {
        fwrite(bgr + (size_t)y * width * 3, 1, (size_t)width * 3, f);
        fwrite(pad, 1, (size_t)(rowbytes - width * 3), f);
    }
    fclose(f);
    return CRT_OK;
}

}  // extern "C"
