/* oracle/ovec.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Minimal float vec3 whose operation ORDER restates the GLM 0.9.7.6 templates the reference
 * instantiates (the library is vendored inside the reference, so the order is readable):
 *   dot       glm/glm/detail/func_geometric.inl:65-72   tmp = x*y; tmp.x + tmp.y + tmp.z
 *   cross     glm/glm/detail/func_geometric.inl:132-141
 *   normalize glm/glm/detail/func_geometric.inl:154-159 x * inversesqrt(dot(x,x))
 *   inversesqrt glm/glm/detail/func_exponential.inl:150-153  1 / sqrt(x)
 *   reflect   glm/glm/detail/func_geometric.inl:176-179 I - N * dot(N,I) * 2
 *   vec/scalar glm/glm/detail/type_vec3.inl:703-709     true division per component
 * Pinned bit-for-bit against the vendored headers by oracle/ref_driver.cpp (tests/test_oracle_pins.py).
 * Must be compiled with -ffp-contract=off and without fast-math.
 */
#ifndef ORACLE_OVEC_H
#define ORACLE_OVEC_H
#include <math.h>

struct ov3 {
    float x, y, z;
    ov3() : x(0.f), y(0.f), z(0.f) {}          /* GLM 0.9.7 zero-initialises (type_vec3.inl:37-43) */
    ov3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit ov3(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    void store(float* p) const { p[0] = x; p[1] = y; p[2] = z; }
};

static inline ov3 operator+(const ov3& a, const ov3& b) { return ov3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ov3 operator-(const ov3& a, const ov3& b) { return ov3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ov3 operator*(const ov3& a, const ov3& b) { return ov3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline ov3 operator*(const ov3& a, float s) { return ov3(a.x * s, a.y * s, a.z * s); }
static inline ov3 operator*(float s, const ov3& a) { return ov3(s * a.x, s * a.y, s * a.z); }
static inline ov3 operator/(const ov3& a, float s) { return ov3(a.x / s, a.y / s, a.z / s); }
static inline ov3 operator-(const ov3& a) { return ov3(-a.x, -a.y, -a.z); }
static inline ov3& operator+=(ov3& a, const ov3& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }

static inline float odot(const ov3& a, const ov3& b) {
    ov3 t(a * b);
    return t.x + t.y + t.z;
}
static inline ov3 ocross(const ov3& x, const ov3& y) {
    return ov3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline ov3 onormalize(const ov3& v) { return v * (1.0f / sqrtf(odot(v, v))); }
static inline ov3 oreflect(const ov3& I, const ov3& N) { return I - N * odot(N, I) * 2.0f; }
/* glm::min(x,y) = x < y ? x : y (func_common.inl:409-414); glm::max(x,y) = x > y ? x : y (:430-435);
 * clamp = min(max(x,lo),hi) (:451-456).  NaN first operand therefore yields the second. */
static inline float omax(float x, float y) { return (x > y) ? x : y; }
static inline float omin(float x, float y) { return (x < y) ? x : y; }
static inline float oclamp(float x, float lo, float hi) { return omin(omax(x, lo), hi); }

#endif
