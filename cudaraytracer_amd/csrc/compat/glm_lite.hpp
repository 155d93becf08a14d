/* compat/glm_lite.hpp -- the few GLM names the reference's scene/BVH/render API spells in its
 * signatures (glm::vec3, glm::vec2 and the free functions its app code calls on them), backed by
 * vec.h so that the operation order is GLM 0.9.7.6's.  An application that already has GLM defines
 * CRT_COMPAT_USE_GLM before including the compat headers and gets the real library instead: the
 * layouts are identical (three packed floats). */
#ifndef CRT_COMPAT_GLM_LITE_HPP
#define CRT_COMPAT_GLM_LITE_HPP
#ifdef CRT_COMPAT_USE_GLM
#include <glm/glm.hpp>
#else
#include "../vec.h"
namespace glm {
struct vec2 { float x, y; vec2() : x(0), y(0) {} vec2(float a, float b) : x(a), y(b) {} };
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    template <class A, class B, class C> vec3(A a, B b, C c) : x((float)a), y((float)b), z((float)c) {}
    vec3(const crt::vec3& v) : x(v.x), y(v.y), z(v.z) {}
    operator crt::vec3() const { return crt::vec3(x, y, z); }
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
    vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return crt::vec3(a) + crt::vec3(b); }
inline vec3 operator-(const vec3& a, const vec3& b) { return crt::vec3(a) - crt::vec3(b); }
inline vec3 operator*(const vec3& a, const vec3& b) { return crt::vec3(a) * crt::vec3(b); }
inline vec3 operator*(const vec3& a, float s) { return crt::vec3(a) * s; }
inline vec3 operator*(float s, const vec3& a) { return s * crt::vec3(a); }
inline vec3 operator/(const vec3& a, float s) { return crt::vec3(a) / s; }
inline vec3 operator-(const vec3& a) { return -crt::vec3(a); }
inline float dot(const vec3& a, const vec3& b) { return crt::dot(a, b); }
inline vec3 cross(const vec3& a, const vec3& b) { return crt::cross(a, b); }
inline vec3 normalize(const vec3& a) { return crt::normalize(a); }
inline vec3 reflect(const vec3& i, const vec3& n) { return crt::reflect(i, n); }
}  // namespace glm
#endif
#endif
