/* oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY; built into oracle/_ref/ (git-ignored).
 *
 * Our own driver, compiled TOGETHER WITH the reference's objloader.cpp and its vendored GLM
 * headers, taken where they lie under /root/reference (nothing is copied into the repo).
 * These are the only reference sources on the render path that compile in this image without
 * stand-ins: Ray.cu and bvh.cu need nvcc's launch syntax and the CUDA runtime.
 *
 *   ref_driver math <in.bin> <out.bin>   n x 9 floats -> n x 32 floats, same probe as orc_math_probe
 *   ref_driver obj  <file.obj> <out.bin> loadOBJ -> int32 ncorners, verts[3n], normals[3n]
 */
#include "glm/glm/glm.hpp"
#include "glm/glm/gtc/matrix_transform.hpp"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include "objloader.hpp"   /* needs <vector> first, as objloader.cpp includes it */

static void put3(float* o, const glm::vec3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: ref_driver math|obj in out\n"); return 2; }
    if (!strcmp(argv[1], "math")) {
        FILE* f = fopen(argv[2], "rb"); if (!f) return 3;
        fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
        int n = (int)(sz / (9 * sizeof(float)));
        std::vector<float> in(9 * (size_t)n), out(32 * (size_t)n);
        if (fread(in.data(), sizeof(float), in.size(), f) != in.size()) return 3;
        fclose(f);
        for (int i = 0; i < n; i++) {
            glm::vec3 a(in[9 * i], in[9 * i + 1], in[9 * i + 2]), b(in[9 * i + 3], in[9 * i + 4], in[9 * i + 5]),
                c(in[9 * i + 6], in[9 * i + 7], in[9 * i + 8]);
            float* o = &out[32 * (size_t)i];
            o[0] = glm::dot(a, b);
            put3(o + 1, glm::cross(a, b));
            put3(o + 4, glm::normalize(a));
            put3(o + 7, glm::reflect(a, b));
            put3(o + 10, a / b.x);
            put3(o + 13, a * b.x);
            put3(o + 16, b.x * a);
            put3(o + 19, glm::clamp(a * 255.0f, glm::vec3(0.0f, 0.0f, 0.0f), glm::vec3(255.0f, 255.0f, 255.0f)));
            o[22] = glm::max(a.x, b.x);
            o[23] = glm::min(a.x, b.x);
            glm::mat4 trans = glm::mat4(1.0);
            trans = glm::rotate(trans, 0.01f, glm::vec3(0.0f, 1.0f, 0.0f));
            put3(o + 24, glm::vec3(trans * glm::vec4(a, 1.0)));
            put3(o + 27, a * b * c);
            o[30] = glm::pow(std::fabs(a.x), c.x);
            o[31] = (a + b * c.x - c).x;
        }
        f = fopen(argv[3], "wb"); if (!f) return 3;
        fwrite(out.data(), sizeof(float), out.size(), f); fclose(f);
        return 0;
    }
    if (!strcmp(argv[1], "obj")) {
        std::vector<glm::vec3> v, nrm; std::vector<glm::vec2> uv;
        if (!loadOBJ(argv[2], v, uv, nrm)) return 4;
        FILE* f = fopen(argv[3], "wb"); if (!f) return 3;
        int n = (int)v.size();
        fwrite(&n, sizeof(int), 1, f);
        fwrite(v.data(), sizeof(glm::vec3), v.size(), f);
        fwrite(nrm.data(), sizeof(glm::vec3), nrm.size(), f);
        fclose(f);
        return 0;
    }
    return 2;
}
