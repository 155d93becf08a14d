/* tools/crt_demo.cpp -- headless replacement of the reference's display loop (main.cu:140-651),
 * written against the reference's own API names (loadOBJ, SceneObject, Light, newBuild/refitTree/
 * freeTree, startRayTracing) through the compat wrappers.  The OpenGL window is replaced by BMP
 * output.  Frame policy: refit every frame, full rebuild every N-th (README: "every 20 frames";
 * main.cu:534 never resets curFrame and so rebuilds every frame after the 19th -- SURVEY.md 9.2).
 *
 *   crt_demo <teapot3.obj> <frames> <out_prefix> [rebuild_every=20] [bounces=2]
 */
#include "../cudaraytracer_amd/csrc/compat/Ray.hpp"
#include "../cudaraytracer_amd/csrc/compat/objloader.hpp"
#include "../include/crt.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: crt_demo teapot3.obj frames out_prefix [rebuild_every] [bounces]\n"); return 2; }
    const int frames = atoi(argv[2]);
    const std::string prefix = argv[3];
    const int rebuild_every = argc > 4 ? atoi(argv[4]) : 20;
    if (argc > 5) numBounces = atoi(argv[5]);
    float width = 720, height = 360;                                         /* main.cu:141-142 */

    std::vector<SceneObject*> scene;
    std::vector<glm::vec3> vertices, normals;
    std::vector<glm::vec2> uvs;
    if (!loadOBJ(argv[1], vertices, uvs, normals)) return 1;
    for (size_t i = 0; i + 2 < vertices.size(); i += 3) {                    /* main.cu:236-260 */
        SceneObject* s1 = new SceneObject();
        s1->objNum = (int)i; s1->radius = 0;
        s1->ambient = glm::vec3(0, 0, 1); s1->specular = glm::vec3(0.9, 0.4, 0); s1->diffuse = glm::vec3(0.8, 0.3, 0.1);
        s1->center = glm::vec3(0, 0, 0); s1->position = s1->center;
        s1->v1 = vertices[i]; s1->v1Norm = glm::normalize(normals[i]);
        s1->v2 = vertices[i + 1]; s1->v2Norm = glm::normalize(normals[i + 1]);
        s1->v3 = vertices[i + 2]; s1->v3Norm = glm::normalize(normals[i + 2]);
        s1->shininess = 64; s1->reflective = glm::vec3(.5, .5, .5); s1->triangle = true; s1->sphere = false;
        scene.push_back(s1);
    }
    {                                                                         /* main.cu:304-321 */
        SceneObject* s2 = new SceneObject();
        const int r[4] = {1804289383, 846930886, 1681692777, 1714636915};    /* unseeded glibc rand() */
        const float rmax = (float)2147483647;
        s2->direction = glm::vec3(-1 + ((float)r[0] / rmax * 2), -1 + ((float)r[1] / rmax * 2), -1 + ((float)r[2] / rmax * 2));
        s2->speed = (float)(.01 + ((float)r[3] / rmax * .3));
        s2->objNum = 1; s2->radius = 20;
        s2->ambient = glm::vec3(1.0, 1, 1); s2->specular = glm::vec3(.5, .5, .5); s2->diffuse = glm::vec3(1, 1, 1);
        s2->center = glm::vec3(-40, 40, 0); s2->position = s2->center;
        s2->reflective = glm::vec3(.5, .5, .5); s2->shininess = 64; s2->sphere = true;
        scene.push_back(s2);
    }
    glm::vec3 cameraDirection(0, 0, 0), cameraPosition(0, 50, 250);         /* main.cu:459,508 */
    std::vector<Light> lights;                                               /* main.cu:464-488 */
    Light l1; l1.direction = glm::vec3(2, -1, -1); l1.color = glm::vec3(1.0, 1.0, 1.0); lights.push_back(l1);
    Light l3; l3.direction = glm::vec3(-2, -1, 1); l3.color = glm::vec3(1, 1, 1); lights.push_back(l3);

    Node* root = new Node();
    std::deque<Node*> leafs;
    newBuild(root, leafs, scene);
    if (*crtCompatLastError()) return 1;
    unsigned char* buffer1 = (unsigned char*)malloc((size_t)width * (size_t)height * 3);

    /* animation state lives in crt_prim form so that the step is the library's (main.cu:545-573) */
    std::vector<crt_prim> anim(scene.size());
    double total_ms = 0;
    for (int f = 0; f < frames; f++) {
        auto t0 = std::chrono::steady_clock::now();
        if (f > 0) {
            if (rebuild_every > 0 && f % rebuild_every == 0) { freeTree(root); leafs.clear(); root = new Node(); newBuild(root, leafs, scene); }
            else refitTree(leafs);
        }
        startRayTracing(width, height, buffer1, cameraPosition, cameraDirection, scene, lights, root);
        if (*crtCompatLastError()) return 1;
        /* animate: marshal to crt_prim, step, copy the moved fields back */
        for (size_t i = 0; i < scene.size(); i++) {
            SceneObject& s = *scene[i]; crt_prim& p = anim[i];
            p.type = s.sphere ? 0 : 1; p.radius = s.radius; p.speed = s.speed;
            for (int k = 0; k < 3; k++) {
                p.position[k] = s.position[k]; p.direction[k] = s.direction[k];
                p.v1[k] = s.v1[k]; p.v2[k] = s.v2[k]; p.v3[k] = s.v3[k];
                p.v1Norm[k] = s.v1Norm[k]; p.v2Norm[k] = s.v2Norm[k]; p.v3Norm[k] = s.v3Norm[k];
            }
        }
        crt_animate_host(anim.data(), (int)anim.size());
        for (size_t i = 0; i < scene.size(); i++) {
            SceneObject& s = *scene[i]; const crt_prim& p = anim[i];
            for (int k = 0; k < 3; k++) {
                s.position[k] = p.position[k]; s.direction[k] = p.direction[k];
                s.v1[k] = p.v1[k]; s.v2[k] = p.v2[k]; s.v3[k] = p.v3[k];
                s.v1Norm[k] = p.v1Norm[k]; s.v2Norm[k] = p.v2Norm[k]; s.v3Norm[k] = p.v3Norm[k];
            }
        }
        auto t1 = std::chrono::steady_clock::now();
        total_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
        char name[512];
        snprintf(name, sizeof name, "%s_%03d.bmp", prefix.c_str(), f);
        if (crt_write_bmp(name, buffer1, (int)width, (int)height) != CRT_OK) return 1;
    }
    printf("%f ms/frame %f FPS\n", total_ms / frames, 1000.0 / (total_ms / frames));     /* main.cu:529 */
    freeTree(root);
    free(buffer1);
    return 0;
}
