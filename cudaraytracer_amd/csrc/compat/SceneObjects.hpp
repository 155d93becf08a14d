/* compat/SceneObjects.hpp -- the reference's scene classes, same names and fields
 * (reference SceneObjects.hpp:16-52), so application code written against them compiles unchanged. */
#ifndef CRT_COMPAT_SCENEOBJECTS_HPP
#define CRT_COMPAT_SCENEOBJECTS_HPP
#include "glm_lite.hpp"
#include <vector>

class Light {
public:
    glm::vec3 color;
    glm::vec3 direction;
    bool point = false;
    float constantTerm = 0, linearTerm = 0, quadraticTerm = 0;
    glm::vec3 position;
    bool area = false;
    float radius = 0;
};

class SceneObject {
public:
    int objNum = 0;
    glm::vec3 position, ambient, diffuse, specular, reflective, direction;
    float speed = 0, radius = 0;
    glm::vec3 center;
    float shininess = 0;
    bool sphere = false, triangle = false;
    glm::vec3 v1, v1Norm, v2, v2Norm, v3, v3Norm;
};
#endif
