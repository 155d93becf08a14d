/* compat/objloader.hpp -- loadOBJ with the reference's signature (reference objloader.hpp:4-9). */
#ifndef CRT_COMPAT_OBJLOADER_HPP
#define CRT_COMPAT_OBJLOADER_HPP
#include "glm_lite.hpp"
#include <vector>
bool loadOBJ(const char* path, std::vector<glm::vec3>& out_vertices, std::vector<glm::vec2>& out_uvs,
             std::vector<glm::vec3>& out_normals);
#endif
