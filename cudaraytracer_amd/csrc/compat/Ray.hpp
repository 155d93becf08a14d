/* compat/Ray.hpp -- startRayTracing with the reference's signature (reference Ray.hpp:35,
 * Ray.cu:665-742).  pixelcolorBuffer is the caller's width*height*3 bytes, BGR, bottom row first. */
#ifndef CRT_COMPAT_RAY_HPP
#define CRT_COMPAT_RAY_HPP
#include "SceneObjects.hpp"
#include "bvh.hpp"

extern int numBounces;          /* Ray.cu:27, default 2 */

void startRayTracing(float width, float height, unsigned char*& pixelcolorBuffer, glm::vec3 cameraPosition,
                     glm::vec3 cameraDirection, std::vector<SceneObject*>& scene, std::vector<Light>& lights, Node* rootnode);

/* last error of the process-wide context the wrappers use; empty when the last call succeeded.
 * (The reference ignores every CUDA error; the wrappers keep its void signatures and park the
 * error text here.) */
const char* crtCompatLastError();
#endif
