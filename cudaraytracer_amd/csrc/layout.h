/* layout.h -- HBM layout of the packed scene + BVH ("scene blob").
 *
 * One contiguous device allocation, so that replicating the scene to another GPU is a single
 * broadcast.  Sections (all 256-byte aligned, offsets in BlobHeader):
 *
 *   nodes   nnodes x 128 B   4-wide BVH node: exact child bounds SoA + child links + parent link.
 *                            Master copy: refit and dumps work on it; the kernels traverse it only
 *                            for the reference topology.
 *   qnodes  nnodes x  64 B   the same tree quantised for traversal (wide mode): node origin + per-axis
 *                            power-of-two scale, 8-bit conservative child bounds, explicit child
 *                            links.  4 x 16-byte loads per lane per visit instead of 7.
 *   geom    nprims x  48 B   hit-test record, indexed by SLOT (leaf order of the packed tree)
 *                              triangle: v1.xyz v2.xyz v3.xyz Ng.xyz   (Ng precomputed exactly as Ray.cu:145)
 *                              sphere:   c.xyz r | dot(c,c) r*r 0 0 | 0 SPHERE_TAG 0 0
 *   shade   nprims x  48 B   touched only at the final hit: n1.xyz n2.xyz n3.xyz material-id pad pad
 *   rank    nprims x   4 B   position of the primitive in the REFERENCE's leaf order (tie-break, Ray.cu:199,216)
 *   sidx    nprims x   4 B   slot -> scene index (the parity id)
 *   slot_of nprims x   4 B   scene index -> slot (used when geometry is re-packed after an update)
 *   mats    nmats  x  64 B   ambient diffuse specular reflective shininess
 *
 * The reference instead chases 584-byte Node objects and 176-byte SceneObjects through pointers in
 * managed memory (bvh.hpp:14-33, SceneObjects.hpp:30-52).
 */
#ifndef CRT_LAYOUT_H
#define CRT_LAYOUT_H
#include <stdint.h>

namespace crt {

constexpr uint32_t kBlobMagic = 0x43525442u;   /* "CRTB" */
constexpr uint32_t kBlobVersion = 3;
constexpr uint32_t kSphereTag = 0x7fc5fe3eu;   /* quiet NaN with a payload no arithmetic produces */
constexpr uint32_t kChildEmpty = 0xffffffffu;
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr int kLeafCountShift = 25;
constexpr uint32_t kLeafStartMask = (1u << kLeafCountShift) - 1u;
constexpr int kMaxLeafPrims = 64;

/* leaf link: bit31 | (count-1) << 25 | first slot */
static inline uint32_t make_leaf(uint32_t start, uint32_t count) {
    return kLeafFlag | ((count - 1u) << kLeafCountShift) | start;
}

struct alignas(128) Node4 {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    uint32_t child[4];        /* internal: node index; leaf: make_leaf(); unused: kChildEmpty */
    uint32_t parent;          /* node index of the parent, kChildEmpty for the root */
    uint32_t parent_slot;     /* which of the parent's 4 children this node is */
    uint32_t n_internal;      /* children that are nodes: arrival count the refit kernel waits for */
    uint32_t pad;
};
static_assert(sizeof(Node4) == 128, "Node4 must be 128 bytes");

/* Quantised traversal node, 64 B = 4 x 16-byte loads per lane per visit (the exact node needs 7).
 * Child k's box is [o + qlo*s, o + qhi*s] per axis with s = 2^(e-127) taken from the exponent
 * byte; quantisation rounds outward, so the box contains the exact one.  Child links are the
 * same 32-bit links as Node4::child (explicit: deriving them from base indices cost ~35 VALU
 * instructions per visit in a kernel that is bound by VALU issue). */
struct alignas(16) QNode {
    float ox, oy, oz;
    uint8_t ex, ey, ez, pad0;
    uint8_t qlo[3][4];
    uint8_t qhi[3][4];
    uint32_t pad1[2];
    uint32_t child[4];
};
static_assert(sizeof(QNode) == 64, "QNode must be 64 bytes");

struct alignas(16) Geom { float w[12]; };
struct alignas(16) Shade { float n[9]; uint32_t mat; uint32_t pad[2]; };
struct alignas(16) Material { float ambient[3], diffuse[3], specular[3], reflective[3]; float shininess; float pad[3]; };
static_assert(sizeof(Geom) == 48 && sizeof(Shade) == 48 && sizeof(Material) == 64, "record sizes");

struct BlobHeader {
    uint32_t magic, version;
    uint32_t nprims, nnodes, nmats, depth;
    uint32_t mode;            /* CRT_BVH_WIDE / CRT_BVH_REFERENCE */
    uint32_t root_is_leaf;    /* reference topology with <=64 primitives: a single leaf (Ray.cu:186-188) */
    uint64_t off_nodes, off_qnodes, off_geom, off_shade, off_rank, off_sidx, off_slot_of, off_mats;
    uint64_t total_bytes;
    float root_box[6];        /* minX,maxX,minY,maxY,minZ,maxZ written by build and refit */
    uint32_t pad[8];
};
static_assert(sizeof(BlobHeader) % 16 == 0, "header alignment");

/* Device-side view: raw pointers into the blob. */
struct SceneView {
    const Node4* nodes;
    const QNode* qnodes;
    const Geom* geom;
    const Shade* shade;
    const uint32_t* rank;
    const uint32_t* sidx;
    const Material* mats;
    uint32_t nprims, nnodes;
};

}  // namespace crt
#endif
