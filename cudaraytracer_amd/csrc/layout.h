/* layout.h -- HBM layout of the packed scene + BVH ("scene blob").
 *
 * One contiguous device allocation, so that replicating the scene to another GPU is a single
 * broadcast.  Sections (all 256-byte aligned, offsets in BlobHeader):
 *
 *   nodes   exact child bounds SoA + child links + parent link; master copy for refit and dumps.
 *                            CRT_BVH_REFERENCE: nnodes x 128 B Node4 (two slots used), traversed as is.
 *                            CRT_BVH_WIDE:      nnodes x 256 B Node8.
 *   qnodes  nnodes x  80 B   CRT_BVH_WIDE only: the 8-wide tree compressed for traversal (QNode8):
 *                            node origin + per-axis power-of-two scale, 8-bit conservative child
 *                            planes, two base indices instead of links.
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
#include <math.h>

namespace crt {

constexpr uint32_t kBlobMagic = 0x43525442u;   /* "CRTB" */
constexpr uint32_t kBlobVersion = 5;
constexpr uint32_t kSphereTag = 0x7fc5fe3eu;   /* quiet NaN with a payload no arithmetic produces */
constexpr uint32_t kChildEmpty = 0xffffffffu;
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr int kLeafCountShift = 25;
constexpr uint32_t kLeafStartMask = (1u << kLeafCountShift) - 1u;
constexpr int kMaxLeafPrims = 64;

/* leaf link: bit31 | (count-1) << 25 | first slot */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t make_leaf(uint32_t start, uint32_t count) {
    return kLeafFlag | ((count - 1u) << kLeafCountShift) | start;
}

/* Sphere boxes of the wide trees.  intersectSphere (Ray.cu:105-133) evaluates disc = b*b - 4*a*(c.c + o.o - 2 c.o - r*r) in
 * float; the cancellation inside the bracket leaves an absolute error of up to E = kSphereNoiseK (|c| + |o|)^2 + kSphereNoiseR r^2
 * in disc / 4a (20 roundings of 2^-24 relative to (|c| + |o|)^2, 5 relative to r^2, counted operation by operation), so the
 * predicate can report a hit for a ray that passes the centre at any distance up to sqrt(r^2 + E) -- 0.026 for a sphere of radius 0
 * in the reference's room, which the randomised tests found -- and the hit point it reports lies in that ball.  The reference finds
 * such a hit whenever the ray crosses the sphere's 64-primitive leaf; a tree with other leaves finds it only if its boxes hold the
 * ball.  `origin_bound` bounds |o| for rays that start inside the scene or the room (BlobHeader::origin_bound, fixed when the tree is
 * built); a ray that starts further out widens its own slab pad by sqrt(kSphereNoiseK) * (|o| - origin_bound) instead (k_trace8's
 * ray set-up).  Returns the half side of the box. */
constexpr float kSphereNoiseK = 1.25e-6f, kSphereNoiseR = 4e-7f;
constexpr float kSphereNoiseSqrtK = 1.1181e-3f;        /* >= sqrt(kSphereNoiseK) */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline float sphere_box_half(float cx, float cy, float cz, float r, float origin_bound) {
    const float ax = cx < 0.f ? -cx : cx, ay = cy < 0.f ? -cy : cy, az = cz < 0.f ? -cz : cz;
    const float m = ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az);
    /* |c|: the Euclidean norm where its squares cannot overflow, sqrt(3) x the largest coordinate otherwise */
    const float norm = m < 1e18f ? sqrtf(cx * cx + cy * cy + cz * cz) * 1.000001f : 1.7320509f * m;
    const float e = kSphereNoiseSqrtK * (norm + origin_bound);
    const float M = r > e ? r : e;
    if (!(M > 0.f)) return 0.f;
    const float rr = r / M, ee = e / M;      /* scaled: r * r overflows for the radii the hostile tests use */
    return M * sqrtf(rr * rr + ee * ee + kSphereNoiseR * rr * rr) * 1.000001f;
}

struct alignas(128) Node4 {
    static constexpr int W = 4;
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    uint32_t child[4];        /* internal: node index; leaf: make_leaf(); unused: kChildEmpty */
    uint32_t parent;          /* node index of the parent, kChildEmpty for the root */
    uint32_t parent_slot;     /* which of the parent's 4 children this node is */
    uint32_t n_internal;      /* children that are nodes: arrival count the refit kernel waits for */
    uint32_t pad;
};
static_assert(sizeof(Node4) == 128, "Node4 must be 128 bytes");

/* Exact 8-wide node of the wide tree (CRT_BVH_WIDE): master copy for refit, dumps and the
 * quantiser.  Internal children of a node are consecutive nodes in slot order and its leaf
 * primitives one contiguous slot range, which is what lets the traversal node below carry two
 * base indices instead of eight links. */
struct alignas(256) Node8 {
    static constexpr int W = 8;
    float lox[8], loy[8], loz[8];
    float hix[8], hiy[8], hiz[8];
    uint32_t child[8];        /* same link format as Node4::child */
    uint32_t parent, parent_slot, n_internal, pad;
    uint32_t pad2[4];
};
static_assert(sizeof(Node8) == 256, "Node8 must be 256 bytes");

/* Compressed 8-wide traversal node, 80 B = 5 x 16-byte loads per lane per visit.  The traversal
 * kernel is bound by the vector L1's request rate (one 16-byte lane request per clock per CU when
 * every lane reads its own node, profiles/r01d), so bytes per visit are what count: an fp16-plane
 * variant of this node (128 B, 8 requests) was slower with fewer instructions.  Child k's box is
 * [o + lo*s, o + hi*s] per axis with s = 2^(e-127) and 8-bit plane
 * codes rounded outward, so the box contains the exact one.  Each 16-byte plane block holds the lo
 * and the hi row of one axis, so every byte loaded is used; near/far rows are picked by the sign of
 * the ray direction.  Slots are assigned by octant (slot bit 4/2/1 = child on the +x/+y/+z side of
 * the node centre): visiting hit children in descending (slot ^ R), R = the ray's octant code, is
 * a front-to-back order without computing or sorting distances. */
struct alignas(16) QNode8 {
    float ox, oy, oz;
    uint32_t exps_imask;      /* ex | ey << 8 | ez << 16 | imask << 24; imask bit k: child k is a node */
    uint32_t base_vmask;      /* first internal child (node index, < 2^24) | vmask << 24; vmask bit k: child k exists */
    uint32_t prim_base;       /* first slot of this node's leaf primitives */
    uint8_t meta[8];          /* leaf child k: 0x80 | (count-1) << 5 | (first slot - prim_base); else 0 */
    uint8_t lo_x[8], hi_x[8], lo_y[8], hi_y[8], lo_z[8], hi_z[8];
};
static_assert(sizeof(QNode8) == 80, "QNode8 must be 80 bytes");
constexpr int kQ8Grid = 255;            /* largest plane code */
/* A code q reaches the traversal's v_fma_mix_f32 as the fp16 value with bits 0x00:q -- a DENORMAL, q * 2^-24 exactly (fp16
 * denormals are on: .amdhsa_float_denorm_mode_16_64 3) -- so that the byte can come from either row word through one v_perm_b32
 * whose selector also says which row (near / far by the ray's direction); the stored scale exponent carries the 2^24 back. */
constexpr uint32_t kQ8Bias = 0;
constexpr uint32_t kQ8ScaleShift = 24;
constexpr int kWide8LeafMax = 4;        /* primitives per leaf of the wide tree (2 bits in meta) */

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
    uint32_t flags;           /* bit 0 (kBlobNonFiniteShading): the scene can shade to a non-finite colour, so paths whose coefficient
                               * is zero are traced on (crt_params.zero_paths); travels with the blob to the ranks that import it */
    float origin_bound;       /* wide trees: the |ray origin| their sphere boxes were padded for (sphere_box_half) */
    uint32_t pad[6];
};
constexpr uint32_t kBlobNonFiniteShading = 1u;
static_assert(sizeof(BlobHeader) % 16 == 0, "header alignment");

/* Device-side view: raw pointers into the blob. */
struct SceneView {
    const Node4* nodes;       /* CRT_BVH_REFERENCE only */
    const QNode8* qnodes;     /* CRT_BVH_WIDE only */
    const Geom* geom;
    const Shade* shade;
    const uint32_t* rank;
    const uint32_t* sidx;
    const Material* mats;
    uint32_t nprims, nnodes;
    uint32_t nmats;
};

}  // namespace crt
#endif
