/* kernels.h -- launch interface between the C-ABI host layer (crt_api.cpp) and the HIP kernels
 * (kernels.hip).  Everything here is plain data; no HIP types leak into crt.h. */
#ifndef CRT_KERNELS_H
#define CRT_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "layout.h"
#include "../../include/crt.h"

namespace crt {

/* tunables (overridable with -D for A/B builds, see tools/build_variant.sh) */
#ifndef CRT_STACK_DEPTH
#define CRT_STACK_DEPTH 32
#endif
#ifndef CRT_REFILL_MIN_IDLE
#define CRT_REFILL_MIN_IDLE 8
#endif
#ifndef CRT_TRACE_MIN_WAVES
#define CRT_TRACE_MIN_WAVES 0          /* 0: let the compiler choose; n: __launch_bounds__(256, n) waves per SIMD */
#endif
/* vote: a node step runs when n_node * PRIM_WEIGHT >= n_prim * NODE_WEIGHT (1:1 = plain majority) */
#ifndef CRT_VOTE_PRIM_WEIGHT
#define CRT_VOTE_PRIM_WEIGHT 1
#endif
#ifndef CRT_VOTE_NODE_WEIGHT
#define CRT_VOTE_NODE_WEIGHT 3
#endif
#ifndef CRT_FETCH_CHUNK
#define CRT_FETCH_CHUNK 64
#endif
constexpr int kFetchChunk = CRT_FETCH_CHUNK;       /* rays a wave takes from the queue per atomic */
constexpr int kFetchStride = 64;       /* uint32 words between the 8 per-region fetch counters (256 B apart) */
constexpr int kFetchDoneWord = 32;     /* word of the same zeroed block (its own 128-byte line): bit r = region r is used up */
constexpr int kTraceBlock = 256;       /* 4 waves of 64 */
constexpr int kStackDepth = CRT_STACK_DEPTH;        /* per-lane entries in LDS: stack[depth][lane], conflict-free per wave */
#ifndef CRT_REFILL8_MIN_IDLE
#define CRT_REFILL8_MIN_IDLE 12         /* k_trace8: idle lanes that trigger a refill (its loads complete under the next traversal step) */
#endif
#ifndef CRT_TRACE8_MIN_WAVES
#define CRT_TRACE8_MIN_WAVES 5         /* __launch_bounds__(256, n): 96 VGPRs, 5 waves per SIMD; 5 x 30.7 KB of LDS per CU */
#endif
#ifndef CRT_STACK_DEPTH8
#define CRT_STACK_DEPTH8 12         /* 24 KB; with the 2 KB octant table and 4.3 KB of work-sharing records a block stays at the 30,720 B that let 5 blocks share a CU */
#endif
constexpr int kRefillMinIdle8 = CRT_REFILL8_MIN_IDLE;
#ifndef CRT_STEAL_MIN_IDLE
#define CRT_STEAL_MIN_IDLE 16          /* k_trace8, queue dry: idle lanes of a wave that start a steal step */
#endif
#ifndef CRT_STEAL_EVERY
#define CRT_STEAL_EVERY 4              /* ... at most every n-th iteration (power of two) */
#endif
constexpr int kStealMinIdle = CRT_STEAL_MIN_IDLE;
constexpr int kStealEvery = CRT_STEAL_EVERY;
constexpr int kStackDepth8 = CRT_STACK_DEPTH8;      /* k_trace8: 64-bit entries, at most one per tree level */
/* in-LDS stack depth of the k_trace8 variant with `block` threads: chosen so that 5 waves per SIMD (20 per CU) fit the
 * CU's 160 KB of LDS -- 256: 12 rows (30.7 KB x 5), 128: 11 rows (15.5 KB x 10), 64: 9 rows (7.7 KB x 20); deeper entries
 * go to the per-thread overflow rows in HBM */
constexpr int stack_depth8_of(int block) { return block >= 256 ? kStackDepth8 : (block == 128 ? (kStackDepth8 < 11 ? kStackDepth8 : 11) : (kStackDepth8 < 9 ? kStackDepth8 : 9)); }
/* k_trace8 vote: a node step runs when n_node * PRIM_WEIGHT >= n_prim * NODE_WEIGHT */
#ifndef CRT_VOTE8_PRIM_WEIGHT
#define CRT_VOTE8_PRIM_WEIGHT 1
#endif
#ifndef CRT_VOTE8_NODE_WEIGHT
#define CRT_VOTE8_NODE_WEIGHT 3
#endif
constexpr int kRefillMinIdle = CRT_REFILL_MIN_IDLE; /* a wave refills from the ray queue when >= this many lanes are idle */

/* Camera basis of Ray.cu:684-689 plus the tile map of this rank. */
struct FrameParams {
    float L[3], u[3], v[3], cam[3];
    int32_t width, height;
    int32_t tile_w, tile_h, tiles_x;
    int32_t ntiles_owned;
    int32_t spp_n, sample;            /* spp = spp_n^2; sample in [0, spp) */
    int32_t bounce, bounces;
    int32_t capture;
    int32_t keep_zero;                /* emit the next ray of a path whose coefficient is (0,0,0) too (crt_params.zero_paths) */
};

/* Shade treats every light as one or more SLOTS: a directional or point light owns one, an area light one per
 * sample.  Slot s of queue entry i owns contrib/shflag[s * cap + i]. */
struct LightsParam {
    crt_light l[CRT_MAX_LIGHTS];
    int32_t n;
    int32_t slot_base[CRT_MAX_LIGHTS];
    int32_t nslots;
    int32_t area_samples;             /* shadow rays per area light and shading point (crt_params.area_samples) */
};

/* A shadow ray carries in rayD.w what the traversal needs to know about its light, as float bits:
 *   -inf                  directional light: shadowed iff a hit exists and the nearest t > 1e-9 (Ray.cu:650-656)
 *   +dtoLight (sign 0)    point light: nearest hit, shadowed iff dtoLight > |t * dir| (Ray.cu:621-629)
 *   -dtoLight (sign 1)    area-light sample: as the point light, and only if the nearest t > 1e-9 (Ray.cu:587-596) */
constexpr uint32_t kShadowDirectional = 0xff800000u;
/* shflag[slot * cap + i], one byte per path and light slot: 0 lit so far (a traversal may set 1), 1 shadowed (the record in
 * contrib still counts: distance * 0, Ray.cu:629,655), and two values only k_shade writes: */
constexpr uint8_t kShDeadPath = 2;          /* the path hit neither object nor wall: it ended, nothing is added (all slots carry it) */
constexpr uint8_t kShNoContribution = 3;    /* shadowed without a ray and the term is finite: adds an exact zero, contrib not written */

/* Area lights (commented out in the reference, Ray.cu:563-603): sample k of a shading point
 * takes entry (h + k) mod kAreaTable of a table of unit-disk samples (u, cos(theta), sin(theta)) built on the
 * host, where h hashes (pixel, light, bounce, sub-pixel sample). */
constexpr int kAreaTable = 1024;
constexpr int kAreaSamplesMax = 64;
__host__ __device__ static inline uint32_t area_hash(uint32_t pixel, uint32_t light, uint32_t bounce, uint32_t sample) {
    uint32_t h = pixel * 0x9E3779B1u + light * 0x85EBCA77u + bounce * 0xC2B2AE3Du + sample * 0x27D4EB2Fu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}

/* Per-frame device buffers.  Queue entry i of the current bounce owns rayO/rayD/coef/pix[i];
 * shading writes rc at [i] and contrib/shflag at [s*cap + i] for light slot s. */
struct FrameBuffers {
    float4* rayO; float4* rayD; float4* coef; uint32_t* pix;           /* current queue */
    float4* nrayO; float4* nrayD; float4* ncoef; uint32_t* npix;       /* next-bounce queue */
    int32_t* hitSlot; float* hitT;
    float4* contrib; uint8_t* shflag;
    const float4* area_table;                                          /* kAreaTable entries (u, cos, sin, 0) */
    float4* shO; float4* shD;                                          /* shadow-ray queue: .w = target / light */
    float4* accum;                                                     /* per pixel float colour (isect.color) */
    int32_t* cap_ids; float* cap_ts;                                   /* capture planes [bounce][pixel] or null */
    uint32_t cap;                                                      /* queue capacity: stride of the per-light planes */
    uint32_t cap_stride;                                               /* pixels per capture plane */
    const uint32_t* tile_ids;
};

enum TraceMode { TRACE_NEAREST = 0, TRACE_SHADOW = 1 };

struct TraceArgs {
    SceneView sv;
    const float4* rayO; const float4* rayD;
    const uint32_t* count;       /* number of rays (device); k_trace8: the nearest-hit queue, null = none */
    const uint32_t* perm;        /* optional visiting order (crt_params.sort_rays): queue position -> ray, or null */
    const float4* shO; const float4* shD;    /* k_trace8 only: the shadow queue served by the same launch, */
    const uint32_t* shcount;                 /* null = none (k_trace takes its one queue in rayO/rayD/count) */
    const uint32_t* shperm;
    uint32_t* next;              /* 8 fetch counters, kFetchStride words apart (device, zeroed) */
    int32_t* hitSlot; float* hitT;           /* NEAREST outputs */
    uint8_t* shflag;                         /* SHADOW output, indexed by the target in rayO.w */
    float scene_abs;             /* max |coordinate| of the root box: scales the slab-test pad */
    float origin_bound;          /* k_trace8: the |ray origin| the tree's sphere boxes are padded for (BlobHeader::origin_bound); a ray from
                                  * further out widens its own pad (layout.h sphere_box_half) */
    uint32_t* ovf; int32_t ovf_depth;        /* per-thread stack overflow area in HBM: ovf_depth entries of 4 B (k_trace) or 8 B (k_trace8) */
    unsigned long long* counters;            /* [0]=box [1]=tri [2]=sph tests [3]=node visits, when counting */
    uint32_t* status;            /* device error word: bit0 = stack overflow beyond ovf, bit1 = node not quantisable */
    int32_t steal_min_idle, steal_every;     /* k_trace8 tail (set by launch_trace): idle lanes that start a steal step; at most every n-th iteration */
    int32_t block;                           /* k_trace8: threads per block, 256 (also 0) / 128 / 64 */
    int32_t tail_probe;                      /* CRT_TAIL_PROBE=1: waves report how long they ran after the queue was dry (counters[30], [31]) */
};

int trace_grid_blocks(int device);
bool trace8_has_block(int block);
void launch_trace(const TraceArgs& a, int mode, bool counting, bool wide, int blocks, hipStream_t s);
void launch_quantize(const Node8* nodes, uint32_t nnodes, QNode8* qnodes, uint32_t* status, hipStream_t s);
void launch_gen_primary(const FrameParams& fp, const FrameBuffers& fb, uint32_t* count, hipStream_t s);
void launch_shade(const FrameParams& fp, const FrameBuffers& fb, const SceneView& sv, const LightsParam& lights,
                  const uint32_t* count, uint32_t* next_count, uint32_t* shadow_count, hipStream_t s);
void launch_resolve(const FrameBuffers& fb, const LightsParam& lights, const uint32_t* count, int bounce, hipStream_t s);
void launch_convert(const FrameParams& fp, const FrameBuffers& fb, const float4* sum_or_null, uint8_t* bgr, hipStream_t s);
void launch_spp_accumulate(const FrameParams& fp, const FrameBuffers& fb, float4* sum, hipStream_t s);
/* optional ray sorting (crt_params.sort_rays): visiting order of a ray queue by the Morton code of the
 * origin's cell in a 32^3 grid over the root box.  keys/perm: one word per ray; hist: kSortBins + 1 words. */
constexpr int kSortBits = 5;
constexpr int kSortBins = 1 << (3 * kSortBits);
void launch_sort_rays(const float4* rayO, const uint32_t* count, uint32_t cap, const float* root_box6, uint32_t* keys,
                      uint32_t* hist, uint32_t* perm, hipStream_t s);
/* device-side build of the wide tree (lbvh.hip): scratch sized by lbvh_scratch_bytes(n); nodes_out holds up to
 * node_cap (>= n) Node8 records, slot_to_scene n words; the node count and depth come back to the host */
size_t lbvh_scratch_bytes(uint32_t n);
/* root_box: of the tree (sphere boxes padded for origin_bound, layout.h sphere_box_half); exact_box: of the primitives as they
 * are -- the scene box the reference's own split starts from (refrank_build) */
int lbvh_build(void* scratch, const crt_prim* d_prims, uint32_t n, float origin_bound, Node8* nodes_out, uint32_t node_cap, uint32_t* slot_to_scene,
               uint32_t* h_nnodes, uint32_t* h_depth, float root_box[6], float exact_box[6], hipStream_t s, std::string* err);
/* rank_of_scene: the reference's leaf-order rank of every primitive (refrank.hip), or null = ties resolve by scene index */
void launch_lbvh_maps(const uint32_t* slot_to_scene, const uint32_t* rank_of_scene, uint32_t n, uint32_t* sidx, uint32_t* slot_of, uint32_t* rank, hipStream_t s);
/* the reference's centroid-mean partition replayed on the device (refrank.hip): rank_of_scene[i] = position of primitive i in
 * the reference tree's leaf order.  CRT_ERR_BUILD where the reference's recursion does not terminate. */
size_t refrank_scratch_bytes(uint32_t n);
int refrank_build(void* scratch, const crt_prim* d_prims, uint32_t n, const float root_box[6], uint32_t* d_rank_of_scene, uint32_t* h_levels,
                  hipStream_t s, std::string* err);
struct AnimParam { float m[16]; };          /* column-major rotation of main.cu:548 */
void animation_matrix(float out16[16]);     /* scene.cpp */
void launch_animate(crt_prim* d_prims, uint32_t n, const AnimParam& a, hipStream_t s);
void launch_pack_prims(const crt_prim* d_prims, const uint32_t* d_matid, const uint32_t* slot_of, uint32_t n,
                       Geom* geom, Shade* shade, hipStream_t s);
void launch_refit(Node4* nodes, uint32_t nnodes, const Geom* geom, uint32_t* arrive, float* root_box, int grow_only,
                  hipStream_t s);
void launch_refit8(Node8* nodes, uint32_t nnodes, const Geom* geom, float origin_bound, uint32_t* arrive, float* root_box, int grow_only,
                   hipStream_t s);      /* origin_bound: BlobHeader::origin_bound, sphere boxes are padded for it */
void launch_slots_to_scene(const int32_t* slots, const uint32_t* sidx, int32_t* out, uint32_t n, hipStream_t s);
void launch_powf_probe(const float* x, const float* y, uint32_t n, float* out, hipStream_t s);
void launch_pack_rays(const float* org_dir, uint32_t n, float4* rayO, float4* rayD, bool shadow, hipStream_t s);
void launch_tiles_copy(const FrameParams& fp, const uint32_t* tile_ids, int ntiles, const uint8_t* src, uint8_t* dst,
                       int to_packed, hipStream_t s);

}  // namespace crt
#endif
