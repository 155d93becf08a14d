/* include/crt.h -- C ABI of the MI355X-native render path (libcrt_amd.so).
 *
 * Drop-in boundary for boonemiller/CudaRayTracer's per-pixel render path.  The reference has no
 * plugin system: its boundary is four free C++ functions plus three POD classes.
 * Each entry point below names the reference interface it replaces.  Plain pointers and sizes
 * only; every function returns 0 on success or a negative crt_status; no exceptions cross the
 * boundary.  One context per device; calls on one context are serialised by the caller.
 * Thin C++ wrappers that keep the reference's own names (startRayTracing, constructTree,
 * refitTree, freeTree, loadOBJ) live in cudaraytracer_amd/csrc/compat/.
 */
#ifndef CRT_H
#define CRT_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CRT_MAX_LIGHTS 8
#define CRT_MAX_PRIMS ((1 << 25) - 64)   /* leaf links hold a 25-bit slot; the compressed wide tree (CRT_BVH_WIDE*) holds 24-bit
                                          * node and slot bases: beyond 2^24 primitives its build returns CRT_ERR_BUILD */

typedef enum crt_status {
    CRT_OK = 0,
    CRT_ERR_ARG = -1,        /* bad argument / size out of range */
    CRT_ERR_STATE = -2,      /* call order: e.g. render before crt_bvh_build */
    CRT_ERR_HIP = -3,        /* a HIP runtime call failed: see crt_last_error */
    CRT_ERR_NOMEM = -4,
    CRT_ERR_IO = -5,
    CRT_ERR_BUILD = -6       /* BVH build failed (e.g. the reference's runaway recursion case) */
} crt_status;

/* POD mirror of class SceneObject (SceneObjects.hpp:30-52): geometry + material + animation
 * state.  `type` replaces the sphere/triangle bools; `center` (set but never read by the render
 * path) is dropped. */
typedef struct crt_prim {
    int32_t type;            /* 0 = sphere, 1 = triangle */
    int32_t objNum;          /* app-assigned, not an ID; carried, never read */
    float position[3];
    float radius;
    float v1[3], v2[3], v3[3];
    float v1Norm[3], v2Norm[3], v3Norm[3];
    float ambient[3], diffuse[3], specular[3], reflective[3];
    float shininess;
    float direction[3];      /* sphere animation (main.cu:558-572) */
    float speed;
} crt_prim;

/* POD mirror of class Light (SceneObjects.hpp:16-28). */
typedef struct crt_light {
    float color[3];
    float direction[3];
    int32_t point;           /* non-zero: point light (Ray.cu:604-632), else directional (Ray.cu:633-658) */
    float constantTerm, linearTerm, quadraticTerm;
    float position[3];
    int32_t area;            /* non-zero: area light (takes precedence over `point`): a disk of `radius` in the XZ plane
                              * about `position`, sampled crt_params.area_samples times per shading point.  The
                              * reference has this branch commented out (Ray.cu:563-603); see crt_area_table */
    float radius;
} crt_light;

/* startRayTracing's camera arguments (Ray.hpp:35): position and look-at point. */
typedef struct crt_camera {
    float position[3];
    float target[3];         /* the reference calls this "cameraDirection"; it is a look-at point (main.cu:459) */
} crt_camera;

/* The reference's knobs are globals and literals (Ray.cu:25-28, main.cu:141-142); defaults from
 * crt_params_default() equal them. */
typedef struct crt_params {
    int32_t width, height;   /* 720 x 360 */
    int32_t bounces;         /* numBounces = 2: passes = 1 primary + (bounces-1) reflections (Ray.cu:715) */
    int32_t spp;             /* 1; n*n stratified samples when a perfect square > 1 (extension) */
    int32_t tile_w, tile_h;  /* image tile for sharding, default 32 x 32 */
    int32_t rank, nranks;    /* this context renders tiles t with crt_tile_owner(t, nranks) == rank; 0, 1 */
    int32_t sort_rays;       /* 0/1: visit reflection and shadow rays in Morton order of their origin's cell (32^3 grid over
                              * the scene box; wide BVH only).  Changes the visiting order only, never the image. */
    int32_t area_samples;    /* shadow rays per area light and shading point, 1..64; 0 = 30 (lightSamples, Ray.cu:566) */
    int32_t use_graph;       /* 0/1: capture the frame's launch sequence in a HIP graph and replay it while camera, parameters,
                              * lights and buffers stay the same (scene contents may move: animation + refit).  For small,
                              * launch-bound frames; per-kernel times in crt_stats are then 0, ms_total stays */
    int32_t tile_sub, tile_nsub; /* tile_nsub > 1: this call renders only every tile_nsub-th tile of the rank's tile list, starting
                              * at tile_sub (a frame split over several frame slots); the rank's tile map -- what pack / unpack
                              * and the other ranks see -- is unchanged.  0, 0 */
    int32_t grid_div;        /* > 1: the persistent traversal grid is the chip's resident capacity divided by this (frames in
                              * flight on several slots share the chip side by side instead of taking turns).  0 */
    int32_t timing;          /* 0: HIP events around every kernel (crt_stats ms_trace, ms_shade ...); 1: frame start / end only */
    int32_t zero_paths;      /* paths whose reflection coefficient has become exactly (0,0,0) -- every path after a wall, Ray.cu:257,262,
                              * or after a material that reflects nothing.  The reference traces them on and adds 0 * colour
                              * (Ray.cu:660, 715-723), which changes the pixel only when that colour is not finite (pow of a
                              * negative base with a fractional shininess, a zero-length normal ...): the pixel then turns NaN.
                              * 0: end such paths when the scene and the lights cannot produce a non-finite term (materials finite
                              * with an integral shininess in [0, 1e6], sphere radii > 0, vertex normals on one side of their
                              * triangle, lights finite with a direction), trace them on otherwise -- the image is the reference's
                              * either way; 1: always trace them on (the reference's ray count); 2: always end them */
} crt_params;

typedef struct crt_stats {
    uint64_t rays_primary, rays_shadow, rays_reflect;   /* traversals issued by the last crt_render* call */
    /* filled only when crt_set_counting(ctx,1): AABB / triangle / sphere tests of the nearest-hit
     * launches and of the shadow launches (the counts behind the algorithmic-bytes figure) */
    uint64_t box_tests, tri_tests, sph_tests;
    uint64_t box_tests_shadow, tri_tests_shadow, sph_tests_shadow;
    double ms_total;         /* device time of the last crt_render* call (HIP events on the context's stream) */
    double ms_trace, ms_shade, ms_other;
    double ms_trace_nearest; /* summed duration of the nearest-hit traversal launches (dominant kernel) */
    double ms_trace_shadow;
    int32_t launches_trace_nearest, launches_trace_shadow;
    int32_t nodes, prims, depth, leaves;
    double ms_build_host, ms_refit;
    uint64_t scene_bytes;    /* packed scene + BVH bytes resident in HBM */
    uint64_t node_visits, node_visits_shadow;   /* counting only: BVH nodes visited (one visit tests all its children) */
    double ms_build_device;  /* wall time of the last CRT_BVH_WIDE_DEVICE build */
    /* the wide path traces the shadow rays of a bounce and the nearest-hit rays of the next in ONE launch: those
     * launches are counted here, the pure ones (primary rays; last bounce's shadow rays) in the fields above */
    double ms_trace_mixed;
    int32_t launches_trace_mixed, pad_;
    /* counting only, wide path: what the persistent waves of k_trace8 did, summed over the frame's launches.  An
     * iteration of a wave is either a node step (every lane that wants one tests the 8 children of its node) or a
     * primitive step (one primitive per lane); bench.py prices them with tools/isa_mix.py's per-region instruction counts */
    uint64_t sched_iterations;          /* loop iterations of all waves */
    uint64_t sched_node_iterations, sched_node_lanes;    /* node steps and the lanes they served */
    uint64_t sched_prim_iterations, sched_prim_lanes;
    uint64_t sched_active_lanes;        /* lanes holding a ray, summed over iterations */
    uint64_t sched_refill_iterations, sched_finish_iterations, sched_setup_iterations;
    uint64_t sched_pop_iterations, sched_decode_iterations, sched_hit_iterations;   /* node steps with a stack pop; primitive steps with a leaf decode / with a hit */
    uint64_t visits_no_child_hit;       /* node visits whose 8 slab tests all failed */
    uint64_t visits_no_child_hit_tlimit; /* ... of which some child would have passed without the ray's current t limit (entries stacked before a nearer hit was found) */
    uint64_t boxes_rejected_by_tlimit;
    uint64_t trace_waves;               /* waves launched by the traversal launches (persistent grid x launches) */
} crt_stats;

typedef struct crt_ctx crt_ctx;

/* ---- context ---- */
crt_ctx* crt_create(int device, int* err);
void crt_destroy(crt_ctx*);
const char* crt_last_error(crt_ctx*);
void crt_params_default(crt_params*);
int crt_set_counting(crt_ctx*, int on);   /* per-ray test counters for the algorithmic-bytes figure (slower) */

/* ---- scene: replaces the app-side std::vector<SceneObject*> / std::vector<Light> (main.cu:230-321,
 *      464-488) that startRayTracing receives (Ray.cu:665).  Coordinates must be finite and sphere radii >= 0 (CRT_ERR_ARG
 *      otherwise: with a negative radius the reference's boxes are inverted and what a ray then hits depends on the
 *      composition of that one tree's leaves); every other value is taken as the reference takes it, NaN and infinity in
 *      materials and lights included (crt_params.zero_paths) ---- */
int crt_scene_set(crt_ctx*, const crt_prim* prims, int n);
int crt_scene_update(crt_ctx*, const crt_prim* prims, int n);   /* moved geometry; same count and order */
int crt_lights_set(crt_ctx*, const crt_light* lights, int n);
/* the animation step of main.cu:545-573 (rotate triangles 0.01 rad about Y, bounce and move spheres)
 * applied to the device copy of the scene and its packed geometry: bit-identical to crt_animate_host +
 * crt_scene_update without the host loop and the re-upload.  Follow with crt_bvh_refit (or crt_bvh_build,
 * which first copies the moved scene back for the host builders). */
int crt_animate_device(crt_ctx*);
int crt_scene_get(crt_ctx*, crt_prim* prims_out, int n);        /* current scene state (after device animation) */

/* ---- BVH: replaces newBuild+constructTree (main.cu:33-136, bvh.cu:12-273), refitTree
 *      (bvh.cu:288-400) and freeTree (bvh.cu:274-286) ---- */
#define CRT_BVH_WIDE 0       /* binned-SAH tree collapsed to compressed 8-wide nodes, <=4 primitives per leaf.  Exact-t ties
                              * resolve by the reference's leaf order; on scenes where the reference's split does not
                              * terminate (below) the tree is still built and ties resolve by scene index */
#define CRT_BVH_REFERENCE 1  /* the reference's topology (mean-centroid split, <=64 per leaf), exact boxes.  CRT_ERR_BUILD where
                              * constructTree would recurse forever: more than 64 primitives sharing a centroid coordinate
                              * on the axis it picks (its box maxima never fall below 0, main.cu:35, so that axis can be one
                              * along which a flat group has no extent at all) */
#define CRT_BVH_WIDE_DEVICE 2 /* the same 8-wide format built on the device (Morton-order linear BVH, one primitive per leaf):
                              * milliseconds instead of a second for 1M primitives, for "rebuild every N frames" on large
                              * animated scenes.  Builds from the device copy of the scene (after crt_animate_device no
                              * copy back).  Exact-t ties resolve by the reference's leaf order here too: refrank.hip replays
                              * newBuild + constructTree's partition on the device (scene index only where that recursion
                              * does not terminate, as for CRT_BVH_WIDE) */
int crt_bvh_build(crt_ctx*, int mode);
/* "Rebuild every N frames" (main.cu:534-541) without the stall of a synchronous rebuild; no counterpart in the reference,
 * whose loop stops for its host build.  crt_bvh_build_async starts a CRT_BVH_WIDE_DEVICE build on a stream and a host
 * thread of its own, from a snapshot of the scene as the context's stream leaves it (take it right after the frame's
 * crt_animate_device), into a spare blob, and returns; frames keep rendering on the current tree (keep refitting it).
 * crt_bvh_swap(ctx, wait, &swapped): once the build is done (wait != 0: wait for it) the new tree becomes current -- its
 * geometry packed from the scene as it is now, its boxes refitted to it, so the call stands in for that frame's
 * crt_bvh_refit -- and *swapped = 1.  Tie-break ranks are those of the snapshot, as a synchronous rebuild at the snapshot's
 * frame would have produced; frames rendered between the snapshot and the swap still use the previous build's ranks (only
 * exact-t ties between duplicated primitives can tell).  One build at a time; crt_bvh_build / crt_scene_set / crt_bvh_free /
 * crt_blob_import_device abandon a build under way. */
int crt_bvh_build_async(crt_ctx*);
int crt_bvh_swap(crt_ctx*, int wait, int* swapped);
#define CRT_REFIT_TIGHT 0
#define CRT_REFIT_GROW 1     /* boxes only grow, as refitTree does */
int crt_bvh_refit(crt_ctx*, int grow_only);
int crt_bvh_free(crt_ctx*);
/* Rank of every primitive in the reference's leaf order (DFS, left first): ties in t resolve to
 * the lower rank, as the strict '<' at Ray.cu:199,216 does.  out[n] indexed by scene index. */
int crt_bvh_reference_order(crt_ctx*, int32_t* rank_out, int n);
/* Node boxes after build/refit, for parity checks of the refit kernel: 6 floats per stored child
 * box in node order (minX,maxX,minY,maxY,minZ,maxZ), plus the root box first.  Every node is
 * dumped with CRT_BVH_DUMP_WIDTH child entries (the reference topology uses the first two).
 * The reference topology holds the reference's boxes (bvh.cu:70-129, 301-360).  In the wide trees a triangle's box is
 * exact and a sphere's is the cube around the ball that intersectSphere's float arithmetic can report a hit in (Ray.cu:110-113
 * cancels: a little more than the sphere, and more than that for spheres much smaller than their distance from the origin). */
#define CRT_BVH_DUMP_WIDTH 8
int crt_bvh_dump(crt_ctx*, int32_t* nnodes, float* root_box6, float* child_boxes /* nnodes*8*6 */,
                 uint32_t* child_links /* nnodes*8 */, int cap_nodes);

/* ---- render: replaces startRayTracing (Ray.hpp:35, Ray.cu:665-742).  bgr_out is the caller's
 *      width*height*3 buffer, BGR, row 0 = bottom scanline (Ray.cu:724-731). ---- */
int crt_render(crt_ctx*, const crt_camera*, const crt_params*, unsigned char* bgr_out_host);
/* same, leaving the frame in HBM: dev_bgr (width*height*3 device bytes) may be NULL to use the
 * context's own buffer; only this rank's tiles are written. */
int crt_render_device(crt_ctx*, const crt_camera*, const crt_params*, void* dev_bgr);
/* The same frame as two calls, for callers that keep several frames (or tile subsets of one frame: crt_params.tile_sub)
 * in flight.  No counterpart in the reference, whose startRayTracing synchronises after every kernel (Ray.cu:698,718,720).
 * A context has CRT_MAX_FRAME_SLOTS frame slots, each with a HIP stream and per-frame buffers of its own; slot 0 is the
 * one crt_render / crt_render_device use (they are submit + wait on slot 0).  crt_render_submit queues the frame's whole
 * launch sequence on the slot's stream and returns without waiting for the device (the first frame of a size on a slot
 * allocates); crt_render_wait returns when that frame is complete, reports its errors and fills crt_get_stats.  A persistent
 * traversal launch ends in a tail of its longest rays: with two or three slots in flight the launches of one frame run in
 * the tails of the other's.  Frames on different slots may read the scene concurrently; calls that change the scene
 * (crt_animate_device, crt_bvh_refit, crt_scene_update) are ordered behind frames already submitted and before frames
 * submitted later, on the device; crt_bvh_build / crt_blob_import_device / crt_scene_set wait for them on the host.
 * Counting (crt_set_counting) and crt_params.use_graph are for slot 0 only.  dev_bgr NULL = the slot's own frame. */
#define CRT_MAX_FRAME_SLOTS 4
int crt_render_submit(crt_ctx*, int slot, const crt_camera*, const crt_params*, void* dev_bgr);
int crt_render_wait(crt_ctx*, int slot);
/* float colour (RGB, before Ray.cu:724-731) and per-bounce nearest-hit ids/t of the last render,
 * for parity tests.  ids: scene index, -2-k for wall k, -1 none.  Enabled by crt_set_capture. */
int crt_set_capture(crt_ctx*, int on);
int crt_get_capture(crt_ctx*, int bounce, int32_t* ids, float* ts, int npixels);
int crt_get_color(crt_ctx*, float* rgb_f32, int npixels);

/* ---- nearest-hit parity hook (no counterpart in the reference, which never exposes ids or t) ---- */
int crt_trace(crt_ctx*, const float* org_dir /* 6n host floats */, int n, int32_t* prim_out, float* t_out);
/* shadow predicate of Ray.cu:650-656 for directional lights: out[i]=1 iff a hit exists and the
 * nearest hit has t > 1e-9 */
int crt_trace_shadow(crt_ctx*, const float* org_dir, int n, uint8_t* shadowed_out);
/* pow(x, y) as Shade's specular term evaluates it on the device (Ray.cu:655 `pow(dot(...), shininess)` on floats), n values,
 * host arrays.  Parity hook without a counterpart in the reference: the kernel restates the host libm's powf and the
 * tests hold it against the oracle's, value by value. */
int crt_powf_probe(crt_ctx*, const float* x, const float* y, int n, float* out);
/* crt_trace for callers that cannot allocate.  No counterpart in the reference (like crt_trace itself); it exists for
 * tight serving loops, harnesses that forbid hipMalloc and code that must not disturb an allocator (e.g. one preparing a
 * graph capture elsewhere): crt_trace allocates and frees its working buffers on every call, this entry point calls no
 * hipMalloc / hipFree / event or other allocating routine.  It does copy the status word to the host and synchronise the
 * context's stream, so it cannot itself run INSIDE a stream capture.  Same kernels, same results; rays and results are DEVICE pointers (6n floats in, n ids and n t out).  The
 * working buffers (packed ray queue, hit slots, fetch counters) live in one fixed device array per GPU of
 * 36 MiB + 4 KiB, held until the process ends, which bounds n: CRT_ERR_ARG beyond CRT_NOALLOC_MAX_RAYS.  All calls on
 * one GPU share that array: calls that overlap on one GPU (two contexts, two threads) race -- serialise them.
 * Returns with the results complete (the context's stream is synchronised). */
#define CRT_NOALLOC_MAX_RAYS (1 << 20)
int crt_trace_noalloc(crt_ctx*, const float* dev_org_dir, int n, int32_t* dev_prim_out, float* dev_t_out);

/* ---- multi-GPU plumbing: the packed scene + BVH is one device blob that the host layer
 *      broadcasts (RCCL) and imports on the other ranks ---- */
int crt_blob_size(crt_ctx*, size_t* bytes);
int crt_blob_device_ptr(crt_ctx*, void** dev_ptr);
/* copies (D2D).  A context that already holds the SAME scene (crt_scene_set with the primitives the blob was built
 * from, e.g. every rank of an animated multi-GPU run) stays a full context afterwards: crt_animate_device,
 * crt_scene_update, crt_bvh_refit and crt_bvh_build work on it.  Without a scene the import is render-only. */
int crt_blob_import_device(crt_ctx*, const void* dev_ptr, size_t bytes);
int crt_tile_owner(int tile_index, int nranks);
/* number of tiles this rank owns and their pixel count; tile list in owned order */
int crt_tiles_owned(const crt_params*, int32_t* ntiles, int64_t* npixels, int32_t* tile_ids, int cap);
/* pack this rank's tiles of dev_bgr tile-major into dev_packed (npixels*3 bytes) and back.  Pack returns
 * with the copy finished (a collective follows); unpacks queue on the context's stream: crt_sync() ends them. */
int crt_tiles_pack(crt_ctx*, const crt_params*, const void* dev_bgr, void* dev_packed);
int crt_tiles_unpack(crt_ctx*, const crt_params*, int src_rank, const void* dev_packed, void* dev_bgr);
/* the pack queued on a frame slot's stream behind the frame submitted there, without waiting: crt_render_wait(slot)
 * then returns with the packed tiles complete.  dev_bgr NULL = the slot's own frame */
int crt_tiles_pack_slot(crt_ctx*, int slot, const crt_params*, const void* dev_bgr, void* dev_packed);
/* the HIP stream of a frame slot (a hipStream_t), for host layers that queue their own copies / collectives behind a frame */
int crt_slot_stream(crt_ctx*, int slot, void** hip_stream);

int crt_get_stats(crt_ctx*, crt_stats*);
int crt_sync(crt_ctx*);

/* ---- the same exchange done for you: one process, every GPU of the node (no counterpart in the
 *      reference, which is single-GPU: startRayTracing in compat/ uses this layer when more than one GPU is visible).
 *      One crt_ctx per device; rank 0's context owns the scene (crt_scene_set / crt_bvh_build / crt_bvh_refit /
 *      crt_animate_device on crt_mgpu_ctx(m, 0)), crt_mgpu_sync_scene replicates it with ONE ncclBroadcast of the
 *      packed blob, crt_mgpu_render queues every rank's tiles (crt_render_submit: one host thread drives all GPUs) and gathers
 *      them to rank 0 with one group of ncclSend / ncclRecv (R - 1 point-to-point transfers, each over its own xGMI link).
 *      No reduction.  With CRT_MGPU_RCCL_ALWAYS on ONE GPU rank 0 sends its own tiles to itself through the same grouped
 *      ncclSend / ncclRecv (a rehearsal of the calls where there is no second GPU).
 *      devices = NULL: all visible GPUs.  A list with repeats (e.g. {0,0,0}) makes logical ranks on one GPU for
 *      rehearsal: RCCL refuses duplicate devices, so the two exchanges become device-to-device copies. ---- */
typedef struct crt_mgpu crt_mgpu;
#define CRT_MGPU_RCCL_ALWAYS 1   /* create the communicator even for a single GPU (default: one GPU never loads RCCL) */
crt_mgpu* crt_mgpu_create(const int* devices, int ndev, int flags, int* err);
void crt_mgpu_destroy(crt_mgpu*);
int crt_mgpu_size(const crt_mgpu*);
int crt_mgpu_uses_rccl(const crt_mgpu*);                       /* 1: RCCL communicator; 0: local device-to-device copies */
crt_ctx* crt_mgpu_ctx(crt_mgpu*, int rank);
const char* crt_mgpu_last_error(crt_mgpu*);
int crt_mgpu_lights_set(crt_mgpu*, const crt_light* lights, int n);
int crt_mgpu_sync_scene(crt_mgpu*);                            /* after every build on rank 0 (and after refit / animation steps done on rank 0 ONLY) */
/* The scene on every rank instead of rank 0 only (R uploads from the one host array, no collective) -- what crt_mgpu_animate
 * needs.  Then crt_bvh_build on crt_mgpu_ctx(m, 0) + crt_mgpu_sync_scene as usual. */
int crt_mgpu_scene_set(crt_mgpu*, const crt_prim* prims, int n);
/* The per-frame update of main.cu:534-541 + 545-573 with no traffic between GPUs (SURVEY.md 8(e) collective (2), the
 * "replicate the deterministic animation step" form): every rank runs crt_animate_device on its own copy of the scene,
 * then crt_bvh_refit(grow_only) -- queued on every GPU before any is waited for -- or, rebuild != 0, crt_bvh_build(
 * CRT_BVH_WIDE_DEVICE) on its own device (one host thread per GPU).  No crt_mgpu_sync_scene afterwards. */
int crt_mgpu_animate(crt_mgpu*, int rebuild, int grow_only);
int crt_mgpu_render(crt_mgpu*, const crt_camera*, const crt_params*, unsigned char* bgr_out_host);
/* leaves the assembled frame on rank 0's device (its address in *dev_frame_rank0, valid until the next render) */
int crt_mgpu_render_device(crt_mgpu*, const crt_camera*, const crt_params*, void** dev_frame_rank0);
/* Throughput mode of the same frame: crt_mgpu_submit queues the frame on every rank (render + tile pack on one of three
 * frame slots per context) and returns without waiting; once three frames are in flight it first collects the oldest and
 * starts its gather, and completes gathers more than three frames old -- the launches of one frame run in the tails of the
 * others', every gather under later renders.  *dev_frame_done = rank 0's assembled frame of the most recent frame this call
 * completed (frames complete in submission order; valid until the next crt_mgpu_* call), or NULL.  crt_mgpu_finish
 * completes what is still in flight and returns the last frame.  Scene changes between submissions (crt_mgpu_animate) are
 * ordered on the device behind the frames already submitted. */
int crt_mgpu_submit(crt_mgpu*, const crt_camera*, const crt_params*, void** dev_frame_done);
int crt_mgpu_finish(crt_mgpu*, void** dev_frame_last);
/* wall milliseconds of the last crt_mgpu_render_device: all ranks' render + pack, the gather, the unpack on rank 0 */
int crt_mgpu_get_times(crt_mgpu*, double* ms_render, double* ms_exchange, double* ms_unpack);

/* ---- host-side scene layer (replaces objloader.cpp:20-112, the scene literals of
 *      main.cu:230-321 and the animation step main.cu:545-573) ---- */
int crt_load_obj(const char* path, float* verts3, float* norms3, int max_corners);   /* returns corner count */
int crt_scene_teapot(const char* obj_path, int with_sphere, crt_prim* out, int cap);  /* returns prim count */
int crt_scene_spheres(int n, uint32_t seed, crt_prim* out);
int crt_scene_soup(int ntris, uint32_t seed, crt_prim* out);
/* The unit-disk sample table of the area lights: n <= 1024 entries of (u, cos(theta), sin(theta)) with u, theta/2pi
 * from a 32-bit LCG (Ray.cu:569-573 draws them from curand).  Sample k of a shading point is entry
 * (hash(pixel, light, bounce, sub-sample) + k) mod 1024, placed at position + u*radius*(cos, 0, sin). */
int crt_area_table(float* out3, int n);
int crt_lights_default(crt_light* out2);                                              /* l1, l3 of main.cu:465-488 */
void crt_camera_default(crt_camera*);                                                  /* main.cu:459,508 */
int crt_animate_host(crt_prim* prims, int n);                                          /* main.cu:545-573 */
int crt_write_bmp(const char* path, const unsigned char* bgr, int width, int height); /* headless output */

/* Host-only view of the two BVH builders (no device needed): what crt_bvh_build packs and uploads.
 * Two-call pattern: pass NULL arrays to get *nnodes, then arrays with cap_nodes >= *nnodes.
 * child_boxes: nnodes*8*6 floats (minX,maxX,minY,maxY,minZ,maxZ per child); child_links: nnodes*8;
 * parents: nnodes*2 (parent node, slot in parent); slot_to_scene, rank_of_scene: n each. */
int crt_host_bvh(const crt_prim* prims, int n, int mode, int32_t* nnodes, int32_t* depth, float* root_box6,
                 float* child_boxes, uint32_t* child_links, uint32_t* parents, uint32_t* slot_to_scene,
                 uint32_t* rank_of_scene, int cap_nodes);

#ifdef __cplusplus
}
#endif
#endif
