/* crt_api.cpp -- implementation of the C ABI in include/crt.h on top of the HIP kernels.
 *
 * Frame flow (replaces startRayTracing, Ray.cu:665-742), all on one HIP stream with queue sizes
 * kept on the device -- no host synchronisation between kernels, where the reference calls
 * cudaDeviceSynchronize after every launch and copies the Isect array to the host every bounce:
 *
 *   gen_primary -> [ trace(nearest) -> shade -> trace(shadow) -> resolve ] x bounces -> convert
 *
 * Paths whose reflection coefficient is zero (after any wall hit, Ray.cu:257,262) are not traced
 * further; they contribute exactly zero in the reference, so the image is unchanged.
 */
#include "../../include/crt.h"
#include "kernels.h"
#include "internal.h"
#include "bvh_build.h"
#include "vec.h"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <map>
#include <array>
#include <chrono>
#include <cstring>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <atomic>

using namespace crt;

namespace {
constexpr int kMaxBounces = 16;
/* per sample pass: [0..15] queue sizes per bounce, then 32 launches x 8 region fetch counters kFetchStride words apart,
 * then (kCtrShadow0) the shadow-queue sizes per bounce */
constexpr int kCtrFetch0 = 64;
/* the shadow-queue sizes sit 4 KB and more from the queue sizes: k_shade adds to one of each per block iteration, and two hot
 * words in one 128-byte line share an L2 channel's atomic unit */
constexpr int kCtrShadow0 = kCtrFetch0 + 32 * 8 * kFetchStride + 1024;
constexpr int kCtrStride = kCtrShadow0 + 1024;
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}

/* Everything one frame in flight needs besides the scene: see crt_ctx::fs */
struct FrameSlot {
    hipStream_t stream = nullptr; bool own_stream = false;
    bool in_flight = false;          /* submitted, not yet collected by crt_render_wait */
    hipEvent_t ev_done = nullptr;    /* recorded behind the last launch of every submitted frame (no timing) */
    hipEvent_t ev_scene = nullptr;   /* slots > 0: the state of the context's stream a frame has to wait for */
    size_t cap = 0, npix_alloc = 0, nlights_alloc = 0;
    float4 *rayO[2] = {nullptr, nullptr}, *rayD[2] = {nullptr, nullptr}, *coef[2] = {nullptr, nullptr};
    uint32_t* pix[2] = {nullptr, nullptr};
    int32_t* hitSlot = nullptr; float* hitT = nullptr;
    float4 *contrib = nullptr, *shO = nullptr, *shD = nullptr, *accum = nullptr, *sum = nullptr;
    uint8_t* shflag = nullptr;
    /* crt_params.use_graph: the frame's launch sequence captured once and replayed while nothing but the scene's
     * contents changes */
    struct GraphKey {
        crt_camera cam; crt_params p; LightsParam lights;
        const void* ptrs[20]; uint32_t nnodes, nprims, nmats, depth, mode; int counting, capture, grid, ovf_depth, block, keep_zero; float scene_abs, origin_bound;
    };
    GraphKey graph_key{}; hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr; bool in_capture = false;
    uint8_t* d_bgr = nullptr; size_t bgr_cap = 0;
    int32_t* cap_ids = nullptr; float* cap_ts = nullptr; size_t cap_planes = 0; size_t cap_npix = 0;
    uint32_t* d_tiles = nullptr; size_t tiles_cap = 0; std::vector<uint32_t> tiles; crt_params tiles_for{}; int64_t owned_pixels = 0;
    uint32_t* d_ctr = nullptr; size_t ctr_cap = 0;
    uint32_t* h_ctr = nullptr; size_t h_ctr_cap = 0;             /* pinned: queue sizes of the frame, copied behind its last launch */
    uint32_t* h_status = nullptr;                                /* pinned: the device status word as the frame left it */
    uint32_t *d_sortkeys = nullptr, *d_sortperm = nullptr, *d_sorthist = nullptr; size_t sort_cap = 0;     /* shadow queue */
    uint32_t *d_sortkeys2 = nullptr, *d_sortperm2 = nullptr;                                                 /* reflection queue */
    bool ev_error = false;             /* a timing event could not be created or recorded since the last submit began */
    std::vector<hipEvent_t> evpool; size_t evused = 0;
    std::vector<std::pair<int, int>> evlog;      /* (kind, event index) */
    /* what crt_render_wait needs to turn the frame's counters into crt_stats */
    crt_params sub_p{}; int sub_spp = 1; bool sub_counting = false; int sub_grid = 0;
    int last_w = 0, last_h = 0, last_bounces = 0;
    bool light_timing = false;       /* crt_params.timing == 1 */
};

/* one device build (device_build_core): where it runs, what it reads, where the tree goes */
struct DevBuild {
    hipStream_t s; const crt_prim* prims; bool pack;      /* in: stream, scene to build from, whether to pack geometry too */
    float origin_bound;                                   /* in: what the sphere boxes are padded for (taken when the build is described: the
                                                           * asynchronous build runs on a thread of its own while the scene may be updated) */
    void* blob; size_t blob_cap;                          /* in/out: target blob (reallocated when too small) */
    BlobHeader hdr; bool rank_valid; uint32_t levels; double ms;      /* out */
};

struct crt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool counting = false, capture = false;

    std::vector<crt_prim> prims;
    std::vector<uint32_t> matid;
    std::vector<Material> mats;
    LightsParam lights{};
    RefTree ref;
    HostBVH bvh;
    bool built = false, have_ref = false, imported = false;
    bool host_stale = false;     /* crt_animate_device moved the device copy of the scene past c->prims */
    /* can a shading term of this scene / these lights be non-finite?  Decides whether paths with a zero coefficient may end
     * (crt_params.zero_paths = 0); the scene's half travels in BlobHeader.flags */
    bool nonfinite_scene = false, nonfinite_lights = false;

    /* packed scene */
    void* d_blob = nullptr; size_t blob_bytes = 0; size_t blob_cap = 0;
    BlobHeader hdr{};
    crt_prim* d_prims = nullptr; uint32_t* d_matid = nullptr; size_t prim_cap = 0;
    uint32_t* d_arrive = nullptr; size_t arrive_cap = 0;
    float scene_abs = 0.f;
    float origin_bound_scene = 0.f;      /* origin_bound_of(prims) as of crt_scene_set / crt_scene_update: what the next device build pads sphere boxes for */

    /* frames: every slot owns a stream and a full set of per-frame buffers, so that up to CRT_MAX_FRAME_SLOTS frames
     * (or tile subsets of one frame) are in flight at once on the read-only scene blob.  Slot 0 runs on the context's own
     * stream -- the synchronous entry points use it -- the others on streams of their own */
    FrameSlot fs[CRT_MAX_FRAME_SLOTS];
    float4* d_area_table = nullptr;      /* unit-disk samples of the area lights, uploaded on first use */
    unsigned char* h_stage = nullptr; size_t stage_cap = 0;     /* pinned staging for crt_render's copy to the caller's buffer */
    void* d_lbvh = nullptr; size_t lbvh_cap = 0;                 /* scratch of the device builder */
    Node8* d_lbvh_nodes = nullptr; size_t lbvh_nodes_cap = 0;    /* its output before the blob is laid out */
    uint32_t* d_lbvh_s2s = nullptr; size_t lbvh_s2s_cap = 0;
    void* d_refrank = nullptr; size_t refrank_cap = 0;          /* scratch of the device-side reference-order pass */
    uint32_t* d_refrank_out = nullptr; size_t refrank_out_cap = 0; bool dev_rank_valid = false; uint32_t refrank_levels = 0;
    uint32_t* d_status = nullptr;
    uint32_t* h_status = nullptr;      /* pinned: crt_trace_noalloc reads the status word without a staging copy */
    void* noalloc_scratch = nullptr;   /* this device's copy of noalloc.hip's scratch array (address looked up once per device) */
    unsigned long long* d_testctr = nullptr;
    uint32_t* d_ovf = nullptr; size_t ovf_cap = 0; size_t ovf_slot_words = 0; int ovf_depth = 0;      /* one area per frame slot */
    int grid_blocks = 1024;            /* persistent traversal grid, in 256-thread blocks: what is resident at once */
    int trace_block = 256;             /* threads per block of k_trace8 (CRT_TRACE_BLOCK = 256 / 128 / 64) */
    /* crt_bvh_build_async: a device build on a stream and a host thread of its own, from a snapshot of the scene, into the
     * spare blob; crt_bvh_swap makes it current */
    std::thread build_thread; std::atomic<int> build_state{0};      /* 0 none, 1 running, 2 ready, 3 failed */
    hipStream_t build_stream = nullptr; hipEvent_t ev_snap = nullptr, ev_build = nullptr;
    void* d_blob2 = nullptr; size_t blob2_cap = 0;
    crt_prim* d_prims_snap = nullptr; size_t snap_cap = 0;
    DevBuild async_build{}; int async_rc = 0; std::string async_err;
    hipEvent_t refit_ev[2] = {nullptr, nullptr}; bool refit_pending = false;      /* ctx_refit_enqueue .. ctx_refit_finish */
    std::vector<uint32_t*> tl_ids; std::vector<int> tl_n; crt_params tl_key{};      /* per-rank tile lists for pack/unpack */
    crt_stats stats{};
};

/* an asynchronous rebuild that is running or waiting to be swapped in is abandoned (its thread joined) */
static void cancel_async_build(crt_ctx* c) {
    if (c->build_thread.joinable()) c->build_thread.join();
    if (c->build_state.load() != 0) { c->d_blob2 = c->async_build.blob; c->blob2_cap = c->async_build.blob_cap; }      /* the build may have reallocated the spare blob */
    c->build_state = 0;
}

namespace {

#define HIPCHECK(ctx, call)                                                                         \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return CRT_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

int fail(crt_ctx* c, int code, const char* msg) { if (c) c->err = msg; return code; }

template <class T>
int ensure(crt_ctx* c, T*& p, size_t& cap, size_t need) {
    if (need <= cap && p) return CRT_OK;
    if (p) { HIPCHECK(c, hipFree(p)); p = nullptr; cap = 0; }
    size_t want = need + need / 8 + 64;
    HIPCHECK(c, hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return CRT_OK;
}
template <class T>
int realloc_exact(crt_ctx* c, T*& p, size_t count) {
    if (p) { HIPCHECK(c, hipFree(p)); p = nullptr; }
    if (count) HIPCHECK(c, hipMalloc((void**)&p, count * sizeof(T)));
    return CRT_OK;
}

SceneView view_of(const crt_ctx* c) {
    SceneView v{};
    const char* b = (const char*)c->d_blob;
    v.nodes = (const Node4*)(b + c->hdr.off_nodes);
    v.qnodes = (const QNode8*)(b + c->hdr.off_qnodes);
    v.geom = (const Geom*)(b + c->hdr.off_geom);
    v.shade = (const Shade*)(b + c->hdr.off_shade);
    v.rank = (const uint32_t*)(b + c->hdr.off_rank);
    v.sidx = (const uint32_t*)(b + c->hdr.off_sidx);
    v.mats = (const Material*)(b + c->hdr.off_mats);
    v.nprims = c->hdr.nprims; v.nnodes = c->hdr.nnodes; v.nmats = c->hdr.nmats;
    return v;
}

/* crt_params.zero_paths: a path whose coefficient is (0,0,0) may end iff nothing it could still add -- coefficient * colour,
 * Ray.cu:660 -- can be non-finite.  Sufficient conditions on the scene, each one a way the reference's Shade (Ray.cu:546-663)
 * or its normals (Ray.cu:135,168-172) produce NaN or infinity from finite rays:
 *   - a material value that is not finite or beyond 1e9 in magnitude (sums of products may overflow; light colours likewise); a shininess that is negative (pow(0, s) = inf), fractional (pow of a negative
 *     base = NaN, Ray.cu:655) or beyond 1e6 (|base| may exceed 1 by a rounding);
 *   - a sphere of radius <= 0 ((X - c) / r, Ray.cu:135);
 *   - a triangle whose vertex normals are not finite or do not all lie strictly on one side of its plane: their
 *     barycentric mix can then be the zero vector, and normalize(0) is NaN (Ray.cu:172).  Zero-area triangles are never hit. */
static bool finite3(const float* v) { return v[0] * 0.f + v[1] * 0.f + v[2] * 0.f == 0.f; }
/* finite and small enough that sums and products of a few of them cannot overflow: a colour is at most
 * 8 lights x 1.5 x lcolor x (.6 diffuse + .2 specular (1 + 1e-6)^shininess) */
static bool tame3(const float* v) { return fabsf(v[0]) <= 1e9f && fabsf(v[1]) <= 1e9f && fabsf(v[2]) <= 1e9f; }
bool shading_may_be_nonfinite(const std::vector<crt_prim>& prims) {
    for (const crt_prim& p : prims) {
        if (!tame3(p.ambient) || !tame3(p.diffuse) || !tame3(p.specular) || !finite3(p.reflective)) return true;
        const float s = p.shininess;
        if (!(s >= 0.f && s <= 1e6f) || s != floorf(s)) return true;
        if (p.type == 0) { if (!(p.radius > 0.f)) return true; continue; }
        const vec3 v1(p.v1), v2(p.v2), v3(p.v3);
        const vec3 g = cross(v2 - v1, v3 - v1);
        if (!(dot(g, g) > 0.f)) continue;
        if (!finite3(p.v1Norm) || !finite3(p.v2Norm) || !finite3(p.v3Norm)) return true;
        const float a = dot(vec3(p.v1Norm), g), b = dot(vec3(p.v2Norm), g), d = dot(vec3(p.v3Norm), g);
        if (!((a > 0.f && b > 0.f && d > 0.f) || (a < 0.f && b < 0.f && d < 0.f))) return true;
    }
    return false;
}
bool lights_may_be_nonfinite(const LightsParam& L) {
    for (int i = 0; i < L.n; i++) {
        const crt_light& l = L.l[i];
        if (!tame3(l.color)) return true;
        if (l.area || l.point) {
            if (!finite3(l.position)) return true;
            if (l.area) { if (!(l.radius * 0.f == 0.f)) return true; }
            else if (!(l.constantTerm * 0.f + l.linearTerm * 0.f + l.quadraticTerm * 0.f == 0.f)) return true;
        } else {
            if (!finite3(l.direction)) return true;
            const float n2 = l.direction[0] * l.direction[0] + l.direction[1] * l.direction[1] + l.direction[2] * l.direction[2];
            if (!(n2 > 0.f) || !(n2 * 0.f == 0.f)) return true;          /* normalize(direction), Ray.cu:635-636 */
        }
    }
    return false;
}

void dedupe_materials(crt_ctx* c) {
    std::map<std::array<uint32_t, 13>, uint32_t> seen;
    c->mats.clear(); c->matid.resize(c->prims.size());
    for (size_t i = 0; i < c->prims.size(); i++) {
        const crt_prim& p = c->prims[i];
        std::array<uint32_t, 13> key;
        memcpy(&key[0], p.ambient, 12); memcpy(&key[3], p.diffuse, 12); memcpy(&key[6], p.specular, 12);
        memcpy(&key[9], p.reflective, 12); memcpy(&key[12], &p.shininess, 4);
        auto it = seen.find(key);
        if (it == seen.end()) {
            Material m{};
            memcpy(m.ambient, p.ambient, 12); memcpy(m.diffuse, p.diffuse, 12); memcpy(m.specular, p.specular, 12);
            memcpy(m.reflective, p.reflective, 12); m.shininess = p.shininess;
            it = seen.emplace(key, (uint32_t)c->mats.size()).first;
            c->mats.push_back(m);
        }
        c->matid[i] = it->second;
    }
}

int upload_prims(crt_ctx* c) {
    const size_t n = c->prims.size();
    if (n > c->prim_cap || !c->d_prims) {
        if (c->d_prims) HIPCHECK(c, hipFree(c->d_prims));
        if (c->d_matid) HIPCHECK(c, hipFree(c->d_matid));
        c->d_prims = nullptr; c->d_matid = nullptr;
        HIPCHECK(c, hipMalloc((void**)&c->d_prims, (n + 1) * sizeof(crt_prim)));
        HIPCHECK(c, hipMalloc((void**)&c->d_matid, (n + 1) * sizeof(uint32_t)));
        c->prim_cap = n;
    }
    HIPCHECK(c, hipMemcpyAsync(c->d_prims, c->prims.data(), n * sizeof(crt_prim), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->d_matid, c->matid.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    return CRT_OK;
}

void layout_blob(BlobHeader& h) {
    size_t off = align_up(sizeof(BlobHeader), 256);
    const bool wide = h.mode == CRT_BVH_WIDE;
    h.off_nodes = off; off = align_up(off + (size_t)h.nnodes * (wide ? sizeof(Node8) : sizeof(Node4)), 256);
    h.off_qnodes = off; off = align_up(off + (wide ? (size_t)h.nnodes * sizeof(QNode8) : 0), 256);
    h.off_geom = off; off = align_up(off + (size_t)h.nprims * sizeof(Geom), 256);
    h.off_shade = off; off = align_up(off + (size_t)h.nprims * sizeof(Shade), 256);
    h.off_rank = off; off = align_up(off + (size_t)h.nprims * 4, 256);
    h.off_sidx = off; off = align_up(off + (size_t)h.nprims * 4, 256);
    h.off_slot_of = off; off = align_up(off + (size_t)h.nprims * 4, 256);
    h.off_mats = off; off = align_up(off + (size_t)h.nmats * sizeof(Material), 256);
    h.total_bytes = off;
}

int size_overflow(crt_ctx* c) {
    /* worst-case stack need: one entry per level, for the 8-wide tree (one 64-bit group entry per
     * visited node) as for the 2-wide reference topology (one 32-bit link).  Every frame slot gets an area of its own
     * (their launches overlap); the 8-wide kernel's in-LDS depth depends on its block size */
    const bool wide = c->hdr.mode == CRT_BVH_WIDE;
    const int bound = (int)c->hdr.depth + 2;
    const int in_lds = wide ? stack_depth8_of(c->trace_block) : kStackDepth;
    c->ovf_depth = bound > in_lds ? bound - in_lds : 0;
    const size_t per_slot = (size_t)c->grid_blocks * kTraceBlock * (size_t)c->ovf_depth * (wide ? 2 : 1);
    const size_t need = per_slot * CRT_MAX_FRAME_SLOTS;
    if (need > c->ovf_cap || !c->d_ovf) {
        if (c->d_ovf) HIPCHECK(c, hipFree(c->d_ovf));
        c->d_ovf = nullptr;
        HIPCHECK(c, hipMalloc((void**)&c->d_ovf, (need + 64) * sizeof(uint32_t)));
        c->ovf_cap = need;
    }
    c->ovf_slot_words = per_slot;
    return CRT_OK;
}

void scene_abs_from_box(crt_ctx* c) {
    float m = 0.f;
    for (int k = 0; k < 6; k++) { float a = fabsf(c->hdr.root_box[k]); if (a < 1e30f && a > m) m = a; }
    c->scene_abs = m;
}

/* events come from a pool (one per frame slot) that only grows; a failure to create or record one is remembered
 * (ev_error) and reported by the call that was timing with it */
bool reserve_events(FrameSlot& f, size_t k) {          /* the next k calls of next_event() cannot fail afterwards */
    while (f.evpool.size() < f.evused + k) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); f.ev_error = true; return false; }
        f.evpool.push_back(e);
    }
    return true;
}
hipEvent_t next_event(FrameSlot& f) {
    if (!reserve_events(f, 1)) return nullptr;
    return f.evpool[f.evused++];
}
enum { EV_START = 0, EV_GEN, EV_TRACE_N, EV_SHADE, EV_TRACE_S, EV_RESOLVE, EV_CONVERT, EV_SORT, EV_TRACE_M };
void mark(FrameSlot& f, int kind) {
    if (f.in_capture) return;             /* per-kernel timing is not available for a replayed graph */
    if (f.light_timing && kind != EV_START && kind != EV_CONVERT) return;
    hipEvent_t e = next_event(f);
    if (!e) return;
    if (hipEventRecord(e, f.stream) != hipSuccess) { f.ev_error = true; f.evused--; return; }
    f.evlog.emplace_back(kind, (int)f.evused - 1);
}

/* a frame slot's stream, events and pinned words; slot 0 borrows the context's stream */
int ensure_slot(crt_ctx* c, int k) {
    FrameSlot& f = c->fs[k];
    if (!f.stream) {
        if (k == 0) f.stream = c->stream;
        else { HIPCHECK(c, hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking)); f.own_stream = true; }
    }
    if (!f.ev_done) HIPCHECK(c, hipEventCreateWithFlags(&f.ev_done, hipEventDisableTiming));
    if (k > 0 && !f.ev_scene) HIPCHECK(c, hipEventCreateWithFlags(&f.ev_scene, hipEventDisableTiming));
    if (!f.h_status) { HIPCHECK(c, hipHostMalloc((void**)&f.h_status, 64, hipHostMallocDefault)); *f.h_status = 0; }
    return CRT_OK;
}

/* Frames in flight on the other slots read the scene blob.  Before the context's stream rewrites it in place (animation,
 * refit, scene update) that stream waits for them -- device-side, the host does not block */
int order_after_frames(crt_ctx* c) {
    for (int k = 1; k < CRT_MAX_FRAME_SLOTS; k++)
        if (c->fs[k].in_flight) HIPCHECK(c, hipStreamWaitEvent(c->stream, c->fs[k].ev_done, 0));
    return CRT_OK;
}
/* ... and before anything is freed or reallocated (build, import, free) the host waits for them */
int drain_frames(crt_ctx* c) {
    for (int k = 0; k < CRT_MAX_FRAME_SLOTS; k++)
        if (c->fs[k].in_flight && c->fs[k].stream) HIPCHECK(c, hipStreamSynchronize(c->fs[k].stream));
    return CRT_OK;
}

int ensure_tiles(crt_ctx* c, FrameSlot& f, const crt_params* p) {
    if (f.d_tiles && memcmp(&f.tiles_for, p, sizeof(crt_params)) == 0) return CRT_OK;
    int nt = 0; int64_t np = 0;
    int rc = crt_tiles_owned(p, &nt, &np, nullptr, 0);
    if (rc) return fail(c, rc, "bad tile parameters");
    std::vector<int32_t> tmp(nt);
    crt_tiles_owned(p, &nt, &np, tmp.data(), nt);
    /* crt_params.tile_nsub > 1: this call renders every tile_nsub-th tile of the rank's list, starting at tile_sub --
     * a frame split over several slots (or calls) without touching the rank's tile map, which pack / unpack keep using */
    const int nsub = p->tile_nsub > 1 ? p->tile_nsub : 1, sub = p->tile_nsub > 1 ? p->tile_sub : 0;
    f.tiles.clear();
    np = 0;
    const int tx = (p->width + p->tile_w - 1) / p->tile_w;
    for (int i = sub; i < nt; i += nsub) {
        f.tiles.push_back((uint32_t)tmp[i]);
        const int x0 = (tmp[i] % tx) * p->tile_w, y0 = (tmp[i] / tx) * p->tile_h;
        np += (int64_t)(x0 + p->tile_w <= p->width ? p->tile_w : p->width - x0) * (y0 + p->tile_h <= p->height ? p->tile_h : p->height - y0);
    }
    rc = ensure(c, f.d_tiles, f.tiles_cap, f.tiles.size() + 1);
    if (rc) return rc;
    HIPCHECK(c, hipMemcpyAsync(f.d_tiles, f.tiles.data(), f.tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice, f.stream));
    HIPCHECK(c, hipStreamSynchronize(f.stream));
    f.tiles_for = *p; f.owned_pixels = np;
    return CRT_OK;
}

int ensure_frame(crt_ctx* c, FrameSlot& f, size_t cap, size_t npix, int bounces, int nlights, bool need_sum) {
    if (cap > f.cap || !f.hitSlot) {
        for (int k = 0; k < 2; k++) {
            int r;
            if ((r = realloc_exact(c, f.rayO[k], cap))) return r;
            if ((r = realloc_exact(c, f.rayD[k], cap))) return r;
            if ((r = realloc_exact(c, f.coef[k], cap))) return r;
            if ((r = realloc_exact(c, f.pix[k], cap))) return r;
        }
        int r;
        if ((r = realloc_exact(c, f.hitSlot, cap))) return r;
        if ((r = realloc_exact(c, f.hitT, cap))) return r;
        f.cap = cap; f.nlights_alloc = 0;
    }
    const size_t nl = nlights > 0 ? (size_t)nlights : 1;
    if (nl > f.nlights_alloc) {
        int r;
        if ((r = realloc_exact(c, f.contrib, f.cap * nl))) return r;
        if ((r = realloc_exact(c, f.shflag, f.cap * nl))) return r;
        if ((r = realloc_exact(c, f.shO, f.cap * nl))) return r;
        if ((r = realloc_exact(c, f.shD, f.cap * nl))) return r;
        f.nlights_alloc = nl;
    }
    if (npix > f.npix_alloc || !f.accum) {
        int r;
        if ((r = realloc_exact(c, f.accum, npix))) return r;
        if (f.sum) { HIPCHECK(c, hipFree(f.sum)); f.sum = nullptr; }
        f.npix_alloc = npix;
    }
    if (need_sum && !f.sum) { int r; if ((r = realloc_exact(c, f.sum, f.npix_alloc))) return r; }
    if (c->capture && ((size_t)bounces > f.cap_planes || npix > f.cap_npix || !f.cap_ids)) {
        int r;
        if ((r = realloc_exact(c, f.cap_ids, (size_t)bounces * npix))) return r;
        if ((r = realloc_exact(c, f.cap_ts, (size_t)bounces * npix))) return r;
        f.cap_planes = bounces; f.cap_npix = npix;
    }
    return CRT_OK;
}

/* the status word as a frame left it (copied into the slot's pinned word behind the frame's last launch) */
int status_error(crt_ctx* c, uint32_t st) {
    if (!st) return CRT_OK;
    (void)hipMemsetAsync(c->d_status, 0, 4, c->stream);
    if (st & 2u) return fail(c, CRT_ERR_BUILD, "wide tree does not fit the compressed node format");
    return fail(c, CRT_ERR_STATE, "traversal stack overflow: tree deeper than the sized overflow area");
}

int check_status(crt_ctx* c) {
    uint32_t st = 0;
    HIPCHECK(c, hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost));
    return status_error(c, st);
}

/* dumps always carry CRT_BVH_DUMP_WIDTH (8) child entries per node; slots a format does not have are empty */
template <class N>
void export_nodes(const std::vector<N>& nodes, float* child_boxes, uint32_t* child_links, uint32_t* parents) {
    for (size_t i = 0; i < nodes.size(); i++) {
        for (int k = 0; k < CRT_BVH_DUMP_WIDTH; k++) {
            float* o = child_boxes + (i * CRT_BVH_DUMP_WIDTH + k) * 6;
            if (k < N::W) {
                o[0] = nodes[i].lox[k]; o[1] = nodes[i].hix[k]; o[2] = nodes[i].loy[k];
                o[3] = nodes[i].hiy[k]; o[4] = nodes[i].loz[k]; o[5] = nodes[i].hiz[k];
                child_links[i * CRT_BVH_DUMP_WIDTH + k] = nodes[i].child[k];
            } else {
                o[0] = o[2] = o[4] = 3.402823466e+38f; o[1] = o[3] = o[5] = -3.402823466e+38f;
                child_links[i * CRT_BVH_DUMP_WIDTH + k] = kChildEmpty;
            }
        }
        if (parents) { parents[2 * i] = nodes[i].parent; parents[2 * i + 1] = nodes[i].parent_slot; }
    }
}

}  // namespace

namespace {
/* the builders and the refit assume finite boxes: a NaN or infinite coordinate is an argument error, not a scene */
/* coordinates must be finite and radii >= 0.  A negative radius is not an error in the reference: intersectSphere squares it
 * (Ray.cu:110) while the boxes of bvh.cu come out inverted (centre - r > centre + r), so whether a ray ever reaches the sphere
 * depends on what else shares its 64-primitive leaf -- a property of that one tree, which no other tree can reproduce.
 * Rejected here instead (CRT_ERR_ARG). */
bool finite_geometry(const crt_prim* prims, int n) {
    for (int i = 0; i < n; i++) {
        const crt_prim& p = prims[i];
        float acc = 0.f;          /* x * 0 is 0 for every finite x and NaN otherwise */
        if (p.type == 0) { acc = p.position[0] * 0.f + p.position[1] * 0.f + p.position[2] * 0.f + p.radius * 0.f; if (p.radius < 0.f) return false; }
        else for (int k = 0; k < 3; k++) acc += p.v1[k] * 0.f + p.v2[k] * 0.f + p.v3[k] * 0.f;
        if (!(acc == 0.f)) return false;
    }
    return true;
}
}  // namespace

namespace crt {
int ctx_fail(crt_ctx* c, int code, const char* msg) { return fail(c, code, msg); }
/* crt_bvh_refit in two halves, so that a host layer driving several contexts (mgpu.cpp) can queue the refit on every
 * GPU before it waits for any: the launches + the copies of the new root box and the status word into pinned words ... */
int ctx_refit_enqueue(crt_ctx* c, int grow_only) {
    if (!c) return CRT_ERR_ARG;
    if (!c->built) return fail(c, CRT_ERR_STATE, "crt_bvh_refit: no BVH (call crt_bvh_build)");
    HIPCHECK(c, hipSetDevice(c->device));
    char* b = (char*)c->d_blob;
    { const int rc = order_after_frames(c); if (rc) return rc; }
    FrameSlot& f0 = c->fs[0];
    if (!reserve_events(f0, 2)) return fail(c, CRT_ERR_HIP, "crt_bvh_refit: hipEventCreate failed");
    c->refit_ev[0] = next_event(f0); c->refit_ev[1] = next_event(f0);
    f0.evused -= 2;
    HIPCHECK(c, hipEventRecord(c->refit_ev[0], c->stream));
    if (c->hdr.mode == CRT_BVH_WIDE) {
        launch_refit8((Node8*)(b + c->hdr.off_nodes), c->hdr.nnodes, (const Geom*)(b + c->hdr.off_geom), c->hdr.origin_bound, c->d_arrive,
                      (float*)(b + offsetof(BlobHeader, root_box)), grow_only ? 1 : 0, c->stream);
        launch_quantize((const Node8*)(b + c->hdr.off_nodes), c->hdr.nnodes, (QNode8*)(b + c->hdr.off_qnodes), c->d_status, c->stream);
    } else {
        launch_refit((Node4*)(b + c->hdr.off_nodes), c->hdr.nnodes, (const Geom*)(b + c->hdr.off_geom), c->d_arrive,
                     (float*)(b + offsetof(BlobHeader, root_box)), grow_only ? 1 : 0, c->stream);
    }
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipEventRecord(c->refit_ev[1], c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->h_status + 8, b + offsetof(BlobHeader, root_box), 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->h_status + 1, c->d_status, 4, hipMemcpyDeviceToHost, c->stream));
    c->refit_pending = true;
    return CRT_OK;
}
/* ... and the wait: the root box (it scales the traversal's slab pad) and the status word reach the host */
int ctx_refit_finish(crt_ctx* c) {
    if (!c) return CRT_ERR_ARG;
    if (!c->refit_pending) return fail(c, CRT_ERR_STATE, "crt_bvh_refit: nothing enqueued");
    c->refit_pending = false;
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    float ms = 0; hipEventElapsedTime(&ms, c->refit_ev[0], c->refit_ev[1]); c->stats.ms_refit = ms;
    memcpy(c->hdr.root_box, c->h_status + 8, 6 * sizeof(float));
    scene_abs_from_box(c);
    const uint32_t st = c->h_status[1];
    c->h_status[1] = 0;
    return status_error(c, st);
}
int ctx_trace_view(crt_ctx* c, TraceView* v) {
    if (!c || !v) return CRT_ERR_ARG;
    v->built = c->built;
    v->sv = c->built ? view_of(c) : SceneView{};
    v->stream = c->stream; v->device = c->device; v->wide = c->hdr.mode == CRT_BVH_WIDE;
    v->scene_abs = c->scene_abs; v->origin_bound = c->hdr.origin_bound; v->ovf = c->d_ovf; v->ovf_depth = c->ovf_depth;
    v->counters = c->d_testctr; v->status = c->d_status; v->h_status = c->h_status; v->grid_blocks = c->grid_blocks; v->block = c->trace_block;
    return CRT_OK;
}
}  // namespace crt

extern "C" {

crt_ctx* crt_create(int device, int* err) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { if (err) *err = CRT_ERR_HIP; return nullptr; }
    if (device < 0 || device >= ndev) { if (err) *err = CRT_ERR_ARG; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { if (err) *err = CRT_ERR_HIP; return nullptr; }
    crt_ctx* c = new crt_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&c->d_status, 4) != hipSuccess ||
        hipHostMalloc((void**)&c->h_status, 64, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&c->d_testctr, 32 * sizeof(unsigned long long)) != hipSuccess) {
        if (err) *err = CRT_ERR_HIP;
        delete c;
        return nullptr;
    }
    hipMemset(c->d_status, 0, 4);
    hipMemset(c->d_testctr, 0, 32 * sizeof(unsigned long long));
    c->grid_blocks = trace_grid_blocks(device);
    if (const char* v = getenv("CRT_TRACE_BLOCK")) { const int b = atoi(v); if (trace8_has_block(b)) c->trace_block = b; }
    {   /* crt_trace_noalloc's scratch array: its address on THIS device, looked up while this device is current */
        hipError_t se = hipSuccess;
        c->noalloc_scratch = noalloc_scratch_for_device(device, &se);
        if (!c->noalloc_scratch) { if (err) *err = CRT_ERR_HIP; crt_destroy(c); return nullptr; }
    }
    *c->h_status = 0;
    if (ensure_slot(c, 0) != CRT_OK) { if (err) *err = CRT_ERR_HIP; crt_destroy(c); return nullptr; }
    if (err) *err = CRT_OK;
    return c;
}

void crt_destroy(crt_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    cancel_async_build(c);
    if (c->build_stream) { hipStreamSynchronize(c->build_stream); hipStreamDestroy(c->build_stream); }
    if (c->ev_snap) hipEventDestroy(c->ev_snap);
    if (c->ev_build) hipEventDestroy(c->ev_build);
    for (FrameSlot& f : c->fs) if (f.stream) hipStreamSynchronize(f.stream);
    hipStreamSynchronize(c->stream);
    void* ptrs[] = {c->d_blob, c->d_blob2, c->d_prims_snap, c->d_prims, c->d_matid, c->d_arrive, c->d_status, c->d_testctr, c->d_ovf,
                    c->d_lbvh, c->d_lbvh_nodes, c->d_lbvh_s2s, c->d_area_table, c->d_refrank, c->d_refrank_out};
    for (void* p : ptrs) if (p) hipFree(p);
    for (FrameSlot& f : c->fs) {
        void* fp[] = {f.rayO[0], f.rayO[1], f.rayD[0], f.rayD[1], f.coef[0], f.coef[1], f.pix[0], f.pix[1], f.hitSlot, f.hitT,
                      f.contrib, f.shO, f.shD, f.accum, f.sum, f.shflag, f.d_bgr, f.cap_ids, f.cap_ts, f.d_tiles, f.d_ctr,
                      f.d_sortkeys, f.d_sortperm, f.d_sorthist, f.d_sortkeys2, f.d_sortperm2};
        for (void* p : fp) if (p) hipFree(p);
        if (f.h_ctr) hipHostFree(f.h_ctr);
        if (f.h_status) hipHostFree(f.h_status);
        if (f.graph_exec) hipGraphExecDestroy(f.graph_exec);
        if (f.graph) hipGraphDestroy(f.graph);
        for (hipEvent_t e : f.evpool) hipEventDestroy(e);
        if (f.ev_done) hipEventDestroy(f.ev_done);
        if (f.ev_scene) hipEventDestroy(f.ev_scene);
        if (f.own_stream && f.stream) hipStreamDestroy(f.stream);
    }
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->h_status) hipHostFree(c->h_status);
    for (uint32_t* q : c->tl_ids) if (q) hipFree(q);
    hipStreamDestroy(c->stream);
    delete c;
}

const char* crt_last_error(crt_ctx* c) { return c ? c->err.c_str() : "null context"; }

void crt_params_default(crt_params* p) {
    memset(p, 0, sizeof *p);
    p->width = 720; p->height = 360;        /* main.cu:141-142 */
    p->bounces = 2;                          /* Ray.cu:27 */
    p->spp = 1; p->tile_w = 32; p->tile_h = 32; p->rank = 0; p->nranks = 1; p->sort_rays = 0;
    p->area_samples = 30;                    /* lightSamples, Ray.cu:566 */
}

int crt_set_counting(crt_ctx* c, int on) { if (!c) return CRT_ERR_ARG; c->counting = on != 0; return CRT_OK; }
int crt_set_capture(crt_ctx* c, int on) { if (!c) return CRT_ERR_ARG; c->capture = on != 0; return CRT_OK; }

int crt_scene_set(crt_ctx* c, const crt_prim* prims, int n) {
    if (!c || n < 0 || (n > 0 && !prims)) return fail(c, CRT_ERR_ARG, "crt_scene_set: bad arguments");
    if (n > CRT_MAX_PRIMS) return fail(c, CRT_ERR_ARG, "crt_scene_set: too many primitives");
    for (int i = 0; i < n; i++)
        if (prims[i].type != 0 && prims[i].type != 1) return fail(c, CRT_ERR_ARG, "crt_scene_set: type must be 0 (sphere) or 1 (triangle)");
    if (!finite_geometry(prims, n)) return fail(c, CRT_ERR_ARG, "crt_scene_set: non-finite coordinate or negative radius");
    HIPCHECK(c, hipSetDevice(c->device));
    cancel_async_build(c);
    { const int rc = drain_frames(c); if (rc) return rc; }
    c->prims.assign(prims, prims + n);
    dedupe_materials(c);
    c->nonfinite_scene = shading_may_be_nonfinite(c->prims);
    c->origin_bound_scene = origin_bound_of(c->prims.data(), n);
    c->built = false; c->have_ref = false; c->imported = false; c->host_stale = false;
    return upload_prims(c);
}

int crt_scene_update(crt_ctx* c, const crt_prim* prims, int n) {
    if (!c || !prims) return fail(c, CRT_ERR_ARG, "crt_scene_update: bad arguments");
    if (c->imported) return fail(c, CRT_ERR_STATE, "crt_scene_update: scene was imported as a blob; set it with crt_scene_set first");
    if ((size_t)n != c->prims.size()) return fail(c, CRT_ERR_ARG, "crt_scene_update: count differs from crt_scene_set");
    for (int i = 0; i < n; i++)
        if (prims[i].type != c->prims[i].type) return fail(c, CRT_ERR_ARG, "crt_scene_update: primitive type changed");
    if (!finite_geometry(prims, n)) return fail(c, CRT_ERR_ARG, "crt_scene_update: non-finite coordinate or negative radius");
    HIPCHECK(c, hipSetDevice(c->device));
    { const int rc = order_after_frames(c); if (rc) return rc; }
    c->host_stale = false;
    /* materials keep their ids: only geometry (and normals) move, main.cu:545-573 */
    for (int i = 0; i < n; i++) {
        crt_prim& d = c->prims[i]; const crt_prim& s = prims[i];
        memcpy(d.position, s.position, 12); d.radius = s.radius;
        memcpy(d.v1, s.v1, 12); memcpy(d.v2, s.v2, 12); memcpy(d.v3, s.v3, 12);
        memcpy(d.v1Norm, s.v1Norm, 12); memcpy(d.v2Norm, s.v2Norm, 12); memcpy(d.v3Norm, s.v3Norm, 12);
        memcpy(d.direction, s.direction, 12); d.speed = s.speed;
    }
    int rc = upload_prims(c);
    if (rc) return rc;
    c->nonfinite_scene = shading_may_be_nonfinite(c->prims);          /* radii and normals moved */
    c->origin_bound_scene = origin_bound_of(c->prims.data(), n);      /* for the next device build; the current tree keeps the bound it was built for */
    if (c->built && ((c->hdr.flags & kBlobNonFiniteShading) != 0u) != c->nonfinite_scene) {
        c->hdr.flags = (c->hdr.flags & ~kBlobNonFiniteShading) | (c->nonfinite_scene ? kBlobNonFiniteShading : 0u);
        HIPCHECK(c, hipMemcpyAsync((char*)c->d_blob + offsetof(BlobHeader, flags), &c->hdr.flags, 4, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));      /* the source is a member of the context, not pinned memory */
    }
    if (c->built) {
        char* b = (char*)c->d_blob;
        launch_pack_prims(c->d_prims, c->d_matid, (const uint32_t*)(b + c->hdr.off_slot_of), (uint32_t)n,
                          (Geom*)(b + c->hdr.off_geom), (Shade*)(b + c->hdr.off_shade), c->stream);
        HIPCHECK(c, hipGetLastError());
    }
    return CRT_OK;
}

int crt_animate_device(crt_ctx* c) {
    if (!c) return CRT_ERR_ARG;
    if (c->imported) return fail(c, CRT_ERR_STATE, "crt_animate_device: scene was imported as a blob; set it with crt_scene_set first");
    HIPCHECK(c, hipSetDevice(c->device));
    { const int rc = order_after_frames(c); if (rc) return rc; }
    const uint32_t n = (uint32_t)c->prims.size();
    AnimParam a;
    animation_matrix(a.m);
    launch_animate(c->d_prims, n, a, c->stream);
    if (c->built) {
        char* b = (char*)c->d_blob;
        launch_pack_prims(c->d_prims, c->d_matid, (const uint32_t*)(b + c->hdr.off_slot_of), n,
                          (Geom*)(b + c->hdr.off_geom), (Shade*)(b + c->hdr.off_shade), c->stream);
    }
    HIPCHECK(c, hipGetLastError());
    c->host_stale = n > 0;
    return CRT_OK;
}

int crt_scene_get(crt_ctx* c, crt_prim* prims_out, int n) {
    if (!c || !prims_out) return CRT_ERR_ARG;
    if (c->imported) return fail(c, CRT_ERR_STATE, "crt_scene_get: scene was imported as a blob");
    if ((size_t)n != c->prims.size()) return fail(c, CRT_ERR_ARG, "crt_scene_get: count differs from crt_scene_set");
    HIPCHECK(c, hipSetDevice(c->device));
    if (c->host_stale) {
        HIPCHECK(c, hipMemcpyAsync(c->prims.data(), c->d_prims, (size_t)n * sizeof(crt_prim), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        c->host_stale = false;
    }
    memcpy(prims_out, c->prims.data(), (size_t)n * sizeof(crt_prim));
    return CRT_OK;
}

int crt_lights_set(crt_ctx* c, const crt_light* lights, int n) {
    if (!c || n < 0 || n > CRT_MAX_LIGHTS || (n > 0 && !lights)) return fail(c, CRT_ERR_ARG, "crt_lights_set: 0..CRT_MAX_LIGHTS lights");
    memset(&c->lights, 0, sizeof c->lights);
    for (int i = 0; i < n; i++) c->lights.l[i] = lights[i];
    c->lights.n = n;
    c->nonfinite_lights = lights_may_be_nonfinite(c->lights);
    return CRT_OK;
}

/* CRT_BVH_WIDE_DEVICE: lbvh.hip builds Node8 records from a device copy of the scene; the blob is laid out
 * once the node count is known, and from there on it is an ordinary wide-mode blob.  One device build, described so
 * that it can run on the context's stream into the current blob (crt_bvh_build) or on the build stream, from a snapshot
 * of the scene, into the spare blob (crt_bvh_build_async: a host thread of its own, so it reports through `err`). */
static int device_build_core(crt_ctx* c, DevBuild& B, std::string& err) {
#define BHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return CRT_ERR_HIP; } } while (0)
    const uint32_t n = (uint32_t)c->prims.size();
    auto t0 = std::chrono::steady_clock::now();
    auto grow = [&](void*& p, size_t& cap, size_t need) -> int {      /* byte-sized scratch that only grows */
        if (need <= cap && p) return CRT_OK;
        if (p) { BHIP(hipFree(p)); p = nullptr; cap = 0; }
        BHIP(hipMalloc(&p, need + need / 8 + 256));
        cap = need + need / 8 + 256;
        return CRT_OK;
    };
    int rc;
    size_t cap_nodes = c->lbvh_nodes_cap * sizeof(Node8), cap_s2s = c->lbvh_s2s_cap * 4, cap_rout = c->refrank_out_cap * 4;
    void *pn = c->d_lbvh_nodes, *ps = c->d_lbvh_s2s, *pr = c->d_refrank_out;
    if ((rc = grow(c->d_lbvh, c->lbvh_cap, lbvh_scratch_bytes(n)))) return rc;
    rc = grow(pn, cap_nodes, (size_t)n * sizeof(Node8)); c->d_lbvh_nodes = (Node8*)pn; c->lbvh_nodes_cap = cap_nodes / sizeof(Node8); if (rc) return rc;
    rc = grow(ps, cap_s2s, (size_t)n * 4); c->d_lbvh_s2s = (uint32_t*)ps; c->lbvh_s2s_cap = cap_s2s / 4; if (rc) return rc;
    uint32_t nnodes = 0, depth = 0;
    float root_box[6], exact_box[6];
    const float origin_bound = B.origin_bound;
    rc = lbvh_build(c->d_lbvh, B.prims, n, origin_bound, c->d_lbvh_nodes, n, c->d_lbvh_s2s, &nnodes, &depth, root_box, exact_box, B.s, &err);
    if (rc) return rc;
    /* tie-break ranks: the reference's own leaf order, replayed on the device (refrank.hip).  Where its recursion does not
     * terminate (more than 64 primitives sharing a centroid coordinate on the split axis) ties resolve by scene index */
    B.rank_valid = false;
    if (!getenv("CRT_DEVICE_BUILD_NO_REFRANK")) {
        if ((rc = grow(c->d_refrank, c->refrank_cap, refrank_scratch_bytes(n)))) return rc;
        rc = grow(pr, cap_rout, (size_t)n * 4); c->d_refrank_out = (uint32_t*)pr; c->refrank_out_cap = cap_rout / 4; if (rc) return rc;
        std::string rerr;
        rc = refrank_build(c->d_refrank, B.prims, n, exact_box, c->d_refrank_out, &B.levels, B.s, &rerr);
        if (rc == CRT_OK) B.rank_valid = true;
        else if (rc != CRT_ERR_BUILD) { err = rerr; return rc; }
    }
    BlobHeader h{};
    h.magic = kBlobMagic; h.version = kBlobVersion;
    h.nprims = n; h.nnodes = nnodes; h.nmats = (uint32_t)c->mats.size(); h.flags = c->nonfinite_scene ? kBlobNonFiniteShading : 0u;
    h.depth = depth; h.mode = CRT_BVH_WIDE; h.origin_bound = origin_bound;
    memcpy(h.root_box, root_box, sizeof h.root_box);
    layout_blob(h);
    if (h.total_bytes > B.blob_cap || !B.blob) {
        if (B.blob) BHIP(hipFree(B.blob));
        B.blob = nullptr; B.blob_cap = 0;
        BHIP(hipMalloc(&B.blob, h.total_bytes + h.total_bytes / 16));      /* a little room: the next rebuild's tree differs by a few nodes */
        B.blob_cap = h.total_bytes + h.total_bytes / 16;
    }
    B.hdr = h;
    char* b = (char*)B.blob;
    BHIP(hipMemcpyAsync(b, &B.hdr, sizeof h, hipMemcpyHostToDevice, B.s));
    BHIP(hipMemcpyAsync(b + h.off_nodes, c->d_lbvh_nodes, (size_t)nnodes * sizeof(Node8), hipMemcpyDeviceToDevice, B.s));
    launch_lbvh_maps(c->d_lbvh_s2s, B.rank_valid ? c->d_refrank_out : nullptr, n, (uint32_t*)(b + h.off_sidx), (uint32_t*)(b + h.off_slot_of), (uint32_t*)(b + h.off_rank), B.s);
    if (h.nmats) BHIP(hipMemcpyAsync(b + h.off_mats, c->mats.data(), (size_t)h.nmats * sizeof(Material), hipMemcpyHostToDevice, B.s));
    if (B.pack) {
        launch_pack_prims(B.prims, c->d_matid, (const uint32_t*)(b + h.off_slot_of), n, (Geom*)(b + h.off_geom), (Shade*)(b + h.off_shade), B.s);
        launch_quantize((const Node8*)(b + h.off_nodes), h.nnodes, (QNode8*)(b + h.off_qnodes), c->d_status, B.s);
    }
    BHIP(hipGetLastError());
    BHIP(hipStreamSynchronize(B.s));
    B.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return CRT_OK;
#undef BHIP
}

/* what follows a device build on the context's own thread: arrival counters, pad scale, overflow rows, bookkeeping */
static int adopt_device_tree(crt_ctx* c, const DevBuild& B) {
    c->d_blob = B.blob; c->blob_cap = B.blob_cap; c->blob_bytes = B.hdr.total_bytes; c->hdr = B.hdr;
    c->dev_rank_valid = B.rank_valid; c->refrank_levels = B.levels;
    int rc = check_status(c);
    if (rc) return rc;
    rc = ensure(c, c->d_arrive, c->arrive_cap, (size_t)B.hdr.nnodes + 1);
    if (rc) return rc;
    scene_abs_from_box(c);
    rc = size_overflow(c);
    if (rc) return rc;
    c->built = true; c->have_ref = false;
    c->ref = RefTree(); c->bvh = HostBVH();
    const int n = (int)c->prims.size();
    c->stats.nodes = (int)B.hdr.nnodes; c->stats.prims = n; c->stats.depth = (int)B.hdr.depth; c->stats.leaves = n;
    c->stats.scene_bytes = B.hdr.total_bytes;
    c->stats.ms_build_device = B.ms;
    return CRT_OK;
}

static int build_on_device(crt_ctx* c) {
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    DevBuild B{};
    B.s = c->stream; B.prims = c->d_prims; B.pack = true; B.blob = c->d_blob; B.blob_cap = c->blob_cap; B.origin_bound = c->origin_bound_scene;
    std::string err;
    const int rc = device_build_core(c, B, err);
    c->d_blob = B.blob; c->blob_cap = B.blob_cap;          /* the core may have reallocated it even when it failed later */
    if (rc) { c->built = false; return fail(c, rc, err.c_str()); }
    return adopt_device_tree(c, B);
}

int crt_bvh_build(crt_ctx* c, int mode) {
    if (!c) return CRT_ERR_ARG;
    if (mode != CRT_BVH_WIDE && mode != CRT_BVH_REFERENCE && mode != CRT_BVH_WIDE_DEVICE) return fail(c, CRT_ERR_ARG, "crt_bvh_build: unknown mode");
    if (c->imported) return fail(c, CRT_ERR_STATE, "crt_bvh_build: scene was imported as a blob");
    HIPCHECK(c, hipSetDevice(c->device));
    cancel_async_build(c);
    { const int rc = drain_frames(c); if (rc) return rc; }      /* the blob may move: no frame may still be reading it */
    if (mode == CRT_BVH_WIDE_DEVICE) {
        if (!c->prims.empty()) return build_on_device(c);
        mode = CRT_BVH_WIDE;                 /* nothing to sort: the host path writes the empty root */
    }
    const int n = (int)c->prims.size();
    if (c->host_stale) {          /* the scene was animated on the device: the host builders need its current state */
        HIPCHECK(c, hipMemcpyAsync(c->prims.data(), c->d_prims, (size_t)n * sizeof(crt_prim), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        c->host_stale = false;
    }
    auto t0 = std::chrono::steady_clock::now();
    /* the reference's own split (its leaf order is the tie-break rank).  It does not terminate when more than 64
     * primitives share a centroid coordinate on the axis it picks (bvh.cu:12-273 would recurse forever): the
     * reference topology then cannot be built at all; the wide tree is built without it and exact-t ties
     * resolve by scene index, as they do for the device build */
    int rc = build_reference_tree(c->prims.data(), n, c->ref);
    c->have_ref = rc == 0; c->dev_rank_valid = false;
    if (rc && mode == CRT_BVH_REFERENCE)
        return fail(c, rc, "BVH build: the reference's split does not terminate on this scene (more than 64 primitives share a centroid "
                           "coordinate on the split axis); use CRT_BVH_WIDE or CRT_BVH_WIDE_DEVICE");
    if (mode == CRT_BVH_WIDE) rc = build_wide(c->prims.data(), n, c->have_ref ? &c->ref : nullptr, c->bvh);
    else rc = pack_reference(c->prims.data(), n, c->ref, c->bvh);
    if (rc) return fail(c, rc, "BVH build failed");
    auto t1 = std::chrono::steady_clock::now();
    c->stats.ms_build_host = std::chrono::duration<double, std::milli>(t1 - t0).count();

    BlobHeader h{};
    h.magic = kBlobMagic; h.version = kBlobVersion;
    h.nprims = (uint32_t)n; h.nnodes = (uint32_t)c->bvh.node_count(); h.nmats = (uint32_t)c->mats.size(); h.flags = c->nonfinite_scene ? kBlobNonFiniteShading : 0u;
    h.depth = (uint32_t)c->bvh.depth; h.mode = (uint32_t)mode; h.origin_bound = mode == CRT_BVH_WIDE ? c->bvh.origin_bound : 0.f;
    memcpy(h.root_box, c->bvh.root_box, sizeof h.root_box);
    layout_blob(h);
    if (h.total_bytes > c->blob_cap || !c->d_blob) {
        if (c->d_blob) HIPCHECK(c, hipFree(c->d_blob));
        c->d_blob = nullptr;
        HIPCHECK(c, hipMalloc(&c->d_blob, h.total_bytes));
        c->blob_cap = h.total_bytes;
    }
    c->blob_bytes = h.total_bytes; c->hdr = h;
    char* b = (char*)c->d_blob;
    std::vector<uint32_t> rank_slot(n), slot_of(n);
    for (int s = 0; s < n; s++) {
        const uint32_t si = c->bvh.slot_to_scene[s];
        rank_slot[s] = c->bvh.rank_of_scene[si];
        slot_of[si] = (uint32_t)s;
    }
    HIPCHECK(c, hipMemcpyAsync(b, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    if (mode == CRT_BVH_WIDE)
        HIPCHECK(c, hipMemcpyAsync(b + h.off_nodes, c->bvh.nodes8.data(), (size_t)h.nnodes * sizeof(Node8), hipMemcpyHostToDevice, c->stream));
    else
        HIPCHECK(c, hipMemcpyAsync(b + h.off_nodes, c->bvh.nodes.data(), (size_t)h.nnodes * sizeof(Node4), hipMemcpyHostToDevice, c->stream));
    if (n) {
        HIPCHECK(c, hipMemcpyAsync(b + h.off_rank, rank_slot.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipMemcpyAsync(b + h.off_sidx, c->bvh.slot_to_scene.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipMemcpyAsync(b + h.off_slot_of, slot_of.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    }
    if (h.nmats) HIPCHECK(c, hipMemcpyAsync(b + h.off_mats, c->mats.data(), (size_t)h.nmats * sizeof(Material), hipMemcpyHostToDevice, c->stream));
    launch_pack_prims(c->d_prims, c->d_matid, (const uint32_t*)(b + h.off_slot_of), (uint32_t)n, (Geom*)(b + h.off_geom),
                      (Shade*)(b + h.off_shade), c->stream);
    if (mode == CRT_BVH_WIDE) launch_quantize((const Node8*)(b + h.off_nodes), h.nnodes, (QNode8*)(b + h.off_qnodes), c->d_status, c->stream);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipStreamSynchronize(c->stream));      /* host staging vectors go out of scope */
    rc = check_status(c);
    if (rc) return rc;
    rc = ensure(c, c->d_arrive, c->arrive_cap, (size_t)h.nnodes + 1);
    if (rc) return rc;
    scene_abs_from_box(c);
    rc = size_overflow(c);
    if (rc) return rc;
    c->built = true;
    c->stats.nodes = (int)h.nnodes; c->stats.prims = n; c->stats.depth = (int)h.depth; c->stats.leaves = c->bvh.leaves;
    c->stats.scene_bytes = h.total_bytes;
    return CRT_OK;
}

/* The rebuild of "rebuild every N frames" (main.cu:534-541) without the hitch: the device build runs on a stream and a host
 * thread of its own, from a snapshot of the scene taken now (device-side, ordered behind whatever the context's stream has
 * queued), into a spare blob, while frames keep rendering on the current tree (keep refitting it).  crt_bvh_swap makes the
 * new tree current. */
int crt_bvh_build_async(crt_ctx* c) {
    if (!c) return CRT_ERR_ARG;
    if (c->imported) return fail(c, CRT_ERR_STATE, "crt_bvh_build_async: scene was imported as a blob");
    if (!c->built) return fail(c, CRT_ERR_STATE, "crt_bvh_build_async: no BVH yet (the first build is crt_bvh_build)");
    if (c->prims.empty()) return fail(c, CRT_ERR_STATE, "crt_bvh_build_async: empty scene");
    if (c->build_state.load() != 0) return fail(c, CRT_ERR_STATE, "crt_bvh_build_async: a rebuild is already under way (crt_bvh_swap collects it)");
    HIPCHECK(c, hipSetDevice(c->device));
    const size_t n = c->prims.size();
    if (!c->build_stream) HIPCHECK(c, hipStreamCreateWithFlags(&c->build_stream, hipStreamNonBlocking));
    if (!c->ev_snap) HIPCHECK(c, hipEventCreateWithFlags(&c->ev_snap, hipEventDisableTiming));
    if (!c->ev_build) HIPCHECK(c, hipEventCreateWithFlags(&c->ev_build, hipEventDisableTiming));
    int rc = ensure(c, c->d_prims_snap, c->snap_cap, n);
    if (rc) return rc;
    /* frames still reading the spare blob (it was the current one until the last swap) finish first */
    for (int k = 0; k < CRT_MAX_FRAME_SLOTS; k++)
        if (c->fs[k].in_flight) HIPCHECK(c, hipStreamWaitEvent(c->build_stream, c->fs[k].ev_done, 0));
    /* the snapshot: behind the context's stream (the animation step just queued), and the context's stream behind it (the
     * next animation step must not move the scene under the copy) */
    HIPCHECK(c, hipEventRecord(c->ev_build, c->stream));
    HIPCHECK(c, hipStreamWaitEvent(c->build_stream, c->ev_build, 0));
    HIPCHECK(c, hipMemcpyAsync(c->d_prims_snap, c->d_prims, n * sizeof(crt_prim), hipMemcpyDeviceToDevice, c->build_stream));
    HIPCHECK(c, hipEventRecord(c->ev_snap, c->build_stream));
    HIPCHECK(c, hipStreamWaitEvent(c->stream, c->ev_snap, 0));
    c->async_build = DevBuild{};
    c->async_build.s = c->build_stream; c->async_build.prims = c->d_prims_snap; c->async_build.pack = false;
    c->async_build.blob = c->d_blob2; c->async_build.blob_cap = c->blob2_cap; c->async_build.origin_bound = c->origin_bound_scene;
    c->async_err.clear(); c->async_rc = 0;
    c->dev_rank_valid = false;          /* the rank buffer crt_bvh_reference_order reads is the builder's from here to the swap */
    c->build_state = 1;
    try {
        c->build_thread = std::thread([c] {
            (void)hipSetDevice(c->device);
            c->async_rc = device_build_core(c, c->async_build, c->async_err);
            c->build_state = c->async_rc ? 3 : 2;
        });
    } catch (const std::system_error&) {
        c->build_state = 0;
        return fail(c, CRT_ERR_NOMEM, "crt_bvh_build_async: cannot start the build thread");
    }
    return CRT_OK;
}

/* *swapped = 1 when the tree of the last crt_bvh_build_async became current: its geometry is packed from the scene as it
 * is NOW and its boxes are refitted to it (the topology is the snapshot's), so the call replaces this frame's crt_bvh_refit.
 * wait = 0 returns at once with *swapped = 0 while the build is still running. */
int crt_bvh_swap(crt_ctx* c, int wait, int* swapped) {
    if (!c) return CRT_ERR_ARG;
    if (swapped) *swapped = 0;
    const int st = c->build_state.load();
    if (st == 0) return fail(c, CRT_ERR_STATE, "crt_bvh_swap: no crt_bvh_build_async under way");
    if (st == 1 && !wait) return CRT_OK;
    if (c->build_thread.joinable()) c->build_thread.join();
    c->build_state = 0;
    HIPCHECK(c, hipSetDevice(c->device));
    DevBuild& B = c->async_build;
    c->d_blob2 = B.blob; c->blob2_cap = B.blob_cap;          /* reallocated or not, the spare blob is this one now */
    if (c->async_rc) return fail(c, c->async_rc, c->async_err.c_str());
    /* pointers change hands: the old tree becomes the spare one and stays valid for the frames still in flight on it */
    std::swap(c->d_blob, c->d_blob2); std::swap(c->blob_cap, c->blob2_cap);
    B.blob = c->d_blob; B.blob_cap = c->blob_cap;
    const int need_lds = stack_depth8_of(c->trace_block);
    if ((int)B.hdr.depth + 2 > need_lds && (size_t)c->grid_blocks * kTraceBlock * (size_t)((int)B.hdr.depth + 2 - need_lds) * 2 * CRT_MAX_FRAME_SLOTS > c->ovf_cap) {
        const int rc = drain_frames(c);          /* the overflow rows are about to move */
        if (rc) return rc;
    }
    int rc = adopt_device_tree(c, B);
    if (rc) return rc;
    char* b = (char*)c->d_blob;
    const uint32_t n = (uint32_t)c->prims.size();
    launch_pack_prims(c->d_prims, c->d_matid, (const uint32_t*)(b + c->hdr.off_slot_of), n, (Geom*)(b + c->hdr.off_geom), (Shade*)(b + c->hdr.off_shade), c->stream);
    HIPCHECK(c, hipGetLastError());
    rc = crt_bvh_refit(c, CRT_REFIT_TIGHT);      /* boxes of the snapshot's topology around today's positions, then the quantised nodes */
    if (rc) return rc;
    if (swapped) *swapped = 1;
    return CRT_OK;
}

int crt_bvh_refit(crt_ctx* c, int grow_only) {
    const int rc = crt::ctx_refit_enqueue(c, grow_only);
    return rc ? rc : crt::ctx_refit_finish(c);
}

int crt_bvh_free(crt_ctx* c) {
    if (!c) return CRT_ERR_ARG;
    HIPCHECK(c, hipSetDevice(c->device));
    cancel_async_build(c);
    { const int rc = drain_frames(c); if (rc) return rc; }
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if (c->d_blob) { HIPCHECK(c, hipFree(c->d_blob)); c->d_blob = nullptr; }
    c->blob_cap = c->blob_bytes = 0; c->built = false; c->imported = false;
    c->ref = RefTree(); c->bvh = HostBVH(); c->have_ref = false;
    return CRT_OK;
}

int crt_bvh_reference_order(crt_ctx* c, int32_t* rank_out, int n) {
    if (!c || !rank_out) return CRT_ERR_ARG;
    if ((size_t)n != c->prims.size()) return fail(c, CRT_ERR_ARG, "crt_bvh_reference_order: n differs from the scene size");
    if (!c->have_ref && c->dev_rank_valid && c->built) {          /* CRT_BVH_WIDE_DEVICE: the ranks were computed on the device */
        HIPCHECK(c, hipSetDevice(c->device));
        HIPCHECK(c, hipMemcpy(rank_out, c->d_refrank_out, (size_t)n * 4, hipMemcpyDeviceToHost));
        return CRT_OK;
    }
    if (!c->have_ref) return fail(c, CRT_ERR_STATE, "crt_bvh_reference_order: no reference order (no build yet, or a scene on which the reference's split does not terminate)");
    for (int i = 0; i < n; i++) rank_out[c->ref.order[i]] = i;
    return CRT_OK;
}

int crt_bvh_dump(crt_ctx* c, int32_t* nnodes, float* root_box6, float* child_boxes, uint32_t* child_links, int cap_nodes) {
    if (!c || !nnodes) return CRT_ERR_ARG;
    if (!c->built) return fail(c, CRT_ERR_STATE, "crt_bvh_dump: no BVH");
    HIPCHECK(c, hipSetDevice(c->device));
    *nnodes = (int32_t)c->hdr.nnodes;
    if (!child_boxes || !child_links) return CRT_OK;
    if (cap_nodes < (int)c->hdr.nnodes) return fail(c, CRT_ERR_ARG, "crt_bvh_dump: buffer too small");
    const char* b = (const char*)c->d_blob;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if (root_box6) HIPCHECK(c, hipMemcpy(root_box6, b + offsetof(BlobHeader, root_box), 24, hipMemcpyDeviceToHost));
    if (c->hdr.mode == CRT_BVH_WIDE) {
        std::vector<Node8> nodes(c->hdr.nnodes);
        HIPCHECK(c, hipMemcpy(nodes.data(), b + c->hdr.off_nodes, nodes.size() * sizeof(Node8), hipMemcpyDeviceToHost));
        export_nodes(nodes, child_boxes, child_links, nullptr);
    } else {
        std::vector<Node4> nodes(c->hdr.nnodes);
        HIPCHECK(c, hipMemcpy(nodes.data(), b + c->hdr.off_nodes, nodes.size() * sizeof(Node4), hipMemcpyDeviceToHost));
        export_nodes(nodes, child_boxes, child_links, nullptr);
    }
    return CRT_OK;
}

/* ---------------------------------------------------------------------------------------- */

int crt_tile_owner(int t, int nranks) {
    if (nranks <= 1) return 0;
    /* skewed round-robin: every rank gets a spatially uniform lattice of tiles (static balance) */
    return (int)(((long long)t + 3LL * (t / nranks)) % nranks);
}

int crt_tiles_owned(const crt_params* p, int32_t* ntiles, int64_t* npixels, int32_t* tile_ids, int cap) {
    if (!p || p->width <= 0 || p->height <= 0 || p->tile_w <= 0 || p->tile_h <= 0 || (p->tile_w & 7) || (p->tile_h & 7) ||
        p->nranks < 1 || p->rank < 0 || p->rank >= p->nranks)
        return CRT_ERR_ARG;
    const int tx = (p->width + p->tile_w - 1) / p->tile_w, ty = (p->height + p->tile_h - 1) / p->tile_h;
    int n = 0; int64_t np = 0;
    for (int t = 0; t < tx * ty; t++) {
        if (crt_tile_owner(t, p->nranks) != p->rank) continue;
        if (tile_ids && n < cap) tile_ids[n] = t;
        const int x0 = (t % tx) * p->tile_w, y0 = (t / tx) * p->tile_h;
        const int w = (x0 + p->tile_w <= p->width) ? p->tile_w : p->width - x0;
        const int h = (y0 + p->tile_h <= p->height) ? p->tile_h : p->height - y0;
        np += (int64_t)w * h;
        n++;
    }
    if (ntiles) *ntiles = n;
    if (npixels) *npixels = np;
    return CRT_OK;
}

static int fill_frame_params(crt_ctx* c, const crt_camera* cam, const crt_params* p, FrameParams& fp) {
    if (!cam || !p) return fail(c, CRT_ERR_ARG, "render: null camera or params");
    if (p->bounces < 1 || p->bounces > kMaxBounces) return fail(c, CRT_ERR_ARG, "render: bounces must be 1..16");
    int sn = 1;
    if (p->spp > 1) { sn = (int)lround(sqrt((double)p->spp)); if (sn * sn != p->spp || sn > 8) return fail(c, CRT_ERR_ARG, "render: spp must be a perfect square <= 64"); }
    if ((int64_t)p->width * p->height > (1LL << 31) - 1) return fail(c, CRT_ERR_ARG, "render: frame too large");
    /* Ray.cu:684-689 */
    const float width = (float)p->width, height = (float)p->height;
    const vec3 cameraPosition(cam->position), cameraDirection(cam->target);
    const vec3 n = normalize(cameraPosition - cameraDirection);
    const vec3 u = normalize(cross(vec3(0, 1, 0), n));
    const vec3 v = cross(n, u);
    const float fov = 45 / (180.0 / 3.14159265359);
    const float d = (height / tanf(fov / 2)) / 2;
    const vec3 L = (cameraPosition - n * d) - u * (width / 2) - v * (height / 2);
    memset(&fp, 0, sizeof fp);
    L.store(fp.L); u.store(fp.u); v.store(fp.v); cameraPosition.store(fp.cam);
    fp.width = p->width; fp.height = p->height; fp.tile_w = p->tile_w; fp.tile_h = p->tile_h;
    fp.tiles_x = (p->width + p->tile_w - 1) / p->tile_w;
    fp.spp_n = sn; fp.bounces = p->bounces; fp.capture = c->capture ? 1 : 0;
    if (p->zero_paths < 0 || p->zero_paths > 2) return fail(c, CRT_ERR_ARG, "render: zero_paths must be 0 (automatic), 1 (trace on) or 2 (end)");
    fp.keep_zero = p->zero_paths == 1 || (p->zero_paths == 0 && (c->nonfinite_scene || c->nonfinite_lights)) ? 1 : 0;
    return CRT_OK;
}

/* One frame enqueued on a slot's stream: gen_primary -> [trace -> shade -> trace(shadow + next bounce) -> resolve] x bounces
 * -> convert, then the frame's queue sizes and the status word into pinned host words, then the slot's done event.
 * Nothing here waits for the device (first use of a slot or of a frame size allocates, which does). */
int crt_render_submit(crt_ctx* c, int slot, const crt_camera* cam, const crt_params* p, void* dev_bgr) {
    if (!c) return CRT_ERR_ARG;
    if (slot < 0 || slot >= CRT_MAX_FRAME_SLOTS) return fail(c, CRT_ERR_ARG, "crt_render_submit: slot must be 0..CRT_MAX_FRAME_SLOTS-1");
    if (!c->built) return fail(c, CRT_ERR_STATE, "render: no BVH (call crt_scene_set + crt_bvh_build, or import a blob)");
    FrameSlot& f = c->fs[slot];
    if (f.in_flight) return fail(c, CRT_ERR_STATE, "crt_render_submit: the slot still holds a frame (crt_render_wait collects it)");
    HIPCHECK(c, hipSetDevice(c->device));
    FrameParams fp;
    int rc = fill_frame_params(c, cam, p, fp);
    if (rc) return rc;
    if (p->tile_nsub > 1 && (p->tile_sub < 0 || p->tile_sub >= p->tile_nsub)) return fail(c, CRT_ERR_ARG, "render: tile_sub must be 0..tile_nsub-1");
    if (p->grid_div < 0 || p->grid_div > 64) return fail(c, CRT_ERR_ARG, "render: grid_div must be 0..64");
    if (c->counting || p->use_graph) {
        /* the test counters are one set per context, and a captured graph is replayed on the stream it was captured on */
        if (slot != 0) return fail(c, CRT_ERR_ARG, "render: counting and graph replay run on slot 0 only");
        if (c->counting) for (int k = 1; k < CRT_MAX_FRAME_SLOTS; k++)
            if (c->fs[k].in_flight) return fail(c, CRT_ERR_STATE, "render: counting needs the other slots idle");
    }
    if ((rc = ensure_slot(c, slot))) return rc;
    rc = ensure_tiles(c, f, p);
    if (rc) return rc;
    fp.ntiles_owned = (int)f.tiles.size();
    const size_t cap = (size_t)fp.ntiles_owned * p->tile_w * p->tile_h;
    const size_t npix = (size_t)p->width * p->height;
    const int spp = fp.spp_n * fp.spp_n;
    /* light slots: one per directional / point light, area_samples per area light */
    LightsParam lights = c->lights;
    lights.area_samples = p->area_samples > 0 ? p->area_samples : 30;
    lights.nslots = 0;
    bool any_area = false;
    for (int j = 0; j < lights.n; j++) {
        lights.slot_base[j] = lights.nslots;
        lights.nslots += lights.l[j].area ? lights.area_samples : 1;
        any_area = any_area || lights.l[j].area;
    }
    if (any_area) {
        if (lights.area_samples > kAreaSamplesMax) return fail(c, CRT_ERR_ARG, "render: area_samples must be 1..64");
        if ((uint64_t)lights.nslots * (cap > 0 ? cap : 64) >= (1ull << 31))
            return fail(c, CRT_ERR_ARG, "render: area_samples x lights x pixels of this rank exceeds 2^31 shadow rays per bounce: render in more tiles/ranks");
        if (!c->d_area_table) {
            std::vector<float> tab((size_t)kAreaTable * 3);
            crt_area_table(tab.data(), kAreaTable);
            std::vector<float4> t4(kAreaTable);
            for (int i = 0; i < kAreaTable; i++) t4[i] = make_float4(tab[3 * i], tab[3 * i + 1], tab[3 * i + 2], 0.f);
            if ((rc = realloc_exact(c, c->d_area_table, (size_t)kAreaTable))) return rc;
            HIPCHECK(c, hipMemcpy(c->d_area_table, t4.data(), sizeof(float4) * kAreaTable, hipMemcpyHostToDevice));
        }
    }
    rc = ensure_frame(c, f, cap > 0 ? cap : 64, npix, p->bounces, lights.nslots, spp > 1);
    if (rc) return rc;
    if (!dev_bgr) {
        rc = ensure(c, f.d_bgr, f.bgr_cap, npix * 3);
        if (rc) return rc;
        dev_bgr = f.d_bgr;
    }
    if (p->sort_rays) {
        const size_t need = f.cap * (size_t)(lights.nslots > 1 ? lights.nslots : 1);
        if (need > f.sort_cap || !f.d_sortkeys) {
            if ((rc = realloc_exact(c, f.d_sortkeys, need))) return rc;
            if ((rc = realloc_exact(c, f.d_sortperm, need))) return rc;
            if ((rc = realloc_exact(c, f.d_sortkeys2, f.cap))) return rc;
            if ((rc = realloc_exact(c, f.d_sortperm2, f.cap))) return rc;
            if (!f.d_sorthist && (rc = realloc_exact(c, f.d_sorthist, (size_t)kSortBins + 1))) return rc;
            f.sort_cap = need;
        }
    }
    rc = ensure(c, f.d_ctr, f.ctr_cap, (size_t)spp * kCtrStride);
    if (rc) return rc;
    if ((size_t)spp * kCtrFetch0 > f.h_ctr_cap || !f.h_ctr) {
        if (f.h_ctr) { HIPCHECK(c, hipHostFree(f.h_ctr)); f.h_ctr = nullptr; f.h_ctr_cap = 0; }
        HIPCHECK(c, hipHostMalloc((void**)&f.h_ctr, (size_t)spp * kCtrFetch0 * 4, hipHostMallocDefault));
        f.h_ctr_cap = (size_t)spp * kCtrFetch0;
    }
    if (slot > 0) {      /* behind whatever the context's stream has queued for the scene (animation, refit, upload) */
        HIPCHECK(c, hipEventRecord(f.ev_scene, c->stream));
        HIPCHECK(c, hipStreamWaitEvent(f.stream, f.ev_scene, 0));
    }
    HIPCHECK(c, hipMemsetAsync(f.d_ctr, 0, (size_t)spp * kCtrStride * 4, f.stream));
    static const bool tail_probe = getenv("CRT_TAIL_PROBE") != nullptr;
    if (c->counting || tail_probe) HIPCHECK(c, hipMemsetAsync(c->d_testctr, 0, 32 * sizeof(unsigned long long), f.stream));
    if (c->capture) {
        HIPCHECK(c, hipMemsetD32Async((hipDeviceptr_t)f.cap_ids, -9, (size_t)p->bounces * npix, f.stream));
        HIPCHECK(c, hipMemsetD32Async((hipDeviceptr_t)f.cap_ts, 0, (size_t)p->bounces * npix, f.stream));
    }
    f.evused = 0; f.evlog.clear(); f.ev_error = false; f.light_timing = p->timing == 1;
    const SceneView sv = view_of(c);
    /* the slab-test pad scales with the scene's extent, a kernel argument: a replayed graph keeps the value it was
     * captured with, so graph mode rounds it up to a power of two (a larger pad is as conservative) and re-captures
     * only when an animated scene outgrows it */
    float sabs = c->scene_abs;
    int gb = c->grid_blocks / (p->grid_div > 1 ? p->grid_div : 1);
    if (gb < 1) gb = 1;
    uint32_t* const ovf = c->d_ovf + (size_t)slot * c->ovf_slot_words;
    if (p->use_graph) { int e = 0; (void)frexpf(sabs > 0.f ? sabs : 1.f, &e); sabs = ldexpf(1.0f, e); }
    auto enqueue = [&]() {
    for (int s = 0; s < spp; s++) {
        uint32_t* ctr = f.d_ctr + (size_t)s * kCtrStride;
        fp.sample = s;
        int cur = 0;
        FrameBuffers fb{};
        fb.hitSlot = f.hitSlot; fb.hitT = f.hitT; fb.contrib = f.contrib; fb.shflag = f.shflag;
        fb.shO = f.shO; fb.shD = f.shD; fb.area_table = c->d_area_table; fb.accum = f.accum; fb.cap_ids = f.cap_ids; fb.cap_ts = f.cap_ts;
        fb.cap = (uint32_t)f.cap; fb.cap_stride = (uint32_t)npix; fb.tile_ids = f.d_tiles;
        auto bind = [&](int k) {
            fb.rayO = f.rayO[k]; fb.rayD = f.rayD[k]; fb.coef = f.coef[k]; fb.pix = f.pix[k];
            fb.nrayO = f.rayO[k ^ 1]; fb.nrayD = f.rayD[k ^ 1]; fb.ncoef = f.coef[k ^ 1]; fb.npix = f.pix[k ^ 1];
        };
        bind(cur);
        launch_gen_primary(fp, fb, ctr + 0, f.stream);
        mark(f, EV_GEN);
        const float* d_root_box = (const float*)((const char*)c->d_blob + offsetof(BlobHeader, root_box));
        const bool wide = c->hdr.mode == CRT_BVH_WIDE;
        auto common = [&](TraceArgs& t, int launch) {
            t.sv = sv; t.next = ctr + kCtrFetch0 + launch * 8 * kFetchStride;
            t.hitSlot = f.hitSlot; t.hitT = f.hitT; t.shflag = f.shflag; t.scene_abs = sabs; t.origin_bound = c->hdr.origin_bound;
            t.ovf = ovf; t.ovf_depth = c->ovf_depth; t.counters = c->d_testctr; t.status = c->d_status; t.block = c->trace_block;
        };
        for (int b = 0; b < p->bounces; b++) {
            fp.bounce = b;
            bind(cur);
            if (!wide || b == 0) {
                /* nearest hits of this bounce: the wide path has traced them already, together with the previous
                 * bounce's shadow rays, except for the primary rays */
                TraceArgs ta{};
                common(ta, 2 * b);
                ta.rayO = fb.rayO; ta.rayD = fb.rayD; ta.count = ctr + b;
                launch_trace(ta, TRACE_NEAREST, c->counting, wide, gb, f.stream);
                mark(f, EV_TRACE_N);
            }
            launch_shade(fp, fb, sv, lights, ctr + b, ctr + b + 1, ctr + kCtrShadow0 + b, f.stream);
            mark(f, EV_SHADE);
            if (wide) {
                /* ONE launch for the shadow rays of this bounce and the nearest-hit rays of the next: they are
                 * independent, and every persistent launch ends in a ~0.25 ms tail of its longest rays */
                TraceArgs tm{};
                common(tm, 2 * b + 1);
                const bool has_sh = c->lights.n > 0, has_nx = b + 1 < p->bounces;
                if (has_sh) {
                    tm.shO = f.shO; tm.shD = f.shD; tm.shcount = ctr + kCtrShadow0 + b;
                    if (p->sort_rays) {
                        launch_sort_rays(f.shO, ctr + kCtrShadow0 + b, (uint32_t)(f.cap * (size_t)lights.nslots), d_root_box, f.d_sortkeys, f.d_sorthist,
                                         f.d_sortperm, f.stream);
                        tm.shperm = f.d_sortperm;
                        mark(f, EV_SORT);
                    }
                }
                if (has_nx) {
                    tm.rayO = fb.nrayO; tm.rayD = fb.nrayD; tm.count = ctr + b + 1;
                    if (p->sort_rays) {      /* reflection rays: visit them by origin cell (primary rays are coherent already) */
                        launch_sort_rays(fb.nrayO, ctr + b + 1, (uint32_t)f.cap, d_root_box, f.d_sortkeys2, f.d_sorthist, f.d_sortperm2, f.stream);
                        tm.perm = f.d_sortperm2;
                        mark(f, EV_SORT);
                    }
                }
                if (has_sh || has_nx) {
                    launch_trace(tm, TRACE_NEAREST, c->counting, true, gb, f.stream);
                    mark(f, has_sh && has_nx ? EV_TRACE_M : (has_sh ? EV_TRACE_S : EV_TRACE_N));
                }
            } else if (c->lights.n > 0) {
                TraceArgs ts{};
                common(ts, 2 * b + 1);
                ts.rayO = f.shO; ts.rayD = f.shD; ts.count = ctr + kCtrShadow0 + b;
                ts.hitSlot = nullptr; ts.hitT = nullptr; ts.counters = c->d_testctr + 16;
                launch_trace(ts, TRACE_SHADOW, c->counting, false, gb, f.stream);
                mark(f, EV_TRACE_S);
            }
            launch_resolve(fb, lights, ctr + b, b, f.stream);
            mark(f, EV_RESOLVE);
            cur ^= 1;
        }
        if (spp > 1) launch_spp_accumulate(fp, fb, f.sum, f.stream);
    }
    {
        FrameBuffers fb{};
        fb.accum = f.accum; fb.tile_ids = f.d_tiles;
        launch_convert(fp, fb, spp > 1 ? f.sum : nullptr, (uint8_t*)dev_bgr, f.stream);
        mark(f, EV_CONVERT);
    }
    };
    if (!p->use_graph) {
        mark(f, EV_START);
        enqueue();
    } else {
        FrameSlot::GraphKey key{};
        key.cam = *cam; key.p = *p; key.lights = lights;
        /* everything a captured launch bakes in: every buffer (size_overflow() may move d_ovf and change ovf_depth on any
         * rebuild or import that leaves the blob's address and counts as they were), every header field the offsets derive from */
        const void* ptrs[20] = {c->d_blob, f.rayO[0], f.rayO[1], f.hitSlot, f.hitT, f.contrib, f.shO, f.accum, f.d_ctr, dev_bgr,
                                f.d_tiles, f.cap_ids, c->d_ovf, f.d_sortkeys, f.d_sortperm, f.d_sortkeys2, f.d_sortperm2, f.d_sorthist,
                                c->d_area_table, f.sum};
        memcpy(key.ptrs, ptrs, sizeof ptrs);
        key.nnodes = c->hdr.nnodes; key.nprims = c->hdr.nprims; key.nmats = c->hdr.nmats; key.depth = c->hdr.depth; key.mode = c->hdr.mode;
        key.counting = c->counting; key.capture = c->capture; key.grid = gb; key.ovf_depth = c->ovf_depth; key.block = c->trace_block; key.keep_zero = fp.keep_zero; key.scene_abs = sabs; key.origin_bound = c->hdr.origin_bound;
        if (!f.graph_exec || memcmp(&key, &f.graph_key, sizeof key) != 0) {
            if (f.graph_exec) { hipGraphExecDestroy(f.graph_exec); f.graph_exec = nullptr; }
            if (f.graph) { hipGraphDestroy(f.graph); f.graph = nullptr; }
            HIPCHECK(c, hipStreamBeginCapture(f.stream, hipStreamCaptureModeRelaxed));
            f.in_capture = true;
            enqueue();
            f.in_capture = false;
            HIPCHECK(c, hipStreamEndCapture(f.stream, &f.graph));
            HIPCHECK(c, hipGraphInstantiate(&f.graph_exec, f.graph, nullptr, nullptr, 0));
            memcpy(&f.graph_key, &key, sizeof key);
        }
        mark(f, EV_START);
        HIPCHECK(c, hipGraphLaunch(f.graph_exec, f.stream));
        mark(f, EV_CONVERT);
    }
    HIPCHECK(c, hipGetLastError());
    for (int s = 0; s < spp; s++)      /* queue sizes only; the fetch counters stay on the device */
    {
        HIPCHECK(c, hipMemcpyAsync(f.h_ctr + (size_t)s * kCtrFetch0, f.d_ctr + (size_t)s * kCtrStride, 16 * 4, hipMemcpyDeviceToHost, f.stream));
        HIPCHECK(c, hipMemcpyAsync(f.h_ctr + (size_t)s * kCtrFetch0 + 16, f.d_ctr + (size_t)s * kCtrStride + kCtrShadow0, 16 * 4, hipMemcpyDeviceToHost, f.stream));
    }
    HIPCHECK(c, hipMemcpyAsync(f.h_status, c->d_status, 4, hipMemcpyDeviceToHost, f.stream));
    HIPCHECK(c, hipEventRecord(f.ev_done, f.stream));
    f.in_flight = true;
    f.sub_p = *p; f.sub_spp = spp; f.sub_counting = c->counting; f.sub_grid = gb;
    return CRT_OK;
}

/* Completes the frame submitted on `slot`: waits for its done event, checks the status word, and turns the frame's
 * counters and events into crt_stats (crt_get_stats then describes this frame). */
int crt_render_wait(crt_ctx* c, int slot) {
    if (!c) return CRT_ERR_ARG;
    if (slot < 0 || slot >= CRT_MAX_FRAME_SLOTS) return fail(c, CRT_ERR_ARG, "crt_render_wait: slot must be 0..CRT_MAX_FRAME_SLOTS-1");
    FrameSlot& f = c->fs[slot];
    if (!f.in_flight) return fail(c, CRT_ERR_STATE, "crt_render_wait: no frame was submitted on this slot");
    HIPCHECK(c, hipSetDevice(c->device));
    f.in_flight = false;
    HIPCHECK(c, hipEventSynchronize(f.ev_done));
    const crt_params* p = &f.sub_p;
    const int spp = f.sub_spp;
    int rc = status_error(c, *f.h_status);
    *f.h_status = 0;
    if (rc) return rc;
    if (f.ev_error || f.evlog.size() < 2) return fail(c, CRT_ERR_HIP, "render: a timing event could not be created or recorded (the frame itself is complete)");

    crt_stats& st = c->stats;
    st.rays_primary = st.rays_shadow = st.rays_reflect = 0;
    for (int s = 0; s < spp; s++) {
        const uint32_t* h = f.h_ctr + (size_t)s * kCtrFetch0;
        (void)h[0];                       /* queue length including holes of clipped tiles */
        st.rays_primary += (uint64_t)f.owned_pixels;
        for (int b = 1; b < p->bounces; b++) st.rays_reflect += h[b];
        for (int b = 0; b < p->bounces; b++) st.rays_shadow += h[16 + b];
    }
    st.ms_trace = st.ms_shade = st.ms_other = st.ms_trace_nearest = st.ms_trace_shadow = st.ms_trace_mixed = 0;
    st.launches_trace_nearest = st.launches_trace_shadow = st.launches_trace_mixed = 0;
    for (size_t i = 1; i < f.evlog.size(); i++) {
        float ms = 0;
        hipEventElapsedTime(&ms, f.evpool[f.evlog[i - 1].second], f.evpool[f.evlog[i].second]);
        switch (f.evlog[i].first) {
            case EV_TRACE_N: st.ms_trace += ms; st.ms_trace_nearest += ms; st.launches_trace_nearest++; break;
            case EV_TRACE_S: st.ms_trace += ms; st.ms_trace_shadow += ms; st.launches_trace_shadow++; break;
            case EV_TRACE_M: st.ms_trace += ms; st.ms_trace_mixed += ms; st.launches_trace_mixed++; break;
            case EV_SHADE: case EV_RESOLVE: st.ms_shade += ms; break;
            default: st.ms_other += ms;
        }
    }
    float tot = 0;
    hipEventElapsedTime(&tot, f.evpool[f.evlog.front().second], f.evpool[f.evlog.back().second]);
    st.ms_total = tot;
    if (f.light_timing) { st.ms_other = 0; }          /* crt_params.timing = 1: start and end only */
    static const bool tail_probe = getenv("CRT_TAIL_PROBE") != nullptr;
    const int waves_per_launch = f.sub_grid * (kTraceBlock / 64);
    if (tail_probe && !f.sub_counting) {
        unsigned long long t[32];
        HIPCHECK(c, hipMemcpy(t, c->d_testctr, sizeof t, hipMemcpyDeviceToHost));
        const int launches = st.launches_trace_nearest + st.launches_trace_shadow + st.launches_trace_mixed;
        fprintf(stderr, "[crt tail] %d launches, %.3f ms of traversal: wave tails (queue dry -> wave done) longest %.0f, mean %.0f shader clocks (s_memtime; / ~2320 for us)\n",
                launches, st.ms_trace, (double)t[30], launches ? (double)t[31] / ((double)waves_per_launch * launches) : 0.0);
    }
    if (f.sub_counting) {
        unsigned long long t[32];
        HIPCHECK(c, hipMemcpy(t, c->d_testctr, sizeof t, hipMemcpyDeviceToHost));
        if (getenv("CRT_DEBUG_SCHED"))      /* wave scheduling statistics of k_trace8 */
        {
            const unsigned long long* q = t + 8;
            fprintf(stderr, "[crt sched] node iterations %llu (%.1f lanes served), prim iterations %llu (%.1f lanes served), active lanes per iteration %.1f, "
                            "visits with no child hit %llu of %llu; shadow rays that found an occluder %llu, their node visits after it %llu; "
                            "tail iterations %llu; steal steps %llu, stack entries handed over %.0f\n",
                    q[0], q[0] ? (double)q[1] / q[0] : 0., q[2], q[2] ? (double)q[3] / q[2] : 0., (q[0] + q[2]) ? (double)q[4] / (q[0] + q[2]) : 0., q[5], t[3] + t[19], q[7], q[6],
                    t[27], t[30], (double)t[31]);
        }
        st.box_tests = t[0]; st.tri_tests = t[1]; st.sph_tests = t[2]; st.node_visits = t[3];
        st.box_tests_shadow = t[16]; st.tri_tests_shadow = t[17]; st.sph_tests_shadow = t[18]; st.node_visits_shadow = t[19];
        st.sched_node_iterations = t[8]; st.sched_node_lanes = t[9]; st.sched_prim_iterations = t[10]; st.sched_prim_lanes = t[11];
        st.sched_active_lanes = t[12]; st.visits_no_child_hit = t[13]; st.visits_no_child_hit_tlimit = t[28]; st.boxes_rejected_by_tlimit = t[29];
        st.sched_iterations = t[20]; st.sched_refill_iterations = t[21]; st.sched_finish_iterations = t[22]; st.sched_setup_iterations = t[23];
        st.sched_pop_iterations = t[24]; st.sched_decode_iterations = t[25]; st.sched_hit_iterations = t[26];
        st.trace_waves = (uint64_t)(st.launches_trace_nearest + st.launches_trace_shadow + st.launches_trace_mixed) * (uint64_t)waves_per_launch;
    }
    f.last_w = p->width; f.last_h = p->height; f.last_bounces = p->bounces;
    return CRT_OK;
}

int crt_render_device(crt_ctx* c, const crt_camera* cam, const crt_params* p, void* dev_bgr) {
    const int rc = crt_render_submit(c, 0, cam, p, dev_bgr);
    return rc ? rc : crt_render_wait(c, 0);
}

int crt_render(crt_ctx* c, const crt_camera* cam, const crt_params* p, unsigned char* bgr_out_host) {
    if (!c || !bgr_out_host) return fail(c, CRT_ERR_ARG, "crt_render: null output buffer");
    int rc = crt_render_device(c, cam, p, nullptr);
    if (rc) return rc;
    FrameSlot& f0 = c->fs[0];
    const size_t bytes = (size_t)p->width * p->height * 3;
    /* The caller's buffer is ordinary pageable memory (the reference mallocs it, main.cu:512): a plain hipMemcpy into it
     * runs at ~7 GB/s.  Frames of a few MB and more go through a pinned staging buffer instead: the device copies it in
     * chunks at PCIe speed while host threads move finished chunks into the caller's buffer. */
    constexpr size_t kStageMin = 2u << 20;
    constexpr int kChunks = 8, kThreads = 4;
    if (bytes < kStageMin) {
        HIPCHECK(c, hipMemcpy(bgr_out_host, f0.d_bgr, bytes, hipMemcpyDeviceToHost));
        return CRT_OK;
    }
    if (bytes > c->stage_cap || !c->h_stage) {
        if (c->h_stage) { HIPCHECK(c, hipHostFree(c->h_stage)); c->h_stage = nullptr; c->stage_cap = 0; }
        if (hipHostMalloc((void**)&c->h_stage, bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->h_stage = nullptr;                    /* no pinned memory to be had: the plain copy still works */
            HIPCHECK(c, hipMemcpy(bgr_out_host, f0.d_bgr, bytes, hipMemcpyDeviceToHost));
            return CRT_OK;
        }
        c->stage_cap = bytes;
    }
    const size_t chunk = ((bytes + kChunks - 1) / kChunks + 4095) & ~(size_t)4095;
    hipEvent_t ev[kChunks];
    int nchunks = 0;
    if (!reserve_events(f0, (size_t)kChunks)) return fail(c, CRT_ERR_HIP, "crt_render: hipEventCreate failed");
    for (size_t off = 0; off < bytes; off += chunk, nchunks++) {
        const size_t len = off + chunk < bytes ? chunk : bytes - off;
        ev[nchunks] = next_event(f0);
        HIPCHECK(c, hipMemcpyAsync(c->h_stage + off, (const unsigned char*)f0.d_bgr + off, len, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipEventRecord(ev[nchunks], c->stream));
    }
    f0.evused -= nchunks;                            /* the events go back to the pool: they are done when this call returns */
    std::thread workers[kThreads];
    hipError_t werr[kThreads];
    for (int t = 0; t < kThreads; t++) {
        werr[t] = hipSuccess;
        workers[t] = std::thread([&, t] {
            (void)hipSetDevice(c->device);
            for (int k = t; k < nchunks; k += kThreads) {
                const hipError_t e = hipEventSynchronize(ev[k]);
                if (e != hipSuccess) { werr[t] = e; return; }
                const size_t off = (size_t)k * chunk, len = off + chunk < bytes ? chunk : bytes - off;
                memcpy(bgr_out_host + off, c->h_stage + off, len);
            }
        });
    }
    for (auto& w : workers) w.join();
    for (int t = 0; t < kThreads; t++) HIPCHECK(c, werr[t]);
    return CRT_OK;
}

int crt_get_capture(crt_ctx* c, int bounce, int32_t* ids, float* ts, int npixels) {
    if (!c || !ids || !ts) return CRT_ERR_ARG;
    FrameSlot& f0 = c->fs[0];
    if (!c->capture || !f0.cap_ids) return fail(c, CRT_ERR_STATE, "crt_get_capture: capture was not enabled before the render");
    if (bounce < 0 || bounce >= f0.last_bounces || npixels != f0.last_w * f0.last_h) return fail(c, CRT_ERR_ARG, "crt_get_capture: bad bounce or size");
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipMemcpy(ids, f0.cap_ids + (size_t)bounce * npixels, (size_t)npixels * 4, hipMemcpyDeviceToHost));
    HIPCHECK(c, hipMemcpy(ts, f0.cap_ts + (size_t)bounce * npixels, (size_t)npixels * 4, hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_get_color(crt_ctx* c, float* rgb, int npixels) {
    if (!c || !rgb) return CRT_ERR_ARG;
    FrameSlot& f0 = c->fs[0];
    if (!f0.accum || npixels != f0.last_w * f0.last_h) return fail(c, CRT_ERR_ARG, "crt_get_color: no frame of that size");
    HIPCHECK(c, hipSetDevice(c->device));
    std::vector<float4> tmp(npixels);
    HIPCHECK(c, hipMemcpy(tmp.data(), f0.accum, (size_t)npixels * sizeof(float4), hipMemcpyDeviceToHost));
    for (int i = 0; i < npixels; i++) { rgb[3 * i] = tmp[i].x; rgb[3 * i + 1] = tmp[i].y; rgb[3 * i + 2] = tmp[i].z; }
    return CRT_OK;
}

static int trace_hook(crt_ctx* c, const float* org_dir, int n, int32_t* prim_out, float* t_out, uint8_t* shadow_out) {
    if (!c || !org_dir || n < 0) return fail(c, CRT_ERR_ARG, "crt_trace: bad arguments");
    if (!c->built) return fail(c, CRT_ERR_STATE, "crt_trace: no BVH");
    if (n == 0) return CRT_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    float* d_od = nullptr; float4 *d_o = nullptr, *d_d = nullptr; int32_t *d_slot = nullptr, *d_id = nullptr; float* d_t = nullptr;
    uint8_t* d_flag = nullptr; uint32_t* d_cnt = nullptr;
    int rc = CRT_OK;
#define TH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { c->err = std::string(#call) + ": " + hipGetErrorString(e_); rc = CRT_ERR_HIP; goto done; } } while (0)
    TH(hipMalloc((void**)&d_od, (size_t)n * 24)); TH(hipMalloc((void**)&d_o, (size_t)n * 16)); TH(hipMalloc((void**)&d_d, (size_t)n * 16));
    TH(hipMalloc((void**)&d_slot, (size_t)n * 4)); TH(hipMalloc((void**)&d_id, (size_t)n * 4)); TH(hipMalloc((void**)&d_t, (size_t)n * 4));
    TH(hipMalloc((void**)&d_flag, (size_t)n)); TH(hipMalloc((void**)&d_cnt, (1 + 8 * kFetchStride) * 4 + 256));
    {
        uint32_t cnt = (uint32_t)n;
        TH(hipMemcpyAsync(d_od, org_dir, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
        TH(hipMemsetAsync(d_cnt, 0, (1 + 8 * kFetchStride) * 4 + 256, c->stream));
        TH(hipMemcpyAsync(d_cnt, &cnt, 4, hipMemcpyHostToDevice, c->stream));
        TH(hipMemsetAsync(d_flag, 0, (size_t)n, c->stream));
        launch_pack_rays(d_od, (uint32_t)n, d_o, d_d, shadow_out != nullptr, c->stream);
        TraceArgs ta{};
        ta.sv = view_of(c); ta.rayO = d_o; ta.rayD = d_d; ta.count = d_cnt; ta.next = d_cnt + 64;
        ta.hitSlot = d_slot; ta.hitT = d_t; ta.shflag = d_flag; ta.scene_abs = c->scene_abs; ta.origin_bound = c->hdr.origin_bound;
        ta.ovf = c->d_ovf; ta.ovf_depth = c->ovf_depth; ta.counters = c->d_testctr; ta.status = c->d_status; ta.block = c->trace_block;
        if (c->counting) TH(hipMemsetAsync(c->d_testctr, 0, 32 * sizeof(unsigned long long), c->stream));
        if (shadow_out && c->hdr.mode == CRT_BVH_WIDE) {      /* k_trace8 takes shadow rays in its second queue */
            ta.shO = d_o; ta.shD = d_d; ta.shcount = d_cnt;
            ta.rayO = nullptr; ta.rayD = nullptr; ta.count = nullptr;
        }
        if (!reserve_events(c->fs[0], 2)) { c->err = "crt_trace: hipEventCreate failed"; rc = CRT_ERR_HIP; goto done; }
        hipEvent_t e0 = next_event(c->fs[0]), e1 = next_event(c->fs[0]);
        c->fs[0].evused -= 2;
        TH(hipEventRecord(e0, c->stream));
        launch_trace(ta, shadow_out ? TRACE_SHADOW : TRACE_NEAREST, c->counting, c->hdr.mode == CRT_BVH_WIDE, c->grid_blocks, c->stream);
        TH(hipGetLastError());
        TH(hipEventRecord(e1, c->stream));
        if (shadow_out) TH(hipMemcpyAsync(shadow_out, d_flag, (size_t)n, hipMemcpyDeviceToHost, c->stream));
        else {
            launch_slots_to_scene(d_slot, ta.sv.sidx, d_id, (uint32_t)n, c->stream);
            TH(hipMemcpyAsync(prim_out, d_id, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
            TH(hipMemcpyAsync(t_out, d_t, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        }
        TH(hipStreamSynchronize(c->stream));
        { float ms = 0; hipEventElapsedTime(&ms, e0, e1); c->stats.ms_trace = ms; }     /* the traversal launch alone */
        if (c->counting) {
            unsigned long long t[32];
            TH(hipMemcpy(t, c->d_testctr, sizeof t, hipMemcpyDeviceToHost));
            const unsigned long long* q = (shadow_out && c->hdr.mode == CRT_BVH_WIDE) ? t + 16 : t;
            c->stats.box_tests = q[0]; c->stats.tri_tests = q[1]; c->stats.sph_tests = q[2]; c->stats.node_visits = q[3];
        }
    }
done:
#undef TH
    hipFree(d_od); hipFree(d_o); hipFree(d_d); hipFree(d_slot); hipFree(d_id); hipFree(d_t); hipFree(d_flag); hipFree(d_cnt);
    if (rc) return rc;
    return check_status(c);
}

int crt_powf_probe(crt_ctx* c, const float* x, const float* y, int n, float* out) {
    if (!c || n < 0 || (n > 0 && (!x || !y || !out))) return fail(c, CRT_ERR_ARG, "crt_powf_probe: null argument");
    if (n == 0) return CRT_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * sizeof(float);
    int rc = CRT_OK;
    if (hipMalloc(&dx, bytes) != hipSuccess || hipMalloc(&dy, bytes) != hipSuccess || hipMalloc(&dout, bytes) != hipSuccess) rc = fail(c, CRT_ERR_HIP, "crt_powf_probe: hipMalloc failed");
    if (rc == CRT_OK && (hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                         hipMemcpyAsync(dy, y, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess)) rc = fail(c, CRT_ERR_HIP, "crt_powf_probe: copy failed");
    if (rc == CRT_OK) {
        launch_powf_probe(dx, dy, (uint32_t)n, dout, c->stream);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, CRT_ERR_HIP, "crt_powf_probe: launch or copy failed");
    }
    if (dx) hipFree(dx);
    if (dy) hipFree(dy);
    if (dout) hipFree(dout);
    return rc;
}

int crt_trace(crt_ctx* c, const float* org_dir, int n, int32_t* prim_out, float* t_out) {
    if (!prim_out || !t_out) return fail(c, CRT_ERR_ARG, "crt_trace: null outputs");
    return trace_hook(c, org_dir, n, prim_out, t_out, nullptr);
}
int crt_trace_shadow(crt_ctx* c, const float* org_dir, int n, uint8_t* shadowed_out) {
    if (!shadowed_out) return fail(c, CRT_ERR_ARG, "crt_trace_shadow: null output");
    return trace_hook(c, org_dir, n, nullptr, nullptr, shadowed_out);
}

/* ---------------------------------------------------------------------------------------- */

int crt_blob_size(crt_ctx* c, size_t* bytes) {
    if (!c || !bytes) return CRT_ERR_ARG;
    if (!c->built) return fail(c, CRT_ERR_STATE, "crt_blob_size: no BVH");
    *bytes = c->blob_bytes;
    return CRT_OK;
}
int crt_blob_device_ptr(crt_ctx* c, void** p) {
    if (!c || !p) return CRT_ERR_ARG;
    if (!c->built) return fail(c, CRT_ERR_STATE, "crt_blob_device_ptr: no BVH");
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    *p = c->d_blob;
    return CRT_OK;
}
int crt_blob_import_device(crt_ctx* c, const void* dev_ptr, size_t bytes) {
    if (!c || !dev_ptr || bytes < sizeof(BlobHeader)) return fail(c, CRT_ERR_ARG, "crt_blob_import_device: bad arguments");
    HIPCHECK(c, hipSetDevice(c->device));
    cancel_async_build(c);
    { const int rc = drain_frames(c); if (rc) return rc; }
    BlobHeader h;
    HIPCHECK(c, hipMemcpy(&h, dev_ptr, sizeof h, hipMemcpyDeviceToHost));
    if (h.magic != kBlobMagic || h.version != kBlobVersion || h.total_bytes != bytes) return fail(c, CRT_ERR_ARG, "crt_blob_import_device: not a scene blob of this version/size");
    BlobHeader chk = h; layout_blob(chk);
    if (chk.total_bytes != h.total_bytes || chk.off_mats != h.off_mats || chk.off_qnodes != h.off_qnodes) return fail(c, CRT_ERR_ARG, "crt_blob_import_device: inconsistent header");
    if (dev_ptr != c->d_blob) {
        if (bytes > c->blob_cap || !c->d_blob) {
            if (c->d_blob) HIPCHECK(c, hipFree(c->d_blob));
            c->d_blob = nullptr;
            HIPCHECK(c, hipMalloc(&c->d_blob, bytes));
            c->blob_cap = bytes;
        }
        HIPCHECK(c, hipMemcpy(c->d_blob, dev_ptr, bytes, hipMemcpyDeviceToDevice));
    }
    /* a context that was given the same scene with crt_scene_set before the import (every rank of an animated
     * multi-GPU run) keeps its crt_prim array and material ids, so crt_animate_device / crt_scene_update /
     * crt_bvh_refit / crt_bvh_build work on it as on the rank that built the blob */
    c->blob_bytes = bytes; c->hdr = h; c->built = true; c->imported = c->prims.size() != (size_t)h.nprims;
    c->nonfinite_scene = (h.flags & kBlobNonFiniteShading) != 0u;
    c->have_ref = c->have_ref && dev_ptr == c->d_blob;
    int rc = ensure(c, c->d_arrive, c->arrive_cap, (size_t)h.nnodes + 1);
    if (rc) return rc;
    scene_abs_from_box(c);
    c->stats.nodes = (int)h.nnodes; c->stats.prims = (int)h.nprims; c->stats.depth = (int)h.depth; c->stats.scene_bytes = bytes;
    return size_overflow(c);
}

/* tile lists of every rank for the gather, resident on the device and cached per frame geometry */
static int tile_list(crt_ctx* c, const crt_params* p, int rank, uint32_t** d_ids, int* ntiles) {
    crt_params key{};          /* what the tile map depends on */
    key.width = p->width; key.height = p->height; key.tile_w = p->tile_w; key.tile_h = p->tile_h; key.nranks = p->nranks;
    if (c->tl_ids.empty() || memcmp(&key, &c->tl_key, sizeof key) != 0) {
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        for (uint32_t* q : c->tl_ids) if (q) hipFree(q);
        c->tl_ids.assign((size_t)p->nranks, nullptr); c->tl_n.assign((size_t)p->nranks, 0);
        c->tl_key = key;
    }
    if (!c->tl_ids[rank]) {
        crt_params q = *p; q.rank = rank;
        int nt = 0; int64_t np = 0;
        int rc = crt_tiles_owned(&q, &nt, &np, nullptr, 0);
        if (rc) return fail(c, rc, "tiles pack/unpack: bad params");
        std::vector<int32_t> ids(nt);
        crt_tiles_owned(&q, &nt, &np, ids.data(), nt);
        HIPCHECK(c, hipMalloc((void**)&c->tl_ids[rank], (size_t)(nt + 1) * 4));
        HIPCHECK(c, hipMemcpy(c->tl_ids[rank], ids.data(), (size_t)nt * 4, hipMemcpyHostToDevice));
        c->tl_n[rank] = nt;
    }
    *d_ids = c->tl_ids[rank]; *ntiles = c->tl_n[rank];
    return CRT_OK;
}

static int tiles_copy(crt_ctx* c, const crt_params* p, int src_rank, const void* src, void* dst, int to_packed, hipStream_t stream, bool sync) {
    if (!c || !p || !src || !dst) return fail(c, CRT_ERR_ARG, "tiles pack/unpack: null argument");
    if (p->nranks < 1 || src_rank < 0 || src_rank >= p->nranks) return fail(c, CRT_ERR_ARG, "tiles pack/unpack: bad rank");
    HIPCHECK(c, hipSetDevice(c->device));
    uint32_t* d_ids = nullptr; int nt = 0;
    int rc = tile_list(c, p, src_rank, &d_ids, &nt);
    if (rc) return rc;
    FrameParams fp{};
    fp.width = p->width; fp.height = p->height; fp.tile_w = p->tile_w; fp.tile_h = p->tile_h;
    fp.tiles_x = (p->width + p->tile_w - 1) / p->tile_w; fp.ntiles_owned = nt;
    launch_tiles_copy(fp, d_ids, nt, (const uint8_t*)src, (uint8_t*)dst, to_packed, stream ? stream : c->stream);
    HIPCHECK(c, hipGetLastError());
    /* packing is followed by a collective on another stream: finish it here.  Unpacks of several
     * ranks queue up on the context's stream; crt_sync() (or the next render) completes them. */
    if (sync) HIPCHECK(c, hipStreamSynchronize(stream ? stream : c->stream));
    return CRT_OK;
}
int crt_tiles_pack(crt_ctx* c, const crt_params* p, const void* dev_bgr, void* dev_packed) {
    return tiles_copy(c, p, p ? p->rank : 0, dev_bgr ? dev_bgr : (c ? c->fs[0].d_bgr : nullptr), dev_packed, 1, nullptr, true);
}
int crt_tiles_unpack(crt_ctx* c, const crt_params* p, int src_rank, const void* dev_packed, void* dev_bgr) {
    return tiles_copy(c, p, src_rank, dev_packed, dev_bgr, 0, nullptr, false);
}
/* the pack queued behind the frame a slot has in flight (or has just finished), on that slot's stream: complete when
 * crt_render_wait(slot) returns -- or, for device-side consumers, when crt_slot_stream's work up to here is */
int crt_tiles_pack_slot(crt_ctx* c, int slot, const crt_params* p, const void* dev_bgr, void* dev_packed) {
    if (!c || slot < 0 || slot >= CRT_MAX_FRAME_SLOTS) return fail(c, CRT_ERR_ARG, "crt_tiles_pack_slot: bad slot");
    FrameSlot& f = c->fs[slot];
    if (!f.stream) return fail(c, CRT_ERR_STATE, "crt_tiles_pack_slot: nothing was ever submitted on this slot");
    const int rc = tiles_copy(c, p, p ? p->rank : 0, dev_bgr ? dev_bgr : f.d_bgr, dev_packed, 1, f.stream, false);
    if (rc) return rc;
    if (f.in_flight) HIPCHECK(c, hipEventRecord(f.ev_done, f.stream));      /* the wait now covers the pack too */
    return CRT_OK;
}
int crt_slot_stream(crt_ctx* c, int slot, void** hip_stream) {
    if (!c || !hip_stream || slot < 0 || slot >= CRT_MAX_FRAME_SLOTS) return fail(c, CRT_ERR_ARG, "crt_slot_stream: bad arguments");
    HIPCHECK(c, hipSetDevice(c->device));
    const int rc = ensure_slot(c, slot);
    if (rc) return rc;
    *hip_stream = (void*)c->fs[slot].stream;
    return CRT_OK;
}

int crt_get_stats(crt_ctx* c, crt_stats* out) { if (!c || !out) return CRT_ERR_ARG; *out = c->stats; return CRT_OK; }
int crt_sync(crt_ctx* c) { if (!c) return CRT_ERR_ARG; HIPCHECK(c, hipSetDevice(c->device)); HIPCHECK(c, hipStreamSynchronize(c->stream)); return CRT_OK; }

int crt_host_bvh(const crt_prim* prims, int n, int mode, int32_t* nnodes, int32_t* depth, float* root_box6,
                 float* child_boxes, uint32_t* child_links, uint32_t* parents, uint32_t* slot_to_scene,
                 uint32_t* rank_of_scene, int cap_nodes) {
    if (n < 0 || (n > 0 && !prims) || !nnodes) return CRT_ERR_ARG;
    if (mode != CRT_BVH_WIDE && mode != CRT_BVH_REFERENCE) return CRT_ERR_ARG;
    RefTree ref; HostBVH bvh;
    int rc = build_reference_tree(prims, n, ref);
    if (rc && mode == CRT_BVH_REFERENCE) return rc;
    rc = mode == CRT_BVH_WIDE ? build_wide(prims, n, rc ? nullptr : &ref, bvh) : pack_reference(prims, n, ref, bvh);
    if (rc) return rc;
    *nnodes = (int32_t)bvh.node_count();
    if (depth) *depth = bvh.depth;
    if (root_box6) memcpy(root_box6, bvh.root_box, 24);
    if (!child_boxes || !child_links) return CRT_OK;
    if (cap_nodes < *nnodes) return CRT_ERR_ARG;
    if (mode == CRT_BVH_WIDE) export_nodes(bvh.nodes8, child_boxes, child_links, parents);
    else export_nodes(bvh.nodes, child_boxes, child_links, parents);
    if (slot_to_scene) memcpy(slot_to_scene, bvh.slot_to_scene.data(), (size_t)n * 4);
    if (rank_of_scene) memcpy(rank_of_scene, bvh.rank_of_scene.data(), (size_t)n * 4);
    return CRT_OK;
}

}  // extern "C"
