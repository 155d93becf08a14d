/* mgpu.cpp -- the multi-GPU host layer in C++ over RCCL (BASELINE north_star: "frames shard by image
 * tile across the GPUs of one node, the BVH and scene replicated via RCCL broadcast, the framebuffer gathered over xGMI").
 *
 * One process, one crt_ctx per GPU; frames are queued asynchronously, so one host thread drives every GPU (only the per-GPU
 * device rebuilds of crt_mgpu_animate run on threads of their own).  The path shards with no reduction:
 *   crt_mgpu_sync_scene   ncclBroadcast of rank 0's packed scene + BVH blob (one contiguous device buffer, layout.h) to
 *                         every other rank, which imports it (crt_blob_import_device); lights are copied host-side
 *   crt_mgpu_render       every rank renders the 32x32 tiles it owns (crt_tile_owner) into its own device frame and packs
 *                         them tile-major; ONE grouped exchange -- ncclSend from every rank r > 0, ncclRecv x (R-1) on rank 0,
 *                         each over its own xGMI link -- brings them to rank 0, which scatters them into the bottom-up frame
 *   crt_mgpu_submit/finish the same frame as a pipeline: K = 3 frames in flight per rank on the contexts' frame slots
 *                         (crt_render_submit: the host thread queues every rank's launches and never blocks on one), the gather of
 *                         a finished frame under the renders of later ones, 2K + 1 frame / 2K pack and gather buffers
 *   crt_mgpu_animate      the per-frame scene update with no traffic at all: every rank replays the deterministic animation step
 *                         on its own copy of the scene (crt_mgpu_scene_set) and refits / rebuilds locally
 * RCCL is loaded with dlopen on first use (RTLD_LOCAL: a process that also holds PyTorch's bundled RCCL keeps the two
 * apart) and not at all for one GPU unless CRT_MGPU_RCCL_ALWAYS asks for the 1-rank communicator.
 *
 * Rehearsal on one GPU: a device list with repeats (e.g. {0,0,0}) makes LOGICAL ranks -- RCCL refuses a communicator with
 * duplicate devices, so the two exchanges become device-to-device copies; everything else (contexts, threads, tile lists,
 * pack, unpack) is the code the real ranks run.
 */
#include "../../include/crt.h"
#include "internal.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <system_error>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
std::mutex g_rccl_mu;
Rccl g_rccl;

bool load_rccl(std::string& err) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.so) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* so = nullptr;
    for (const char* n : names) if ((so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!so) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    Rccl r; r.so = so;
#define SYM(F) r.F = (decltype(r.F))dlsym(so, "nccl" #F); if (!r.F) { err = "librccl lacks nccl" #F; dlclose(so); return false; }
    SYM(CommInitAll) SYM(CommDestroy) SYM(Broadcast) SYM(Send) SYM(Recv) SYM(GroupStart) SYM(GroupEnd) SYM(GetErrorString)
#undef SYM
    g_rccl = r;
    return true;
}

}  // namespace

/* one frame on its way through the pipeline of crt_mgpu_submit */
struct MgpuFrame {
    std::vector<crt_params> p;       /* per rank */
    std::vector<size_t> pb;          /* packed bytes per rank */
    int fb = 0, kb = 0, slot = 0;    /* rank 0's frame buffer (of 2K + 1), pack / gather buffer (of 2K), frame slot (the other ranks' frame buffer goes by it) */
};

struct crt_mgpu {
    int n = 0;
    std::vector<int> dev;
    std::vector<crt_ctx*> ctx;
    std::vector<hipStream_t> stream;           /* one per rank, on its device: the collectives run here */
    std::vector<ncclComm_t> comm;              /* empty: local-copy transport (logical ranks, or one GPU without RCCL) */
    bool rccl = false;
    bool self_exchange = false;                /* CRT_MGPU_RCCL_ALWAYS on one GPU: rank 0 sends its own tiles to itself through ncclSend / ncclRecv */
    std::vector<void*> blob_rx; std::vector<size_t> blob_rx_cap;      /* per rank > 0: where the broadcast blob lands */
    /* per rank, per buffer index: device frames (w*h*3, own tiles written), this rank's tiles tile-major, and -- on rank 0's
     * device -- where each source rank's tiles land */
    static constexpr int K = 3;                /* frames in flight per rank in crt_mgpu_submit (frame slots 1..K of every context) */
    static constexpr int NB = 2 * K;
    std::vector<std::vector<void*>> frame, packed, gathered;
    std::vector<std::vector<size_t>> frame_cap, packed_cap, gathered_cap;
    std::vector<crt_light> lights;
    bool have_lights = false;
    std::string err;
    double ms_render = 0, ms_exchange = 0, ms_unpack = 0;
    uint64_t seq = 0;
    std::deque<MgpuFrame> inflight, gathering;
    bool have_prims = false;                   /* crt_mgpu_scene_set: every rank holds the crt_prim array (crt_mgpu_animate needs it) */
};

namespace {

int mfail(crt_mgpu* m, int code, const std::string& msg) { if (m) m->err = msg; return code; }

#define MHIP(m, call)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return mfail((m), CRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define MNCCL(m, call)                                                                                    \
    do {                                                                                                  \
        ncclResult_t r_ = (call);                                                                         \
        if (r_ != ncclSuccess) return mfail((m), CRT_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

int ensure_dev(crt_mgpu* m, int device, void*& p, size_t& cap, size_t need) {
    if (need <= cap && p) return CRT_OK;
    MHIP(m, hipSetDevice(device));
    if (p) { MHIP(m, hipFree(p)); p = nullptr; cap = 0; }
    MHIP(m, hipMalloc(&p, need + need / 8 + 256));
    cap = need + need / 8 + 256;
    return CRT_OK;
}

size_t packed_bytes(const crt_params* p, int rank) {
    crt_params q = *p; q.rank = rank;
    int32_t nt = 0; int64_t np = 0;
    if (crt_tiles_owned(&q, &nt, &np, nullptr, 0) != CRT_OK) return 0;
    return (size_t)nt * (size_t)q.tile_w * (size_t)q.tile_h * 3;
}

int rank_fail(crt_mgpu* m, int rc, const char* what, int r) {
    return mfail(m, rc, std::string(what) + " on rank " + std::to_string(r) + ": " + crt_last_error(m->ctx[r]));
}

/* Which of a rank's device frames a frame in flight renders into.  Rank 0 keeps 2K + 1 (a frame stays there until its gather is
 * unpacked and the caller has seen it).  The other ranks only need their frame until the tile pack queued behind the render has
 * run: one per frame SLOT -- frames in flight never share a slot, whereas seq mod (2K + 1) mod K maps two consecutive frames to
 * one buffer whenever the sequence wraps. */
int frame_index(const MgpuFrame& f, int r) { return r == 0 ? f.fb : (f.slot > 0 ? f.slot - 1 : 0); }

/* per-rank parameters, packed sizes and the buffers frame (fb, kb) needs */
int prepare_frame(crt_mgpu* m, const crt_params* params, MgpuFrame& f) {
    const int R = m->n;
    const size_t fbytes = (size_t)params->width * (size_t)params->height * 3;
    f.p.assign((size_t)R, *params); f.pb.assign((size_t)R, 0);
    int rc;
    for (int r = 0; r < R; r++) {
        f.p[r].rank = r; f.p[r].nranks = R;
        f.pb[r] = packed_bytes(&f.p[r], r);
        const int fb = frame_index(f, r);
        if ((rc = ensure_dev(m, m->dev[r], m->frame[r][fb], m->frame_cap[r][fb], fbytes))) return rc;
        if (r > 0 || m->self_exchange) {
            if ((rc = ensure_dev(m, m->dev[r], m->packed[r][f.kb], m->packed_cap[r][f.kb], f.pb[r] ? f.pb[r] : 16))) return rc;
            if ((rc = ensure_dev(m, m->dev[0], m->gathered[r][f.kb], m->gathered_cap[r][f.kb], f.pb[r] ? f.pb[r] : 16))) return rc;
        }
    }
    return CRT_OK;
}
void* frame_of(crt_mgpu* m, const MgpuFrame& f, int r) { return m->frame[r][frame_index(f, r)]; }

/* queue the frame on every rank's frame slot, each with its tile pack behind it: no waiting */
int submit_frame(crt_mgpu* m, const crt_camera* cam, MgpuFrame& f) {
    for (int r = 0; r < m->n; r++) {
        int q = crt_render_submit(m->ctx[r], f.slot, cam, &f.p[r], frame_of(m, f, r));
        if (q == CRT_OK && (r > 0 || m->self_exchange) && f.pb[r]) q = crt_tiles_pack_slot(m->ctx[r], f.slot, &f.p[r], frame_of(m, f, r), m->packed[r][f.kb]);
        if (q) return rank_fail(m, q, "render", r);
    }
    return CRT_OK;
}
int wait_frame(crt_mgpu* m, MgpuFrame& f) {
    int first = CRT_OK, at = 0;
    for (int r = 0; r < m->n; r++) {        /* every rank is collected even after a failure: no slot stays busy */
        const int q = crt_render_wait(m->ctx[r], f.slot);
        if (q && !first) { first = q; at = r; }
    }
    return first ? rank_fail(m, first, "render", at) : CRT_OK;
}
/* the framebuffer gather: R - 1 point-to-point transfers into rank 0, each over its own link, as one group; queued on the
 * ranks' collective streams, not waited for */
int start_exchange(crt_mgpu* m, MgpuFrame& f) {
    const int R = m->n;
    const int first = m->self_exchange ? 0 : 1;
    if (R == 1 && !m->self_exchange) return CRT_OK;
    if (m->rccl) {
        MNCCL(m, g_rccl.GroupStart());
        for (int r = first; r < R; r++) {
            if (!f.pb[r]) continue;
            ncclResult_t q = g_rccl.Send(m->packed[r][f.kb], f.pb[r], ncclUint8, 0, m->comm[r], m->stream[r]);
            if (q == ncclSuccess) q = g_rccl.Recv(m->gathered[r][f.kb], f.pb[r], ncclUint8, r, m->comm[0], m->stream[0]);
            if (q != ncclSuccess) { g_rccl.GroupEnd(); return mfail(m, CRT_ERR_HIP, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(q)); }
        }
        MNCCL(m, g_rccl.GroupEnd());
    } else {
        MHIP(m, hipSetDevice(m->dev[0]));
        for (int r = first; r < R; r++)
            if (f.pb[r]) {
                if (m->dev[r] == m->dev[0]) MHIP(m, hipMemcpyAsync(m->gathered[r][f.kb], m->packed[r][f.kb], f.pb[r], hipMemcpyDeviceToDevice, m->stream[0]));
                else MHIP(m, hipMemcpyPeerAsync(m->gathered[r][f.kb], m->dev[0], m->packed[r][f.kb], m->dev[r], f.pb[r], m->stream[0]));
            }
    }
    return CRT_OK;
}
/* wait for the exchange and scatter the other ranks' tiles into rank 0's frame */
int finish_exchange(crt_mgpu* m, MgpuFrame& f) {
    const int R = m->n;
    const int first = m->self_exchange ? 0 : 1;
    if (R == 1 && !m->self_exchange) return CRT_OK;
    for (int r = 0; r < R; r++) { MHIP(m, hipSetDevice(m->dev[r])); MHIP(m, hipStreamSynchronize(m->stream[r])); }
    int rc;
    for (int r = first; r < R; r++) {
        if (!f.pb[r]) continue;
        if ((rc = crt_tiles_unpack(m->ctx[0], &f.p[0], r, m->gathered[r][f.kb], frame_of(m, f, 0)))) return rank_fail(m, rc, "crt_tiles_unpack", 0);
    }
    if ((rc = crt_sync(m->ctx[0]))) return rank_fail(m, rc, "crt_sync", 0);
    return CRT_OK;
}

/* oldest frame in flight: wait for its render + pack on every rank, start its gather; complete gathers older than `keep`
 * frames.  *done = rank 0's frame of the last frame completed here, if any */
int retire(crt_mgpu* m, size_t keep, void** done) {
    int rc;
    if (!m->inflight.empty()) {
        MgpuFrame f = std::move(m->inflight.front());
        m->inflight.pop_front();
        if ((rc = wait_frame(m, f))) return rc;
        if ((rc = start_exchange(m, f))) return rc;
        m->gathering.push_back(std::move(f));
    }
    while (m->gathering.size() > keep) {
        MgpuFrame f = std::move(m->gathering.front());
        m->gathering.pop_front();
        if ((rc = finish_exchange(m, f))) return rc;
        if (done) *done = frame_of(m, f, 0);
    }
    return CRT_OK;
}

}  // namespace

extern "C" {

crt_mgpu* crt_mgpu_create(const int* devices, int ndev, int flags, int* err) {
    auto bail = [&](int code) -> crt_mgpu* { if (err) *err = code; return nullptr; };
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible == 0) return bail(CRT_ERR_HIP);
    std::vector<int> dev;
    if (!devices || ndev <= 0) { for (int d = 0; d < visible; d++) dev.push_back(d); }
    else dev.assign(devices, devices + ndev);
    for (int d : dev) if (d < 0 || d >= visible) return bail(CRT_ERR_ARG);
    crt_mgpu* m = new (std::nothrow) crt_mgpu();
    if (!m) return bail(CRT_ERR_NOMEM);
    m->n = (int)dev.size(); m->dev = dev;
    const size_t n = dev.size();
    m->ctx.assign(n, nullptr); m->stream.assign(n, nullptr);
    m->blob_rx.assign(n, nullptr); m->blob_rx_cap.assign(n, 0);
    m->frame.assign(n, std::vector<void*>(crt_mgpu::NB + 1, nullptr)); m->frame_cap.assign(n, std::vector<size_t>(crt_mgpu::NB + 1, 0));
    m->packed.assign(n, std::vector<void*>(crt_mgpu::NB, nullptr)); m->packed_cap.assign(n, std::vector<size_t>(crt_mgpu::NB, 0));
    m->gathered.assign(n, std::vector<void*>(crt_mgpu::NB, nullptr)); m->gathered_cap.assign(n, std::vector<size_t>(crt_mgpu::NB, 0));
    for (size_t r = 0; r < n; r++) {
        int e = 0;
        m->ctx[r] = crt_create(dev[r], &e);
        if (!m->ctx[r] || hipSetDevice(dev[r]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking) != hipSuccess) {
            crt_mgpu_destroy(m);
            return bail(e ? e : CRT_ERR_HIP);
        }
    }
    const bool distinct = std::set<int>(dev.begin(), dev.end()).size() == n;
    if (distinct && (n > 1 || (flags & CRT_MGPU_RCCL_ALWAYS))) {
        std::string lerr;
        if (!load_rccl(lerr)) { fprintf(stderr, "crt_mgpu_create: %s\n", lerr.c_str()); crt_mgpu_destroy(m); return bail(CRT_ERR_HIP); }
        m->comm.assign(n, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), (int)n, dev.data());
        if (r != ncclSuccess) {
            fprintf(stderr, "crt_mgpu_create: ncclCommInitAll: %s\n", g_rccl.GetErrorString(r));
            m->comm.clear(); crt_mgpu_destroy(m);
            return bail(CRT_ERR_HIP);
        }
        m->rccl = true;
        m->self_exchange = n == 1;          /* rehearsal of the point-to-point calls where there is no second GPU */
    }
    if (err) *err = CRT_OK;
    return m;
}

void crt_mgpu_destroy(crt_mgpu* m) {
    if (!m) return;
    for (size_t r = 0; r < m->ctx.size(); r++) {
        if (r < m->dev.size()) (void)hipSetDevice(m->dev[r]);
        if (m->ctx[r]) for (int k = 0; k < CRT_MAX_FRAME_SLOTS; k++) (void)crt_render_wait(m->ctx[r], k);      /* frames still in flight */
        if (r < m->stream.size() && m->stream[r]) { (void)hipStreamSynchronize(m->stream[r]); }
    }
    for (size_t r = 0; r < m->ctx.size(); r++) {
        if (r < m->dev.size()) (void)hipSetDevice(m->dev[r]);
        if (m->rccl && r < m->comm.size() && m->comm[r]) g_rccl.CommDestroy(m->comm[r]);
        if (m->blob_rx[r]) (void)hipFree(m->blob_rx[r]);
        for (void* p : m->frame[r]) if (p) (void)hipFree(p);
        for (void* p : m->packed[r]) if (p) (void)hipFree(p);
    }
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    for (auto& v : m->gathered) for (void* p : v) if (p) (void)hipFree(p);      /* all on rank 0's device */
    for (size_t r = 0; r < m->ctx.size(); r++) {
        if (r < m->dev.size()) (void)hipSetDevice(m->dev[r]);
        if (r < m->stream.size() && m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
        if (m->ctx[r]) crt_destroy(m->ctx[r]);
    }
    delete m;
}

int crt_mgpu_size(const crt_mgpu* m) { return m ? m->n : 0; }
int crt_mgpu_uses_rccl(const crt_mgpu* m) { return m && m->rccl ? 1 : 0; }
crt_ctx* crt_mgpu_ctx(crt_mgpu* m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
const char* crt_mgpu_last_error(crt_mgpu* m) { return m ? m->err.c_str() : "null crt_mgpu"; }

int crt_mgpu_lights_set(crt_mgpu* m, const crt_light* lights, int n) {
    if (!m || n < 0 || n > CRT_MAX_LIGHTS || (n > 0 && !lights)) return mfail(m, CRT_ERR_ARG, "crt_mgpu_lights_set: 0..CRT_MAX_LIGHTS lights");
    m->lights.assign(lights, lights + n); m->have_lights = true;
    for (int r = 0; r < m->n; r++) {
        const int rc = crt_lights_set(m->ctx[r], lights, n);
        if (rc) return rank_fail(m, rc, "crt_lights_set", r);
    }
    return CRT_OK;
}

/* The scene on EVERY rank (one process holds the host array, so this is R uploads, no collective): what crt_mgpu_animate
 * needs, since the ranks then advance their own copy of the scene instead of receiving moved primitives.  Follow with
 * crt_bvh_build on crt_mgpu_ctx(m, 0) and crt_mgpu_sync_scene, as for a scene set on rank 0 only. */
int crt_mgpu_scene_set(crt_mgpu* m, const crt_prim* prims, int n) {
    if (!m) return CRT_ERR_ARG;
    int rc = crt_mgpu_finish(m, nullptr);
    if (rc) return rc;
    for (int r = 0; r < m->n; r++)
        if ((rc = crt_scene_set(m->ctx[r], prims, n))) return rank_fail(m, rc, "crt_scene_set", r);
    m->have_prims = true;
    return CRT_OK;
}

int crt_mgpu_sync_scene(crt_mgpu* m) {
    if (!m) return CRT_ERR_ARG;
    if (m->n == 1 && !m->rccl) return CRT_OK;
    size_t bytes = 0; void* src = nullptr;
    int rc = crt_blob_size(m->ctx[0], &bytes);
    if (rc == CRT_OK) rc = crt_blob_device_ptr(m->ctx[0], &src);       /* returns with rank 0's stream drained */
    if (rc) return mfail(m, rc, std::string("crt_mgpu_sync_scene: rank 0 has no built scene: ") + crt_last_error(m->ctx[0]));
    for (int r = 1; r < m->n; r++)
        if ((rc = ensure_dev(m, m->dev[r], m->blob_rx[r], m->blob_rx_cap[r], bytes))) return rc;
    if (m->rccl) {
        /* one broadcast, rank 0 in place: every rank's call inside one group, each on its own stream and with buffers on
         * its own device (the send buffer is read on the root only) */
        MNCCL(m, g_rccl.GroupStart());
        for (int r = 0; r < m->n; r++) {
            void* buf = r == 0 ? src : m->blob_rx[r];
            const ncclResult_t q = g_rccl.Broadcast(buf, buf, bytes, ncclUint8, 0, m->comm[r], m->stream[r]);
            if (q != ncclSuccess) { g_rccl.GroupEnd(); return mfail(m, CRT_ERR_HIP, std::string("ncclBroadcast: ") + g_rccl.GetErrorString(q)); }
        }
        MNCCL(m, g_rccl.GroupEnd());
    } else {
        for (int r = 1; r < m->n; r++) {
            MHIP(m, hipSetDevice(m->dev[r]));
            if (m->dev[r] == m->dev[0]) MHIP(m, hipMemcpyAsync(m->blob_rx[r], src, bytes, hipMemcpyDeviceToDevice, m->stream[r]));
            else MHIP(m, hipMemcpyPeerAsync(m->blob_rx[r], m->dev[r], src, m->dev[0], bytes, m->stream[r]));
        }
    }
    for (int r = 0; r < m->n; r++) { MHIP(m, hipSetDevice(m->dev[r])); MHIP(m, hipStreamSynchronize(m->stream[r])); }
    for (int r = 1; r < m->n; r++) {
        if ((rc = crt_blob_import_device(m->ctx[r], m->blob_rx[r], bytes))) return rank_fail(m, rc, "crt_blob_import_device", r);
    }
    return CRT_OK;
}

/* One step of the reference's frame loop for the scene (main.cu:534-541, 545-573) on every rank, with NO exchange: the
 * animation step is deterministic, so every rank advances its own copy (crt_animate_device), then refits -- or, rebuild != 0
 * (the "every 20th frame" of main.cu:534), rebuilds on its own device.  The refits are queued on all GPUs before any is
 * waited for; the rebuilds run on one host thread per GPU.  Needs crt_mgpu_scene_set. */
int crt_mgpu_animate(crt_mgpu* m, int rebuild, int grow_only) {
    if (!m) return CRT_ERR_ARG;
    if (!m->have_prims && m->n > 1) return mfail(m, CRT_ERR_STATE, "crt_mgpu_animate: only rank 0 holds the scene (crt_mgpu_scene_set hands it to every rank)");
    int rc;
    for (int r = 0; r < m->n; r++)
        if ((rc = crt_animate_device(m->ctx[r]))) return rank_fail(m, rc, "crt_animate_device", r);
    if (!rebuild) {
        for (int r = 0; r < m->n; r++)
            if ((rc = crt::ctx_refit_enqueue(m->ctx[r], grow_only))) return rank_fail(m, rc, "crt_bvh_refit", r);
        int first = CRT_OK, at = 0;
        for (int r = 0; r < m->n; r++) { rc = crt::ctx_refit_finish(m->ctx[r]); if (rc && !first) { first = rc; at = r; } }
        return first ? rank_fail(m, first, "crt_bvh_refit", at) : CRT_OK;
    }
    std::vector<int> rcs((size_t)m->n, CRT_OK);
    auto work = [&](int r) { rcs[r] = crt_bvh_build(m->ctx[r], CRT_BVH_WIDE_DEVICE); };
    std::vector<std::thread> th;
    try {
        for (int r = 1; r < m->n; r++) th.emplace_back(work, r);
    } catch (const std::system_error&) {          /* no thread to be had: the ranks not started build one after the other below */
    }
    work(0);
    for (auto& t : th) t.join();
    for (int r = (int)th.size() + 1; r < m->n; r++) work(r);
    for (int r = 0; r < m->n; r++) if (rcs[r]) return rank_fail(m, rcs[r], "crt_bvh_build", r);
    return CRT_OK;
}

/* Throughput mode: the frame is queued on every rank (render + tile pack on a frame slot's stream) and the call returns
 * without waiting for it.  K = 3 frames are in flight per rank; when all slots are busy the oldest frame is waited for and
 * its gather started on the collective streams, and gathers more than K frames old are completed and unpacked -- so the
 * launches of one frame fill the tails of another's persistent launches and every gather runs under later renders.
 * *dev_frame_done = rank 0's assembled frame of the last frame completed by this call (valid until the next call), or NULL. */
int crt_mgpu_submit(crt_mgpu* m, const crt_camera* cam, const crt_params* params, void** dev_frame_done) {
    if (!m || !cam || !params) return mfail(m, CRT_ERR_ARG, "crt_mgpu_submit: null argument");
    if (dev_frame_done) *dev_frame_done = nullptr;
    int rc;
    if (m->inflight.size() >= (size_t)crt_mgpu::K && (rc = retire(m, (size_t)crt_mgpu::K, dev_frame_done))) return rc;
    MgpuFrame f;
    f.fb = (int)(m->seq % (crt_mgpu::NB + 1)); f.kb = (int)(m->seq % crt_mgpu::NB); f.slot = 1 + (int)(m->seq % crt_mgpu::K);
    if ((rc = prepare_frame(m, params, f))) return rc;
    for (auto& p : f.p) p.timing = 1;
    if ((rc = submit_frame(m, cam, f))) {
        /* the ranks before the failing one hold the frame on this slot: collect it, or the slot stays busy for good */
        const std::string why = m->err;
        (void)wait_frame(m, f);
        m->err = why;
        return rc;
    }
    m->inflight.push_back(std::move(f));
    m->seq++;
    return CRT_OK;
}

/* completes every frame crt_mgpu_submit still has in flight; *dev_frame_last = the last one on rank 0's device */
int crt_mgpu_finish(crt_mgpu* m, void** dev_frame_last) {
    if (!m) return CRT_ERR_ARG;
    if (dev_frame_last) *dev_frame_last = nullptr;
    while (!m->inflight.empty() || !m->gathering.empty()) {
        const int rc = retire(m, m->inflight.empty() ? 0 : (size_t)crt_mgpu::K, dev_frame_last);
        if (rc) return rc;
    }
    return CRT_OK;
}

int crt_mgpu_render_device(crt_mgpu* m, const crt_camera* cam, const crt_params* params, void** dev_frame_rank0) {
    if (!m || !cam || !params) return mfail(m, CRT_ERR_ARG, "crt_mgpu_render: null argument");
    int rc;
    if ((rc = crt_mgpu_finish(m, nullptr))) return rc;          /* frames of the throughput mode use the same buffers */
    MgpuFrame f;
    f.fb = 0; f.kb = 0; f.slot = 0;
    if ((rc = prepare_frame(m, params, f))) return rc;
    /* every rank renders and packs its tiles: the submissions return at once, the GPUs work side by side */
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = submit_frame(m, cam, f))) { const std::string why = m->err; (void)wait_frame(m, f); m->err = why; return rc; }
    if ((rc = wait_frame(m, f))) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    if ((rc = start_exchange(m, f))) return rc;
    for (int r = 0; r < m->n; r++) { MHIP(m, hipSetDevice(m->dev[r])); MHIP(m, hipStreamSynchronize(m->stream[r])); }
    const auto t2 = std::chrono::steady_clock::now();
    if ((rc = finish_exchange(m, f))) return rc;
    const auto t3 = std::chrono::steady_clock::now();
    m->ms_render = std::chrono::duration<double, std::milli>(t1 - t0).count();
    m->ms_exchange = std::chrono::duration<double, std::milli>(t2 - t1).count();
    m->ms_unpack = std::chrono::duration<double, std::milli>(t3 - t2).count();
    if (dev_frame_rank0) *dev_frame_rank0 = frame_of(m, f, 0);
    return CRT_OK;
}

int crt_mgpu_render(crt_mgpu* m, const crt_camera* cam, const crt_params* params, unsigned char* bgr_out_host) {
    if (!m || !cam || !params || !bgr_out_host) return mfail(m, CRT_ERR_ARG, "crt_mgpu_render: null argument");
    if (m->n == 1 && !m->self_exchange) {            /* one GPU: the single-context path with its pinned staging copy */
        int rc = crt_mgpu_finish(m, nullptr);
        if (rc) return rc;
        crt_params p = *params; p.rank = 0; p.nranks = 1;
        rc = crt_render(m->ctx[0], cam, &p, bgr_out_host);
        return rc ? mfail(m, rc, crt_last_error(m->ctx[0])) : CRT_OK;
    }
    void* f = nullptr;
    const int rc = crt_mgpu_render_device(m, cam, params, &f);
    if (rc) return rc;
    MHIP(m, hipSetDevice(m->dev[0]));
    MHIP(m, hipMemcpy(bgr_out_host, f, (size_t)params->width * (size_t)params->height * 3, hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_mgpu_get_times(crt_mgpu* m, double* ms_render, double* ms_exchange, double* ms_unpack) {
    if (!m) return CRT_ERR_ARG;
    if (ms_render) *ms_render = m->ms_render;
    if (ms_exchange) *ms_exchange = m->ms_exchange;
    if (ms_unpack) *ms_unpack = m->ms_unpack;
    return CRT_OK;
}

}  // extern "C"
