/* mgpu.cpp -- the multi-GPU host layer in C++ over RCCL (BASELINE north_star: "frames shard by image
 * tile across the GPUs of one node, the BVH and scene replicated via RCCL broadcast, the framebuffer gathered over xGMI").
 *
 * One process, one crt_ctx per GPU, one host thread per GPU while a frame renders.  The path shards with no reduction:
 *   crt_mgpu_sync_scene   ncclBroadcast of rank 0's packed scene + BVH blob (one contiguous device buffer, layout.h) to
 *                         every other rank, which imports it (crt_blob_import_device); lights are copied host-side
 *   crt_mgpu_render       every rank renders the 32x32 tiles it owns (crt_tile_owner) into its own device frame and packs
 *                         them tile-major; ONE grouped exchange -- ncclSend from every rank r > 0, ncclRecv x (R-1) on rank 0,
 *                         each over its own xGMI link -- brings them to rank 0, which scatters them into the bottom-up frame
 * RCCL is loaded with dlopen on first use (RTLD_LOCAL: a process that also holds PyTorch's bundled RCCL keeps the two
 * apart) and not at all for one GPU unless CRT_MGPU_RCCL_ALWAYS asks for the 1-rank communicator.
 *
 * Rehearsal on one GPU: a device list with repeats (e.g. {0,0,0}) makes LOGICAL ranks -- RCCL refuses a communicator with
 * duplicate devices, so the two exchanges become device-to-device copies; everything else (contexts, threads, tile lists,
 * pack, unpack) is the code the real ranks run.
 */
#include "../../include/crt.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
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

struct crt_mgpu {
    int n = 0;
    std::vector<int> dev;
    std::vector<crt_ctx*> ctx;
    std::vector<hipStream_t> stream;           /* one per rank, on its device: the collectives run here */
    std::vector<ncclComm_t> comm;              /* empty: local-copy transport (logical ranks, or one GPU without RCCL) */
    bool rccl = false;
    std::vector<void*> blob_rx; std::vector<size_t> blob_rx_cap;      /* per rank > 0: where the broadcast blob lands */
    std::vector<void*> frame; std::vector<size_t> frame_cap;          /* per rank: device frame (w*h*3), own tiles written */
    std::vector<void*> packed; std::vector<size_t> packed_cap;        /* per rank: its tiles, tile-major */
    std::vector<void*> gathered; std::vector<size_t> gathered_cap;    /* on rank 0's device, per source rank > 0 */
    std::vector<crt_light> lights;
    bool have_lights = false;
    std::string err;
    double ms_render = 0, ms_exchange = 0, ms_unpack = 0;
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
    crt_mgpu* m = new crt_mgpu();
    m->n = (int)dev.size(); m->dev = dev;
    const size_t n = dev.size();
    m->ctx.assign(n, nullptr); m->stream.assign(n, nullptr);
    m->blob_rx.assign(n, nullptr); m->blob_rx_cap.assign(n, 0); m->frame.assign(n, nullptr); m->frame_cap.assign(n, 0);
    m->packed.assign(n, nullptr); m->packed_cap.assign(n, 0); m->gathered.assign(n, nullptr); m->gathered_cap.assign(n, 0);
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
    }
    if (err) *err = CRT_OK;
    return m;
}

void crt_mgpu_destroy(crt_mgpu* m) {
    if (!m) return;
    for (size_t r = 0; r < m->ctx.size(); r++) {
        if (r < m->dev.size()) (void)hipSetDevice(m->dev[r]);
        if (r < m->stream.size() && m->stream[r]) { (void)hipStreamSynchronize(m->stream[r]); }
        if (m->rccl && r < m->comm.size() && m->comm[r]) g_rccl.CommDestroy(m->comm[r]);
        void* bufs[] = {m->blob_rx[r], m->frame[r], m->packed[r], m->gathered[r]};
        for (void* p : bufs) if (p) (void)hipFree(p);
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
        if (rc) return mfail(m, rc, std::string("crt_lights_set on rank ") + std::to_string(r) + ": " + crt_last_error(m->ctx[r]));
    }
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
        /* one broadcast, rank 0 in place: every rank's call inside one group, each on its own stream */
        MNCCL(m, g_rccl.GroupStart());
        for (int r = 0; r < m->n; r++) {
            const ncclResult_t q = g_rccl.Broadcast(src, r == 0 ? src : m->blob_rx[r], bytes, ncclUint8, 0, m->comm[r], m->stream[r]);
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
        if ((rc = crt_blob_import_device(m->ctx[r], m->blob_rx[r], bytes)))
            return mfail(m, rc, std::string("crt_blob_import_device on rank ") + std::to_string(r) + ": " + crt_last_error(m->ctx[r]));
    }
    return CRT_OK;
}

int crt_mgpu_render_device(crt_mgpu* m, const crt_camera* cam, const crt_params* params, void** dev_frame_rank0) {
    if (!m || !cam || !params) return mfail(m, CRT_ERR_ARG, "crt_mgpu_render: null argument");
    const int R = m->n;
    const size_t fbytes = (size_t)params->width * (size_t)params->height * 3;
    std::vector<crt_params> p((size_t)R, *params);
    std::vector<size_t> pb((size_t)R, 0);
    int rc;
    for (int r = 0; r < R; r++) {
        p[r].rank = r; p[r].nranks = R;
        pb[r] = packed_bytes(&p[r], r);
        if ((rc = ensure_dev(m, m->dev[r], m->frame[r], m->frame_cap[r], fbytes))) return rc;
        if (r > 0) {
            if ((rc = ensure_dev(m, m->dev[r], m->packed[r], m->packed_cap[r], pb[r] ? pb[r] : 16))) return rc;
            if ((rc = ensure_dev(m, m->dev[0], m->gathered[r], m->gathered_cap[r], pb[r] ? pb[r] : 16))) return rc;
        }
    }
    /* every rank renders and packs its tiles: one host thread per GPU (crt_render_device returns with the frame complete) */
    std::vector<int> rcs((size_t)R, CRT_OK);
    auto work = [&](int r) {
        int q = crt_render_device(m->ctx[r], cam, &p[r], m->frame[r]);
        if (q == CRT_OK && r > 0 && pb[r]) q = crt_tiles_pack(m->ctx[r], &p[r], m->frame[r], m->packed[r]);
        rcs[r] = q;
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (R == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int r = 1; r < R; r++) th.emplace_back(work, r);
        work(0);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < R; r++)
        if (rcs[r]) return mfail(m, rcs[r], std::string("render on rank ") + std::to_string(r) + ": " + crt_last_error(m->ctx[r]));
    const auto t1 = std::chrono::steady_clock::now();
    /* the framebuffer gather: R - 1 point-to-point transfers into rank 0, each over its own link, as one group */
    if (R > 1) {
        if (m->rccl) {
            MNCCL(m, g_rccl.GroupStart());
            for (int r = 1; r < R; r++) {
                if (!pb[r]) continue;
                ncclResult_t q = g_rccl.Send(m->packed[r], pb[r], ncclUint8, 0, m->comm[r], m->stream[r]);
                if (q == ncclSuccess) q = g_rccl.Recv(m->gathered[r], pb[r], ncclUint8, r, m->comm[0], m->stream[0]);
                if (q != ncclSuccess) { g_rccl.GroupEnd(); return mfail(m, CRT_ERR_HIP, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(q)); }
            }
            MNCCL(m, g_rccl.GroupEnd());
            for (int r = 0; r < R; r++) { MHIP(m, hipSetDevice(m->dev[r])); MHIP(m, hipStreamSynchronize(m->stream[r])); }
        } else {
            MHIP(m, hipSetDevice(m->dev[0]));
            for (int r = 1; r < R; r++)
                if (pb[r]) {
                    if (m->dev[r] == m->dev[0]) MHIP(m, hipMemcpyAsync(m->gathered[r], m->packed[r], pb[r], hipMemcpyDeviceToDevice, m->stream[0]));
                    else MHIP(m, hipMemcpyPeerAsync(m->gathered[r], m->dev[0], m->packed[r], m->dev[r], pb[r], m->stream[0]));
                }
            MHIP(m, hipStreamSynchronize(m->stream[0]));
        }
    }
    const auto t2 = std::chrono::steady_clock::now();
    for (int r = 1; r < R; r++) {
        if (!pb[r]) continue;
        if ((rc = crt_tiles_unpack(m->ctx[0], &p[0], r, m->gathered[r], m->frame[0])))
            return mfail(m, rc, std::string("crt_tiles_unpack: ") + crt_last_error(m->ctx[0]));
    }
    if ((rc = crt_sync(m->ctx[0]))) return mfail(m, rc, crt_last_error(m->ctx[0]));
    const auto t3 = std::chrono::steady_clock::now();
    m->ms_render = std::chrono::duration<double, std::milli>(t1 - t0).count();
    m->ms_exchange = std::chrono::duration<double, std::milli>(t2 - t1).count();
    m->ms_unpack = std::chrono::duration<double, std::milli>(t3 - t2).count();
    if (dev_frame_rank0) *dev_frame_rank0 = m->frame[0];
    return CRT_OK;
}

int crt_mgpu_render(crt_mgpu* m, const crt_camera* cam, const crt_params* params, unsigned char* bgr_out_host) {
    if (!m || !bgr_out_host) return mfail(m, CRT_ERR_ARG, "crt_mgpu_render: null output buffer");
    if (m->n == 1) {            /* one GPU: the single-context path with its pinned staging copy */
        crt_params p = *params; p.rank = 0; p.nranks = 1;
        const int rc = crt_render(m->ctx[0], cam, &p, bgr_out_host);
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
