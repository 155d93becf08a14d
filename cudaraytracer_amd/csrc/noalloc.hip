/* noalloc.hip -- crt_trace_noalloc: the nearest-hit parity hook (crt_trace) for callers that cannot allocate.
 *
 * crt_trace allocates its working buffers per call (packed ray queue, hit slots, queue / fetch counters:
 * hipMalloc + hipFree, crt_api.cpp trace_hook).  Code under HIP graph capture, tight serving loops and harnesses
 * that forbid hipMalloc cannot call that.  This entry point runs the SAME kernels (k_trace8 / k_trace,
 * k_slots_to_scene) with those buffers in a file-scope device array and calls no hipMalloc, hipFree,
 * hipEventCreate or any other allocating routine: the caller passes device pointers for the rays and the results.
 *
 * The array is one per GPU (each device that loads this code object holds its own copy, at its own address, until
 * the process ends: kScratchBytes = 36 MiB + 4 KiB).  In host code the array's NAME is only a handle for
 * hipGetSymbolAddress -- it is not a device address -- so the address is looked up once per device
 * (noalloc_scratch_for_device, called by crt_create with that device current) and only that pointer is ever
 * handed to kernels and memsets.
 *
 * Every call on a GPU shares that GPU's array and finds the previous call's contents in it: the counters are
 * cleared per call, the ray queue and the hit slots are rewritten for [0, n) before they are read.  Calls that
 * overlap on one GPU (two contexts on one device, two host threads) race on it: serialise them.
 */
#include "../../include/crt.h"
#include "internal.h"
#include <mutex>

namespace crt {

constexpr size_t kNoAllocMax = (size_t)CRT_NOALLOC_MAX_RAYS;
constexpr size_t kOffRayO = 0;                                        /* float4[kNoAllocMax] */
constexpr size_t kOffRayD = kOffRayO + kNoAllocMax * sizeof(float4);  /* float4[kNoAllocMax] */
constexpr size_t kOffSlot = kOffRayD + kNoAllocMax * sizeof(float4);  /* int32[kNoAllocMax]: hit slots before the slot -> scene index map */
constexpr size_t kOffCtr = kOffSlot + kNoAllocMax * sizeof(int32_t);  /* [0] ray count, [64 ..) 8 fetch counters kFetchStride words apart */
constexpr size_t kCtrWords = 64 + 8 * kFetchStride;
constexpr size_t kScratchBytes = kOffCtr + 4096;
static_assert(kCtrWords * 4 <= 4096, "counter block");
static_assert(kOffRayD % 256 == 0 && kOffSlot % 256 == 0 && kOffCtr % 256 == 0, "sections stay 256-byte aligned");

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_mu;
static void* g_ptr[kMaxDevices];          /* per device: where THAT device's copy of `scratch` lives */
static bool g_done[kMaxDevices];

void* noalloc_scratch_for_device(int device, hipError_t* err) {
    if (err) *err = hipSuccess;
    if (device < 0 || device >= kMaxDevices) { if (err) *err = hipErrorInvalidDevice; return nullptr; }
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_done[device]) {
        /* the caller has made `device` current: the lookup returns the address on the current device */
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) { if (err) *err = e; return nullptr; }
        g_ptr[device] = p;
        g_done[device] = true;
    }
    return g_ptr[device];
}

/* crt_trace's k_pack_rays plus the queue length (thread 0), so that no host word has to be copied in */
__global__ __launch_bounds__(256) void k_noalloc_pack(const float* od, uint32_t n, float4* rayO, float4* rayD, uint32_t* count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *count = n;
    if (i >= n) return;
    rayO[i] = make_float4(od[6 * i], od[6 * i + 1], od[6 * i + 2], __uint_as_float(i));
    rayD[i] = make_float4(od[6 * i + 3], od[6 * i + 4], od[6 * i + 5], 0.f);
}

}  // namespace crt

using namespace crt;

#define NA_HIP(c, call)                                                                                       \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) return ctx_fail((c), CRT_ERR_HIP, (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

extern "C" int crt_trace_noalloc(crt_ctx* c, const float* dev_org_dir, int n, int32_t* dev_prim_out, float* dev_t_out) {
    TraceView v;
    if (ctx_trace_view(c, &v) != CRT_OK) return CRT_ERR_ARG;
    if (n < 0 || (n > 0 && (!dev_org_dir || !dev_prim_out || !dev_t_out))) return ctx_fail(c, CRT_ERR_ARG, "crt_trace_noalloc: bad arguments");
    if ((size_t)n > kNoAllocMax)
        return ctx_fail(c, CRT_ERR_ARG, "crt_trace_noalloc: n exceeds CRT_NOALLOC_MAX_RAYS (1048576): the scratch array is fixed; split the batch or use crt_trace");
    if (!v.built) return ctx_fail(c, CRT_ERR_STATE, "crt_trace_noalloc: no BVH");
    if (n == 0) return CRT_OK;
    NA_HIP(c, hipSetDevice(v.device));
    hipError_t se = hipSuccess;
    unsigned char* base = (unsigned char*)noalloc_scratch_for_device(v.device, &se);      /* cached by crt_create: a table read here */
    if (!base) return ctx_fail(c, CRT_ERR_HIP, (std::string("crt_trace_noalloc: hipGetSymbolAddress(scratch): ") + hipGetErrorString(se)).c_str());
    float4* rayO = (float4*)(base + kOffRayO);
    float4* rayD = (float4*)(base + kOffRayD);
    int32_t* slot = (int32_t*)(base + kOffSlot);
    uint32_t* ctr = (uint32_t*)(base + kOffCtr);

    NA_HIP(c, hipMemsetAsync(ctr, 0, 4096, v.stream));                  /* the last call's fetch counters and used-up mask */
    hipLaunchKernelGGL(k_noalloc_pack, dim3(((uint32_t)n + 255u) / 256u), dim3(256), 0, v.stream, dev_org_dir, (uint32_t)n, rayO, rayD, ctr);
    TraceArgs ta{};
    ta.sv = v.sv; ta.rayO = rayO; ta.rayD = rayD; ta.count = ctr; ta.next = ctr + 64;
    ta.hitSlot = slot; ta.hitT = dev_t_out; ta.shflag = nullptr; ta.scene_abs = v.scene_abs; ta.origin_bound = v.origin_bound;
    ta.ovf = v.ovf; ta.ovf_depth = v.ovf_depth; ta.counters = v.counters; ta.status = v.status;
    ta.block = v.block;
    launch_trace(ta, TRACE_NEAREST, false, v.wide, v.grid_blocks, v.stream);
    launch_slots_to_scene(slot, v.sv.sidx, dev_prim_out, (uint32_t)n, v.stream);
    NA_HIP(c, hipGetLastError());
    NA_HIP(c, hipMemcpyAsync(v.h_status, v.status, 4, hipMemcpyDeviceToHost, v.stream));       /* pinned word: no staging */
    NA_HIP(c, hipStreamSynchronize(v.stream));
    if (*v.h_status) {
        NA_HIP(c, hipMemsetAsync(v.status, 0, 4, v.stream));
        NA_HIP(c, hipStreamSynchronize(v.stream));
        return ctx_fail(c, CRT_ERR_STATE, "traversal stack overflow: tree deeper than the sized overflow area");
    }
    return CRT_OK;
}
