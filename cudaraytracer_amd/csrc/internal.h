/* internal.h -- what other translation units of libcrt_amd.so may see of a crt_ctx (defined in crt_api.cpp). */
#ifndef CRT_INTERNAL_H
#define CRT_INTERNAL_H
#include "kernels.h"

namespace crt {

/* the part of a context a traversal launch needs; filled by ctx_trace_view() */
struct TraceView {
    SceneView sv;
    hipStream_t stream;
    int device;
    bool built, wide;
    float scene_abs;
    float origin_bound;                      /* BlobHeader::origin_bound */
    uint32_t* ovf; int32_t ovf_depth;        /* sized by crt_bvh_build / crt_blob_import_device, never by a trace call */
    unsigned long long* counters;
    uint32_t* status;                        /* device error word */
    uint32_t* h_status;                      /* pinned host word, allocated by crt_create */
    int grid_blocks;                         /* persistent grid in 256-thread blocks */
    int block;                               /* threads per block of k_trace8 (256 / 128 / 64) */
};
int ctx_trace_view(crt_ctx* c, TraceView* out);          /* CRT_OK, or CRT_ERR_ARG for a null context */
int ctx_fail(crt_ctx* c, int code, const char* msg);
/* crt_bvh_refit as enqueue + wait (mgpu.cpp queues the refit on every GPU before it waits for any) */
int ctx_refit_enqueue(crt_ctx* c, int grow_only);
int ctx_refit_finish(crt_ctx* c);     /* records msg for crt_last_error and returns code */

/* noalloc.hip: address of this device's copy of the scratch array, looked up once per device by crt_create
 * (with that device current); null + *err set if the lookup failed */
void* noalloc_scratch_for_device(int device, hipError_t* err);

}  // namespace crt
#endif
