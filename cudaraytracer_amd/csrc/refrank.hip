/* refrank.hip -- the reference's leaf order, computed on the device ("... the reference's
 * centroid-mean split as the parity mode").
 *
 * Exact-t ties resolve to the primitive the reference's DFS meets first (strict '<', Ray.cu:199,216), i.e. to the lower
 * rank in the leaf order of ITS tree: newBuild + constructTree (main.cu:33-136, bvh.cu:12-273) -- split on the node's
 * longest axis at `midpoint` (root: box centre; child: mean of the child's centroids, accumulated in object order),
 * stable partition `centroid <= midpoint -> left`, leaf at <= 64 objects.  The device-built tree (lbvh.hip) is a
 * different tree, so CRT_BVH_WIDE_DEVICE used to break ties by scene index; this pass replays the reference's partition
 * level by level and hands the same ranks to the device build -- bit for bit what bvh_build.cpp's build_reference_tree
 * produces on the host, including the order-dependent float sums:
 *
 *   host        the root: box = the scene box lbvh.hip has already reduced, midpoint = box centre (main.cu:108-131)
 *   per level   k_flags      flag[pos] = centroid[axis(seg)] <= mid(seg) for positions of active (> 64 objects) segments
 *               rocPRIM      exclusive scan of the flags
 *               k_scatter    stable partition inside every segment; centroids (one array per axis) and bounds move along
 *               k_children   one block per child that is still > 64 objects.  Wave 0: the three centroid sums as
 *                            sequential float additions in object order -- lanes 0-2 walk the x, y, z arrays, which the
 *                            wave stages through LDS in 1024-element tiles (double buffered, 128-bit LDS reads).  Waves
 *                            1-2: the child's box (min/max are exact in any order).  Then the child's longest axis
 *                            (strict comparisons; a tie leaves axis 0 / midpoint 0, bvh.cu:205-253), midpoint = sum / count
 *   host        reads the number of children that still hold > 64 objects; stops at 0; more than 257 levels is the
 *               reference's runaway recursion -> CRT_ERR_BUILD (the caller falls back to scene-index ranks)
 *
 * The sequential sums bound the time: the root's two children are ~n/2-element chains of dependent v_add_f32; the levels
 * below halve.
 */
#include <algorithm>
#include <cstring>
#include <string>
#include "kernels.h"
#include <rocprim/device/device_scan.hpp>

namespace crt {

namespace {

constexpr uint32_t kNone = 0xffffffffu;
constexpr int kLeafMax = 64;                      /* bvh.cu:14 */
constexpr float kFmax = 3.402823466e+38f;
constexpr float kFminPos = 1.17549435e-38f;       /* numeric_limits<float>::min(): the reference's max-init (main.cu:35) */

struct Seg { uint32_t start, count, axis; float mid; };

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

struct RScratch {
    float* cen[2][3];                             /* centroids, one array per axis: the sum lanes read theirs contiguously */
    float4 *lo[2], *hi[2];
    uint32_t *order[2], *segid[2], *flag, *scan;
    Seg *segs[2], *kids;
    uint32_t* slotmap;      /* child index of this level -> segment index of the next, or none */
    uint32_t* ctr;          /* [0] next level's segment count */
    void* scan_tmp; size_t scan_tmp_bytes;
};

size_t scan_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, p, p, 0u, (size_t)n, rocprim::plus<uint32_t>(), (hipStream_t)0);
    return bytes;
}

constexpr int kTile = 1024;     /* elements staged per step and axis */

RScratch carve(void* base, uint32_t n, size_t* total) {
    RScratch S{};
    size_t off = 0;
    char* b = (char*)base;
    auto take = [&](size_t bytes) { void* p = b ? b + off : nullptr; off += up256(bytes); return p; };
    const size_t nseg = (size_t)n / 32 + 64;              /* active segments hold > 64 objects each; children twice as many */
    for (int k = 0; k < 2; k++) {
        for (int a = 0; a < 3; a++) S.cen[k][a] = (float*)take((size_t)n * 4 + kTile * 4);     /* tiles are read whole */
        S.lo[k] = (float4*)take((size_t)n * 16); S.hi[k] = (float4*)take((size_t)n * 16);
        S.order[k] = (uint32_t*)take((size_t)n * 4); S.segid[k] = (uint32_t*)take((size_t)n * 4);
        S.segs[k] = (Seg*)take(nseg * sizeof(Seg));
    }
    S.flag = (uint32_t*)take((size_t)n * 4); S.scan = (uint32_t*)take((size_t)n * 4 + 4);
    S.kids = (Seg*)take(2 * nseg * sizeof(Seg));
    S.slotmap = (uint32_t*)take(2 * nseg * 4);
    S.ctr = (uint32_t*)take(256);
    S.scan_tmp_bytes = scan_temp_bytes(n);
    S.scan_tmp = take(S.scan_tmp_bytes + 256);
    if (total) *total = off;
    return S;
}

/* centroid and bounds exactly as bvh_build.cpp centroid_of / grow (bvh.cu:61-66,70-129) */
__global__ __launch_bounds__(256) void k_rr_init(const crt_prim* prims, uint32_t n, float* cx, float* cy, float* cz, float4* lo, float4* hi,
                                                 uint32_t* order, uint32_t* segid, uint32_t root_slot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const crt_prim& o = prims[i];
    float c[3], l[3], h[3];
    if (o.type == 0) {
        for (int a = 0; a < 3; a++) { c[a] = o.position[a]; l[a] = o.position[a] - o.radius; h[a] = o.position[a] + o.radius; }
    } else {
        for (int a = 0; a < 3; a++) {
            c[a] = (o.v1[a] + o.v2[a] + o.v3[a]) / 3.0f;
            l[a] = fminf(o.v1[a], fminf(o.v2[a], o.v3[a]));
            h[a] = fmaxf(o.v1[a], fmaxf(o.v2[a], o.v3[a]));
        }
    }
    cx[i] = c[0]; cy[i] = c[1]; cz[i] = c[2];
    lo[i] = make_float4(l[0], l[1], l[2], 0.f);
    hi[i] = make_float4(h[0], h[1], h[2], 0.f);
    order[i] = i;
    segid[i] = root_slot;
}

/* segid holds the child index the last scatter wrote; slotmap (k_children) turns it into this level's segment index */
__global__ __launch_bounds__(256) void k_rr_flags(const float* cx, const float* cy, const float* cz, uint32_t* segid, const uint32_t* slotmap,
                                                  const Seg* segs, uint32_t n, uint32_t* flag) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    uint32_t s = segid[pos];
    if (slotmap && s != kNone) { s = slotmap[s]; segid[pos] = s; }
    uint32_t f = 0;
    if (s != kNone) {
        const Seg sg = segs[s];
        const float v = sg.axis == 0 ? cx[pos] : (sg.axis == 1 ? cy[pos] : cz[pos]);
        f = v <= sg.mid ? 1u : 0u;                     /* bvh.cu:58-68 */
    }
    flag[pos] = f;
}

/* stable partition inside every active segment; child k of segment s is kids[2 s + k].  segid2 holds the CHILD index
 * (k_children maps it to the next level's segment index through slotmap, the next k_flags applies the map). */
__global__ __launch_bounds__(256) void k_rr_scatter(uint32_t n, const uint32_t* flag, const uint32_t* scan, const uint32_t* segid, const Seg* segs,
                                                     const float* cx, const float* cy, const float* cz, const float4* lo, const float4* hi, const uint32_t* order,
                                                     float* cx2, float* cy2, float* cz2, float4* lo2, float4* hi2, uint32_t* order2, uint32_t* segid2,
                                                     Seg* kids) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const uint32_t s = segid[pos];
    uint32_t np = pos, ns = kNone;
    if (s != kNone) {
        const Seg sg = segs[s];
        const uint32_t last = sg.start + sg.count - 1u;
        const uint32_t nl = scan[last] + flag[last] - scan[sg.start];
        const uint32_t l = scan[pos] - scan[sg.start];
        const uint32_t f = flag[pos];
        np = f ? sg.start + l : sg.start + nl + (pos - sg.start - l);
        ns = 2u * s + (f ? 0u : 1u);
        if (pos == sg.start) {
            kids[2u * s].start = sg.start; kids[2u * s].count = nl;
            kids[2u * s + 1u].start = sg.start + nl; kids[2u * s + 1u].count = sg.count - nl;
        }
    }
    cx2[np] = cx[pos]; cy2[np] = cy[pos]; cz2[np] = cz[pos];
    lo2[np] = lo[pos]; hi2[np] = hi[pos]; order2[np] = order[pos]; segid2[np] = ns;
}

/* bvh.cu:205-253 / main.cu:108-131: strict comparisons; on a tie nothing is assigned (axis 0, midpoint 0) */
__host__ __device__ inline void choose_axis(float minX, float maxX, float minY, float maxY, float minZ, float maxZ, const float mid[3],
                                            uint32_t& axis, float& m) {
    const float ex = maxX - minX, ey = maxY - minY, ez = maxZ - minZ;
    axis = 0; m = 0.f;
    if (ex > ey && ex > ez) { axis = 0; m = mid[0]; }
    if (ey > ex && ey > ez) { axis = 1; m = mid[1]; }
    if (ez > ex && ez > ey) { axis = 2; m = mid[2]; }
}

/* the centroid tile at `tile0` (a multiple of 4): kPer aligned 128-bit loads per lane and axis; addresses beyond the child are
 * redirected to a valid one and masked when written to LDS (the arrays are padded by a tile) */
template <int kPer>
__device__ __forceinline__ void load_cen(float4 (&pre)[3 * kPer], const float* cx, const float* cy, const float* cz, uint32_t tile0, uint32_t end, int lane) {
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        uint32_t e = tile0 + (j * 64 + lane) * 4;
        e = e < end ? e : tile0;
        pre[0 * kPer + j] = *reinterpret_cast<const float4*>(cx + e);
        pre[1 * kPer + j] = *reinterpret_cast<const float4*>(cy + e);
        pre[2 * kPer + j] = *reinterpret_cast<const float4*>(cz + e);
    }
}
/* lo/hi of kBox elements per lane from `first` on; positions outside the child read its first element again (harmless for min/max) */
template <int kBox>
__device__ __forceinline__ void load_box(float4 (&pre)[2 * kBox], const float4* lo, const float4* hi, uint32_t first, uint32_t start, uint32_t end, int lane) {
#pragma unroll
    for (int j = 0; j < kBox; j++) {
        uint32_t e = first + j * 64 + lane;
        e = (e >= start && e < end) ? e : start;
        pre[2 * j] = lo[e]; pre[2 * j + 1] = hi[e];
    }
}

/* One block of three waves per child that still holds > 64 objects.
 *   wave 0      the three centroid sums: one float addition per element, in object order (bvh.cu:85,130,151,196).  The wave
 *               stages 1024-element tiles of the x, y, z arrays through LDS (loads for tile t+1 in flight while tile t is
 *               summed); lanes 0, 1, 2 then walk their axis with 128-bit LDS reads, 32 elements ahead of the additions.
 *   waves 1, 2  the child's box over the same tile: min/max are exact in any order (bvh.cu:70-129).
 * Tiles start at a multiple of 4 elements below the child's first position so that the 128-bit loads are aligned; elements
 * outside the child are replaced by -0.0f, the identity of float addition (x + -0 = x for every x). */
__global__ __launch_bounds__(192) void k_rr_children(const Seg* kids, const uint32_t* nseg_this_level, const float* cx, const float* cy, const float* cz,
                                                      const float4* lo, const float4* hi, Seg* next, uint32_t* next_count, uint32_t* slotmap) {
    __shared__ __align__(16) float s_c[2][3][kTile];
    __shared__ float s_box[2][6];
    const uint32_t k = blockIdx.x;
    if (k >= 2u * *nseg_this_level) return;            /* the grid is an upper bound when the host skipped a read-back */
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t start = kids[k].start, count = kids[k].count;
    if (count <= (uint32_t)kLeafMax) {                 /* a leaf: needs neither box nor midpoint */
        if (threadIdx.x == 0) slotmap[k] = kNone;
        return;
    }
    const uint32_t base = start & ~3u;
    const uint32_t end = start + count;
    const uint32_t ntiles = (end - base + kTile - 1) / kTile;
    constexpr int kPer = kTile / 4 / 64;               /* wave 0: float4 per lane, axis and tile */
    constexpr int kBox = kTile / 128;                  /* waves 1, 2: elements per lane and tile */
    float4 cpre[3 * kPer];                             /* wave 0: the staged centroids of the next tile */
    float4 bpre[2 * kBox];                             /* waves 1, 2: lo, hi of their elements of the next tile */
    float mn[3] = {kFmax, kFmax, kFmax}, mx[3] = {-kFmax, -kFmax, -kFmax};
    float sum = 0.f;                                   /* wave 0, lanes 0, 1, 2: the x, y, z chain */
    if (wave == 0) load_cen<kPer>(cpre, cx, cy, cz, base, end, lane);
    else load_box<kBox>(bpre, lo, hi, base + (wave - 1) * kBox * 64, start, end, lane);
    for (uint32_t t = 0; t < ntiles; t++) {
        const int buf = t & 1;
        if (wave == 0) {
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int j = 0; j < kPer; j++) {
                    const uint32_t e = base + t * kTile + (j * 64 + lane) * 4;
                    float4 v = cpre[a * kPer + j];
                    v.x = (e + 0 >= start && e + 0 < end) ? v.x : -0.f;
                    v.y = (e + 1 >= start && e + 1 < end) ? v.y : -0.f;
                    v.z = (e + 2 >= start && e + 2 < end) ? v.z : -0.f;
                    v.w = (e + 3 >= start && e + 3 < end) ? v.w : -0.f;
                    reinterpret_cast<float4*>(s_c[buf][a])[j * 64 + lane] = v;
                }
        } else {
#pragma unroll
            for (int j = 0; j < kBox; j++) {
                const float4 l = bpre[2 * j], h = bpre[2 * j + 1];
                mn[0] = fminf(mn[0], l.x); mn[1] = fminf(mn[1], l.y); mn[2] = fminf(mn[2], l.z);
                mx[0] = fmaxf(mx[0], h.x); mx[1] = fmaxf(mx[1], h.y); mx[2] = fmaxf(mx[2], h.z);
            }
        }
        __syncthreads();
        if (t + 1 < ntiles) {                          /* in flight while this tile is summed */
            if (wave == 0) load_cen<kPer>(cpre, cx, cy, cz, base + (t + 1) * kTile, end, lane);
            else load_box<kBox>(bpre, lo, hi, base + (t + 1) * kTile + (wave - 1) * kBox * 64, start, end, lane);
        }
        if (wave == 0 && lane < 3) {
            const float4* p = reinterpret_cast<const float4*>(s_c[buf][lane]);
            const uint32_t left = end - (base + t * kTile);          /* elements from this tile's start to the child's end */
            const int nq = left < (uint32_t)kTile ? (int)((left + 3u) / 4u) : kTile / 4;
            /* 32 elements per group; the reads of the next group are issued before the additions of this one */
            float4 v[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = p[u];
#pragma unroll 1
            for (int q = 0; q < nq; q += 16) {
#pragma unroll
                for (int u = 0; u < 8; u++) w[u] = p[q + 8 + u];
#pragma unroll
                for (int u = 0; u < 8; u++) { sum += v[u].x; sum += v[u].y; sum += v[u].z; sum += v[u].w; }
                const int qn = q + 16 < kTile / 4 ? q + 16 : q;      /* the last group reads itself again; the values are not used */
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = p[qn + u];
#pragma unroll
                for (int u = 0; u < 8; u++) { sum += w[u].x; sum += w[u].y; sum += w[u].z; sum += w[u].w; }
            }
        }
        __syncthreads();
    }
    if (wave > 0) {
        for (int off = 32; off > 0; off >>= 1)
            for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off)); }
        if (lane == 0) for (int a = 0; a < 3; a++) { s_box[wave - 1][a] = mn[a]; s_box[wave - 1][3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const float sx = __shfl(sum, 0), sy = __shfl(sum, 1), sz = __shfl(sum, 2);
        if (lane == 0) {
            const float minX = fminf(s_box[0][0], s_box[1][0]), minY = fminf(s_box[0][1], s_box[1][1]), minZ = fminf(s_box[0][2], s_box[1][2]);
            /* box_reset: max starts at the smallest POSITIVE float, so no maximum is ever below it */
            const float maxX = fmaxf(fmaxf(s_box[0][3], s_box[1][3]), kFminPos), maxY = fmaxf(fmaxf(s_box[0][4], s_box[1][4]), kFminPos),
                        maxZ = fmaxf(fmaxf(s_box[0][5], s_box[1][5]), kFminPos);
            const float c = (float)(size_t)count;
            const float mid[3] = {sx / c, sy / c, sz / c};
            uint32_t axis; float m;
            choose_axis(minX, maxX, minY, maxY, minZ, maxZ, mid, axis, m);
            const uint32_t slot = atomicAdd(next_count, 1u);
            Seg s; s.start = start; s.count = count; s.axis = axis; s.mid = m;
            next[slot] = s;
            slotmap[k] = slot;
        }
    }
}

__global__ __launch_bounds__(256) void k_rr_ranks(const uint32_t* order, uint32_t n, uint32_t* rank_of_scene) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < n) rank_of_scene[order[pos]] = pos;
}

}  // namespace

size_t refrank_scratch_bytes(uint32_t n) {
    size_t total = 0;
    (void)carve(nullptr, n ? n : 1, &total);
    return total;
}

#define RR(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { if (err) *err = std::string(#call) + ": " + hipGetErrorString(e_); return CRT_ERR_HIP; } \
    } while (0)

int refrank_build(void* scratch, const crt_prim* d_prims, uint32_t n, const float root_box[6], uint32_t* d_rank_of_scene, uint32_t* h_levels,
                  hipStream_t s, std::string* err) {
    if (n == 0) return CRT_OK;
    RScratch S = carve(scratch, n, nullptr);
    const uint32_t blocks = (n + 255u) / 256u;
    int cur = 0, segbuf = 0;
    /* the root (newBuild, main.cu:33-136): the scene box with the reference's max-init, midpoint = box centre */
    uint32_t nseg = 0;
    if (n > (uint32_t)kLeafMax) {
        const float minX = root_box[0], minY = root_box[2], minZ = root_box[4];
        const float maxX = std::max(root_box[1], kFminPos), maxY = std::max(root_box[3], kFminPos), maxZ = std::max(root_box[5], kFminPos);
        const float mid[3] = {(maxX + minX) / 2, (maxY + minY) / 2, (maxZ + minZ) / 2};
        Seg root; root.start = 0; root.count = n;
        choose_axis(minX, maxX, minY, maxY, minZ, maxZ, mid, root.axis, root.mid);
        RR(hipMemcpyAsync(S.segs[0], &root, sizeof root, hipMemcpyHostToDevice, s));
        nseg = 1;
    }
    hipLaunchKernelGGL(k_rr_init, dim3(blocks), dim3(256), 0, s, d_prims, n, S.cen[0][0], S.cen[0][1], S.cen[0][2], S.lo[0], S.hi[0],
                       S.order[0], S.segid[0], nseg ? 0u : kNone);
    /* S.ctr[0], S.ctr[1]: the number of segments of this level and of the next, alternating.  The host needs the count for
     * the grid of k_rr_children and to know when to stop; a read-back is a stream synchronisation (~40 us), which for a
     * small scene is most of the pass, so there it is taken only every fourth level: the grid is then the bound
     * min(2 x last known count x 2^levels since, n / 32 + 2) and a level without segments is a handful of empty launches. */
    RR(hipMemcpyAsync(S.ctr + 0, &nseg, 4, hipMemcpyHostToDevice, s));
    const uint32_t kid_cap = n / 32u + 2u;
    const int readback_every = n >= 200000u ? 1 : 4;
    uint32_t level = 0, bound = nseg;                  /* bound >= the true count */
    int cbuf = 0, since = 0;
    while (bound > 0) {
        if (level + 1 > 257 && since > 0) {
            /* `bound` is only an upper estimate between read-backs: a tree whose last split is at depth 256 -- which the host
             * builder accepts -- must not fail here because of launches that would have found nothing to do */
            RR(hipMemcpyAsync(&nseg, S.ctr + cbuf, 4, hipMemcpyDeviceToHost, s));
            RR(hipStreamSynchronize(s));
            bound = nseg; since = 0;
            if (bound == 0) break;
        }
        if (++level > 257) {       /* level L splits the nodes of depth L-1; bvh_build.cpp gives up beyond depth 256 */
            RR(hipStreamSynchronize(s));
            if (err) *err = "refrank_build: the reference's split does not terminate on this scene";
            return CRT_ERR_BUILD;
        }
        hipLaunchKernelGGL(k_rr_flags, dim3(blocks), dim3(256), 0, s, S.cen[cur][0], S.cen[cur][1], S.cen[cur][2], S.segid[cur],
                           level > 1 ? S.slotmap : (const uint32_t*)nullptr, S.segs[segbuf], n, S.flag);
        size_t tmp = S.scan_tmp_bytes;
        RR(rocprim::exclusive_scan(S.scan_tmp, tmp, S.flag, S.scan, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
        const int nxt = cur ^ 1;
        hipLaunchKernelGGL(k_rr_scatter, dim3(blocks), dim3(256), 0, s, n, S.flag, S.scan, S.segid[cur], S.segs[segbuf],
                           S.cen[cur][0], S.cen[cur][1], S.cen[cur][2], S.lo[cur], S.hi[cur], S.order[cur],
                           S.cen[nxt][0], S.cen[nxt][1], S.cen[nxt][2], S.lo[nxt], S.hi[nxt], S.order[nxt], S.segid[nxt], S.kids);
        cur = nxt;
        RR(hipMemsetAsync(S.ctr + (cbuf ^ 1), 0, 4, s));
        const uint32_t kids_bound = 2u * bound < kid_cap ? 2u * bound : kid_cap;
        hipLaunchKernelGGL(k_rr_children, dim3(kids_bound), dim3(192), 0, s, S.kids, S.ctr + cbuf, S.cen[cur][0], S.cen[cur][1], S.cen[cur][2],
                           S.lo[cur], S.hi[cur], S.segs[segbuf ^ 1], S.ctr + (cbuf ^ 1), S.slotmap);
        segbuf ^= 1; cbuf ^= 1;
        if (++since >= readback_every) {
            RR(hipMemcpyAsync(&nseg, S.ctr + cbuf, 4, hipMemcpyDeviceToHost, s));
            RR(hipStreamSynchronize(s));
            bound = nseg; since = 0;
        } else bound = kids_bound;
    }
    level -= (uint32_t)since;      /* levels launched after the last read-back that found segments may have been empty */
    hipLaunchKernelGGL(k_rr_ranks, dim3(blocks), dim3(256), 0, s, S.order[cur], n, d_rank_of_scene);
    RR(hipGetLastError());
    if (h_levels) *h_levels = level;
    return CRT_OK;
}
#undef RR

}  // namespace crt
