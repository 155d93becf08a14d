/* lbvh.hip -- device-side build of the 8-wide tree (CRT_BVH_WIDE_DEVICE).
 *
 * The host builders (bvh_build.cpp) take about a second for a million primitives; "rebuild every N
 * frames" (main.cu:534-541) on a large animated scene needs the rebuild on the device.  This is a
 * linear BVH: 30-bit Morton codes of the primitive centres, radix sort (rocPRIM), the binary radix
 * tree of Karras 2012 built with one thread per internal node, boxes fitted bottom-up with arrival
 * counters, then a level-by-level collapse into the same Node8 records the SAH builder produces
 * (largest-area child opened first, children in octant slots, internal children consecutive, leaf
 * primitives in one consecutive slot range) -- so refit, quantiser and traversal are shared.
 * Leaves hold one primitive.  There is no reference-topology pass in this mode: exact-t ties
 * resolve by scene index instead of the reference's leaf order (SURVEY.md 9.1 class G2).
 */
#include <cstring>
#include <string>
#include "kernels.h"
#include "vec.h"
#include <rocprim/device/device_radix_sort.hpp>

namespace crt {

namespace {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNone = 0xffffffffu;
enum { CTR_NODES = 0, CTR_SLOTS, CTR_QUEUE_A, CTR_QUEUE_B, CTR_DEPTH, CTR_WORDS = 8 };

struct QItem { uint32_t b, w, parent, pslot_depth; };     /* binary node, wide node, wide parent, slot | depth << 8 */

struct Scratch {
    float4 *plo, *phi;            /* primitive boxes, scene order */
    float4 *ilo, *ihi;            /* boxes of the binary internal nodes */
    uint32_t *keys, *keys2, *vals, *vals2;
    uint32_t *left, *right;       /* children of the binary internal nodes: index | kLeafBit */
    uint32_t *parent;             /* [0, n-1): of internal nodes; [n-1, 2n-1): of leaves */
    uint32_t *flags;
    uint32_t *ctr;
    int32_t *cbox;                /* as ordered ints: [0..5] bounds of the box centres, lo xyz, hi xyz; [6..11] the scene box from the exact primitive boxes */
    QItem *qa, *qb;
    void* sort_tmp; size_t sort_tmp_bytes;
};

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    uint32_t* k = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, (size_t)n, 0u, 30u, (hipStream_t)0);
    return bytes;
}

Scratch carve(void* base, uint32_t n, size_t* total) {
    Scratch s{};
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = base ? (char*)base + off : nullptr; off += up256(bytes); return p; };
    const size_t nn = n ? n : 1;
    s.plo = (float4*)take(nn * 16); s.phi = (float4*)take(nn * 16);
    s.ilo = (float4*)take(nn * 16); s.ihi = (float4*)take(nn * 16);
    s.keys = (uint32_t*)take(nn * 4); s.keys2 = (uint32_t*)take(nn * 4);
    s.vals = (uint32_t*)take(nn * 4); s.vals2 = (uint32_t*)take(nn * 4);
    s.left = (uint32_t*)take(nn * 4); s.right = (uint32_t*)take(nn * 4);
    s.parent = (uint32_t*)take(nn * 8);
    s.flags = (uint32_t*)take(nn * 4);
    s.ctr = (uint32_t*)take(CTR_WORDS * 4);
    s.cbox = (int32_t*)take(12 * 4);
    s.qa = (QItem*)take(nn * sizeof(QItem)); s.qb = (QItem*)take(nn * sizeof(QItem));
    s.sort_tmp_bytes = sort_temp_bytes(n);
    s.sort_tmp = take(s.sort_tmp_bytes + 256);
    if (total) *total = off;
    return s;
}

/* order-preserving float <-> int for atomicMin / atomicMax */
__device__ __forceinline__ int32_t f2ord(float f) { const int32_t i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int32_t i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
static inline float ord2f_host(int32_t i) { const int32_t b = i >= 0 ? i : i ^ 0x7fffffff; float f; memcpy(&f, &b, 4); return f; }

constexpr int kBoundsPerThread = 4;
/* plo / phi: the primitive boxes the tree is built from -- a sphere's is padded (layout.h sphere_box_half).  The scene box of the
 * EXACT primitive boxes is reduced beside the centre bounds: it is the root of the reference's own split, which refrank.hip replays */
__global__ __launch_bounds__(256) void k_lbvh_bounds(const crt_prim* prims, uint32_t n, float origin_bound, float4* plo, float4* phi, int32_t* cbox) {
    __shared__ float s_red[4][12];
    float cmin[6] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float cmax[6] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int r = 0; r < kBoundsPerThread; r++) {
        const uint32_t i = (blockIdx.x * kBoundsPerThread + r) * blockDim.x + threadIdx.x;
        if (i >= n) break;
        const crt_prim& o = prims[i];
        float lo[3], hi[3];
        if (o.type == 0) {
            const float hs = sphere_box_half(o.position[0], o.position[1], o.position[2], o.radius, origin_bound);
            for (int a = 0; a < 3; a++) {
                cmin[3 + a] = fminf(cmin[3 + a], o.position[a] - o.radius); cmax[3 + a] = fmaxf(cmax[3 + a], o.position[a] + o.radius);
                lo[a] = o.position[a] - hs; hi[a] = o.position[a] + hs;
            }
        } else {
            for (int a = 0; a < 3; a++) {
                lo[a] = fminf(o.v1[a], fminf(o.v2[a], o.v3[a]));
                hi[a] = fmaxf(o.v1[a], fmaxf(o.v2[a], o.v3[a]));
                cmin[3 + a] = fminf(cmin[3 + a], lo[a]); cmax[3 + a] = fmaxf(cmax[3 + a], hi[a]);
            }
        }
        plo[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
        phi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
        for (int a = 0; a < 3; a++) {
            const float c = 0.5f * (lo[a] + hi[a]);
            cmin[a] = fminf(cmin[a], c); cmax[a] = fmaxf(cmax[a], c);
        }
    }
    /* bounds of the box centres: wave reduction, the block's four waves through LDS, then one atomic per block and component
     * (atomics on one address serialise at ~10 ns each: one per wave cost 1 ms for 1M primitives) */
    for (int a = 0; a < 6; a++)
        for (int off = 32; off > 0; off >>= 1) {
            cmin[a] = fminf(cmin[a], __shfl_xor(cmin[a], off));
            cmax[a] = fmaxf(cmax[a], __shfl_xor(cmax[a], off));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) for (int a = 0; a < 6; a++) { s_red[wave][a] = cmin[a]; s_red[wave][6 + a] = cmax[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;          /* 0..2 centre bounds -> cbox[a], cbox[3 + a]; 3..5 exact scene box -> cbox[6 + (a - 3)], cbox[9 + (a - 3)] */
        const float mn = fminf(fminf(s_red[0][a], s_red[1][a]), fminf(s_red[2][a], s_red[3][a]));
        const float mx = fmaxf(fmaxf(s_red[0][6 + a], s_red[1][6 + a]), fmaxf(s_red[2][6 + a], s_red[3][6 + a]));
        const int lo_at = a < 3 ? a : 3 + a, hi_at = a < 3 ? 3 + a : 6 + a;
        if (mn <= mx) { atomicMin(&cbox[lo_at], f2ord(mn)); atomicMax(&cbox[hi_at], f2ord(mx)); }
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_lbvh_morton(const float4* plo, const float4* phi, uint32_t n, const int32_t* cbox,
                                                      uint32_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 lo = plo[i], hi = phi[i];
    const float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        const float l = ord2f(cbox[a]), h = ord2f(cbox[3 + a]);
        const float ext = h - l;
        float t = ext > 0.f ? (c[a] - l) / ext * 1024.0f : 0.f;
        t = fminf(fmaxf(t, 0.f), 1023.0f);
        q[a] = (uint32_t)t;
    }
    keys[i] = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    vals[i] = i;
}

__device__ __forceinline__ int delta(const uint32_t* k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = k[i], b = k[j];
    if (a == b) return 32 + __clz((int)((uint32_t)i ^ (uint32_t)j));
    return __clz((int)(a ^ b));
}

/* Karras 2012, one thread per internal node */
__global__ __launch_bounds__(256) void k_lbvh_tree(const uint32_t* k, int n, uint32_t* left, uint32_t* right, uint32_t* parent) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = (delta(k, n, i, i + 1) - delta(k, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(k, n, i, i - d);
    int lmax = 2;
    while (delta(k, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(k, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(k, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(k, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const bool lleaf = lo == gamma, rleaf = hi == gamma + 1;
    left[i] = lleaf ? ((uint32_t)gamma | kLeafBit) : (uint32_t)gamma;
    right[i] = rleaf ? ((uint32_t)(gamma + 1) | kLeafBit) : (uint32_t)(gamma + 1);
    parent[lleaf ? (n - 1 + gamma) : gamma] = (uint32_t)i;
    parent[rleaf ? (n - 1 + gamma + 1) : (gamma + 1)] = (uint32_t)i;
    if (i == 0) parent[0] = kNone;
}

/* Agent-coherent accesses to the internal boxes while k_lbvh_fit runs: 64-bit relaxed atomics at agent scope compile to
 * global loads / stores that go through to the level all XCDs share, so no cache write-back or invalidate is needed
 * around them (a release/acquire fence pair per node -- an L2 write-back each -- made this kernel 3.7 ms for 1M leaves). */
__device__ __forceinline__ void store_agent(float4* p, const float4& v) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    const unsigned long long a = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
    const unsigned long long b = (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32);
    __hip_atomic_store(q, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 load_agent(const float4* p) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<float4*>(p));
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)b), __uint_as_float((uint32_t)(b >> 32)));
}

/* Ordering of the bottom-up fit.  On gfx950 (as on gfx94x) an agent-scope relaxed atomic store is a write-through store (sc1)
 * that is visible to every CU of the device once it is acknowledged, and s_waitcnt vmcnt(0) waits for that acknowledgement;
 * an agent-scope relaxed atomic load (sc1) bypasses the non-coherent levels.  So "stores, s_waitcnt vmcnt(0), counter" on
 * the writer and "counter, loads" on the reader order the boxes without the release / acquire fence pair, each of which is
 * an L2 write-back on a chip with eight L2s (3.67 ms -> 0.17 ms for 1M primitives).  This is an ARCHITECTURE assumption, outside
 * the HIP / C++ memory model (relaxed atomics give no happens-before): it is taken only when compiling for gfx950; any
 * other target gets the fences. */
#if defined(__gfx950__)
#define CRT_FIT_PUBLISH() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define CRT_FIT_ACQUIRE() asm volatile("" ::: "memory")
#else
#define CRT_FIT_PUBLISH() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define CRT_FIT_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#endif

/* boxes bottom-up, one thread per leaf.  A thread carries the box of the subtree it has finished; at the parent the
 * first child to arrive (agent-scope counter) stops, the second reads its sibling's box, merges, stores the parent's
 * box and carries on.  Order at every node: box stores (agent scope), s_waitcnt vmcnt(0) = the stores are acknowledged,
 * then the counter; the reader's loads come after its counter operation returned. */
__global__ __launch_bounds__(256) void k_lbvh_fit(const float4* plo, const float4* phi, const uint32_t* vals, int n, const uint32_t* left,
                                                   const uint32_t* right, const uint32_t* parent, uint32_t* flags, float4* ilo, float4* ihi) {
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= n) return;
    uint32_t me = (uint32_t)j | kLeafBit;
    float4 mlo = plo[vals[j]], mhi = phi[vals[j]];
    uint32_t node = parent[n - 1 + j];
    for (int guard = 0; guard < 4096 && node != kNone; guard++) {
        CRT_FIT_PUBLISH();
        const uint32_t prev = __hip_atomic_fetch_add(&flags[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0u) return;
        CRT_FIT_ACQUIRE();
        const uint32_t cl = left[node], cr = right[node];
        const uint32_t sib = cl == me ? cr : cl;
        float4 slo, shi;
        if (sib & kLeafBit) { const uint32_t p = vals[sib & ~kLeafBit]; slo = plo[p]; shi = phi[p]; }
        else { slo = load_agent(&ilo[sib]); shi = load_agent(&ihi[sib]); }
        mlo = make_float4(fminf(mlo.x, slo.x), fminf(mlo.y, slo.y), fminf(mlo.z, slo.z), 0.f);
        mhi = make_float4(fmaxf(mhi.x, shi.x), fmaxf(mhi.y, shi.y), fmaxf(mhi.z, shi.z), 0.f);
        store_agent(&ilo[node], mlo);
        store_agent(&ihi[node], mhi);
        me = node;
        node = parent[node];
    }
}

__device__ __forceinline__ float box_area(const float4& lo, const float4& hi) {
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return 2.f * (dx * dy + dy * dz + dz * dx);
}

/* one level of the collapse: every queue item becomes one Node8 */
__global__ __launch_bounds__(128) void k_lbvh_collapse(const QItem* qin, const uint32_t* qin_count, QItem* qout, uint32_t* qout_count,
                                                        uint32_t* ctr, const float4* plo, const float4* phi, const float4* ilo, const float4* ihi,
                                                        const uint32_t* vals, const uint32_t* left, const uint32_t* right, Node8* nodes,
                                                        uint32_t node_cap, uint32_t* slot_to_scene, uint32_t nprims) {
    const uint32_t total = *qin_count;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        const QItem it = qin[q];
        uint32_t kids[8];
        int nk = 0;
        kids[nk++] = left[it.b]; kids[nk++] = right[it.b];
        while (nk < 8) {                      /* open the internal child with the largest surface area */
            int best = -1; float bestA = -1.f;
            for (int k = 0; k < nk; k++) {
                if (kids[k] & kLeafBit) continue;
                const float a = box_area(ilo[kids[k]], ihi[kids[k]]);
                if (a > bestA) { bestA = a; best = k; }
            }
            if (best < 0) break;
            const uint32_t c = kids[best];
            kids[best] = left[c];
            kids[nk++] = right[c];
        }
        float4 klo[8], khi[8];
        for (int k = 0; k < nk; k++) {
            if (kids[k] & kLeafBit) { const uint32_t p = vals[kids[k] & ~kLeafBit]; klo[k] = plo[p]; khi[k] = phi[p]; }
            else { klo[k] = ilo[kids[k]]; khi[k] = ihi[kids[k]]; }
        }
        /* octant slots (as build_wide): greedy (child, slot) pairs by the projection of the child's offset on the slot diagonal */
        const float4 nlo = ilo[it.b], nhi = ihi[it.b];
        const float cx = 0.5f * (nlo.x + nhi.x), cy = 0.5f * (nlo.y + nhi.y), cz = 0.5f * (nlo.z + nhi.z);
        int kid_at[8];
        for (int s = 0; s < 8; s++) kid_at[s] = -1;
        uint32_t done = 0, used = 0;
        for (int round = 0; round < nk; round++) {
            int bi = -1, bs = -1; float bc = -3.402823466e+38f;
            for (int i = 0; i < nk; i++) {
                if (done & (1u << i)) continue;
                const float vx = 0.5f * (klo[i].x + khi[i].x) - cx, vy = 0.5f * (klo[i].y + khi[i].y) - cy, vz = 0.5f * (klo[i].z + khi[i].z) - cz;
                for (int s = 0; s < 8; s++) {
                    if (used & (1u << s)) continue;
                    const float c = ((s & 4) ? vx : -vx) + ((s & 2) ? vy : -vy) + ((s & 1) ? vz : -vz);
                    if (c > bc || bi < 0) { bc = c; bi = i; bs = s; }
                }
            }
            done |= 1u << bi; used |= 1u << bs; kid_at[bs] = bi;
        }
        uint32_t ni = 0, nl = 0;
        for (int k = 0; k < nk; k++) { if (kids[k] & kLeafBit) nl++; else ni++; }
        const uint32_t cbase = ni ? atomicAdd(&ctr[CTR_NODES], ni) : 0u;
        const uint32_t pbase = nl ? atomicAdd(&ctr[CTR_SLOTS], nl) : 0u;
        const uint32_t qbase = ni ? atomicAdd(qout_count, ni) : 0u;
        const uint32_t depth = it.pslot_depth >> 8;
        atomicMax(&ctr[CTR_DEPTH], depth);
        if (it.w >= node_cap || cbase + ni > node_cap || pbase + nl > nprims) continue;       /* cannot happen: caps are n and n */
        Node8 nd;
        for (int s = 0; s < 8; s++) {
            nd.lox[s] = nd.loy[s] = nd.loz[s] = 3.402823466e+38f;
            nd.hix[s] = nd.hiy[s] = nd.hiz[s] = -3.402823466e+38f;
            nd.child[s] = kChildEmpty;
        }
        nd.parent = it.parent; nd.parent_slot = it.pslot_depth & 0xffu; nd.n_internal = ni; nd.pad = 0;
        nd.pad2[0] = nd.pad2[1] = nd.pad2[2] = nd.pad2[3] = 0;
        uint32_t ii = 0, li = 0;
        for (int s = 0; s < 8; s++) {
            const int k = kid_at[s];
            if (k < 0) continue;
            nd.lox[s] = klo[k].x; nd.loy[s] = klo[k].y; nd.loz[s] = klo[k].z;
            nd.hix[s] = khi[k].x; nd.hiy[s] = khi[k].y; nd.hiz[s] = khi[k].z;
            if (kids[k] & kLeafBit) {
                slot_to_scene[pbase + li] = vals[kids[k] & ~kLeafBit];
                nd.child[s] = make_leaf(pbase + li, 1u);
                li++;
            } else {
                nd.child[s] = cbase + ii;
                QItem c; c.b = kids[k]; c.w = cbase + ii; c.parent = it.w; c.pslot_depth = (uint32_t)s | ((depth + 1u) << 8);
                qout[qbase + ii] = c;
                ii++;
            }
        }
        nodes[it.w] = nd;
    }
}

__global__ __launch_bounds__(256) void k_lbvh_maps(const uint32_t* slot_to_scene, const uint32_t* rank_of_scene, uint32_t n, uint32_t* sidx,
                                                    uint32_t* slot_of, uint32_t* rank) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t si = slot_to_scene[s];
    sidx[s] = si; slot_of[si] = s;
    rank[s] = rank_of_scene ? rank_of_scene[si] : si;    /* the reference's leaf order (refrank.hip), else ties resolve by scene index */
}

/* n == 1: a root whose only child is the leaf */
__global__ void k_lbvh_single(const float4* plo, const float4* phi, Node8* nodes, uint32_t* slot_to_scene) {
    Node8 nd;
    for (int s = 0; s < 8; s++) {
        nd.lox[s] = nd.loy[s] = nd.loz[s] = 3.402823466e+38f;
        nd.hix[s] = nd.hiy[s] = nd.hiz[s] = -3.402823466e+38f;
        nd.child[s] = kChildEmpty;
    }
    nd.parent = kChildEmpty; nd.parent_slot = 0; nd.n_internal = 0; nd.pad = 0;
    nd.pad2[0] = nd.pad2[1] = nd.pad2[2] = nd.pad2[3] = 0;
    nd.lox[0] = plo[0].x; nd.loy[0] = plo[0].y; nd.loz[0] = plo[0].z;
    nd.hix[0] = phi[0].x; nd.hiy[0] = phi[0].y; nd.hiz[0] = phi[0].z;
    nd.child[0] = make_leaf(0u, 1u);
    slot_to_scene[0] = 0;
    nodes[0] = nd;
}

}  // namespace

size_t lbvh_scratch_bytes(uint32_t n) {
    size_t total = 0;
    (void)carve(nullptr, n, &total);
    return total;
}

#define LB(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { if (err) *err = std::string(#call) + ": " + hipGetErrorString(e_); return CRT_ERR_HIP; } \
    } while (0)

void launch_lbvh_maps(const uint32_t* slot_to_scene, const uint32_t* rank_of_scene, uint32_t n, uint32_t* sidx, uint32_t* slot_of, uint32_t* rank, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_lbvh_maps, dim3((n + 255u) / 256u), dim3(256), 0, s, slot_to_scene, rank_of_scene, n, sidx, slot_of, rank);
}

int lbvh_build(void* scratch, const crt_prim* d_prims, uint32_t n, float origin_bound, Node8* nodes_out, uint32_t node_cap, uint32_t* slot_to_scene,
               uint32_t* h_nnodes, uint32_t* h_depth, float root_box[6], float exact_box[6], hipStream_t s, std::string* err) {
    if (n == 0 || node_cap == 0) { if (err) *err = "lbvh_build: empty scene"; return CRT_ERR_ARG; }
    Scratch S = carve(scratch, n, nullptr);
    const uint32_t blocks = (n + 255u) / 256u;
    const int32_t big = 0x7f7fffff;       /* ordered int of +FLT_MAX; its negation orders as -FLT_MAX */
    const int32_t init[12] = {big, big, big, -big, -big, -big, big, big, big, -big, -big, -big};
    LB(hipMemcpyAsync(S.cbox, init, sizeof init, hipMemcpyHostToDevice, s));
    LB(hipMemsetAsync(S.ctr, 0, CTR_WORDS * 4, s));
    hipLaunchKernelGGL(k_lbvh_bounds, dim3((n + 256u * kBoundsPerThread - 1u) / (256u * kBoundsPerThread)), dim3(256), 0, s, d_prims, n, origin_bound, S.plo, S.phi, S.cbox);
    if (n == 1) {
        hipLaunchKernelGGL(k_lbvh_single, dim3(1), dim3(1), 0, s, S.plo, S.phi, nodes_out, slot_to_scene);
        float4 lo, hi;
        int32_t eb[6];
        LB(hipMemcpyAsync(&lo, S.plo, 16, hipMemcpyDeviceToHost, s));
        LB(hipMemcpyAsync(&hi, S.phi, 16, hipMemcpyDeviceToHost, s));
        LB(hipMemcpyAsync(eb, S.cbox + 6, sizeof eb, hipMemcpyDeviceToHost, s));
        LB(hipStreamSynchronize(s));
        root_box[0] = lo.x; root_box[1] = hi.x; root_box[2] = lo.y; root_box[3] = hi.y; root_box[4] = lo.z; root_box[5] = hi.z;
        for (int a = 0; a < 3; a++) { exact_box[2 * a] = ord2f_host(eb[a]); exact_box[2 * a + 1] = ord2f_host(eb[3 + a]); }
        *h_nnodes = 1; *h_depth = 1;
        return CRT_OK;
    }
    hipLaunchKernelGGL(k_lbvh_morton, dim3(blocks), dim3(256), 0, s, S.plo, S.phi, n, S.cbox, S.keys, S.vals);
    size_t tmp = S.sort_tmp_bytes;
    LB(rocprim::radix_sort_pairs(S.sort_tmp, tmp, S.keys, S.keys2, S.vals, S.vals2, (size_t)n, 0u, 30u, s));
    LB(hipMemsetAsync(S.flags, 0, (size_t)n * 4, s));
    hipLaunchKernelGGL(k_lbvh_tree, dim3(blocks), dim3(256), 0, s, S.keys2, (int)n, S.left, S.right, S.parent);
    hipLaunchKernelGGL(k_lbvh_fit, dim3(blocks), dim3(256), 0, s, S.plo, S.phi, S.vals2, (int)n, S.left, S.right, S.parent, S.flags, S.ilo, S.ihi);
    /* collapse, one launch per level; the level sizes come back to the host (a build is not a per-frame operation) */
    QItem root; root.b = 0; root.w = 0; root.parent = kChildEmpty; root.pslot_depth = (1u << 8);
    uint32_t one = 1;
    LB(hipMemcpyAsync(S.qa, &root, sizeof root, hipMemcpyHostToDevice, s));
    LB(hipMemcpyAsync(S.ctr + CTR_QUEUE_A, &one, 4, hipMemcpyHostToDevice, s));
    LB(hipMemcpyAsync(S.ctr + CTR_NODES, &one, 4, hipMemcpyHostToDevice, s));
    uint32_t level_count = 1;
    int cur = 0;
    for (int level = 0; level < 512 && level_count > 0; level++) {
        QItem* qin = cur ? S.qb : S.qa; QItem* qout = cur ? S.qa : S.qb;
        uint32_t* cin = S.ctr + (cur ? CTR_QUEUE_B : CTR_QUEUE_A); uint32_t* cout = S.ctr + (cur ? CTR_QUEUE_A : CTR_QUEUE_B);
        LB(hipMemsetAsync(cout, 0, 4, s));
        uint32_t gb = (level_count + 127u) / 128u;
        if (gb > 4096u) gb = 4096u;
        hipLaunchKernelGGL(k_lbvh_collapse, dim3(gb), dim3(128), 0, s, qin, cin, qout, cout, S.ctr, S.plo, S.phi, S.ilo, S.ihi, S.vals2,
                           S.left, S.right, nodes_out, node_cap, slot_to_scene, n);
        LB(hipMemcpyAsync(&level_count, cout, 4, hipMemcpyDeviceToHost, s));
        LB(hipStreamSynchronize(s));
        cur ^= 1;
    }
    LB(hipGetLastError());
    uint32_t h[CTR_WORDS];
    float4 rlo, rhi;
    int32_t eb[6];
    LB(hipMemcpyAsync(eb, S.cbox + 6, sizeof eb, hipMemcpyDeviceToHost, s));
    LB(hipMemcpyAsync(h, S.ctr, sizeof h, hipMemcpyDeviceToHost, s));
    LB(hipMemcpyAsync(&rlo, S.ilo, 16, hipMemcpyDeviceToHost, s));
    LB(hipMemcpyAsync(&rhi, S.ihi, 16, hipMemcpyDeviceToHost, s));
    LB(hipStreamSynchronize(s));
    if (level_count != 0 || h[CTR_SLOTS] != n || h[CTR_NODES] > node_cap) { if (err) *err = "lbvh_build: collapse did not place every primitive"; return CRT_ERR_BUILD; }
    root_box[0] = rlo.x; root_box[1] = rhi.x; root_box[2] = rlo.y; root_box[3] = rhi.y; root_box[4] = rlo.z; root_box[5] = rhi.z;
    for (int a = 0; a < 3; a++) { exact_box[2 * a] = ord2f_host(eb[a]); exact_box[2 * a + 1] = ord2f_host(eb[3 + a]); }
    *h_nnodes = h[CTR_NODES]; *h_depth = h[CTR_DEPTH];
    return CRT_OK;
}
#undef LB

}  // namespace crt
