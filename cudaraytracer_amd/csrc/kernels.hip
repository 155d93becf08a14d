/* kernels.hip -- hand-written CDNA4 (gfx950) kernels of the render path.
 *
 * Built with -ffp-contract=off: the reference's hit/no-hit predicates are threshold tests on float
 * expressions (Ray.cu:114,122,126,148,151,158) and must be evaluated with the same roundings as a
 * non-contracting build; IEEE divide/sqrt are hipcc's default.  The only fused operations are the
 * explicit fmaf() of the box test, which only has to be conservative.
 *
 * Design (wave64, no MFMA -- nothing here is a dense contraction):
 *   k_trace8     THE traversal kernel (CRT_BVH_WIDE, ~90 % of a frame): persistent waves pull rays from two queues in HBM
 *                (the shadow rays of a bounce and the nearest-hit rays of the next share a launch); each lane walks the
 *                compressed 8-wide tree (QNode8: 80-byte nodes, 8-bit plane codes decoded and near/far-selected by v_perm_b32, used by v_fma_mix_f32,
 *                an inline-asm slab test yielding a hit MASK, no distance sort) with one 64-bit stack entry per level in
 *                LDS (stack[depth][lane]); the wave votes between a node step and a primitive step each iteration; rays are
 *                fetched in XCD-affine chunks; a second instantiation of the loop shares the work of a wave's last rays
 *                among its idle lanes.  bvhTraverse (Ray.cu:184-236) with the reference's strict-'<' tie-break made
 *                tree-shape independent via reference ranks; shadow predicates of Ray.cu:621-629 / 650-656.
 *   k_trace      the same contract on the reference's own topology (CRT_BVH_REFERENCE: exact boxes, two children per
 *                node, up to 64 primitives per leaf), 4-wide node format with a distance sort; one queue per launch.
 *   k_shade      RayIntersection's bookkeeping (Ray.cu:479-525), wallIntersection (Ray.cu:238-457)
 *                and Shade's light loop (Ray.cu:560-659), emitting compacted shadow / reflection
 *                queues with wave-aggregated atomics.
 *   k_resolve    Ray.cu:660 once the shadow results are in.
 *   k_gen_primary GeneratePrimaryRays (Ray.cu:31-52) over this rank's image tiles, 8x8 pixels per wave.
 *   k_convert    Ray.cu:724-731 on the device.
 *   k_refit_*    refitTree (bvh.cu:288-400) as leaf pass + bottom-up pass with arrival counters.
 */
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "kernels.h"
#include "vec.h"

namespace crt {

#define CRT_DEV __device__ __forceinline__
/* lane mask of a predicate.  HIP's __ballot takes an int, and the compiler then rebuilds the mask it already holds in
 * SGPRs with v_cndmask_b32 + v_cmp_ne_u32 -- two half-rate VALU instructions per vote, ten per iteration of the traversal
 * loops; the intrinsic on the i1 is one s_and_b64 with exec. */
#define crt_ballot(P) __builtin_amdgcn_ballot_w64((bool)(P))
/* population count of a lane mask on the scalar unit, as a 32-bit value: __popcll's 64-bit result drew 64-bit compares, which
 * exist only on the vector unit */
static __device__ __forceinline__ int popc_mask(unsigned long long m) {
    int r;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(r) : "s"(m) : "scc");
    return r;
}

static_assert(kFetchChunk >= 1 && (kFetchChunk & (kFetchChunk - 1)) == 0, "CRT_FETCH_CHUNK must be a power of two");
constexpr uint32_t kRegionAlign = kFetchChunk > 64 ? (uint32_t)kFetchChunk : 64u;      /* region size granule: a multiple of the chunk and of a wave */

/* ------------------------------------------------------------------------------------------
 * primitive tests
 * ------------------------------------------------------------------------------------------ */
struct GeomRec { float4 a, b, c; };

CRT_DEV GeomRec load_geom(const Geom* g, uint32_t slot) {
    const float4* p = reinterpret_cast<const float4*>(g + slot);
    GeomRec r; r.a = p[0]; r.b = p[1]; r.c = p[2];
    return r;
}
CRT_DEV bool is_sphere(const GeomRec& g) { return __float_as_uint(g.c.y) == kSphereTag; }

/* intersectTriangle, Ray.cu:143-158, hit test only; Ng is the precomputed value of Ray.cu:145.
 * `tlimit` is an early-out that cannot change which hit wins. */
CRT_DEV bool tri_test(const GeomRec& g, const vec3& o, const vec3& d, float tlimit, float& tout) {
    const vec3 v1(g.a.x, g.a.y, g.a.z), v2(g.a.w, g.b.x, g.b.y), v3(g.b.z, g.b.w, g.c.x), ng(g.c.y, g.c.z, g.c.w);
    const float denom = dot(ng, d);
    if (!(fabsf(denom) > .0001f)) return false;
    const float t = dot((v1 - o), ng) / denom;
    if (!(t >= -0.0001f)) return false;
    if (t > tlimit) return false;
    const vec3 X = o + t * d;
    const float test1 = dot(cross((v2 - v1), (X - v1)), ng);
    const float test2 = dot(cross((v3 - v2), (X - v2)), ng);
    const float test3 = dot(cross((v1 - v3), (X - v3)), ng);
    if (test1 >= 0.0f && test2 >= 0.0f && test3 >= 0.0f) { tout = t; return true; }
    return false;
}

/* intersectSphere, Ray.cu:105-133: returns isectT; `tsmall` is the root used for the hit point */
CRT_DEV bool sphere_test(const GeomRec& g, const vec3& o, const vec3& d, float& tout, float& tsmall) {
    const float RAY_EPSILON = 0.000000001f;
    const vec3 c(g.a.x, g.a.y, g.a.z);
    const float a = dot(d, d);
    const float b = 2.0f * dot(d, o - c);
    const float cc = g.b.x + dot(o, o) + (-2.0f * dot(c, o)) - g.b.y;
    const float disc = b * b - 4.0f * a * cc;
    if (!(disc > RAY_EPSILON)) return false;
    const float t = (-b - sqrtf(disc)) / (2.0f * a);
    const float t2 = (-b + sqrtf(disc)) / (2.0f * a);
    if (!(t2 > RAY_EPSILON)) return false;
    tout = (t > RAY_EPSILON) ? t : t2;
    tsmall = t;
    return true;
}

/* ------------------------------------------------------------------------------------------
 * k_trace
 *
 * Every active lane always holds one work item: an internal node (`cur`) or a pending range of
 * primitives of a leaf ([lp, lend)).  Each iteration the WAVE votes: if more lanes hold a node,
 * the wave runs one node step (those lanes test 4 child boxes), otherwise one primitive step
 * (those lanes test one primitive).  The branch is wave-uniform, so the two bodies never
 * serialise against each other inside one iteration; the minority lanes keep their item.
 * (First version ran "if node ... else leaf-loop" per lane: VALU lane utilisation was 27 %,
 * profiles/r01a_first_path.md.)
 * ------------------------------------------------------------------------------------------ */
#define CSWAP(ta, la, tb, lb)                                   \
    {                                                           \
        const bool s_ = (tb) < (ta);                            \
        const float t0_ = s_ ? (tb) : (ta), t1_ = s_ ? (ta) : (tb); \
        const uint32_t l0_ = s_ ? (lb) : (la), l1_ = s_ ? (la) : (lb); \
        ta = t0_; tb = t1_; la = l0_; lb = l1_;                 \
    }

#if CRT_TRACE_MIN_WAVES > 0
#define CRT_TRACE_BOUNDS __launch_bounds__(kTraceBlock, CRT_TRACE_MIN_WAVES)
#else
#define CRT_TRACE_BOUNDS __launch_bounds__(kTraceBlock)
#endif
template <int MODE, bool COUNT>
__global__ CRT_TRACE_BOUNDS void k_trace(TraceArgs A) {
    __shared__ uint32_t s_stack[kStackDepth + 1][kTraceBlock];   /* last row: write-only trash for the branch-free push */
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t total = *A.count;
    const uint32_t gthread = blockIdx.x * kTraceBlock + tid;
    uint32_t* ovf = A.ovf + (size_t)gthread * (size_t)(A.ovf_depth > 0 ? A.ovf_depth : 0);

    bool active = false;
    bool exhausted = false;          /* wave-uniform: the queue has no more chunks for this wave */
    uint32_t cnext = 0, cend = 0;    /* wave-uniform: this wave's private chunk [cnext, cend) */
    /* HW_REG_XCC_ID (id 20), bits [3:0]: which XCD this wave runs on.  Used for L2 affinity only. */
    const uint32_t xcd = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const uint32_t per_region = (((total + 7u) >> 3) + kRegionAlign - 1u) & ~(kRegionAlign - 1u);     /* whole chunks: a chunk never straddles two regions or two queues */
    uint32_t region_k = 0;           /* wave-uniform: regions (starting at this XCD's) already found empty */
    vec3 o, d;
    float ix = 0, iy = 0, iz = 0, ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
    float tbest = 0;
    int best = -1;
    uint32_t cur = 0, ray = 0;
    uint32_t lp = 0, lend = 0;      /* pending primitives of the current leaf */
    int sp = 0;
    int found = 0;          /* SHADOW: a hit with t > 1e-9 exists */
    uint32_t lkind = kShadowDirectional;   /* SHADOW: float bits of the ray's light distance (kernels.h) */
    unsigned long long nbox = 0, ntri = 0, nsph = 0, nvis = 0;

    /* take the next item off the stack, or report that the ray is finished */
#define POP_ITEM(FINISH)                                                                 \
    {                                                                                    \
        if (sp == 0) FINISH = true;                                                      \
        else {                                                                           \
            sp--;                                                                        \
            uint32_t l_;                                                                 \
            if (sp < kStackDepth) l_ = s_stack[sp][tid];                                 \
            else if (sp - kStackDepth < A.ovf_depth) l_ = ovf[sp - kStackDepth];         \
            else { l_ = 0; FINISH = true; } /* dropped entry: status word already set */ \
            if (l_ & kLeafFlag) { lp = l_ & kLeafStartMask; lend = lp + ((l_ >> kLeafCountShift) & 63u) + 1u; } \
            else cur = l_;                                                               \
        }                                                                                \
    }
#define PUSH_ITEM(L)                                                                     \
    {                                                                                    \
        if (sp < kStackDepth) s_stack[sp][tid] = (L);                                    \
        else if (sp - kStackDepth < A.ovf_depth) ovf[sp - kStackDepth] = (L);            \
        else atomicOr(A.status, 1u);                                                     \
        sp++;                                                                            \
    }

    for (;;) {
        /* ---- refill idle lanes: each wave owns a private chunk of the queue and takes a new one
         *      with ONE atomic when it runs dry (per-refill atomics on the single fetch counter
         *      serialise at ~10 ns each and were the limit of this kernel) ---- */
        if (!exhausted) {
            const unsigned long long idle = crt_ballot(!active);
            const int nidle = __popcll(idle);
            if (nidle >= kRefillMinIdle) {
                if (cnext >= cend) {
                    /* the queue is cut into 8 contiguous regions, one fetch counter each (256 B apart).
                     * A wave starts in the region of its own XCD -- the XCD's L2 then serves one part
                     * of the image / one part of the secondary rays -- and moves on to the next
                     * region when one is used up, so that no XCD idles at the tail. */
                    bool got = false;
                    while (region_k < 8 && !got) {
                        const uint32_t r = (xcd + region_k) & 7u;
                        const uint32_t rbeg = r * per_region;
                        const uint32_t rend = rbeg + per_region < total ? rbeg + per_region : total;
                        uint32_t base = 0xffffffffu;
                        if (rbeg < rend) {
                            if (lane == 0) base = atomicAdd(A.next + r * kFetchStride, (uint32_t)kFetchChunk);
                            base = __builtin_amdgcn_readfirstlane(base);
                        }
                        if (rbeg < rend && base < rend - rbeg) {
                            cnext = rbeg + base;
                            cend = cnext + (uint32_t)kFetchChunk < rend ? cnext + (uint32_t)kFetchChunk : rend;
                            got = true;
                        } else {
                            /* used up.  When a launch ends every wave finds every region used up, and eight failing atomics
                             * per wave on eight addresses serialise (a tenth of a millisecond per launch): the first wave to
                             * fail on a region says so in a shared mask, and a wave that has just failed skips what the mask
                             * already rules out */
                            region_k++;
                            uint32_t done = 0;
                            if (lane == 0) {
                                done = __hip_atomic_load(A.next + kFetchDoneWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (rbeg < rend && !((done >> r) & 1u)) atomicOr(A.next + kFetchDoneWord, 1u << r);
                            }
                            done = __builtin_amdgcn_readfirstlane(done);
                            while (region_k < 8 && ((done >> ((xcd + region_k) & 7u)) & 1u)) region_k++;
                        }
                    }
                    if (!got) exhausted = true;
                }
                if (!exhausted) {
                    const uint32_t idx = cnext + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    const uint32_t take = (uint32_t)nidle < cend - cnext ? (uint32_t)nidle : cend - cnext;
                    const bool mine = !active && idx < cend;
                    cnext += take;
                    if (mine) {
                        const float4 ro = A.rayO[idx], rd = A.rayD[idx];
                        if (MODE == TRACE_NEAREST && __float_as_uint(rd.w) == 0xffffffffu) {
                            A.hitSlot[idx] = -3;            /* hole in the primary queue: pixel outside the frame */
                            A.hitT[idx] = 0.f;
                        } else {
                            ray = idx;
                            o = vec3(ro.x, ro.y, ro.z); d = vec3(rd.x, rd.y, rd.z);
                            if (MODE == TRACE_SHADOW) { ray = __float_as_uint(ro.w); lkind = __float_as_uint(rd.w); }
                            /* clamped reciprocal: a zero component (the centre pixel column has
                             * direction.x == 0 exactly) becomes +-1e20 instead of inf, so 0*inf never appears */
                            float rx = 1.0f / d.x, ry = 1.0f / d.y, rz = 1.0f / d.z;
                            if (!(fabsf(d.x) > 1e-20f)) rx = copysignf(1e20f, d.x);
                            if (!(fabsf(d.y) > 1e-20f)) ry = copysignf(1e20f, d.y);
                            if (!(fabsf(d.z) > 1e-20f)) rz = copysignf(1e20f, d.z);
                            ix = rx; iy = ry; iz = rz;
                            /* boxes are stored exact; the test pads them: lo' = lo - pad, hi' = hi + pad */
                            const float px = 1e-3f + 1e-6f * (fabsf(o.x) + A.scene_abs);
                            const float py = 1e-3f + 1e-6f * (fabsf(o.y) + A.scene_abs);
                            const float pz = 1e-3f + 1e-6f * (fabsf(o.z) + A.scene_abs);
                            ax = (o.x + px) * ix; bx = (o.x - px) * ix;
                            ay = (o.y + py) * iy; by = (o.y - py) * iy;
                            az = (o.z + pz) * iz; bz = (o.z - pz) * iz;
                            tbest = 1000000000.0f; best = -1; cur = 0; sp = 0; found = 0; lp = lend = 0;
                            active = true;
                        }
                    }
                }
            }
        }
        if (crt_ballot(active) == 0ull) { if (exhausted) break; else continue; }

        /* ---- traverse until enough lanes idle to be worth a refill ---- */
        do {
            const bool want_prim = active && (lp < lend);
            const bool want_node = active && !want_prim;
            const int n_node = __popcll(crt_ballot(want_node));
            const int n_prim = __popcll(crt_ballot(want_prim));
            bool finish = false;
            if (n_node * CRT_VOTE_PRIM_WEIGHT >= n_prim * CRT_VOTE_NODE_WEIGHT) {
                if (want_node) {
                    float tn0, tn1, tn2, tn3;
                    uint32_t c0, c1, c2, c3;
                    if (COUNT) nvis++;
#define SLAB_T(K, X0, X1, Y0, Y1, Z0, Z1, C)                                                    \
    {                                                                                           \
        const float tn = fmaxf(fmaxf(fminf(X0, X1), fminf(Y0, Y1)), fmaxf(fminf(Z0, Z1), -1.0001e-4f)); \
        const float tf = fminf(fminf(fmaxf(X0, X1), fmaxf(Y0, Y1)), fminf(fmaxf(Z0, Z1), tbest)); \
        K = ((C) != kChildEmpty && tn <= tf) ? tn : INFINITY;                                   \
        if (COUNT && (C) != kChildEmpty) nbox++;                                                \
    }
                    {
                        const float4* np = reinterpret_cast<const float4*>(A.sv.nodes + cur);
                        const float4 lox = np[0], loy = np[1], loz = np[2], hix = np[3], hiy = np[4], hiz = np[5];
                        const uint4 ch = reinterpret_cast<const uint4*>(np)[6];
#ifdef CRT_EXPERIMENT_EXTRA_LOADS   /* timing experiment only: is the vector L1's access rate the limit? */
                        for (int e_ = 0; e_ < CRT_EXPERIMENT_EXTRA_LOADS; e_++) {
                            typedef float f4v_ __attribute__((ext_vector_type(4)));
                            const f4v_ x_ = reinterpret_cast<const volatile f4v_*>(np)[7];
                            asm volatile("" ::"v"(x_));
                        }
#endif
#define SLAB(K, LX, LY, LZ, HX, HY, HZ, C)                                                       \
    {                                                                                            \
        const float x0 = fmaf(LX, ix, -ax), x1 = fmaf(HX, ix, -bx);                              \
        const float y0 = fmaf(LY, iy, -ay), y1 = fmaf(HY, iy, -by);                              \
        const float z0 = fmaf(LZ, iz, -az), z1 = fmaf(HZ, iz, -bz);                              \
        SLAB_T(K, x0, x1, y0, y1, z0, z1, C)                                                     \
    }
                        SLAB(tn0, lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, ch.x)
                        SLAB(tn1, lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, ch.y)
                        SLAB(tn2, lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, ch.z)
                        SLAB(tn3, lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, ch.w)
#undef SLAB
                        c0 = ch.x; c1 = ch.y; c2 = ch.z; c3 = ch.w;
                    }
#undef SLAB_T
                    CSWAP(tn0, c0, tn1, c1)
                    CSWAP(tn2, c2, tn3, c3)
                    CSWAP(tn0, c0, tn2, c2)
                    CSWAP(tn1, c1, tn3, c3)
                    CSWAP(tn1, c1, tn2, c2)
                    /* nearest child becomes the next item; the others go on the stack far-first.
                     * After the sort the children that were hit form a prefix, so the three stores
                     * go to computed rows (or to the trash row) without branches. */
                    const int npush = (tn1 < INFINITY ? 1 : 0) + (tn2 < INFINITY ? 1 : 0) + (tn3 < INFINITY ? 1 : 0);
                    if (sp + 3 <= kStackDepth) {
                        const int r1 = npush >= 1 ? sp + npush - 1 : kStackDepth;
                        const int r2 = npush >= 2 ? sp + npush - 2 : kStackDepth;
                        const int r3 = npush >= 3 ? sp : kStackDepth;
                        s_stack[r1][tid] = c1;
                        s_stack[r2][tid] = c2;
                        s_stack[r3][tid] = c3;
                        sp += npush;
                    } else {                   /* deep trees only: spill rows live in HBM */
                        if (tn3 < INFINITY) PUSH_ITEM(c3)
                        if (tn2 < INFINITY) PUSH_ITEM(c2)
                        if (tn1 < INFINITY) PUSH_ITEM(c1)
                    }
                    if (tn0 < INFINITY) {
                        if (c0 & kLeafFlag) { lp = c0 & kLeafStartMask; lend = lp + ((c0 >> kLeafCountShift) & 63u) + 1u; }
                        else cur = c0;
                    } else POP_ITEM(finish)
                }
            } else {
                if (want_prim) {
                    const uint32_t p = lp++;
                    const GeomRec g = load_geom(A.sv.geom, p);
                    float t = 0.f, tsm;
                    bool hit;
                    if (is_sphere(g)) { if (COUNT) nsph++; hit = sphere_test(g, o, d, t, tsm); }
                    else { if (COUNT) ntri++; hit = tri_test(g, o, d, tbest, t); }
                    if (hit) {
                        if (MODE == TRACE_NEAREST) {
                            /* strict '<' as Ray.cu:199,216; equal t resolves to the primitive the
                             * reference would have met first (lower rank in its leaf order) */
                            if (t < tbest || (t == tbest && best >= 0 && A.sv.rank[p] < A.sv.rank[best])) { tbest = t; best = (int)p; }
                        } else {
                            if (lkind != kShadowDirectional) {  /* point light or area-light sample: nearest t decides, Ray.cu:621-629 */
                                if (t < tbest) { tbest = t; best = (int)p; }
                            } else {                              /* Ray.cu:650-656: hit exists and min t > 1e-9 */
                                if (!(t > 0.000000001f)) { found = 0; finish = true; }
                                else { found = 1; if (tbest > 0.000000001f) tbest = 0.000000001f; }
                            }
                        }
                    }
                    if (lp == lend && !finish) POP_ITEM(finish)
                }
            }
            if (finish) {
                if (MODE == TRACE_NEAREST) {
                    A.hitSlot[ray] = best;
                    A.hitT[ray] = tbest;
                } else {
                    bool shadowed;
                    if (lkind != kShadowDirectional) {
                        shadowed = false;
                        if (best >= 0) {
                            /* Ray.cu:625-629; k_shade computed dtoLight from this very origin */
                            const vec3 ipoint = o + tbest * d;
                            const float dtoLight = __uint_as_float(lkind & 0x7fffffffu);
                            const float fx = ipoint.x - o.x, fy = ipoint.y - o.y, fz = ipoint.z - o.z;
                            const float dtoLightIntersection = sqrtf(fx * fx + fy * fy + fz * fz);
                            shadowed = dtoLight > dtoLightIntersection;
                            if ((lkind >> 31) && !(tbest > 0.000000001f)) shadowed = false;   /* area samples: Ray.cu:589 */
                        }
                    } else shadowed = (found != 0);
                    if (shadowed) A.shflag[ray] = 1;
                }
                active = false; lp = lend = 0;
            }
        } while (exhausted ? (crt_ballot(active) != 0ull) : (__popcll(crt_ballot(active)) > 64 - kRefillMinIdle));
    }
#undef POP_ITEM
#undef PUSH_ITEM
    if (COUNT) {
        atomicAdd(&A.counters[0], nbox);
        atomicAdd(&A.counters[1], ntri);
        atomicAdd(&A.counters[2], nsph);
        atomicAdd(&A.counters[3], nvis);
    }
}

/* ------------------------------------------------------------------------------------------
 * k_trace8: traversal of the compressed 8-wide tree (QNode8, layout.h) -- CRT_BVH_WIDE.
 *
 * The 4-wide quantised kernel of the earlier commits was bound by VALU issue (profiles/r01c:
 * SQ_INSTS_VALU = 100 % of the SIMDs' issue slots, ~250 VALU per node visit, 54 visits per ray on
 * the 1M soup).  This kernel spends the instructions differently:
 *   - no distance sort and no per-child stack pushes: a node visit produces an 8-bit hit mask; the
 *     internal hits stay together as ONE 64-bit stack entry (child base | imask | hits permuted
 *     into front-to-back priority by the ray's octant, via a 2 KB LDS table), leaf hits are
 *     tested before descending;
 *   - 8 children per visit cut the visits (dependent round trips) per ray from 54 to 38;
 *   - 80-byte nodes: with half the VALU work the limit became the vector L1's request rate (every
 *     lane reads its own node: one 16-byte request per clock per CU), so the node is 5 requests.
 * Scheduling (wave vote between node steps and primitive steps, chunked XCD-affine ray fetch,
 * per-lane stack rows in LDS) is the same as in k_trace.
 * ------------------------------------------------------------------------------------------ */
/* Slab test of one child of a QNode8.  The six plane codes arrive as fp16 DENORMALS -- the byte q under a zero high byte,
 * q * 2^-24, two codes per word, assembled by ONE v_perm_b32 per word whose per-ray selector also picks the lo or the hi row
 * (near / far) -- so a plane distance is ONE v_fma_mix_f32 (fp16 x fp32 + fp32) instead of v_cvt_f32_ubyte + v_fma_f32; the
 * stored scale exponent carries the 2^24 (layout.h kQ8ScaleShift).  Until round 3 the codes were 1024 + q under the constant
 * byte 0x64 and the rows were selected by twelve v_cndmask + three v_cmp per node step.  HI picks the upper half of the words.
 * The whole ten-instruction sequence is written out because the compiler canonicalises (v_max_f32 x,x) every asm result that
 * feeds fmaxf/fminf.
 *   tn = max(tnx, tny, tnz, -1.0001e-4)      tf = min(tfx, tfy, tfz, tbest)      raw = 2 raw + (tn <= tf)
 * (v_addc_co_u32 shifts the hit bit in through the per-lane carry: children are tested 7 .. 0). */
template <bool HI>
CRT_DEV void slab8(uint32_t& raw, uint32_t wnx, uint32_t wfx, uint32_t wny, uint32_t wfy, uint32_t wnz, uint32_t wfz, float sx,
                   float sy, float sz, float bnx, float bfx, float bny, float bfy, float bnz, float bfz, float tbest) {
    float tn, tf, t1, t2;
#define SLAB8_OPERANDS                                                                                                   \
    : [tn] "=&v"(tn), [tf] "=&v"(tf), [t1] "=&v"(t1), [t2] "=&v"(t2), [raw] "+v"(raw)                                    \
    : [wnx] "v"(wnx), [wfx] "v"(wfx), [wny] "v"(wny), [wfy] "v"(wfy), [wnz] "v"(wnz), [wfz] "v"(wfz), [sx] "v"(sx),      \
      [sy] "v"(sy), [sz] "v"(sz), [bnx] "v"(bnx), [bfx] "v"(bfx), [bny] "v"(bny), [bfy] "v"(bfy), [bnz] "v"(bnz),        \
      [bfz] "v"(bfz), [tb] "v"(tbest)                                                                                    \
    : "vcc"
    if (HI)
        asm(
            "v_fma_mix_f32 %[t1], %[wnx], %[sx], %[bnx] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[t2], %[wny], %[sy], %[bny] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[tn], %[wnz], %[sz], %[bnz] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_max_f32 %[tn], 0xb8d1bc76, %[tn]\n\t"
            "v_max3_f32 %[tn], %[t1], %[t2], %[tn]\n\t"
            "v_fma_mix_f32 %[t1], %[wfx], %[sx], %[bfx] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[t2], %[wfy], %[sy], %[bfy] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[tf], %[wfz], %[sz], %[bfz] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_min_f32 %[tf], %[tf], %[tb]\n\t"
            "v_min3_f32 %[tf], %[t1], %[t2], %[tf]\n\t"
            "v_cmp_le_f32 vcc, %[tn], %[tf]\n\t"
            "v_addc_co_u32 %[raw], vcc, %[raw], %[raw], vcc"
            SLAB8_OPERANDS);
    else
        asm(
            "v_fma_mix_f32 %[t1], %[wnx], %[sx], %[bnx] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[t2], %[wny], %[sy], %[bny] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[tn], %[wnz], %[sz], %[bnz] op_sel_hi:[1,0,0]\n\t"
            "v_max_f32 %[tn], 0xb8d1bc76, %[tn]\n\t"
            "v_max3_f32 %[tn], %[t1], %[t2], %[tn]\n\t"
            "v_fma_mix_f32 %[t1], %[wfx], %[sx], %[bfx] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[t2], %[wfy], %[sy], %[bfy] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %[tf], %[wfz], %[sz], %[bfz] op_sel_hi:[1,0,0]\n\t"
            "v_min_f32 %[tf], %[tf], %[tb]\n\t"
            "v_min3_f32 %[tf], %[t1], %[t2], %[tf]\n\t"
            "v_cmp_le_f32 vcc, %[tn], %[tf]\n\t"
            "v_addc_co_u32 %[raw], vcc, %[raw], %[raw], vcc"
            SLAB8_OPERANDS);
#undef SLAB8_OPERANDS
}

/* region markers: comment lines in the generated assembly (no instruction, no clobber) that tools/isa_mix.py uses to
 * attribute the kernel's instructions to the parts of an iteration */
#define CRT_MARK(NAME) asm volatile("; CRT_MARK " NAME);

#if CRT_TRACE8_MIN_WAVES > 0
#define CRT_TRACE8_BOUNDS __launch_bounds__(BLOCK, CRT_TRACE8_MIN_WAVES)
#else
#define CRT_TRACE8_BOUNDS __launch_bounds__(BLOCK)
#endif
/* BLOCK threads per block.  256 is what ships.  A block's registers and LDS come free only when its last wave ends, so
 * 128- and 64-thread blocks (-DCRT_TRACE_BLOCK_VARIANTS, CRT_TRACE_BLOCK=128|64; the per-block octant table then costs
 * more LDS, which a shallower in-LDS stack -- stack_depth8_of -- gives back so that 5 waves per SIMD still fit) were tried
 * as a way to hand the tail of a launch to the frames other slots have in flight sooner.  Measured (round 3,
 * profiles/r03_rank_slice_async.jsonl): the whole 4K frame 17.05 / 17.81 / 17.98 ms for 256 / 128 / 64, and with three frames of
 * an eighth of it in flight 2.37 / 2.45 / 2.46 ms -- the same 0.90-0.91 of each variant's own ideal: what overlaps is not
 * block retirement, so the fastest single launch wins */
/* OVF = false: the tree is no deeper than the in-LDS stack (TraceArgs::ovf_depth == 0, which the host derives from the blob
 * header's depth: one stack entry per level at most), so neither pop nor push looks at the overflow rows in HBM -- two compares
 * and two branches less per node step, and no spilled row pointer */
template <bool COUNT, int BLOCK, bool OVF>
__global__ CRT_TRACE8_BOUNDS void k_trace8(TraceArgs A) {
    constexpr int kDepth = stack_depth8_of(BLOCK);
    __shared__ uint2 s_stack[kDepth][BLOCK];
    __shared__ uint8_t s_perm[8][256];     /* s_perm[R][m]: bit k of m moved to bit k ^ R */
    /* work sharing in the tail (see "steal" below), indexed by the thread that OWNS the ray */
    __shared__ unsigned long long s_key[BLOCK];      /* minimum over the ray's lanes: ordered t << 32 | rank, or a shadow outcome code */
    __shared__ uint32_t s_cnt[BLOCK];                /* lanes still working on the ray */
    __shared__ uint32_t s_best[BLOCK];               /* slot that holds the minimum */
    __shared__ uint8_t s_pair[BLOCK];                /* steal step: the wave's offering lanes, compacted */
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wbase = tid & ~63;
    for (int i = tid; i < 2048; i += BLOCK) {
        const int R_ = i >> 8, m = i & 255;
        int y = 0;
        for (int k = 0; k < 8; k++) if ((m >> k) & 1) y |= 1 << (k ^ R_);
        s_perm[R_][m] = (uint8_t)y;
    }
    __syncthreads();
    /* one launch serves TWO independent queues: the shadow rays of a bounce and the nearest-hit rays of the
     * next one (either may be absent).  A launch of even a thousand rays lasts 0.23 ms -- a ray is ~100
     * dependent node visits -- so two launches per bounce meant two such tails.  Virtual queue positions
     * [0, ns64) are shadow rays, [ns64, total) nearest-hit rays; ns64 is a multiple of the chunk size, so a
     * chunk never straddles the two.  Bit 31 of `ray` marks a lane that carries a shadow ray. */
    const uint32_t total_s = A.shcount ? *A.shcount : 0u, total_n = A.count ? *A.count : 0u;
    const uint32_t ns64 = (total_s + (uint32_t)kFetchChunk - 1u) / (uint32_t)kFetchChunk * (uint32_t)kFetchChunk;
    const uint32_t total = ns64 + total_n;
    const uint32_t gthread = blockIdx.x * BLOCK + tid;
    uint2* ovf = OVF ? reinterpret_cast<uint2*>(A.ovf) + (size_t)gthread * (size_t)(A.ovf_depth > 0 ? A.ovf_depth : 0) : nullptr;
    const char* qbase = reinterpret_cast<const char*>(A.sv.qnodes);

    bool active = false;
    bool exhausted = false;
    uint32_t cnext = 0, cend = 0;
    const uint32_t xcd = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const uint32_t per_region = (((total + 7u) >> 3) + kRegionAlign - 1u) & ~(kRegionAlign - 1u);     /* whole chunks: a chunk never straddles two regions or two queues */
    uint32_t region_k = 0;
    vec3 o, d;
    float ix = 0, iy = 0, iz = 0, anx = 0, afx = 0, any_ = 0, afy = 0, anz = 0, afz = 0;
    uint32_t R = 0;        /* bit 2 / 1 / 0: the direction is NEGATIVE along x / y / z; child slot ^ R = its place in front-to-back order */
    /* v_perm selector of the NEAR plane codes of children 2,3 / 6,7 of an axis: bytes 2,3 of the hi row (second v_perm operand)
     * for a negative direction, of the lo row (first operand) for a positive one, each under a zero high byte.  The far codes come
     * from the same selector with the two rows swapped; children 0,1 / 4,5 = - 0x00020002 */
    uint32_t selx = 0, sely = 0, selz = 0;
    float tbest = 0;
    int best = -1;
    uint32_t ray = 0;
    uint32_t gbase = 0, gmask = 0;                 /* internal-child group: base node | imask << 8 | permuted hits */
    uint32_t plm = 0, meta0 = 0, meta1 = 0;    /* leaf children hit at the last node: first leaf slot (< 2^24) | hit mask by child << 24 */
    uint32_t lp = 0, lend = 0;
    int sp = 0;
    /* bit 0: a shadow ray has found an occluder with t > 1e-9; bit 1: a directional shadow ray met a hit with t <= 1e-9 (lit,
     * overrides); bits 8-13: the lane that owns the ray; bits 16-19: stack entries given away from the bottom; bit 24: the
     * ray is shared with other lanes of the wave (result goes through s_key / s_cnt) */
    uint32_t fstate = 1u << 31;          /* kIdle, below */
    uint32_t tick = 0;
    uint32_t lkind = kShadowDirectional;   /* shadow lanes: float bits of the ray's light distance (kernels.h) */
    unsigned long long nbox[2] = {0, 0}, ntri[2] = {0, 0}, nsph[2] = {0, 0}, nvis[2] = {0, 0};      /* [0] nearest-hit rays, [1] shadow rays */
    unsigned long long sch[5] = {0, 0, 0, 0, 0}, nzero = 0, nstale = 0, nfound = 0, nzero_t = 0, nculled_t = 0;
    unsigned long long itc[8] = {0, 0, 0, 0, 0, 0, 0, 0};      /* wave-level iterations: all, with a refill fetch, with a finishing lane, with a ray set-up,
                                                                 * node steps with a stack pop, primitive steps with a leaf decode, with a hit */

    /* One iteration of the traversal loop, instantiated twice: TAIL = false is the loop that runs while the ray queue lasts
     * (refill, vote, node / primitive step, write-out); TAIL = true runs once this wave has found the queue dry -- no refill,
     * and the idle lanes share the work of the busy ones (the steal step).  Keeping the two apart keeps the sharing logic
     * (shared write-out, empty stack entries, 20 lane shuffles) out of the loop where 97 % of the time goes: in one loop
     * it cost 19 % of the frame.  Returns true when the loop should end. */
    /* result of a finished ray whose lanes did not share it: nearest hit, or the shadow predicate */
    auto write_result = [&]() __attribute__((always_inline)) {
        if (!(ray >> 31)) {
            A.hitSlot[ray] = best;
            A.hitT[ray] = tbest;
        } else {
            bool shadowed;
            if (lkind != kShadowDirectional && !(fstate & 1u)) {
                shadowed = false;
                if (best >= 0) {
                    /* Ray.cu:625-629; k_shade computed dtoLight from this very origin */
                    const vec3 ipoint = o + tbest * d;
                    const float dtoLight = __uint_as_float(lkind & 0x7fffffffu);
                    const float fx = ipoint.x - o.x, fy = ipoint.y - o.y, fz = ipoint.z - o.z;
                    const float dtoLightIntersection = sqrtf(fx * fx + fy * fy + fz * fz);
                    shadowed = dtoLight > dtoLightIntersection;
                    if ((lkind >> 31) && !(tbest > 0.000000001f)) shadowed = false;   /* area samples: Ray.cu:589 */
                }
            } else shadowed = ((fstate & 1u) != 0);
            if (shadowed) A.shflag[ray & 0x7fffffffu] = 1;
        }
    };
    /* fstate bit 25: the lane's ray has finished in the main loop and its result is still in the lane's registers.  Writing it out
     * in the finishing iteration ran the write-out for one or two lanes in a third of all iterations; a finished lane idles until
     * the next refill anyway, so the result is written when the lane is set up for its next ray (or when the queue runs dry) */
    constexpr uint32_t kZombie = 1u << 25;
    /* fstate bit 31 == !active, kept beside the bool: the loop's vote on "which lanes have a ray" is then one signed compare of a
     * VGPR; a vote on the loop-carried bool costs two VALU instructions (the compiler rebuilds its lane mask through a VGPR) */
    constexpr uint32_t kIdle = 1u << 31;
#define CRT_MARK8(NAME) if constexpr (!TAIL) { CRT_MARK(NAME) }
    auto iteration = [&](auto tail_tag) __attribute__((always_inline)) -> bool {
        constexpr bool TAIL = decltype(tail_tag)::value;
        CRT_MARK8("refill1_begin")
        /* ---- refill, first half: when enough lanes are idle, take rays off the queue and ISSUE their
         *      loads; the set-up runs after this iteration's traversal step, whose own node / primitive
         *      loads return later (loads return in order), so the refill costs the wave no extra stall ---- */
        bool pend = false, pend_sh = false;
        float4 pro4, prd;      /* read only by the lanes that set pend; a ray record carries its own index / result slot in .w of its origin */
        /* one vote on `active` serves the refill trigger, the "anything to do" test and the node count below */
        const unsigned long long amask = crt_ballot((int32_t)fstate >= 0);
        const int nactive = popc_mask(amask);
        if constexpr (TAIL) {
        if (COUNT && lane == 0) itc[7]++;
        /* ---- steal: the queue is dry and this wave is draining.  A launch ends when its longest ray does (a chain of a few
         *      hundred dependent node visits), while most lanes of most waves idle: an idle lane now takes the BOTTOM stack
         *      entry (the largest untouched subtree) of a lane that is still descending, together with a copy of its ray, and
         *      traverses it as its own; the lanes of a ray fold their results into the owner's LDS record -- minimum of
         *      (t, rank) for nearest-hit rays, which is what one lane's strict '<' plus rank tie-break computes too, and
         *      the strongest outcome for directional shadow rays -- and the last one to finish writes it out.  Point-light and
         *      area-sample shadow rays keep their lane (their predicate needs the one nearest hit's own arithmetic). ---- */
        if (amask != 0ull && ((++tick) & (uint32_t)(A.steal_every - 1)) == 0u && __popcll(amask) <= 64 - A.steal_min_idle) {
            const uint32_t nst = (fstate >> 16) & 15u;
            const bool offer = active && (!(ray >> 31) || lkind == kShadowDirectional) && nst < 12u && (int)nst < kDepth && sp > (int)nst;
            const unsigned long long omask = crt_ballot(offer);
            const int noffer = __popcll(omask);
            if (noffer > 0) {
                const unsigned long long lt = (1ull << lane) - 1ull;
                const int orank = __popcll(omask & lt), irank = __popcll(~amask & lt);
                const int ntake = noffer < 64 - __popcll(amask) ? noffer : 64 - __popcll(amask);
                if (offer) s_pair[wbase + orank] = (uint8_t)lane;
                if (offer && orank < ntake && !(fstate & (1u << 24))) { s_cnt[tid] = 1u; s_key[tid] = ~0ull; }      /* an unshared victim owns its ray */
                __builtin_amdgcn_wave_barrier();
                const bool take = !active && irank < ntake;
                const int v = take ? (int)s_pair[wbase + irank] : lane;
                const float c_ox = __shfl(o.x, v), c_oy = __shfl(o.y, v), c_oz = __shfl(o.z, v);
                const float c_dx = __shfl(d.x, v), c_dy = __shfl(d.y, v), c_dz = __shfl(d.z, v);
                const float c_ix = __shfl(ix, v), c_iy = __shfl(iy, v), c_iz = __shfl(iz, v);
                const float c_anx = __shfl(anx, v), c_afx = __shfl(afx, v), c_any = __shfl(any_, v), c_afy = __shfl(afy, v);
                const float c_anz = __shfl(anz, v), c_afz = __shfl(afz, v), c_tb = __shfl(tbest, v);
                const uint32_t c_R = __shfl(R, v), c_ray = __shfl(ray, v), c_lk = __shfl(lkind, v), c_fs = __shfl(fstate, v);
                const int c_best = __shfl(best, v);
                if (take) {
                    o = vec3(c_ox, c_oy, c_oz); d = vec3(c_dx, c_dy, c_dz);
                    ix = c_ix; iy = c_iy; iz = c_iz; anx = c_anx; afx = c_afx; any_ = c_any; afy = c_afy; anz = c_anz; afz = c_afz;
                    tbest = c_tb; best = c_best; R = c_R; ray = c_ray; lkind = c_lk;
                    selx = (c_R & 4u) ? 0x0c030c02u : 0x0c070c06u; sely = (c_R & 2u) ? 0x0c030c02u : 0x0c070c06u; selz = (c_R & 1u) ? 0x0c030c02u : 0x0c070c06u;
                    const uint32_t vnst = (c_fs >> 16) & 15u;
                    const uint2 e = s_stack[vnst][wbase + v];
                    gbase = e.x; gmask = e.y;
                    s_stack[vnst][wbase + v] = make_uint2(e.x, e.y & ~0xffu);       /* taken: the victim will pop an empty group */
                    sp = 0; lp = lend = 0; plm = 0;
                    const uint32_t own = (c_fs >> 8) & 63u;
                    fstate = (c_fs & 1u) | (own << 8) | (1u << 24);
                    atomicAdd(&s_cnt[wbase + own], 1u);
                    active = true;
                }
                if (offer && orank < ntake) fstate = (fstate + (1u << 16)) | (1u << 24);
                __builtin_amdgcn_wave_barrier();
                if (COUNT && !A.tail_probe && lane == 0) { atomicAdd(&A.counters[30], 1ull); atomicAdd(&A.counters[31], (unsigned long long)ntake); }
            }
        }
        }
        if constexpr (!TAIL)
        if (!exhausted) {
            const unsigned long long idle = ~amask;
            const int nidle = 64 - nactive;
            if (nidle >= kRefillMinIdle8) {
                if (COUNT && lane == 0) itc[1]++;
                if (cnext >= cend) {
                    bool got = false;
                    while (region_k < 8 && !got) {
                        const uint32_t r = (xcd + region_k) & 7u;
                        const uint32_t rbeg = r * per_region;
                        const uint32_t rend = rbeg + per_region < total ? rbeg + per_region : total;
                        uint32_t base = 0xffffffffu;
                        if (rbeg < rend) {
                            if (lane == 0) base = atomicAdd(A.next + r * kFetchStride, (uint32_t)kFetchChunk);
                            base = __builtin_amdgcn_readfirstlane(base);
                        }
                        if (rbeg < rend && base < rend - rbeg) {
                            cnext = rbeg + base;
                            cend = cnext + (uint32_t)kFetchChunk < rend ? cnext + (uint32_t)kFetchChunk : rend;
                            got = true;
                        } else {
                            /* used up.  When a launch ends every wave finds every region used up, and eight failing atomics
                             * per wave on eight addresses serialise (a tenth of a millisecond per launch): the first wave to
                             * fail on a region says so in a shared mask, and a wave that has just failed skips what the mask
                             * already rules out */
                            region_k++;
                            uint32_t done = 0;
                            if (lane == 0) {
                                done = __hip_atomic_load(A.next + kFetchDoneWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (rbeg < rend && !((done >> r) & 1u)) atomicOr(A.next + kFetchDoneWord, 1u << r);
                            }
                            done = __builtin_amdgcn_readfirstlane(done);
                            while (region_k < 8 && ((done >> ((xcd + region_k) & 7u)) & 1u)) region_k++;
                        }
                    }
                    if (!got) exhausted = true;
                }
                if (!exhausted) {
                    const uint32_t idx = cnext + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    const uint32_t take = (uint32_t)nidle < cend - cnext ? (uint32_t)nidle : cend - cnext;
                    const bool sh = cnext < ns64;          /* wave-uniform: the chunk lies in one queue */
                    pend_sh = sh;
                    const bool mine = !active && idx < cend && (!sh || idx < total_s);
                    cnext += take;
                    if (mine) {          /* issue the loads only: they complete under the traversal step below */
                        if (sh) {
                            const uint32_t q = A.shperm ? A.shperm[idx] : idx;      /* sorted visiting order, if any */
                            pro4 = A.shO[q]; prd = A.shD[q];
                        } else {
                            const uint32_t q = A.perm ? A.perm[idx - ns64] : idx - ns64;
                            pro4 = A.rayO[q]; prd = A.rayD[q];
                        }
                        pend = true;
                    }
                }
            }
        }
        CRT_MARK8("refill1_end")
        if (COUNT && lane == 0) itc[0]++;
        if (amask != 0ull) {
            CRT_MARK8("vote_begin")
            /* a lane without a ray has lp == lend == 0 and plm == 0 (set where `active` is cleared), and lp <= lend always:
             * one compare on one value, which the vote takes as its lane mask without a detour through a VGPR */
            const bool want_prim = ((lend - lp) | (plm >> 24)) != 0u;
            const bool want_node = active && !want_prim;
            const int n_prim = popc_mask(crt_ballot(want_prim));
            const int n_node = nactive - n_prim;
            bool finish = false;
            if (COUNT && lane == 0) {     /* scheduling statistics: [8] node iterations [9] lanes served [10] prim iterations [11] lanes served [12] active lanes */
                if (n_node * CRT_VOTE8_PRIM_WEIGHT >= n_prim * CRT_VOTE8_NODE_WEIGHT) { sch[0]++; sch[1] += n_node; } else { sch[2]++; sch[3] += n_prim; }
                sch[4] += n_node + n_prim;
            }
            CRT_MARK8("vote_end")
            if (n_node * CRT_VOTE8_PRIM_WEIGHT >= n_prim * CRT_VOTE8_NODE_WEIGHT) {
                CRT_MARK8("node_begin")
                if (want_node) {
                    if (COUNT) { const bool any_ = crt_ballot((gmask & 0xffu) == 0u) != 0ull; if (lane == 0 && any_) itc[4]++; }
                    CRT_MARK8("pop_begin")
                    if ((gmask & 0xffu) == 0u) {
                        if (sp == 0) finish = true;
                        else {
                            sp--;
                            uint2 e;
                            if (!OVF || sp < kDepth) e = s_stack[sp][tid];
                            else if (sp - kDepth < A.ovf_depth) e = ovf[sp - kDepth];
                            else { e = make_uint2(0u, 0u); finish = true; }
                            gbase = e.x; gmask = e.y;
                        }
                    }
                    CRT_MARK8("pop_end")
                    if (!finish && (!TAIL || (gmask & 0xffu) != 0u)) {          /* tail: a popped entry is empty when its group was given to another lane */
                        /* pending children sit in visiting order from bit 0 up (bit = slot ^ R, R = the NEGATIVE directions): the next
                         * one is the lowest set bit, taken and cleared without building a mask (bits 0-7 are not all zero here) */
                        const uint32_t pbit = (uint32_t)__builtin_ctz(gmask);
                        const uint32_t slot = pbit ^ R;
                        gmask &= gmask - 1u;
                        const uint32_t node = gbase + (uint32_t)__popc((gmask >> 8) & ((1u << slot) - 1u));
                        if (gmask & 0xffu) {
                            if (!OVF || sp < kDepth) s_stack[sp][tid] = make_uint2(gbase, gmask);
                            else if (sp - kDepth < A.ovf_depth) ovf[sp - kDepth] = make_uint2(gbase, gmask);
                            else atomicOr(A.status, 1u);
                            sp++;
                        }
                        /* 32-bit byte offsets from the uniform base (nodes < 2^24): scalar base + vector offset addressing */
                        static_assert(sizeof(QNode8) == 80, "node * 80 below");
                        const uint4* np = reinterpret_cast<const uint4*>(qbase + __umul24(node, 80u));     /* v_mul_u32_u24: full rate, nodes < 2^24 */
                        const uint4 h0 = np[0], h1 = np[1];
                        const uint4 PX = np[2], PY = np[3], PZ = np[4];       /* lo_x[8] hi_x[8] | lo_y hi_y | lo_z hi_z */
                        const uint32_t em = h0.w;
                        const float nox = __uint_as_float(h0.x), noy = __uint_as_float(h0.y), noz = __uint_as_float(h0.z);
                        const float sx = __uint_as_float((em << 23) & 0x7f800000u) * ix;
                        const float sy = __uint_as_float((em << 15) & 0x7f800000u) * iy;
                        const float sz = __uint_as_float((em << 7) & 0x7f800000u) * iz;
                        const float bnx = fmaf(nox, ix, -anx), bfx = fmaf(nox, ix, -afx);
                        const float bny = fmaf(noy, iy, -any_), bfy = fmaf(noy, iy, -afy);
                        const float bnz = fmaf(noz, iz, -anz), bfz = fmaf(noz, iz, -afz);
                        /* rows: PX = lo_x[0..3] lo_x[4..7] hi_x[0..3] hi_x[4..7].  One v_perm_b32 per pair of codes takes them from the
                         * lo or the hi row as the ray's selector says (near / far) and puts each under a zero byte: fp16 denormals
                         * q * 2^-24, which v_fma_mix_f32 multiplies by the scale (stored with 2^24 folded in) */
                        /* the far plane comes from the other row: the same selector with the two rows swapped */
                        const uint32_t X23 = selx, X01 = selx - 0x00020002u;
                        const uint32_t Y23 = sely, Y01 = sely - 0x00020002u;
                        const uint32_t Z23 = selz, Z01 = selz - 0x00020002u;
                        uint32_t raw;        /* eight compare results are shifted in; whatever was there ends above bit 7, which every use masks off */
                        if (COUNT) raw = 0; else asm volatile("" : "=v"(raw));
#define SLAB8x4(RAW, TB, LOX, HIX, LOY, HIY, LOZ, HIZ)                                                                 \
    {                                                                                                         \
        const uint32_t c0 = __builtin_amdgcn_perm(LOX, HIX, X23), c1 = __builtin_amdgcn_perm(HIX, LOX, X23),             \
                       c2 = __builtin_amdgcn_perm(LOY, HIY, Y23), c3 = __builtin_amdgcn_perm(HIY, LOY, Y23),             \
                       c4 = __builtin_amdgcn_perm(LOZ, HIZ, Z23), c5 = __builtin_amdgcn_perm(HIZ, LOZ, Z23);             \
        slab8<true>(RAW, c0, c1, c2, c3, c4, c5, sx, sy, sz, bnx, bfx, bny, bfy, bnz, bfz, TB);            \
        slab8<false>(RAW, c0, c1, c2, c3, c4, c5, sx, sy, sz, bnx, bfx, bny, bfy, bnz, bfz, TB);           \
        const uint32_t a0 = __builtin_amdgcn_perm(LOX, HIX, X01), a1 = __builtin_amdgcn_perm(HIX, LOX, X01),             \
                       a2 = __builtin_amdgcn_perm(LOY, HIY, Y01), a3 = __builtin_amdgcn_perm(HIY, LOY, Y01),             \
                       a4 = __builtin_amdgcn_perm(LOZ, HIZ, Z01), a5 = __builtin_amdgcn_perm(HIZ, LOZ, Z01);             \
        slab8<true>(RAW, a0, a1, a2, a3, a4, a5, sx, sy, sz, bnx, bfx, bny, bfy, bnz, bfz, TB);            \
        slab8<false>(RAW, a0, a1, a2, a3, a4, a5, sx, sy, sz, bnx, bfx, bny, bfy, bnz, bfz, TB);           \
    }
                        SLAB8x4(raw, tbest, PX.y, PX.w, PY.y, PY.w, PZ.y, PZ.w)      /* children 7, 6, 5, 4 */
                        SLAB8x4(raw, tbest, PX.x, PX.z, PY.x, PY.z, PZ.x, PZ.z)      /* children 3, 2, 1, 0 */
                        uint32_t raw_far = 0;            /* counting only: what the visit would have hit without the t limit */
                        if (COUNT) {
                            const float tinf = INFINITY;
                            SLAB8x4(raw_far, tinf, PX.y, PX.w, PY.y, PY.w, PZ.y, PZ.w)
                            SLAB8x4(raw_far, tinf, PX.x, PX.z, PY.x, PY.z, PZ.x, PZ.z)
                        }
#undef SLAB8x4
                        const uint32_t vmask = h1.x >> 24, imask = em >> 24;
                        raw &= vmask;                       /* an empty slot can pass the padded test of a degenerate node */
                        if (COUNT) { nbox[ray >> 31] += (unsigned long long)__popc(vmask); nvis[ray >> 31]++; if (raw == 0u) nzero++; if ((ray >> 31) && (fstate & 1u)) nstale++;
                                     if (raw == 0u && (raw_far & vmask) != 0u) nzero_t++; nculled_t += (unsigned long long)__popc(raw_far & vmask & ~raw); }
                        gbase = h1.x & 0xffffffu;
                        gmask = (imask << 8) | (uint32_t)s_perm[R][raw & imask];
                        plm = h1.y | ((raw & ~imask) << 24);
                        meta0 = h1.z; meta1 = h1.w;
                        if (raw == 0u && sp == 0) finish = true;
                    }
                }
                CRT_MARK8("node_end")
            } else {
                CRT_MARK8("prim_begin")
                if (want_prim) {
                    if (COUNT) { const bool any_ = crt_ballot(lp == lend) != 0ull; if (lane == 0 && any_) itc[5]++; }
                    CRT_MARK8("decode_begin")
                    if (lp == lend) {
                        const uint32_t k = (uint32_t)__builtin_ctz(plm >> 24);      /* not zero: want_prim with lp == lend means a leaf hit is pending */
                        plm &= ~(1u << (24u + k));
                        const uint32_t m = ((k & 4u) ? meta1 : meta0) >> ((k & 3u) * 8u);
                        lp = (plm & 0xffffffu) + (m & 31u);
                        lend = lp + ((m >> 5) & 3u) + 1u;
                    }
                    CRT_MARK8("decode_end")
                    const uint32_t p = lp++;
                    const GeomRec g = load_geom(A.sv.geom, p);
                    float t = 0.f, tsm;
                    bool hit;
                    const bool shray = (ray >> 31) != 0u;
                    if (is_sphere(g)) { CRT_MARK8("sphere_begin") if (COUNT) nsph[shray]++; hit = sphere_test(g, o, d, t, tsm); CRT_MARK8("sphere_end") }
                    else { if (COUNT) ntri[shray]++; hit = tri_test(g, o, d, tbest, t); }
                    if (COUNT) { const bool any_ = crt_ballot(hit) != 0ull; if (lane == 0 && any_) itc[6]++; }
                    CRT_MARK8("hit_begin")
                    if (hit) {
                        if (!shray) {
                            if (t < tbest || (t == tbest && best >= 0 && A.sv.rank[p] < A.sv.rank[best])) { tbest = t; best = (int)p; }
                        } else {
                            if (lkind != kShadowDirectional) {
                                /* point light / area sample: the reference takes the nearest hit t0 and calls the point
                                 * shadowed iff dtoLight > |(o + t0 d) - o| (Ray.cu:625-629; area samples also need
                                 * t0 > 1e-9, Ray.cu:589).  Any hit t >= t0 that lies before the light by a margin far
                                 * above the rounding of that expression decides the same without knowing t0 */
                                const bool area = (lkind >> 31) != 0u;
                                if (fstate & 1u) {            /* area sample known to be occluded: only a hit at t <= 1e-9 lights it */
                                    if (!(t > 0.000000001f)) { fstate &= ~1u; best = -1; finish = true; }
                                } else {
                                    if (t < tbest) { tbest = t; best = (int)p; }
                                    if (area && !(t > 0.000000001f)) { best = -1; finish = true; }     /* t0 <= 1e-9: lit */
                                    else {
                                        const vec3 ip = o + t * d;
                                        const float gx = ip.x - o.x, gy = ip.y - o.y, gz = ip.z - o.z;
                                        /* |(o + t0 d) - o| for any t0 in [-1e-4, t] is below this bound: relative slack for |d| and the
                                         * square root, absolute slack for the rounding of o + t d (a few ulp of |o| + t) and for t0 < 0 */
                                        const float slack = 2e-6f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z)) + 2e-4f;
                                        if (sqrtf(gx * gx + gy * gy + gz * gz) * 1.00001f + slack < __uint_as_float(lkind & 0x7fffffffu)) {
                                            if (COUNT) nfound++;
                                            fstate |= 1u;
                                            if (area) { if (tbest > 0.000000001f) tbest = 0.000000001f; }
                                            else finish = true;
                                        }
                                    }
                                }
                            } else {
                                if (!(t > 0.000000001f)) { fstate = (fstate & ~1u) | 2u; finish = true; }
                                else { if (COUNT && !(fstate & 1u)) nfound++; fstate |= 1u; if (tbest > 0.000000001f) tbest = 0.000000001f; }
                            }
                        }
                    }
                    CRT_MARK8("hit_end")
                    if (lp == lend && plm < (1u << 24) && (gmask & 0xffu) == 0u && sp == 0) finish = true;
                }
                CRT_MARK8("prim_end")
            }
            CRT_MARK8("finish_begin")
            if (COUNT) { const bool any_ = crt_ballot(finish) != 0ull; if (lane == 0 && any_) itc[2]++; }
            if (finish) {
                if (TAIL && (fstate & (1u << 24))) {
                    /* the ray is shared: fold this lane's part into the owner's record; the last lane to finish writes the result */
                    const uint32_t oi = (uint32_t)wbase + ((fstate >> 8) & 63u);
                    const bool shray = (ray >> 31) != 0u;
                    unsigned long long key;
                    if (!shray) {
                        key = ~0ull;
                        if (best >= 0) {
                            uint32_t tb = __float_as_uint(tbest);
                            if (tb == 0x80000000u) tb = 0u;                        /* -0 == +0 for the '<' of Ray.cu:199,216: one key, the rank decides */
                            tb ^= (tb >> 31) ? 0xffffffffu : 0x80000000u;          /* unsigned order = float order */
                            key = ((unsigned long long)tb << 32) | (unsigned long long)A.sv.rank[best];
                        }
                    } else {
                        key = (fstate & 2u) ? 0ull : ((fstate & 1u) ? 1ull : 2ull);      /* directional: lit-override < occluder < nothing */
                    }
                    atomicMin(&s_key[oi], key);
                    __builtin_amdgcn_wave_barrier();
                    /* bit 31: the winner's t is -0 (the key holds +0 for it; slots are < 2^24) */
                    if (!shray && best >= 0 && s_key[oi] == key) s_best[oi] = (uint32_t)best | (__float_as_uint(tbest) == 0x80000000u ? 0x80000000u : 0u);
                    __builtin_amdgcn_wave_barrier();
                    if (atomicSub(&s_cnt[oi], 1u) == 1u) {
                        const unsigned long long kmin = s_key[oi];
                        if (!shray) {
                            if (kmin == ~0ull) { A.hitSlot[ray] = -1; A.hitT[ray] = 1000000000.0f; }
                            else {
                                uint32_t tb = (uint32_t)(kmin >> 32);
                                tb ^= (tb >> 31) ? 0x80000000u : 0xffffffffu;
                                const uint32_t sb = s_best[oi];
                                A.hitSlot[ray] = (int32_t)(sb & 0x7fffffffu);
                                A.hitT[ray] = __uint_as_float(tb | (sb & 0x80000000u));
                            }
                        } else if (kmin == 1ull) A.shflag[ray & 0x7fffffffu] = 1;
                    }
                } else if constexpr (TAIL) write_result();
                else fstate |= kZombie;          /* main loop: the result is written when the lane takes its next ray (refill2) */
                active = false; fstate |= kIdle; lp = lend = 0; plm = 0;
            }
            CRT_MARK8("finish_end")
        }
        CRT_MARK8("refill2_begin")
        if (COUNT) { const bool any_ = crt_ballot(pend) != 0ull; if (lane == 0 && any_) itc[3]++; }
        /* ---- refill, second half: ray set-up for the lanes whose loads were issued above ---- */
        if constexpr (!TAIL)
        if (pend) {
            if (fstate & kZombie) write_result();
            const float4 ro = pro4, rd = prd;
            const bool shray = pend_sh;
            const uint32_t idx = __float_as_uint(ro.w);      /* nearest-hit ray: its own queue index; shadow ray: the flag it may set */
            if (!shray && __float_as_uint(rd.w) == 0xffffffffu) {
                A.hitSlot[idx] = -3;
                A.hitT[idx] = 0.f;
                fstate &= ~kZombie;
            } else {
                ray = idx;
                o = vec3(ro.x, ro.y, ro.z); d = vec3(rd.x, rd.y, rd.z);
                if (shray) { ray = idx | 0x80000000u; lkind = __float_as_uint(rd.w); }
                /* v_rcp_f32 (1 ulp) instead of an IEEE division: the box test only has to be
                 * conservative and the pad below is 1e4 times that error */
                float rx = __builtin_amdgcn_rcpf(d.x), ry = __builtin_amdgcn_rcpf(d.y), rz = __builtin_amdgcn_rcpf(d.z);
                if (!(fabsf(d.x) > 1e-20f)) rx = copysignf(1e20f, d.x);
                if (!(fabsf(d.y) > 1e-20f)) ry = copysignf(1e20f, d.y);
                if (!(fabsf(d.z) > 1e-20f)) rz = copysignf(1e20f, d.z);
                ix = rx; iy = ry; iz = rz;
                /* boxes are conservative already; the pad covers the rounding of the plane arithmetic.  A ray that starts further out
                 * than the sphere boxes were padded for (layout.h sphere_box_half) carries the rest of that pad itself */
                const float far_out = kSphereNoiseSqrtK * fmaxf(__builtin_amdgcn_sqrtf(o.x * o.x + o.y * o.y + o.z * o.z) * 1.00001f - A.origin_bound, 0.f);      /* v_sqrt_f32: 1 ulp */
                const float px = 1e-3f + 1e-6f * (fabsf(o.x) + A.scene_abs) + far_out;
                const float py = 1e-3f + 1e-6f * (fabsf(o.y) + A.scene_abs) + far_out;
                const float pz = 1e-3f + 1e-6f * (fabsf(o.z) + A.scene_abs) + far_out;
                const bool ngx = ix < 0.f, ngy = iy < 0.f, ngz = iz < 0.f;
                /* near plane: lo row shifted by -pad for a positive direction, hi row by +pad for a negative one */
                anx = (ngx ? o.x - px : o.x + px) * ix; afx = (ngx ? o.x + px : o.x - px) * ix;
                any_ = (ngy ? o.y - py : o.y + py) * iy; afy = (ngy ? o.y + py : o.y - py) * iy;
                anz = (ngz ? o.z - pz : o.z + pz) * iz; afz = (ngz ? o.z + pz : o.z - pz) * iz;
                R = (ngx ? 4u : 0u) | (ngy ? 2u : 0u) | (ngz ? 1u : 0u);
                selx = ngx ? 0x0c030c02u : 0x0c070c06u; sely = ngy ? 0x0c030c02u : 0x0c070c06u; selz = ngz ? 0x0c030c02u : 0x0c070c06u;
                tbest = 1000000000.0f; best = -1; sp = 0; fstate = (uint32_t)lane << 8; lp = lend = 0; plm = 0;
                gbase = 0; gmask = (1u << 8) | (1u << R);      /* a group whose only child is the root */
                active = true;
            }
        }
        CRT_MARK8("refill2_end")
        if constexpr (TAIL) return crt_ballot(active) == 0ull;
        else return exhausted;
    };
    while (!iteration(std::false_type{})) {}
    if (fstate & kZombie) { write_result(); fstate &= ~kZombie; }      /* the queue is dry: nothing will be set up on this lane again */
    unsigned long long t_dry = 0;
    if (A.tail_probe) t_dry = __builtin_readcyclecounter();      /* s_memtime; CRT_TAIL_PROBE=1 only */
    CRT_MARK("tail_begin")
    if (crt_ballot(active) != 0ull)
        while (!iteration(std::true_type{})) {}
    CRT_MARK("tail_end")
    if (A.tail_probe && lane == 0) {   /* [30] the longest, [31] the sum of the waves' tails (queue dry -> wave done), in s_memtime ticks */
        const unsigned long long dt = __builtin_readcyclecounter() - t_dry;
        atomicMax(&A.counters[30], dt);
        atomicAdd(&A.counters[31], dt);
    }
#undef CRT_MARK8
    if (COUNT) {
        for (int m = 0; m < 2; m++) {          /* nearest-hit counters at [0..3], shadow counters at [16..19] */
            atomicAdd(&A.counters[16 * m + 0], nbox[m]);
            atomicAdd(&A.counters[16 * m + 1], ntri[m]);
            atomicAdd(&A.counters[16 * m + 2], nsph[m]);
            atomicAdd(&A.counters[16 * m + 3], nvis[m]);
        }
        if (lane == 0) for (int i = 0; i < 5; i++) atomicAdd(&A.counters[8 + i], sch[i]);
        atomicAdd(&A.counters[13], nzero);
        atomicAdd(&A.counters[14], nstale);      /* node visits of shadow rays that already hold a hit with t > 1e-9 */
        atomicAdd(&A.counters[15], nfound);
        atomicAdd(&A.counters[28], nzero_t);        /* visits that hit no child only because of the t limit (stale stack entries) */
        atomicAdd(&A.counters[29], nculled_t);      /* child boxes the t limit rejected */
        if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&A.counters[20 + i], itc[i]);
    }
}

int trace_grid_blocks(int device) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 1024;
    /* persistent grid = what is resident at once: limited by the LDS stack per block and by the
     * kernel's VGPR allocation; ask the runtime for the tighter of the two */
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace8<false, 256, true>, 256, 0) != hipSuccess || per_cu < 1)
        per_cu = 4;
    if (per_cu > 8) per_cu = 8;
    return p.multiProcessorCount * per_cu;
}

/* which block sizes this build holds: 256, and 128 / 64 with -DCRT_TRACE_BLOCK_VARIANTS (an experiment, see k_trace8) */
bool trace8_has_block(int block) {
#ifdef CRT_TRACE_BLOCK_VARIANTS
    return block == 256 || block == 128 || block == 64;
#else
    return block == 256;
#endif
}

template <int BLOCK>
static void launch_trace8(const TraceArgs& a, bool counting, int blocks256, hipStream_t s) {
    dim3 g((unsigned)(blocks256 * (256 / BLOCK))), b(BLOCK);      /* the same number of waves whatever the block size */
    static const bool force_ovf = getenv("CRT_TRACE8_FORCE_OVF") != nullptr;      /* tests: the variant with the overflow code on a shallow tree */
    if (counting) hipLaunchKernelGGL((k_trace8<true, BLOCK, true>), g, b, 0, s, a);
    else if (a.ovf_depth > 0 || BLOCK != 256 || force_ovf) hipLaunchKernelGGL((k_trace8<false, BLOCK, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_trace8<false, BLOCK, BLOCK != 256>), g, b, 0, s, a);      /* 256 threads, shallow tree: no overflow code */
}

void launch_trace(const TraceArgs& a_in, int mode, bool counting, bool wide, int blocks, hipStream_t s) {
    dim3 g(blocks), b(kTraceBlock);
    TraceArgs a = a_in;
    /* tail work sharing of k_trace8; CRT_STEAL="min_idle,every" overrides the built-in setting (tuning experiments) */
    static const int steal[2] = {[] { int i = kStealMinIdle, e = kStealEvery; if (const char* v = getenv("CRT_STEAL")) sscanf(v, "%d,%d", &i, &e); return i < 1 ? 1 : (i > 64 ? 65 : i); }(),
                                 [] { int i = kStealMinIdle, e = kStealEvery; if (const char* v = getenv("CRT_STEAL")) sscanf(v, "%d,%d", &i, &e); int p2 = 1; while (p2 < e && p2 < 64) p2 <<= 1; return p2; }()};
    a.steal_min_idle = steal[0]; a.steal_every = steal[1];
    static const int probe = getenv("CRT_TAIL_PROBE") ? 1 : 0;
    a.tail_probe = probe;
    if (wide) {          /* k_trace8 serves both of its queues in one launch; `mode` is not used */
#ifdef CRT_TRACE_BLOCK_VARIANTS
        if (a.block == 64) launch_trace8<64>(a, counting, blocks, s);
        else if (a.block == 128) launch_trace8<128>(a, counting, blocks, s);
        else
#endif
        launch_trace8<256>(a, counting, blocks, s);
    } else {
        if (mode == TRACE_NEAREST) {
            if (counting) hipLaunchKernelGGL((k_trace<TRACE_NEAREST, true>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_trace<TRACE_NEAREST, false>), g, b, 0, s, a);
        } else {
            if (counting) hipLaunchKernelGGL((k_trace<TRACE_SHADOW, true>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_trace<TRACE_SHADOW, false>), g, b, 0, s, a);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * k_quantize8: Node8 (exact child bounds) -> QNode8.  Plane codes are integers 0..255 on a
 * power-of-two grid anchored at the node's own box, rounded outward and checked in double so that
 * every quantised box contains the exact one.  Runs after every build and refit of the wide tree.
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_quantize8(const Node8* nodes, uint32_t nnodes, QNode8* qnodes, uint32_t* status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnodes) return;
    const Node8& nd = nodes[i];
    QNode8 q;
    uint32_t imask = 0, vmask = 0, cbase = 0xffffffffu, pb = 0xffffffffu;
    for (int k = 0; k < 8; k++) {
        const uint32_t l = nd.child[k];
        if (l == kChildEmpty) continue;
        vmask |= 1u << k;
        if (l & kLeafFlag) { const uint32_t st = l & kLeafStartMask; if (st < pb) pb = st; }
        else { imask |= 1u << k; if (l < cbase) cbase = l; }
    }
    if (cbase == 0xffffffffu) cbase = 0;
    if (pb == 0xffffffffu) pb = 0;
    bool bad = false;
    uint32_t nint = 0;
    for (int k = 0; k < 8; k++) {
        const uint32_t l = nd.child[k];
        q.meta[k] = 0;
        if (l == kChildEmpty) continue;
        if (l & kLeafFlag) {
            const uint32_t off = (l & kLeafStartMask) - pb, cnt = ((l >> kLeafCountShift) & 63u) + 1u;
            if (off > 31u || cnt > (uint32_t)kWide8LeafMax) bad = true;
            q.meta[k] = (uint8_t)(0x80u | ((cnt - 1u) << 5) | (off & 31u));
        } else {
            if (l != cbase + nint) bad = true;          /* internal children must be consecutive in slot order */
            nint++;
        }
    }
    float org[3];
    uint32_t ex[3];
    for (int a = 0; a < 3; a++) {
        const float* clo = a == 0 ? nd.lox : a == 1 ? nd.loy : nd.loz;
        const float* chi = a == 0 ? nd.hix : a == 1 ? nd.hiy : nd.hiz;
        uint8_t* qlo = a == 0 ? q.lo_x : a == 1 ? q.lo_y : q.lo_z;
        uint8_t* qhi = a == 0 ? q.hi_x : a == 1 ? q.hi_y : q.hi_z;
        float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
        for (int k = 0; k < 8; k++) if (vmask & (1u << k)) { lo = fminf(lo, clo[k]); hi = fmaxf(hi, chi[k]); }
        const float o = lo <= hi ? lo : 0.f;
        const float ext = lo <= hi ? hi - lo : 0.f;
        /* smallest power of two s with (kQ8Grid - 1) * s >= extent (one code of headroom for outward rounding) */
        int e = 0;
        (void)frexpf(ext / (float)(kQ8Grid - 1), &e);
        float sc = ldexpf(1.0f, e);
        if (!(ext > 0.f)) sc = ldexpf(1.0f, -60);
        /* the traversal reads a code q as kQ8Bias + q (0 since the codes became fp16 denormals, slab8 above; 1024 before), so the
         * stored origin is os = o - kQ8Bias s and a plane sits at os + (kQ8Bias + q) s; the containment check uses exactly that */
        float os = o;
        for (int attempt = 0; attempt < 4; attempt++) {
            bool fits = true;
            os = o - (float)kQ8Bias * sc;
            /* round the shifted origin DOWN: code 0 must not lie above the node's own lower bound */
            if ((double)os + (double)kQ8Bias * (double)sc > (double)o) os = __uint_as_float(os > 0.f ? __float_as_uint(os) - 1u : (os < 0.f ? __float_as_uint(os) + 1u : 0x80000001u));
            for (int k = 0; k < 8; k++) {
                uint32_t ql = (uint32_t)kQ8Grid, qh = 0;       /* empty slot: inverted */
                if (vmask & (1u << k)) {
                    float fl = floorf((clo[k] - o) / sc), fh = ceilf((chi[k] - o) / sc);
                    fl = fminf(fmaxf(fl, 0.f), (float)kQ8Grid); fh = fminf(fmaxf(fh, 0.f), (float)kQ8Grid);
                    ql = (uint32_t)fl; qh = (uint32_t)fh;
                    while (ql > 0 && (double)os + (double)(kQ8Bias + ql) * (double)sc > (double)clo[k]) ql--;
                    while (qh < (uint32_t)kQ8Grid && (double)os + (double)(kQ8Bias + qh) * (double)sc < (double)chi[k]) qh++;
                    if ((double)os + (double)(kQ8Bias + ql) * (double)sc > (double)clo[k] ||
                        (double)os + (double)(kQ8Bias + qh) * (double)sc < (double)chi[k]) fits = false;
                }
                qlo[k] = (uint8_t)ql; qhi[k] = (uint8_t)qh;
            }
            if (fits) break;
            sc *= 2.0f;
        }
        ex[a] = ((__float_as_uint(sc) >> 23) & 0xffu) + kQ8ScaleShift;      /* < 255: sc <= the scene's extent */
        if (ex[a] > 254u) bad = true;
        org[a] = os;
    }
    if (bad || cbase >= (1u << 24) || pb >= (1u << 24)) atomicOr(status, 2u);     /* both bases share a word with an 8-bit mask */
    q.ox = org[0]; q.oy = org[1]; q.oz = org[2];
    q.exps_imask = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (imask << 24);
    q.base_vmask = cbase | (vmask << 24);
    q.prim_base = pb;
    qnodes[i] = q;
}
void launch_quantize(const Node8* nodes, uint32_t nnodes, QNode8* qnodes, uint32_t* status, hipStream_t s) {
    if (nnodes) hipLaunchKernelGGL(k_quantize8, dim3((nnodes + 255u) / 256u), dim3(256), 0, s, nodes, nnodes, qnodes, status);
}

/* ------------------------------------------------------------------------------------------
 * optional ray sorting (the reference's own sketch is the nw/ne/sw/se counters of
 * Ray.cu:526-541): a counting sort of queue positions by the Morton code of the origin's cell.
 * Only the visiting order changes (TraceArgs::perm); every ray still writes its result at its own
 * queue index, so images are identical with and without it.
 * ------------------------------------------------------------------------------------------ */
CRT_DEV int lane_id() { return (int)(threadIdx.x & 63u); }
CRT_DEV uint32_t spread3(uint32_t v) {        /* 5 bits -> every third bit */
    v = (v | (v << 8)) & 0x0000f00fu;
    v = (v | (v << 4)) & 0x000c30c3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
__global__ __launch_bounds__(256) void k_sort_count(const float4* rayO, const uint32_t* count, const float* box, uint32_t* keys, uint32_t* hist) {
    const uint32_t total = *count;
    const float lx = box[0], ly = box[2], lz = box[4];
    const float sx = (float)(1 << kSortBits) / fmaxf(box[1] - box[0], 1e-20f), sy = (float)(1 << kSortBits) / fmaxf(box[3] - box[2], 1e-20f),
                sz = (float)(1 << kSortBits) / fmaxf(box[5] - box[4], 1e-20f);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const float4 o = rayO[i];
        const int m = (1 << kSortBits) - 1;
        int cx = (int)((o.x - lx) * sx), cy = (int)((o.y - ly) * sy), cz = (int)((o.z - lz) * sz);
        cx = cx < 0 ? 0 : (cx > m ? m : cx); cy = cy < 0 ? 0 : (cy > m ? m : cy); cz = cz < 0 ? 0 : (cz > m ? m : cz);
        const uint32_t key = spread3((uint32_t)cx) | (spread3((uint32_t)cy) << 1) | (spread3((uint32_t)cz) << 2);
        keys[i] = key;
        /* neighbouring rays mostly share a cell: one atomic per distinct key of the wave, not one per lane
         * (same-address atomics serialise) */
        unsigned long long todo = crt_ballot(1);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t k = (uint32_t)__shfl((int)key, leader);
            const unsigned long long same = crt_ballot(key == k);
            if (lane_id() == leader) atomicAdd(&hist[k], (uint32_t)__popcll(same));
            todo &= ~same;
            if (key == k) break;
        }
    }
}
/* exclusive scan of the kSortBins counters in place: one block, each thread owns a run of bins */
__global__ __launch_bounds__(1024) void k_sort_scan(uint32_t* hist) {
    __shared__ uint32_t s_sum[1024];
    constexpr int per = kSortBins / 1024;
    uint32_t local[per];
    uint32_t acc = 0;
    for (int k = 0; k < per; k++) { local[k] = acc; acc += hist[threadIdx.x * per + k]; }
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= (unsigned)off ? s_sum[threadIdx.x - off] : 0u;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t base = threadIdx.x ? s_sum[threadIdx.x - 1] : 0u;
    for (int k = 0; k < per; k++) hist[threadIdx.x * per + k] = base + local[k];
}
__global__ __launch_bounds__(256) void k_sort_scatter(const uint32_t* keys, const uint32_t* count, uint32_t* cursor, uint32_t* perm) {
    const uint32_t total = *count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t key = keys[i];
        unsigned long long todo = crt_ballot(1);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t k = (uint32_t)__shfl((int)key, leader);
            const unsigned long long same = crt_ballot(key == k);
            uint32_t base = 0;
            if (lane_id() == leader) base = atomicAdd(&cursor[k], (uint32_t)__popcll(same));
            base = (uint32_t)__shfl((int)base, leader);
            todo &= ~same;
            if (key == k) { perm[base + (uint32_t)__popcll(same & ((1ull << lane_id()) - 1ull))] = i; break; }
        }
    }
}
void launch_sort_rays(const float4* rayO, const uint32_t* count, uint32_t cap, const float* root_box6, uint32_t* keys,
                      uint32_t* hist, uint32_t* perm, hipStream_t s) {
    (void)hipMemsetAsync(hist, 0, sizeof(uint32_t) * kSortBins, s);
    uint32_t blocks = (cap + 255u) / 256u;
    if (blocks > 8192u) blocks = 8192u;
    if (blocks < 1u) blocks = 1u;
    hipLaunchKernelGGL(k_sort_count, dim3(blocks), dim3(256), 0, s, rayO, count, root_box6, keys, hist);
    hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, s, hist);
    hipLaunchKernelGGL(k_sort_scatter, dim3(blocks), dim3(256), 0, s, keys, count, hist, perm);
}

/* ------------------------------------------------------------------------------------------
 * wave-aggregated queue append: one atomic per wave
 * ------------------------------------------------------------------------------------------ */
CRT_DEV uint32_t queue_slot(bool want, uint32_t* counter) {
    const unsigned long long m = crt_ballot(want);
    if (m == 0ull) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

/* ------------------------------------------------------------------------------------------
 * k_gen_primary: GeneratePrimaryRays (Ray.cu:31-52).  Path q of this rank -> tile -> 8x8 pixel
 * block per wave.  The reference's literal 720 (Ray.cu:37-38) becomes `width`.
 * ------------------------------------------------------------------------------------------ */
/* Ray.cu:39-47 for pixel (px, py): shared by k_gen_primary and by k_shade, which recomputes the primary ray of a path from
 * its pixel instead of reading it back (same expressions, same order, no contraction: the same bits) */
CRT_DEV vec3 primary_direction(const FrameParams& P, int px, int py) {
    const vec3 L(P.L), u(P.u), v(P.v), cam(P.cam);
    vec3 pix;
    if (P.spp_n <= 1) {
        pix = (L + u * float(px) + v * float(py));
    } else {        /* stratified sub-pixel offsets along u and v: an extension, SURVEY.md 8(d) config 5 */
        const float ox = ((float)(P.sample % P.spp_n) + .5f) / (float)P.spp_n - .5f;
        const float oy = ((float)(P.sample / P.spp_n) + .5f) / (float)P.spp_n - .5f;
        pix = (L + u * (float(px) + ox) + v * (float(py) + oy));
    }
    const float xoffset = 0, yoffset = 0;
    return normalize(vec3(pix.x + xoffset, pix.y + yoffset, pix.z) - cam);
}

__global__ __launch_bounds__(256) void k_gen_primary(FrameParams P, FrameBuffers B, uint32_t* count) {
    const uint32_t per_tile = (uint32_t)(P.tile_w * P.tile_h);
    const uint32_t total = (uint32_t)P.ntiles_owned * per_tile;
    const uint32_t stride = gridDim.x * blockDim.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) *count = total;
    /* queue slot == path index: no compaction, no atomics.  Pixels of a clipped tile that fall
     * outside the frame leave a flagged hole that k_trace and k_shade skip. */
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const uint32_t tile = B.tile_ids[q / per_tile];
        const uint32_t r = q % per_tile;
        const uint32_t blk = r >> 6, l = r & 63u;
        const uint32_t bpr = (uint32_t)P.tile_w >> 3;
        const int px = (int)((tile % (uint32_t)P.tiles_x) * (uint32_t)P.tile_w + (blk % bpr) * 8u + (l & 7u));
        const int py = (int)((tile / (uint32_t)P.tiles_x) * (uint32_t)P.tile_h + (blk / bpr) * 8u + (l >> 3));
        if (!(px < P.width && py < P.height)) {
            B.rayO[q] = make_float4(0.f, 0.f, 0.f, __uint_as_float(q));
            B.rayD[q] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xffffffffu));
            B.pix[q] = 0xffffffffu;
            continue;
        }
        const vec3 cam(P.cam);
        const vec3 sample = primary_direction(P, px, py);
        B.rayO[q] = make_float4(cam.x, cam.y, cam.z, __uint_as_float(q));      /* .w: the ray's own queue index (k_trace8 takes it from here) */
        B.rayD[q] = make_float4(sample.x, sample.y, sample.z, 0.f);
        B.pix[q] = (uint32_t)(py * P.width + px);
    }
}

void launch_gen_primary(const FrameParams& fp, const FrameBuffers& fb, uint32_t* count, hipStream_t s) {
    const uint64_t total = (uint64_t)fp.ntiles_owned * fp.tile_w * fp.tile_h;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_gen_primary, dim3(blocks), dim3(256), 0, s, fp, fb, count);
}

/* ------------------------------------------------------------------------------------------
 * k_shade
 * ------------------------------------------------------------------------------------------ */
struct WallDef { float n[3]; float p[3]; int kind; float diffuse[3]; float ambient[3]; };
/* Ray.cu:238-457: floor, left, front, back, right, ceiling -- tested in this order, first
 * in-bounds hit wins.  kind 0 bounds (z,x), 1 bounds (z,y), 2 bounds (x,y). */
__constant__ WallDef c_walls[6] = {
    {{0, 1, 0}, {0, -200.0f, 0}, 0, {1.0f, 0.2f, 0.2f}, {1.0f, 0.2f, 0.2f}},
    {{1, 0, 0}, {-200.0f, 0, 0}, 1, {0.2f, 0.2f, 1.0f}, {0.2f, 0.2f, 1.0f}},
    {{0, 0, 1}, {0, 0, -300.0f}, 2, {0.0f, 1.0f, 0.0f}, {0.0f, 1.0f, 0.0f}},
    {{0, 0, -1}, {0, 0, 300.0f}, 2, {1.0f, 1.0f, 0.0f}, {1.0f, 1.0f, 0.0f}},
    {{-1, 0, 0}, {200.0f, 0, 0}, 1, {0.2f, 0.2f, 1.0f}, {0.2f, 0.2f, 1.0f}},
    {{0, -1, 0}, {0, 200.0f, 0}, 0, {0.9f, 0.9f, 0.9f}, {1.0f, 1.0f, 1.0f}},
};

/* ------------------------------------------------------------------------------------------
 * powf as the host's libm computes it.  Shade's specular term is pow(dot, shininess) on floats (Ray.cu:655); the parity
 * target (oracle/, and any host build of the reference) evaluates it with glibc's powf.  Until round 2 the kernel called
 * the device's double-precision pow and rounded -- correct to the last bit in practice, and most of k_shade's instructions.
 * This is glibc 2.35's algorithm restated (sysdeps/ieee754/flt-32/e_powf.c: log2 by a 16-entry table and a degree-5
 * polynomial, exp2 by a 32-entry table and a cubic, all in double), with the multiply-adds fused where the x86-64 FMA build
 * of libm fuses them: bit-identical to the host's powf on 10^8 random (dot, shininess) pairs, subnormal results included
 * (tools/exp/powf_check.c; the unfused variant differs once in 10^8), and on the special values (zeros, infinities, NaNs,
 * subnormal bases, negative bases) whose branches are restated too.  Measured: k_shade is not bound by it either way
 * (1.66 ms per frame with the device's pow, 1.60 with no pow at all, 1.67 with this) -- the point is that the term is now
 * the parity target's own arithmetic instead of "correct to the last bit in practice".
 * ------------------------------------------------------------------------------------------ */
/* constants of glibc 2.35 powf (sysdeps/ieee754/flt-32/e_powf.c, powf_log2_data.c, exp2f_data.c; S. Nagy, ARM optimized routines) */
static __device__ const double d_powlog2_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2},
    {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2},
    {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2},
    {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3},
    {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},
    {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},
    {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
    {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
static __device__ const double d_powlog2_poly[5] = {0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp+0};
static __device__ const unsigned long long d_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
static __device__ const double d_exp2f_poly[3] = {0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1};

/* ltab / etab: the two tables, in LDS for k_shade (a global gather costs two dependent cache round trips per call) */
CRT_DEV int powf_checkint(uint32_t iy) {          /* 0: not an integer, 1: odd, 2: even (e_powf.c checkint) */
    const int e = (int)(iy >> 23 & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    return (iy & (1u << (0x7f + 23 - e))) ? 1 : 2;
}
CRT_DEV float powf_libm(float x, float y, const double (*ltab)[2] = d_powlog2_tab, const unsigned long long* etab = d_exp2f_tab) {
    uint32_t ix = __float_as_uint(x), sign_bias = 0;
    const uint32_t iy = __float_as_uint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || 2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
        /* either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan): e_powf.c's special cases, in its order */
        if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
            if (2u * iy == 0u) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;      /* |x| < 1 && y == inf, or |x| > 1 && y == -inf */
            return y * y;
        }
        if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && powf_checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {                       /* finite x < 0: needs an integer y */
            const int yint = powf_checkint(iy);
            if (yint == 0) return __uint_as_float(0x7fc00000u);
            if (yint == 1) sign_bias = 1u << (5 + 11);
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                       /* subnormal x: normalise, the exponent goes negative */
            ix = __float_as_uint(__uint_as_float(ix) * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    /* log2_inline: x = 2^k z, z in [OFF, 2 OFF), c near the centre of z's subinterval: log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k */
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> (23 - 4)) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const int k = (int32_t)top >> 23;
    const double z = (double)__uint_as_float(ix - top);
    const double r = fma(z, ltab[i][0], -1.0);
    const double y0 = ltab[i][1] + (double)k;
    const double r2 = r * r;
    double yy = fma(d_powlog2_poly[0], r, d_powlog2_poly[1]);
    const double p = fma(d_powlog2_poly[2], r, d_powlog2_poly[3]);
    const double r4 = r2 * r2;
    double q = fma(d_powlog2_poly[4], r, y0);
    q = fma(p, r2, q);
    yy = fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    if (((unsigned long long)__double_as_longlong(ylogx) >> 47 & 0xffffull) >= ((unsigned long long)__double_as_longlong(126.0) >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return __uint_as_float((sign_bias ? 0x80000000u : 0u) | 0x7f800000u);
        if (ylogx <= -150.0) return __uint_as_float(sign_bias ? 0x80000000u : 0u);
    }
    /* exp2_inline: ylogx = k/32 + r, |r| <= 1/64; 2^(k/32) from the table with the exponent added into its bits */
    const double shift = 0x1.8p+47;
    double kd = ylogx + shift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= shift;
    const double rr = ylogx - kd;
    unsigned long long t = etab[ki & 31ull];
    t += (ki + sign_bias) << (52 - 5);
    const double sc = __longlong_as_double((long long)t);
    const double zz = fma(d_exp2f_poly[0], rr, d_exp2f_poly[1]);
    const double rr2 = rr * rr;
    double w = fma(d_exp2f_poly[2], rr, 1.0);
    w = fma(zz, rr2, w);
    w = w * sc;
    return (float)w;
}

constexpr int kShadeMatsLds = 64;
/* Threads per block of k_shade.  Every block iteration reserves its queue space with ONE returning atomic per queue, and those
 * atomics are what the kernel waits for: all resident blocks queue on the same two words, a round trip is (blocks in the queue) x
 * ~12 ns, so the launch lasts (block iterations) x ~12 ns whatever else it does.  128 / 256 / 384 / 512 / 640 / 768 / 1024 threads:
 * 2.51 / 1.49 / 1.50 / 1.32 / 1.57 / 1.34 / 1.58 ms of k_shade + k_resolve per 4K frame (6 waves per SIMD = 24 per CU: 512 and 768 fill them) */
#ifndef CRT_SHADE_BLOCK
#define CRT_SHADE_BLOCK 512
#endif
constexpr int kShadeBlock = CRT_SHADE_BLOCK;
#ifndef CRT_SHADE_MIN_WAVES
#define CRT_SHADE_MIN_WAVES 0          /* no floor: 74 VGPRs (materials from LDS, no SLP vectorizer), 6 waves per SIMD, no scratch */
#endif
#if CRT_SHADE_MIN_WAVES > 0
#define CRT_SHADE_BOUNDS __launch_bounds__(kShadeBlock, CRT_SHADE_MIN_WAVES)
#else
#define CRT_SHADE_BOUNDS __launch_bounds__(kShadeBlock)
#endif
__global__ CRT_SHADE_BOUNDS void k_shade(FrameParams P, FrameBuffers B, SceneView sv, LightsParam LP,
                                                const uint32_t* count, uint32_t* next_count, uint32_t* shadow_count) {
    /* queue appends are reserved once per BLOCK iteration (LDS partial sums + one atomic per
     * queue): per-wave atomics on the two counters made this kernel atomic-bound */
    __shared__ uint32_t s_cnt[2][kShadeBlock / 64];
    __shared__ uint32_t s_base[2];
    __shared__ double s_ltab[16][2];
    __shared__ unsigned long long s_etab[32];
    /* the first kShadeMatsLds materials, in LDS: the material is the third of three dependent gathers per path (hit slot -> shading
     * record -> material), and scenes have few materials */
    __shared__ float4 s_mat[kShadeMatsLds][4];
    static_assert(sizeof(Material) == 64, "four float4 per material");
    {
        const uint32_t nm4 = (sv.nmats < (uint32_t)kShadeMatsLds ? sv.nmats : (uint32_t)kShadeMatsLds) * 4u;
        for (uint32_t t = threadIdx.x; t < nm4; t += blockDim.x) s_mat[t >> 2][t & 3u] = reinterpret_cast<const float4*>(sv.mats)[t];
    }
    if (threadIdx.x < 32) { reinterpret_cast<double*>(s_ltab)[threadIdx.x] = reinterpret_cast<const double*>(d_powlog2_tab)[threadIdx.x]; s_etab[threadIdx.x] = d_exp2f_tab[threadIdx.x]; }
    __syncthreads();
    const uint32_t total = *count;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounded = (total + (uint32_t)kShadeBlock - 1u) / (uint32_t)kShadeBlock * (uint32_t)kShadeBlock;
    const int raytype = P.bounce > 0 ? 1 : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += stride) {
        const int slot_i = i < total ? B.hitSlot[i] : -3;
        const bool valid = slot_i != -3;                        /* -3: hole left by a clipped tile (k_resolve skips it by its pixel) */
        bool dead = !valid;
        vec3 o, d, cin, P_, N, incident, rc, newcoef, diffuse, specular;
        float shininess = 0.f;
        uint32_t pix = 0;
        if (valid) {
            pix = B.pix[i];
            if (raytype == 0) {
                /* a primary ray is a function of its pixel (Ray.cu:31-52): recomputed, not read back -- 48 B per path less */
                o = vec3(P.cam);
                d = primary_direction(P, (int)(pix % (uint32_t)P.width), (int)(pix / (uint32_t)P.width));
            } else {
                const float4 ro = B.rayO[i], rd = B.rayD[i], ci = B.coef[i];
                o = vec3(ro.x, ro.y, ro.z); d = vec3(rd.x, rd.y, rd.z); cin = vec3(ci.x, ci.y, ci.z);
            }
            const int slot = slot_i;
            int cap_id = -1;
            if (slot >= 0) {
                /* RayIntersection hit branch, Ray.cu:479-511, with the full intersect functions */
                const GeomRec g = load_geom(sv.geom, (uint32_t)slot);
                const float4* sp4 = reinterpret_cast<const float4*>(sv.shade + slot);
                const float4 s0 = sp4[0], s1 = sp4[1], s2 = sp4[2];
                vec3 X;
                if (is_sphere(g)) {
                    float t, tsm;
                    sphere_test(g, o, d, t, tsm);
                    const vec3 c(g.a.x, g.a.y, g.a.z);
                    X = o + tsm * d;                                  /* Ray.cu:135 uses the smaller root */
                    N = normalize((X - c) / g.a.w);
                } else {
                    /* intersectTriangle, Ray.cu:150-173 */
                    const vec3 v1(g.a.x, g.a.y, g.a.z), v2(g.a.w, g.b.x, g.b.y), v3(g.b.z, g.b.w, g.c.x), ng(g.c.y, g.c.z, g.c.w);
                    const float denom = dot(ng, d);
                    const float t = dot((v1 - o), ng) / denom;
                    X = o + t * d;
                    const vec3 e0 = v2 - v1, e1 = v3 - v1, e2 = X - v1;
                    const float d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1), d20 = dot(e2, e0), d21 = dot(e2, e1);
                    const float den = d00 * d11 - d01 * d01;
                    const float vv = (d11 * d20 - d01 * d21) / den;
                    const float ww = (d00 * d21 - d01 * d20) / den;
                    const float uu = 1.0f - vv - ww;
                    const vec3 n1(s0.x, s0.y, s0.z), n2(s0.w, s1.x, s1.y), n3(s1.z, s1.w, s2.x);
                    N = normalize(uu * n1 + vv * n2 + ww * n3);
                }
                const uint32_t mi = __float_as_uint(s2.y);
                Material m;
                if (mi < (uint32_t)kShadeMatsLds) {
                    const float4 m0 = s_mat[mi][0], m1 = s_mat[mi][1], m2 = s_mat[mi][2], m3 = s_mat[mi][3];
                    m.ambient[0] = m0.x; m.ambient[1] = m0.y; m.ambient[2] = m0.z; m.diffuse[0] = m0.w; m.diffuse[1] = m1.x; m.diffuse[2] = m1.y;
                    m.specular[0] = m1.z; m.specular[1] = m1.w; m.specular[2] = m2.x; m.reflective[0] = m2.y; m.reflective[1] = m2.z; m.reflective[2] = m2.w;
                    m.shininess = m3.x;
                } else m = sv.mats[mi];
                const vec3 refl(m.reflective), amb(m.ambient);
                if (raytype == 0) {
                    newcoef = refl; rc = vec3(1.0f, 1.0f, 1.0f);
                    const vec3 c0 = 0.2f * amb;
                    B.accum[pix] = make_float4(c0.x, c0.y, c0.z, 0.f);
                } else { rc = cin; newcoef = cin * refl; }
                P_ = X + N * 0.7f;
                incident = d;
                diffuse = vec3(m.diffuse); specular = vec3(m.specular); shininess = m.shininess;
                cap_id = (int)sv.sidx[slot];
            } else {
                /* miss branch, Ray.cu:513-525 -> wallIntersection */
                int w = -1;
                for (int k = 0; k < 6 && w < 0; k++) {
                    const vec3 up(c_walls[k].n);
                    const float denom = dot(up, d);
                    if (fabsf(denom) > .0001f) {
                        const float t = dot((vec3(c_walls[k].p) - o), up) / denom;
                        if (t >= -0.0001f) {
                            const vec3 X = o + t * d;
                            const float x = 200.0f, y = 200.0f, z = 300.0f;
                            bool in;
                            if (c_walls[k].kind == 0) in = X.z > -z && X.z < z && X.x > -x && X.x < x;
                            else if (c_walls[k].kind == 1) in = X.z > -z && X.z < z && X.y < y && X.y > -y;
                            else in = X.x > -x && X.x < x && X.y < y && X.y > -y;
                            if (in) { w = k; P_ = X; N = up; }
                        }
                    }
                }
                if (w >= 0) {
                    if (raytype == 0) {
                        const vec3 c0 = 0.2f * vec3(c_walls[w].ambient);
                        B.accum[pix] = make_float4(c0.x, c0.y, c0.z, 0.f);
                        rc = vec3(1.0f, 1.0f, 1.0f);
                    } else rc = cin;
                    /* Ray.cu:257,262: nothing is reflected off a wall -- literally zero behind a primary ray, coefficient * 0 behind a
                     * reflected one, which is NaN once the coefficient has overflowed (reflective > 1 compounds) and keeps the path alive */
                    newcoef = raytype == 0 ? vec3(0.f, 0.f, 0.f) : cin * vec3(0.f, 0.f, 0.f);
                    incident = normalize(reflect(d, N));
                    diffuse = vec3(c_walls[w].diffuse); specular = vec3(0.f, 0.f, 0.f); shininess = 2;
                    cap_id = -2 - w;
                } else {
                    /* neither object nor wall: the reference shades an uninitialised Isect (NaN);
                     * defined here as end of path, colour unchanged */
                    dead = true;
                    if (raytype == 0) B.accum[pix] = make_float4(0.f, 0.f, 0.f, 0.f);
                    /* k_resolve learns that a path is dead from the first light's flag.  Without lights it would add coefficient * 0 for
                     * it -- NaN once the coefficient has overflowed (found by the randomised tests: two mirrors of reflectance 1e20,
                     * then nothing): the coefficient this entry was read from becomes zero, and 0 * 0 leaves the colour as it is */
                    else if (LP.n == 0) B.coef[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (P.capture) {
                B.cap_ids[(size_t)P.bounce * B.cap_stride + pix] = cap_id;
                B.cap_ts[(size_t)P.bounce * B.cap_stride + pix] = B.hitT[i];
            }
        }

        /* ---- reserve queue space: count this block's shadow and reflection rays ---- */
        /* a path whose coefficient is (0,0,0) adds 0 * colour from here on: it ends, unless a later colour could be non-finite
         * (0 * NaN is NaN in Ray.cu:660) -- P.keep_zero, decided per scene on the host (crt_params.zero_paths) */
        const bool emit = !dead && (P.bounce + 1 < P.bounces) && (P.keep_zero != 0 || newcoef.x != 0.f || newcoef.y != 0.f || newcoef.z != 0.f);
        uint32_t wave_sh = 0;
        for (int j = 0; j < LP.n; j++) {
            bool want = false;
            if (!dead) {
                if (LP.l[j].area || LP.l[j].point) want = true;
                else want = !((double)dot(normalize(vec3(LP.l[j].direction)), N) > -0.00001);
            }
            wave_sh += (uint32_t)__popcll(crt_ballot(want)) * (LP.l[j].area ? (uint32_t)LP.area_samples : 1u);
        }
        const unsigned long long emit_mask = crt_ballot(emit);
        if (lane == 0) { s_cnt[0][wv] = wave_sh; s_cnt[1][wv] = (uint32_t)__popcll(emit_mask); }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tsh = 0, trf = 0;
            for (int w = 0; w < kShadeBlock / 64; w++) { tsh += s_cnt[0][w]; trf += s_cnt[1][w]; }
            s_base[0] = tsh ? atomicAdd(shadow_count, tsh) : 0u;
            s_base[1] = trf ? atomicAdd(next_count, trf) : 0u;
        }
        __syncthreads();
        uint32_t sh_run = s_base[0], rf_base = s_base[1];
        for (int w = 0; w < wv; w++) { sh_run += s_cnt[0][w]; rf_base += s_cnt[1][w]; }

        /* ---- Shade's light loop, Ray.cu:560-659 ---- */
        vec3 direction, intersection = P_;
        if (!dead) direction = normalize(incident);
        for (int j = 0; j < LP.n; j++) {
            const crt_light& l = LP.l[j];
            /* an area light is a disk of l.radius in the XZ plane about l.position, sampled area_samples times
             * (Ray.cu:563-603, commented out there); every sample is shaded like an unattenuated point light */
            const int ns = l.area ? LP.area_samples : 1;
            const uint32_t ah = l.area ? area_hash(pix, (uint32_t)j, (uint32_t)P.bounce, (uint32_t)P.sample) : 0u;
            for (int k = 0; k < ns; k++) {
                const size_t plane = (size_t)(LP.slot_base[j] + k) * B.cap + i;
                bool want = false;
                vec3 toLight, sorg;
                uint32_t lbits = kShadowDirectional;
                if (!dead) {
                    vec3 reflectFromLight;
                    float distance;
                    uint8_t flag = 0;
                    if (l.area || l.point) {
                        vec3 lpos(l.position);
                        if (l.area) {
                            const float4 e = B.area_table[(ah + (uint32_t)k) & (uint32_t)(kAreaTable - 1)];
                            const float radius = e.x * l.radius;                  /* Ray.cu:569-575 */
                            lpos.x = lpos.x + radius * e.y;
                            lpos.z = lpos.z + radius * e.z;
                            distance = 1.0f;
                        } else {
                            const float ex = intersection.x - lpos.x, ey = intersection.y - lpos.y, ez = intersection.z - lpos.z;
                            const float dd = sqrtf(ex * ex + ey * ey + ez * ez);
                            distance = 1.0f / (l.constantTerm + l.linearTerm * dd + l.quadraticTerm * (dd * dd));
                            if (distance > 1.5f) distance = .5f;
                        }
                        toLight = normalize(lpos - intersection);
                        reflectFromLight = -toLight;
                        intersection = intersection + .01f * N;           /* cumulative, Ray.cu:585,620 */
                        /* dtoLight of Ray.cu:592,626, from the origin the shadow ray starts at */
                        const float fx = intersection.x - lpos.x, fy = intersection.y - lpos.y, fz = intersection.z - lpos.z;
                        lbits = __float_as_uint(sqrtf(fx * fx + fy * fy + fz * fz)) | (l.area ? 0x80000000u : 0u);
                        want = true;
                    } else {
                        distance = 1.0f;
                        toLight = -normalize(vec3(l.direction));
                        reflectFromLight = normalize(vec3(l.direction));
                        const float dotP = dot(reflectFromLight, N);
                        if ((double)dotP > -0.00001) flag = 1;            /* faces away: shadowed without a ray */
                        else want = true;
                    }
                    sorg = intersection;
                    const float sdot = dot(reflect(reflectFromLight, N), -direction);
                                        const float spec = powf_libm(sdot, shininess, s_ltab, s_etab);
                    const vec3 X = (.6f * diffuse * gmax(dot(toLight, N), 0.0f) + .2f * specular * spec);
                    /* a light that is shadowed without a ray contributes distance * 0 * lcolor * X (Ray.cu:629,655): exactly zero
                     * unless X or the light's colour is not finite, so the 16-byte record is written only where Ray.cu:660 can see it */
                    const bool lfinite = l.color[0] * 0.f + l.color[1] * 0.f + l.color[2] * 0.f == 0.f;
                    if (flag && lfinite && X.x - X.x == 0.f && X.y - X.y == 0.f && X.z - X.z == 0.f) flag = kShNoContribution;
                    else B.contrib[plane] = make_float4(X.x, X.y, X.z, distance);
                    B.shflag[plane] = flag;
                } else if (valid) B.shflag[plane] = kShDeadPath;
                const unsigned long long wm = crt_ballot(want);
                if (want) {
                    const uint32_t ks = sh_run + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
                    B.shO[ks] = make_float4(sorg.x, sorg.y, sorg.z, __uint_as_float((uint32_t)plane));
                    B.shD[ks] = make_float4(toLight.x, toLight.y, toLight.z, __uint_as_float(lbits));
                }
                sh_run += (uint32_t)__popcll(wm);
            }
        }

        /* ---- reflected ray, Ray.cu:494-496 / 519-520; paths whose coefficient is zero end here ---- */
        if (emit) {
            const uint32_t kn = rf_base + (uint32_t)__popcll(emit_mask & ((1ull << lane) - 1ull));
            const vec3 rdir = normalize(reflect(d, N));
            B.nrayO[kn] = make_float4(P_.x, P_.y, P_.z, __uint_as_float(kn));
            B.nrayD[kn] = make_float4(rdir.x, rdir.y, rdir.z, 0.f);
            B.ncoef[kn] = make_float4(newcoef.x, newcoef.y, newcoef.z, 0.f);
            B.npix[kn] = pix;
        }
        __syncthreads();     /* s_cnt / s_base are reused by the next iteration */
    }
}

void launch_shade(const FrameParams& fp, const FrameBuffers& fb, const SceneView& sv, const LightsParam& lights,
                  const uint32_t* count, uint32_t* next_count, uint32_t* shadow_count, hipStream_t s) {
    /* grid-stride over the queue; no more blocks than the queue can have entries */
    uint32_t blocks = (fb.cap + (uint32_t)kShadeBlock - 1u) / (uint32_t)kShadeBlock;
    /* the kernel is bound by the latency of its dependent gathers (hit slot -> geometry + shading record -> material), not by
     * bytes: more, shorter blocks keep more of them in flight (4096 blocks: 1.60 ms of k_shade + k_resolve per 4K frame,
     * 16384: 1.52; tools/exp/shade_grid.sh) */
    static const uint32_t maxb = [] { const char* v = getenv("CRT_SHADE_BLOCKS"); return v ? (uint32_t)atoi(v) : 16384u; }();
    blocks = blocks < 1u ? 1u : (blocks > maxb ? maxb : blocks);
    hipLaunchKernelGGL(k_shade, dim3(blocks), dim3(kShadeBlock), 0, s, fp, fb, sv, lights, count, next_count, shadow_count);
}

/* isectPoints[i].color += reflectionCoef * color (Ray.cu:660), colour summed over lights in order
 * with `distance * 0` for shadowed lights (Ray.cu:629,655) */
__global__ __launch_bounds__(256) void k_resolve(FrameBuffers B, LightsParam LP, const uint32_t* count, int bounce) {
    const uint32_t total = *count;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint32_t pix = B.pix[i];
        if (pix == 0xffffffffu) continue;              /* hole left by a clipped tile */
        vec3 color;
        bool deadpath = false;
        for (int j = 0; j < LP.n; j++) {
            const vec3 lcolor(LP.l[j].color);
            if (LP.l[j].area) {                       /* Ray.cu:565-601: mean over the samples */
                vec3 avgcolor;
                for (int k = 0; k < LP.area_samples; k++) {
                    const size_t plane = (size_t)(LP.slot_base[j] + k) * B.cap + i;
                    const uint8_t fl = B.shflag[plane];
                    if (fl == kShDeadPath) { deadpath = true; break; }
                    const float4 c = B.contrib[plane];
                    float distance = c.w;
                    if (fl) distance = distance * 0;
                    avgcolor += distance * lcolor * vec3(c.x, c.y, c.z);
                }
                color += avgcolor / (float)LP.area_samples;
            } else {
                const size_t plane = (size_t)LP.slot_base[j] * B.cap + i;
                const uint8_t fl = B.shflag[plane];
                if (fl == kShDeadPath) { deadpath = true; break; }
                if (fl == kShNoContribution) continue;            /* adds an exact zero */
                const float4 c = B.contrib[plane];
                float distance = c.w;
                if (fl) distance = distance * 0;
                color += distance * lcolor * vec3(c.x, c.y, c.z);
            }
        }
        if (deadpath) continue;                        /* neither object nor wall: the path ended, colour unchanged */
        /* isect.reflectionCoef: 1 for a primary hit, the running product carried by the ray otherwise (Ray.cu:482-491) */
        vec3 rc(1.0f, 1.0f, 1.0f);
        if (bounce > 0) { const float4 ci = B.coef[i]; rc = vec3(ci.x, ci.y, ci.z); }
        float4 a = B.accum[pix];
        vec3 acc(a.x, a.y, a.z);
        acc += rc * color;
        B.accum[pix] = make_float4(acc.x, acc.y, acc.z, 0.f);
    }
}

void launch_resolve(const FrameBuffers& fb, const LightsParam& lights, const uint32_t* count, int bounce, hipStream_t s) {
    uint32_t blocks = (fb.cap + 255u) / 256u;
    blocks = blocks < 1u ? 1u : (blocks > 4096u ? 4096u : blocks);
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, s, fb, lights, count, bounce);
}

/* ------------------------------------------------------------------------------------------
 * output conversion, Ray.cu:724-731: *255, clamp, truncate, BGR, row 0 = bottom
 * ------------------------------------------------------------------------------------------ */
CRT_DEV bool tile_pixel(const FrameParams& P, const uint32_t* tile_ids, uint32_t q, uint32_t& pixel) {
    const uint32_t per_tile = (uint32_t)(P.tile_w * P.tile_h);
    const uint32_t tile = tile_ids[q / per_tile];
    const uint32_t r = q % per_tile;
    const uint32_t px = (tile % (uint32_t)P.tiles_x) * (uint32_t)P.tile_w + r % (uint32_t)P.tile_w;
    const uint32_t py = (tile / (uint32_t)P.tiles_x) * (uint32_t)P.tile_h + r / (uint32_t)P.tile_w;
    pixel = py * (uint32_t)P.width + px;
    return px < (uint32_t)P.width && py < (uint32_t)P.height;
}

__global__ __launch_bounds__(256) void k_convert(FrameParams P, FrameBuffers B, const float4* sum, uint8_t* bgr) {
    const uint32_t total = (uint32_t)P.ntiles_owned * (uint32_t)(P.tile_w * P.tile_h);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        uint32_t pixel;
        if (!tile_pixel(P, B.tile_ids, q, pixel)) continue;
        vec3 c;
        if (sum) { const float4 s = sum[pixel]; const float inv = (float)(P.spp_n * P.spp_n); c = vec3(s.x / inv, s.y / inv, s.z / inv); }
        else { const float4 a = B.accum[pixel]; c = vec3(a.x, a.y, a.z); }
        c = c * 255.0f;
        c = vec3(gclamp(c.x, 0.0f, 255.0f), gclamp(c.y, 0.0f, 255.0f), gclamp(c.z, 0.0f, 255.0f));
        bgr[(size_t)pixel * 3] = (uint8_t)c.z;
        bgr[(size_t)pixel * 3 + 1] = (uint8_t)c.y;
        bgr[(size_t)pixel * 3 + 2] = (uint8_t)c.x;
    }
}
void launch_convert(const FrameParams& fp, const FrameBuffers& fb, const float4* sum, uint8_t* bgr, hipStream_t s) {
    hipLaunchKernelGGL(k_convert, dim3(2048), dim3(256), 0, s, fp, fb, sum, bgr);
}

__global__ __launch_bounds__(256) void k_spp_accumulate(FrameParams P, FrameBuffers B, float4* sum) {
    const uint32_t total = (uint32_t)P.ntiles_owned * (uint32_t)(P.tile_w * P.tile_h);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        uint32_t pixel;
        if (!tile_pixel(P, B.tile_ids, q, pixel)) continue;
        const float4 a = B.accum[pixel];
        float4 s = P.sample == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : sum[pixel];
        s.x += a.x; s.y += a.y; s.z += a.z;
        sum[pixel] = s;
    }
}
void launch_spp_accumulate(const FrameParams& fp, const FrameBuffers& fb, float4* sum, hipStream_t s) {
    hipLaunchKernelGGL(k_spp_accumulate, dim3(2048), dim3(256), 0, s, fp, fb, sum);
}

/* tile-major packing of this rank's pixels for the framebuffer gather */
__global__ __launch_bounds__(256) void k_tiles_copy(FrameParams P, const uint32_t* tile_ids, int ntiles, const uint8_t* src,
                                                     uint8_t* dst, int to_packed) {
    const uint32_t total = (uint32_t)ntiles * (uint32_t)(P.tile_w * P.tile_h);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        uint32_t pixel;
        if (!tile_pixel(P, tile_ids, q, pixel)) continue;
        for (int c = 0; c < 3; c++) {
            if (to_packed) dst[(size_t)q * 3 + c] = src[(size_t)pixel * 3 + c];
            else dst[(size_t)pixel * 3 + c] = src[(size_t)q * 3 + c];
        }
    }
}
void launch_tiles_copy(const FrameParams& fp, const uint32_t* tile_ids, int ntiles, const uint8_t* src, uint8_t* dst,
                       int to_packed, hipStream_t s) {
    hipLaunchKernelGGL(k_tiles_copy, dim3(2048), dim3(256), 0, s, fp, tile_ids, ntiles, src, dst, to_packed);
}

/* ------------------------------------------------------------------------------------------
 * scene packing: crt_prim (scene order) -> Geom/Shade records at their slots
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_pack_prims(const crt_prim* prims, const uint32_t* matid, const uint32_t* slot_of,
                                                     uint32_t n, Geom* geom, Shade* shade) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const crt_prim& s = prims[i];
    const uint32_t slot = slot_of[i];
    float4* g = reinterpret_cast<float4*>(geom + slot);
    float4* h = reinterpret_cast<float4*>(shade + slot);
    if (s.type == 0) {
        const vec3 c(s.position);
        g[0] = make_float4(c.x, c.y, c.z, s.radius);
        g[1] = make_float4(dot(c, c), s.radius * s.radius, 0.f, 0.f);     /* Ray.cu:110 terms that do not depend on the ray */
        g[2] = make_float4(0.f, __uint_as_float(kSphereTag), 0.f, 0.f);
        h[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        h[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const vec3 v1(s.v1), v2(s.v2), v3(s.v3);
        vec3 ng = normalize(cross((v2 - v1), (v3 - v1)));                 /* Ray.cu:145, same operation order */
        if (ng.x != ng.x) ng.x = __uint_as_float(0x7fc00000u);           /* degenerate: never the sphere tag */
        g[0] = make_float4(v1.x, v1.y, v1.z, v2.x);
        g[1] = make_float4(v2.y, v2.z, v3.x, v3.y);
        g[2] = make_float4(v3.z, ng.x, ng.y, ng.z);
        h[0] = make_float4(s.v1Norm[0], s.v1Norm[1], s.v1Norm[2], s.v2Norm[0]);
        h[1] = make_float4(s.v2Norm[1], s.v2Norm[2], s.v3Norm[0], s.v3Norm[1]);
    }
    h[2] = make_float4(s.type == 0 ? 0.f : s.v3Norm[2], __uint_as_float(matid[i]), 0.f, 0.f);
}
void launch_pack_prims(const crt_prim* d_prims, const uint32_t* d_matid, const uint32_t* slot_of, uint32_t n, Geom* geom,
                       Shade* shade, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pack_prims, dim3((n + 255) / 256), dim3(256), 0, s, d_prims, d_matid, slot_of, n, geom, shade);
}

/* ------------------------------------------------------------------------------------------
 * animation step (main.cu:545-573) on the device copy of the scene: triangles (positions and
 * normals) times the per-frame rotation -- mat4 * vec4(v, 1) evaluated as (m0*x + m1*y) +
 * (m2*z + m3*1), glm/detail/type_mat4x4.inl:640-651 -- spheres bounce off the room walls and move.
 * Same expressions, same order, no contraction: bit-identical to crt_animate_host.
 * ------------------------------------------------------------------------------------------ */
CRT_DEV void anim_apply(const AnimParam& A, float* v) {
    float o[3];
    for (int k = 0; k < 3; k++) {
        const float a0 = A.m[0 * 4 + k] * v[0] + A.m[1 * 4 + k] * v[1];
        const float a1 = A.m[2 * 4 + k] * v[2] + A.m[3 * 4 + k] * 1.0f;
        o[k] = a0 + a1;
    }
    v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
}
__global__ __launch_bounds__(256) void k_animate(crt_prim* prims, uint32_t n, AnimParam A) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    crt_prim& s = prims[i];
    if (s.type == 1) {
        anim_apply(A, s.v1); anim_apply(A, s.v2); anim_apply(A, s.v3);
        anim_apply(A, s.v1Norm); anim_apply(A, s.v2Norm); anim_apply(A, s.v3Norm);
    } else if (s.type == 0) {
        vec3 d(s.direction);
        if (s.position[0] + s.radius > 200.0f) d = reflect(d, vec3(-1, 0, 0));
        if (s.position[0] - s.radius < -200.0f) d = reflect(d, vec3(1, 0, 0));
        if (s.position[1] + s.radius > 200.0f) d = reflect(d, vec3(0, -1, 0));
        if (s.position[1] - s.radius < -200.0f) d = reflect(d, vec3(0, 1, 0));
        if (s.position[2] + s.radius > 300.0f) d = reflect(d, vec3(0, 0, -1));
        if (s.position[2] - s.radius < -300.0f) d = reflect(d, vec3(0, 0, 1));
        d.store(s.direction);
        (vec3(s.position) + d * s.speed * 8.0f).store(s.position);
    }
}
void launch_animate(crt_prim* d_prims, uint32_t n, const AnimParam& a, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_animate, dim3((n + 255u) / 256u), dim3(256), 0, s, d_prims, n, a);
}

/* ------------------------------------------------------------------------------------------
 * refit (refitTree, bvh.cu:288-400): pass 1 recomputes every leaf child's box from its
 * primitives; pass 2 climbs from the nodes that have no internal children, the last child to
 * arrive at a parent (agent-scope arrival counter) carries on.  grow_only keeps the reference's
 * "boxes never shrink" rule; min/max are exact, so the result is order independent.
 * ------------------------------------------------------------------------------------------ */
template <class N>
__global__ __launch_bounds__(256) void k_refit_leaves(N* nodes, uint32_t nnodes, const Geom* geom, float origin_bound, int grow_only) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nnodes * (uint32_t)N::W) return;
    N& nd = nodes[id / (uint32_t)N::W];
    const int k = (int)(id % (uint32_t)N::W);
    const uint32_t link = nd.child[k];
    if (link == kChildEmpty || !(link & kLeafFlag)) return;
    const uint32_t start = link & kLeafStartMask, cnt = ((link >> kLeafCountShift) & 63u) + 1u;
    float lo[3], hi[3];
    if (grow_only) { lo[0] = nd.lox[k]; lo[1] = nd.loy[k]; lo[2] = nd.loz[k]; hi[0] = nd.hix[k]; hi[1] = nd.hiy[k]; hi[2] = nd.hiz[k]; }
    else { lo[0] = lo[1] = lo[2] = 3.402823466e+38f; hi[0] = hi[1] = hi[2] = -3.402823466e+38f; }
    for (uint32_t p = start; p < start + cnt; p++) {
        const GeomRec g = load_geom(geom, p);
        if (is_sphere(g)) {
            const float c[3] = {g.a.x, g.a.y, g.a.z};
            /* the reference topology keeps the reference's boxes (bvh.cu:301-360); the wide tree's hold what the hit predicate can reach */
            const float hs = N::W == 8 ? sphere_box_half(c[0], c[1], c[2], g.a.w, origin_bound) : g.a.w;
            for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], c[a] - hs); hi[a] = fmaxf(hi[a], c[a] + hs); }
        } else {
            const float vx[3] = {g.a.x, g.a.w, g.b.z}, vy[3] = {g.a.y, g.b.x, g.b.w}, vz[3] = {g.a.z, g.b.y, g.c.x};
            for (int q = 0; q < 3; q++) {
                lo[0] = fminf(lo[0], vx[q]); hi[0] = fmaxf(hi[0], vx[q]);
                lo[1] = fminf(lo[1], vy[q]); hi[1] = fmaxf(hi[1], vy[q]);
                lo[2] = fminf(lo[2], vz[q]); hi[2] = fmaxf(hi[2], vz[q]);
            }
        }
    }
    nd.lox[k] = lo[0]; nd.loy[k] = lo[1]; nd.loz[k] = lo[2];
    nd.hix[k] = hi[0]; nd.hiy[k] = hi[1]; nd.hiz[k] = hi[2];
}

template <class N>
__global__ __launch_bounds__(256) void k_refit_up(N* nodes, uint32_t nnodes, uint32_t* arrive, float* root_box, int grow_only) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nnodes) return;
    if (nodes[id].n_internal != 0) return;
    uint32_t cur = id;
    for (int guard = 0; guard < 4096; guard++) {
        const N& nd = nodes[cur];
        float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
        float hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        for (int k = 0; k < N::W; k++) {
            lo[0] = fminf(lo[0], nd.lox[k]); lo[1] = fminf(lo[1], nd.loy[k]); lo[2] = fminf(lo[2], nd.loz[k]);
            hi[0] = fmaxf(hi[0], nd.hix[k]); hi[1] = fmaxf(hi[1], nd.hiy[k]); hi[2] = fmaxf(hi[2], nd.hiz[k]);
        }
        const uint32_t parent = nd.parent;
        if (parent == kChildEmpty) {
            if (grow_only) {
                lo[0] = fminf(lo[0], root_box[0]); hi[0] = fmaxf(hi[0], root_box[1]);
                lo[1] = fminf(lo[1], root_box[2]); hi[1] = fmaxf(hi[1], root_box[3]);
                lo[2] = fminf(lo[2], root_box[4]); hi[2] = fmaxf(hi[2], root_box[5]);
            }
            root_box[0] = lo[0]; root_box[1] = hi[0]; root_box[2] = lo[1]; root_box[3] = hi[1]; root_box[4] = lo[2]; root_box[5] = hi[2];
            return;
        }
        N& pn = nodes[parent];
        const int k = (int)nd.parent_slot;
        if (grow_only) {
            lo[0] = fminf(lo[0], pn.lox[k]); lo[1] = fminf(lo[1], pn.loy[k]); lo[2] = fminf(lo[2], pn.loz[k]);
            hi[0] = fmaxf(hi[0], pn.hix[k]); hi[1] = fmaxf(hi[1], pn.hiy[k]); hi[2] = fmaxf(hi[2], pn.hiz[k]);
        }
        pn.lox[k] = lo[0]; pn.loy[k] = lo[1]; pn.loz[k] = lo[2];
        pn.hix[k] = hi[0]; pn.hiy[k] = hi[1]; pn.hiz[k] = hi[2];
        const uint32_t need = pn.n_internal;
        /* publish, then arrive: stores drained, agent-scope release, relaxed agent-scope ticket */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t prev = __hip_atomic_fetch_add(&arrive[parent], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u < need) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur = parent;
    }
}

template <class N>
static void launch_refit_t(N* nodes, uint32_t nnodes, const Geom* geom, float origin_bound, uint32_t* arrive, float* root_box, int grow_only, hipStream_t s) {
    if (nnodes == 0) return;
    (void)hipMemsetAsync(arrive, 0, sizeof(uint32_t) * nnodes, s);
    hipLaunchKernelGGL((k_refit_leaves<N>), dim3((nnodes * (uint32_t)N::W + 255u) / 256u), dim3(256), 0, s, nodes, nnodes, geom, origin_bound, grow_only);
    hipLaunchKernelGGL((k_refit_up<N>), dim3((nnodes + 255u) / 256u), dim3(256), 0, s, nodes, nnodes, arrive, root_box, grow_only);
}
void launch_refit(Node4* nodes, uint32_t nnodes, const Geom* geom, uint32_t* arrive, float* root_box, int grow_only, hipStream_t s) {
    launch_refit_t(nodes, nnodes, geom, 0.f, arrive, root_box, grow_only, s);
}
void launch_refit8(Node8* nodes, uint32_t nnodes, const Geom* geom, float origin_bound, uint32_t* arrive, float* root_box, int grow_only, hipStream_t s) {
    launch_refit_t(nodes, nnodes, geom, origin_bound, arrive, root_box, grow_only, s);
}

/* ------------------------------------------------------------------------------------------
 * small helpers for the parity hooks
 * ------------------------------------------------------------------------------------------ */
__global__ void k_slots_to_scene(const int32_t* slots, const uint32_t* sidx, int32_t* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = slots[i] >= 0 ? (int32_t)sidx[slots[i]] : -1;
}
void launch_slots_to_scene(const int32_t* slots, const uint32_t* sidx, int32_t* out, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_slots_to_scene, dim3((n + 255) / 256), dim3(256), 0, s, slots, sidx, out, n);
}
__global__ void k_powf_probe(const float* x, const float* y, uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = powf_libm(x[i], y[i]);
}
void launch_powf_probe(const float* x, const float* y, uint32_t n, float* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_powf_probe, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n, out);
}

__global__ void k_pack_rays(const float* od, uint32_t n, float4* rayO, float4* rayD, uint32_t dw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rayO[i] = make_float4(od[6 * i], od[6 * i + 1], od[6 * i + 2], __uint_as_float(i));
    rayD[i] = make_float4(od[6 * i + 3], od[6 * i + 4], od[6 * i + 5], __uint_as_float(dw));
}
void launch_pack_rays(const float* org_dir, uint32_t n, float4* rayO, float4* rayD, bool shadow, hipStream_t s) {
    /* shadow rays of the parity hook follow the directional-light predicate (Ray.cu:650-656) */
    if (n) hipLaunchKernelGGL(k_pack_rays, dim3((n + 255) / 256), dim3(256), 0, s, org_dir, n, rayO, rayD, shadow ? kShadowDirectional : 0u);
}

}  // namespace crt
