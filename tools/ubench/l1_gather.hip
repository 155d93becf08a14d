/* tools/ubench/l1_gather.hip -- the vector-memory ceiling for k_trace8's access pattern: every lane of a wave reads
 * ITS OWN record (a BVH node) with 16-byte loads, records picked at random from a table that lives in L1 / L2 /
 * Infinity Cache.  What is the rate of such lane requests per clock per CU, and does it depend on how many of
 * them one record takes (80-byte QNode8 = 5, a 64-byte node = 4), on alignment, on lanes sharing a line?
 *
 * Patterns (each lane: R dependent-free 16-byte loads per step, xor-reduced into a register):
 *   lane80    record = 80 bytes at 80-byte stride (QNode8 as it is), lane-random record, 5 loads
 *   lane64    record = 64 bytes, 64-byte aligned, lane-random record, 4 loads
 *   lane128   record = 128 bytes, 128-byte aligned, lane-random record, 5 loads (first 80 bytes)
 *   lane16    ONE 16-byte load per step from a lane-random 16-byte slot
 *   quad64    the 4 lanes of a quad read the 4 consecutive 16-byte pieces of ONE random 64-byte line (1 load per step)
 *   wave1k    fully coalesced: the wave reads 1 KiB contiguous at a random 1-KiB-aligned place (1 load per step)
 * Table sizes: 16 KiB (L1), 2 MiB (L2), 24 MiB (the qnodes of the 1M-triangle soup: L2 + Infinity Cache), 96 MiB.
 * Geometry as valu_issue.hip: k workgroups of 256 threads per CU via dynamic LDS; k = 5 (k_trace8's occupancy) and 8.
 *
 * Reported: G lane-requests/s and GB/s chip-wide (HIP events), the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz,
 * median over waves) and with it lane requests per shader clock per CU.
 *
 * build:  hipcc --offload-arch=gfx950 -O2 -o l1_gather l1_gather.hip
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

/* dynamic LDS per workgroup such that exactly k workgroups fit a CU: LDS is handed out in 1280-byte granules (128 per CU);
 * k x floor(120 / k) granules fit with room to spare, k + 1 such workgroups do not (a first version asked for 160 KiB / k - 64
 * bytes, which rounds UP to a granule: only 4 of "5" workgroups were resident and the fifth ran alone afterwards) */
static size_t lds_bytes_for(int k) { return (size_t)(120 / k) * 1280; }

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kSteps = 2048;
enum Pat { P_LANE80 = 0, P_LANE64, P_LANE128, P_LANE16, P_QUAD64, P_WAVE1K, P_N };
static const char* kPat[P_N] = {"lane80 (5 x 16 B of a lane-random 80-byte record)", "lane64 (4 x 16 B of a lane-random aligned 64-byte record)",
                                "lane128 (5 x 16 B of a lane-random aligned 128-byte record)", "lane16 (1 x 16 B, lane-random)",
                                "quad64 (a quad reads one random 64-byte line)", "wave1k (a wave reads 1 KiB contiguous)"};
static const int kLoads[P_N] = {5, 4, 5, 1, 1, 1};

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int PAT>
__global__ __launch_bounds__(256) void k_gather(const char* table, uint32_t nmask, uint4* sink, unsigned long long* cycles) {
    extern __shared__ char lds_pad[];
    if (threadIdx.x == 1023) lds_pad[0] = 1;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = gid * 2654435761u + 12345u;
    uint32_t sq = (gid >> 2) * 2246822519u + 777u, sw = (gid >> 6) * 3266489917u + 99u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 2
    for (int it = 0; it < kSteps; it++) {
        if constexpr (PAT == P_LANE80) {
            const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)(lcg(s) & nmask) * 80u);
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
            acc.x ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x; acc.y ^= a.y ^ b.z ^ c.w ^ d.x ^ e.y; acc.z ^= a.z ^ b.w ^ c.x ^ d.y ^ e.z; acc.w ^= a.w ^ b.x ^ c.y ^ d.z ^ e.w;
        } else if constexpr (PAT == P_LANE64) {
            const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)(lcg(s) & nmask) * 64u);
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y ^= a.y ^ b.z ^ c.w ^ d.x; acc.z ^= a.z ^ b.w ^ c.x ^ d.y; acc.w ^= a.w ^ b.x ^ c.y ^ d.z;
        } else if constexpr (PAT == P_LANE128) {
            const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)(lcg(s) & nmask) * 128u);
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
            acc.x ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x; acc.y ^= a.y ^ b.z ^ c.w ^ d.x ^ e.y; acc.z ^= a.z ^ b.w ^ c.x ^ d.y ^ e.z; acc.w ^= a.w ^ b.x ^ c.y ^ d.z ^ e.w;
        } else if constexpr (PAT == P_LANE16) {
            const uint4 a = *reinterpret_cast<const uint4*>(table + (size_t)(lcg(s) & nmask) * 16u);
            acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w;
        } else if constexpr (PAT == P_QUAD64) {
            const uint4 a = *reinterpret_cast<const uint4*>(table + (size_t)(lcg(sq) & nmask) * 64u + (threadIdx.x & 3u) * 16u);
            acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w;
        } else {
            const uint4 a = *reinterpret_cast<const uint4*>(table + (size_t)(lcg(sw) & nmask) * 1024u + (threadIdx.x & 63u) * 16u);
            acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { cycles[2 * (gid >> 6)] = t1 - t0; cycles[2 * (gid >> 6) + 1] = r1 - r0; }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

template <int PAT>
static void run_pat(int cus, const char* d_table, size_t table_bytes, const char* where, uint4* sink, unsigned long long* d_cycles,
                    std::vector<unsigned long long>& h) {
    const uint32_t rec = PAT == P_LANE80 ? 80 : PAT == P_LANE64 ? 64 : PAT == P_LANE128 ? 128 : PAT == P_LANE16 ? 16 : PAT == P_QUAD64 ? 64 : 1024;
    uint32_t nrec = 1;
    while ((size_t)nrec * 2 * rec <= table_bytes) nrec *= 2;      /* power of two: the index is a mask, not a division */
    const int ks[] = {5, 8};
    for (int k : ks) {
        const size_t lds = lds_bytes_for(k);
        CHECK(hipFuncSetAttribute((const void*)k_gather<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int blocks = cus * k;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float best_ms = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather<PAT>, dim3(blocks), dim3(256), lds, 0, d_table, nrec - 1u, sink, d_cycles);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best_ms) best_ms = ms;
        }
        const int nw = blocks * 4;
        CHECK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost));
        std::vector<double> clk(nw);
        for (int i = 0; i < nw; i++) clk[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.;      /* s_memrealtime ticks at 100 MHz */
        std::sort(clk.begin(), clk.end());
        const double clock_ghz = clk[nw / 2];
        const double lane_req_per_wave = (double)kSteps * kLoads[PAT] * 64;
        const double total_req = lane_req_per_wave * nw;
        const double cu_clks = best_ms * 1e-3 * clock_ghz * 1e9;                /* shader clocks the launch lasted */
        printf("{\"bench\": \"l1_gather\", \"pattern\": \"%s\", \"table_bytes\": %zu, \"records\": %u, \"served_from\": \"%s\", \"waves_per_simd\": %d, "
               "\"lane_requests_per_clk_per_cu\": %.4f, \"clks_per_wave_load_instruction_per_cu\": %.2f, "
               "\"kernel_ms\": %.4f, \"g_lane_requests_per_s\": %.2f, \"gb_per_s\": %.1f, \"clock_ghz\": %.3f}\n",
               kPat[PAT], table_bytes, nrec, where, k, total_req / cus / cu_clks, cu_clks / (total_req / 64 / cus),
               best_ms, total_req / (best_ms * 1e-3) / 1e9, total_req * 16 / (best_ms * 1e-3) / 1e9, clock_ghz);
        fflush(stdout);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t sizes[] = {16u << 10, 2u << 20, 24u << 20, 96u << 20};
    const char* where[] = {"L1", "L2", "L2 + Infinity Cache (size of the 1M-triangle soup's qnodes)", "Infinity Cache"};
    char* d_table; uint4* sink; unsigned long long* d_cycles;
    CHECK(hipMalloc(&d_table, (96u << 20) + 4096));
    std::vector<uint32_t> init((96u << 20) / 4);
    uint32_t s = 1;
    for (auto& v : init) { s = s * 1664525u + 1013904223u; v = s; }
    CHECK(hipMemcpy(d_table, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * 2 * cus * 8 * 4));
    std::vector<unsigned long long> h((size_t)2 * cus * 8 * 4);
    for (int i = 0; i < 4; i++) {
        run_pat<P_LANE80>(cus, d_table, sizes[i], where[i], sink, d_cycles, h);
        run_pat<P_LANE64>(cus, d_table, sizes[i], where[i], sink, d_cycles, h);
        run_pat<P_LANE128>(cus, d_table, sizes[i], where[i], sink, d_cycles, h);
        run_pat<P_LANE16>(cus, d_table, sizes[i], where[i], sink, d_cycles, h);
        run_pat<P_QUAD64>(cus, d_table, sizes[i], where[i], sink, d_cycles, h);
        run_pat<P_WAVE1K>(cus, d_table, sizes[i], where[i], sink, d_cycles, h);
    }
    return 0;
}
