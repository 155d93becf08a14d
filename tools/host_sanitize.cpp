#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "bvh_build.h"
using namespace crt;
int main() {
    const int n = 200000;
    std::vector<crt_prim> prims(n);
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-180.f, 180.f), O(-1.5f, 1.5f);
    for (auto& p : prims) {
        memset(&p, 0, sizeof p);
        p.type = 1;
        float c[3] = {U(rng), U(rng), U(rng)};
        for (int a = 0; a < 3; a++) { p.v1[a] = c[a] + O(rng); p.v2[a] = c[a] + O(rng); p.v3[a] = c[a] + O(rng); }
    }
    RefTree ref; HostBVH bvh;
    int rc = build_reference_tree(prims.data(), n, ref);
    printf("ref rc=%d nodes=%zu leaves=%zu depth=%d\n", rc, ref.nodes.size(), ref.leafs.size(), ref.depth);
    rc = build_wide(prims.data(), n, &ref, bvh);
    printf("wide rc=%d nodes=%zu depth=%d leaves=%d\n", rc, bvh.nodes8.size(), bvh.depth, bvh.leaves);
    return 0;
}
