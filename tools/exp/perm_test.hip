#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t w, uint32_t* out, float* f) {
    uint32_t a = __builtin_amdgcn_perm(0x64646464u, w, 0x04010400u);
    uint32_t b = __builtin_amdgcn_perm(0x64646464u, w, 0x04030402u);
    out[0] = a; out[1] = b;
    float r0, r1;
    float s = 2.0f, c = 1.0f;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(s), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(s), "v"(c));
    f[0] = r0; f[1] = r1;
}
int main() {
    uint32_t* d; float* f; hipMalloc(&d, 8); hipMalloc(&f, 8);
    k<<<1, 1>>>(0x44332211u, d, f);
    uint32_t h[2]; float g[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(g, f, 8, hipMemcpyDeviceToHost);
    printf("%08x %08x  %f %f (expect (1024+0x11)*2+1=%f, (1024+0x22)*2+1=%f)\n", h[0], h[1], g[0], g[1], (1024 + 0x11) * 2.0 + 1, (1024 + 0x22) * 2.0 + 1);
}
