#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    int l = threadIdx.x;
    int v = l;
    out[l] = (float)__builtin_amdgcn_update_dpp(0, v, 0x15B, 0xf, 0xf, false);        // row_newbcast:11
    out[64 + l] = (float)__builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);   // row_ror:4
    out[128 + l] = (float)__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    out[192 + l] = (float)__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
    float h[256]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"row_newbcast11", "row_ror4", "quad_xor1", "quad_xor2"};
    for (int r = 0; r < 4; r++) { printf("%s:", names[r]); for (int i = 0; i < 36; i++) printf(" %g", h[r * 64 + i]); printf("\n"); }
    return 0;
}
