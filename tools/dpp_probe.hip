#include <hip/hip_runtime.h>
#include <cstdio>
#include "../gnuradio-wifi-imagetransfer_amd/csrc/wr_device.h"
__global__ void k(float* out) {
    int l = threadIdx.x;
    float v = (float)l;
    out[l] = wr::dpp_zero<0x111>(v);
    out[64 + l] = wr::dpp_zero<0x101>(v);
    out[128 + l] = wr::row_prefix16(1.0f);
    out[192 + l] = wr::row_suffix16(1.0f);
    out[256 + l] = wr::dpp_zero<0x118>(v);
}
int main() {
    float* d; hipMalloc(&d, 320 * 4);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
    float h[320]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"row_shr1", "row_shl1", "prefix", "suffix", "row_shr8"};
    for (int r = 0; r < 5; r++) { printf("%s:", names[r]); for (int i = 0; i < 34; i++) printf(" %g", h[r * 64 + i]); printf("\n"); }
    return 0;
}
