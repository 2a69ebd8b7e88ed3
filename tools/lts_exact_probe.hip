// lts_exact_probe.hip -- the exact stage of the LTS search (spec rule 6 stage 2) as one MFMA tile against the plain fmaf
// chain, on random data: every fetched value must be bit-identical.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DWR_DEBUG_EXACT -Iinclude -Ignuradio-wifi-imagetransfer_amd/csrc tools/lts_exact_probe.hip -o tools/lts_exact_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wifirx.h"
#include "wr_device.h"
#include "wr_kernels.h"
#include "wr_quad.h"

__global__ void probe(const float* y, const int* cands, float* out)
{
    __shared__ float lds[2 * WR_PRE_FRAME_FLOATS];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2 * WR_PRE_FRAME_FLOATS; i += 64) lds[i] = y[i];
    __syncthreads();
    int c0[8], c1[8];
    for (int k = 0; k < 8; k++) { c0[k] = cands[k]; c1[k] = cands[8 + k]; }
    int lag, lag2;
    const float a = wr::lts_exact_pair(lds, lane, c0, c1, lag);
    const float b = wr::lts_exact_pair_valu(lds, lane, c0, c1, lag2);
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = (float)lag; out[192 + lane] = (float)lag2;
}

int main()
{
    std::vector<float> y(2 * WR_PRE_FRAME_FLOATS);
    srand(3);
    for (auto& v : y) v = (float)(rand() % 2001 - 1000) / 977.0f;
    int cands[16] = { 9, 73, 0, 0, 0, 0, 0, 0,   201, 137, 5, 318, 64, 11, 250, 99 };
    float *dy, *dout; int* dc;
    hipMalloc(&dy, y.size() * 4); hipMalloc(&dout, 256 * 4); hipMalloc(&dc, 64);
    hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, cands, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dy, dc, dout);
    std::vector<float> o(256);
    hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 32; l++) {
        const bool ne = __builtin_memcmp(&o[l], &o[64 + l], 4) != 0;
        if (ne) bad++;
        printf("lane %2d lag %3.0f/%3.0f mfma % .9e valu % .9e %s\n", l, o[128 + l], o[192 + l], o[l], o[64 + l], ne ? "DIFF" : "");
    }
    printf("%d of 32 differ\n", bad);
    return bad != 0;
}
