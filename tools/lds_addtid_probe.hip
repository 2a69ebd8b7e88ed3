// lds_addtid_probe.hip -- does ds_read_addtid_b32 read LDS[M0 + offset + 4 * lane] on gfx950?  (the lane-private column
// read of decode_kernel: the row comes from the scalar unit, no address VGPR and no v_add per read)
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_addtid_probe.hip -o tools/lds_addtid_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(uint32_t* out, int row_a, int row_b)
{
    __shared__ uint32_t pad[100];          // so that the array does not start at LDS address 0
    __shared__ uint32_t t[4][24 * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    pad[threadIdx.x % 100] = 7;
    for (int r = 0; r < 24; r++) t[wv][r * 64 + lane] = 1000000u * wv + 1000u * r + lane;
    __syncthreads();
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&t[wv][0]);
    uint32_t a, b;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_read_addtid_b32 %0\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\tds_read_addtid_b32 %1 offset:256\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "s"(base + row_a * 256), "s"(base + row_b * 256) : "m0", "memory");
    out[threadIdx.x * 2] = a;
    out[threadIdx.x * 2 + 1] = b;
    if (pad[3] == 0) out[0] = 0;
}

int main()
{
    uint32_t* d;
    hipMalloc(&d, 256 * 2 * 4);
    const int ra = 5, rb = 17;
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, ra, rb);
    std::vector<uint32_t> h(512);
    hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; t++) {
        const uint32_t wa = 1000000u * (t >> 6) + 1000u * ra + (t & 63), wb = 1000000u * (t >> 6) + 1000u * (rb + 1) + (t & 63);
        if (h[2 * t] != wa || h[2 * t + 1] != wb) { if (bad < 5) printf("thread %d: got %u %u want %u %u\n", t, h[2 * t], h[2 * t + 1], wa, wb); bad++; }
    }
    printf("ds_read_addtid_b32: LDS[M0 + offset + 4 * lane] %s (%d mismatches)\n", bad ? "NOT confirmed" : "confirmed", bad);
    return bad != 0;
}
