// mfma_i8_probe.hip -- operand layout of v_mfma_i32_16x16x64_i8 on gfx950, checked with exact integer data:
// lane l holds A[row l & 15][k = 16 (l >> 4) + j] and B[k = 16 (l >> 4) + j][col l & 15] in byte j = 0..15 of its four
// operand registers (little endian); C/D: col = l & 15, row = 4 (l >> 4) + reg.  Also times a dependent chain.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_i8_probe.hip -o tools/mfma_i8_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void probe(const int8_t* A, const int8_t* B, int* C, int reps)
{
    const int l = threadIdx.x & 63;
    v4i a, b, c = { 0, 0, 0, 0 };
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; j++) {
        ab[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j];
        bb[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)];
    }
    __builtin_memcpy(&a, ab, 16);
    __builtin_memcpy(&b, bb, 16);
    for (int r = 0; r < reps; r++) c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int g = 0; g < 4; g++) C[(4 * (l >> 4) + g) * 16 + (l & 15)] = c[g];
}

int main()
{
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    srand(7);
    for (auto& v : A) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : B) v = (int8_t)(rand() % 255 - 127);
    int8_t *dA, *dB; int* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, 1);
    std::vector<int> C(256);
    hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            int s = 0;
            for (int k = 0; k < 64; k++) s += (int)A[i * 64 + k] * (int)B[k * 16 + j];
            if (s != C[i * 16 + j]) { if (bad < 5) printf("C[%d][%d] = %d, want %d\n", i, j, C[i * 16 + j], s); bad++; }
        }
    printf("v_mfma_i32_16x16x64_i8 layout %s (%d of 256 wrong)\n", bad ? "NOT as assumed" : "confirmed", bad);
    // rate: 4096 waves x 1000 dependent instructions
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, dA, dB, dC, 1000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, dA, dB, dC, 1000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("4096 waves x 1000 dependent i8 16x16x64: %.3f ms -> %.1f cycles per instruction per SIMD slot at 2.1 GHz (4 waves per SIMD)\n", ms, ms * 1e-3 * 2.1e9 / 1000 / 4);
    return bad != 0;
}
