// Do f32 matrix instructions (v_mfma_f32_16x16x4_f32) run BESIDE the vector ALU work of the other waves of a SIMD,
// or do they take the vector ALU's time?  Workgroups of 4 x W waves (W per SIMD); in a "mixed" launch the waves whose
// index within their SIMD is below M run an MFMA loop and the others a v_fma_f32 loop, both with the same iteration
// count as in the "alone" launches.  If the two pipes overlap, mixed time ~ max(alone); if not, ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(float* out, int it_mfma, int it_valu, int n_mfma_waves_per_simd)
{
    const int wave = threadIdx.x >> 6;              // waves go to SIMDs round-robin: wave >> 2 = index within its SIMD
    const bool mf = (wave >> 2) < n_mfma_waves_per_simd;
    float r = 0.0f;
    if (mf) {
        f4 a0 = { 0, 0, 0, 0 }, a1 = a0, a2 = a0, a3 = a0, a4 = a0;
        float x = threadIdx.x * 1e-3f, y = 1.0f - x;
        for (int i = 0; i < it_mfma; i++) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3] + a4[0];
    } else {
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        const float b = 1.0001f, c = 0.5f;
        for (int i = 0; i < it_valu; i++) {
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    if (r == 123.456f) out[0] = r;
}
static float run(int W, int M, int it_mfma, int it_valu)
{
    static float* o = nullptr;
    if (!o) (void)hipMalloc(&o, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256 * W), 0, 0, o, it_mfma, it_valu, M);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main()
{
    const int W = 4;
    // 1 MFMA wave + 3 VALU waves per SIMD; iteration counts chosen so that each side alone takes about the same time:
    // MFMA wave: 5 per iteration x 32 cycles = 160 cycles; VALU wave: 8 per iteration, 3 waves share the SIMD
    const int it_m = 20000, it_v = 20000 * 160 / (8 * 3 * 4) * 1;      // 3 waves x 8 fma x 4 cycles = 96 cycles per iteration
    float t_m = run(W, 1, it_m, 0);          // the VALU waves do nothing
    float t_v = run(W, 1, 0, it_v);          // the MFMA wave does nothing
    float t_b = run(W, 1, it_m, it_v);
    printf("1 MFMA wave + 3 VALU waves per SIMD: MFMA alone %.3f ms, VALU alone %.3f ms, together %.3f ms  (overlap if ~max, serial if ~sum %.3f)\n",
           t_m, t_v, t_b, t_m + t_v);
    // all four waves VALU vs all four MFMA for reference rates
    float t4v = run(W, 0, 0, it_v);
    float t4m = run(W, 4, it_m, 0);
    printf("4 VALU waves: %.3f ms (%.2f cycles per v_fma_f32 per SIMD at 2.4 GHz); 4 MFMA waves: %.3f ms (%.1f cycles per MFMA per SIMD)\n",
           t4v, t4v * 1e-3 * 2.4e9 / ((double)it_v * 8 * 4), t4m, t4m * 1e-3 * 2.4e9 / ((double)it_m * 5 * 4));
    // 2 + 2
    float t2m = run(W, 2, it_m, 0), t2v = run(W, 2, 0, it_v), t2b = run(W, 2, it_m, it_v);
    printf("2 MFMA + 2 VALU waves per SIMD: MFMA alone %.3f, VALU alone %.3f, together %.3f (sum %.3f)\n", t2m, t2v, t2b, t2m + t2v);
    return 0;
}
