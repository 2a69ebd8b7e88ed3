// Numerics + layout probe for the f32 matrix instructions the LTS correlation uses:
//   v_mfma_f32_16x16x4_f32:  D[16x16] = A[16x4] * B[4x16] + C      (lane l: A[l&15][l>>4], B[l>>4][l&15])
//   v_mfma_f32_32x32x2_f32:  D[32x32] = A[32x2] * B[2x32] + C      (lane l: A[l&31][l>>5], B[l>>5][l&31])
// Questions: (1) is the documented operand / result layout right, (2) is every element bit for bit the fmaf chain
// d = fma(a[k], b[k], d) over k ascending, starting from C -- on random values of mixed magnitude (cancellation),
// with denormals, and chained over several instructions?  Also times a chain of each form on one wave.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_f32_probe.hip -o tools/mfma_f32_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

// KSTEPS chained instructions: K = 4 * KSTEPS (16x16) or 2 * KSTEPS (32x32)
template <int KSTEPS>
__global__ void k16(const float* a, const float* b, const float* c, float* d)
{
    const int l = threadIdx.x;
    f4 acc;
    for (int i = 0; i < 4; i++) acc[i] = c[(4 * (l >> 4) + i) * 16 + (l & 15)];        // C/D: row 4*(l>>4)+i, col l&15
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
        const int k = 4 * s + (l >> 4);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(l & 15) * (4 * KSTEPS) + k], b[k * 16 + (l & 15)], acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) d[(4 * (l >> 4) + i) * 16 + (l & 15)] = acc[i];
}

template <int KSTEPS>
__global__ void k32(const float* a, const float* b, const float* c, float* d)
{
    const int l = threadIdx.x;
    f16 acc;
    for (int i = 0; i < 16; i++) acc[i] = c[((i & 3) + 8 * (i >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
        const int k = 2 * s + (l >> 5);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(l & 31) * (2 * KSTEPS) + k], b[k * 32 + (l & 31)], acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; i++) d[((i & 3) + 8 * (i >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[i];
}

// issue-rate probe: N dependent / independent instructions on one wave per SIMD
__global__ void rate16(float* out, int iters)
{
    f4 a0 = { 0, 0, 0, 0 }, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f - x;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ void rate32(float* out, int iters)
{
    f16 a0, a1;
    for (int i = 0; i < 16; i++) { a0[i] = 0; a1[i] = 0; }
    float x = threadIdx.x * 1e-3f, y = 1.0f - x;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[5];
}

static float rnd(int spread)
{
    float m = (float)rand() / (float)RAND_MAX - 0.5f;
    return ldexpf(m, rand() % (2 * spread + 1) - spread);
}

template <int M, int KSTEPS, int KPER>
static int check(const char* name, void (*kern)(const float*, const float*, const float*, float*), int reps, int spread, bool denorm)
{
    const int K = KPER * KSTEPS;
    static float ha[64 * 64], hb[64 * 64], hc[32 * 32], hd[32 * 32];
    float *da, *db, *dc, *dd;
    (void)hipMalloc(&da, sizeof ha); (void)hipMalloc(&db, sizeof hb); (void)hipMalloc(&dc, sizeof hc); (void)hipMalloc(&dd, sizeof hd);
    long total = 0, same = 0, same_rev = 0, same_pair = 0;
    for (int rep = 0; rep < reps; rep++) {
        for (int i = 0; i < M * K; i++) { ha[i] = rnd(spread); hb[i] = rnd(spread); }
        for (int i = 0; i < M * M; i++) hc[i] = rnd(spread);
        if (denorm) for (int i = 0; i < M * K; i += 3) { ha[i] = ldexpf(ha[i], -120); hb[i] = ldexpf(hb[i], -20); }
        if (denorm) for (int i = 0; i < M * M; i += 2) hc[i] = ldexpf(hc[i], -135);
        (void)hipMemcpy(da, ha, sizeof(float) * M * K, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, hb, sizeof(float) * M * K, hipMemcpyHostToDevice);
        (void)hipMemcpy(dc, hc, sizeof(float) * M * M, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        (void)hipMemcpy(hd, dd, sizeof(float) * M * M, hipMemcpyDeviceToHost);
        for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) {
            float f = hc[i * M + j], r = hc[i * M + j], p = hc[i * M + j];
            for (int k = 0; k < K; k++) f = fmaf(ha[i * K + k], hb[k * M + j], f);                 // ascending chain
            for (int s = 0; s < KSTEPS; s++) for (int k = KPER - 1; k >= 0; k--)                  // descending inside an instruction
                r = fmaf(ha[i * K + KPER * s + k], hb[(KPER * s + k) * M + j], r);
            for (int s = 0; s < KSTEPS; s++) {                                                   // products summed first, then added
                float q = 0.0f;
                for (int k = 0; k < KPER; k++) q = fmaf(ha[i * K + KPER * s + k], hb[(KPER * s + k) * M + j], q);
                p = p + q;
            }
            float g = hd[i * M + j];
            total++;
            same += memcmp(&g, &f, 4) == 0;
            same_rev += memcmp(&g, &r, 4) == 0;
            same_pair += memcmp(&g, &p, 4) == 0;
        }
    }
    printf("%-26s K=%3d spread 2^+-%-2d %s: %ld elements, == ascending fmaf chain %ld, == descending-in-instruction %ld, == sum-then-add %ld\n",
           name, K, spread, denorm ? "denormals" : "normal   ", total, same, same_rev, same_pair);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dd);
    return same == total ? 0 : 1;
}

int main()
{
    srand(7);
    int bad = 0;
    bad += check<16, 1, 4>("mfma_f32_16x16x4_f32", k16<1>, 400, 12, false);
    bad += check<16, 1, 4>("mfma_f32_16x16x4_f32", k16<1>, 200, 2, false);
    bad += check<16, 1, 4>("mfma_f32_16x16x4_f32", k16<1>, 200, 6, true);
    bad += check<16, 16, 4>("mfma_f32_16x16x4_f32 x16", k16<16>, 200, 8, false);
    bad += check<16, 16, 4>("mfma_f32_16x16x4_f32 x16", k16<16>, 100, 4, true);
    bad += check<32, 1, 2>("mfma_f32_32x32x2_f32", k32<1>, 200, 12, false);
    bad += check<32, 1, 2>("mfma_f32_32x32x2_f32", k32<1>, 100, 6, true);
    bad += check<32, 32, 2>("mfma_f32_32x32x2_f32 x32", k32<32>, 100, 8, false);
    printf(bad ? "NOT an ascending fmaf chain somewhere (see above)\n" : "all cases: bit-identical to the k-ascending fmaf chain\n");

    // issue rate, 1 / 2 / 4 waves per SIMD on every CU
    float* out;
    (void)hipMalloc(&out, 1024 * 1024 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int iters = 20000, blocks = 256, threads = 256 * wps;
        for (int which = 0; which < 2; which++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0, 0);
                if (which == 0) hipLaunchKernelGGL(rate16, dim3(blocks), dim3(threads), 0, 0, out, iters);
                else hipLaunchKernelGGL(rate32, dim3(blocks), dim3(threads), 0, 0, out, iters);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double n_inst = (double)iters * (which == 0 ? 4 : 2) * wps;       // per SIMD
            const double flop = n_inst * (which == 0 ? 2048.0 : 4096.0) * 1024.0;   // 1024 SIMDs
            printf("%s, %d wave(s)/SIMD: %.3f ms, %.1f ns per instruction per SIMD, %.1f TFLOP/s\n",
                   which == 0 ? "16x16x4 " : "32x32x2 ", wps, ms, ms * 1e6 / n_inst, flop / (ms * 1e-3) / 1e12);
        }
    }
    return bad;
}
