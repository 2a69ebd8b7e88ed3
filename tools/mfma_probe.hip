// Numerics probe: is v_mfma_f32_4x4x1_16b_f32 (one multiply-add per element, K = 1) an IEEE fused multiply-add,
// and what does it do with denormals?  D[b][i][j] = A[b][i] * B[b][j] + C[b][i][j], 16 blocks of 4x4.
// Lane l supplies A[l/4][l%4] and B[l/4][l%4]; it holds D[l/4][0..3][l%4]?  (layout printed below)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
typedef float float4_ __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, const float* c, float* d)
{
    const int l = threadIdx.x;
    float4_ acc = { c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3] };
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) d[l * 4 + i] = acc[i];
}
int main()
{
    float ha[64], hb[64], hc[256], hd[256];
    float *da, *db, *dc, *dd;
    (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&dc, 1024); (void)hipMalloc(&dd, 1024);
    // pass 1: find the layout with distinct small integers
    for (int i = 0; i < 64; i++) { ha[i] = (float)(i + 1); hb[i] = (float)(100 + i); }
    for (int i = 0; i < 256; i++) hc[i] = 0.0f;
    (void)hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    (void)hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    // D value = A[x]*B[y]: recover (x, y) for lanes 0..7
    for (int l = 0; l < 8; l++) {
        printf("lane %d:", l);
        for (int i = 0; i < 4; i++) {
            int found = 0;
            for (int x = 0; x < 64 && !found; x++) for (int y = 0; y < 64 && !found; y++)
                if (ha[x] * hb[y] == hd[l * 4 + i]) { printf("  d[%d]=A[lane %d]*B[lane %d]", i, x, y); found = 1; }
        }
        printf("\n");
    }
    // pass 2: random values, compare with fmaf and with mul-then-add
    srand(1);
    int n_fma = 0, n_muladd = 0, n_other = 0, total = 0;
    for (int rep = 0; rep < 2000; rep++) {
        for (int i = 0; i < 64; i++) { ha[i] = ldexpf((float)rand() / (float)RAND_MAX - 0.5f, rand() % 40 - 20); hb[i] = ldexpf((float)rand() / (float)RAND_MAX - 0.5f, rand() % 40 - 20); }
        for (int i = 0; i < 256; i++) hc[i] = ldexpf((float)rand() / (float)RAND_MAX - 0.5f, rand() % 40 - 20);
        (void)hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        (void)hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
            // layout assumption checked in pass 1: d[l][i] = A[4*(l/4) + i] * B[l]  (printed above; adjust if different)
            float av = ha[4 * (l / 4) + i], bv = hb[l], cv = hc[l * 4 + i];
            float f = fmaf(av, bv, cv);
            volatile float p = av * bv; float m = p + cv;
            float g = hd[l * 4 + i];
            total++;
            if (memcmp(&g, &f, 4) == 0) n_fma++; else if (memcmp(&g, &m, 4) == 0) n_muladd++; else n_other++;
        }
    }
    printf("random: total %d  == fmaf %d  == mul+add only %d  neither %d\n", total, n_fma, n_muladd, n_other);
    // pass 3: denormals
    float tiny = 1e-30f, den = 1e-41f;
    float cases[6][3] = { { tiny, tiny, 0.0f }, { den, 1.0f, 0.0f }, { 1.0f, 1.0f, den }, { tiny, 1e-10f, den }, { den, den, 1.0f }, { 1e-20f, 1e-20f, 1e-40f } };
    for (int t = 0; t < 6; t++) {
        for (int i = 0; i < 64; i++) { ha[i] = cases[t][0]; hb[i] = cases[t][1]; }
        for (int i = 0; i < 256; i++) hc[i] = cases[t][2];
        (void)hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        (void)hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        printf("denormal case a=%g b=%g c=%g: mfma %g  fmaf %g\n", cases[t][0], cases[t][1], cases[t][2], hd[0], fmaf(cases[t][0], cases[t][1], cases[t][2]));
    }
    return 0;
}
