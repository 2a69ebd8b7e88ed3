// VALU issue-rate probe for gfx950: independent f32 ops per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
            a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
            a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
        } else if (KIND == 1) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(const double*)&b), "v"(*(const double*)&c));
        } else if (KIND == 2) {
            double d0 = a0, d1 = a1; (void)d0; (void)d1;
            asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(1.0001), "v"(0.5));
            asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(1.0001), "v"(0.5));
        } else if (KIND == 3) {
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 4) {
            asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 5) {
            asm volatile("v_pk_min_u16 %0, %0, %4\n v_pk_min_u16 %1, %1, %4\n v_pk_min_u16 %2, %2, %4\n v_pk_min_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 6) {
            asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 7) {
            asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_cmp_lt_i32 vcc, %1, %3\n v_cndmask_b32 %1, %1, %3, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");
        } else if (KIND == 9) {
            asm volatile("v_pk_sub_i16 %0, %0, %4\n v_pk_lshrrev_b16 %1, 15, %1\n v_pk_mad_u16 %2, %2, %4, %1\n v_pk_sub_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 10) {
            asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 11) {
            asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 12) {
            asm volatile("v_and_or_b32 %0, %0, %4, %4\n v_and_or_b32 %1, %1, %4, %4\n v_and_or_b32 %2, %2, %4, %4\n v_and_or_b32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 13) {
            asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 14) {
            asm volatile("v_cmp_lt_u16 vcc, %0, %1\n v_cmp_lt_u16 vcc, %1, %2\n v_cmp_lt_u16 vcc, %2, %3\n v_cmp_lt_u16 vcc, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");
        } else if (KIND == 15) {
            asm volatile("v_pk_sub_i16 %0, %0, %4\n v_pk_sub_i16 %1, %1, %4\n v_pk_sub_i16 %2, %2, %4\n v_pk_sub_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 16) {
            asm volatile("v_pk_lshrrev_b16 %0, 15, %0\n v_pk_lshrrev_b16 %1, 15, %1\n v_pk_lshrrev_b16 %2, 15, %2\n v_pk_lshrrev_b16 %3, 15, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 17) {
            asm volatile("v_pk_mad_u16 %0, %0, %4, %1\n v_pk_mad_u16 %1, %1, %4, %2\n v_pk_mad_u16 %2, %2, %4, %3\n v_pk_mad_u16 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 18) {
            asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 19) {
            asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (KIND == 20) {
            asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else {
            asm volatile("v_pk_fma_f16 %0, %0, %4, %4\n v_pk_fma_f16 %1, %1, %4, %4\n v_pk_fma_f16 %2, %2, %4, %4\n v_pk_fma_f16 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[0] = a0;
}
template <int KIND> void run(const char* name, int per_iter, int waves_per_simd)
{
    float* o; (void)hipMalloc(&o, 4);
    int iters = 20000;
    dim3 grid(256 * waves_per_simd), block(256);      // 1 block = 4 waves = 1 per SIMD of a CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, o, 100);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, o, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)iters * per_iter * waves_per_simd;
    printf("%-14s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main()
{
    for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", 8, w); }
    for (int w : {1, 4}) { run<1>("v_pk_fma_f32", 4, w); run<2>("v_fma_f64", 4, w); run<3>("v_rcp_f32", 4, w); }
    for (int w : {4, 8}) { run<4>("v_pk_add_u16", 4, w); run<5>("v_pk_min_u16", 4, w); run<6>("v_add_u32", 4, w); run<7>("cmp+cndmask+addc", 6, w); run<8>("v_pk_fma_f16", 4, w); run<9>("pk sub/lshr/mad", 4, w);
        run<10>("v_add3_u32", 4, w); run<11>("v_lshl_add_u32", 4, w); run<12>("v_and_or_b32", 4, w); run<13>("v_min_u32", 4, w);
        run<14>("v_cmp_lt_u16", 4, w); run<15>("v_pk_sub_i16", 4, w); run<16>("v_pk_lshrrev_b16", 4, w); run<17>("v_pk_mad_u16", 4, w);
        run<18>("v_sub_u32", 4, w); run<19>("v_bfi_b32", 4, w); run<20>("v_lshrrev_b32", 4, w); }
    return 0;
}
