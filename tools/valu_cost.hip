// Issue cost of the vector-ALU forms the demod kernel's symbol loop is made of (round 3; the kernel is bound by vector issue):
// 32 independent instructions of ONE form per loop iteration over 8 registers, W waves per SIMD, ns per wave-instruction
// per SIMD and the ratio to v_fma_f32.   hipcc --offload-arch=gfx950 -O3 tools/valu_cost.hip -o tools/valu_cost.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY(STR) REP8(STR) REP8(STR) REP8(STR) REP8(STR)

#define KERNEL(NAME, ASMSTR)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters)                                               \
    {                                                                                                                 \
        float a[8];                                                                                                   \
        for (int j = 0; j < 8; j++) a[j] = (float)(threadIdx.x + j);                                                  \
        float b = 1.0001f, c = 0.5f;                                                                                  \
        asm volatile("" : "+v"(b), "+v"(c));                                                                          \
        for (int i = 0; i < iters; i++) {                                                                             \
            asm volatile(ASMSTR : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                         : "v"(b), "v"(c) : "vcc", "s20", "s21");                                                                   \
        }                                                                                                             \
        float s = 0.f;                                                                                                \
        for (int j = 0; j < 8; j++) s += a[j];                                                                        \
        if (s == 123.456f) out[0] = s;                                                                                \
    }

#define X(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
KERNEL(k_fma, BODY(X))
#undef X
#define X(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
KERNEL(k_mul, BODY(X))
#undef X
#define X(n) "v_add_f32 %" #n ", %" #n ", %8\n"
KERNEL(k_add, BODY(X))
#undef X
#define X(n) "v_fmac_f32 %" #n ", %8, %9\n"
KERNEL(k_fmac, BODY(X))
#undef X
#define X(n) "v_fmaak_f32 %" #n ", %" #n ", %8, 0x3fc00000\n"
KERNEL(k_fmaak, BODY(X))
#undef X
#define X(n) "v_mul_f32 %" #n ", |%" #n "|, -%8\n"
KERNEL(k_mul_mod, BODY(X))
#undef X
#define X(n) "v_mov_b32 %" #n ", %8\n"
KERNEL(k_mov, BODY(X))
#undef X
#define X(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
KERNEL(k_xor, BODY(X))
#undef X
#define X(n) "v_and_b32 %" #n ", %" #n ", %8\n"
KERNEL(k_and, BODY(X))
#undef X
#define X(n) "v_add_u32 %" #n ", %" #n ", %8\n"
KERNEL(k_addu, BODY(X))
#undef X
#define X(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
KERNEL(k_shl, BODY(X))
#undef X
#define X(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
KERNEL(k_cnd, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n"
KERNEL(k_cmp, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
KERNEL(k_cmpcnd, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 s[20:21], %" #n ", %8\n"
KERNEL(k_cmp_s, BODY(X))
#undef X
#define X(n) "v_mov_b32_dpp %" #n ", %" #n " row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
KERNEL(k_dpp_bc, BODY(X))
#undef X
#define X(n) "v_add_f32_dpp %" #n ", %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
KERNEL(k_dpp_add, BODY(X))
#undef X
#define X(n) "v_max_f32 %" #n ", %" #n ", %8\n"
KERNEL(k_max, BODY(X))
#undef X
#define X(n) "v_min_i32 %" #n ", %" #n ", %8\n"
KERNEL(k_mini, BODY(X))
#undef X
#define X(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
KERNEL(k_cvt, BODY(X))
#undef X
#define X(n) "v_rsq_f32 %" #n ", %" #n "\n"
KERNEL(k_rsq, BODY(X))
#undef X
#define X(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
KERNEL(k_perm, BODY(X))
#undef X
#define X(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %8\n"
KERNEL(k_lshladd, BODY(X))
#undef X
#define X(n) "v_sub_f32 %" #n ", %8, %" #n "\n"
KERNEL(k_sub, BODY(X))
#undef X
#define X(n) "v_readlane_b32 s20, %" #n ", 5\n"
KERNEL(k_readlane, BODY(X))
#undef X


// ---- v_cndmask in context (it reads its mask through the scalar operand path) ----
#define X(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
KERNEL(k_cmp_2cnd, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
KERNEL(k_cmp_4cnd, BODY(X))
#undef X
#define X(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
KERNEL(k_cnd_sgpr, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 s[20:21], %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, s[20:21]\n"
KERNEL(k_cmps_cnd, BODY(X))
#undef X
#define X(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n v_add_f32 %" #n ", %" #n ", %9\n"
KERNEL(k_cnd_add, BODY(X))
#undef X
#define X(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n v_add_f32 %" #n ", %" #n ", %9\n v_add_f32 %" #n ", %" #n ", %9\n v_add_f32 %" #n ", %" #n ", %9\n"
KERNEL(k_cnd_3add, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n v_add_f32 %" #n ", %" #n ", %9\n v_add_f32 %" #n ", %" #n ", %9\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
KERNEL(k_cmp_2add_cnd, BODY(X))
#undef X
#define X(n) "v_mul_f32 %" #n ", %" #n ", %8\n v_max_f32 %" #n ", %" #n ", %9\n"
KERNEL(k_mul_max, BODY(X))
#undef X


#define X(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
KERNEL(k_cmp_3cnd, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_add_f32 %" #n ", %" #n ", %9\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n v_add_f32 %" #n ", %" #n ", %9\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_add_f32 %" #n ", %" #n ", %9\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
KERNEL(k_cmp_4cnd_spaced, BODY(X))
#undef X
#define X(n) "s_mov_b64 vcc, s[20:21]\n s_nop 3\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n v_add_f32 %" #n ", %" #n ", %9\n"
KERNEL(k_smov_cnd, BODY(X))
#undef X
#define X(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n"
KERNEL(k_cnd_e64_vcc, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n s_nop 1\n v_cndmask_b32_e64 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n v_cndmask_b32_e64 %" #n ", %" #n ", %9, vcc\n v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n"
KERNEL(k_cmp_4cnd_e64, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 s[20:21], %" #n ", %8\n s_nop 1\n v_cndmask_b32 %" #n ", %" #n ", %9, s[20:21]\n v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n v_cndmask_b32 %" #n ", %" #n ", %9, s[20:21]\n v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
KERNEL(k_cmps_4cnd, BODY(X))
#undef X
#define X(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n s_nop 1\n v_addc_co_u32 %" #n ", vcc, %" #n ", %8, vcc\n"
KERNEL(k_cmp_addc, BODY(X))
#undef X

template <typename K> static double run(K kern, int w)
{
    float* o; (void)hipMalloc(&o, 4);
    const int iters = 4000;
    dim3 grid(256 * w), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, o, 50);
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, grid, block, 0, 0, o, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / ((double)iters * 32 * w);
        if (ns < best) best = ns;
    }
    (void)hipFree(o);
    return best;
}

int main()
{
    for (int w : {4, 8}) {
        const double f = run(k_fma, w);
        printf("-- %d waves per SIMD; v_fma_f32 = %.3f ns per wave-instruction per SIMD\n", w, f);
#define R(name, kern, per) { const double t = run(kern, w) / per; printf("%-34s %.3f ns  = %.2f x v_fma_f32\n", name, t, t / f); }
        R("v_mul_f32", k_mul, 1) R("v_add_f32", k_add, 1) R("v_sub_f32", k_sub, 1) R("v_fmac_f32", k_fmac, 1) R("v_fmaak_f32 (literal)", k_fmaak, 1)
        R("v_mul_f32 |a|, -b (VOP3)", k_mul_mod, 1) R("v_max_f32", k_max, 1) R("v_mov_b32", k_mov, 1) R("v_xor_b32", k_xor, 1) R("v_and_b32", k_and, 1)
        R("v_add_u32", k_addu, 1) R("v_lshlrev_b32", k_shl, 1) R("v_lshl_add_u32", k_lshladd, 1) R("v_min_i32", k_mini, 1) R("v_perm_b32", k_perm, 1)
        R("v_cndmask_b32 (vcc)", k_cnd, 1) R("v_cmp_gt_f32 -> vcc", k_cmp, 1) R("v_cmp_gt_f32 -> sgpr pair", k_cmp_s, 1)
        R("v_cmp, s_nop 1, v_cndmask (per pair)", k_cmpcnd, 1) R("v_mov_b32_dpp row_newbcast", k_dpp_bc, 1) R("v_add_f32_dpp row_shr", k_dpp_add, 1)
        R("v_cmp, nop, 2 x v_cndmask (per group of 3)", k_cmp_2cnd, 1) R("v_cmp, nop, 4 x v_cndmask (per group of 5)", k_cmp_4cnd, 1)
        R("v_cndmask_b32 with an SGPR-pair mask", k_cnd_sgpr, 1) R("v_cmp -> sgpr, nop, v_cndmask sgpr (per pair)", k_cmps_cnd, 1)
        R("v_cndmask, v_add_f32 (per pair)", k_cnd_add, 1) R("v_cndmask, 3 x v_add_f32 (per group of 4)", k_cnd_3add, 1)
        R("v_cmp, 2 x v_add_f32, v_cndmask (per group of 4)", k_cmp_2add_cnd, 1) R("v_mul_f32, v_max_f32 (per pair)", k_mul_max, 1)
        R("v_cmp, nop, 3 x v_cndmask vcc (per group)", k_cmp_3cnd, 1) R("v_cmp, nop, 4 x (v_cndmask vcc, v_add) (per group)", k_cmp_4cnd_spaced, 1)
        R("s_mov vcc, nop 3, v_cndmask, v_add (per group)", k_smov_cnd, 1) R("v_cndmask_b32_e64 ..., vcc", k_cnd_e64_vcc, 1)
        R("v_cmp, nop, 4 x v_cndmask_e64 vcc (per group)", k_cmp_4cnd_e64, 1) R("v_cmp -> sgpr, nop, 4 x v_cndmask sgpr (per group)", k_cmps_4cnd, 1)
        R("v_cmp, nop, v_addc_co (per pair)", k_cmp_addc, 1)
        R("v_cvt_f32_i32", k_cvt, 1) R("v_rsq_f32", k_rsq, 1) R("v_readlane_b32", k_readlane, 1)
    }
    return 0;
}
