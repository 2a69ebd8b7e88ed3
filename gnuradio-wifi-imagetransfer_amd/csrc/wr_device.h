// wr_device.h -- device-side building blocks of the wifirx kernels (gfx950 / CDNA4, wave64).
//
// Everything here follows the wifirx numerics spec: IEEE-754 binary32 operations in a
// fixed order, explicit fma where the spec writes fma, no contraction (compile with
// -ffp-contract=off), own sincos / atan2 / log2 polynomials.  The CPU oracle implements the same
// spec in plain C (oracle/wifirx_oracle.c); the two must agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WR_TABLE_QUAL __device__
#define WR_WANT_T4_TABLE
#define WR_WANT_LTS_MFMA_TABLE
#include "wifirx_tables.h"

namespace wr {

struct c32 { float re, im; };

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---- spec math ------------------------------------------------------------------------------
__device__ __forceinline__ void sp_sincos(float x, float& s, float& c)
{
    float kf = __builtin_rintf(x * WR_TWO_OVER_PI);
    int   k  = (int)kf;
    float r  = fma_(-kf, WR_PIO2_HI, x);
    r = fma_(-kf, WR_PIO2_MID, r);
    r = fma_(-kf, WR_PIO2_LO, r);
    float z  = r * r;
    float ps = fma_(z, WR_S3, WR_S2);
    ps = fma_(ps, z, WR_S1);
    float sr = fma_(ps * z, r, r);
    float pc = fma_(z, WR_C3, WR_C2);
    pc = fma_(pc, z, WR_C1);
    float cr = fma_(pc * z, z, fma_(-0.5f, z, 1.0f));
    // quadrant: swap for odd k, negate sin for k&2, negate cos for (k+1)&2
    bool  odd = k & 1;
    float s0 = odd ? cr : sr;
    float c0 = odd ? sr : cr;
    s = (k & 2) ? -s0 : s0;
    c = ((k + 1) & 2) ? -c0 : c0;
}

// sp_sincos for callers whose angles are usually small: when every lane has |x * 2/pi| < 0.5 the reduction
// of sp_sincos yields k = 0 and r = x exactly, so the polynomials alone give bit-identical results.
__device__ __forceinline__ void sp_sincos_small(float x, float& s, float& c)
{
    if (__all(__builtin_fabsf(x * WR_TWO_OVER_PI) < 0.5f)) {
        float z  = x * x;
        float ps = fma_(z, WR_S3, WR_S2);
        ps = fma_(ps, z, WR_S1);
        s = fma_(ps * z, x, x);
        float pc = fma_(z, WR_C3, WR_C2);
        pc = fma_(pc, z, WR_C1);
        c = fma_(pc * z, z, fma_(-0.5f, z, 1.0f));
    } else {
        sp_sincos(x, s, c);
    }
}

// sine/cosine of a phase given in 2^-62 quarter turns modulo 2^64 (spec section 4.8): reduction = shift, quadrant = top bits
__device__ __forceinline__ void sp_sincos_q(uint32_t p_hi, uint32_t p_lo, float& s, float& c)
{
    const uint32_t th = p_hi + 0x20000000u;
    const int      k  = (int)(th >> 30);
    const uint32_t uh = __builtin_amdgcn_alignbit(th, p_lo, 30);           // (th << 2) | (p_lo >> 30)
    const int      gh = (int)(uh ^ 0x80000000u);
    float r  = (float)gh * WR_PIO2_2M32;
    float z  = r * r;
    float ps = fma_(z, WR_S3, WR_S2);
    ps = fma_(ps, z, WR_S1);
    float sr = fma_(ps * z, r, r);
    float pc = fma_(z, WR_C3, WR_C2);
    pc = fma_(pc, z, WR_C1);
    float cr = fma_(pc * z, z, fma_(-0.5f, z, 1.0f));
    bool  odd = k & 1;
    float s0 = odd ? cr : sr;
    float c0 = odd ? sr : cr;
    s = (k & 2) ? -s0 : s0;
    c = ((k + 1) & 2) ? -c0 : c0;
}

// 1/sqrt(v) for normal positive v: seed by halving the exponent, three Newton steps y <- y (1.5 - 0.5 v y^2)
// (plain multiplies and fmas: bit-identical to the oracle's, relative error 1.3e-7; v_rsq_f32 would not be reproducible)
__device__ __forceinline__ float sp_rsqrt(float v)
{
    float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(v) >> 1));
    const float h = 0.5f * v;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float t = y * y;
        const float u = fma_(-h, t, 1.5f);
        y = y * u;
    }
    return y;
}

// 1/v for normal positive v (spec rule 11): seed by negating the exponent, three Newton steps y <- y + y (1 - v y);
// seven full-rate instructions instead of the division's ten with a quarter-rate v_rcp_f32 among them
__device__ __forceinline__ float sp_recip(float v)
{
    float y = __uint_as_float(0x7EF127EAu - __float_as_uint(v));
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float e = fma_(-v, y, 1.0f);
        y = fma_(y, e, y);
    }
    return y;
}

__device__ __forceinline__ float sp_atan2(float y, float x)
{
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float a   = mn / mx;
    bool  big = a > WR_TAN_PIO8;
    float t   = a;                                  // a/1
    if (__any(big)) {                               // wave-uniform: most calls see small angles only
        float num = big ? a - 1.0f : a;
        float den = big ? a + 1.0f : 1.0f;
        t = num / den;
    }
    float z   = t * t;
    float p   = fma_(z, WR_A4, WR_A3);
    p = fma_(p, z, WR_A2);
    p = fma_(p, z, WR_A1);
    float r = fma_(p * z, t, t);
    if (big) r = r + WR_PIO4;
    if (ay > ax) r = WR_PIO2 - r;
    if (x < 0.0f) r = WR_PI - r;
    if (y < 0.0f) r = -r;
    return (mx == 0.0f) ? 0.0f : r;
}

__device__ __forceinline__ float sp_log2(float v)
{
    uint32_t b  = __float_as_uint(v);
    int      e  = (int)((b >> 23) & 0xffu) - 127;
    float    m  = __uint_as_float((b & 0x7fffffu) | 0x3f800000u);
    if (m > WR_SQRT2) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = fma_(f, WR_L9, WR_L8);
    p = fma_(p, f, WR_L7);
    p = fma_(p, f, WR_L6);
    p = fma_(p, f, WR_L5);
    p = fma_(p, f, WR_L4);
    p = fma_(p, f, WR_L3);
    p = fma_(p, f, WR_L2);
    p = fma_(p, f, WR_L1);
    float ln = fma_(p * z, f, fma_(-0.5f, z, f));
    return fma_(ln, WR_LOG2E, (float)e);
}

__device__ __forceinline__ float sp_snr_db(float signal, float noise)
{
    float v = (signal / noise) / 2.0f;
    if (!(v > 1e-10f)) return -100.0f;
    if (v > 1e10f) return 100.0f;
    return WR_10LOG10_2 * sp_log2(v);
}

// x * (c + j s)
__device__ __forceinline__ c32 sp_rot(c32 x, float s, float c)
{
    c32 r;
    r.re = fma_(-x.im, s, x.re * c);
    r.im = fma_(x.im, c, x.re * s);
    return r;
}
__device__ __forceinline__ c32 sp_cmul(c32 x, c32 w)
{
    c32 r;
    r.re = fma_(-x.im, w.im, x.re * w.re);
    r.im = fma_(x.im, w.re, x.re * w.im);
    return r;
}
// conj(a) * b
__device__ __forceinline__ c32 sp_conj_mul(c32 a, c32 b)
{
    c32 r;
    r.re = fma_(a.im, b.im, a.re * b.re);
    r.im = fma_(-a.im, b.re, a.re * b.im);
    return r;
}
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return { a.re + b.re, a.im + b.im }; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return { a.re - b.re, a.im - b.im }; }
__device__ __forceinline__ c32 cneg(c32 a) { return { -a.re, -a.im }; }
// a with the sign bits of both parts xor-ed with m (m = 0 or 0x80000000: a or -a)
__device__ __forceinline__ c32 cflip(c32 a, uint32_t m)
{
    return { __uint_as_float(__float_as_uint(a.re) ^ m), __uint_as_float(__float_as_uint(a.im) ^ m) };
}

// ---- cross-lane helpers (wave64) --------------------------------------------------------------
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int   shfl(int v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ c32   shfl(c32 v, int src) { return { __shfl(v.re, src, 64), __shfl(v.im, src, 64) }; }
__device__ __forceinline__ float bcast(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
__device__ __forceinline__ c32   bcast(c32 v, int lane) { return { bcast(v.re, lane), bcast(v.im, lane) }; }

// DPP row shifts inside rows of 16 lanes; lanes shifted in from outside the row read 0.
template <int CTRL>
__device__ __forceinline__ float dpp_zero(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// inclusive Kogge-Stone prefix sum over each row of 16 lanes: v[i] += v[i-k], k = 1,2,4,8
__device__ __forceinline__ float row_prefix16(float v)
{
    v = dpp_zero<0x111>(v) + v;   // row_shr:1
    v = dpp_zero<0x112>(v) + v;   // row_shr:2
    v = dpp_zero<0x114>(v) + v;   // row_shr:4
    v = dpp_zero<0x118>(v) + v;   // row_shr:8
    return v;
}
// maximum over each row of 16 lanes, valid in the row's last lane (lanes shifted in from outside the row keep INT_MIN)
__device__ __forceinline__ int row_max16(int v)
{
#define WR_ROW_MAX_STEP(CTRL) { int o = __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, 0xf, 0xf, false); v = o > v ? o : v; }
    WR_ROW_MAX_STEP(0x111)   // row_shr:1
    WR_ROW_MAX_STEP(0x112)   // row_shr:2
    WR_ROW_MAX_STEP(0x114)   // row_shr:4
    WR_ROW_MAX_STEP(0x118)   // row_shr:8
#undef WR_ROW_MAX_STEP
    return v;
}
// inclusive suffix sum: v[i] += v[i+k]
__device__ __forceinline__ float row_suffix16(float v)
{
    v = dpp_zero<0x101>(v) + v;   // row_shl:1
    v = dpp_zero<0x102>(v) + v;
    v = dpp_zero<0x104>(v) + v;
    v = dpp_zero<0x108>(v) + v;
    return v;
}
}  // namespace wr
