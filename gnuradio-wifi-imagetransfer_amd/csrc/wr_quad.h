// wr_quad.h -- the per-frame part of the chain, four frames per wavefront.
//
// Layout: row f = lanes 16f..16f+15 owns frame f of the wave; lane r of a row holds the four
// sub-carriers i = r + 16 j (j = 0..3) of the current OFDM symbol.  The four frames walk their
// symbols in lock step, so
//   * the per-frame scalar chain of frame_equalizer (pilot phase, residual-offset IIR, two atan2, the
//     double-precision sampling-offset factor) is computed once per row, i.e. for 4 frames per
//     instruction instead of 1;
//   * the radix-4 FFT runs in registers: stage k of a lane is one 4-point butterfly; between stages the
//     64 values of a row are transposed through 512 B of wave-private LDS;
//   * pilots reach every lane of their row by DPP row broadcast, not through the LDS crossbar.
// The preamble work (short-preamble scan, 64-tap LTS correlation over 320 lags) keeps lane <-> sample /
// lag and is done for the four frames one after the other by the whole wave.
//
// Arithmetic is the numerics spec of DESIGN.md section 4, value for value: only the lane that computes a value
// changed with respect to the one-frame-per-wave version, so outputs stay bit-identical to the oracle.
#pragma once
#include "wr_device.h"
#include "wr_kernels.h"
#include <type_traits>

namespace wr {

#ifndef WR_SPLIT_SYMBOL_LOOP
#define WR_SPLIT_SYMBOL_LOOP 1
#endif
#ifndef WR_PLAIN_STORES
#define WR_PLAIN_STORES 1
#endif
#ifndef WR_CONST_DATA_MASK
#define WR_CONST_DATA_MASK 1
#endif
#ifndef WR_NB_LOOPS
#define WR_NB_LOOPS 2       // 1: BPSK and QPSK only, 2: all four constellations (loops of their own for 16- and 64-QAM bring nothing by themselves; they carry the whole-line stores)
#endif
#ifndef WR_POLARITY_WINDOW
#define WR_POLARITY_WINDOW 1
#endif
#ifndef WR_T4_POINTER
#define WR_T4_POINTER 1
#endif
#ifndef WR_STORE_AS_LINES
#define WR_STORE_AS_LINES 3       // 1: BPSK / QPSK rows in the kernels without plane output only, 2: in both, 3: 16- / 64-QAM rows as well (needs WR_NB_LOOPS 2)
#endif
#ifndef WR_FLAT_STAGING
#define WR_FLAT_STAGING 1        // store_bins_lines: the four bins of a lane staged without per-bin execution masks (BPSK .. 16-QAM)
#endif
#ifndef WR_GLOBAL_SAMPLE_LOADS
#define WR_GLOBAL_SAMPLE_LOADS 1
#endif
#ifndef WR_DMA_PREFETCH
#define WR_DMA_PREFETCH 1        // BPSK / QPSK loops of the usual output set: the next symbol's samples by LDS-DMA (global_load_lds_dwordx4), requested while this symbol is computed
#endif
#define WR_QLDS_PFX (WR_DMA_PREFETCH ? 256 : 0)       // floats behind the LLR-weight area that complete the 2-kB prefetch buffer (4 frames x 64 samples x 8 B)
#ifndef WR_LTS_SKIP_IDLE
#define WR_LTS_SKIP_IDLE 1       // LTS candidate rounds 3..7 run for the frame of a pair that needs them only (was: both frames of the pair)
#endif
#ifndef WR_X_LOOPS
#define WR_X_LOOPS 1             // constellation loops with whole-line stores for the other output sets too (carrier, weights, moments, planes alone)
#endif
#ifndef WR_NT_LOADS
#define WR_NT_LOADS 1            // the symbol loop's sample loads as non-temporal (streaming) loads
#endif
#ifndef WR_NT_STORES
#define WR_NT_STORES 1           // store_bins_lines / store_rows_x: the whole-line 16-byte pieces as non-temporal stores (2: the decisions' dwords too -- slower)
#endif
// LDS of the symbol loop, per wave (floats).  The scratch area at the front serves the FFT transposes (512 floats), the SIGNAL
// decoder's survivor words, STA's window exchange and the staging of a symbol's output rows (store_bins_lines: 432 floats
// for BPSK / QPSK rows, 1200 for 64-QAM = 4 x 1152 B of LLRs + 4 x 48 decisions).  The COMB instance keeps the short scratch
// area (its running estimate d_H takes another 2 kB; with the long one sixteen waves would no longer fit a CU's 160 kB) and
// with it the per-bin stores for 16- / 64-QAM rows.
#ifndef WR_QLDS_LONG
#define WR_QLDS_LONG 1200
#endif
#define WR_QLDS_SCRATCH_EQ(EQ) ((EQ) == WIFIRX_EQ_COMB ? 768 : WR_QLDS_LONG)
#define WR_QLDS_H(S)       (S)                        // 4 x 64 float2: channel estimate, lane-private slots
#define WR_QLDS_TW(S)      (WR_QLDS_H(S) + 512)       // 6 x 16 float2: stage-1/2 twiddles by row lane
#define WR_QLDS_PREV(S)    (WR_QLDS_TW(S) + 192)      // 4 rows x 4 float2: pilots of the previous symbol (LMS / COMB / STA: rows of 6 float2, the fifth = the frame's phase increment Qp)
#define WR_QLDS_W(S)       (WR_QLDS_PREV(S) + 48)     // 4 x 64 floats: |H|^2 of the LS estimate (LLR weight, lane-private slots)
#define WR_QLDS_PF(S)      (WR_QLDS_W(S))             // the prefetch buffer of the usual output set's loops: the weight area (idle there: weights are an XK output) + WR_QLDS_PFX
#define WR_QLDS_STAT(S)    (WR_QLDS_W(S) + 256 + WR_QLDS_PFX)       // 4 rows x 4 floats: running sums of |y|, |y|^2, |y|^4 (sym_stats output)
#define WR_QLDS_FLOATS(S)  (WR_QLDS_STAT(S) + 16)     // per wave; the preamble phase uses the first 1536 floats for two frames' samples
#define WR_QLDS_DH(S)      (WR_QLDS_FLOATS(S))        // COMB only: 4 x 64 float2, the running estimate d_H
#define WR_QLDS_FLOATS_EQ(EQ) (WR_QLDS_FLOATS(WR_QLDS_SCRATCH_EQ(EQ)) + ((EQ) == WIFIRX_EQ_COMB ? 512 : 0))

// a sample through the global-memory path: the symbol loop's sample pointer is rebuilt from lane exchanges, which hides
// its address space from the compiler (it would take the flat path, which also counts on the LDS counter)
typedef float wr_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 load_global_f2(const float2* p)
{
#if WR_GLOBAL_SAMPLE_LOADS && defined(__HIP_DEVICE_COMPILE__)
#if WR_NT_LOADS
    const wr_f2 v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) wr_f2*>(reinterpret_cast<uintptr_t>(p)));
#else
    const wr_f2 v = *reinterpret_cast<const __attribute__((address_space(1))) wr_f2*>(reinterpret_cast<uintptr_t>(p));
#endif
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}

// a wave-uniform float of a constant table through the scalar data cache (s_load_dword: counted on lgkmcnt, not on the vector
// memory counter the LDS-DMA prefetch waits on)
__device__ __forceinline__ float load_const_f(const float* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<uintptr_t>(p));
#else
    return *p;
#endif
}

__device__ __forceinline__ c32 load_sample(const float2* __restrict__ x, long n, long n_samp)
{
    c32 z = { 0.0f, 0.0f };
    if (n >= 0 && n < n_samp) {
        float2 t = x[n];
        z.re = t.x;
        z.im = t.y;
    }
    return z;
}

// y[m] of a frame, zero outside [m_lo, m_hi): branch-free (clamped address, selected value)
__device__ __forceinline__ c32 load_y(const float2* __restrict__ xb, int m, int m_lo, int m_hi)
{
    int mc = m < m_hi - 1 ? m : m_hi - 1;
    mc = mc > m_lo ? mc : m_lo;
    const bool ok = (m >= m_lo) && (m < m_hi);
    float2 t = make_float2(0.0f, 0.0f);
    if (m_hi > m_lo) t = load_global_f2(xb + mc);
    return { ok ? t.x : 0.0f, ok ? t.y : 0.0f };
}

__device__ __forceinline__ uint8_t decide(c32 y, int n_bpsc)
{
    float re = y.re, im = y.im, are = __builtin_fabsf(re), aim = __builtin_fabsf(im);
    unsigned r;
    if (n_bpsc == 1) {
        r = re > 0.0f;
    } else if (n_bpsc == 2) {
        r = (re > 0.0f) | ((im > 0.0f) << 1);
    } else if (n_bpsc == 4) {
        r = (re > 0.0f) | ((are < WR_T16_2) << 1) | ((im > 0.0f) << 2) | ((aim < WR_T16_2) << 3);
    } else {
        r = (re > 0.0f) | ((are < WR_T64_4) << 1) | (((are < WR_T64_6) && (are > WR_T64_2)) << 2) |
            ((im > 0.0f) << 3) | ((aim < WR_T64_4) << 4) | (((aim < WR_T64_6) && (aim > WR_T64_2)) << 5);
    }
    return (uint8_t)r;
}

// Viterbi over the 24 SIGNAL bits of the FOUR frames of a wave at once, lane <-> state.  cb[f]: the 48 de-interleaved hard
// decisions of frame f (bit j = coded bit j; 0 for a row without a frame).  sig[f] = its 24 decoded bits (bit t = decoded
// bit t), wave-uniform.  The path metrics of the four frames share one register per lane, a byte each: a metric is at most
// 48, and the "unreachable" start value only has to exceed that (64; the oracle's 2^28 orders every comparison the same
// way: each path has one origin) -- so no byte ever reaches 128 and the byte-wise compare is one subtraction with a guard
// bit: byte f of (m0 + 0x7f7f7f7f) - m1 has its top bit set iff m1 < m0.  Survivor bits stay in the lane of their state (below);
// the trace-back runs in lanes 0..3, one frame each.
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ void viterbi_signal4(const uint64_t (&cb)[4], int lane, uint32_t (&sig)[4])
{
    const int s = lane, u = s & 1, p0 = s >> 1, p1 = (s >> 1) | 32;
    // the branch bits of the transition p0 -> s; those of p1 -> s are their complements (bit 6 is a tap of both generators:
    // 0155, 0117), so its branch metric is 2 - (that of p0 -> s) in every byte
    const int f0 = (p0 << 1) | u;
    const uint32_t A0 = (__builtin_popcount(f0 & 0155) & 1) * 0x01010101u, B0 = (__builtin_popcount(f0 & 0117) & 1) * 0x01010101u;
    uint32_t pm = (s == 0) ? 0u : 0x40404040u;
    // received bits, frame f in byte f (bit 0: first coded bit of the step, bit 1: second): lane t forms those of step t once, the
    // loop fetches them with one lane read and splits them on the scalar side
    uint32_t r2_l = 0;
#pragma unroll
    for (int f = 0; f < 4; f++) r2_l |= ((uint32_t)(cb[f] >> (2 * (lane & 31))) & 3u) << (8 * f);
    // Survivor bits (round 5): every lane = state keeps ITS OWN decisions, byte f of a word = frame f, bit i = step 8 k + i -- two
    // instructions per step.  (Until round 5: four ballots per step and lane 0 writing four 64-bit words to LDS -- 24 instructions
    // per step, half of the decoder's.)  The trace-back fetches the word of the state it stands in across the lanes.  Three rounds
    // of eight steps, NOT unrolled across the rounds: fully unrolled, the scheduler hoists the 48 lane reads to the front and the
    // scalar registers they need spill into vector lanes of the whole kernel (scratch 24 -> 44 bytes in the LS instance).
    uint32_t h0 = 0u, h1 = 0u, h2 = 0u;                                    // steps 16..23, 8..15, 0..7 after the loop
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
        uint32_t acc = 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int t = 8 * k + i;
            const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)r2_l, t);
            const uint32_t ra = r2 & 0x01010101u, rb = (r2 >> 1) & 0x01010101u;
            const uint32_t x0 = (ra ^ A0) + (rb ^ B0);
            const uint32_t m0 = (uint32_t)__shfl((int)pm, p0, 64) + x0;
            const uint32_t m1 = (uint32_t)__shfl((int)pm, p1, 64) + (0x02020202u - x0);
            const uint32_t top = ((m0 + 0x7f7f7f7fu) - m1) & 0x80808080u;     // byte f: 0x80 iff m1 < m0
            const uint32_t mask = (top - (top >> 7)) | top;                   // ... 0xff
            pm = (m1 & mask) | (m0 & ~mask);
            acc |= top >> (7 - i);
        }
        h2 = h1; h1 = h0; h0 = acc;
    }
    // best final state per frame: smallest metric, lowest index on ties -- keys (metric << 6 | state) of two frames per register
    uint32_t k01 = ((((pm & 0xffu) << 6) | (uint32_t)s)) | (((((pm >> 8) & 0xffu) << 6) | (uint32_t)s) << 16);
    uint32_t k23 = (((((pm >> 16) & 0xffu) << 6) | (uint32_t)s)) | ((((pm >> 24) << 6) | (uint32_t)s) << 16);
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        k01 = pk_min_u16(k01, (uint32_t)__shfl_xor((int)k01, k, 64));
        k23 = pk_min_u16(k23, (uint32_t)__shfl_xor((int)k23, k, 64));
    }
    // trace-back: lane f = frame f (the other lanes repeat frames 0..3 and are not read)
    const int fr = lane & 3;
    const uint32_t kk = (fr & 2) ? k23 : k01;
    int st = (int)(((fr & 1) ? (kk >> 16) : kk) & 63u);
    uint32_t bits = 0;
#pragma unroll 1
    for (int k = 2; k >= 0; k--) {
        uint32_t b8 = 0;
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            b8 |= (uint32_t)(st & 1) << i;
            const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * st, (int)h0);      // the decisions of state st
            const uint32_t h = (w >> (8 * fr + i)) & 1u;
            st = (st >> 1) | (int)(h << 5);
        }
        bits |= b8 << (8 * k);
        h0 = h1; h1 = h2;
    }
#pragma unroll
    for (int f = 0; f < 4; f++) sig[f] = (uint32_t)__builtin_amdgcn_readlane((int)bits, f);
}

__device__ __forceinline__ bool parse_signal(uint32_t bits, int& enc, int& len)
{
    int par = __builtin_popcount(bits & 0x1ffff) & 1;
    if (par != (int)((bits >> 17) & 1)) return false;
    int r = bits & 15;
    len = (bits >> 5) & 0xfff;
    switch (r) {
    case 11: enc = 0; break;
    case 15: enc = 1; break;
    case 10: enc = 2; break;
    case 14: enc = 3; break;
    case 9:  enc = 4; break;
    case 13: enc = 5; break;
    case 8:  enc = 6; break;
    case 12: enc = 7; break;
    default: return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// a2 copy + a3: sync_short's first 384 copied samples, LTS correlation, frame start, fine CFO -- for TWO frames at a
// time ("pair"): their derotated samples sit side by side in LDS, as float32 and as 8-bit integers.  WHERE the LTS peaks
// are is searched on the integers -- the 2 x 320 x 64 complex correlation is 15 v_mfma_i32_16x16x64_i8 on the matrix
// cores --, WHAT they are (the values the fine CFO is taken from) is computed in float32 for the two (rarely eight)
// candidates of a frame alone (spec rule 6).
//
// LDS: frame e (0/1) of the pair at floats [768 e, 768 e + 768): y[m] = (re, im) interleaved, m = 0..383; its 8-bit
// image at byte 4 * WR_PRE_Q8 + 832 e: 768 bytes (re, im) interleaved + 64 zero bytes (the rows of the last lag blocks
// run past the samples; their taps are zero there).
#define WR_PRE_FRAME_FLOATS 768
#define WR_PRE_Q8           1536                 // float index where the 8-bit images start
#define WR_PRE_Q8_FRAME     832                  // bytes per frame
#define WR_QLDS_PRE_FLOATS  (WR_PRE_Q8 + 2 * WR_PRE_Q8_FRAME / 4)      // LDS of the preamble phase, per wave

typedef int   wr_i4 __attribute__((ext_vector_type(4)));
typedef float wr_f4 __attribute__((ext_vector_type(4)));

// One frame before the correlation: what sync_short found, and whether the LTS search can run.
struct PreFrame {
    const float2* x;    // always a readable pointer
    long  n_samp;       // 0: no such slot
    long  t, L;         // trigger (-1: none), usable copied samples
    float cfo_c;
    bool  search;       // enough copied samples for the LTS search
    long  out;          // record / output row, -1: none
};

// Coarse derotation, y[m] = x[t - 16 + m] exp(-j float(cfo_c m)), m = 0..383 (lane <-> sample, six passes).
// The samples are requested by preamble_load() -- for all frames of a wave together, before the first of them is
// needed --, rotated and written to LDS pair by pair by preamble_derotate_pair().  A frame without a search loads
// nothing (its LDS rows are never looked at).
struct PreSamples { c32 xs[6]; };

__device__ __forceinline__ void preamble_load(const PreFrame& f, int lane, PreSamples& ps)
{
    const long ns = f.search ? f.n_samp : 0;
    if (f.search && f.t >= 16 && f.t + 368 <= f.n_samp) {
        // (wave-uniform; the usual case) all 384 samples lie inside the slot / stream buffer: one pointer, six loads at constant
        // offsets -- the per-sample range tests of load_sample() are 64-bit compares and a branch around every load
        const float2* p = f.x + (f.t - 16) + lane;
#pragma unroll
        for (int pass = 0; pass < 6; pass++) { const float2 v = p[64 * pass]; ps.xs[pass] = { v.x, v.y }; }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < 6; pass++) ps.xs[pass] = load_sample(f.x, f.t - 16 + (pass * 64 + lane), ns);
}

__device__ __forceinline__ int wave_max_int(int v)
{
    v = row_max16(v);
    const int m0 = __builtin_amdgcn_readlane(v, 15), m1 = __builtin_amdgcn_readlane(v, 31);
    const int m2 = __builtin_amdgcn_readlane(v, 47), m3 = __builtin_amdgcn_readlane(v, 63);
    const int m01 = m0 > m1 ? m0 : m1, m23 = m2 > m3 ? m2 : m3;
    return m01 > m23 ? m01 : m23;
}

// spec rule 6, stage 1: a component scaled into the 8-bit range, rounded to nearest even and clamped to [-127, 127]; NaN -> 0.
// v_cvt_i32_f32 saturates and turns a NaN into 0 (the integer clamp then does the rest), which is the rule's wording.
__device__ __forceinline__ int quant8(float x, float scale)
{
    const float r = __builtin_rintf(x * scale);
    int q;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(q) : "v"(r));
    q = q > 127 ? 127 : q;
    return q < -127 ? -127 : q;
}

// w64_0 / w64_1: exp(-j float(cfo_c 64)) of the two frames (spec rule 5a), formed by the caller for all frames of the wave in ONE
// pass of the sine / cosine (row f of the wave computes frame f's; the values are wave-uniform here).
__device__ __forceinline__ c32 preamble_w64_rows(float cfo_c_row)
{
    c32 w;
    sp_sincos(-cfo_c_row * 64.0f, w.im, w.re);
    return w;
}

__device__ __forceinline__ void preamble_derotate_pair(const PreFrame& f0, const PreFrame& f1, const PreSamples& s0,
                                                       const PreSamples& s1, c32 w64_0, c32 w64_1, float* ylds, int lane)
{
    if (!(f0.search || f1.search)) return;                  // wave-uniform
    // Both frames in one piece of straight-line code: their chains (two sincos, the carried phasor, the maximum over the
    // wave) are independent and interleave.  A frame without a search runs along on zeros; nothing reads its area.
    c32 yv[2][6];
    uint32_t mx[2] = { 0u, 0u };
    // spec rule 5a: the phasor of sample m = 64 p + l is that of sample l (exact: sp_sincos of the float angle) carried
    // p times by exp(-j float(cfo_c 64)) -- two sincos per lane and frame instead of six
    c32 wl[2], w64[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const float cfo = e ? f1.cfo_c : f0.cfo_c;
        sp_sincos(-cfo * (float)lane, wl[e].im, wl[e].re);
        w64[e] = e ? w64_1 : w64_0;
    }
#pragma unroll
    for (int pass = 0; pass < 6; pass++) {
        const int m = pass * 64 + lane;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if (pass) wl[e] = sp_cmul(wl[e], w64[e]);
            const c32 y = sp_cmul(e ? s1.xs[pass] : s0.xs[pass], wl[e]);
            *reinterpret_cast<float2*>(ylds + WR_PRE_FRAME_FLOATS * e + 2 * m) = make_float2(y.re, y.im);
            yv[e][pass] = y;
            const uint32_t br = __float_as_uint(y.re) & 0x7fffffffu, bi = __float_as_uint(y.im) & 0x7fffffffu;
            mx[e] = br > mx[e] ? br : mx[e];
            mx[e] = bi > mx[e] ? bi : mx[e];
        }
    }
    // the 8-bit image: a power of two puts the largest component of the frame into [64, 128)
    float scale[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int E = wave_max_int((int)mx[e]) >> 23;        // largest biased exponent (255: an Inf or NaN among the samples)
        const int sfield = 260 - E > 254 ? 254 : 260 - E;
        scale[e] = __uint_as_float((uint32_t)sfield << 23);
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
        uint8_t* q8 = reinterpret_cast<uint8_t*>(ylds + WR_PRE_Q8) + WR_PRE_Q8_FRAME * e;
#pragma unroll
        for (int pass = 0; pass < 6; pass++) {
            const int m = pass * 64 + lane;
            const int qr = quant8(yv[e][pass].re, scale[e]), qi = quant8(yv[e][pass].im, scale[e]);
            *reinterpret_cast<uint16_t*>(q8 + 2 * m) = (uint16_t)((qr & 0xff) | ((qi & 0xff) << 8));
        }
        if (lane < 16) *reinterpret_cast<uint32_t*>(q8 + 768 + 4 * lane) = 0u;
    }
}

// Stage 1 of the search: corr_q[i] = sum_k conj(lq[k]) yq[i + k] for the 320 lags of both frames of the pair in exact
// integer arithmetic, as a GEMM on v_mfma_i32_16x16x64_i8 (operand layout: tools/mfma_i8_probe.hip).  Round 5: the TAPS are
// the row-side operand (WR_LTS_MFMA_A8), the SAMPLES the column side, so that the real and the imaginary sum of a lag land
// in the same lane -- |corr_q|^2 then needs no exchange, and all 64 lanes hold lags (with the samples on the row side the
// two sums sat 8 lanes apart and half the lanes idled through the candidate search).  Lags in blocks of 8, i = 8 a + b:
//   N: 80 columns (frame e, block a) = 5 tiles of 16; the column of (e, a) is the 144 contiguous bytes from sample 8a on
//      (read as 192: the taps beyond are zero);
//   M: 16 rows = (b, part): row r' = 4 q + rho <-> b = 2 q + (rho >> 1), part = rho & 1;      K: 192 = 3 instructions of 64.
// Instruction t takes the bytes phi = 64 t + 16 kk + j (kk = lane >> 4, j = 0..15): one ds_read_b128 per lane, tile and t;
// the matching tap bytes come as one 16-byte load from WR_LTS_MFMA_A8.
// Tiles 0, 1 = frame 0 blocks 0..31; tile 2 = frame 0 blocks 32..39 (columns 0..7) and frame 1 blocks 0..7 (columns 8..15);
// tiles 3, 4 = frame 1 blocks 8..39.  Result register rho of lane (q = lane >> 4, col = lane & 15) of a tile: rho = 0 / 1 the
// real / imaginary sum of lag 8 a(col) + 2 q, rho = 2 / 3 those of lag 8 a(col) + 2 q + 1.
__device__ __forceinline__ void lts_corr_pair_q8(const float* ylds, int lane, wr_i4 (&acc)[5])
{
    const int col = lane & 15, kk = lane >> 4;
    const uint8_t* q8 = reinterpret_cast<const uint8_t*>(ylds + WR_PRE_Q8) + 16 * kk;
    const uint8_t* q8b = q8 + WR_PRE_Q8_FRAME;
    const uint8_t* bcol[5];
    bcol[0] = q8 + 16 * col;
    bcol[1] = q8 + 16 * (16 + col);
    bcol[2] = col < 8 ? q8 + 16 * (32 + col) : q8b + 16 * (col - 8);
    bcol[3] = q8b + 16 * (8 + col);
    bcol[4] = q8b + 16 * (24 + col);
    const wr_i4* __restrict__ at = reinterpret_cast<const wr_i4*>(WR_LTS_MFMA_A8) + lane;
#pragma unroll
    for (int t = 0; t < 5; t++) acc[t] = wr_i4{ 0, 0, 0, 0 };
#pragma unroll
    for (int t3 = 0; t3 < 3; t3++) {
        const wr_i4 aq = at[64 * t3];
#pragma unroll
        for (int t = 0; t < 5; t++) {
            const wr_i4 bq = *reinterpret_cast<const wr_i4*>(bcol[t] + 64 * t3);
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aq, bq, acc[t], 0, 0, 0);
        }
    }
}

// Stage 1 for both frames of a pair: the candidate lags.  Lane (q, col) holds six slots per frame, slot s = 2 T + h of the
// frame's tile T = 0..2 (h = rho >> 1): lag = L + 128 (s >> 1) + (s & 1) with L = 8 col + 2 q - 64 e; of the shared tile the
// half that belongs to the other frame is -1 (frame 0: slots 4, 5 of the columns 8..15; frame 1: slots 0, 1 of the columns
// 0..7).  The integers are below 2^24, so their float32 images are exact and |corr_q|^2 = fma(im, im, re re) is the same number
// on the CPU; it is >= +0 and no NaN, so its bit pattern orders like a signed integer; already-taken entries are -1.  Per round:
// the lane maximum, one integer max over the wave, the lane that holds it (a ballot), that lane's six values through the scalar
// unit (the first slot that matches: the lowest lag inside a lane), one v_writelane to strike it out.  Several lanes with the
// maximum (rare): the lowest lag among all entries that hold it, by a minimum over the wave.  The two largest when they are
// exactly 64 lags apart and the third largest value is below 7/8 of the second (spec rule 6: no rounding of the integer stage can
// then have changed the leaders); otherwise the eight largest; n_cand = their number.
// v with lane `lane_s` (wave-uniform) set to -1 (one scalar register per VOP3 on gfx9: the value is the inline constant)
__device__ __forceinline__ int strike_lane(int v, int lane_s)
{
    asm("v_writelane_b32 %0, -1, %1" : "+v"(v) : "s"(lane_s));
    return v;
}

__device__ __forceinline__ int lts_mag_key(int re, int im)
{
    const float fr = (float)re, fi = (float)im;
    return (int)__float_as_uint(fma_(fi, fi, fr * fr));
}

__device__ __forceinline__ void lts_candidates2(const wr_i4 (&acc)[5], const bool (&run)[2], int lane,
                                                int (&cand)[2][8], int (&n_cand)[2])
{
    const int col = lane & 15;
    const bool lo = col < 8;
    int K[2][6];
    {
        const int s0 = lts_mag_key(acc[2][0], acc[2][1]), s1 = lts_mag_key(acc[2][2], acc[2][3]);
        K[0][0] = lts_mag_key(acc[0][0], acc[0][1]); K[0][1] = lts_mag_key(acc[0][2], acc[0][3]);
        K[0][2] = lts_mag_key(acc[1][0], acc[1][1]); K[0][3] = lts_mag_key(acc[1][2], acc[1][3]);
        K[0][4] = lo ? s0 : -1;                      K[0][5] = lo ? s1 : -1;
        K[1][0] = lo ? -1 : s0;                      K[1][1] = lo ? -1 : s1;
        K[1][2] = lts_mag_key(acc[3][0], acc[3][1]); K[1][3] = lts_mag_key(acc[3][2], acc[3][3]);
        K[1][4] = lts_mag_key(acc[4][0], acc[4][1]); K[1][5] = lts_mag_key(acc[4][2], acc[4][3]);
    }
    const int L0 = 8 * col + 2 * (lane >> 4);
    n_cand[0] = n_cand[1] = 0;
    bool need[2] = { run[0], run[1] };
    int second[2] = { 0, 0 };                                   // the second largest |corr_q|^2 (bit pattern)
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
        for (int e = 0; e < 2; e++) cand[e][r] = 0;
        if (!(need[0] || need[1])) continue;
        int best[2] = { 0, 0 }, lmax[2] = { 0, 0 };
#pragma unroll
        for (int e = 0; e < 2; e++) {                           // rounds 0..2: no branch in here (both frames run them); the candidate rounds
            if (WR_LTS_SKIP_IDLE && r > 2 && !need[e]) continue; // behind them only for the frame that needs them (wave-uniform)
            int m = K[e][0];
#pragma unroll
            for (int n = 1; n < 6; n++) m = K[e][n] > m ? K[e][n] : m;
            lmax[e] = m;
            best[e] = wave_max_int(m);
            if (r == 1) second[e] = best[e];
        }
        if (r == 2) {
            // the two leaders stand when they are 64 lags apart and the third largest value is below 7/8 of the second
            // (wave-uniform: the usual case); otherwise the eight largest are the candidates
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int d01 = cand[e][0] > cand[e][1] ? cand[e][0] - cand[e][1] : cand[e][1] - cand[e][0];
                if (d01 == 64 && __int_as_float(best[e]) < 0.875f * __int_as_float(second[e])) need[e] = false;
            }
            if (!(need[0] || need[1])) continue;
        }
        int w[2] = { 0, 0 };
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if (WR_LTS_SKIP_IDLE && r >= 2 && !need[e]) continue;
            const int L = L0 - 64 * e;
            const uint64_t hit = __ballot(lmax[e] == best[e]);
            if (__builtin_popcountll(hit) == 1) {               // wave-uniform; the usual case
                const int wl = (int)__builtin_ctzll(hit);
                int sl = 5;
#pragma unroll
                for (int n = 4; n >= 0; n--) sl = __builtin_amdgcn_readlane(K[e][n], wl) == best[e] ? n : sl;
                w[e] = __builtin_amdgcn_readlane(L, wl) + 128 * (sl >> 1) + (sl & 1);
                // (a switch over the slot, not a select per slot: one v_writelane is issued, not six)
                switch (sl) {
                case 0:  K[e][0] = strike_lane(K[e][0], wl); break;
                case 1:  K[e][1] = strike_lane(K[e][1], wl); break;
                case 2:  K[e][2] = strike_lane(K[e][2], wl); break;
                case 3:  K[e][3] = strike_lane(K[e][3], wl); break;
                case 4:  K[e][4] = strike_lane(K[e][4], wl); break;
                default: K[e][5] = strike_lane(K[e][5], wl); break;
                }
            } else {
                int lagl = 0x7fffffff;
#pragma unroll
                for (int n = 5; n >= 0; n--) lagl = K[e][n] == best[e] ? L + 128 * (n >> 1) + (n & 1) : lagl;
                const int wlag = -wave_max_int(-lagl);          // the lowest lag that holds the maximum (a lag names one lane and slot)
                w[e] = wlag;
#pragma unroll
                for (int n = 0; n < 6; n++)
                    if (L + 128 * (n >> 1) + (n & 1) == wlag) K[e][n] = -1;
            }
        }
#pragma unroll
        for (int e = 0; e < 2; e++)
            if (need[e]) { cand[e][r] = w[e]; n_cand[e] = r + 1; }
    }
}

// Stage 2 for the pair: the float32 correlation values of the candidates, on ONE tile of v_mfma_f32_16x16x4_f32: row
// r' = 8 e + c of the tile is the lag block of candidate c of frame e (lag i = 8a + b reads the 144 floats from sample 8a
// on), the 16 columns are (b', real part) b' = 0..7 and (b', imaginary part) with the coefficients of WR_LTS_MFMA_B, and
// 36 instructions sum the floats in the order j = 0..8, s' = 0..3, kk = 0..3 of phi = 16 j + 4 kk + s' -- a k-ascending
// fmaf chain per output (tools/mfma_f32_probe.hip), the chain the rule writes down.  Of the 16 x 16 outputs the two of each
// candidate (column b and b + 8 of its row) are fetched into lane (e, c, part) = (lane >> 4 & 1, lane >> 1 & 7, lane & 1);
// `lag` = that lane's candidate.
__device__ __forceinline__ float lts_exact_pair(const float* ylds, int lane, const int (&cand0)[8], const int (&cand1)[8], int& lag)
{
    // as a row of the tile: lane & 15 = 8 e + c, lane >> 4 = kk
    const int re_ = (lane >> 3) & 1, rc = lane & 7, kk = lane >> 4;
    int rlag = re_ ? cand1[0] : cand0[0];
#pragma unroll
    for (int k = 1; k < 8; k++) rlag = (rc == k) ? (re_ ? cand1[k] : cand0[k]) : rlag;
    const float* arow = ylds + WR_PRE_FRAME_FLOATS * re_ + 16 * (rlag >> 3) + 4 * kk;
    const float4* __restrict__ bt = reinterpret_cast<const float4*>(WR_LTS_MFMA_B) + lane;
    wr_f4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const float4 bq = bt[64 * j];
        const float4 aq = *reinterpret_cast<const float4*>(arow + 16 * j);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.x, bq.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.y, bq.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.z, bq.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.w, bq.w, acc, 0, 0, 0);
    }
    // as the holder of one value: lane = 16 e + 2 c + part wants row r' = 8 e + c, column (lag & 7) + 8 part, i.e. result
    // register r' & 3 of lane 16 (r' >> 2) + column
    const int e = (lane >> 4) & 1, c = (lane >> 1) & 7, part = lane & 1;
    lag = e ? cand1[0] : cand0[0];
#pragma unroll
    for (int k = 1; k < 8; k++) lag = (c == k) ? (e ? cand1[k] : cand0[k]) : lag;
    const int rr = 8 * e + c;
    const int src = 4 * (16 * (rr >> 2) + (lag & 7) + 8 * part);
    // (copies first: __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index -- hipcc 7.2)
    const float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
    const int v0 = __builtin_amdgcn_ds_bpermute(src, __float_as_int(a0));
    const int v1 = __builtin_amdgcn_ds_bpermute(src, __float_as_int(a1));
    const int v2 = __builtin_amdgcn_ds_bpermute(src, __float_as_int(a2));
    const int v3 = __builtin_amdgcn_ds_bpermute(src, __float_as_int(a3));
    const int sel = (rr & 3) == 0 ? v0 : (rr & 3) == 1 ? v1 : (rr & 3) == 2 ? v2 : v3;
    return __builtin_bit_cast(float, sel);
}

#ifdef WR_DEBUG_EXACT
// the same values as one plain fmaf chain per lane (what the tile replaced): tools/lts_exact_probe.hip compares the two
__device__ __forceinline__ float lts_exact_pair_valu(const float* ylds, int lane, const int (&cand0)[8], const int (&cand1)[8], int& lag)
{
    const int e = (lane >> 4) & 1, c = (lane >> 1) & 7, part = lane & 1;
    lag = e ? cand1[0] : cand0[0];
#pragma unroll
    for (int k = 1; k < 8; k++) lag = (c == k) ? (e ? cand1[k] : cand0[k]) : lag;
    const int a = lag >> 3, col = (lag & 7) + 8 * part;
    const float4* __restrict__ arow = reinterpret_cast<const float4*>(ylds + WR_PRE_FRAME_FLOATS * e + 16 * a);
    const float4* __restrict__ bt = reinterpret_cast<const float4*>(WR_LTS_MFMA_B) + col;
    float acc = 0.0f;
#pragma unroll 1
    for (int j = 0; j < 9; j++) {            // not unrolled: 32 operand registers per step are enough to keep in flight
        float4 av[4], bv[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { av[kk] = arow[4 * j + kk]; bv[kk] = bt[64 * j + 16 * kk]; }
#pragma unroll
        for (int sp = 0; sp < 4; sp++) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const float as = sp == 0 ? av[kk].x : sp == 1 ? av[kk].y : sp == 2 ? av[kk].z : av[kk].w;
                const float bs = sp == 0 ? bv[kk].x : sp == 1 ? bv[kk].y : sp == 2 ? bv[kk].z : bv[kk].w;
                acc = fma_(as, bs, acc);
            }
        }
    }
    return acc;
}

#endif
// The (up to) four largest candidates of both frames of the pair by |corr|^2 = fma(im, im, re re) of their float32 values
// -- lowest lag first among equal values, a NaN is never a peak.  `ex`, `lag` as lts_exact_pair left them: rows 0 and 1 of
// the wave hold frame 0 and 1, lane 2c the real part of candidate c, lane 2c + 1 the imaginary part.  Per round one
// integer max over each row (the bit pattern of a magnitude >= +0 orders like a signed integer), then the lowest lag
// among the lanes that hold it.
__device__ __forceinline__ void lts_top4_pair(float ex, int lag, int lane, const int (&n_cand)[2], int (&top_off)[2][4], c32 (&top_val)[2][4])
{
    const int e = (lane >> 4) & 1, c = (lane >> 1) & 7, part = lane & 1;
    const float im = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ex), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]: the neighbour
    const float mag = fma_(im, im, ex * ex);
    const int nc = e ? n_cand[1] : n_cand[0];
    int key = (lane < 32 && part == 0 && c < nc && mag >= 0.0f) ? (int)__float_as_uint(mag) : -1;
    const int n_rounds = n_cand[0] > n_cand[1] ? n_cand[0] : n_cand[1];      // wave-uniform; usually 2
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r >= n_rounds) {                                    // no candidate left in either frame: what the round would find
#pragma unroll
            for (int f = 0; f < 2; f++) { top_off[f][r] = -1; top_val[f][r] = { 0.0f, 0.0f }; }
            continue;
        }
        const int m = row_max16(key);
        const int b0 = __builtin_amdgcn_readlane(m, 15), b1 = __builtin_amdgcn_readlane(m, 31);
        const int best = e ? b1 : b0;
        const bool hit = lane < 32 && key >= 0 && key == best;
        const int nl = row_max16(hit ? -lag : (int)0x80000000);                  // lowest lag = largest negated lag
        const int l0 = -__builtin_amdgcn_readlane(nl, 15), l1 = -__builtin_amdgcn_readlane(nl, 31);
        const bool win = hit && lag == (e ? l1 : l0);
        const uint64_t wb = __ballot(win);
#pragma unroll
        for (int f = 0; f < 2; f++) {
            const uint32_t wf = (uint32_t)(wb >> (16 * f)) & 0xffffu;
            top_off[f][r] = -1;
            top_val[f][r] = { 0.0f, 0.0f };
            if (wf) {                                                           // wave-uniform
                const int wl = 16 * f + (int)__builtin_ctz(wf);
                top_off[f][r] = f ? l1 : l0;
                top_val[f][r] = { bcast(ex, wl), bcast(ex, wl + 1) };
            }
        }
        if (win) key = -1;
    }
}

// sync_long's pair search over the (up to) four largest peaks of a frame: the frame start and the two peaks whose phase
// difference is the fine CFO (cfo_f = arg(conj(first) second) / diff, formed by the caller for both frames of a pair in one
// pass).  Returns the distance of the chosen pair (64, or the last 63 / 65 seen), 0 when no LTS pair was found.
__device__ __forceinline__ int lts_pair_search(const int (&top_off)[4], const c32 (&top_val)[4], int& fs, c32& first, c32& second)
{
    int found = 0;
    fs = WIFIRX_SYNC_LENGTH;
    first = { 0.0f, 0.0f };
    second = { 0.0f, 0.0f };
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int k = i + 1; k < 4; k++) {
            if (found == 64) continue;
            const int oi = top_off[i], ok = top_off[k];
            const int diff = oi > ok ? oi - ok : ok - oi;
            if ((diff == 64 || diff == 63 || diff == 65) && oi >= 0 && ok >= 0) {      // -1: fewer than four valid lags
                first = oi > ok ? top_val[k] : top_val[i];
                second = oi > ok ? top_val[i] : top_val[k];
                fs = oi < ok ? oi : ok;
                found = diff;
            }
        }
    }
    return found;
}

// What the preamble phase hands to the symbol phase.  One copy per lane, uniform inside a row: row f
// of the wave takes the values of frame f, so nothing wave-uniform has to be kept for four frames at once.
struct QuadSeed {
    const float2* x;      // stream / slot the frame lives in
    long     n_samp;
    long     t;           // trigger index, -1: no frame in this row
    long     L;           // usable copied samples
    float    cfo_c;
    float    cfo_f;
    int      fs;
    uint32_t flags;       // DETECTED [| SYNC] [| TRUNCATED]
    long     out;         // index of the frame's record / output slices, -1: none
};

__device__ __forceinline__ QuadSeed quad_seed_none()
{
    QuadSeed q;
    q.x = nullptr; q.n_samp = 0; q.t = -1; q.L = 0; q.cfo_c = 0.0f; q.cfo_f = 0.0f; q.fs = 0; q.flags = 0; q.out = -1;
    return q;
}

// The LTS search of a pair whose frames were derotated into `lds`, and the seeds of rows 2 p + e of the wave.
__device__ __forceinline__ void preamble_pair_finish(const PreFrame& f0, const PreFrame& f1, int p, const float* lds, int lane, QuadSeed& seed)
{
    const PreFrame pf[2] = { f0, f1 };
    const bool run[2] = { pf[0].out >= 0 && pf[0].t >= 0 && pf[0].search, pf[1].out >= 0 && pf[1].t >= 0 && pf[1].search };   // wave-uniform
    int cand[2][8], n_cand[2] = { 0, 0 };
    int top_off[2][4] = { { -1, -1, -1, -1 }, { -1, -1, -1, -1 } };
    c32 top_val[2][4];
    float ex = 0.0f;
    if (run[0] || run[1]) {
        wr_i4 acc[5];
        lts_corr_pair_q8(lds, lane, acc);
        lts_candidates2(acc, run, lane, cand, n_cand);
        int lag;
        ex = lts_exact_pair(lds, lane, cand[0], cand[1], lag);
        lts_top4_pair(ex, lag, lane, n_cand, top_off, top_val);
    }
    // the pair search of both frames, then ONE pass of the arctangent for the two fine CFOs (even lanes: frame 0, odd: 1)
    int fs2[2] = { 0, 0 }, found2[2] = { 0, 0 };
    c32 fi[2] = { { 0.0f, 0.0f }, { 0.0f, 0.0f } }, se[2] = { { 0.0f, 0.0f }, { 0.0f, 0.0f } };
#pragma unroll
    for (int e = 0; e < 2; e++)
        if (pf[e].out >= 0 && pf[e].t >= 0 && pf[e].search) found2[e] = lts_pair_search(top_off[e], top_val[e], fs2[e], fi[e], se[e]);
    float cfo2;
    {
        const bool odd = lane & 1;
        const c32 a = odd ? fi[1] : fi[0], b = odd ? se[1] : se[0];
        const int dv = odd ? found2[1] : found2[0];
        const float pr = fma_(a.im, b.im, a.re * b.re);
        const float pi = fma_(a.im, b.re, -(a.re * b.im));
        cfo2 = sp_atan2(pi, pr) / (float)(dv ? dv : 64);
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
        if (pf[e].out < 0) continue;
        int fs = 0;
        float cfo_f = 0.0f;
        uint32_t flags = 0;
        if (pf[e].t >= 0) {
            flags = WIFIRX_F_DETECTED | WIFIRX_F_TRUNCATED;
            if (pf[e].search) {
                const bool ok = found2[e] != 0;
                flags = ok ? (WIFIRX_F_DETECTED | WIFIRX_F_SYNC) : WIFIRX_F_DETECTED;
                if (ok) { fs = fs2[e]; cfo_f = bcast(cfo2, e); }
            }
        }
        if ((lane >> 4) == 2 * p + e) {
            seed.x = pf[e].x; seed.n_samp = pf[e].n_samp; seed.t = pf[e].t; seed.L = pf[e].L; seed.cfo_c = pf[e].cfo_c;
            seed.cfo_f = cfo_f; seed.fs = fs; seed.flags = flags; seed.out = pf[e].out;
        }
    }
}

// in-register 4-point DIF butterfly (spec section 4.4)
__device__ __forceinline__ void bfly4_reg(c32& a, c32& b, c32& c, c32& d)
{
    c32 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    b = { t1.re + t3.im, t1.im - t3.re };     // t1 - j t3
    c = csub(t0, t2);
    d = { t1.re - t3.im, t1.im + t3.re };     // t1 + j t3
}

template <int LANE>
__device__ __forceinline__ float row_bcast(float v)
{
    // every lane is written (row_newbcast, full masks): mov_dpp leaves the previous value of the destination undefined
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x150 + LANE, 0xf, 0xf, false));
}
template <int LANE>
__device__ __forceinline__ c32 row_bcast(c32 v) { return { row_bcast<LANE>(v.re), row_bcast<LANE>(v.im) }; }

// sum over the 16 lanes of a row in the order of the spec's xor tree (steps 1,2,4,8)
__device__ __forceinline__ float row_xor_sum16(float v)
{
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));  // xor 1
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));  // xor 2
    v = v + __shfl_xor(v, 4, 64);
    v = v + __shfl_xor(v, 8, 64);
    return v;
}

// a6 + a7 stores of one lane's four bins for constellation NB (compile-time: no per-lane branching on it).
// ok = row active; the pointers are the row's output slices.
// CSI: every LLR is multiplied by w[j] = |H|^2 of its sub-carrier (spec rule 12).
// idx / car / llr: the rows of the wave's FIRST record (wave-uniform pointers: the stores take them from scalar
// registers); row_o / row_l: this row's distance from there in decisions / in LLR values (32 bits per lane).
// PLAIN: the caller has established (wave-uniform) that decisions and LLRs are wanted by every row with a symbol and the
// equalised points by none -- the usual output set; the flags are then compile-time and the four bins leave without a
// scalar branch per output and bin.
template <int NB, bool CSI, bool PLAIN = false>
__device__ __forceinline__ void store_bins(const c32 (&Y)[4], const int (&carrier)[4], bool ok, int q,
                                           uint8_t* __restrict__ idx, float2* __restrict__ car,
                                           float* __restrict__ llr, bool has_idx_, bool has_car_, bool want_llr_,
                                           const float (&w)[4], uint32_t row_o, uint32_t row_l)
{
    const bool has_idx = PLAIN ? true : has_idx_, has_car = PLAIN ? false : has_car_, want_llr = PLAIN ? true : want_llr_;
#pragma unroll
    for (int j = 0; j < 4; j++) {
#if WR_CONST_DATA_MASK
        // which lanes hold a data sub-carrier in register j is a constant of the lane <-> bin map (bins r + 16 j): the
        // execution mask comes from a scalar constant, not from a comparison of the carrier number
        const uint64_t row_bits = j == 0 ? 0xF7C0ull : j == 1 ? 0xFDFFull : j == 2 ? 0xFF7Eull : 0x07DFull;      // (the loop is unrolled)
        if (!(ok && __builtin_amdgcn_inverse_ballot_w64(row_bits * 0x0001000100010001ull))) continue;
#else
        if (!(ok && carrier[j] >= 0)) continue;
#endif
        const uint32_t oq = (uint32_t)(q * 48 + carrier[j]);
        const uint32_t o = row_o + oq;
        if (has_idx) idx[o] = decide(Y[j], NB);
        // byte distances in 32 bits: base pointer from scalar registers + one offset register per store
        if (has_car) *reinterpret_cast<float2*>(reinterpret_cast<char*>(car) + (uint32_t)(o * 8u)) = make_float2(Y[j].re, Y[j].im);
        if (want_llr) {
            const float are = __builtin_fabsf(Y[j].re), aim = __builtin_fabsf(Y[j].im);
            float* lp = reinterpret_cast<float*>(reinterpret_cast<char*>(llr) + (uint32_t)((row_l + oq * NB) * 4u));
            const float wj = CSI ? w[j] : 1.0f;
#define WR_WT(v) (CSI ? (v) * wj : (v))
            if (NB == 1) {
                lp[0] = WR_WT(Y[j].re);
            } else if (NB == 2) {
                *reinterpret_cast<float2*>(lp) = make_float2(WR_WT(Y[j].re), WR_WT(Y[j].im));
            } else if (NB == 4) {
                *reinterpret_cast<float4*>(lp) = make_float4(WR_WT(Y[j].re), WR_WT(WR_T16_2 - are), WR_WT(Y[j].im), WR_WT(WR_T16_2 - aim));
            } else {
                float2* l2 = reinterpret_cast<float2*>(lp);
                l2[0] = make_float2(WR_WT(Y[j].re), WR_WT(WR_T64_4 - are));
                l2[1] = make_float2(WR_WT(WR_T64_2 - __builtin_fabsf(are - WR_T64_4)), WR_WT(Y[j].im));
                l2[2] = make_float2(WR_WT(WR_T64_4 - aim), WR_WT(WR_T64_2 - __builtin_fabsf(aim - WR_T64_4)));
            }
#undef WR_WT
        }
    }
}

// a 16-byte piece / a dword of an output row (WR_NT_STORES: as a streaming store -- the rows are written once and never read here)
typedef float wr_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_piece(char* p, float4 v)
{
#if WR_NT_STORES
    __builtin_nontemporal_store(wr_f4v{ v.x, v.y, v.z, v.w }, reinterpret_cast<wr_f4v*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}
// a dword of a row's decisions: ALWAYS a plain store.  A row of decisions is 48 bytes -- never a whole 128-byte line --, and a
// streaming store of part of a line leaves the L2 before the rest of the line has arrived: the memory probe's stores alone take
// 6.5 ms with streaming decision dwords against 4.1 ms with plain ones on a box whose store path is slow, the whole pattern
// 12.0 against 9.75 ms (bench.py roofline.box, profiles/r04_box_nt_decisions.json) -- the "slow box" of round 3's driver run.
__device__ __forceinline__ void store_word(uint8_t* p, uint32_t v)
{
#if WR_NT_STORES > 1
    __builtin_nontemporal_store(v, reinterpret_cast<uint32_t*>(p));
#else
    *reinterpret_cast<uint32_t*>(p) = v;
#endif
}

// The same stores as whole 16-byte pieces: a row's LLRs of one symbol are 192 NB contiguous bytes (192 / 384 / 768 / 1152)
// and its decisions 48, but lane r holds bins r + 16 j, so store_bins() writes them as four pieces per row that start and end
// inside 128-byte lines -- for 64-QAM as three 8-byte stores per bin with the lanes 24 bytes apart -- and the decisions as
// single bytes.  Here every lane drops its values into the scratch LDS area (free at this point of the symbol) in carrier
// order and the row's lanes write the row back out as 16-byte pieces (decisions: one dword for lanes 0..11): 2 / 3 / 4 / 6
// store instructions per symbol instead of 8 / 8 / 8 / 16, every line written whole.  The memory probes run 11 % (QPSK,
// tools/mem_floor.hip) and 15 % (64-QAM, tools/mem_floor64.hip: 5.71 -> 4.83 ms on the config-3 geometry) faster in this
// shape; the kernel follows since the data loop was slimmed down: -3 % on config 2, -13 % on the config-3 geometry, -5 % on
// long 16-QAM frames.  Same values, same addresses.
// Requires (caller, wave-uniform): decisions and LLRs wanted by every active row, no weights, idx 4-byte and llr 16-byte aligned.
// OFF32 (LMS / COMB / STA instances): every piece is addressed by the row base from scalar registers + a 32-bit offset of its own.
// Written as a pointer sum (lp + 256 k), the compiler keeps the 64-bit address of a piece in a register pair across the loop; in
// those instances the pair was spilled, and its reload inside the loop is a vector memory operation the counted wait of the
// prefetch cannot skip (tests/test_isa_prefetch_wait.py; LMS -6 %).  The LS instances have the registers and lose 0.8 % (plane
// output) to the extra additions: they keep the pointer sum (profiles/r05_ab_spills_in_prefetch_loops.txt).
template <int NB, bool OFF32 = false>
__device__ __forceinline__ void store_bins_lines(const c32 (&Y)[4], const int (&carrier)[4], bool ok, int q,
                                                 uint8_t* __restrict__ idx, float* __restrict__ llr,
                                                 uint32_t row_o, uint32_t row_l, float* stage, int row, int r)
{
    static_assert(NB == 1 || NB == 2 || NB == 4 || NB == 6, "constellation");
    // 16- / 64-QAM rows (768 / 1152 bytes of LLRs per symbol: 48 / 72 pieces) need the long scratch area (WR_QLDS_SCRATCH_EQ)
    constexpr int ROWB = 192 * NB, IDX0 = NB <= 2 ? 1536 : 4 * ROWB;
    char* srow = reinterpret_cast<char*>(stage) + row * ROWB;
    uint8_t* irow = reinterpret_cast<uint8_t*>(stage) + IDX0 + row * 48;
    __builtin_amdgcn_wave_barrier();
#if WR_FLAT_STAGING
    if (NB <= 4) {
        // one straight piece of code for the four bins of a lane: EVERY lane writes, the lanes whose bin carries no data (pilots,
        // DC, guards: a constant of the lane <-> bin map) into a dump area behind the rows.  The four per-bin regions under an
        // execution mask of their own cost three scalar instructions and a branch each, and kept every compare -> select pair
        // of the decisions back to back (two idle issue slots per pair).  (64-QAM: the scratch area has no room for a dump.)
        constexpr int ESZ = 4 * NB, DUMP_L = IDX0 + 192, DUMP_I = DUMP_L + 64 * ESZ;
        const int lane = 16 * row + r;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool data = carrier[j] >= 0;
            char* lw = data ? srow + ESZ * carrier[j] : reinterpret_cast<char*>(stage) + DUMP_L + ESZ * lane;
            uint8_t* iw = data ? irow + carrier[j] : reinterpret_cast<uint8_t*>(stage) + DUMP_I + lane;
            if (NB == 1)      *reinterpret_cast<float*>(lw) = Y[j].re;
            else if (NB == 2) *reinterpret_cast<float2*>(lw) = make_float2(Y[j].re, Y[j].im);
            else              *reinterpret_cast<float4*>(lw) = make_float4(Y[j].re, WR_T16_2 - __builtin_fabsf(Y[j].re), Y[j].im, WR_T16_2 - __builtin_fabsf(Y[j].im));
#ifndef WR_DBG_NO_IDX_STAGING      // (measurement only: profiles/r05_lds_bank_conflicts.txt)
            *iw = decide(Y[j], NB);
#endif
        }
    } else
#endif
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t row_bits = j == 0 ? 0xF7C0ull : j == 1 ? 0xFDFFull : j == 2 ? 0xFF7Eull : 0x07DFull;      // (the loop is unrolled)
        if (!__builtin_amdgcn_inverse_ballot_w64(row_bits * 0x0001000100010001ull)) continue;
        const float are = __builtin_fabsf(Y[j].re), aim = __builtin_fabsf(Y[j].im);
        if (NB == 1)      *reinterpret_cast<float*>(srow + 4 * carrier[j]) = Y[j].re;
        else if (NB == 2) *reinterpret_cast<float2*>(srow + 8 * carrier[j]) = make_float2(Y[j].re, Y[j].im);
        else if (NB == 4) *reinterpret_cast<float4*>(srow + 16 * carrier[j]) = make_float4(Y[j].re, WR_T16_2 - are, Y[j].im, WR_T16_2 - aim);
        else if (OFF32) {
            // ONE lane constant per bin, ci = carrier + 48 row: the decision goes to byte IDX0 + ci and the LLRs to byte 24 ci
            // (ROWB = 24 x 48) -- the product formed HERE, by an instruction the compiler cannot hoist.  As two lane constants
            // per bin (eight registers across the 64-QAM loop) one of them was spilled in the LMS / STA instances, and its reload
            // inside the loop is a vector memory operation the counted wait of the prefetch cannot skip.
            const uint32_t ci = (uint32_t)(carrier[j] + 48 * row);
            uint32_t l24;
            asm volatile("v_mul_u32_u24 %0, 24, %1" : "=v"(l24) : "v"(ci));
            float2* l2 = reinterpret_cast<float2*>(reinterpret_cast<char*>(stage) + l24);
            l2[0] = make_float2(Y[j].re, WR_T64_4 - are);
            l2[1] = make_float2(WR_T64_2 - __builtin_fabsf(are - WR_T64_4), Y[j].im);
            l2[2] = make_float2(WR_T64_4 - aim, WR_T64_2 - __builtin_fabsf(aim - WR_T64_4));
            reinterpret_cast<uint8_t*>(stage)[IDX0 + ci] = decide(Y[j], NB);
            continue;
        } else {
            float2* l2 = reinterpret_cast<float2*>(srow + 24 * carrier[j]);
            l2[0] = make_float2(Y[j].re, WR_T64_4 - are);
            l2[1] = make_float2(WR_T64_2 - __builtin_fabsf(are - WR_T64_4), Y[j].im);
            l2[2] = make_float2(WR_T64_4 - aim, WR_T64_2 - __builtin_fabsf(aim - WR_T64_4));
        }
        irow[carrier[j]] = decide(Y[j], NB);
    }
    __builtin_amdgcn_wave_barrier();
    // piece r + 16 k of the row, k = 0 .. (12 NB - 1) / 16 (the last k for the lanes that still have one)
    constexpr int NK = (12 * NB + 15) / 16, TAIL = 12 * NB - 16 * (NK - 1);       // BPSK 1 / 12, QPSK 2 / 8, 16-QAM 3 / 16, 64-QAM 5 / 8
    // (separate variables, not an array: an array indexed in a loop under a lane condition stays in scratch memory)
#define WR_PIECE(k) (*reinterpret_cast<const float4*>(srow + 256 * (k) + 16 * (((k) == NK - 1 && TAIL == 8) ? (r & 7) : r)))     // (BPSK: lanes 12..15 read into the next row's area, and store nothing)
    const float4 p0 = WR_PIECE(0);
    float4 p1 = p0, p2 = p0, p3 = p0, p4 = p0;
    if (NK > 1) p1 = WR_PIECE(1);
    if (NK > 2) p2 = WR_PIECE(2);
    if (NK > 3) p3 = WR_PIECE(3);
    if (NK > 4) p4 = WR_PIECE(4);
#undef WR_PIECE
    const uint32_t d = reinterpret_cast<const uint32_t*>(irow)[r < 12 ? r : 0];
    __builtin_amdgcn_wave_barrier();
    if (ok) {
        char* const lb = reinterpret_cast<char*>(llr);
        const uint32_t lo = (row_l + (uint32_t)(q * 48 * NB)) * 4u + 16u * r;
        char* const lp = reinterpret_cast<char*>(llr) + (uint32_t)((row_l + (uint32_t)(q * 48 * NB)) * 4u + 16u * r);
#define WR_PIECE_AT(K) (OFF32 ? lb + (uint32_t)(lo + 256u * (K)) : lp + 256 * (K))
        if (NK > 1 || r < TAIL)            store_piece(lp, p0);
        if (NK > 2 || (NK == 2 && r < TAIL)) store_piece(WR_PIECE_AT(1), p1);
        if (NK > 3 || (NK == 3 && r < TAIL)) store_piece(WR_PIECE_AT(2), p2);
        if (NK > 4 || (NK == 4 && r < TAIL)) store_piece(WR_PIECE_AT(3), p3);
        if (NK == 5 && r < TAIL)             store_piece(WR_PIECE_AT(4), p4);
#undef WR_PIECE_AT
        if (r < 12) store_word(idx + (row_o + (uint32_t)(q * 48) + 4u * r), d);
    }
}

// The rows of a symbol as whole 16-byte pieces for ANY output set (round 4): decisions, LLRs -- with the channel-state weight
// when asked for (spec rule 12) --, and the equalised points of the `carrier` port (IRS_AP.py:293: frame_equalizer.symbols; a row
// of 48 points is 384 bytes = three whole lines).  Which outputs are wanted is wave-uniform and constant for the launch (scalar
// branches); the constellation is compile time.  Same values, same addresses as store_bins().  For QPSK rows without weights the
// LLR row IS the carrier row (re, im per carrier): the pieces are stored twice, not staged twice.
// Requires (caller, wave-uniform): idx 4-byte, llr and car 16-byte aligned rows; has_llr = every active row wants LLRs.
template <int NB, bool OFF32 = false>      // OFF32: as store_bins_lines
__device__ __forceinline__ void store_rows_x(const c32 (&Y)[4], const int (&carrier)[4], bool ok, int q,
                                             uint8_t* __restrict__ idx, float* __restrict__ llr, float2* __restrict__ car,
                                             bool has_idx, bool has_llr, bool has_car, bool csi, const float* Wl,
                                             uint32_t row_o, uint32_t row_l, float* stage, int row, int r)
{
    static_assert(NB == 1 || NB == 2 || NB == 4 || NB == 6, "constellation");
    constexpr int ROWB = 192 * NB, IDX0 = NB <= 2 ? 1536 : 4 * ROWB;
    constexpr int NK = (12 * NB + 15) / 16, TAIL = 12 * NB - 16 * (NK - 1);
    char* srow = reinterpret_cast<char*>(stage) + row * ROWB;
    uint8_t* irow = reinterpret_cast<uint8_t*>(stage) + IDX0 + row * 48;
    float4 p0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), p1 = p0;
    if (has_idx || has_llr) {
        float w[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
        if (csi) {
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = Wl[64 * j];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t row_bits = j == 0 ? 0xF7C0ull : j == 1 ? 0xFDFFull : j == 2 ? 0xFF7Eull : 0x07DFull;      // (the loop is unrolled)
            if (!__builtin_amdgcn_inverse_ballot_w64(row_bits * 0x0001000100010001ull)) continue;
            const float are = __builtin_fabsf(Y[j].re), aim = __builtin_fabsf(Y[j].im);
            // (x * 1.0f is x for every float: the unweighted values are those of store_bins_lines)
#define WR_WT(v) (csi ? (v) * w[j] : (v))
            if (NB == 1)      *reinterpret_cast<float*>(srow + 4 * carrier[j]) = WR_WT(Y[j].re);
            else if (NB == 2) *reinterpret_cast<float2*>(srow + 8 * carrier[j]) = make_float2(WR_WT(Y[j].re), WR_WT(Y[j].im));
            else if (NB == 4) *reinterpret_cast<float4*>(srow + 16 * carrier[j]) = make_float4(WR_WT(Y[j].re), WR_WT(WR_T16_2 - are), WR_WT(Y[j].im), WR_WT(WR_T16_2 - aim));
            else {
                float2* l2 = reinterpret_cast<float2*>(srow + 24 * carrier[j]);
                l2[0] = make_float2(WR_WT(Y[j].re), WR_WT(WR_T64_4 - are));
                l2[1] = make_float2(WR_WT(WR_T64_2 - __builtin_fabsf(are - WR_T64_4)), WR_WT(Y[j].im));
                l2[2] = make_float2(WR_WT(WR_T64_4 - aim), WR_WT(WR_T64_2 - __builtin_fabsf(aim - WR_T64_4)));
            }
#undef WR_WT
            irow[carrier[j]] = decide(Y[j], NB);
        }
        __builtin_amdgcn_wave_barrier();
#define WR_PIECE(k) (*reinterpret_cast<const float4*>(srow + 256 * (k) + 16 * (((k) == NK - 1 && TAIL == 8) ? (r & 7) : r)))
        p0 = WR_PIECE(0);
        float4 p2 = p0, p3 = p0, p4 = p0;
        p1 = p0;
        if (NK > 1) p1 = WR_PIECE(1);
        if (NK > 2) p2 = WR_PIECE(2);
        if (NK > 3) p3 = WR_PIECE(3);
        if (NK > 4) p4 = WR_PIECE(4);
#undef WR_PIECE
        const uint32_t d = reinterpret_cast<const uint32_t*>(irow)[r < 12 ? r : 0];
        __builtin_amdgcn_wave_barrier();
        if (ok) {
            if (has_llr) {
                char* const lb = reinterpret_cast<char*>(llr);
                const uint32_t lo = (row_l + (uint32_t)(q * 48 * NB)) * 4u + 16u * r;
                char* const lp = reinterpret_cast<char*>(llr) + (uint32_t)((row_l + (uint32_t)(q * 48 * NB)) * 4u + 16u * r);
#define WR_PIECE_AT(K) (OFF32 ? lb + (uint32_t)(lo + 256u * (K)) : lp + 256 * (K))
                if (NK > 1 || r < TAIL)            store_piece(lp, p0);
                if (NK > 2 || (NK == 2 && r < TAIL)) store_piece(WR_PIECE_AT(1), p1);
                if (NK > 3 || (NK == 3 && r < TAIL)) store_piece(WR_PIECE_AT(2), p2);
                if (NK > 4 || (NK == 4 && r < TAIL)) store_piece(WR_PIECE_AT(3), p3);
                if (NK == 5 && r < TAIL)             store_piece(WR_PIECE_AT(4), p4);
#undef WR_PIECE_AT
            }
            if (has_idx && r < 12) store_word(idx + (row_o + (uint32_t)(q * 48) + 4u * r), d);
        }
    }
    if (has_car) {
        if (!(NB == 2 && has_llr && !csi)) {       // (wave-uniform) the points staged on their own: 4 rows x 384 bytes
            char* crow = reinterpret_cast<char*>(stage) + row * 384;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t row_bits = j == 0 ? 0xF7C0ull : j == 1 ? 0xFDFFull : j == 2 ? 0xFF7Eull : 0x07DFull;
                if (!__builtin_amdgcn_inverse_ballot_w64(row_bits * 0x0001000100010001ull)) continue;
                *reinterpret_cast<float2*>(crow + 8 * carrier[j]) = make_float2(Y[j].re, Y[j].im);
            }
            __builtin_amdgcn_wave_barrier();
            p0 = *reinterpret_cast<const float4*>(crow + 16 * r);
            p1 = *reinterpret_cast<const float4*>(crow + 256 + 16 * (r & 7));
            __builtin_amdgcn_wave_barrier();
        }
        if (ok) {
            if (OFF32) {
                char* const cb = reinterpret_cast<char*>(car);
                const uint32_t co = (row_o + (uint32_t)(q * 48)) * 8u + 16u * r;
                store_piece(cb + co, p0);
                if (r < 8) store_piece(cb + (uint32_t)(co + 256u), p1);
            } else {
                char* cp = reinterpret_cast<char*>(car) + (uint32_t)((row_o + (uint32_t)(q * 48)) * 8u + 16u * r);
                store_piece(cp, p0);
                if (r < 8) store_piece(cp + 256, p1);
            }
        }
    }
}

#define WR_HB_DATA_LO 0xFDFFF7C0u     // bins 6..31 without the pilots 11, 25
#define WR_HB_DATA_HI 0x07DFFF7Eu     // bins 33..58 without the pilots 39, 53 (bit k = bin 32 + k)

// decision bit B of a point for constellation NB -- the comparisons of decide(), one at a time
template <int NB, int B>
__device__ __forceinline__ bool decide_bit(c32 y)
{
    const float are = __builtin_fabsf(y.re), aim = __builtin_fabsf(y.im);
    if (NB == 1) return y.re > 0.0f;
    if (NB == 2) return B == 0 ? y.re > 0.0f : y.im > 0.0f;
    if (NB == 4) return B == 0 ? y.re > 0.0f : B == 1 ? are < WR_T16_2 : B == 2 ? y.im > 0.0f : aim < WR_T16_2;
    return B == 0 ? y.re > 0.0f : B == 1 ? are < WR_T64_4 : B == 2 ? (are < WR_T64_6) && (are > WR_T64_2)
         : B == 3 ? y.im > 0.0f : B == 4 ? aim < WR_T64_4 : (aim < WR_T64_6) && (aim > WR_T64_2);
}

// the two words of bit plane B (bins 0..31, 32..63) of the four frames of the wave, into lanes 0, 16, 32, 48 of w0 / w1.
// A comparison over the wave IS the plane: its result mask holds, for the row of frame f, bins 16 j .. 16 j + 15 in bits
// 16 f .. 16 f + 15 when every lane compares its bin r + 16 j; the scalar unit cuts the four masks (j = 0..3) into the
// frames' words, one write per frame puts them into the lane that stores.
template <int NB, int B>
__device__ __forceinline__ void plane_words(const c32 (&Y)[4], uint32_t& w0, uint32_t& w1)
{
    const uint64_t m0 = __ballot(decide_bit<NB, B>(Y[0])), m1 = __ballot(decide_bit<NB, B>(Y[1]));
    const uint64_t m2 = __ballot(decide_bit<NB, B>(Y[2])), m3 = __ballot(decide_bit<NB, B>(Y[3]));
    // frames 0, 1 sit in the low words of the masks, 2, 3 in the high words; s_pack_{ll,hh} puts a frame's two halves together
    const uint32_t a0 = (uint32_t)m0, a1 = (uint32_t)m1, a2 = (uint32_t)m2, a3 = (uint32_t)m3;
    const uint32_t b0 = (uint32_t)(m0 >> 32), b1 = (uint32_t)(m1 >> 32), b2 = (uint32_t)(m2 >> 32), b3 = (uint32_t)(m3 >> 32);
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const uint32_t p0 = f < 2 ? a0 : b0, p1 = f < 2 ? a1 : b1, p2 = f < 2 ? a2 : b2, p3 = f < 2 ? a3 : b3;
        uint32_t lo, hi;
        if (f & 1) {
            asm("s_pack_hh_b32_b16 %0, %1, %2" : "=s"(lo) : "s"(p0), "s"(p1));
            asm("s_pack_hh_b32_b16 %0, %1, %2" : "=s"(hi) : "s"(p2), "s"(p3));
        } else {
            asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(lo) : "s"(p0), "s"(p1));
            asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(hi) : "s"(p2), "s"(p3));
        }
        asm("v_writelane_b32 %0, %1, %2" : "+v"(w0) : "s"(lo & WR_HB_DATA_LO), "n"(16 * f));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(w1) : "s"(hi & WR_HB_DATA_HI), "n"(16 * f));
    }
}

// a6 as bit planes (wifirx_out.hbits): word 2 b + h of the symbol = bit b of the decisions of bins 32 h .. 32 h + 31.
// Lane 0 of a row stores the row's words; hb = the frame's first word.
template <int NB>
__device__ __forceinline__ void store_hbits(const c32 (&Y)[4], bool ok, int q, uint32_t* __restrict__ hb, int r)
{
    uint32_t* dst = hb + (unsigned)(q * 2 * NB);
    if (NB == 1) {
        uint32_t w0 = 0, w1 = 0;
        plane_words<1, 0>(Y, w0, w1);
        if (ok && r == 0) *reinterpret_cast<uint2*>(dst) = make_uint2(w0, w1);
    } else {
        // two planes (four words) at a time: what is live stays small
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        plane_words<NB, 0>(Y, w.x, w.y);
        plane_words<NB, 1>(Y, w.z, w.w);
        if (ok && r == 0) *reinterpret_cast<uint4*>(dst) = w;
        if (NB >= 4) {
            plane_words<NB, 2>(Y, w.x, w.y);
            plane_words<NB, (NB >= 4 ? 3 : 0)>(Y, w.z, w.w);
            if (ok && r == 0) *reinterpret_cast<uint4*>(dst + 4) = w;
        }
        if (NB == 6) {
            plane_words<NB, (NB == 6 ? 4 : 0)>(Y, w.x, w.y);
            plane_words<NB, (NB == 6 ? 5 : 0)>(Y, w.z, w.w);
            if (ok && r == 0) *reinterpret_cast<uint4*>(dst + 8) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// a3 copy + a4 + a5 + a6 + a7 for the four frames of a wave.
// the constellation point of an index (levels formed in float32 as the upstream constellations do)
__device__ __forceinline__ c32 point_of(unsigned idx, int n_bpsc)
{
    c32 p;
    if (n_bpsc == 1) { p.re = (idx & 1) ? 1.0f : -1.0f; p.im = 0.0f; return p; }
    if (n_bpsc == 2) {
        p.re = (idx & 1) ? WR_LEVEL_QPSK : -WR_LEVEL_QPSK;
        p.im = (idx & 2) ? WR_LEVEL_QPSK : -WR_LEVEL_QPSK;
        return p;
    }
    if (n_bpsc == 4) {
        const float l = WR_T16_2 * 0.5f;
        float ar = (idx & 2) ? l : 3.0f * l, ai = (idx & 8) ? l : 3.0f * l;
        p.re = (idx & 1) ? ar : -ar;
        p.im = (idx & 4) ? ai : -ai;
        return p;
    }
    const float l = WR_T64_2 * 0.5f;
    unsigned r = (idx >> 1) & 3, q = (idx >> 4) & 3;
    unsigned kr = ((r & 1) << 1) | (r >> 1), kq = ((q & 1) << 1) | (q >> 1);     // 0,1,2,3 -> 7a,5a,1a,3a
    float ar = (kr == 0 ? 7.0f : kr == 1 ? 5.0f : kr == 2 ? 1.0f : 3.0f) * l;
    float ai = (kq == 0 ? 7.0f : kq == 1 ? 5.0f : kq == 2 ? 1.0f : 3.0f) * l;
    p.re = (idx & 1) ? ar : -ar;
    p.im = (idx & 8) ? ai : -ai;
    return p;
}

// EQ = WIFIRX_EQ_LS:   channel estimate from the LTS, held for the frame; multiplier form.
// EQ = WIFIRX_EQ_LMS:  decision-directed (ieee802_11.LMS): Y = X/H, then H = H/2 + (X/point)/2 on every data bin.
// EQ = WIFIRX_EQ_COMB: the four pilots of every symbol, interpolated over the band, smoothed over time.
// EQ = WIFIRX_EQ_STA:  spectral-temporal averaging of the per-bin estimates X/point.
// HB: the decisions also leave as bit planes (DemodOut.hbits); a template parameter so that the kernels without planes
// keep their register allocation.
// XK: the kernel instance for every output set but decisions + LLRs (wr_kernels_x.hip): its constellation loops end in
// store_rows_x(); the instance with XK = false keeps the loops of the usual set and nothing else (wr_demod.h).
template <int EQ, bool HB, bool XK>
__device__ __forceinline__ void frames_quad(const QuadSeed& seed, const DemodParams& prm, float* qlds, int lane,
                                            const DemodOut& dout)
{
    wifirx_frame* __restrict__ frames = dout.frames;
    uint8_t* __restrict__ idx_all = dout.idx;
    float* __restrict__ llr_all = dout.llr;
    float2* __restrict__ car_all = dout.carrier;
    float2* __restrict__ csi_all = dout.csi;
    float4* __restrict__ stat_all = dout.sym_stats;
    uint32_t* __restrict__ hb_all = dout.hbits;
    constexpr bool LMS = EQ == WIFIRX_EQ_LMS, COMB = EQ == WIFIRX_EQ_COMB, STA = EQ == WIFIRX_EQ_STA;
    constexpr int QS = WR_QLDS_SCRATCH_EQ(EQ);      // where the per-wave state behind the scratch area begins
    constexpr bool DIV = EQ != WIFIRX_EQ_LS;         // Y = X / H by division (LS multiplies by G = conj(H)/|H|^2)
    const int row = lane >> 4, r = lane & 15;
    // ---- row-uniform frame state, one copy per lane ----
    // Samples are addressed as xb[m], m = index into the copied stream y (xb = x + trigger - 16); m is valid
    // for m_lo <= m < m_hi (the part of y that lies inside the slot / stream buffer).  Rows without a frame
    // get an empty range on a harmless pointer.  32-bit indices: L <= 43200 + 320.
    const bool  has = seed.out >= 0 && seed.x != nullptr && seed.t >= 0;
    const long  t16 = seed.t - 16;
    // rows without a frame point at the samples of the wave's first frame: whatever they load is finite and never used
    const uint64_t has_m = __ballot(has);
    const int has_l = has_m ? (int)__builtin_ctzll(has_m) : 0;
    const uint64_t xv = has ? reinterpret_cast<uint64_t>(seed.x + t16) : 0ull;
    const uint64_t xf = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(xv >> 32), has_l) << 32) |
                        (uint32_t)__builtin_amdgcn_readlane((int)xv, has_l);
    const float2* xb = has ? seed.x + t16 : reinterpret_cast<const float2*>(xf);
    const int   m_lo = has ? (int)(t16 < 0 ? -t16 : 0) : 0;
    const long  m_hi_l = seed.n_samp - t16;
    const int   m_hi = has ? (int)(m_hi_l > 0x7fffff00l ? 0x7fffff00l : m_hi_l) : 0;
    const int   L = (int)seed.L;
    // wave-uniform: no row starts before its stream does, and the first 64 copied samples of every frame exist (what a
    // row without a symbol reads in the fast path below)
    const bool  lo_zero = __all(m_lo == 0 && L <= m_hi && (!has || m_hi >= 64));
    const int   out = (int)seed.out;
    const int   trig = (int)seed.t;
    const float cfo_c = seed.cfo_c, cfo_f = seed.cfo_f;
    const int   fs = seed.fs;
    uint32_t flags = seed.flags;
    bool alive = (out >= 0) && (flags & WIFIRX_F_SYNC);
    const double bw = prm.bandwidth, fc = prm.frequency;
    const double two_pi = 2 * 3.14159265358979323846;
    const double tag = (double)cfo_c - (double)cfo_f;
    // spec rule 9: the control chain of the sampling-offset compensation in float32 (formed once in double, rounded)
    const float eps0 = (float)(tag * bw / (two_pi * fc));
    const float er_scale = (float)(bw / (two_pi * fc * 80));
    float d_er = 0.0f;
    // total derotation as an exact integer phase: Qp = 2^-62 quarter turns per sample; sample m has phase Qp*m
    // mod 2^64 (a 64 x 32-bit product where a phasor is formed from it, spec rule 8).
    const double theta_d = (double)cfo_f - (double)cfo_c;
    const unsigned long long Qp = (unsigned long long)(long long)__builtin_rint(theta_d * WR_TWO_OVER_PI_D * 4611686018427387904.0);
    // LMS / COMB / STA: ... kept in LDS for the data symbols, behind the row's previous pilots (every row's lane 0 writes its
    // frame's; a symbol that forms exact phasors -- every eighth -- reads it back): two registers fewer across the loops.  In these
    // instances the pair was spilled and reloaded INSIDE the prefetch loops (tests/test_isa_prefetch_wait.py); the LS instance has
    // the registers, and the read would cost it 1 % (profiles/r05_ab_spills_in_prefetch_loops.txt).
    constexpr bool QP_LDS = EQ != WIFIRX_EQ_LS;
    constexpr int PVS = QP_LDS ? 6 : 4;                         // float2 per row of the previous-pilot area (16-byte aligned rows)
    uint2* const qpl = reinterpret_cast<uint2*>(qlds + WR_QLDS_PREV(WR_QLDS_SCRATCH_EQ(EQ))) + PVS * (lane >> 4) + 4;
    if (QP_LDS && (lane & 15) == 0) *qpl = make_uint2((uint32_t)Qp, (uint32_t)(Qp >> 32));
    c32 u16;                                                   // exp(j theta 16)
    {
        const unsigned long long q16 = Qp * 16ull;
        sp_sincos_q((uint32_t)(q16 >> 32), (uint32_t)q16, u16.im, u16.re);
    }
    c32 u80;                                                   // exp(j theta 80): one symbol further on
    {
        const unsigned long long q80 = Qp * 80ull;
        sp_sincos_q((uint32_t)(q80 >> 32), (uint32_t)q80, u80.im, u80.re);
    }
    c32 wbase = { 1.0f, 0.0f };                                // phasor of this lane's first sample of the current symbol
    const float rp_bpsk = sp_recip(1.0f);                      // LMS / STA: 1 / |BPSK point|^2 as the rule forms it
    const float rp_qpsk = sp_recip(fma_(WR_LEVEL_QPSK, WR_LEVEL_QPSK, WR_LEVEL_QPSK * WR_LEVEL_QPSK));     // LMS / STA: 1 / |QPSK point|^2
    int n_sym = 0, n_bpsc = 1, n_out = 0, enc = 0, psdu_len = 0;
    bool have_signal = false, want_llr = false;
    float snr = 0.0f;

    // ---- twiddles of stage 1 (output q = j at n = r) and stage 2 (lane (q1 = r>>2, m = r&3)): a 6 x 16 table in
    //      LDS, read back every symbol (registers are the scarce resource of this kernel, the LDS pipe is idle) ----
    float2* twl = reinterpret_cast<float2*>(qlds + WR_QLDS_TW(QS));
    if (row == 0) {
#pragma unroll
        for (int j = 1; j < 4; j++) {
            int e1 = (j * r) & 63, e2 = (j * (r & 3) * 4) & 63;
            twl[(j - 1) * 16 + r] = make_float2(WR_TWIDDLE64[2 * e1], WR_TWIDDLE64[2 * e1 + 1]);
            twl[(j + 2) * 16 + r] = make_float2(WR_TWIDDLE64[2 * e2], WR_TWIDDLE64[2 * e2 + 1]);
        }
    }
    float* stl = qlds + WR_QLDS_STAT(QS) + 4 * row;                           // the row's running sums (lane r = 0 updates them)
    if (stat_all != nullptr && r < 4) stl[r] = 0.0f;
    float2* Hl = reinterpret_cast<float2*>(qlds + WR_QLDS_H(QS)) + lane;      // element j at Hl[64 j]
    float2* pvl = reinterpret_cast<float2*>(qlds + WR_QLDS_PREV(QS)) + PVS * row;
    float* Wl = qlds + WR_QLDS_W(QS) + lane;                                  // element j at Wl[64 j]
    float2* DHl = reinterpret_cast<float2*>(qlds + (COMB ? WR_QLDS_DH(QS) : WR_QLDS_H(QS))) + lane;   // COMB: d_H; else = Hl
    float cw[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, cu[4] = { 0.0f, 0.0f, 0.0f, 0.0f };           // COMB: interpolation weights
    if (COMB) {
#pragma unroll
        for (int j = 0; j < 4; j++) { cw[j] = WR_COMB_W[r + 16 * j]; cu[j] = WR_COMB_U[r + 16 * j]; }
    }
    // STA: the float32 reciprocal of the number of used bins within +-2 of bin r + 16 j is 1/5 but for eight bins next to the band
    // edges and the unused centre bin -- formed where it is used from lane masks (scalar constants), not kept in four registers
    // across the loops (the STA instance spilled inside its prefetch loops: tests/test_isa_prefetch_wait.py)
    auto sta_inv_of = [&](int j) __attribute__((always_inline)) -> float {
        // bins with three neighbours in the window: 6 (j 0, r 6), 58 (j 3, r 10); with four: 7; 30, 31; 33, 34; 57
        const uint64_t m3 = j == 0 ? 0x0040ull : j == 3 ? 0x0400ull : 0ull;
        const uint64_t m4 = j == 0 ? 0x0080ull : j == 1 ? 0xC000ull : j == 2 ? 0x0006ull : 0x0200ull;
        float v = 0x1.99999ap-3f;                                                        // 1/5
        v = __builtin_amdgcn_inverse_ballot_w64(m4 * 0x0001000100010001ull) ? 0.25f : v;
        if (m3) v = __builtin_amdgcn_inverse_ballot_w64(m3 * 0x0001000100010001ull) ? 0x1.555556p-2f : v;      // 1/3
        return v;
    };
    int carrier0[4];                     // data carrier number 0..47 of bin r + 16 j, -1 for pilots / DC / guards
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = r + 16 * j;
        const bool data = (i >= 6 && i <= 58 && i != 11 && i != 25 && i != 32 && i != 39 && i != 53);
        carrier0[j] = data ? (i - 6 - (i > 11) - (i > 25) - (i > 32) - (i > 39) - (i > 53)) : -1;
    }
    __builtin_amdgcn_wave_barrier();
    // LDS transposes (8-byte elements).  Element (hi, mid, lo) of a row lives at
    //   TP_PAD:  144*row + 33*hi + 4*mid + lo            (round 5; every instance but COMB, whose scratch area is too short)
    //   else:    128*(row>>1) + 32*hi + 16*(row&1) + 4*mid + (hi ^ lo)
    // Either way the 32 lanes of a row pair hit 32 different 8-byte banks both when a register index is fixed and the lane pair
    // (mid, lo) varies and when lo is fixed and (hi, mid) varies.  The xor swizzle pays for that with an address register per
    // access (the offset of register j is not a constant: 16 registers across the symbol loops); with the PADDING -- a stride of 33
    // between the hi-planes, 144 between rows (= 16 banks apart modulo 32) -- the four accesses of a phase are ONE lane address +
    // constants: four address registers instead of sixteen, and the accesses pair into ds_write2_b64 / ds_read2_b64 (eight LDS
    // instructions fewer per symbol).  4 x 144 x 8 = 4 608 bytes of the scratch area (4 800).
    #ifndef WR_TP_PAD
#define WR_TP_PAD 1
#endif
    constexpr bool TP_PAD = WR_TP_PAD && EQ != WIFIRX_EQ_COMB;
    float2* ql = reinterpret_cast<float2*>(qlds) + (TP_PAD ? 144 * row : 128 * (row >> 1) + 16 * (row & 1));
    const int m4 = 4 * (r & 3), c2 = r >> 2;
    // transpose A: stage-1 output q of lane r=(m, c) is element (q, m, c); lane (q1=c2, m) reads (q1, m, j)
    // transpose B: stage-2 output q2 of lane (q1=c2, m) is element (q2, q1, m); lane (q1=r&3, q2=c2) reads (q2, q1, j)
    const size_t per = (size_t)prm.max_sym * 48;
    // The rows of a wave hold consecutive records: the output rows are addressed from the wave's first record (scalar
    // registers) plus a 32-bit distance per lane -- no 64-bit pointer per output and lane.
    const uint64_t have_out = __ballot(out >= 0);
    const int out_base = have_out ? __builtin_amdgcn_readlane(out, (int)__builtin_ctzll(have_out)) : 0;
    const uint32_t row_o = out >= 0 ? (uint32_t)(out - out_base) * (uint32_t)per : 0u;
    const uint32_t row_l = row_o * prm.llr_bits;
    uint8_t* idx = idx_all ? idx_all + (size_t)out_base * per : nullptr;
    float*   llr = llr_all ? llr_all + (size_t)out_base * per * prm.llr_bits : nullptr;
    float2*  car = car_all ? car_all + (size_t)out_base * per : nullptr;
    // BPSK / QPSK rows can leave as whole 16-byte pieces (store_bins_lines) when the output rows are aligned for it
    const bool lines_ok = (reinterpret_cast<uintptr_t>(idx) & 3) == 0 && (reinterpret_cast<uintptr_t>(llr) & 15) == 0 && ((per * prm.llr_bits) & 3) == 0;

    int pk = 0;                                                // (s - 2) mod 127: index into the pilot polarity sequence
    uint64_t polw = WR_POLARITY_NEG_LO;                        // ... or: the sequence from the current symbol on,
    int poln = 64;                                             //     the bits left in the window
    int polhi = 0;                                             //     and which half of the 127 it came from (scalar 0 / 1)
    (void)pk;
    float t4 = WR_T4_64F[0];                                   // float32 (2 pi s 80) / 64 of the current symbol
    const float* t4_next = WR_T4_64F + 1;                      // data loop: where the factor after next stands
    // One symbol of the four frames.  DATA (compile time): the symbol is known to be a data symbol (s >= 3) -- the steady
    // state of the loop, compiled without the tests and flag juggling of the LTS and SIGNAL symbols (the scalar side of an
    // iteration costs almost what its vector side does: 32 more scalar instructions per symbol = +2 % time).  Returns false
    // when no row has a symbol left.
    int s_end = 0;          // data loop: first symbol index the row no longer has
    int nbu_all = 0;        // data loop: the constellation all rows with data symbols share, 0 if they differ
    bool plain_all = false; // data loop: every row with data symbols wants decisions + LLRs, none the equalised points
    // NBC (compile time, data symbols only): > 0 = every row with data symbols carries this constellation AND the outputs are
    // the usual set (decisions + LLRs without weights, no equalised points): the store code is picked at compile time; 0 = decided
    // per symbol.
    // XC (compile time, with NBC > 0): the rows leave through store_rows_x() -- any output set as whole lines (x_* below say which)
    bool x_idx = false, x_llr = false, x_car = false, x_csi = false;     // wave-uniform, settled when the data symbols begin
    // PC (compile time, BPSK / QPSK loops of the usual output set): the samples of a symbol arrive by LDS-DMA, requested one symbol
    // ahead (pf_* below); the wave waits for them with a COUNTED s_waitcnt that leaves the stores of the symbol before in flight.
    uint32_t pf_voff0 = 0, pf_voff1 = 0;         // byte offsets of this lane's 16 bytes (frames 0 / 1 and 2 / 3) from pf_next
    const float2* pf_next = nullptr;             // wave-uniform: first sample of the NEXT symbol of the reference row
    int pf_end = 0;                              // the symbol index at which every row with data symbols ends
    bool pf_row = false;                         // this row has data symbols (in a prefetch loop: the row is active in every iteration)
    int pf_nst = 0;                              // XK instances: global stores one symbol issues (wave-uniform, constant for the launch)
    // LDS byte address of the buffer and of its second half (wave-uniform: in scalar registers, whatever the compiler can prove)
    const uint32_t pf_m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(qlds + WR_QLDS_PF(QS)));
    const uint32_t pf_m1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pf_m0 + 1024u));
    auto pf_issue = [&](float dep) __attribute__((always_inline)) {
        // two instructions of 1 kB: lanes 0..31 -> 64 samples of frame 0 (2), lanes 32..63 -> frame 1 (3); LDS address = M0 + 16 lane
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                     "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %4"
                     :: "s"(pf_m0), "s"(pf_m1), "v"(pf_voff0), "v"(pf_voff1), "s"(pf_next), "v"(dep) : "memory", "m0");
        pf_next += 80;
    };
    // Can the data loop that begins at symbol s0 take the prefetch?  Every row with data symbols must end at the same symbol and
    // its samples lie within 2^29 bytes above the first such row's.  Sets the lanes' offsets up and requests the first symbol.
    auto pf_setup = [&](uint64_t has_data, int s0) __attribute__((always_inline)) -> bool {
        const int first = (int)__builtin_ctzll(has_data);
        const bool mine = s_end > 3;
        pf_end = __builtin_amdgcn_readlane(s_end, first);
        pf_row = mine;
        const uint64_t xbv = reinterpret_cast<uint64_t>(xb);
        const uint64_t xref = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(xbv >> 32), first) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane((int)xbv, first);
        const int64_t dist = (int64_t)(xbv - xref) / (int64_t)sizeof(float2);        // samples
        if (!__all(!mine || (s_end == pf_end && dist >= 0 && dist < (1 << 26)))) return false;
        const int c_own = (int)dist + fs;
        const int c_ref = __builtin_amdgcn_readlane(c_own, first);
        const int c = mine ? c_own : c_ref;                  // rows without data symbols read the reference row's samples
        const int c0 = __builtin_amdgcn_readlane(c, 0), c1 = __builtin_amdgcn_readlane(c, 16);
        const int c2 = __builtin_amdgcn_readlane(c, 32), c3 = __builtin_amdgcn_readlane(c, 48);
        pf_voff0 = (uint32_t)((lane < 32 ? c0 : c1) + 2 * (lane & 31)) * 8u;
        pf_voff1 = (uint32_t)((lane < 32 ? c2 : c3) + 2 * (lane & 31)) * 8u;
        pf_next = reinterpret_cast<const float2*>(xref) + (128 + 80 * (s0 - 2) + 16);
        __builtin_amdgcn_wave_barrier();
        pf_issue(0.0f);                                      // the first data symbol's samples: nothing to hide behind yet, and
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no stores behind them that the loop's counted wait could count on
        return true;
    };
    auto symbol = [&](auto data_c, auto nb_c, auto x_c, auto p_c, const int s) __attribute__((always_inline)) -> bool {
        constexpr bool DATA = decltype(data_c)::value;
        constexpr int NBC = decltype(nb_c)::value;
        constexpr bool XC = decltype(x_c)::value;
        constexpr bool PC = decltype(p_c)::value;
        const int off0 = fs + ((!DATA && s < 2) ? 64 * s : 128 + 80 * (s - 2) + 16);
        bool act;
        if (DATA) {
            // how far the row's frame goes was settled when the data symbols began; in a prefetch loop every row with data symbols
            // ends at the same symbol (pf_end, where the loop ends): which rows are active does not change from symbol to symbol
            act = PC ? pf_row : s < s_end;
        } else {
            act = alive && (s <= n_sym + 2);
            if (act && (off0 + 64 > L || (s > 2 && (s - 3) >= (int)prm.max_sym))) {
                flags |= WIFIRX_F_TRUNCATED;
                alive = false;
                act = false;
            }
        }
        if (!(DATA && PC) && !__any(act)) return false;       // (a prefetch loop runs to pf_end: a scalar bound, no vote per symbol)

        // ---- samples r + 16 j of the symbol (rows without a symbol get zeros) ----
        c32 v[4], cur[4];
        {
            if (PC) {
                // the DMA of this symbol's samples was issued before the stores of the symbol before: NK pieces + one dword of
                // decisions (+ the stores of the plane words: 1 / 1 / 2 / 3) may still be in flight behind it (the counter is in order)
#ifdef WR_PF_WAIT_ALL
                constexpr int NST = 0;
#else
                constexpr int NST = (12 * (NBC ? NBC : 1) + 15) / 16 + 1 + (HB ? ((NBC ? NBC : 1) + 1) / 2 : 0);
#endif
                if (XC) {
                    // the XK instances: how many stores a symbol issues is a constant of the launch (which outputs are wanted),
                    // not of the instance -- the immediate of the wait is picked by a scalar switch
                    switch (pf_nst) {
#define WR_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
                    WR_W(1) WR_W(2) WR_W(3) WR_W(4) WR_W(5) WR_W(6) WR_W(7) WR_W(8) WR_W(9) WR_W(10) WR_W(11)
#undef WR_W
                    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    }
                } else
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
                const float2* pb = reinterpret_cast<const float2*>(qlds + WR_QLDS_PF(QS)) + 64 * row + r;
#pragma unroll
                for (int j = 0; j < 4; j++) { const float2 t = pb[16 * j]; cur[j] = { t.x, t.y }; }
            } else
            if (NBC != 0 || lo_zero) {  // act => off0 + 63 < L <= m_hi: every sample of the symbol is in range (the constellation loops are entered only then)
                // rows without a symbol read the first 64 samples of their (or the wave's first) frame instead: finite
                // numbers that no output sees -- cheaper than zeroing eight registers per symbol for them
                const float2* p = xb + ((act ? off0 : 0) + r);
#pragma unroll
                for (int j = 0; j < 4; j++) { const float2 t = load_global_f2(p + 16 * j); cur[j] = { t.x, t.y }; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) cur[j] = load_y(xb, off0 + r + 16 * j, m_lo, act ? m_hi : 0);
            }
        }
        {   // one rotation by the total offset (spec rule 8): the lane's base phasor from the exact integer phase at
            // s = 0, 1, 9, 17, ..., in between carried from symbol to symbol by exp(j theta 80); then steps of exp(j theta 16)
            if ((!DATA && s < 2) || ((s - 1) & 7) == 0) {        // wave-uniform
                unsigned long long qp = Qp;
                if (DATA && QP_LDS) { const uint2 q2 = *qpl; qp = ((unsigned long long)q2.y << 32) | q2.x; }
                const unsigned long long ph = qp * (unsigned long long)(unsigned)(off0 + r);
                sp_sincos_q((uint32_t)(ph >> 32), (uint32_t)ph, wbase.im, wbase.re);
            } else {
                wbase = sp_cmul(wbase, u80);
            }
            c32 w = wbase;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[j] = sp_cmul(cur[j], w);                         // rows without a symbol loaded zeros above
                w = sp_cmul(w, u16);
            }
            // the buffer has been read (the products above used it): the next symbol's samples may land in it
            if (PC && s + 1 < pf_end) pf_issue(v[3].re);
        }
        // ---- FFT-64: three in-register radix-4 stages, two transposes through LDS ----
        bfly4_reg(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            c32 y = v[0];
            if (j) { float2 w = twl[(j - 1) * 16 + r]; y = sp_cmul(v[j], c32{ w.x, w.y }); }
            ql[TP_PAD ? 33 * j + m4 + c2 : 32 * j + m4 + (j ^ c2)] = make_float2(y.re, y.im);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float2 t = ql[TP_PAD ? 33 * c2 + m4 + j : 32 * c2 + m4 + (c2 ^ j)];
            v[j] = { t.x, t.y };
        }
        __builtin_amdgcn_wave_barrier();
        bfly4_reg(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            c32 y = v[0];
            if (j) { float2 w = twl[(j + 2) * 16 + r]; y = sp_cmul(v[j], c32{ w.x, w.y }); }
            ql[TP_PAD ? 33 * j + 4 * c2 + (r & 3) : 32 * j + 4 * c2 + (j ^ (r & 3))] = make_float2(y.re, y.im);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float2 t = ql[TP_PAD ? 33 * c2 + m4 + j : 32 * c2 + m4 + (c2 ^ j)];
            v[j] = { t.x, t.y };
        }
        __builtin_amdgcn_wave_barrier();
        bfly4_reg(v[0], v[1], v[2], v[3]);
        // stage-3 output q3 is sub-carrier k = r + 16 q3, i.e. shifted bin i = r + 16 ((q3 + 2) & 3)
        c32 X[4] = { v[2], v[3], v[0], v[1] };

        // (1) sampling offset (spec rule 9)
        {
            // upstream: 2 pi s 80 (eps0 + d_er) / 64 in double; spec: the factor that depends on s alone comes from a
            // float32 table, eps0 and d_er are float32
            const float kf = t4 * (eps0 + d_er);
#if WR_T4_POINTER
            if (DATA) { t4 = PC ? load_const_f(t4_next) : *t4_next; t4_next++; }             // the next symbol's factor (s + 1 <= 514: inside the table), requested a whole iteration early
            else { t4 = WR_T4_64F[s + 1]; t4_next = WR_T4_64F + (s + 2); }
#else
            t4 = WR_T4_64F[s < 518 ? s + 1 : 519];              // the next symbol's factor: requested a whole iteration early
#endif
            // b = phasor of bin r + 16; lane 0 of the row holds exp(-j kf 16), whose conjugate is the step
            c32 b;
            sp_sincos_small(kf * (float)(r - 16), b.im, b.re);
            const c32 b0 = row_bcast<0>(b), step = { b0.re, -b0.im };
            const c32 q0 = sp_cmul(b, b0), q2 = sp_cmul(b, step), q3 = sp_cmul(q2, step);
            X[0] = sp_cmul(X[0], q0);
            X[1] = sp_cmul(X[1], b);
            X[2] = sp_cmul(X[2], q2);
            X[3] = sp_cmul(X[3], q3);
        }
        // (2) pilots: bins 11, 25, 39, 53 = (lane 11, j 0), (lane 9, j 1), (lane 7, j 2), (lane 5, j 3)
        c32 X11 = row_bcast<11>(X[0]), X25 = row_bcast<9>(X[1]), X39 = row_bcast<7>(X[2]), X53 = row_bcast<5>(X[3]);
        // polarity of symbol s - 2 as a sign-bit mask (wave-uniform: scalar registers): negating = xor; pilot sum S,
        // (3) the pilots with the polarity removed and the residual offset estimate -- one branch on the kind of symbol
        uint32_t sgn = 0;
        c32 S, cur0, cur1, cur2, cur3;
        float er = 0.0f;
        if (DATA || s >= 2) {
#if WR_POLARITY_WINDOW
            // the polarity sequence as a window that moves one bit per symbol (two scalar shifts; refilled every 64 / 63 symbols)
            sgn = (uint32_t)polw << 31;
            polw >>= 1;
            // (the refill -- every 64th / 63rd symbol -- as a branch the scalar unit almost never takes: written without the hint it
            //  became ten always-executed scalar selects and a detour of the flag through a vector register per symbol)
            if (__builtin_expect(--poln == 0, 0)) { polhi ^= 1; polw = polhi ? WR_POLARITY_NEG_HI : WR_POLARITY_NEG_LO; poln = 64 - polhi; }
#else
            const uint64_t bits = pk < 64 ? (WR_POLARITY_NEG_LO >> pk) : (WR_POLARITY_NEG_HI >> (pk - 64));
            sgn = (uint32_t)(bits & 1ull) << 31;
#endif
            S = cflip(csub(cadd(cadd(X11, X39), X25), X53), sgn);
            cur0 = cflip(X11, sgn);
            cur1 = cflip(X25, sgn);
            cur2 = cflip(X39, sgn);
            cur3 = cflip(X53, sgn ^ 0x80000000u);
            const float2 q0 = pvl[0], q1 = pvl[1], q2 = pvl[2], q3 = pvl[3];
            const c32 prev0 = { q0.x, q0.y }, prev1 = { q1.x, q1.y }, prev2 = { q2.x, q2.y }, prev3 = { q3.x, q3.y };
            c32 acc = cadd(cadd(cadd(sp_conj_mul(prev0, cur0), sp_conj_mul(prev1, cur1)),
                                sp_conj_mul(prev2, cur2)), sp_conj_mul(prev3, cur3));
            er = sp_atan2(acc.im, acc.re) * er_scale;
#if !WR_POLARITY_WINDOW
            pk = pk == 126 ? 0 : pk + 1;                       // (s - 2) mod 127 of the next symbol
#endif
        } else {
            S = cadd(cadd(csub(X11, X25), X39), X53);
            cur0 = X11; cur1 = cneg(X25); cur2 = X39; cur3 = X53;
        }
        __builtin_amdgcn_wave_barrier();
        if (r == 0) {
            pvl[0] = make_float2(cur0.re, cur0.im); pvl[1] = make_float2(cur1.re, cur1.im);
            pvl[2] = make_float2(cur2.re, cur2.im); pvl[3] = make_float2(cur3.re, cur3.im);
        }
        __builtin_amdgcn_wave_barrier();
        // (4) common phase: exp(-j beta) = conj(S) rsqrt(|S|^2) (spec rule 10)
        {
            float n2 = fma_(S.im, S.im, S.re * S.re);
            float inv = sp_rsqrt(n2);
            float cs = (n2 > 0.0f) ? S.re * inv : 1.0f;
            float sn = (n2 > 0.0f) ? -(S.im * inv) : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; j++) X[j] = sp_rot(X[j], sn, cs);
        }
        // (5) IIR
        if (DATA || s >= 2) d_er = fma_(0.1f, er, 0.9f * d_er);
        // (6a) COMB: this symbol's pilots (polarity removed) are the channel at bins 11, 25, 39, 53, their mean stands at
        //      the band edges (bins 0 and 64); linear interpolation, then d_H = 0.8 d_H + 0.2 H (d_H = H at s = 0)
        if (COMB) {
            c32 n1 = row_bcast<11>(X[0]), n2 = row_bcast<9>(X[1]), n3 = row_bcast<7>(X[2]), n4 = row_bcast<5>(X[3]);
            if (!DATA && s < 2) n2 = cneg(n2);
            else { n1 = cflip(n1, sgn); n2 = cflip(n2, sgn); n3 = cflip(n3, sgn); n4 = cflip(n4, sgn ^ 0x80000000u); }
            const c32 sum = cadd(cadd(cadd(n1, n2), n3), n4);
            const c32 n0 = { 0.25f * sum.re, 0.25f * sum.im };
            const c32 node[6] = { n0, n1, n2, n3, n4, n0 };
            const int edge[4] = { 11, 25, 39, 53 };
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool lo = (r + 16 * j) <= edge[j];            // segment j (nodes j, j+1) or j+1 (nodes j+1, j+2)
                const c32 a = lo ? node[j] : node[j + 1], b = lo ? node[j + 1] : node[j + 2];
                const float hr = fma_(b.re, cw[j], a.re * cu[j]), hi = fma_(b.im, cw[j], a.im * cu[j]);
                if (!DATA && s == 0) DHl[64 * j] = make_float2(hr, hi);
                else {
                    const float2 o = DHl[64 * j];
                    DHl[64 * j] = make_float2(0.8f * o.x + 0.2f * hr, 0.8f * o.y + 0.2f * hi);
                }
            }
        }
        // (6) LS equalizer
        if (!DATA && s == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) Hl[64 * j] = make_float2(X[j].re, X[j].im);
        } else if (!DATA && s == 1) {
            float nv[4], sv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float2 h0 = Hl[64 * j];
                const c32 Hj = { h0.x, h0.y };
                c32 d = csub(Hj, X[j]), u = cadd(Hj, X[j]);
                const int i = r + 16 * j;
                const bool usedj = (i >= 6 && i <= 58 && i != 32);
                nv[j] = usedj ? fma_(d.im, d.im, d.re * d.re) : 0.0f;
                sv[j] = usedj ? fma_(u.im, u.im, u.re * u.re) : 0.0f;
                float g = 0.5f * WR_LTS_FREQ[i];
                const float hr = u.re * g, hi = u.im * g;
                const float dd = fma_(hi, hi, hr * hr);             // |H|^2: LS equaliser's divisor, LLR weight
                if (prm.llr_csi) Wl[64 * j] = usedj ? dd : 0.0f;
                if (csi_all && usedj && act) csi_all[(size_t)out * 52 + (i - 6 - (i > 32))] = make_float2(hr, hi);
                if (LMS || STA) {
                    Hl[64 * j] = usedj ? make_float2(hr, hi) : make_float2(1.0f, 0.0f);
                } else if (COMB) {
                    // COMB equalises with d_H (own LDS area); the LS estimate only feeds the SNR figure and the CSI
                } else {
                    // G = conj(H)/|H|^2 replaces H in LDS: the one-tap equaliser as a multiplier
                    Hl[64 * j] = usedj ? make_float2(hr / dd, -hi / dd) : make_float2(0.0f, 0.0f);
                }
                // the spec's xor tree over the 64 bins: steps 1,2,4,8 inside the row ...
                nv[j] = row_xor_sum16(nv[j]);
                sv[j] = row_xor_sum16(sv[j]);
            }
            // ... step 16 pairs bins i, i^16 (registers j, j^1), step 32 pairs registers j, j^2
            float n01 = nv[0] + nv[1], n23 = nv[2] + nv[3], s01 = sv[0] + sv[1], s23 = sv[2] + sv[3];
            float noise = n01 + n23, signal = s01 + s23;
            if (act) snr = sp_snr_db(signal, noise);
        } else {
            c32 Y[4];
            int carrier[4];
            c32 HU[4];                                   // STA: this symbol's per-bin estimates of my bins
            // (in a constellation loop the rows' constellation is the compile-time NBC: decisions and points of LMS / STA are then
            //  compiled for it, without the per-lane walk through all four)
            const bool const_mag = (LMS || STA) && (NBC != 0 ? NBC <= 2 : __all((!DATA && s == 2) || n_bpsc <= 2));      // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; j++) {
                carrier[j] = carrier0[j];
                if (NBC == 0) asm volatile("" : "+v"(carrier[j]));     // keeps base + carrier out of loop-invariant registers (the constellation loops do better without)
                Y[j] = { 0.0f, 0.0f };
                HU[j] = X[j];
                // every bin is equalised, data or not (LS: G = 0 on unused bins; the others: H = 1 there from the LTS step on;
                // pilot and unused bins are never stored, and no branch has to fence them off -- except in the STA instance's
                // general loop, whose register allocation was the better for it: 20.1 vs 21.1 ms; in its constellation loops the
                // fence costs 1.4 %)
                if (!(STA && NBC == 0) || carrier[j] >= 0) {
                    const float2 g0 = DHl[64 * j];
                    if (DIV) {
                        // spec rule 11: one reciprocal of |H|^2 per bin, then products
                        const float rd = sp_recip(fma_(g0.y, g0.y, g0.x * g0.x));
                        Y[j].re = fma_(X[j].im, g0.y, X[j].re * g0.x) * rd;
                        Y[j].im = fma_(X[j].im, g0.x, -(X[j].re * g0.y)) * rd;
                        if (LMS || STA) {
                            const int nbl = NBC != 0 ? NBC : (!DATA && s == 2) ? 1 : n_bpsc;
                            const c32 pt = point_of(decide(Y[j], nbl), nbl);
                            // 1 / |point|^2: BPSK and QPSK points all have one magnitude, the quotient is formed once per wave
                            // (same value); the wave divides per bin only when a row carries 16- or 64-QAM
                            float rp;
                            if (const_mag) rp = nbl == 1 ? rp_bpsk : rp_qpsk;
                            else           rp = sp_recip(fma_(pt.im, pt.im, pt.re * pt.re));
                            const float tr = fma_(X[j].im, pt.im, X[j].re * pt.re) * rp;
                            const float ti = fma_(X[j].im, pt.re, -(X[j].re * pt.im)) * rp;
                            // (no test of `act`: a row without a symbol has none left either, and what it writes into its own estimate is never read)
                            if (LMS) Hl[64 * j] = make_float2(0.5f * g0.x + 0.5f * tr, 0.5f * g0.y + 0.5f * ti);
                            HU[j] = { tr, ti };
                        }
                    } else {
                        Y[j] = sp_cmul(X[j], c32{ g0.x, g0.y });
                    }
                }
            }
            if (STA) {
                // pilots: X times the known pilot value; then every used bin takes the mean of the estimates of the used
                // bins within +-2 (ascending), and H = H/2 + mean/2.  The exchange goes through the FFT's LDS area.
                if (r == 11) HU[0] = cflip(X[0], sgn);
                if (r == 9)  HU[1] = cflip(X[1], sgn);
                if (r == 7)  HU[2] = cflip(X[2], sgn);
                if (r == 5)  HU[3] = cflip(X[3], sgn ^ 0x80000000u);
                // rows of 68 entries in the FFT's LDS area: the 64 bins with the unused ones written as zero, two zeros on
                // either side -- the window sum then needs no per-term test (spec rule 11: zeros take part in the sum)
                float2* hu = reinterpret_cast<float2*>(qlds) + 68 * row + 2;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = r + 16 * j;
                    const bool usedj = (i >= 6 && i <= 58 && i != 32);
                    hu[i] = usedj ? make_float2(HU[j].re, HU[j].im) : make_float2(0.0f, 0.0f);
                }
                if (r < 2) { hu[r - 2] = make_float2(0.0f, 0.0f); hu[64 + r] = make_float2(0.0f, 0.0f); }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = r + 16 * j;
                    const bool usedj = (i >= 6 && i <= 58 && i != 32);
                    const float2 v0 = hu[i - 2], v1 = hu[i - 1], v2 = hu[i], v3 = hu[i + 1], v4 = hu[i + 2];
                    const c32 sum = cadd(cadd(cadd(cadd(c32{ v0.x, v0.y }, c32{ v1.x, v1.y }), c32{ v2.x, v2.y }), c32{ v3.x, v3.y }),
                                         c32{ v4.x, v4.y });
                    const float2 o = Hl[64 * j];
                    const float inv = sta_inv_of(j);
                    if (act && usedj) Hl[64 * j] = make_float2(0.5f * o.x + 0.5f * (sum.re * inv), 0.5f * o.y + 0.5f * (sum.im * inv));
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (!DATA && s == 2) {
                // (7) SIGNAL: per frame, gather the 48 BPSK decisions in carrier order, de-interleave, Viterbi
                uint64_t bal[4];
#pragma unroll
                for (int j = 0; j < 4; j++) bal[j] = __ballot(carrier[j] >= 0 && Y[j].re > 0.0f);
                const uint64_t actmask = __ballot(act);
                // lane jj < 48 picks de-interleaved bit jj = carrier bit 3 (jj mod 16) + jj / 16 of every frame
                const int dsrc = lane < 48 ? 3 * (lane & 15) + (lane >> 4) : 63;      // bit 63 of cm is 0
                uint64_t de[4];
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    uint64_t b = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) b |= ((bal[j] >> (16 * f)) & 0xffffull) << (16 * j);
                    const uint64_t cm = ((b >> 6) & 0x1full) | (((b >> 12) & 0x1fffull) << 5) |
                                        (((b >> 26) & 0x3full) << 18) | (((b >> 33) & 0x3full) << 24) |
                                        (((b >> 40) & 0x1fffull) << 30) | (((b >> 54) & 0x1full) << 43);
                    de[f] = __ballot(((cm >> dsrc) & 1ull) != 0);
                }
                uint32_t sig4[4];
                viterbi_signal4(de, lane, sig4);
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    if (!((actmask >> (16 * f)) & 1)) continue;
                    const uint32_t sig = sig4[f];
                    int e = 0, len = 0;
                    bool ok = parse_signal(sig, e, len);
                    if (row == f) {
                        if (!ok) {
                            alive = false;
                        } else {
                            const int nbpsc_tab[8] = { 1, 1, 2, 2, 4, 4, 6, 6 };
                            const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
                            have_signal = true;
                            enc = e;
                            psdu_len = len;
                            n_bpsc = nbpsc_tab[e];
                            int nd = ndbps_tab[e];
                            n_sym = (16 + 8 * len + 6 + nd - 1) / nd;
                            flags |= WIFIRX_F_SIGNAL;
                            want_llr = (llr != nullptr) && ((int)prm.llr_bits >= n_bpsc);
                            if (want_llr) flags |= WIFIRX_F_LLR;
                        }
                    }
                }
            } else {
                // (8) data symbol q.  The constellation is usually the same in all active rows: then the slicer and
                // the LLR stores are picked by one scalar branch; mixed rows take the per-constellation passes.
                const int q = s - 3;
                const bool has_idx = idx_all != nullptr, has_car = car_all != nullptr;
                int out_l = out;                                  // the row's plane words are addressed from the record index
                if (HB) asm volatile("" : "+v"(out_l));           // every symbol anew: no loop-invariant pointer in registers
                const uint64_t act_m = __ballot(act);
                // DATA: whether the rows with data symbols share a constellation was settled once (nbu_all > 0: they do)
                // ... if not (rows of different rates and lengths), the rows that are left may still agree: the first active row's
                // constellation against the others', symbol by symbol
                const bool settled = DATA && nbu_all > 0;                    // wave-uniform
                const int nbu = settled ? nbu_all : __builtin_amdgcn_readlane(n_bpsc, act_m ? (int)__builtin_ctzll(act_m) : 0);
                const bool uniform = settled || (act_m & ~__ballot(n_bpsc == nbu)) == 0;
                const float w1[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
                const bool csi = prm.llr_csi != 0 && llr_all != nullptr;     // wave-uniform
                // (a caller that wants PDUs only asks for the bit planes alone: then the per-bin stores have nothing to write)
                const bool no_bins = HB && !has_idx && !has_car && llr_all == nullptr;     // wave-uniform, constant for the launch
#define WR_STORE(NB, OK)                                                                                        \
                { if (no_bins) { }                                                                                      \
                  else if (csi) { const float wq[4] = { Wl[0], Wl[64], Wl[128], Wl[192] };                                   \
                             store_bins<NB, true>(Y, carrier, OK, q, idx, car, llr, has_idx, has_car, want_llr, wq, row_o, row_l); }  \
                  else if (DATA && plain_all) store_bins<NB, false, true>(Y, carrier, OK, q, idx, car, llr, true, false, true, w1, row_o, row_l); \
                  else     store_bins<NB, false>(Y, carrier, OK, q, idx, car, llr, has_idx, has_car, want_llr, w1, row_o, row_l); \
                  if (HB) { __builtin_amdgcn_sched_barrier(0);                                                          \
                            store_hbits<NB>(Y, OK, q, hb_all + (size_t)(unsigned)out_l * (prm.max_sym * 12u), r); } }
                if (NBC != 0 && XC) {
                    store_rows_x<(NBC ? NBC : 1), EQ != WIFIRX_EQ_LS>(Y, carrier, act, q, idx, llr, car, x_idx, x_llr, x_car, x_csi, Wl, row_o, row_l, qlds, row, r);
                    if (HB) { __builtin_amdgcn_sched_barrier(0);
                              store_hbits<(NBC ? NBC : 1)>(Y, act, q, hb_all + (size_t)(unsigned)out_l * (prm.max_sym * 12u), r); }
                } else if (NBC != 0) {
                    if (WR_STORE_AS_LINES && (NBC <= 2 || (WR_STORE_AS_LINES > 2 && !COMB)) && (!HB || WR_STORE_AS_LINES > 1) && (PC || lines_ok))     // (a prefetch loop is entered only with lines_ok)
                        store_bins_lines<(NBC ? NBC : 1), EQ != WIFIRX_EQ_LS>(Y, carrier, act, q, idx, llr, row_o, row_l, qlds, row, r);
                    else
                    store_bins<(NBC ? NBC : 1), false, true>(Y, carrier, act, q, idx, car, llr, true, false, true, w1, row_o, row_l);
                    if (HB) { __builtin_amdgcn_sched_barrier(0);
                              store_hbits<(NBC ? NBC : 1)>(Y, act, q, hb_all + (size_t)(unsigned)out_l * (prm.max_sym * 12u), r); }
                } else if (uniform) {
                    if (nbu == 1)      WR_STORE(1, act)
                    else if (nbu == 2) WR_STORE(2, act)
                    else if (nbu == 4) WR_STORE(4, act)
                    else               WR_STORE(6, act)
                } else {
                    WR_STORE(1, act && n_bpsc == 1)
                    WR_STORE(2, act && n_bpsc == 2)
                    WR_STORE(4, act && n_bpsc == 4)
                    WR_STORE(6, act && n_bpsc == 6)
                }
#undef WR_STORE
                if ((NBC == 0 || XC) && stat_all != nullptr) {     // (the plain constellation loops are entered only without it)
                    // moments of the equalised points for the probe_mpsk_snr_est consumer (IRS_AP.py:275,312): per lane
                    // the data bins r + 16 j in ascending j (others 0), the row by the spec's xor tree, the frame symbol
                    // after symbol
                    float p1 = 0.0f, p2 = 0.0f, p4 = 0.0f;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float m2 = fma_(Y[j].im, Y[j].im, Y[j].re * Y[j].re);
                        const bool d = carrier[j] >= 0;
                        const float a1 = d ? __builtin_sqrtf(m2) : 0.0f, a2 = d ? m2 : 0.0f, a4 = d ? m2 * m2 : 0.0f;
                        p1 = j ? p1 + a1 : a1; p2 = j ? p2 + a2 : a2; p4 = j ? p4 + a4 : a4;
                    }
                    p1 = row_xor_sum16(p1); p2 = row_xor_sum16(p2); p4 = row_xor_sum16(p4);
                    if (act && r == 0) { stl[0] += p1; stl[1] += p2; stl[2] += p4; }
                }
                if (act) n_out = q + 1;
            }
        }
        return true;
    };
    {
        int s = 0;
        bool more = true;
        typedef std::integral_constant<int, 0> nb_any;
        typedef std::false_type no_x;
        typedef std::false_type no_p;
        for (; more && s < 3; s++) more = symbol(std::false_type{}, nb_any{}, no_x{}, no_p{}, s);
#if defined(WR_ABLATE) && WR_ABLATE == 3      // timing experiment: preamble + the two LTS symbols + SIGNAL, no data symbol
        more = false;
#endif
#if WR_SPLIT_SYMBOL_LOOP
        uint64_t has_data = 0;
        if (more) {
            // the data symbols of a row: 3 .. n_sym + 2, as far as the copied samples (off0 + 64 <= L) and the output rows
            // (s - 3 < max_sym) go; a frame that is cut short is flagged here, once
            const int s_fit = 2 + (L - fs - 208) / 80, s_max = (int)prm.max_sym + 2;
            const int s_lim = s_fit < s_max ? s_fit : s_max;
            const bool cut = alive && s_lim < n_sym + 2;
            if (cut) { flags |= WIFIRX_F_TRUNCATED; alive = false; }
            s_end = (alive || cut) ? (cut ? s_lim : n_sym + 2) + 1 : 0;
            has_data = __ballot(s_end > 3);
            const int nb_first = has_data ? __builtin_amdgcn_readlane(n_bpsc, (int)__builtin_ctzll(has_data)) : 0;
            nbu_all = (has_data & ~__ballot(n_bpsc == nb_first)) == 0 ? nb_first : 0;
            // the usual output set: decisions + LLRs for every row with data symbols, no equalised points
            plain_all = WR_PLAIN_STORES && idx_all != nullptr && car_all == nullptr && (has_data & ~__ballot(want_llr)) == 0;
        }
        const bool special = !XK && WR_NB_LOOPS && more && nbu_all > 0 && plain_all && !(prm.llr_csi != 0 && llr_all != nullptr) &&
                             lo_zero && stat_all == nullptr;      // wave-uniform
#if WR_X_LOOPS
      if constexpr (XK) {
        // any other output set of a wave whose rows share a constellation (round 4): the equalised points of the `carrier` port,
        // weighted LLRs, the probe's moments, planes alone -- constellation loops with store_rows_x()
        const uint64_t llr_rows = has_data & __ballot(want_llr);
        const bool special_x = WR_NB_LOOPS && more && nbu_all > 0 && !special && lo_zero && (llr_rows == 0 || llr_rows == has_data) &&
                               (reinterpret_cast<uintptr_t>(idx) & 3) == 0 && (reinterpret_cast<uintptr_t>(llr) & 15) == 0 &&
                               (reinterpret_cast<uintptr_t>(car) & 15) == 0 && ((per * prm.llr_bits) & 3) == 0 && (per & 1) == 0;
        x_idx = idx_all != nullptr;
        x_llr = llr_rows != 0;
        x_car = car_all != nullptr;
        x_csi = prm.llr_csi != 0 && llr_all != nullptr;
        typedef std::true_type with_x;
#if WR_DMA_PREFETCH
        // ... with the samples by LDS-DMA when the weight area is idle (no weighted LLRs) and every row ends at the same symbol
        bool pfx = false;
        // (... and LLR or point rows are written: with the planes alone the wait has next to nothing to skip, 9.99 vs 10.06 ms)
        if (special_x && !x_csi && (x_llr || x_car) && (!COMB || nbu_all <= 2)) {
            pfx = pf_setup(has_data, s);
            const int nk = (12 * nbu_all + 15) / 16;
            pf_nst = (x_llr ? nk : 0) + (x_idx ? 1 : 0) + (x_car ? 2 : 0) + (HB ? (nbu_all + 1) / 2 : 0);
        }
        typedef std::true_type with_px;
        if (pfx && nbu_all == 1)           { for (; s < pf_end; s++) (void)symbol(std::true_type{}, std::integral_constant<int, 1>{}, with_x{}, with_px{}, s); more = false; }
        else if (pfx && nbu_all == 2)      { for (; s < pf_end; s++) (void)symbol(std::true_type{}, std::integral_constant<int, 2>{}, with_x{}, with_px{}, s); more = false; }
#if WR_NB_LOOPS > 1
        else if (!COMB && pfx && nbu_all == 4) { for (; s < pf_end; s++) (void)symbol(std::true_type{}, std::integral_constant<int, 4>{}, with_x{}, with_px{}, s); more = false; }
        else if (!COMB && pfx && nbu_all == 6) { for (; s < pf_end; s++) (void)symbol(std::true_type{}, std::integral_constant<int, 6>{}, with_x{}, with_px{}, s); more = false; }
#endif
        else
#endif
        if (special_x && nbu_all == 1)      for (; more; s++) more = symbol(std::true_type{}, std::integral_constant<int, 1>{}, with_x{}, no_p{}, s);
        else if (special_x && nbu_all == 2) for (; more; s++) more = symbol(std::true_type{}, std::integral_constant<int, 2>{}, with_x{}, no_p{}, s);
#if WR_NB_LOOPS > 1
        else if (!COMB && special_x && nbu_all == 4) for (; more; s++) more = symbol(std::true_type{}, std::integral_constant<int, 4>{}, with_x{}, no_p{}, s);
        else if (!COMB && special_x && nbu_all == 6) for (; more; s++) more = symbol(std::true_type{}, std::integral_constant<int, 6>{}, with_x{}, no_p{}, s);
#endif
        else                              for (; more; s++) more = symbol(std::true_type{}, nb_any{}, no_x{}, no_p{}, s);
      } else
#endif
      {
#if WR_DMA_PREFETCH
        // The samples by LDS-DMA, one symbol ahead: when every row with data symbols ends at the same symbol, the rows leave as
        // whole lines (the store count the wait relies on), and the rows' samples lie within 2^30 bytes above the first row's.
        bool pf = false;
        if (special && (!COMB || nbu_all <= 2) && lines_ok) pf = pf_setup(has_data, s);        // wave-uniform
        typedef std::true_type with_p;
        if (pf && nbu_all == 1)           { for (; s < pf_end; s++) (void)symbol(std::true_type{}, std::integral_constant<int, 1>{}, no_x{}, with_p{}, s); more = false; }
        else if (pf && nbu_all == 2)      { for (; s < pf_end; s++) (void)symbol(std::true_type{}, std::integral_constant<int, 2>{}, no_x{}, with_p{}, s); more = false; }
#if WR_NB_LOOPS > 1
        else if (!COMB && pf && nbu_all == 4) { for (; s < pf_end; s++) (void)symbol(std::true_type{}, std::integral_constant<int, 4>{}, no_x{}, with_p{}, s); more = false; }
        else if (!COMB && pf && nbu_all == 6) { for (; s < pf_end; s++) (void)symbol(std::true_type{}, std::integral_constant<int, 6>{}, no_x{}, with_p{}, s); more = false; }
#endif
        else
#endif
        if (special && nbu_all == 1)      for (; more; s++) more = symbol(std::true_type{}, std::integral_constant<int, 1>{}, no_x{}, no_p{}, s);
        else if (special && nbu_all == 2) for (; more; s++) more = symbol(std::true_type{}, std::integral_constant<int, 2>{}, no_x{}, no_p{}, s);
#if WR_NB_LOOPS > 1     // (not in the COMB instance: its scratch area is too short for these rows' line stores, and the loops alone cost it 6 %)
        else if (!COMB && special && nbu_all == 4) for (; more; s++) more = symbol(std::true_type{}, std::integral_constant<int, 4>{}, no_x{}, no_p{}, s);
        else if (!COMB && special && nbu_all == 6) for (; more; s++) more = symbol(std::true_type{}, std::integral_constant<int, 6>{}, no_x{}, no_p{}, s);
#endif
        else                              for (; more; s++) more = symbol(std::true_type{}, nb_any{}, no_x{}, no_p{}, s);
      }
#else
        for (; more; s++) more = symbol(std::false_type{}, nb_any{}, no_x{}, no_p{}, s);
#endif
    }
    if (r == 0 && out >= 0) {
        wifirx_frame fr;
        if (have_signal && n_out == n_sym) flags |= WIFIRX_F_COMPLETE;
        bool sync = (flags & WIFIRX_F_SYNC) != 0;
        fr.flags = flags;
        fr.trigger = trig;
        fr.frame_start = sync ? fs : 0;
        fr.cfo_coarse = (flags & WIFIRX_F_DETECTED) ? cfo_c : 0.0f;
        fr.cfo_fine = sync ? cfo_f : 0.0f;
        fr.snr_db = snr;
        fr.psdu_len = (uint16_t)psdu_len;
        fr.encoding = (uint8_t)enc;
        fr.n_bpsc = have_signal ? (uint8_t)n_bpsc : 0;
        fr.n_sym = (uint16_t)n_sym;
        fr.n_sym_out = (uint16_t)n_out;
        frames[out] = fr;
        if (stat_all != nullptr) stat_all[out] = make_float4(stl[0], stl[1], stl[2], 0.0f);
    }
}

}  // namespace wr
