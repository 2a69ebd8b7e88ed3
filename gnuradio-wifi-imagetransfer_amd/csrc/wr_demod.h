// wr_demod.h -- the fused demod kernels (batch and stream form) and their detect phase, as templates: instantiated by
// wr_kernels.hip (XK = false: the contract's usual output set -- decisions + LLRs -- in the constellation loops) and by
// wr_kernels_x.hip (XK = true: every other output set in them: equalised points, weighted LLRs, moments, planes alone).
// Two translation units so that the XK instances leave the timed kernels' code untouched, register allocation included
// (in one kernel the additional loops cost the timed one 0.7 %, profiles/r04_ab_x_loops.txt), and compile side by side.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wifirx.h"
#include "wr_device.h"
#include "wr_kernels.h"
#include "wr_quad.h"

#ifndef WR_ABLATE
#define WR_ABLATE 0
#endif

namespace wr {

// ---------------------------------------------------------------------------------------------
// detect phase (a1 + a2), stream form (stream_detect_kernel): one tile = 64 consecutive samples, lane <-> sample; the
// window sums follow the blocked scheme of the spec (section 4.2): Kogge-Stone prefix H / exclusive suffix T inside blocks of
// 16 lanes (DPP row shifts), block totals and tails of the 3-4 previous blocks fetched across rows.  The batch kernel
// uses detect_quad() below.
struct DetectState {       // what a tile needs from the tile before it
    float Hr, Hi, Hp, Tr, Ti, Tp;
};

__device__ __forceinline__ DetectState detect_state_zero() { return { 0, 0, 0, 0, 0, 0 }; }

// Processes tile [n0, n0+64) given this lane's sample xn = x[n0 + lane] and xd = x[n0 + lane - 16] (zero outside the
// stream).  Returns the ballot of c[n] > thr; Ar/Ai = A[n] of this lane's sample.
__device__ __forceinline__ uint64_t detect_tile_core(c32 xn, c32 xd, long n_samp, long n0, float thr,
                                                     int lane, DetectState& ps, float& Ar, float& Ai)
{
    const int q = lane >> 4;
    const int l16 = (lane + 16) & 63;              // lane that holds index lane-48 (mod 64)
    const int b1 = (((q - 1) & 3) << 4) | 15;      // last lane of block m-1 / m-2 / m-3
    const int b2 = (((q - 2) & 3) << 4) | 15;
    const int b3 = (((q - 3) & 3) << 4) | 15;
    const long n = n0 + lane;
    float ar = fma_(xn.im, xd.im, xn.re * xd.re);
    float ai = fma_(xn.im, xd.re, -(xn.re * xd.im));
    float pw = fma_(xn.im, xn.im, xn.re * xn.re);
    float Hr = row_prefix16(ar), Hi = row_prefix16(ai), Hp = row_prefix16(pw);
    float Tr = dpp_zero<0x101>(row_suffix16(ar));
    float Ti = dpp_zero<0x101>(row_suffix16(ai));
    float Tp = dpp_zero<0x101>(row_suffix16(pw));
    // Every shuffle is issued by all 64 lanes (a shuffle under a divergent condition would read
    // inactive source lanes); the choice between this tile and the previous one is a select.
    float cT3r = shfl(Tr, l16), oT3r = shfl(ps.Tr, l16), cT3i = shfl(Ti, l16), oT3i = shfl(ps.Ti, l16);
    float cB2r = shfl(Hr, b2), oB2r = shfl(ps.Hr, b2), cB2i = shfl(Hi, b2), oB2i = shfl(ps.Hi, b2);
    float cB1r = shfl(Hr, b1), oB1r = shfl(ps.Hr, b1), cB1i = shfl(Hi, b1), oB1i = shfl(ps.Hi, b1);
    float cB3p = shfl(Hp, b3), oB3p = shfl(ps.Hp, b3);
    float cB2p = shfl(Hp, b2), oB2p = shfl(ps.Hp, b2);
    float cB1p = shfl(Hp, b1), oB1p = shfl(ps.Hp, b1);
    float t3r = (q >= 3) ? cT3r : oT3r;            // tail of block m-3 at the same r
    float t3i = (q >= 3) ? cT3i : oT3i;
    float B2r = (q >= 2) ? cB2r : oB2r;
    float B2i = (q >= 2) ? cB2i : oB2i;
    float B1r = (q >= 1) ? cB1r : oB1r;
    float B1i = (q >= 1) ? cB1i : oB1i;
    float B3p = (q >= 3) ? cB3p : oB3p;
    float B2p = (q >= 2) ? cB2p : oB2p;
    float B1p = (q >= 1) ? cB1p : oB1p;
    Ar = ((t3r + B2r) + B1r) + Hr;
    Ai = ((t3i + B2i) + B1i) + Hi;
    float P  = (((ps.Tp + B3p) + B2p) + B1p) + Hp; // tail of block m-4: same lane, previous tile
    float m2 = fma_(Ai, Ai, Ar * Ar);
    float tp = thr * P;
    bool above = (m2 > tp * tp) && (n < n_samp);
    ps = { Hr, Hi, Hp, Tr, Ti, Tp };
    return __ballot(above);
}

__device__ __forceinline__ uint64_t detect_tile(const float2* __restrict__ x, long n_samp, long n0, float thr,
                                                int lane, DetectState& ps, float& Ar, float& Ai)
{
    const long n = n0 + lane;
    return detect_tile_core(load_sample(x, n, n_samp), load_sample(x, n - 16, n_samp), n_samp, n0, thr, lane, ps, Ar, Ai);
}

// first sync_short trigger of each of the FOUR slots of a wave, in lock step (batch mode): row f = lanes 16f..16f+15 walks
// slot f in blocks of 16 samples, lane r of the row <-> sample 16m + r of block m.  The spec's 16-sample blocks (rule 3) are
// then exactly a DPP row: prefix H and suffix T are row scans, the block total B = H[15] is a row broadcast, and what a
// window needs of the three or four blocks before -- T of block m-3 (m-4 for the power window), B of blocks m-1 .. m-3 --
// are this lane's own values of earlier steps, carried in registers.  x[n-16] is the lane's sample of the step before.
// No cross-row traffic at all (the lane <-> sample form above fetched 18 values per 64-sample tile through ds_bpermute and
// walked the four slots one after the other).  Same sums in the same order: bit-identical A, P and triggers.
// The samples are requested four blocks (64 samples per slot) ahead.
struct DetectQuadState {
    c32   xd;                     // x[n - 16]
    float Tr[3], Ti[3], Tp[4];    // tails of blocks m-1, m-2, m-3 (, m-4) at this lane's r
    float Br[2], Bi[2], Bp[3];    // totals of blocks m-1, m-2 (, m-3)
};

#ifndef WR_DQ_GROUP
#define WR_DQ_GROUP 4             // blocks per request group
#endif

__device__ __forceinline__ void detect_quad_load(const float2* __restrict__ x, int n_samp, int m0, int r, c32 (&v)[WR_DQ_GROUP])
{
#pragma unroll
    for (int k = 0; k < WR_DQ_GROUP; k++) {
        const int n = 16 * (m0 + k) + r;
        float2 t = make_float2(0.0f, 0.0f);
        if (n < n_samp) t = x[n];
        v[k] = { t.x, t.y };
    }
}

// One block of all four slots.  Returns the ballot of c[n] > thr (bits 16f .. 16f+15: slot f's block); Ar/Ai = A[n] of this
// lane's sample.
__device__ __forceinline__ uint64_t detect_quad_step(c32 xn, int n, int n_samp, float thr, DetectQuadState& st, float& Ar, float& Ai)
{
    const c32 xd = st.xd;
    const float ar = fma_(xn.im, xd.im, xn.re * xd.re);
    const float ai = fma_(xn.im, xd.re, -(xn.re * xd.im));
    const float pw = fma_(xn.im, xn.im, xn.re * xn.re);
    const float Hr = row_prefix16(ar), Hi = row_prefix16(ai), Hp = row_prefix16(pw);
    const float Tr = dpp_zero<0x101>(row_suffix16(ar));
    const float Ti = dpp_zero<0x101>(row_suffix16(ai));
    const float Tp = dpp_zero<0x101>(row_suffix16(pw));
    Ar = ((st.Tr[2] + st.Br[1]) + st.Br[0]) + Hr;
    Ai = ((st.Ti[2] + st.Bi[1]) + st.Bi[0]) + Hi;
    const float P = (((st.Tp[3] + st.Bp[2]) + st.Bp[1]) + st.Bp[0]) + Hp;
    const float m2 = fma_(Ai, Ai, Ar * Ar);
    const float tp = thr * P;
    const bool above = (m2 > tp * tp) && (n < n_samp);
    st.xd = xn;
    st.Tr[2] = st.Tr[1]; st.Tr[1] = st.Tr[0]; st.Tr[0] = Tr;
    st.Ti[2] = st.Ti[1]; st.Ti[1] = st.Ti[0]; st.Ti[0] = Ti;
    st.Tp[3] = st.Tp[2]; st.Tp[2] = st.Tp[1]; st.Tp[1] = st.Tp[0]; st.Tp[0] = Tp;
    st.Br[1] = st.Br[0]; st.Br[0] = row_bcast<15>(Hr);
    st.Bi[1] = st.Bi[0]; st.Bi[0] = row_bcast<15>(Hi);
    st.Bp[2] = st.Bp[1]; st.Bp[1] = st.Bp[0]; st.Bp[0] = row_bcast<15>(Hp);
    return __ballot(above);
}

// x / n_samp: this lane's slot (uniform inside a row; n_samp = 0: no such slot).  t[f] = trigger of slot f or -1, A[f] = A[t[f]].
__device__ __forceinline__ void detect_quad(const float2* __restrict__ x, int n_samp, float thr, int min_plateau, int lane,
                                            int (&t)[4], c32 (&A)[4])
{
    const int r = lane & 15;
    DetectQuadState st = {};
    // per slot: the c > thr bits of the last four blocks, newest block in bits 48..63 (bit 48 + k = sample 16 m + k)
    uint64_t hist[4] = { 0, 0, 0, 0 };
    int ns[4];
#pragma unroll
    for (int f = 0; f < 4; f++) { t[f] = -1; A[f] = { 0.0f, 0.0f }; ns[f] = __builtin_amdgcn_readlane(n_samp, 16 * f); }
    const int n_max = max(max(ns[0], ns[1]), max(ns[2], ns[3]));
    c32 nx[WR_DQ_GROUP];
    detect_quad_load(x, n_samp, 0, r, nx);
    for (int m0 = 0; 16 * m0 < n_max; m0 += WR_DQ_GROUP) {
        c32 cx[WR_DQ_GROUP];
#pragma unroll
        for (int k = 0; k < WR_DQ_GROUP; k++) cx[k] = nx[k];
        detect_quad_load(x, n_samp, m0 + WR_DQ_GROUP, r, nx);
#pragma unroll
        for (int k = 0; k < WR_DQ_GROUP; k++) {
            const int m = m0 + k;
            float Ar, Ai;
            const uint64_t bal = detect_quad_step(cx[k], 16 * m + r, n_samp, thr, st, Ar, Ai);
            bool open = false;                     // a slot still searching with samples left
#pragma unroll
            for (int f = 0; f < 4; f++) {
                hist[f] = (hist[f] >> 16) | (((bal >> (16 * f)) & 0xffffull) << 48);
                if (t[f] < 0 && (hist[f] >> 48)) {                        // wave-uniform
                    uint64_t hit = hist[f];
                    for (int j = 1; j <= min_plateau; j++) hit &= hist[f] << j;       // min_plateau <= 32: three blocks of history suffice
                    hit >>= 48;
                    if (hit) {
                        const int l = __builtin_ctzll(hit);
                        t[f] = 16 * m + l;
                        A[f] = { bcast(Ar, 16 * f + l), bcast(Ai, 16 * f + l) };
                    }
                }
                open |= t[f] < 0 && 16 * (m + 1) < ns[f];
            }
            if (!open) return;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// batch kernel: one wave = 4 consecutive slots, WR_WAVES_PER_BLOCK waves per workgroup.
// Preamble phase per slot with the whole wave (lane = sample / lag), then the four frames walk their
// symbols together (wr_quad.h).
template <int EQ, bool HB, bool XK>
__global__ __launch_bounds__(64 * WR_WAVES_PER_BLOCK, EQ == WIFIRX_EQ_STA ? WR_DEMOD_WAVES_PER_SIMD_STA : WR_DEMOD_WAVES_PER_SIMD)
void demod_batch_kernel(const float2* __restrict__ iq, uint32_t slot_len, uint32_t n_slots,
                        DemodParams prm, DemodOut out, const uint64_t* __restrict__ slot_off)
{
    __shared__ __attribute__((aligned(16))) float lds[WR_WAVES_PER_BLOCK][WR_QLDS_FLOATS_EQ(EQ) > WR_QLDS_PRE_FLOATS ? WR_QLDS_FLOATS_EQ(EQ) : WR_QLDS_PRE_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t slot0 = (blockIdx.x * WR_WAVES_PER_BLOCK + wave) * 4;
    if (slot0 >= n_slots) return;
    QuadSeed seed = quad_seed_none();
    // preamble phase.  Detection of the four slots with the whole wave (lane = sample), their first tiles requested
    // together; then two slots at a time: coarse derotation of both into LDS (their loads in flight together), the LTS
    // correlation of both on the matrix cores, the peak search per slot.
    PreFrame pf[4];
    {
        const int row_ = lane >> 4;
        const float2* xrow = iq;                    // this lane's slot (row f = slot f)
        int nrow = 0;
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const uint32_t slot = slot0 + f;
            const bool has = slot < n_slots;                // wave-uniform
            // uniform slots, or slot k = samples [slot_off[k], slot_off[k+1]) of iq (wifirx_demod_batch_v)
            const uint32_t sk = has ? slot : slot0;
            const size_t off = slot_off ? (size_t)slot_off[sk] : (size_t)sk * slot_len;
            const long len = slot_off ? (long)(slot_off[sk + 1] - slot_off[sk]) : (long)slot_len;
            pf[f] = { iq + off, has ? len : 0l, -1, 0, 0.0f, false, has ? (long)slot : -1l };
            if (row_ == f) { xrow = pf[f].x; nrow = (int)pf[f].n_samp; }
        }
        c32 A4[4];
        int t4[4];
        detect_quad(xrow, nrow, prm.threshold, prm.min_plateau, lane, t4, A4);
#pragma unroll
        for (int f = 0; f < 4; f++) pf[f].t = t4[f];
        // coarse CFO of the four slots in ONE pass of the arctangent: row f of the wave works on slot f
        float cfo4;
        {
            const int row = lane >> 4;
            const float y = row == 0 ? A4[0].im : row == 1 ? A4[1].im : row == 2 ? A4[2].im : A4[3].im;
            const float x = row == 0 ? A4[0].re : row == 1 ? A4[1].re : row == 2 ? A4[2].re : A4[3].re;
            cfo4 = sp_atan2(y, x) / 16.0f;
        }
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int t = (int)pf[f].t;
            if (t >= 0) {
                pf[f].cfo_c = bcast(cfo4, 16 * f);
                long L = pf[f].n_samp - (t - 16);
                if (L > WIFIRX_MAX_SAMPLES) L = WIFIRX_MAX_SAMPLES;
                pf[f].L = L;
                pf[f].search = L >= WIFIRX_SYNC_LENGTH + 63;
            }
        }
    }
#if WR_ABLATE == 2   // timing experiment: detection (a1 + a2) only
    if (lane < 4) { wifirx_frame* frames = out.frames; const int f = lane; long o = f == 0 ? pf[0].out : f == 1 ? pf[1].out : f == 2 ? pf[2].out : pf[3].out;
                    int t = f == 0 ? pf[0].t : f == 1 ? pf[1].t : f == 2 ? pf[2].t : pf[3].t; float c = f == 0 ? pf[0].cfo_c : f == 1 ? pf[1].cfo_c : f == 2 ? pf[2].cfo_c : pf[3].cfo_c;
                    if (o >= 0) { frames[o].trigger = t; frames[o].cfo_coarse = c; } }
    return;
#endif
    PreSamples ps[4];
#pragma unroll
    for (int f = 0; f < 4; f++) preamble_load(pf[f], lane, ps[f]);
    // exp(-j float(cfo_c 64)) of the four frames in one pass: row f of the wave works on frame f
    c32 w64r;
    {
        const int row = lane >> 4;
        w64r = preamble_w64_rows(row == 0 ? pf[0].cfo_c : row == 1 ? pf[1].cfo_c : row == 2 ? pf[2].cfo_c : pf[3].cfo_c);
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
        preamble_derotate_pair(pf[2 * p], pf[2 * p + 1], ps[2 * p], ps[2 * p + 1], bcast(w64r, 32 * p), bcast(w64r, 32 * p + 16), lds[wave], lane);
        __builtin_amdgcn_wave_barrier();
        preamble_pair_finish(pf[2 * p], pf[2 * p + 1], p, lds[wave], lane, seed);
        __builtin_amdgcn_wave_barrier();
    }
#if WR_ABLATE == 1   // timing experiment: preamble phase only
    if ((lane & 15) == 0 && seed.out >= 0) { wifirx_frame* frames = out.frames; frames[seed.out].flags = seed.flags; frames[seed.out].frame_start = seed.fs; frames[seed.out].cfo_fine = seed.cfo_f; frames[seed.out].trigger = (int)seed.t; }
    return;
#endif
    frames_quad<EQ, HB, XK>(seed, prm, lds[wave], lane, out);
}

// one wave per four selected triggers of the stream
template <int EQ, bool HB, bool XK>
__global__ __launch_bounds__(64 * WR_WAVES_PER_BLOCK, EQ == WIFIRX_EQ_STA ? WR_DEMOD_WAVES_PER_SIMD_STA : WR_DEMOD_WAVES_PER_SIMD)
void demod_stream_kernel(const float2* __restrict__ x, long n_samp, const StreamTrig* __restrict__ trig,
                         uint32_t n_trig, DemodParams prm, const float2* __restrict__ A, DemodOut out)
{
    __shared__ __attribute__((aligned(16))) float lds[WR_WAVES_PER_BLOCK][WR_QLDS_FLOATS_EQ(EQ) > WR_QLDS_PRE_FLOATS ? WR_QLDS_FLOATS_EQ(EQ) : WR_QLDS_PRE_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t k0 = (blockIdx.x * WR_WAVES_PER_BLOCK + wave) * 4;
    if (k0 >= n_trig) return;
    QuadSeed seed = quad_seed_none();
    PreFrame pf[4];
    PreSamples ps[4];
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const uint32_t k = k0 + f;
        pf[f] = { x, 0, -1, 0, 0.0f, false, -1 };
        if (k < n_trig) {                               // wave-uniform
            const StreamTrig tg = trig[k];
            float cfo_c = tg.cfo;              // carried over from an earlier push of the same stream
            if (!tg.pad) {
                const float2 At = A[tg.pos];
                cfo_c = sp_atan2(At.y, At.x) / 16.0f;
            }
            pf[f].n_samp = n_samp; pf[f].t = tg.pos; pf[f].L = tg.usable; pf[f].cfo_c = cfo_c; pf[f].out = k;
            pf[f].search = tg.usable >= WIFIRX_SYNC_LENGTH + 63;
        }
        preamble_load(pf[f], lane, ps[f]);
    }
    c32 w64r;
    {
        const int row = lane >> 4;
        w64r = preamble_w64_rows(row == 0 ? pf[0].cfo_c : row == 1 ? pf[1].cfo_c : row == 2 ? pf[2].cfo_c : pf[3].cfo_c);
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
        preamble_derotate_pair(pf[2 * p], pf[2 * p + 1], ps[2 * p], ps[2 * p + 1], bcast(w64r, 32 * p), bcast(w64r, 32 * p + 16), lds[wave], lane);
        __builtin_amdgcn_wave_barrier();
        preamble_pair_finish(pf[2 * p], pf[2 * p + 1], p, lds[wave], lane, seed);
        __builtin_amdgcn_wave_barrier();
    }
    frames_quad<EQ, HB, XK>(seed, prm, lds[wave], lane, out);
}

}  // namespace wr
