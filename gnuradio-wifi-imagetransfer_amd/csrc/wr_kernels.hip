// wr_kernels.hip -- HIP kernels of the wifirx receive chain for gfx950 (MI355X, CDNA4).
//
// One wavefront (64 lanes) owns one frame: lane <-> sample while scanning for the short preamble
// and correlating against the long training symbol, lane <-> sub-carrier from the FFT on.  The
// per-frame state of the reference's frame_equalizer (d_er, previous pilots, channel estimate H)
// lives in registers for the whole frame, so the symbols of a frame are walked in order by the
// same wave and thousands of frames run side by side.  The whole chain is fused: the samples of
// a slot are read from HBM once, the only HBM writes are the decisions / LLRs / frame record.
//
// Replaces, per the reference's flowgraph (gnu_radio/IRS_AP.py:268-285,294-311):
//   a1  delay(16), conjugate_cc, multiply_vcc, moving_average_cc(48), complex_to_mag_squared,
//       moving_average_ff(64), complex_to_mag, divide_ff            -> detect phase
//   a2  ieee802_11.sync_short(0.56, 2)                              -> detect phase + derotation
//   a3  delay(320) + ieee802_11.sync_long(320)                      -> lts phase + derotation
//   a4  stream_to_vector(64) + fft_vcc(64, forward, rect, shift)    -> fft64
//   a5  ieee802_11.frame_equalizer(LS, freq, bw)                    -> symbol loop
//   a6  constellation decision makers, a7 LLR (north-star addition)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wifirx.h"
#include "wr_device.h"
#include "wr_kernels.h"

namespace wr {

// per-lane constants that do not depend on the frame
struct LaneConst {
    c32   tw1, tw2;     // FFT twiddles of stage 1 and 2 for the output this lane computes
    int   src_shift;    // lane that holds, after the 3 stages, the bin this lane wants (fftshift)
    int   carrier;      // data carrier number 0..47 of bin `lane`, -1 for pilots / DC / guards
    float lts;          // L_i of the long training symbol (+-1, 0)
    bool  used;         // one of the 52 occupied bins
};

__device__ __forceinline__ LaneConst lane_const(int lane)
{
    LaneConst k;
    {
        int n = lane & 15, q = lane >> 4;
        int e = (q * n) & 63;
        k.tw1 = { WR_TWIDDLE64[2 * e], WR_TWIDDLE64[2 * e + 1] };
    }
    {
        int n = lane & 3, q = (lane >> 2) & 3;
        int e = (q * n * 4) & 63;
        k.tw2 = { WR_TWIDDLE64[2 * e], WR_TWIDDLE64[2 * e + 1] };
    }
    {
        int kk = (lane + 32) & 63;                    // sub-carrier index (0..63) wanted by this lane
        k.src_shift = 16 * (kk & 3) + 4 * ((kk >> 2) & 3) + (kk >> 4);
    }
    int i = lane;
    bool data = (i >= 6 && i <= 58 && i != 11 && i != 25 && i != 32 && i != 39 && i != 53);
    k.carrier = data ? (i - 6 - (i > 11) - (i > 25) - (i > 32) - (i > 39) - (i > 53)) : -1;
    k.lts = WR_LTS_FREQ[i];
    k.used = (i >= 6 && i <= 58 && i != 32);
    return k;
}

// radix-4 DIF butterfly output q of inputs x0..x3 (spec section 4.3)
__device__ __forceinline__ c32 bfly4(c32 x0, c32 x1, c32 x2, c32 x3, int q)
{
    bool odd = q & 1;
    c32 x2s = odd ? cneg(x2) : x2;
    c32 x3s = odd ? cneg(x3) : x3;
    c32 tac = cadd(x0, x2s);          // q even: x0+x2, q odd: x0-x2
    c32 tbd = cadd(x1, x3s);          // q even: x1+x3, q odd: x1-x3
    c32 u;
    if (q == 0)      u = tbd;
    else if (q == 2) u = cneg(tbd);
    else if (q == 1) u = { tbd.im, -tbd.re };    // -j * tbd
    else             u = { -tbd.im, tbd.re };    // +j * tbd
    return cadd(tac, u);
}

// 64-point forward DFT of one value per lane, output bin i (sub-carrier i-32) on lane i
__device__ __forceinline__ c32 fft64_wave(c32 v, int lane, const LaneConst& k)
{
    {   // stage 1: span 16
        int n = lane & 15, q = lane >> 4;
        c32 x0 = shfl(v, n), x1 = shfl(v, n + 16), x2 = shfl(v, n + 32), x3 = shfl(v, n + 48);
        v = sp_cmul(bfly4(x0, x1, x2, x3, q), k.tw1);
    }
    {   // stage 2: span 4 inside blocks of 16
        int base = lane & 48, n = lane & 3, q = (lane >> 2) & 3;
        c32 x0 = shfl(v, base + n), x1 = shfl(v, base + n + 4), x2 = shfl(v, base + n + 8),
            x3 = shfl(v, base + n + 12);
        v = sp_cmul(bfly4(x0, x1, x2, x3, q), k.tw2);
    }
    {   // stage 3: span 1 inside blocks of 4 (twiddle 1: the spec's multiply by (1,0) is exact)
        int base = lane & 60, q = lane & 3;
        c32 x0 = shfl(v, base), x1 = shfl(v, base + 1), x2 = shfl(v, base + 2), x3 = shfl(v, base + 3);
        v = bfly4(x0, x1, x2, x3, q);
    }
    return shfl(v, k.src_shift);
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

// Viterbi over the 24 SIGNAL bits, lane <-> state.  cbits: the 48 de-interleaved hard decisions
// (bit j = coded bit j).  Returns the 24 decoded bits (bit t = decoded bit t), wave-uniform.
__device__ __forceinline__ uint32_t viterbi_signal(uint64_t cbits, int lane)
{
    const int s = lane, u = s & 1, p0 = s >> 1, p1 = (s >> 1) | 32;
    const int f0 = (p0 << 1) | u, f1 = (p1 << 1) | u;
    const int a0 = __builtin_popcount(f0 & 0155) & 1, b0 = __builtin_popcount(f0 & 0117) & 1;
    const int a1 = __builtin_popcount(f1 & 0155) & 1, b1 = __builtin_popcount(f1 & 0117) & 1;
    int pm = (s == 0) ? 0 : (1 << 28);
    uint64_t dec[24];
#pragma unroll
    for (int t = 0; t < 24; t++) {
        int ra = (int)((cbits >> (2 * t)) & 1), rb = (int)((cbits >> (2 * t + 1)) & 1);
        int m0 = __shfl(pm, p0, 64) + (ra != a0) + (rb != b0);
        int m1 = __shfl(pm, p1, 64) + (ra != a1) + (rb != b1);
        bool sel = m1 < m0;
        pm = sel ? m1 : m0;
        dec[t] = __ballot(sel);
    }
    // best final state: smallest metric, lowest index on ties
    int key = (pm << 6) | s;      // pm < 2^28/… : metrics stay far below 2^25 here
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        int o = __shfl_xor(key, k, 64);
        key = o < key ? o : key;
    }
    int st = key & 63;
    uint32_t bits = 0;
#pragma unroll
    for (int t = 23; t >= 0; t--) {
        bits |= (uint32_t)(st & 1) << t;
        int h = (int)((dec[t] >> st) & 1);
        st = (st >> 1) | (h << 5);
    }
    return bits;
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
// detect phase (a1 + a2).  One tile = 64 consecutive samples, lane <-> sample; the window sums
// follow the blocked scheme of the spec (section 4.2): Kogge-Stone prefix H / exclusive suffix T inside
// blocks of 16 lanes (DPP row shifts), block totals and tails of the 3-4 previous blocks.
struct DetectState {       // what a tile needs from the tile before it
    float Hr, Hi, Hp, Tr, Ti, Tp;
};

__device__ __forceinline__ DetectState detect_state_zero() { return { 0, 0, 0, 0, 0, 0 }; }

// Processes tile [n0, n0+64).  Returns the ballot of c[n] > thr; Ar/Ai = A[n] of this lane's sample.
__device__ __forceinline__ uint64_t detect_tile(const float2* __restrict__ x, long n_samp, long n0, float thr,
                                                int lane, DetectState& ps, float& Ar, float& Ai)
{
    const int q = lane >> 4;
    const int l16 = (lane + 16) & 63;              // lane that holds index lane-48 (mod 64)
    const int b1 = (((q - 1) & 3) << 4) | 15;      // last lane of block m-1 / m-2 / m-3
    const int b2 = (((q - 2) & 3) << 4) | 15;
    const int b3 = (((q - 3) & 3) << 4) | 15;
    long n = n0 + lane;
    c32 xn = load_sample(x, n, n_samp);
    c32 xd = load_sample(x, n - 16, n_samp);
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

// positions where c > thr held for min_plateau+1 consecutive samples ending there
__device__ __forceinline__ uint64_t plateau_hits(uint64_t mask, uint64_t prev_mask, int min_plateau)
{
    uint64_t hit = mask;
    for (int j = 1; j <= min_plateau; j++) hit &= (mask << j) | (prev_mask >> (64 - j));
    return hit;
}

// first sync_short trigger of a slot.  Returns the trigger index or -1; A_t = A[trigger].
__device__ __forceinline__ int detect_first(const float2* __restrict__ x, long n_samp, float thr,
                                            int min_plateau, int lane, c32& A_t)
{
    DetectState ps = detect_state_zero();
    uint64_t prev_mask = 0;
    for (long n0 = 0; n0 < n_samp; n0 += 64) {
        float Ar, Ai;
        uint64_t mask = detect_tile(x, n_samp, n0, thr, lane, ps, Ar, Ai);
        uint64_t hit = plateau_hits(mask, prev_mask, min_plateau);
        if (hit) {
            int l = __builtin_ctzll(hit);
            A_t = { bcast(Ar, l), bcast(Ai, l) };
            return (int)(n0 + l);
        }
        prev_mask = mask;
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------
// Everything after the trigger, for one frame owned by this wave.
//   x: slot / stream base, n_samp: its length, t: trigger, cfo_c: coarse CFO, L: usable copied samples
//   ylds: wave-private LDS scratch of WR_YLDS_FLOATS floats
__device__ __forceinline__ void frame_body(const float2* __restrict__ x, long n_samp, long t, float cfo_c,
                                           long L, const DemodParams& prm, float* ylds, int lane,
                                           wifirx_frame* fr_out, uint8_t* __restrict__ idx,
                                           float* __restrict__ llr, float2* __restrict__ carrier)
{
    wifirx_frame fr;
    fr.flags = WIFIRX_F_DETECTED;
    fr.trigger = (int32_t)t;
    fr.frame_start = 0;
    fr.cfo_coarse = cfo_c;
    fr.cfo_fine = 0.0f;
    fr.snr_db = 0.0f;
    fr.psdu_len = 0; fr.encoding = 0; fr.n_bpsc = 0; fr.n_sym = 0; fr.n_sym_out = 0;

    const LaneConst K = lane_const(lane);
    bool alive = (L >= WIFIRX_SYNC_LENGTH + 63);
    if (!alive) fr.flags |= WIFIRX_F_TRUNCATED;
    int   fs = 0;
    float cfo_f = 0.0f;

    if (alive) {
        // -- sync_short copy of the first 383 samples into LDS (6 x 64 >= 383) --
#pragma unroll
        for (int pass = 0; pass < 6; pass++) {
            int m = pass * 64 + lane;
            c32 xs = load_sample(x, t - 16 + m, n_samp);
            float s, c;
            sp_sincos(-cfo_c * (float)m, s, c);
            c32 y = sp_rot(xs, s, c);
            ylds[2 * m] = y.re;
            ylds[2 * m + 1] = y.im;
        }
        __builtin_amdgcn_wave_barrier();
        // -- sync_long: 64-tap LTS correlation over 320 lags, lane <-> lag --
        c32   corr[5];
        float mag[5];
#pragma unroll
        for (int pass = 0; pass < 5; pass++) {
            int i = pass * 64 + lane;
            float ar = 0.0f, ai = 0.0f;
#pragma unroll 8
            for (int k = 0; k < 64; k++) {
                float lr = WR_LTS_TIME[2 * k], li = WR_LTS_TIME[2 * k + 1];
                float yr = ylds[2 * (i + k)], yi = ylds[2 * (i + k) + 1];
                ar = fma_(lr, yr, ar);
                ar = fma_(li, yi, ar);
                ai = fma_(lr, yi, ai);
                ai = fma_(-li, yr, ai);
            }
            corr[pass] = { ar, ai };
            mag[pass] = fma_(ai, ai, ar * ar);
        }
        // top 4 magnitudes, ties -> lower lag
        int   top_off[4];
        c32   top_val[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // local best of this lane's 5 candidates (lower pass = lower lag wins ties)
            unsigned long long key = 0;
#pragma unroll
            for (int pass = 0; pass < 5; pass++) {
                int i = pass * 64 + lane;
                unsigned long long kk = ((unsigned long long)__float_as_uint(mag[pass]) << 32) |
                                        (unsigned)(0xffffffffu - (unsigned)i);
                bool valid = mag[pass] >= 0.0f;
                if (valid && kk > key) key = kk;
            }
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) {
                unsigned long long o = __shfl_xor(key, k, 64);
                key = o > key ? o : key;
            }
            int w = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
            int wl = w & 63, wp = w >> 6;
            c32 val = { 0.0f, 0.0f };
#pragma unroll
            for (int pass = 0; pass < 5; pass++) {
                if (pass == wp) {
                    val = bcast(corr[pass], wl);
                    if (lane == wl) mag[pass] = -1.0f;
                }
            }
            top_off[r] = w;
            top_val[r] = val;
        }
        int found = 0;
        fs = WIFIRX_SYNC_LENGTH;
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int k = i + 1; k < 4; k++) {
                if (found == 64) continue;
                int oi = top_off[i], ok = top_off[k];
                c32 first = oi > ok ? top_val[k] : top_val[i];
                c32 second = oi > ok ? top_val[i] : top_val[k];
                int diff = oi > ok ? oi - ok : ok - oi;
                if (diff == 64 || diff == 63 || diff == 65) {
                    float pr = fma_(first.im, second.im, first.re * second.re);
                    float pi = fma_(first.im, second.re, -(first.re * second.im));
                    fs = oi < ok ? oi : ok;
                    cfo_f = sp_atan2(pi, pr) / (float)diff;
                    found = diff;
                }
            }
        }
        alive = found != 0;
        if (alive) {
            fr.flags |= WIFIRX_F_SYNC;
            fr.frame_start = fs;
            fr.cfo_fine = cfo_f;
        }
    }

    if (alive) {
        const double bw = prm.bandwidth, fc = prm.frequency;
        const double two_pi = 2 * 3.14159265358979323846;
        const double tag = (double)cfo_c - (double)cfo_f;
        const double eps0 = tag * bw / (two_pi * fc);
        const double er_scale = bw / (two_pi * fc * 80);
        double d_er = 0.0;
        c32 prev0 = { 0, 0 }, prev1 = { 0, 0 }, prev2 = { 0, 0 }, prev3 = { 0, 0 };
        c32 H = { 0, 0 };
        int n_sym = 0, n_bpsc = 1, n_out = 0;
        bool have_signal = false, want_llr = false;

        for (int s = 0; s <= n_sym + 2; s++) {
            long off0 = fs + (s < 2 ? 64 * s : 128 + 80 * (s - 2) + 16);
            if (off0 + 64 > L || (s > 2 && (s - 3) >= (int)prm.max_sym)) { fr.flags |= WIFIRX_F_TRUNCATED; break; }
            long m = off0 + lane;
            c32 xs = load_sample(x, t - 16 + m, n_samp);
            float s1, c1, s2, c2;
            sp_sincos(-cfo_c * (float)m, s1, c1);
            sp_sincos((float)m * cfo_f, s2, c2);
            c32 X = fft64_wave(sp_rot(sp_rot(xs, s1, c1), s2, c2), lane, K);

            // (1) sampling offset
            {
                double t4 = two_pi * s * 80 * (eps0 + d_er);
                float ang = (float)(t4 * (double)(lane - 32) / 64);
                float sn, cs;
                sp_sincos(ang, sn, cs);
                X = sp_rot(X, sn, cs);
            }
            // (2) pilots
            c32 X11 = bcast(X, 11), X25 = bcast(X, 25), X39 = bcast(X, 39), X53 = bcast(X, 53);
            bool pneg = (s >= 2) && (WR_POLARITY[(s - 2) % 127] < 0);
            c32 S;
            if (s < 2) S = cadd(cadd(csub(X11, X25), X39), X53);
            else {
                S = csub(cadd(cadd(X11, X39), X25), X53);
                if (pneg) S = cneg(S);
            }
            float beta = sp_atan2(S.im, S.re);
            // (3) residual offset estimate
            c32 cur0, cur1, cur2, cur3;
            if (s < 2) { cur0 = X11; cur1 = cneg(X25); cur2 = X39; cur3 = X53; }
            else {
                cur0 = pneg ? cneg(X11) : X11;
                cur1 = pneg ? cneg(X25) : X25;
                cur2 = pneg ? cneg(X39) : X39;
                cur3 = pneg ? X53 : cneg(X53);
            }
            double er = 0.0;
            if (s >= 2) {
                c32 acc = cadd(cadd(cadd(sp_conj_mul(prev0, cur0), sp_conj_mul(prev1, cur1)),
                                    sp_conj_mul(prev2, cur2)), sp_conj_mul(prev3, cur3));
                er = (double)sp_atan2(acc.im, acc.re) * er_scale;
            }
            prev0 = cur0; prev1 = cur1; prev2 = cur2; prev3 = cur3;
            // (4) common phase
            {
                float sn, cs;
                sp_sincos(-beta, sn, cs);
                X = sp_rot(X, sn, cs);
            }
            // (5) IIR
            if (s >= 2) {
                double alpha = 0.1;
                d_er = (1 - alpha) * d_er + alpha * er;
            }
            // (6) LS equalizer
            if (s == 0) {
                H = X;
            } else if (s == 1) {
                c32 d = csub(H, X), u = cadd(H, X);
                float nv = K.used ? fma_(d.im, d.im, d.re * d.re) : 0.0f;
                float sv = K.used ? fma_(u.im, u.im, u.re * u.re) : 0.0f;
                float g = 0.5f * K.lts;
                if (K.used) { H.re = u.re * g; H.im = u.im * g; }
                float signal = tree_sum64(sv), noise = tree_sum64(nv);
                fr.snr_db = sp_snr_db(signal, noise);
            } else {
                int  nb = (s == 2) ? 1 : n_bpsc;
                bool is_data = K.carrier >= 0;
                c32  Y = { 0.0f, 0.0f };
                if (is_data) {
                    float d = fma_(H.im, H.im, H.re * H.re);
                    Y.re = fma_(X.im, H.im, X.re * H.re) / d;
                    Y.im = fma_(X.im, H.re, -(X.re * H.im)) / d;
                }
                uint8_t bits = decide(Y, nb);
                if (s == 2) {
                    // (7) SIGNAL: gather the 48 decisions in carrier order, de-interleave, Viterbi
                    uint64_t b = __ballot(is_data && (bits & 1));
                    uint64_t cm = ((b >> 6) & 0x1full) | (((b >> 12) & 0x1fffull) << 5) |
                                  (((b >> 26) & 0x3full) << 18) | (((b >> 33) & 0x3full) << 24) |
                                  (((b >> 40) & 0x1fffull) << 30) | (((b >> 54) & 0x1full) << 43);
                    uint64_t de = 0;
#pragma unroll
                    for (int j = 0; j < 48; j++) de |= ((cm >> (3 * (j % 16) + j / 16)) & 1ull) << j;
                    uint32_t sig = viterbi_signal(de, lane);
                    int enc = 0, len = 0;
                    if (!parse_signal(sig, enc, len)) break;
                    const int nbpsc_tab[8] = { 1, 1, 2, 2, 4, 4, 6, 6 };
                    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
                    have_signal = true;
                    n_bpsc = nbpsc_tab[enc];
                    int nd = ndbps_tab[enc];
                    n_sym = (16 + 8 * len + 6 + nd - 1) / nd;
                    fr.flags |= WIFIRX_F_SIGNAL;
                    fr.psdu_len = (uint16_t)len;
                    fr.encoding = (uint8_t)enc;
                    fr.n_bpsc = (uint8_t)n_bpsc;
                    fr.n_sym = (uint16_t)n_sym;
                    want_llr = (llr != nullptr) && ((int)prm.llr_bits >= n_bpsc);
                    if (want_llr) fr.flags |= WIFIRX_F_LLR;
                } else {
                    // (8) data symbol q
                    int q = s - 3;
                    if (is_data) {
                        size_t o = (size_t)q * 48 + K.carrier;
                        if (idx) idx[o] = bits;
                        if (carrier) carrier[o] = make_float2(Y.re, Y.im);
                        if (want_llr) {
                            float are = __builtin_fabsf(Y.re), aim = __builtin_fabsf(Y.im);
                            float* lp = llr + o * n_bpsc;
                            if (n_bpsc == 1) {
                                lp[0] = Y.re;
                            } else if (n_bpsc == 2) {
                                *reinterpret_cast<float2*>(lp) = make_float2(Y.re, Y.im);
                            } else if (n_bpsc == 4) {
                                *reinterpret_cast<float4*>(lp) = make_float4(Y.re, WR_T16_2 - are, Y.im, WR_T16_2 - aim);
                            } else {
                                float2* l2 = reinterpret_cast<float2*>(lp);
                                l2[0] = make_float2(Y.re, WR_T64_4 - are);
                                l2[1] = make_float2(WR_T64_2 - __builtin_fabsf(are - WR_T64_4), Y.im);
                                l2[2] = make_float2(WR_T64_4 - aim, WR_T64_2 - __builtin_fabsf(aim - WR_T64_4));
                            }
                        }
                    }
                    n_out = q + 1;
                }
            }
        }
        fr.n_sym_out = (uint16_t)n_out;
        if (have_signal && n_out == n_sym) fr.flags |= WIFIRX_F_COMPLETE;
    }
    if (lane == 0) *fr_out = fr;
}

// ---------------------------------------------------------------------------------------------
// batch kernel: one wave per slot, WR_WAVES_PER_BLOCK slots per workgroup
__global__ __launch_bounds__(64 * WR_WAVES_PER_BLOCK)
void demod_batch_kernel(const float2* __restrict__ iq, uint32_t slot_len, uint32_t n_slots,
                        DemodParams prm, wifirx_frame* __restrict__ frames, uint8_t* __restrict__ idx,
                        float* __restrict__ llr, float2* __restrict__ carrier)
{
    __shared__ float lds[WR_WAVES_PER_BLOCK][WR_YLDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * WR_WAVES_PER_BLOCK + wave;
    if (slot >= n_slots) return;
    const float2* x = iq + (size_t)slot * slot_len;
    const size_t idx_stride = (size_t)prm.max_sym * 48;

    c32 A_t = { 0, 0 };
    int t = detect_first(x, slot_len, prm.threshold, prm.min_plateau, lane, A_t);
    if (t < 0) {
        if (lane == 0) {
            wifirx_frame fr;
            fr.flags = 0; fr.trigger = -1; fr.frame_start = 0; fr.cfo_coarse = 0; fr.cfo_fine = 0;
            fr.snr_db = 0; fr.psdu_len = 0; fr.encoding = 0; fr.n_bpsc = 0; fr.n_sym = 0; fr.n_sym_out = 0;
            frames[slot] = fr;
        }
        return;
    }
    float cfo_c = sp_atan2(A_t.im, A_t.re) / 16.0f;
    long L = (long)slot_len - (t - 16);
    if (L > WIFIRX_MAX_SAMPLES) L = WIFIRX_MAX_SAMPLES;
    frame_body(x, slot_len, t, cfo_c, L, prm, lds[wave], lane, frames + slot,
               idx ? idx + slot * idx_stride : nullptr,
               llr ? llr + slot * idx_stride * prm.llr_bits : nullptr,
               carrier ? carrier + slot * idx_stride : nullptr);
}

// ---------------------------------------------------------------------------------------------
// stream mode (the GNU Radio block's work()): detection over a span of the stream buffer.
// Every wave owns WR_STREAM_SPAN tiles; it first replays the tile in front of its span to get the
// block sums that tile hands over (tile -1 of the buffer is the all-zero past of the stream start).
__global__ __launch_bounds__(256)
void stream_detect_kernel(const float2* __restrict__ x, long n_samp, long tile0, long n_tiles, float thr,
                          uint64_t* __restrict__ masks, float2* __restrict__ A)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long first = tile0 + wave * WR_STREAM_SPAN;
    if (first >= tile0 + n_tiles) return;
    long last = first + WR_STREAM_SPAN;
    if (last > tile0 + n_tiles) last = tile0 + n_tiles;
    DetectState ps = detect_state_zero();
    float Ar, Ai;
    if (first > 0) (void)detect_tile(x, n_samp, (first - 1) * 64, thr, lane, ps, Ar, Ai);
    for (long tl = first; tl < last; tl++) {
        uint64_t mask = detect_tile(x, n_samp, tl * 64, thr, lane, ps, Ar, Ai);
        long n = tl * 64 + lane;
        if (n < n_samp) A[n] = make_float2(Ar, Ai);
        if (lane == 0) masks[tl] = mask;
    }
}

// one wave per selected trigger of the stream
__global__ __launch_bounds__(64 * WR_WAVES_PER_BLOCK)
void demod_stream_kernel(const float2* __restrict__ x, long n_samp, const StreamTrig* __restrict__ trig,
                         uint32_t n_trig, DemodParams prm, const float2* __restrict__ A,
                         wifirx_frame* __restrict__ frames, uint8_t* __restrict__ idx,
                         float* __restrict__ llr, float2* __restrict__ carrier)
{
    __shared__ float lds[WR_WAVES_PER_BLOCK][WR_YLDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t k = blockIdx.x * WR_WAVES_PER_BLOCK + wave;
    if (k >= n_trig) return;
    const StreamTrig tg = trig[k];
    float cfo_c = tg.cfo;                 // carried over from an earlier push of the same stream
    if (!tg.pad) {
        const float2 At = A[tg.pos];
        cfo_c = sp_atan2(At.y, At.x) / 16.0f;
    }
    const size_t idx_stride = (size_t)prm.max_sym * 48;
    frame_body(x, n_samp, tg.pos, cfo_c, tg.usable, prm, lds[wave], lane, frames + k,
               idx ? idx + k * idx_stride : nullptr,
               llr ? llr + k * idx_stride * prm.llr_bits : nullptr,
               carrier ? carrier + k * idx_stride : nullptr);
}

}  // namespace wr

extern "C" hipError_t wr_launch_demod_batch(hipStream_t st, const float2* iq, uint32_t slot_len,
                                            uint32_t n_slots, const wr::DemodParams* prm,
                                            wifirx_frame* frames, uint8_t* idx, float* llr, float2* carrier)
{
    if (n_slots == 0) return hipSuccess;
    dim3 grid((n_slots + WR_WAVES_PER_BLOCK - 1) / WR_WAVES_PER_BLOCK), block(64 * WR_WAVES_PER_BLOCK);
    hipLaunchKernelGGL(wr::demod_batch_kernel, grid, block, 0, st, iq, slot_len, n_slots, *prm, frames, idx, llr, carrier);
    return hipGetLastError();
}

extern "C" hipError_t wr_launch_stream_detect(hipStream_t st, const float2* x, int64_t n_samp, int64_t tile0,
                                              int64_t n_tiles, float thr, uint64_t* masks, float2* A)
{
    if (n_tiles <= 0) return hipSuccess;
    int64_t waves = (n_tiles + WR_STREAM_SPAN - 1) / WR_STREAM_SPAN;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    hipLaunchKernelGGL(wr::stream_detect_kernel, grid, block, 0, st, x, (long)n_samp, (long)tile0, (long)n_tiles, thr, masks, A);
    return hipGetLastError();
}

extern "C" hipError_t wr_launch_demod_stream(hipStream_t st, const float2* x, int64_t n_samp, const wr::StreamTrig* trig,
                                             uint32_t n_trig, const wr::DemodParams* prm, const float2* A,
                                             wifirx_frame* frames, uint8_t* idx, float* llr, float2* carrier)
{
    if (n_trig == 0) return hipSuccess;
    dim3 grid((n_trig + WR_WAVES_PER_BLOCK - 1) / WR_WAVES_PER_BLOCK), block(64 * WR_WAVES_PER_BLOCK);
    hipLaunchKernelGGL(wr::demod_stream_kernel, grid, block, 0, st, x, (long)n_samp, trig, n_trig, *prm, A, frames, idx, llr, carrier);
    return hipGetLastError();
}
