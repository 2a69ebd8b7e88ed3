/*
 * wifirx_oracle.c -- CPU restatement of the IEEE 802.11a/g OFDM receive chain of the reference.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (libwifirx.so, the wifirx Python package) may
 * import, link or execute this file; only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py do, and only as the checker / the timed CPU baseline ("port").
 *
 * PARITY UNPINNED.  The reference (OedonLestrange42/GNURadio-WiFI-ImageTransfer) holds no PHY
 * arithmetic, tests or golden vectors of its own: the receive chain it wires together in
 * gnu_radio/IRS_AP.py:267-311 and gnu_radio/wifi_phy_hier.grc:100-260,480-569,698-768 lives in
 * the un-vendored, un-pinned third-party modules gr-ieee802-11 (maint-3.10 line) and GNU Radio
 * 3.10.7.0 (gnu_radio/IRS_AP.py:10), neither of which exists in this image.  This file restates
 * the published algorithms of those blocks from the reference's own call
 * sites and parameters; it is pinned only by (a) the constants the reference carries in
 * wifi_phy_hier.grc:346-398 (tests/golden/grc_constants.json), (b) IEEE 802.11 known answers
 * (Annex example FCS / CRC residue, SIGNAL bits, 127-bit scrambler sequence) and (c) loop-back
 * through the independent NumPy transmitter wifirx/txgen.py.
 *
 * Two arithmetic modes:
 *   ORC_MATH_SPEC (0)  the wifirx numerics spec: IEEE-754 binary32 operations in a
 *                      fixed order, fmaf where written, own sincos/atan2/log2 polynomials, blocked
 *                      window sums, radix-4 DIF FFT.  The HIP kernels follow the same spec, so
 *                      their outputs must equal this mode bit for bit.
 *   ORC_MATH_LIBM (1)  the upstream blocks' arithmetic: libm sincosf/atan2f/log10, C99 complex
 *                      division, running sums in double, O(N^2) DFT in double.  Used by the tests
 *                      to bound the distance between the spec and an upstream-style evaluation.
 *
 * Compile with -ffp-contract=off (see oracle/Makefile): a contracted a*b+c would break the spec.
 */
#define _GNU_SOURCE
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "wifirx.h"
#define WR_WANT_T4_TABLE
#include "wifirx_tables.h"

#define ORC_MATH_SPEC 0
#define ORC_MATH_LIBM 1

typedef struct { float re, im; } c32;

typedef struct orc_params {
    double  bandwidth;
    double  frequency;
    float   threshold;
    int32_t min_plateau;
    int32_t math_mode;
    int32_t max_sym;      /* output capacity per frame (data symbols) */
    int32_t llr_bits;     /* 0: no llr */
    int32_t chan_est;     /* WIFIRX_EQ_LS / LMS / COMB / STA */
    int32_t llr_csi;      /* 1: every LLR is multiplied by |H|^2 of its sub-carrier (LS estimate of the preamble) */
    int32_t lts_search;   /* SPEC mode only.  0: rule 6 as the kernels run it (candidates from the 8-bit search);
                           * 1: the float32 values of ALL 320 lags, the four largest of those -- sync_long's exhaustive
                           *    search in the spec's arithmetic.  The tests compare the two (tests/test_lts_rule6.py). */
    int32_t no_pair_fallback;  /* What sync_long does when no two of the four strongest lags are 64 / 63 / 65 apart.  0 (the build's
                           * semantics): the frame is dropped.  1: upstream's -- d_frame_start keeps its
                           * default, sync_length = 320, and d_freq_offset keeps the PREVIOUS frame's value (SURVEY App. A.3;
                           * call site gnu_radio/IRS_AP.py:269,309): orc_demod_batch runs such frames again, in slot order,
                           * with frame start 320 and the fine CFO of the last frame before them that found a pair (0 at
                           * the start).  A measurement switch (tests/campaigns/lts_no_pair.py), not a product mode.
                           * 2 (internal): this call IS such a re-run, fallback_cfo_f holds the value. */
    int32_t* dbg_top4;    /* test hook of orc_frame (single-threaded callers only): the four lags handed to the pair search */
    float*   dbg_mag4;    /* ... and their magnitudes (|corr|^2 in SPEC mode, |corr| in LIBM mode); NULL: off */
    float    fallback_cfo_f;
    int32_t  libm_exact_phase;  /* LIBM mode only, a measurement switch (tests/test_gpu_configs.py).  0: upstream's literal derotation --
                           * the angles -cfo_c * m and cfo_f * m are float32 PRODUCTS before sincosf sees them (sync_short /
                           * sync_long: `exp(gr_complex(0, -d_freq_offset * d_copied))`): at m = 4000 and 0.037 rad / sample the
                           * angle is 150 rad and its ulp 1.5e-5 rad.  1: the same two rotations with the angles formed and reduced
                           * in double (sin / cos in double, rounded to float once): what is left of the distance to the GPU is then
                           * everything BUT upstream's angle rounding. */
} orc_params;

/* ------------------------------------------------------------------------------------------- */
/* spec math                                                                 */

static inline void sp_sincos(float x, float* s, float* c)
{
    float kf = rintf(x * WR_TWO_OVER_PI);
    int   k  = (int)kf;
    float r  = fmaf(-kf, WR_PIO2_HI, x);
    r = fmaf(-kf, WR_PIO2_MID, r);
    r = fmaf(-kf, WR_PIO2_LO, r);
    float z  = r * r;
    float ps = fmaf(z, WR_S3, WR_S2);
    ps = fmaf(ps, z, WR_S1);
    float sr = fmaf(ps * z, r, r);
    float pc = fmaf(z, WR_C3, WR_C2);
    pc = fmaf(pc, z, WR_C1);
    float cr = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    switch (k & 3) {
    case 0:  *s = sr;  *c = cr;  break;
    case 1:  *s = cr;  *c = -sr; break;
    case 2:  *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr;  break;
    }
}

/* sine/cosine of a phase given in 2^-62 quarter turns, modulo 2^64 (spec section 4.8): the integer phase of sample
 * m is Q*m with wrap-around, so the reduction is a shift and the quadrant the two top bits. */
static inline void sp_sincos_q(uint64_t P, float* s, float* c)
{
    uint32_t th = (uint32_t)(P >> 32) + 0x20000000u, tl = (uint32_t)P;     /* + half a quarter turn: round to nearest */
    unsigned k  = th >> 30;
    uint32_t uh = (th << 2) | (tl >> 30);                                   /* fraction of the quarter turn, offset 1/2 */
    int32_t  gh = (int32_t)(uh ^ 0x80000000u);                              /* [-2^31, 2^31) <-> [-1/2, 1/2) quarter turn */
    float r  = (float)gh * WR_PIO2_2M32;
    float z  = r * r;
    float ps = fmaf(z, WR_S3, WR_S2);
    ps = fmaf(ps, z, WR_S1);
    float sr = fmaf(ps * z, r, r);
    float pc = fmaf(z, WR_C3, WR_C2);
    pc = fmaf(pc, z, WR_C1);
    float cr = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    switch (k & 3) {
    case 0:  *s = sr;  *c = cr;  break;
    case 1:  *s = cr;  *c = -sr; break;
    case 2:  *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr;  break;
    }
}

/* 1/sqrt(v) for normal positive v: seed by halving the exponent, three Newton steps y <- y (1.5 - 0.5 v y^2) */
static inline float sp_rsqrt(float v)
{
    uint32_t b;
    memcpy(&b, &v, 4);
    b = 0x5f3759dfu - (b >> 1);
    float y;
    memcpy(&y, &b, 4);
    const float h = 0.5f * v;
    for (int k = 0; k < 3; k++) {
        const float t = y * y;
        const float u = fmaf(-h, t, 1.5f);
        y = y * u;
    }
    return y;
}

/* 1/v for normal positive v (spec rule 11): seed by negating the exponent, three Newton steps y <- y + y (1 - v y) */
static inline float sp_recip(float v)
{
    uint32_t b;
    memcpy(&b, &v, 4);
    b = 0x7EF127EAu - b;
    float y;
    memcpy(&y, &b, 4);
    for (int k = 0; k < 3; k++) {
        const float e = fmaf(-v, y, 1.0f);
        y = fmaf(y, e, y);
    }
    return y;
}

static inline float sp_atan2(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    if (mx == 0.0f) return 0.0f;
    float a   = mn / mx;
    int   big = a > WR_TAN_PIO8;
    float num = big ? a - 1.0f : a;
    float den = big ? a + 1.0f : 1.0f;
    float t   = num / den;
    float z   = t * t;
    float p   = fmaf(z, WR_A4, WR_A3);
    p = fmaf(p, z, WR_A2);
    p = fmaf(p, z, WR_A1);
    float r = fmaf(p * z, t, t);
    if (big) r = r + WR_PIO4;
    if (ay > ax) r = WR_PIO2 - r;
    if (x < 0.0f) r = WR_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

static inline float sp_log2(float v)   /* v finite, > 0, normal */
{
    uint32_t b;
    memcpy(&b, &v, 4);
    int      e  = (int)((b >> 23) & 0xffu) - 127;
    uint32_t mb = (b & 0x7fffffu) | 0x3f800000u;
    float    m;
    memcpy(&m, &mb, 4);
    if (m > WR_SQRT2) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float p = fmaf(f, WR_L9, WR_L8);
    p = fmaf(p, f, WR_L7);
    p = fmaf(p, f, WR_L6);
    p = fmaf(p, f, WR_L5);
    p = fmaf(p, f, WR_L4);
    p = fmaf(p, f, WR_L3);
    p = fmaf(p, f, WR_L2);
    p = fmaf(p, f, WR_L1);
    float ln = fmaf(p * z, f, fmaf(-0.5f, z, f));   /* f - z/2 + f*z*p */
    return fmaf(ln, WR_LOG2E, (float)e);
}

static inline float sp_snr_db(float signal, float noise)
{
    float v = (signal / noise) / 2.0f;
    if (!(v > 1e-10f)) return -100.0f;
    if (v > 1e10f) return 100.0f;
    return WR_10LOG10_2 * sp_log2(v);
}

/* x * (c + j s) */
static inline c32 sp_rot(c32 x, float s, float c)
{
    c32 r;
    r.re = fmaf(-x.im, s, x.re * c);
    r.im = fmaf(x.im, c, x.re * s);
    return r;
}

/* generic complex product */
static inline c32 sp_cmul(c32 x, c32 w)
{
    c32 r;
    r.re = fmaf(-x.im, w.im, x.re * w.re);
    r.im = fmaf(x.im, w.re, x.re * w.im);
    return r;
}

/* conj(a) * b */
static inline c32 sp_conj_mul(c32 a, c32 b)
{
    c32 r;
    r.re = fmaf(a.im, b.im, a.re * b.re);
    r.im = fmaf(-a.im, b.re, a.re * b.im);
    return r;
}

static inline c32 cadd(c32 a, c32 b) { c32 r = { a.re + b.re, a.im + b.im }; return r; }
static inline c32 csub(c32 a, c32 b) { c32 r = { a.re - b.re, a.im - b.im }; return r; }
static inline c32 cneg(c32 a) { c32 r = { -a.re, -a.im }; return r; }

/* ------------------------------------------------------------------------------------------- */
/* a1: autocorrelation graph (delay 16, conjugate, multiply, moving averages 48 / 64)          */
/*     gnu_radio/IRS_AP.py:277-285,294-305                                                      */

/* Kogge-Stone inclusive prefix (dir=+1) or suffix (dir=-1) sums over one block of 16 */
static void ks16(const float* in, float* out, int suffix)
{
    float v[16], w[16];
    memcpy(v, in, sizeof v);
    for (int k = 1; k < 16; k <<= 1) {
        for (int i = 0; i < 16; i++) {
            if (!suffix) w[i] = (i >= k) ? v[i - k] + v[i] : v[i];
            else         w[i] = (i + k < 16) ? v[i + k] + v[i] : v[i];
        }
        memcpy(v, w, sizeof v);
    }
    memcpy(out, v, sizeof v);
}

/* Computes for samples n in [0, n_samp): above[n] = (c[n] > thr) and the A[n] the trigger reads.
 * x[n<0] = 0.  n_samp is processed in blocks of 16 (a trailing partial block is padded by zeros
 * and only its valid part is written). */
static void autocorr_spec(const c32* x, long n_samp, float thr, uint8_t* above, c32* A)
{
    long nblk = (n_samp + 15) / 16;
    /* per block: prefix H, suffix S of (a.re, a.im, p) */
    float* H = (float*)malloc((size_t)nblk * 3 * 16 * sizeof(float));
    float* S = (float*)malloc((size_t)nblk * 3 * 16 * sizeof(float));
    for (long m = 0; m < nblk; m++) {
        float ar[16], ai[16], pw[16];
        for (int r = 0; r < 16; r++) {
            long n = 16 * m + r;
            c32 xn = { 0, 0 }, xd = { 0, 0 };
            if (n < n_samp) xn = x[n];
            if (n - 16 >= 0 && n - 16 < n_samp) xd = x[n - 16];
            ar[r] = fmaf(xn.im, xd.im, xn.re * xd.re);
            ai[r] = fmaf(xn.im, xd.re, -(xn.re * xd.im));
            pw[r] = fmaf(xn.im, xn.im, xn.re * xn.re);
        }
        ks16(ar, H + (m * 3 + 0) * 16, 0);  ks16(ar, S + (m * 3 + 0) * 16, 1);
        ks16(ai, H + (m * 3 + 1) * 16, 0);  ks16(ai, S + (m * 3 + 1) * 16, 1);
        ks16(pw, H + (m * 3 + 2) * 16, 0);  ks16(pw, S + (m * 3 + 2) * 16, 1);
    }
#define HB(m, c, r) (((m) < 0) ? 0.0f : H[((m) * 3 + (c)) * 16 + (r)])
#define TB(m, c, r) ((((m) < 0) || (r) == 15) ? 0.0f : S[((m) * 3 + (c)) * 16 + (r) + 1])
    for (long n = 0; n < n_samp; n++) {
        long m = n / 16;
        int  r = (int)(n % 16);
        float Ar = ((TB(m - 3, 0, r) + HB(m - 2, 0, 15)) + HB(m - 1, 0, 15)) + HB(m, 0, r);
        float Ai = ((TB(m - 3, 1, r) + HB(m - 2, 1, 15)) + HB(m - 1, 1, 15)) + HB(m, 1, r);
        float P  = (((TB(m - 4, 2, r) + HB(m - 3, 2, 15)) + HB(m - 2, 2, 15)) + HB(m - 1, 2, 15)) + HB(m, 2, r);
        float m2 = fmaf(Ai, Ai, Ar * Ar);
        float tp = thr * P;
        above[n] = m2 > tp * tp;
        A[n].re = Ar;
        A[n].im = Ai;
    }
#undef HB
#undef TB
    free(H);
    free(S);
}

/* upstream-style: float products, running window sums kept in double, c = |A|/P compared in float */
static void autocorr_libm(const c32* x, long n_samp, float thr, uint8_t* above, c32* A)
{
    double sr = 0, si = 0, sp = 0;
    for (long n = 0; n < n_samp; n++) {
        float complex xn = x[n].re + I * x[n].im;
        float complex xd = (n >= 16) ? x[n - 16].re + I * x[n - 16].im : 0;
        float complex a  = xn * conjf(xd);
        sr += crealf(a);  si += cimagf(a);
        sp += (double)(crealf(xn) * crealf(xn) + cimagf(xn) * cimagf(xn));
        if (n >= 48) {
            float complex xo = x[n - 48].re + I * x[n - 48].im;
            float complex xod = (n - 48 >= 16) ? x[n - 64].re + I * x[n - 64].im : 0;
            float complex ao = xo * conjf(xod);
            sr -= crealf(ao);  si -= cimagf(ao);
        }
        if (n >= 64) {
            float complex xo = x[n - 64].re + I * x[n - 64].im;
            sp -= (double)(crealf(xo) * crealf(xo) + cimagf(xo) * cimagf(xo));
        }
        float Ar = (float)sr, Ai = (float)si, P = (float)sp;
        float c = hypotf(Ar, Ai) / P;
        above[n] = c > thr;   /* NaN (0/0) compares false, as in sync_short */
        A[n].re = Ar;
        A[n].im = Ai;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* a2: sync_short state machine, gnu_radio/IRS_AP.py:268                   */
/* Returns the number of triggers; trig[k] = index n of the trigger sample (of the 16-delayed   */
/* stream), cfo[k] = arg(A[n])/16.  `first_only`: stop after the first trigger (batch slots).   */

long orc_sync_short(const c32* x, long n_samp, float thr, int min_plateau, int math_mode,
                    int first_only, int32_t* trig, float* cfo, long cap)
{
    uint8_t* above = (uint8_t*)malloc((size_t)n_samp + 16);
    c32*     A     = (c32*)malloc(((size_t)n_samp + 16) * sizeof(c32));
    if (math_mode == ORC_MATH_SPEC) autocorr_spec(x, n_samp, thr, above, A);
    else                            autocorr_libm(x, n_samp, thr, above, A);
    long n_trig = 0;
    int  state = 0 /* SEARCH */, plateau = 0;
    long copied = 0;
    long i = 0;
    while (i < n_samp && n_trig < cap) {
        if (state == 0) {
            if (above[i]) {
                if (plateau < min_plateau) {
                    plateau++;
                } else {
                    state = 1; copied = 0; plateau = 0;
                    trig[n_trig] = (int32_t)i;
                    cfo[n_trig]  = (math_mode == ORC_MATH_SPEC ? sp_atan2(A[i].im, A[i].re)
                                                               : atan2f(A[i].im, A[i].re)) / 16.0f;
                    n_trig++;
                    if (first_only) break;
                    continue;           /* the trigger sample is not consumed in SEARCH */
                }
            } else {
                plateau = 0;
            }
            i++;
        } else {
            if (above[i]) {
                if (plateau < min_plateau) {
                    plateau++;
                } else if (copied > WIFIRX_MIN_GAP) {
                    copied = 0; plateau = 0;
                    trig[n_trig] = (int32_t)i;
                    cfo[n_trig]  = (math_mode == ORC_MATH_SPEC ? sp_atan2(A[i].im, A[i].re)
                                                               : atan2f(A[i].im, A[i].re)) / 16.0f;
                    n_trig++;
                    continue;           /* re-processed with the new frame's counters */
                }
            } else {
                plateau = 0;
            }
            copied++;
            i++;
            if (copied == WIFIRX_MAX_SAMPLES) state = 0;
        }
    }
    free(above);
    free(A);
    return n_trig;
}

/* ------------------------------------------------------------------------------------------- */
/* a4: 64-point DFT, output fft-shifted (bin i <-> sub-carrier i-32), gnu_radio/IRS_AP.py:273   */

static void fft64_spec(const c32* in, c32* out_shifted)
{
    c32 a[64], b[64];
    memcpy(a, in, sizeof a);
    /* three radix-4 DIF stages: span 16, 4, 1; twiddle W64^(q*n*step) */
    int span = 16, step = 1;
    for (int stage = 0; stage < 3; stage++) {
        for (int base = 0; base < 64; base += 4 * span) {
            for (int n = 0; n < span; n++) {
                c32 x0 = a[base + n], x1 = a[base + n + span], x2 = a[base + n + 2 * span],
                    x3 = a[base + n + 3 * span];
                c32 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = csub(x1, x3);
                c32 mj = { t3.im, -t3.re };       /* -j * t3 */
                c32 pj = { -t3.im, t3.re };       /* +j * t3 */
                c32 y[4];
                y[0] = cadd(t0, t2);
                y[1] = cadd(t1, mj);
                y[2] = csub(t0, t2);
                y[3] = cadd(t1, pj);
                for (int q = 0; q < 4; q++) {
                    int k = (q * n * step) & 63;
                    c32 w = { WR_TWIDDLE64[2 * k], WR_TWIDDLE64[2 * k + 1] };
                    b[base + n + q * span] = sp_cmul(y[q], w);
                }
            }
        }
        memcpy(a, b, sizeof a);
        span >>= 2;
        step <<= 2;
    }
    /* position p = 16 q1 + 4 q2 + q3 holds X[k], k = q1 + 4 q2 + 16 q3; shifted index (k+32)%64 */
    for (int p = 0; p < 64; p++) {
        int q1 = p >> 4, q2 = (p >> 2) & 3, q3 = p & 3;
        int k = q1 + 4 * q2 + 16 * q3;
        out_shifted[(k + 32) & 63] = a[p];
    }
}

static void fft64_libm(const c32* in, c32* out_shifted)
{
    for (int i = 0; i < 64; i++) {
        int k = i - 32;
        double sr = 0, si = 0;
        for (int n = 0; n < 64; n++) {
            double ang = -2.0 * M_PI * (double)k * (double)n / 64.0;
            double c = cos(ang), s = sin(ang);
            sr += in[n].re * c - in[n].im * s;
            si += in[n].re * s + in[n].im * c;
        }
        out_shifted[i].re = (float)sr;
        out_shifted[i].im = (float)si;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Viterbi K=7 (133,171), hard decisions with erasures (value 2), full traceback               */

static int parity7(int v) { v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return v & 1; }

/* coded: 2*n_bits entries in {0,1,2}; out: n_bits decoded bits. Start state 0; final state = the
 * smallest metric (lowest index on ties); on equal path metrics the predecessor with the older
 * bit 0 survives. */
static void viterbi_decode(const uint8_t* coded, int n_bits, uint8_t* out)
{
    const int BIG = 1 << 28;
    int pm[64], nm[64];
    uint64_t* dec = (uint64_t*)malloc((size_t)n_bits * sizeof(uint64_t));
    for (int s = 0; s < 64; s++) pm[s] = BIG;
    pm[0] = 0;
    for (int t = 0; t < n_bits; t++) {
        int ra = coded[2 * t], rb = coded[2 * t + 1];
        uint64_t d = 0;
        for (int s = 0; s < 64; s++) {
            int u = s & 1, p0 = s >> 1, p1 = (s >> 1) | 32;
            int f0 = (p0 << 1) | u, f1 = (p1 << 1) | u;
            int a0 = parity7(f0 & 0155), b0 = parity7(f0 & 0117);
            int a1 = parity7(f1 & 0155), b1 = parity7(f1 & 0117);
            int bm0 = ((ra != 2) && (ra != a0)) + ((rb != 2) && (rb != b0));
            int bm1 = ((ra != 2) && (ra != a1)) + ((rb != 2) && (rb != b1));
            int m0 = pm[p0] + bm0, m1 = pm[p1] + bm1;
            if (m1 < m0) { nm[s] = m1; d |= (uint64_t)1 << s; }
            else         { nm[s] = m0; }
        }
        dec[t] = d;
        memcpy(pm, nm, sizeof pm);
    }
    int best = 0;
    for (int s = 1; s < 64; s++) if (pm[s] < pm[best]) best = s;
    int s = best;
    for (int t = n_bits - 1; t >= 0; t--) {
        out[t] = (uint8_t)(s & 1);
        int h = (int)((dec[t] >> s) & 1);
        s = (s >> 1) | (h << 5);
    }
    free(dec);
}

/* rate table: index = encoding */
static const int N_BPSC[8] = { 1, 1, 2, 2, 4, 4, 6, 6 };
static const int N_DBPS[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
static const int PUNCT[8]  = { 0, 2, 0, 2, 0, 2, 1, 2 };   /* 0: 1/2, 1: 2/3, 2: 3/4 */

/* SIGNAL field: 48 hard BPSK decisions -> (encoding, length); returns 1 when parity+rate are ok */
static int decode_signal(const uint8_t* rx48, int* encoding, int* length)
{
    uint8_t deint[48], bits[24];
    for (int i = 0; i < 48; i++) deint[i] = rx48[3 * (i % 16) + i / 16];
    viterbi_decode(deint, 24, bits);
    int r = 0, len = 0, par = 0;
    for (int i = 0; i < 17; i++) {
        par ^= bits[i];
        if (i < 4 && bits[i]) r |= 1 << i;
        if (bits[i] && i > 4 && i < 17) len |= 1 << (i - 5);
    }
    if (par != bits[17]) return 0;
    int enc;
    switch (r) {
    case 11: enc = 0; break;
    case 15: enc = 1; break;
    case 10: enc = 2; break;
    case 14: enc = 3; break;
    case 9:  enc = 4; break;
    case 13: enc = 5; break;
    case 8:  enc = 6; break;
    case 12: enc = 7; break;
    default: return 0;
    }
    *encoding = enc;
    *length = len;
    return 1;
}

/* a6: hard decision, index LSB = first transmitted bit */
static inline uint8_t decide(c32 y, int n_bpsc)
{
    float re = y.re, im = y.im, are = fabsf(re), aim = fabsf(im);
    unsigned r = 0;
    switch (n_bpsc) {
    case 1:
        return re > 0.0f;
    case 2:
        return (uint8_t)((re > 0.0f) | ((im > 0.0f) << 1));
    case 4:
        r |= re > 0.0f;
        r |= (are < WR_T16_2) << 1;
        r |= (im > 0.0f) << 2;
        r |= (aim < WR_T16_2) << 3;
        return (uint8_t)r;
    default:
        r |= re > 0.0f;
        r |= (are < WR_T64_4) << 1;
        r |= ((are < WR_T64_6) && (are > WR_T64_2)) << 2;
        r |= (im > 0.0f) << 3;
        r |= (aim < WR_T64_4) << 4;
        r |= ((aim < WR_T64_6) && (aim > WR_T64_2)) << 5;
        return (uint8_t)r;
    }
}

/* the constellation point of an index (levels formed in float32 as the upstream constellations do) */
static inline c32 point_of(unsigned idx, int n_bpsc)
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
    {
        const float l = WR_T64_2 * 0.5f;
        /* per axis: bit1 = |u| < 4a, bit2 = 2a < |u| < 6a  ->  |u| = 7a,5a,1a,3a for (b1,b2) = 00,01,10,11 */
        unsigned r = (idx >> 1) & 3, q = (idx >> 4) & 3;
        float tab[4] = { 7.0f * l, 5.0f * l, 1.0f * l, 3.0f * l };
        float ar = tab[((r & 1) << 1) | (r >> 1)], ai = tab[((q & 1) << 1) | (q >> 1)];
        p.re = (idx & 1) ? ar : -ar;
        p.im = (idx & 8) ? ai : -ai;
        return p;
    }
}

/* a7: max-log piecewise-linear LLRs, positive <=> bit 1 (unscaled form) */
static inline void llr_of(c32 y, int n_bpsc, float* out)
{
    float re = y.re, im = y.im, are = fabsf(re), aim = fabsf(im);
    switch (n_bpsc) {
    case 1: out[0] = re; break;
    case 2: out[0] = re; out[1] = im; break;
    case 4:
        out[0] = re; out[1] = WR_T16_2 - are;
        out[2] = im; out[3] = WR_T16_2 - aim;
        break;
    default:
        out[0] = re; out[1] = WR_T64_4 - are; out[2] = WR_T64_2 - fabsf(are - WR_T64_4);
        out[3] = im; out[4] = WR_T64_4 - aim; out[5] = WR_T64_2 - fabsf(aim - WR_T64_4);
        break;
    }
}

/* pairwise (xor-butterfly) sum over 64 values: the reduction tree of the spec */
static float tree_sum64(const float* v)
{
    float a[64];
    memcpy(a, v, sizeof a);
    for (int k = 1; k < 64; k <<= 1) {
        float b[64];
        for (int i = 0; i < 64; i++) b[i] = a[i] + a[i ^ k];
        memcpy(a, b, sizeof a);
    }
    return a[0];
}

/* ------------------------------------------------------------------------------------------- */
/* a3 + a4 + a5 + a6 + a7 for one trigger.                                                      */
/* x: the stream (or slot); n_samp its length; t: trigger index; L: number of copied samples    */
/* y[0..L) that belong to this trigger (y[m] = x[t-16+m] * exp(-j cfo_c m)).                    */
/* Outputs for this frame: idx [max_sym*48], llr [max_sym*48*llr_bits], eq [max_sym*48].        */

static inline c32 x_at(const c32* x, long n_samp, long n)
{
    c32 z = { 0, 0 };
    return (n >= 0 && n < n_samp) ? x[n] : z;
}

void orc_frame(const c32* x, long n_samp, long t, float cfo_c, long L, const orc_params* prm,
               wifirx_frame* fr, uint8_t* idx, float* llr, c32* eq, c32* csi)
{
    const int spec = prm->math_mode == ORC_MATH_SPEC;
    fr->flags = WIFIRX_F_DETECTED;
    fr->trigger = (int32_t)t;
    fr->frame_start = 0;
    fr->cfo_coarse = cfo_c;
    fr->cfo_fine = 0;
    fr->snr_db = 0;
    fr->psdu_len = 0; fr->encoding = 0; fr->n_bpsc = 0; fr->n_sym = 0; fr->n_sym_out = 0;
    if (L < WIFIRX_SYNC_LENGTH + 63) { fr->flags |= WIFIRX_F_TRUNCATED; return; }  /* not enough samples for the LTS search */

    /* -- sync_short COPY: coarse derotation of the first 383 samples (a 384th is formed too: the spec's correlation
     *    multiplies it by zero coefficients, DESIGN.md section 4 rule 6) -- */
    c32 y[WIFIRX_SYNC_LENGTH + 64];
    if (spec) {
        /* Spec rule 5a: the phasor of sample m = 64 p + l is that of sample l (sp_sincos of the float angle -cfo_c l)
         * carried p times by exp(-j float(cfo_c 64)): rule-2 products, two sincos per residue l instead of six. */
        c32 w64;
        sp_sincos(-cfo_c * 64.0f, &w64.im, &w64.re);
        for (int l = 0; l < 64; l++) {
            c32 w;
            sp_sincos(-cfo_c * (float)l, &w.im, &w.re);
            for (int p = 0; p < (WIFIRX_SYNC_LENGTH + 64) / 64; p++) {
                if (p) w = sp_cmul(w, w64);
                y[64 * p + l] = sp_cmul(x_at(x, n_samp, t - 16 + 64 * p + l), w);
            }
        }
    } else {
        for (int m = 0; m < WIFIRX_SYNC_LENGTH + 64; m++) {
            c32 xs = x_at(x, n_samp, t - 16 + m);
            float ang = -cfo_c * (float)m;
            float s, c;
            sincosf(ang, &s, &c);
            float complex v = (xs.re + I * xs.im) * (c + I * s); y[m].re = crealf(v); y[m].im = cimagf(v);
        }
    }
    /* -- sync_long SYNC: 64-tap LTS correlation over 320 lags -- */
    c32   corr[WIFIRX_SYNC_LENGTH];
    float mag[WIFIRX_SYNC_LENGTH];
    int   top[4] = { -1, -1, -1, -1 };
    if (spec) {
        /* Spec rule 6.  The search runs in two stages: WHERE the peaks are is decided on 8-bit integers (exact integer
         * arithmetic: any order of summation gives the same numbers), WHAT they are -- the values the fine CFO is taken
         * from -- is computed in float32 for the candidates alone.
         * Stage 1.  The 384 copied samples are scaled by a power of two that puts the largest component into [64, 128)
         * and rounded to integers in [-127, 127] (NaN -> 0); the taps are rint(64 l) (WR_LTS_Q8).  For every lag the
         * integer correlation conj(lq) yq, its squared magnitude in float32 (the integers are below 2^24: exact
         * conversions).  The two largest (lowest lag first among equals) are the candidates when they are exactly 64 lags
         * apart AND the third largest squared magnitude is below 7/8 of the second (float32 product) -- then no rounding
         * of the integer stage can have changed which two lags lead, and sync_long's pair search ends at that pair
         * whatever ranks 3 and 4 are; otherwise the eight largest are the candidates.  (Round 2 took the top two whenever
         * they were 64 apart: tests/campaigns/lts_rule6.py found 4 frames in 154 000 at 0..8 dB, two equally strong paths,
         * where a third lag within 0.3 % led the float32 ranking and the exhaustive search chose another pair.) */
        int E = 0;                                               /* largest biased exponent among the 768 components */
        for (int m = 0; m < WIFIRX_SYNC_LENGTH + 64; m++) {
            uint32_t br, bi;
            memcpy(&br, &y[m].re, 4); memcpy(&bi, &y[m].im, 4);
            const int er = (int)((br >> 23) & 0xffu), ei = (int)((bi >> 23) & 0xffu);
            if (er > E) E = er;
            if (ei > E) E = ei;
        }
        int sfield = 260 - E;                                    /* x < 2^(E-126)  ->  x 2^(133-E) < 128 */
        if (sfield > 254) sfield = 254;
        const uint32_t sbits = (uint32_t)sfield << 23;
        float scale;
        memcpy(&scale, &sbits, 4);
        int8_t q[2 * (WIFIRX_SYNC_LENGTH + 64)];
        for (int m = 0; m < 2 * (WIFIRX_SYNC_LENGTH + 64); m++) {
            const float v = (&y[0].re)[m] * scale;
            const float r = rintf(v);
            q[m] = (int8_t)((r >= -127.0f && r <= 127.0f) ? (int)r : (r > 127.0f ? 127 : (r < -127.0f ? -127 : 0)));
        }
        float mag1[WIFIRX_SYNC_LENGTH];
        for (int i = 0; i < WIFIRX_SYNC_LENGTH; i++) {
            int32_t cr = 0, ci = 0;
            for (int k = 0; k < 64; k++) {
                const int lr = WR_LTS_Q8[2 * k], li = WR_LTS_Q8[2 * k + 1];
                const int yr = q[2 * (i + k)], yi = q[2 * (i + k) + 1];
                cr += lr * yr + li * yi;
                ci += lr * yi - li * yr;
            }
            const float fr_ = (float)cr, fi_ = (float)ci;
            mag1[i] = fmaf(fi_, fi_, fr_ * fr_);
        }
        int cand[WIFIRX_SYNC_LENGTH], n_cand = 0;
        if (prm->lts_search == 1) {                              /* every lag is a candidate: the exhaustive float32 search */
            for (int i = 0; i < WIFIRX_SYNC_LENGTH; i++) cand[n_cand++] = i;
        } else
        for (int r = 0; r < 8; r++) {
            int best = -1;
            for (int i = 0; i < WIFIRX_SYNC_LENGTH; i++) {
                int used = 0;
                for (int c = 0; c < n_cand; c++) used |= (cand[c] == i);
                if (used) continue;
                if (best < 0 || mag1[i] > mag1[best]) best = i;
            }
            if (r == 2 && abs(cand[0] - cand[1]) == 64 && mag1[best] < 0.875f * mag1[cand[1]])
                break;                                          /* the usual case: the two LTS peaks, clear of every other lag */
            cand[n_cand++] = best;
        }
        /* Stage 2.  The candidates' correlation values in float32: lag i = 8a + b reads the 144 floats (72 samples, re/im
         * interleaved) from sample 8a on; float phi = 2m + part meets tap k = m - b (coefficient 0 outside 0..63).  Both
         * sums are ONE fmaf chain each over the floats in the order j = 0..8, s' = 0..3, kk = 0..3 of phi = 16 j + 4 kk + s'
         * (the order in which round 2's matrix-instruction form accumulated them; kept so that the values did not move
         * when the search went to integers).  conj(l) y: real part lr yr + li yi, imaginary part lr yi - li yr; -li is
         * formed as 0 - li (no negative zero among the coefficients). */
        for (int c = 0; c < n_cand; c++) {
            const int i = cand[c];
            const float* A = &y[8 * (i >> 3)].re;
            const int b = i & 7;
            float ar = 0.0f, ai = 0.0f;
            for (int j = 0; j < 9; j++)
                for (int sp = 0; sp < 4; sp++)
                    for (int kk = 0; kk < 4; kk++) {
                        const int phi = 16 * j + 4 * kk + sp, m = phi >> 1, part = phi & 1, k = m - b;
                        float cr = 0.0f, ci = 0.0f;
                        if (k >= 0 && k < 64) {
                            const float lr = WR_LTS_TIME[2 * k], li = WR_LTS_TIME[2 * k + 1];
                            cr = part ? li : lr;
                            ci = part ? lr : 0.0f - li;
                        }
                        ar = fmaf(A[phi], cr, ar);
                        ai = fmaf(A[phi], ci, ai);
                    }
            corr[i].re = ar; corr[i].im = ai;
            mag[i] = fmaf(ai, ai, ar * ar);
        }
        /* the (up to) four largest candidates by their float32 magnitude, ties -> lower lag; a NaN magnitude is never a peak */
        for (int r = 0; r < 4; r++) {
            int best = -1;
            for (int c = 0; c < n_cand; c++) {
                const int i = cand[c];
                int used = 0;
                for (int qq = 0; qq < r; qq++) used |= (top[qq] == i);
                if (used || !(mag[i] >= 0.0f)) continue;
                if (best < 0 || mag[i] > mag[best] || (mag[i] == mag[best] && i < best)) best = i;
            }
            top[r] = best;
        }
    } else {
        for (int i = 0; i < WIFIRX_SYNC_LENGTH; i++) {
            double ar = 0, ai = 0;
            for (int k = 0; k < 64; k++) {
                double lr = WR_LTS_TIME[2 * k], li = WR_LTS_TIME[2 * k + 1];
                ar += lr * y[i + k].re + li * y[i + k].im;
                ai += lr * y[i + k].im - li * y[i + k].re;
            }
            corr[i].re = (float)ar; corr[i].im = (float)ai;
            mag[i] = hypotf(corr[i].re, corr[i].im);
        }
        /* top 4 by magnitude, ties -> lower offset first (stable sort of the upstream list) */
        for (int r = 0; r < 4; r++) {
            int best = -1;
            for (int i = 0; i < WIFIRX_SYNC_LENGTH; i++) {
                int used = 0;
                for (int q = 0; q < r; q++) used |= (top[q] == i);
                if (used || !(mag[i] >= 0.0f)) continue;            /* a NaN magnitude is never a peak */
                if (best < 0 || mag[i] > mag[best]) best = i;
            }
            top[r] = best;
        }
    }
    if (prm->dbg_top4) {                                /* test hook: the four peaks the pair search sees, and their magnitudes */
        for (int r = 0; r < 4; r++) {
            prm->dbg_top4[r] = top[r];
            if (prm->dbg_mag4) prm->dbg_mag4[r] = top[r] >= 0 ? mag[top[r]] : 0.0f;
        }
    }
    int   fs = WIFIRX_SYNC_LENGTH, found = 0;
    float cfo_f = 0.0f;
    for (int i = 0; i < 3 && found != 64; i++) {
        for (int k = i + 1; k < 4; k++) {
            int oi = top[i], ok = top[k];
            if (oi < 0 || ok < 0) continue;                     /* fewer than four valid lags */
            c32 first = oi > ok ? corr[ok] : corr[oi];
            c32 second = oi > ok ? corr[oi] : corr[ok];
            int diff = abs(oi - ok);
            if (diff == 64 || diff == 63 || diff == 65) {
                /* first * conj(second) */
                float pr, pi;
                if (spec) {
                    pr = fmaf(first.im, second.im, first.re * second.re);
                    pi = fmaf(first.im, second.re, -(first.re * second.im));
                } else {
                    float complex v = (first.re + I * first.im) * conjf(second.re + I * second.im);
                    pr = crealf(v); pi = cimagf(v);
                }
                float ang = spec ? sp_atan2(pi, pr) : atan2f(pi, pr);
                fs = oi < ok ? oi : ok;
                cfo_f = ang / (float)diff;
                found = diff;
                if (diff == 64) break;
            }
        }
    }
    if (!found) {
        if (prm->no_pair_fallback != 2) return;        /* no LTS pair: frame dropped */
        fs = WIFIRX_SYNC_LENGTH;                       /* upstream: the default start and the previous frame's offset */
        cfo_f = prm->fallback_cfo_f;
    }
    fr->flags |= WIFIRX_F_SYNC;
    fr->frame_start = fs;
    fr->cfo_fine = cfo_f;

    /* -- frame_equalizer state -- */
    const double bw = prm->bandwidth, fc = prm->frequency;
    const double tag = (double)cfo_c - (double)cfo_f;          /* sync_long's wifi_start tag */
    const double eps0 = tag * bw / (2 * M_PI * fc);
    double d_er = 0.0;
    /* Spec rule 9: the control chain of the sampling-offset compensation in float32 -- eps0 and the scale of the residual
     * estimate are formed once per frame in double and rounded; the IIR state, its input and the angle factor are float.
     * (Upstream keeps doubles; the distance to that is measured: profiles/r02_spec_vs_libm.json.) */
    const float eps0_f = (float)eps0, er_scale_f = (float)(bw / (2 * M_PI * fc * 80));
    float d_er_f = 0.0f, er_f = 0.0f;
    const double theta_d = (double)cfo_f - (double)cfo_c;       /* total derotation, rad/sample */
    /* ... as an integer phase increment: 2^-62 quarter turns per sample */
    const uint64_t Qp = (uint64_t)(int64_t)rint(theta_d * WR_TWO_OVER_PI_D * 4611686018427387904.0);
    c32 u16, u80;                                               /* exp(j theta 16), exp(j theta 80) */
    sp_sincos_q(Qp * 16u, &u16.im, &u16.re);
    sp_sincos_q(Qp * 80u, &u80.im, &u80.re);
    c32 wbase[16];                                              /* phasor of the first sample of lane r of the current symbol */
    memset(wbase, 0, sizeof wbase);
    c32 prev[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
    c32 H[64], G[64], DH[64];                                   /* DH: running estimate of COMB / STA */
    float W[64];                                                /* |H|^2 of the LS estimate: the LLR weight (llr_csi) */
    memset(W, 0, sizeof W);
    memset(H, 0, sizeof H);
    memset(G, 0, sizeof G);
    memset(DH, 0, sizeof DH);
    int n_sym = 0, n_bpsc = 1, enc = 0, psdu_len = 0, have_signal = 0;
    int n_out = 0;
    float snr = 0.0f;

    for (int s = 0; s <= n_sym + 2; s++) {
        long off0 = fs + (s < 2 ? 64 * s : 128 + 80 * (s - 2) + 16);
        if (off0 + 64 > L || (s > 2 && (s - 3) >= prm->max_sym)) {   /* samples / output capacity ran out */
            fr->flags |= WIFIRX_F_TRUNCATED;
            break;
        }
        /* sync_short copy + sync_long copy.  Upstream: two float rotations exp(-j cfo_c m), exp(+j cfo_f m)
         * per sample.  Spec (section 4.8): one rotation by the total offset; the phasor of sample m0 + r (r = 0..15)
         * comes from a double angle reduced in double, the samples 16, 32, 48 further on from three
         * multiplications by the frame's exp(j theta 16).  The phase is kept as an exact integer (Q*m mod 2^64).
         * From one symbol to the next (80 samples on, from the second long training symbol) the base phasor is carried by
         * one multiplication with exp(j theta 80); every eighth symbol (s = 1, 9, 17, ...) it is formed from the exact
         * phase again, so that at most seven products separate a phasor from an exact one. */
        c32 z[64], X[64];
        if (spec) {
            const int exact = (s < 2) || (((s - 1) & 7) == 0);
            for (int r = 0; r < 16; r++) {
                if (exact) sp_sincos_q(Qp * (uint64_t)(off0 + r), &wbase[r].im, &wbase[r].re);
                else       wbase[r] = sp_cmul(wbase[r], u80);
                c32 w = wbase[r];
                for (int j = 0; j < 4; j++) {
                    c32 xs = x_at(x, n_samp, t - 16 + off0 + r + 16 * j);
                    z[r + 16 * j] = sp_cmul(xs, w);
                    w = sp_cmul(w, u16);
                }
            }
        } else {
            for (int i = 0; i < 64; i++) {
                long m = off0 + i;
                c32 xs = x_at(x, n_samp, t - 16 + m);
                float a1 = -cfo_c * (float)m, a2 = (float)m * cfo_f, s1, c1, s2, c2;
                if (prm->libm_exact_phase) {
                    const double d1 = -(double)cfo_c * (double)m, d2 = (double)m * (double)cfo_f;
                    s1 = (float)sin(d1); c1 = (float)cos(d1); s2 = (float)sin(d2); c2 = (float)cos(d2);
                } else {
                    sincosf(a1, &s1, &c1); sincosf(a2, &s2, &c2);
                }
                float complex v = ((xs.re + I * xs.im) * (c1 + I * s1)) * (c2 + I * s2);
                z[i].re = crealf(v); z[i].im = cimagf(v);
            }
        }
        if (spec) fft64_spec(z, X); else fft64_libm(z, X);

        /* (1) sampling offset compensation */
        double t4 = 2 * M_PI * s * 80 * (eps0 + d_er);
        if (spec) {
            /* Spec rule 9: b(r) = exp(j kf (r - 16)) is the phasor of bin r + 16; the step between bins 16 apart is
             * conj(b(0)) = exp(j kf 16); bins r, r + 32, r + 48 take b(r) b(0), b(r) step, (b(r) step) step. */
            const float kf = WR_T4_64F[s] * (eps0_f + d_er_f);      /* T4F[s] = float(2 pi s 80 / 64) */
            float s0, c0;
            sp_sincos(kf * (float)(0 - 16), &s0, &c0);
            const c32 b0 = { c0, s0 }, step = { c0, -s0 };
            for (int r = 0; r < 16; r++) {
                float bs, bc;
                sp_sincos(kf * (float)(r - 16), &bs, &bc);
                const c32 b = { bc, bs };
                const c32 q0 = sp_cmul(b, b0), q2 = sp_cmul(b, step), q3 = sp_cmul(q2, step);
                X[r]      = sp_cmul(X[r], q0);
                X[r + 16] = sp_cmul(X[r + 16], b);
                X[r + 32] = sp_cmul(X[r + 32], q2);
                X[r + 48] = sp_cmul(X[r + 48], q3);
            }
        } else {
            for (int i = 0; i < 64; i++) {
                float ang = (float)(t4 * (i - 32) / 64), sn, cs;
                sincosf(ang, &sn, &cs);
                float complex v = (X[i].re + I * X[i].im) * (cs + I * sn);
                X[i].re = crealf(v); X[i].im = cimagf(v);
            }
        }
        /* (2) pilot common phase */
        float p = (s >= 2) ? (float)WR_POLARITY[(s - 2) % 127] : 1.0f;
        c32 S;
        if (s < 2) S = cadd(cadd(csub(X[11], X[25]), X[39]), X[53]);
        else {
            S = csub(cadd(cadd(X[11], X[39]), X[25]), X[53]);
            if (p < 0) S = cneg(S);
        }
        float beta = spec ? 0.0f : atan2f(S.im, S.re);    /* the spec never forms beta itself */
        /* (3) residual frequency offset from pilot rotation between symbols */
        c32 cur[4];
        if (s < 2) { cur[0] = X[11]; cur[1] = cneg(X[25]); cur[2] = X[39]; cur[3] = X[53]; }
        else {
            cur[0] = p < 0 ? cneg(X[11]) : X[11];
            cur[1] = p < 0 ? cneg(X[25]) : X[25];
            cur[2] = p < 0 ? cneg(X[39]) : X[39];
            cur[3] = p < 0 ? X[53] : cneg(X[53]);
        }
        double er = 0.0;
        if (s >= 2) {
            c32 acc;
            if (spec) {
                acc = cadd(cadd(cadd(sp_conj_mul(prev[0], cur[0]), sp_conj_mul(prev[1], cur[1])),
                                sp_conj_mul(prev[2], cur[2])), sp_conj_mul(prev[3], cur[3]));
            } else {
                float complex v = 0;
                for (int q = 0; q < 4; q++)
                    v += conjf(prev[q].re + I * prev[q].im) * (cur[q].re + I * cur[q].im);
                acc.re = crealf(v); acc.im = cimagf(v);
            }
            float erf = spec ? sp_atan2(acc.im, acc.re) : atan2f(acc.im, acc.re);
            er = (double)erf * (bw / (2 * M_PI * fc * 80));
            er_f = erf * er_scale_f;
        }
        memcpy(prev, cur, sizeof prev);
        /* (4) derotate by -beta.  Spec rule 10: exp(-j beta) = conj(S) * rsqrt(|S|^2), the reciprocal root by three
         * Newton steps from the exponent-halving seed (plain multiplies and fmas: the same bits on CPU and GPU, relative
         * error 1.3e-7); no atan2/sincos round trip; S = 0 rotates by 0 like arg(0) = 0. */
        {
            float sn, cs;
            if (spec) {
                float n2 = fmaf(S.im, S.im, S.re * S.re);
                float inv = sp_rsqrt(n2);
                cs = (n2 > 0.0f) ? S.re * inv : 1.0f;
                sn = (n2 > 0.0f) ? -(S.im * inv) : 0.0f;
            } else {
                sincosf(-beta, &sn, &cs);
            }
            for (int i = 0; i < 64; i++) {
                if (spec) X[i] = sp_rot(X[i], sn, cs);
                else {
                    float complex v = (X[i].re + I * X[i].im) * (cs + I * sn);
                    X[i].re = crealf(v); X[i].im = cimagf(v);
                }
            }
        }
        /* (5) residual offset IIR */
        if (s >= 2) {
            double alpha = 0.1;
            d_er = (1 - alpha) * d_er + alpha * er;
            d_er_f = fmaf(0.1f, er_f, 0.9f * d_er_f);
        }
                /* (6a) COMB (ieee802_11.COMB; definition: DESIGN.md section 4.11 -- upstream's source is absent, restated from
         * the published comb-pilot scheme): the four pilots of THIS symbol, polarity removed, are the channel at bins
         * 11, 25, 39, 53; the band edges (bins 0 and 64) take their mean; linear interpolation in between; the frame's
         * estimate follows with d_H = 0.8 d_H + 0.2 H (d_H = H on the first long training symbol). */
        if (prm->chan_est == WIFIRX_EQ_COMB) {
            const float pl = (s >= 2) ? (float)WR_POLARITY[(s - 2) % 127] : 1.0f;
            c32 node[6];
            if (s < 2) { node[1] = X[11]; node[2] = cneg(X[25]); node[3] = X[39]; node[4] = X[53]; }
            else {
                node[1] = pl < 0 ? cneg(X[11]) : X[11];
                node[2] = pl < 0 ? cneg(X[25]) : X[25];
                node[3] = pl < 0 ? cneg(X[39]) : X[39];
                node[4] = pl < 0 ? X[53] : cneg(X[53]);
            }
            c32 sum = cadd(cadd(cadd(node[1], node[2]), node[3]), node[4]);
            node[0].re = 0.25f * sum.re; node[0].im = 0.25f * sum.im;
            node[5] = node[0];
            for (int i = 0; i < 64; i++) {
                const c32 a = node[WR_COMB_SEG[i]], b = node[WR_COMB_SEG[i] + 1];
                c32 h;
                h.re = fmaf(b.re, WR_COMB_W[i], a.re * WR_COMB_U[i]);
                h.im = fmaf(b.im, WR_COMB_W[i], a.im * WR_COMB_U[i]);
                if (s == 0) DH[i] = h;
                else {
                    DH[i].re = 0.8f * DH[i].re + 0.2f * h.re;
                    DH[i].im = 0.8f * DH[i].im + 0.2f * h.im;
                }
            }
        }
/* (6) LS equalizer */
        if (s == 0) {
            memcpy(H, X, sizeof H);
        } else if (s == 1) {
            if (spec) {
                float nv[64], sv[64];
                for (int i = 0; i < 64; i++) {
                    nv[i] = 0.0f; sv[i] = 0.0f;
                    if (i == 32 || i < 6 || i > 58) continue;
                    c32 d = csub(H[i], X[i]), u = cadd(H[i], X[i]);
                    nv[i] = fmaf(d.im, d.im, d.re * d.re);
                    sv[i] = fmaf(u.im, u.im, u.re * u.re);
                    float g = 0.5f * WR_LTS_FREQ[i];
                    H[i].re = u.re * g; H[i].im = u.im * g;
                    /* G = conj(H)/|H|^2: the one-tap equaliser as a multiplier, two divisions per bin per frame */
                    float dd = fmaf(H[i].im, H[i].im, H[i].re * H[i].re);
                    G[i].re = H[i].re / dd;
                    G[i].im = -H[i].im / dd;
                }
                snr = sp_snr_db(tree_sum64(sv), tree_sum64(nv));
            } else {
                double signal = 0, noise = 0;
                for (int i = 0; i < 64; i++) {
                    if (i == 32 || i < 6 || i > 58) continue;
                    c32 d = csub(H[i], X[i]), u = cadd(H[i], X[i]);
                    noise += pow(hypotf(d.re, d.im), 2);
                    signal += pow(hypotf(u.re, u.im), 2);
                    float complex hv = (u.re + I * u.im) / (WR_LTS_FREQ[i] * (2.0f + 0.0f * I));
                    H[i].re = crealf(hv); H[i].im = cimagf(hv);
                }
                snr = (float)(10 * log10(signal / noise / 2));
            }
            for (int i = 6; i <= 58; i++) if (i != 32) W[i] = fmaf(H[i].im, H[i].im, H[i].re * H[i].re);
            if (prm->chan_est == WIFIRX_EQ_STA) memcpy(DH, H, sizeof DH);     /* STA starts from the LS estimate */
            fr->snr_db = snr;
            if (csi) {            /* channel state: the LS estimate on the 52 occupied bins, ascending */
                int k = 0;
                for (int i = 6; i <= 58; i++) if (i != 32) csi[k++] = H[i];
            }
        } else {
            /* spec rule 11: X / H as conj(H) X times ONE reciprocal, r = sp_recip(|H|^2) (Newton from an exponent seed: no
             * division), then two products */
#define SP_CDIV(X_, H_, OUT_) do { const float d_ = fmaf((H_).im, (H_).im, (H_).re * (H_).re); const float r_ = sp_recip(d_);         \
                                   (OUT_).re = fmaf((X_).im, (H_).im, (X_).re * (H_).re) * r_;                                      \
                                   (OUT_).im = fmaf((X_).im, (H_).re, -((X_).re * (H_).im)) * r_; } while (0)
            uint8_t bits48[48];
            int     bin48[48];
            c32     sym48[48];
            c32     HU[64];                                     /* STA: per-bin estimates of this symbol */
            memset(HU, 0, sizeof HU);
            int     c = 0;
            int     nb = (s == 2) ? 1 : n_bpsc;
            for (int i = 0; i < 64; i++) {
                if (i == 11 || i == 25 || i == 32 || i == 39 || i == 53 || i < 6 || i > 58) continue;
                c32 yq;
                if (prm->chan_est == WIFIRX_EQ_COMB || prm->chan_est == WIFIRX_EQ_STA) {
                    /* Y = X / d_H (for STA: the estimate of the previous symbol; it is updated below) */
                    SP_CDIV(X[i], DH[i], yq);
                    if (prm->chan_est == WIFIRX_EQ_STA) {
                        c32 pt = point_of(decide(yq, nb), nb);
                        SP_CDIV(X[i], pt, HU[i]);
                    }
                } else if (spec && prm->chan_est == WIFIRX_EQ_LMS) {
                    /* LMS (decision directed): Y = X/H, then H = H/2 + (X/point)/2 */
                    SP_CDIV(X[i], H[i], yq);
                    c32 pt = point_of(decide(yq, nb), nb), tq;
                    SP_CDIV(X[i], pt, tq);
                    H[i].re = 0.5f * H[i].re + 0.5f * tq.re;
                    H[i].im = 0.5f * H[i].im + 0.5f * tq.im;
                } else if (spec) {
                    yq = sp_cmul(X[i], G[i]);
                } else if (prm->chan_est == WIFIRX_EQ_LMS) {
                    float complex hv = H[i].re + I * H[i].im, xv = X[i].re + I * X[i].im;
                    float complex v = xv / hv;
                    yq.re = crealf(v); yq.im = cimagf(v);
                    c32 pt = point_of(decide(yq, nb), nb);
                    float complex nh = (0.5f + 0.0f * I) * hv + (0.5f + 0.0f * I) * (xv / (pt.re + I * pt.im));
                    H[i].re = crealf(nh); H[i].im = cimagf(nh);
                } else {
                    float complex v = (X[i].re + I * X[i].im) / (H[i].re + I * H[i].im);
                    yq.re = crealf(v); yq.im = cimagf(v);
                }
                sym48[c] = yq;
                bin48[c] = i;
                bits48[c] = decide(yq, nb);
                c++;
            }
            if (prm->chan_est == WIFIRX_EQ_STA) {
                /* STA (ieee802_11.STA; definition: DESIGN.md section 4.11, after Fernandez et al., spectral-temporal
                 * averaging with alpha = 0.5, beta = 2): per-bin estimates X/point on the data bins and X * (known
                 * pilot) on the pilots, averaged over the used bins within +-2, then d_H = d_H/2 + average/2. */
                const float pl = (float)WR_POLARITY[(s - 2) % 127];
                HU[11] = pl < 0 ? cneg(X[11]) : X[11];
                HU[25] = pl < 0 ? cneg(X[25]) : X[25];
                HU[39] = pl < 0 ? cneg(X[39]) : X[39];
                HU[53] = pl < 0 ? X[53] : cneg(X[53]);
                c32 NH[64];
                for (int i = 6; i <= 58; i++) {
                    if (i == 32) continue;
                    /* the window sum takes the five bins i-2 .. i+2 in ascending order, an unused bin as +0 (spec rule 11) */
                    c32 sum = { 0.0f, 0.0f };
                    int cnt = 0;
                    for (int k = i - 2; k <= i + 2; k++) {
                        const int used = !(k == 32 || k < 6 || k > 58);
                        const c32 z = used ? HU[k] : (c32){ 0.0f, 0.0f };
                        sum = (k == i - 2) ? z : cadd(sum, z);
                        cnt += used;
                    }
                    /* the mean over 3, 4 or 5 bins: times the float32 reciprocal of the count */
                    const float inv = cnt == 3 ? 0x1.555556p-2f : cnt == 4 ? 0.25f : 0x1.99999ap-3f;
                    NH[i].re = 0.5f * DH[i].re + 0.5f * (sum.re * inv);
                    NH[i].im = 0.5f * DH[i].im + 0.5f * (sum.im * inv);
                }
                for (int i = 6; i <= 58; i++) if (i != 32) DH[i] = NH[i];
            }
            if (s == 2) {
                /* (7) SIGNAL field */
                if (!decode_signal(bits48, &enc, &psdu_len)) break;
                have_signal = 1;
                n_bpsc = N_BPSC[enc];
                n_sym = (int)ceil((16 + 8 * psdu_len + 6) / (double)N_DBPS[enc]);
                fr->flags |= WIFIRX_F_SIGNAL;
                fr->psdu_len = (uint16_t)psdu_len;
                fr->encoding = (uint8_t)enc;
                fr->n_bpsc = (uint8_t)n_bpsc;
                fr->n_sym = (uint16_t)n_sym;
                if (prm->llr_bits >= n_bpsc && llr) fr->flags |= WIFIRX_F_LLR;
            } else {
                /* (8) data symbol s-3 */
                int q = s - 3;
                if (idx) memcpy(idx + (size_t)q * 48, bits48, 48);
                if (eq) memcpy(eq + (size_t)q * 48, sym48, sizeof sym48);
                if (llr && prm->llr_bits >= n_bpsc)
                    for (int k = 0; k < 48; k++) {
                        float* lp = llr + ((size_t)q * 48 + k) * n_bpsc;
                        llr_of(sym48[k], n_bpsc, lp);
                        if (prm->llr_csi)                       /* spec rule 12: channel-state weight */
                            for (int b = 0; b < n_bpsc; b++) lp[b] = lp[b] * W[bin48[k]];
                    }
                n_out = q + 1;
            }
        }
    }
    fr->n_sym_out = (uint16_t)n_out;
    if (have_signal && n_out == n_sym) fr->flags |= WIFIRX_F_COMPLETE;
}

/* ------------------------------------------------------------------------------------------- */
/* f2: decode_mac, gnu_radio/IRS_AP.py:272                                  */

static uint32_t crc32_ieee(const uint8_t* p, size_t n)
{
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
    }
    return ~c;
}

/* idx: n_sym*48 constellation indices. psdu_out: psdu_len bytes (incl. FCS). returns 1 if CRC ok,
 * 0 if CRC bad, -1 if the frame is too large for decode_mac. */
int orc_decode_mac(const uint8_t* idx, int encoding, int psdu_len, uint8_t* psdu_out)
{
    int n_bpsc = N_BPSC[encoding], n_dbps = N_DBPS[encoding], n_cbps = 48 * n_bpsc;
    int n_sym = (int)ceil((16 + 8 * psdu_len + 6) / (double)n_dbps);
    if (n_sym > WIFIRX_MAX_SYM || psdu_len > WIFIRX_MAX_PSDU) return -1;
    int n_data = n_sym * n_dbps;
    uint8_t* rx    = (uint8_t*)malloc((size_t)n_sym * n_cbps);
    uint8_t* deint = (uint8_t*)malloc((size_t)n_sym * n_cbps);
    uint8_t* dep   = (uint8_t*)malloc((size_t)n_data * 2);
    uint8_t* dec   = (uint8_t*)malloc((size_t)n_data);
    for (int i = 0; i < n_sym * 48; i++)
        for (int k = 0; k < n_bpsc; k++) rx[i * n_bpsc + k] = (idx[i] >> k) & 1;
    /* de-interleave: transmitted position j carries coded bit k */
    int s = n_bpsc / 2 > 1 ? n_bpsc / 2 : 1;
    for (int sym = 0; sym < n_sym; sym++) {
        for (int k = 0; k < n_cbps; k++) {
            int i = (n_cbps / 16) * (k % 16) + k / 16;
            int j = s * (i / s) + (i + n_cbps - (16 * i) / n_cbps) % s;
            deint[sym * n_cbps + k] = rx[sym * n_cbps + j];
        }
    }
    /* de-puncture */
    int pos = 0;
    for (int i = 0; i < 2 * n_data; i++) {
        int keep = 1;
        if (PUNCT[encoding] == 1) keep = (i % 4) != 3;
        else if (PUNCT[encoding] == 2) keep = !((i % 6) == 3 || (i % 6) == 4);
        dep[i] = keep ? deint[pos++] : 2;
    }
    viterbi_decode(dep, n_data, dec);
    /* descramble: state from the first 7 bits (SERVICE bits are zero before scrambling) */
    int state = 0;
    for (int i = 0; i < 7; i++) if (dec[i]) state |= 1 << (6 - i);
    uint8_t* bytes = (uint8_t*)calloc((size_t)psdu_len + 2, 1);
    for (int i = 7; i < psdu_len * 8 + 16; i++) {
        int fb = (!!(state & 64)) ^ (!!(state & 8));
        int bit = fb ^ (dec[i] & 1);
        bytes[i / 8] |= (uint8_t)(bit << (i % 8));
        state = ((state << 1) & 0x7e) | fb;
    }
    memcpy(psdu_out, bytes + 2, (size_t)psdu_len);
    int ok = psdu_len >= 4 && crc32_ieee(bytes + 2, (size_t)psdu_len) == 558161692u;
    free(rx); free(deint); free(dep); free(dec); free(bytes);
    return ok;
}

/* ------------------------------------------------------------------------------------------- */
/* batch driver: every slot is an independent stream, first frame only                         */

int orc_demod_batch(const c32* iq, uint32_t slot_len, uint32_t n_slots, const orc_params* prm,
                    wifirx_frame* frames, uint8_t* idx, float* llr, c32* eq, c32* csi, int n_threads)
{
    long i;
    size_t idx_stride = (size_t)prm->max_sym * 48;
    size_t llr_stride = idx_stride * (size_t)prm->llr_bits;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (i = 0; i < (long)n_slots; i++) {
        const c32* x = iq + (size_t)i * slot_len;
        wifirx_frame* fr = frames + i;
        int32_t trig;
        float   cfo;
        long nt = orc_sync_short(x, slot_len, prm->threshold, prm->min_plateau, prm->math_mode, 1,
                                 &trig, &cfo, 1);
        if (nt == 0) {
            memset(fr, 0, sizeof *fr);
            fr->trigger = -1;
            continue;
        }
        long L = (long)slot_len - (trig - 16);
        if (L > WIFIRX_MAX_SAMPLES) L = WIFIRX_MAX_SAMPLES;
        orc_frame(x, slot_len, trig, cfo, L, prm, fr,
                  idx ? idx + i * idx_stride : NULL,
                  (llr && prm->llr_bits) ? llr + i * llr_stride : NULL,
                  eq ? eq + i * idx_stride : NULL, csi ? csi + (size_t)i * 52 : NULL);
    }
    if (prm->no_pair_fallback == 1) {
        /* upstream's no-pair behaviour, measured: slot order = stream order */
        orc_params p2 = *prm;
        p2.no_pair_fallback = 2;
        p2.dbg_top4 = NULL; p2.dbg_mag4 = NULL;
        float last = 0.0f;
        for (i = 0; i < (long)n_slots; i++) {
            wifirx_frame* fr = frames + i;
            if (fr->flags & WIFIRX_F_SYNC) { last = fr->cfo_fine; continue; }
            if (!(fr->flags & WIFIRX_F_DETECTED) || (fr->flags & WIFIRX_F_TRUNCATED)) continue;
            const c32* x = iq + (size_t)i * slot_len;
            long L = (long)slot_len - (fr->trigger - 16);
            if (L > WIFIRX_MAX_SAMPLES) L = WIFIRX_MAX_SAMPLES;
            p2.fallback_cfo_f = last;
            orc_frame(x, slot_len, fr->trigger, fr->cfo_coarse, L, &p2, fr,
                      idx ? idx + i * idx_stride : NULL,
                      (llr && prm->llr_bits) ? llr + i * llr_stride : NULL,
                      eq ? eq + i * idx_stride : NULL, csi ? csi + (size_t)i * 52 : NULL);
        }
    }
    return 0;
}

/* decode_mac over a demodulated batch; psdu: [n_slots][psdu_stride] */
int orc_decode_batch(uint32_t n_slots, const orc_params* prm, wifirx_frame* frames,
                     const uint8_t* idx, uint8_t* psdu, uint32_t psdu_stride, int n_threads)
{
    long i;
    size_t idx_stride = (size_t)prm->max_sym * 48;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (i = 0; i < (long)n_slots; i++) {
        wifirx_frame* fr = frames + i;
        if (!(fr->flags & WIFIRX_F_COMPLETE)) continue;
        if (fr->psdu_len > psdu_stride) continue;
        int r = orc_decode_mac(idx + i * idx_stride, fr->encoding, fr->psdu_len, psdu + (size_t)i * psdu_stride);
        if (r < 0) continue;
        fr->flags |= WIFIRX_F_DECODED;
        if (r == 1) fr->flags |= WIFIRX_F_CRC_OK;
    }
    return 0;
}

/* stream driver: the whole sync_short state machine over one continuous stream, then every
 * trigger's frame (usable samples end at the next trigger or after MAX_SAMPLES) */
long orc_demod_stream(const c32* x, long n_samp, const orc_params* prm, wifirx_frame* frames,
                      uint8_t* idx, float* llr, c32* eq, long cap)
{
    int32_t* trig = (int32_t*)malloc((size_t)cap * sizeof(int32_t));
    float*   cfo  = (float*)malloc((size_t)cap * sizeof(float));
    long nt = orc_sync_short(x, n_samp, prm->threshold, prm->min_plateau, prm->math_mode, 0, trig, cfo, cap);
    size_t idx_stride = (size_t)prm->max_sym * 48;
    size_t llr_stride = idx_stride * (size_t)prm->llr_bits;
    for (long k = 0; k < nt; k++) {
        long L = n_samp - (trig[k] - 16);
        if (k + 1 < nt && trig[k + 1] - trig[k] < L) L = trig[k + 1] - trig[k];
        if (L > WIFIRX_MAX_SAMPLES) L = WIFIRX_MAX_SAMPLES;
        orc_frame(x, n_samp, trig[k], cfo[k], L, prm, frames + k,
                  idx ? idx + k * idx_stride : NULL,
                  (llr && prm->llr_bits) ? llr + k * llr_stride : NULL,
                  eq ? eq + k * idx_stride : NULL, NULL);
    }
    free(trig);
    free(cfo);
    return nt;
}

/* Spec rule 13: moments of the equalised points of a frame, what a digital.probe_mpsk_snr_est_c fed from the `symbols`
 * port accumulates (gnu_radio/IRS_AP.py:275,312).  eq: [n_sym_out][48] in carrier order.  out4 = sum |y|, sum |y|^2,
 * sum |y|^4, 0 in float32: per symbol the 64 bins (non-data bins 0) are folded as ((a[r] + a[r+16]) + a[r+32]) + a[r+48],
 * r = 0..15, those 16 values by the xor tree (steps 1, 2, 4, 8), and the symbols are added up in order. */
void orc_sym_stats(const c32* eq, int n_sym_out, float* out4)
{
    float s[3] = { 0.0f, 0.0f, 0.0f };
    for (int q = 0; q < n_sym_out; q++) {
        float a[3][64];
        memset(a, 0, sizeof a);
        int k = 0;
        for (int i = 6; i <= 58; i++) {
            if (i == 11 || i == 25 || i == 32 || i == 39 || i == 53) continue;
            const c32 y = eq[(size_t)q * 48 + k++];
            const float m2 = fmaf(y.im, y.im, y.re * y.re);
            a[0][i] = sqrtf(m2); a[1][i] = m2; a[2][i] = m2 * m2;
        }
        for (int w = 0; w < 3; w++) {
            float p[16], t[16];
            for (int r = 0; r < 16; r++) p[r] = ((a[w][r] + a[w][r + 16]) + a[w][r + 32]) + a[w][r + 48];
            for (int step = 1; step < 16; step <<= 1) {
                for (int r = 0; r < 16; r++) t[r] = p[r] + p[r ^ step];
                memcpy(p, t, sizeof p);
            }
            s[w] += p[0];
        }
    }
    out4[0] = s[0]; out4[1] = s[1]; out4[2] = s[2]; out4[3] = 0.0f;
}

/* small entry points for unit tests of the spec routines */
void orc_sincos(const float* x, float* s, float* c, long n) { for (long i = 0; i < n; i++) sp_sincos(x[i], s + i, c + i); }
void orc_atan2(const float* y, const float* x, float* r, long n) { for (long i = 0; i < n; i++) r[i] = sp_atan2(y[i], x[i]); }
void orc_log2(const float* x, float* r, long n) { for (long i = 0; i < n; i++) r[i] = sp_log2(x[i]); }
void orc_rsqrt(const float* x, float* r, long n) { for (long i = 0; i < n; i++) r[i] = sp_rsqrt(x[i]); }
void orc_recip(const float* x, float* r, long n) { for (long i = 0; i < n; i++) r[i] = sp_recip(x[i]); }
void orc_fft64(const c32* in, c32* out, long n, int math_mode)
{
    for (long i = 0; i < n; i++) {
        if (math_mode == ORC_MATH_SPEC) fft64_spec(in + 64 * i, out + 64 * i);
        else fft64_libm(in + 64 * i, out + 64 * i);
    }
}
void orc_viterbi(const uint8_t* coded, int n_bits, uint8_t* out) { viterbi_decode(coded, n_bits, out); }
uint32_t orc_crc32(const uint8_t* p, long n) { return crc32_ieee(p, (size_t)n); }
int orc_decode_signal(const uint8_t* rx48, int* enc, int* len) { return decode_signal(rx48, enc, len); }
