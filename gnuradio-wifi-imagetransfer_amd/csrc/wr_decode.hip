// wr_decode.hip -- decode_mac on the device: replaces ieee802_11.decode_mac
// (gnu_radio/IRS_AP.py:272,291-292): demap indices to bits, de-interleave, de-puncture, Viterbi
// K=7 (133,171), descramble, CRC-32.
//
// One wavefront decodes 128 frames: lane l owns frames base + l ("A", low halves) and base + 64 + l ("B", high
// halves).  The 64 path metrics of both frames live in 64 VGPRs as packed 16-bit pairs, and the
// add-compare-select of all 32 butterflies of a trellis step is straight-line packed arithmetic
// (v_pk_add_u16 / v_pk_min_u16 / v_pk_sub_i16: one instruction works on both frames, no cross-lane traffic);
// metrics are updated in place, which rotates the state <-> register map by one bit per step, so the code is
// unrolled over the 6 steps after which the map is the identity again (all rates have n_data % 12 == 0).
// The survivor bit of a state is the sign of (candidate 1 - candidate 0), shifted into a packed accumulator by
// v_pk_lshrrev_b16 + v_pk_mad_u16; the 64 bits per frame and step leave as one 16-byte store per lane.
// 16-bit metrics: the start penalty of the states != 0 only has to outlast the first six steps (from then on every
// state has a survivor that started in state 0), and the common minimum is subtracted every 120 steps, so values
// stay far below 2^15 and the signed difference orders them; decisions depend on differences only.
// Traceback, descrambling and the CRC run per lane, for its two frames.
//
// The received coded bits are gathered beforehand by decode_gather_kernel (one wave per frame, lane <-> trellis
// step: de-puncture + de-interleave + bit extraction from the hard decisions) into per-frame bit masks in global
// memory, which reach the lane that owns the frame through LDS, one 60-step chunk at a time.
//
// Results are bit-identical to the oracle's viterbi_decode(): same metrics (Hamming, erasures free),
// same tie rule (the survivor with older bit 0 wins), same final state rule (smallest metric, lowest
// state), start state 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wifirx.h"
#include "wr_kernels.h"

namespace wr {

#ifndef WR_DEC_WAVES_PER_SIMD
#define WR_DEC_WAVES_PER_SIMD 4
#endif
#define WR_DEC_CHUNK      60                   // trellis steps per gathered mask word (10 groups of 6)
#define WR_DEC_LDS_WORDS  (4 * WR_DECODE_FRAMES_PER_WAVE)   // per wave, 8-byte words: [A1,AV,B1,BV][frame] of the current chunk

// Where the coded bit at position `ci` of the de-punctured stream of ONE OFDM symbol comes from: carrier (bits 0..5)
// and bit of its decision (bits 6..8), or WR_SRC_PUNCT when the transmitter dropped it.  Every symbol carries
// 2 * n_dbps de-punctured positions and exactly n_cbps transmitted bits, so the map repeats from symbol to symbol.
// Rate parameters are template constants: every division below is by a compile-time constant.
#define WR_SRC_PUNCT 0x200u
template <int PUNCT, int N_BPSC>
__device__ __forceinline__ uint32_t coded_src(int ci)
{
    constexpr int n_cbps = 48 * N_BPSC;
    constexpr int s = (N_BPSC / 2) < 1 ? 1 : (N_BPSC / 2);
    int pidx;
    if (PUNCT == 0) {
        pidx = ci;
    } else if (PUNCT == 1) {               // 2/3: every 4th bit dropped
        int r = ci & 3;
        if (r == 3) return WR_SRC_PUNCT;
        pidx = (ci >> 2) * 3 + r;
    } else {                               // 3/4: bits 3,4 of every 6 dropped
        int g = ci / 6, r = ci - 6 * g;
        if (r == 3 || r == 4) return WR_SRC_PUNCT;
        pidx = g * 4 + (r < 3 ? r : 3);
    }
    const int k = pidx;                    // < n_cbps: first symbol
    int i = (n_cbps >> 4) * (k & 15) + (k >> 4);
    int j = s * (i / s) + (i + n_cbps - (16 * i) / n_cbps) % s;
    int carrier = j / N_BPSC, bit = j - carrier * N_BPSC;
    return (uint32_t)carrier | ((uint32_t)bit << 6);
}

__device__ __forceinline__ uint32_t coded_src_of(int enc, int ci)
{
    switch (enc) {
    case 0:  return coded_src<0, 1>(ci);
    case 1:  return coded_src<2, 1>(ci);
    case 2:  return coded_src<0, 2>(ci);
    case 3:  return coded_src<2, 2>(ci);
    case 4:  return coded_src<0, 4>(ci);
    case 5:  return coded_src<2, 4>(ci);
    case 6:  return coded_src<1, 6>(ci);
    default: return coded_src<2, 6>(ci);
    }
}

#define WR_DEC_TAB_STRIDE 216              // steps per OFDM symbol at the highest rate
// the workgroup's source table: entry [enc][tt] = source of coded bit A (low half) and B (high half) of step tt
__device__ __forceinline__ void build_src_table(uint32_t* tab)
{
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    for (int e = threadIdx.x; e < 8 * WR_DEC_TAB_STRIDE; e += blockDim.x) {
        const int enc = e / WR_DEC_TAB_STRIDE, tt = e - enc * WR_DEC_TAB_STRIDE;
        uint32_t v = WR_SRC_PUNCT | (WR_SRC_PUNCT << 16);
        if (tt < ndbps_tab[enc]) v = coded_src_of(enc, 2 * tt) | (coded_src_of(enc, 2 * tt + 1) << 16);
        tab[e] = v;
    }
}

// The received value (0, 1, or 2 = punctured / beyond the tile) of the coded bits A and B of step t of one frame.
// `tile` holds the decisions of the symbols sym0.. of the frame (48 bytes each, LDS); n_dbps and its reciprocal
// (ceil(2^32 / n_dbps): exact quotients for every t < 2^17) are wave-uniform.
__device__ __forceinline__ void gather_step(const uint8_t* tile, int sym0, const uint32_t* __restrict__ tab_enc,
                                            int n_dbps, uint32_t recip, int t, bool in_range, int& ra, int& rb)
{
    ra = 2; rb = 2;
    if (in_range) {
        const int sym = (int)__umulhi((uint32_t)t, recip);
        const int tt = t - sym * n_dbps;
        const uint32_t e = tab_enc[tt];
        const uint32_t ea = e & 0xffffu, eb = e >> 16;
        const uint8_t* sp = tile + (sym - sym0) * 48;
        if (!(ea & WR_SRC_PUNCT)) ra = (sp[ea & 63u] >> (ea >> 6)) & 1;
        if (!(eb & WR_SRC_PUNCT)) rb = (sp[eb & 63u] >> (eb >> 6)) & 1;
    }
}

// trellis steps of a frame decode_mac accepts (a multiple of 12), 0 for a frame it leaves alone
__device__ __forceinline__ int frame_steps(uint32_t flags, int enc, int len, uint32_t psdu_stride, uint32_t max_sym,
                                           uint32_t n_steps_cap)
{
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    const int n_dbps = ndbps_tab[enc & 7];
    const int n_sym = (16 + 8 * len + 6 + n_dbps - 1) / n_dbps;
    const bool ok = (flags & WIFIRX_F_COMPLETE) && len <= (int)psdu_stride && len <= WIFIRX_MAX_PSDU &&
                    n_sym <= WIFIRX_MAX_SYM && n_sym <= (int)max_sym && (uint32_t)(n_sym * n_dbps) <= n_steps_cap;
    return ok ? n_sym * n_dbps : 0;
}

#define WR_RECIP32(d) (uint32_t)((0x100000000ull + (d) - 1) / (d))      /* ceil(2^32 / d) */

// ---- gather kernel: the received coded bits of every frame as 60-step bit masks, one wave per frame with lane <->
//      trellis step.  The decisions of a frame come in tiles of 60 OFDM symbols (a whole number of chunks at every
//      rate), copied into the wave's LDS by coalesced 16-byte loads.  Output layout: for decode task T (the
//      frames_per_wave frames one wave of decode_kernel handles) [chunk][A1,AV,B1,BV][frame position 0..127]. ----
__global__ __launch_bounds__(256)
void decode_gather_kernel(uint32_t n_slots, uint32_t max_sym, const wifirx_frame* __restrict__ frames,
                          const uint8_t* __restrict__ idx_all, uint32_t psdu_stride, uint64_t* __restrict__ masks_all,
                          uint32_t n_steps_cap, uint32_t frames_per_wave)
{
    __shared__ uint8_t tile_all[4][60 * 48];
    __shared__ uint32_t src_tab[8 * WR_DEC_TAB_STRIDE];
    build_src_table(src_tab);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    uint8_t* tile = tile_all[wv];
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    const uint32_t recip_tab[8] = { WR_RECIP32(24), WR_RECIP32(36), WR_RECIP32(48), WR_RECIP32(72),
                                    WR_RECIP32(96), WR_RECIP32(144), WR_RECIP32(192), WR_RECIP32(216) };
    const bool idx16 = ((reinterpret_cast<uintptr_t>(idx_all) | ((size_t)max_sym * 48)) & 15) == 0;
    const uint32_t fA = frames_per_wave < 64 ? frames_per_wave : 64;
    const size_t task_words = ((size_t)n_steps_cap / WR_DEC_CHUNK + 2) * 4 * WR_DECODE_FRAMES_PER_WAVE;
    for (uint32_t slot = blockIdx.x * 4 + wv; slot < n_slots; slot += gridDim.x * 4) {
        const wifirx_frame fr = frames[slot];
        const int f_enc = fr.encoding & 7;
        const int f_ndata = frame_steps(fr.flags, f_enc, fr.psdu_len, psdu_stride, max_sym, n_steps_cap);
        if (f_ndata == 0) continue;
        const uint32_t task = slot / frames_per_wave, rr = slot - task * frames_per_wave;
        const uint32_t f = rr < fA ? rr : 64 + (rr - fA);           // position in the decode wave: lane f & 63, half f >> 6
        uint64_t* masks = masks_all + (size_t)task * task_words;
        const uint8_t* fidx = idx_all + (size_t)slot * max_sym * 48;
        const int f_ndbps = ndbps_tab[f_enc];
        const uint32_t f_recip = recip_tab[f_enc];
        const int f_nsym = f_ndata / f_ndbps;
        for (int sym0 = 0; sym0 < f_nsym; sym0 += 60) {
            const int nsy = f_nsym - sym0 < 60 ? f_nsym - sym0 : 60;
            const int nbytes = nsy * 48;
            __builtin_amdgcn_wave_barrier();
            if (idx16) {
                for (int o = lane * 16; o < nbytes; o += 1024)
                    *reinterpret_cast<uint4*>(tile + o) = *reinterpret_cast<const uint4*>(fidx + sym0 * 48 + o);
            } else {
                for (int o = lane; o < nbytes; o += 64) tile[o] = fidx[sym0 * 48 + o];
            }
            __builtin_amdgcn_wave_barrier();
            const int t_hi = (sym0 + nsy) * f_ndbps;
            for (int c = sym0 * f_ndbps / WR_DEC_CHUNK; c * WR_DEC_CHUNK < t_hi; c++) {
                const int t = c * WR_DEC_CHUNK + lane;
                int ra, rb;
                gather_step(tile, sym0, src_tab + f_enc * WR_DEC_TAB_STRIDE, f_ndbps, f_recip, t,
                            lane < WR_DEC_CHUNK && t < t_hi, ra, rb);
                const uint64_t A1 = __ballot(ra == 1), AV = __ballot(ra != 2);
                const uint64_t B1 = __ballot(rb == 1), BV = __ballot(rb != 2);
                if (lane < 4) {
                    uint64_t wsel = lane == 0 ? A1 : lane == 1 ? AV : lane == 2 ? B1 : BV;
                    masks[((size_t)c * 4 + lane) * WR_DECODE_FRAMES_PER_WAVE + f] = wsel;
                }
            }
        }
    }
}

constexpr __host__ __device__ int rotr6(int s, int p) { return ((s >> p) | (s << (6 - p))) & 63; }
constexpr __host__ __device__ int parity_of(int v) { return __builtin_popcount(v) & 1; }

// ---- packed 16-bit arithmetic (both halves at once; the compiler may schedule these freely) ----
// Issue cost on gfx950 at 4-8 waves per SIMD (tools/valu_rate.hip): v_pk_* and the three-operand integer forms
// (v_bfi_b32, v_add3_u32, ...) ~2.0 ns per wave-instruction, v_add_u32 1.5 ns, v_lshrrev_b32 1.3 ns.  Hence:
//   * sums of two packed halves are plain 32-bit adds (no half ever reaches 2^15, so no carry crosses);
//   * a survivor bit is the sign of (candidate 1 - candidate 0), and because all path metrics of a frame lie
//     within 255 of each other (start penalty 64, the common minimum leaves every 120 steps) bits 8..15 of
//     that difference all equal its sign: one v_bfi_b32 drops it into any of the bit positions 15..8 of the
//     accumulator, eight decisions per half without a shift, then the accumulator moves down by eight.
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// (d & mask) | (acc & ~mask)
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t d, uint32_t acc)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(d), "v"(acc));
    return r;
}

#define WR_DEC_START_PENALTY 0x00400040u      // metric of the states != 0 at step 0 (per half): only has to outlast six steps

// one trellis step at register phase P: logical state s lives in pm[rotr6(s, P)] (low half: frame A, high: frame B).
// M[a][b]: packed branch metrics of a transition whose expected coded pair is (a, b).
// acc[k]: survivor bits of the states 16k..16k+15, state 16k + i in bit (7 - i) & 15 of each half.
template <int P>
__device__ __forceinline__ void acs_step(uint32_t (&pm)[64], const uint32_t (&M)[2][2], uint32_t (&acc)[4])
{
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int a = parity_of((j << 1) & 0155), b = parity_of((j << 1) & 0117);
        const uint32_t m = M[a][b], mb = M[a ^ 1][b ^ 1];
        const int r0 = rotr6(j, P), r1 = rotr6(j + 32, P);
        const uint32_t p0 = pm[r0], p1 = pm[r1];
        // state 2j (input bit 0): from j with m (candidate 0), from j+32 with mb (candidate 1);  state 2j+1: metrics swapped
        const uint32_t c00 = p0 + m, c01 = p1 + mb, c10 = p0 + mb, c11 = p1 + m;
        uint32_t& w = acc[j >> 3];
        const int i = 2 * (j & 7);                         // decisions i, i + 1 of this word
        if (i == 8) w >>= 8;                               // bits 15..8 are full: make room (what leaks across the halves is overwritten)
        const uint32_t s0 = pk_sub(c01, c00), s1 = pk_sub(c11, c10);      // candidate 1 < candidate 0: survivor from j+32
        if (i == 0) w = s0 & 0x80008000u;
        else        w = bfi(0x80008000u >> (i & 7), s0, w);
        w = bfi(0x80008000u >> ((i & 7) + 1), s1, w);
        pm[r0] = pk_min(c00, c01);      // = register of logical state 2j at phase P+1
        pm[r1] = pk_min(c10, c11);      // = register of logical state 2j+1 at phase P+1
    }
}

__device__ __forceinline__ uint32_t crc32_bit(uint32_t c, uint32_t bit)
{
    uint32_t x = (c ^ bit) & 1u;
    return (c >> 1) ^ (0xedb88320u & (0u - x));
}

// Tables of the per-frame finish (workgroup LDS, built once per workgroup): crc[k][b] = CRC-32 (reflected 0xedb88320)
// of byte b followed by k zero bytes ("slicing by 4"), scr[s] = the next 32 scrambler bits from LFSR state s.
struct FinishTables { uint32_t crc[4][256]; uint32_t scr[128]; };

__device__ __forceinline__ void build_finish_tables(FinishTables& ft)
{
    for (int e = threadIdx.x; e < 256; e += blockDim.x) {
        uint32_t c = (uint32_t)e;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
        ft.crc[0][e] = c;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 256; e += blockDim.x) {
        uint32_t c = ft.crc[0][e];
        for (int k = 1; k < 4; k++) { c = (c >> 8) ^ ft.crc[0][c & 0xffu]; ft.crc[k][e] = c; }
    }
    for (int e = threadIdx.x; e < 128; e += blockDim.x) {
        int state = e;
        uint32_t w = 0;
        for (int k = 0; k < 32; k++) {
            const int fb = ((state >> 6) ^ (state >> 3)) & 1;
            state = ((state << 1) & 0x7e) | fb;
            w |= (uint32_t)fb << k;
        }
        ft.scr[e] = w;
    }
    __syncthreads();
}

// descramble (x^7+x^4+1, state from the first 7 decoded bits), bytes, CRC-32 of one frame; db = its decoded words
// (stride 128 dwords, two spare words behind the last one).  Four PSDU bytes per iteration: the 32 decoded bits from
// position 16 + 32 k on (a funnel shift of two decoded words), the 32 scrambler bits from the table (the state after
// them is their last seven, reversed), CRC by four table look-ups.  Bytes leave four at a time when the row is
// dword-aligned (wave-uniform `dword_ok`).
__device__ __forceinline__ void finish_frame(const uint32_t* __restrict__ db, int psdu_len, uint8_t* __restrict__ psdu,
                                             bool dword_ok, wifirx_frame* __restrict__ rec, uint32_t flags,
                                             const FinishTables& ft)
{
    uint32_t cur = db[0];
    int state = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) state |= (int)((cur >> i) & 1) << (6 - i);
    // positions 7..15 belong to the SERVICE field: advance the scrambler
#pragma unroll
    for (int i = 7; i < 16; i++) {
        int fb = ((state >> 6) ^ (state >> 3)) & 1;
        state = ((state << 1) & 0x7e) | fb;
    }
    uint32_t crc = 0xffffffffu;
    uint32_t nxt = db[128];
    const int n_words = psdu_len >> 2;
    for (int k = 0; k < n_words; k++) {
        const uint32_t nn = db[(size_t)(k + 2) * 128];                     // spare words behind the last one keep this in range
        const uint32_t sc = ft.scr[state];
        state = (int)(__builtin_bitreverse32(sc) & 0x7fu);
        const uint32_t d = __builtin_amdgcn_alignbit(nxt, cur, 16) ^ sc;    // positions 16 + 32 k .. + 31, descrambled
        cur = nxt; nxt = nn;
        if (dword_ok) *reinterpret_cast<uint32_t*>(psdu + 4 * k) = d;
        else { psdu[4 * k] = (uint8_t)d; psdu[4 * k + 1] = (uint8_t)(d >> 8); psdu[4 * k + 2] = (uint8_t)(d >> 16); psdu[4 * k + 3] = (uint8_t)(d >> 24); }
        const uint32_t x = crc ^ d;
        crc = ft.crc[3][x & 0xffu] ^ ft.crc[2][(x >> 8) & 0xffu] ^ ft.crc[1][(x >> 16) & 0xffu] ^ ft.crc[0][x >> 24];
    }
    {   // the last one to three bytes
        const uint32_t sc = ft.scr[state];
        const uint32_t d = __builtin_amdgcn_alignbit(nxt, cur, 16) ^ sc;
        for (int b = 4 * n_words; b < psdu_len; b++) {
            const uint32_t byte = (d >> (8 * (b & 3))) & 0xffu;
            psdu[b] = (uint8_t)byte;
            crc = (crc >> 8) ^ ft.crc[0][(crc ^ byte) & 0xffu];
        }
    }
    crc = ~crc;
    uint32_t fl = flags | WIFIRX_F_DECODED;
    if (psdu_len >= 4 && crc == 558161692u) fl |= WIFIRX_F_CRC_OK; else fl &= ~WIFIRX_F_CRC_OK;
    rec->flags = fl;
}

__global__ __launch_bounds__(256, WR_DEC_WAVES_PER_SIMD)
void decode_kernel(uint32_t n_slots, uint32_t max_sym, wifirx_frame* __restrict__ frames,
                   const uint8_t* __restrict__ idx_all, uint8_t* __restrict__ psdu_all, uint32_t psdu_stride,
                   uint8_t* __restrict__ scratch, size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves_total,
                   uint32_t frames_per_wave, const uint64_t* __restrict__ masks_all)
{
    __shared__ uint64_t lds_all[4][WR_DEC_LDS_WORDS];
    __shared__ FinishTables ft;
    build_finish_tables(ft);
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4 + wv;
    if (wave >= n_waves_total) return;
    uint64_t* lds = lds_all[wv];
    const size_t n_data_cap = n_steps_cap;               // trellis steps the scratch slice of a wave holds
    uint32_t* surv = reinterpret_cast<uint32_t*>(scratch + (size_t)wave * scratch_stride);   // [step][lane][4 pieces]
    uint32_t* dbits = surv + n_data_cap * 256;                                               // [word][A/B][lane]
    const uint32_t k1 = 0x00010001u;

    // A wave's tasks (frames_per_wave <= 128 frames each, grid-stride); lane l owns frames base + l (l < fA) and
    // base + fA + l (l < fB).  The coded-bit masks of every task were written by decode_gather_kernel.
    const uint32_t fA = frames_per_wave < 64 ? frames_per_wave : 64, fB = frames_per_wave - fA;
    const size_t task_words = ((size_t)n_data_cap / WR_DEC_CHUNK + 2) * 4 * WR_DECODE_FRAMES_PER_WAVE;
    const uint32_t n_tasks = (n_slots + frames_per_wave - 1) / frames_per_wave;
    for (uint32_t task = wave; task < n_tasks; task += n_waves_total) {
        const uint32_t base = task * frames_per_wave;
        const uint64_t* masks = masks_all + (size_t)task * task_words;
        // ---- my two frames ----
        int n_data[2];
        int n_max = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t slot = base + (h ? fA : 0u) + lane;
            n_data[h] = 0;
            if ((uint32_t)lane < (h ? fB : fA) && slot < n_slots)
                n_data[h] = frame_steps(frames[slot].flags, frames[slot].encoding, frames[slot].psdu_len, psdu_stride, max_sym,
                                        n_steps_cap);
            n_max = n_data[h] > n_max ? n_data[h] : n_max;
        }
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            int o = __shfl_xor(n_max, k, 64);
            n_max = o > n_max ? o : n_max;
        }
        if (n_max == 0) continue;

        // ---- phase 2: add-compare-select ----
        uint32_t pm[64];
#pragma unroll
        for (int s = 0; s < 64; s++) pm[s] = (s == 0) ? 0u : WR_DEC_START_PENALTY;
        int best[2] = { 0, 0 };                         // final states, taken when the frames end
        for (int t0 = 0, c = 0; t0 < n_max; t0 += WR_DEC_CHUNK, c++) {
            // my two frames' mask words of this chunk: global -> LDS (lane-private slots), re-read six steps at a time
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int w = 0; w < 4; w++) {
                lds[w * WR_DECODE_FRAMES_PER_WAVE + lane] = masks[((size_t)c * 4 + w) * WR_DECODE_FRAMES_PER_WAVE + lane];
                lds[w * WR_DECODE_FRAMES_PER_WAVE + 64 + lane] = masks[((size_t)c * 4 + w) * WR_DECODE_FRAMES_PER_WAVE + 64 + lane];
            }
            __builtin_amdgcn_wave_barrier();
            if ((c & 1) == 0 && c > 0) {
                // every 120 steps: subtract the common minimum of each frame (register phase 0 here; decisions see
                // differences only)
                uint32_t mn = pm[0];
#pragma unroll
                for (int s = 1; s < 64; s++) mn = pk_min(mn, pm[s]);
#pragma unroll
                for (int s = 0; s < 64; s++) pm[s] = pk_sub(pm[s], mn);
            }
            {
                for (int g = 0; g < WR_DEC_CHUNK / 6; g++) {
                    const int tg = t0 + 6 * g;
                    if (tg >= n_max) break;
                    // six steps of the four mask words, frame A in bits 0..5, frame B in bits 16..21 (re-read from LDS
                    // every group: registers are what this kernel is short of)
                    uint32_t pw[4];
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const uint64_t wa = lds[w * WR_DECODE_FRAMES_PER_WAVE + lane];
                        const uint64_t wb = lds[w * WR_DECODE_FRAMES_PER_WAVE + 64 + lane];
                        pw[w] = ((uint32_t)(wa >> (6 * g)) & 0x3fu) | (((uint32_t)(wb >> (6 * g)) & 0x3fu) << 16);
                    }
                    const bool mine0 = tg < n_data[0], mine1 = tg < n_data[1];
                    if (mine0 || mine1) {
#define WR_ACS(P)                                                                                         \
                        {                                                                                 \
                            const uint32_t ta = (pw[0] >> P) & k1, va = (pw[1] >> P) & k1;                \
                            const uint32_t tb = (pw[2] >> P) & k1, vb = (pw[3] >> P) & k1;                \
                            const uint32_t nv = va + vb;          /* a set bit implies its valid bit */   \
                            uint32_t M[2][2];                                                             \
                            M[0][0] = ta + tb;                                                            \
                            M[0][1] = ta + vb - tb;                                                       \
                            M[1][1] = nv - M[0][0];                                                       \
                            M[1][0] = nv - M[0][1];                                                       \
                            uint32_t acc[4];                                                              \
                            acs_step<P>(pm, M, acc);                                             \
                            *reinterpret_cast<uint4*>(surv + ((size_t)(tg + P) * 64 + lane) * 4) =        \
                                make_uint4(acc[0], acc[1], acc[2], acc[3]);                               \
                        }
                        WR_ACS(0) WR_ACS(1) WR_ACS(2) WR_ACS(3) WR_ACS(4) WR_ACS(5)
#undef WR_ACS
                    }
                    const bool end0 = mine0 && tg + 6 == n_data[0], end1 = mine1 && tg + 6 == n_data[1];
                    if (__any(end0 || end1)) {
                        // a frame just ended (register phase 0 again): smallest metric, lowest state
                        uint32_t bm0 = pm[0] & 0xffffu, bm1 = pm[0] >> 16;
                        int bs0 = 0, bs1 = 0;
#pragma unroll
                        for (int s = 1; s < 64; s++) {
                            const uint32_t v0 = pm[s] & 0xffffu, v1 = pm[s] >> 16;
                            if (v0 < bm0) { bm0 = v0; bs0 = s; }
                            if (v1 < bm1) { bm1 = v1; bs1 = s; }
                        }
                        if (end0) best[0] = bs0;
                        if (end1) best[1] = bs1;
                    }
                }
            }
        }
        __threadfence_block();
        // ---- traceback of both frames: 32 decoded bits per word, words stored [word][A/B][lane].  The survivor
        //      row of a step does not depend on the path (only the piece picked from it does), so the rows are
        //      loaded ahead of the dependent state updates.
        //      Fast form, when both frames of every lane run the full n_max steps: blocks of 96 steps = three whole
        //      words.  Before step t the state holds the decoded bits u_t .. u_(t-5) in its bits 0..5, so the bits
        //      leave six at a time (one bit reversal and one shift-or at a compile-time offset per six steps), and a
        //      step is the survivor-bit pick plus two instructions.
        //      General form (frames of different lengths in the wave; the steps above the last multiple of 96): one
        //      decoded bit per step, every access predicated on the frame's own length. ----
        {
            int st0 = best[0], st1 = best[1];
            uint32_t word0 = 0, word1 = 0;
            const bool uni = __all(n_data[0] == n_max && n_data[1] == n_max);
            const int n_fast = uni ? (n_max / 96) * 96 : 0;
            for (int t1 = n_max - 1; t1 >= n_fast; t1 -= 16) {
                uint4 rows[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int t = t1 - k;
                    rows[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (t >= n_fast && (t < n_data[0] || t < n_data[1]))
                        rows[k] = *reinterpret_cast<const uint4*>(surv + ((size_t)t * 64 + lane) * 4);
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int t = t1 - k;
                    const bool mine0 = t >= n_fast && t < n_data[0], mine1 = t >= n_fast && t < n_data[1];
                    const int i0 = st0 >> 4, i1 = st1 >> 4;
                    const uint32_t p0 = i0 == 0 ? rows[k].x : i0 == 1 ? rows[k].y : i0 == 2 ? rows[k].z : rows[k].w;
                    const uint32_t p1 = i1 == 0 ? rows[k].x : i1 == 1 ? rows[k].y : i1 == 2 ? rows[k].z : rows[k].w;
                    const uint32_t h0 = (p0 >> ((7 - st0) & 15)) & 1u, h1 = (p1 >> (16 + ((7 - st1) & 15))) & 1u;
                    if (mine0) {
                        word0 |= (uint32_t)(st0 & 1) << (t & 31);
                        st0 = (st0 >> 1) | (int)(h0 << 5);
                        if ((t & 31) == 0) { dbits[(size_t)(t >> 5) * 128 + lane] = word0; word0 = 0; }
                    }
                    if (mine1) {
                        word1 |= (uint32_t)(st1 & 1) << (t & 31);
                        st1 = (st1 >> 1) | (int)(h1 << 5);
                        if ((t & 31) == 0) { dbits[(size_t)(t >> 5) * 128 + 64 + lane] = word1; word1 = 0; }
                    }
                }
            }
            for (int blk = n_fast / 96 - 1; blk >= 0; blk--) {
                uint32_t a0[3] = { 0u, 0u, 0u }, a1[3] = { 0u, 0u, 0u };
                const uint32_t* srow = surv + ((size_t)(blk * 96) * 64 + lane) * 4;
#pragma unroll
                for (int sub = 7; sub >= 0; sub--) {
                    uint4 rows[12];
#pragma unroll
                    for (int k = 0; k < 12; k++) rows[k] = *reinterpret_cast<const uint4*>(srow + (size_t)(sub * 12 + k) * 256);
#pragma unroll
                    for (int gg = 1; gg >= 0; gg--) {
                        const int off = 6 * (2 * sub + gg), w = off >> 5, o = off & 31;      // compile-time after unrolling
                        const uint32_t v0 = __builtin_bitreverse32((uint32_t)st0) >> 26, v1 = __builtin_bitreverse32((uint32_t)st1) >> 26;
                        a0[w] |= v0 << o;
                        a1[w] |= v1 << o;
                        if (o > 26) { a0[w + 1] |= v0 >> (32 - o); a1[w + 1] |= v1 >> (32 - o); }
#pragma unroll
                        for (int q = 5; q >= 0; q--) {
                            const uint4 r = rows[6 * gg + q];
                            const uint32_t l0 = (st0 & 16) ? r.y : r.x, u0 = (st0 & 16) ? r.w : r.z, p0 = (st0 & 32) ? u0 : l0;
                            const uint32_t l1 = (st1 & 16) ? r.y : r.x, u1 = (st1 & 16) ? r.w : r.z, p1 = (st1 & 32) ? u1 : l1;
                            const uint32_t h0 = __builtin_amdgcn_ubfe(p0, (uint32_t)((7 - st0) & 15), 1u);
                            const uint32_t h1 = __builtin_amdgcn_ubfe(p1, (uint32_t)(16 + ((7 - st1) & 15)), 1u);
                            st0 = (st0 >> 1) | (int)(h0 << 5);
                            st1 = (st1 >> 1) | (int)(h1 << 5);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // keep the next twelve rows' loads behind these steps (registers)
                }
#pragma unroll
                for (int w = 0; w < 3; w++) {
                    dbits[(size_t)(blk * 3 + w) * 128 + lane] = a0[w];
                    dbits[(size_t)(blk * 3 + w) * 128 + 64 + lane] = a1[w];
                }
            }
        }
        __threadfence_block();
        // ---- descramble, bytes, CRC-32 ----
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (n_data[h] > 0) {                               // the record is re-read: nothing of it was kept in registers
                const uint32_t slot = base + (h ? fA : 0u) + lane;
                finish_frame(dbits + 64 * h + lane, frames[slot].psdu_len, psdu_all + (size_t)slot * psdu_stride,
                             ((reinterpret_cast<uintptr_t>(psdu_all) | psdu_stride) & 3) == 0, frames + slot, frames[slot].flags, ft);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The low-latency variant for small batches (stream mode: a few hundred frames per push): one wave per frame,
// lane <-> trellis state.  An add-compare-select step is two cross-lane reads (ds_bpermute), a handful of vector
// instructions and one ballot that yields the 64 survivor bits of the step; a frame takes ~0.2 ms instead of the
// ~4 ms a lone wave of decode_kernel needs for its 2 x 64 frames' worth of butterflies, and hundreds of frames run
// side by side.  Same metrics, tie rule and final-state rule, hence the same bytes.
__device__ __forceinline__ uint32_t crc32_update(uint32_t c, uint32_t byte)
{
    c ^= byte;
#pragma unroll
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
    return c;
}

__global__ __launch_bounds__(256)
void decode_small_kernel(uint32_t n_slots, uint32_t max_sym, wifirx_frame* __restrict__ frames,
                         const uint8_t* __restrict__ idx_all, uint8_t* __restrict__ psdu_all, uint32_t psdu_stride,
                         uint8_t* __restrict__ scratch, size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves_total)
{
    __shared__ uint8_t tile_all[4][60 * 48];
    __shared__ uint32_t src_tab[8 * WR_DEC_TAB_STRIDE];
    build_src_table(src_tab);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4 + wv;
    if (wave >= n_waves_total) return;
    uint8_t* tile = tile_all[wv];
    uint64_t* dec = reinterpret_cast<uint64_t*>(scratch + (size_t)wave * scratch_stride);     // survivor word per step
    uint64_t* words = dec + n_steps_cap;                                                      // decoded bits, 60 per word

    // trellis constants of state `lane`: predecessors p0 = s >> 1 and p1 = p0 | 32, input bit s & 1
    const int s = lane, u = s & 1, p0 = s >> 1, p1 = (s >> 1) | 32;
    const int f0 = (p0 << 1) | u;
    const int a0 = __builtin_popcount(f0 & 0155) & 1, b0 = __builtin_popcount(f0 & 0117) & 1;
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
#define WR_RECIP32(d) (uint32_t)((0x100000000ull + (d) - 1) / (d))
    const uint32_t recip_tab[8] = { WR_RECIP32(24), WR_RECIP32(36), WR_RECIP32(48), WR_RECIP32(72),
                                    WR_RECIP32(96), WR_RECIP32(144), WR_RECIP32(192), WR_RECIP32(216) };
#undef WR_RECIP32
    const bool idx16 = ((reinterpret_cast<uintptr_t>(idx_all) | ((size_t)max_sym * 48)) & 15) == 0;

    for (uint32_t slot = wave; slot < n_slots; slot += n_waves_total) {
        const wifirx_frame fr = frames[slot];
        const int enc = fr.encoding & 7, psdu_len = fr.psdu_len;
        const int n_dbps = ndbps_tab[enc];
        const int n_sym = (16 + 8 * psdu_len + 6 + n_dbps - 1) / n_dbps;
        const bool ok = (fr.flags & WIFIRX_F_COMPLETE) && psdu_len <= (int)psdu_stride && psdu_len <= WIFIRX_MAX_PSDU &&
                        n_sym <= WIFIRX_MAX_SYM && n_sym <= (int)max_sym && (uint32_t)(n_sym * n_dbps) <= n_steps_cap;
        if (!ok) continue;                                       // wave-uniform: one frame per wave
        const int n_data = n_sym * n_dbps;
        const uint8_t* fidx = idx_all + (size_t)slot * max_sym * 48;
        const uint32_t recip = recip_tab[enc];
        const uint32_t* tab_enc = src_tab + enc * WR_DEC_TAB_STRIDE;

        // ---- add-compare-select: tiles of 60 symbols, chunks of 60 trellis steps ----
        int pm = (s == 0) ? 0 : (1 << 24);
        for (int sym0 = 0; sym0 < n_sym; sym0 += 60) {
            const int nsy = n_sym - sym0 < 60 ? n_sym - sym0 : 60;
            const int nbytes = nsy * 48;
            __builtin_amdgcn_wave_barrier();
            if (idx16) {
                for (int o = lane * 16; o < nbytes; o += 1024)
                    *reinterpret_cast<uint4*>(tile + o) = *reinterpret_cast<const uint4*>(fidx + sym0 * 48 + o);
            } else {
                for (int o = lane; o < nbytes; o += 64) tile[o] = fidx[sym0 * 48 + o];
            }
            __builtin_amdgcn_wave_barrier();
            const int t_hi = (sym0 + nsy) * n_dbps;
            for (int t0 = sym0 * n_dbps; t0 < t_hi; t0 += WR_DEC_CHUNK) {
                const int t = t0 + lane;
                int ra, rb;
                gather_step(tile, sym0, tab_enc, n_dbps, recip, t, lane < WR_DEC_CHUNK && t < t_hi, ra, rb);
                const uint64_t A1 = __ballot(ra == 1), AV = __ballot(ra != 2);
                const uint64_t B1 = __ballot(rb == 1), BV = __ballot(rb != 2);
                const int jn = t_hi - t0 < WR_DEC_CHUNK ? t_hi - t0 : WR_DEC_CHUNK;
                uint64_t mydec = 0;
                for (int j = 0; j < jn; j++) {
                    const int sa = (int)((A1 >> j) & 1), va = (int)((AV >> j) & 1);
                    const int sb = (int)((B1 >> j) & 1), vb = (int)((BV >> j) & 1);
                    const int bm0 = (va & (sa ^ a0)) + (vb & (sb ^ b0));
                    const int bm1 = (va + vb) - bm0;
                    const int m0 = __shfl(pm, p0, 64) + bm0;
                    const int m1 = __shfl(pm, p1, 64) + bm1;
                    const bool sel = m1 < m0;
                    pm = sel ? m1 : m0;
                    const uint64_t d = __ballot(sel);
                    if (lane == j) mydec = d;
                }
                if (lane < jn) dec[t] = mydec;
            }
        }
        // ---- best final state: smallest metric, lowest state on ties ----
        int key = (pm << 6) | s;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            int o = __shfl_xor(key, k, 64);
            key = o < key ? o : key;
        }
        int st = key & 63;
        __threadfence_block();
        // ---- traceback, one chunk of survivor words in registers at a time; decoded bits 60 per word ----
        const int n_chunks = (n_data + WR_DEC_CHUNK - 1) / WR_DEC_CHUNK;
        for (int c = n_chunks - 1; c >= 0; c--) {
            const int t = c * WR_DEC_CHUNK + lane;
            const uint64_t dw = (lane < WR_DEC_CHUNK && t < n_data) ? dec[t] : 0;
            const uint32_t lo = (uint32_t)dw, hi = (uint32_t)(dw >> 32);
            const int jn = n_data - c * WR_DEC_CHUNK < WR_DEC_CHUNK ? n_data - c * WR_DEC_CHUNK : WR_DEC_CHUNK;
            uint64_t word = 0;
            for (int j = jn - 1; j >= 0; j--) {
                word |= (uint64_t)(st & 1) << j;
                const uint32_t dlo = (uint32_t)__builtin_amdgcn_readlane((int)lo, j);
                const uint32_t dhi = (uint32_t)__builtin_amdgcn_readlane((int)hi, j);
                const uint32_t h = (st < 32 ? (dlo >> st) : (dhi >> (st - 32))) & 1u;
                st = (st >> 1) | (int)(h << 5);
            }
            if (lane == 0) words[c] = word;
        }
        __threadfence_block();
        // ---- descramble: x^7+x^4+1, state from the first 7 decoded bits; the feedback sequence has period 127 ----
        const uint64_t w0 = words[0];
        int state = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) state |= (int)((w0 >> i) & 1) << (6 - i);
        uint64_t seq_lo = 0, seq_hi = 0;             // feedback bit for decoded positions 7, 8, ...
        for (int i = 0; i < 127; i++) {
            const int fb = ((state >> 6) ^ (state >> 3)) & 1;
            if (i < 64) seq_lo |= (uint64_t)fb << i; else seq_hi |= (uint64_t)fb << (i - 64);
            state = ((state << 1) & 0x7e) | fb;
        }
        uint8_t* psdu = psdu_all + (size_t)slot * psdu_stride;
        uint32_t crc = 0xffffffffu;
        for (int b0_ = 0; b0_ < psdu_len; b0_ += 64) {
            const int b = b0_ + lane;
            unsigned byte = 0;
            if (b < psdu_len) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = 16 + 8 * b + k;
                    const int q = (i - 7) % 127;
                    const unsigned fb = (unsigned)(((q < 64 ? (seq_lo >> q) : (seq_hi >> (q - 64)))) & 1);
                    const int wi = i / WR_DEC_CHUNK, wb = i - wi * WR_DEC_CHUNK;
                    const unsigned d = (unsigned)((words[wi] >> wb) & 1);
                    byte |= (d ^ fb) << k;
                }
                psdu[b] = (uint8_t)byte;
            }
            // CRC-32 over the PSDU incl. FCS (residue 0x2144DF1C): the 64 bytes of this pass in order
            const int jn = psdu_len - b0_ < 64 ? psdu_len - b0_ : 64;
            for (int j = 0; j < jn; j++) crc = crc32_update(crc, (uint32_t)__builtin_amdgcn_readlane((int)byte, j));
        }
        crc = ~crc;
        if (lane == 0) {
            uint32_t fl = fr.flags | WIFIRX_F_DECODED;
            if (psdu_len >= 4 && crc == 558161692u) fl |= WIFIRX_F_CRC_OK; else fl &= ~WIFIRX_F_CRC_OK;
            frames[slot].flags = fl;
        }
    }
}

// longest trellis (in steps) among the frames decode_kernel would accept
__global__ __launch_bounds__(256)
void decode_maxsteps_kernel(uint32_t n_slots, uint32_t max_sym, const wifirx_frame* __restrict__ frames,
                            uint32_t psdu_stride, uint32_t* __restrict__ out)
{
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    uint32_t best = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += gridDim.x * blockDim.x) {
        const wifirx_frame fr = frames[i];
        const int nd = ndbps_tab[fr.encoding & 7], len = fr.psdu_len;
        const int n_sym = (16 + 8 * len + 6 + nd - 1) / nd;
        if ((fr.flags & WIFIRX_F_COMPLETE) && len <= (int)psdu_stride && len <= WIFIRX_MAX_PSDU &&
            n_sym <= WIFIRX_MAX_SYM && n_sym <= (int)max_sym) {
            uint32_t v = (uint32_t)(n_sym * nd);
            best = v > best ? v : best;
        }
    }
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        uint32_t o = __shfl_xor(best, k, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}

}  // namespace wr

extern "C" hipError_t wr_launch_decode_maxsteps(hipStream_t st, uint32_t n_slots, uint32_t max_sym,
                                                const wifirx_frame* frames, uint32_t psdu_stride, uint32_t* out)
{
    if (n_slots == 0) return hipSuccess;
    uint32_t blocks = (n_slots + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(wr::decode_maxsteps_kernel, dim3(blocks), dim3(256), 0, st, n_slots, max_sym, frames, psdu_stride, out);
    return hipGetLastError();
}

extern "C" hipError_t wr_launch_decode(hipStream_t st, uint32_t n_slots, uint32_t max_sym, wifirx_frame* frames,
                                       const uint8_t* idx, uint8_t* psdu, uint32_t psdu_stride, uint8_t* scratch,
                                       size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves, uint32_t frames_per_wave,
                                       uint64_t* masks)
{
    if (n_slots == 0 || n_waves == 0) return hipSuccess;
    // the coded-bit masks of all frames first (one wave per frame, as many as fit), then the decoder proper
    uint32_t gblocks = (n_slots + 3) / 4;
    if (gblocks > 16384) gblocks = 16384;
    hipLaunchKernelGGL(wr::decode_gather_kernel, dim3(gblocks), dim3(256), 0, st, n_slots, max_sym, frames, idx, psdu_stride,
                       masks, n_steps_cap, frames_per_wave);
    uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL(wr::decode_kernel, dim3(blocks), dim3(256), 0, st, n_slots, max_sym, frames, idx, psdu,
                       psdu_stride, scratch, scratch_stride, n_steps_cap, n_waves, frames_per_wave, masks);
    return hipGetLastError();
}

extern "C" hipError_t wr_launch_decode_small(hipStream_t st, uint32_t n_slots, uint32_t max_sym, wifirx_frame* frames,
                                             const uint8_t* idx, uint8_t* psdu, uint32_t psdu_stride, uint8_t* scratch,
                                             size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves)
{
    if (n_slots == 0 || n_waves == 0) return hipSuccess;
    hipLaunchKernelGGL(wr::decode_small_kernel, dim3((n_waves + 3) / 4), dim3(256), 0, st, n_slots, max_sym, frames, idx, psdu,
                       psdu_stride, scratch, scratch_stride, n_steps_cap, n_waves);
    return hipGetLastError();
}
