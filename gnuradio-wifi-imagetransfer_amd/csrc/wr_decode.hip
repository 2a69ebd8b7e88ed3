// wr_decode.hip -- decode_mac on the device: replaces ieee802_11.decode_mac
// (gnu_radio/IRS_AP.py:272,291-292): demap indices to bits, de-interleave, de-puncture, Viterbi
// K=7 (133,171), descramble, CRC-32.
//
// One wavefront decodes 128 frames: lane l owns frames base + l ("A", low halves) and base + 64 + l ("B", high
// halves).  The 64 path metrics of both frames live in 64 VGPRs as packed 16-bit pairs, and the
// add-compare-select of all 32 butterflies of a trellis step is straight-line packed arithmetic
// (32-bit adds / v_pk_min_u16 / v_pk_sub_i16: one instruction works on both frames, no cross-lane traffic);
// metrics are updated in place, which rotates the state <-> register map by one bit per step, so the code is
// unrolled over the 6 steps after which the map is the identity again (all rates have n_data % 12 == 0).
// The survivor bit of a state is the sign of (candidate 1 - candidate 0), dropped into a packed accumulator by one
// v_bfi_b32 (see acs_step); the 64 bits per frame and step leave as one 16-byte store per lane.
// 16-bit metrics: the start penalty of the states != 0 only has to outlast the first six steps (from then on every
// state has a survivor that started in state 0), and the common minimum is subtracted every 120 steps, so values
// stay far below 2^15 and the signed difference orders them; decisions depend on differences only.
// Traceback, descrambling and the CRC run per lane, for its two frames.
//
// The received coded bits come as the bit planes of the hard decisions (wifirx_out.hbits: written by the demod
// kernels, or packed from `idx` by decode_pack_kernel): each lane copies the words of the current OFDM symbol of its
// two frames into lane-private LDS slots and picks the coded bits of a trellis step from there -- where a coded bit
// sits (de-puncturing + de-interleaving) is a per-rate table of one OFDM symbol, read through the scalar unit when all
// frames of the wave share a rate.
//
// Results are bit-identical to the oracle's viterbi_decode(): same metrics (Hamming, erasures free),
// same tie rule (the survivor with older bit 0 wins), same final state rule (smallest metric, lowest
// state), start state 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "wifirx.h"
#include "wr_kernels.h"

namespace wr {

#ifndef WR_DEC_WAVES_PER_SIMD
#define WR_DEC_WAVES_PER_SIMD 4
#endif
#define WR_DEC_CHUNK      60                   // decode_small_kernel: trellis steps per pass (lane <-> step)
#define WR_DEC_NORM_STEPS 120                  // decode_kernel: the common minimum leaves the metrics every so many steps

// Where the coded bit at position `ci` of the de-punctured stream of ONE OFDM symbol comes from, in the bit-plane
// form of the decisions (wifirx_out.hbits: word 2 b + h of a symbol = bit b of bins 32 h .. 32 h + 31): bits 0..4 the
// bit of the word (= bin & 31), bits 5..8 the word, or WR_SRC_PUNCT when the transmitter dropped the coded bit.
// Read as 16-bit planes (plane p = 4 b + (bin >> 4), bit bin & 15) the same entry is p = bits 4..8, bit = bits 0..3.
// Every symbol carries 2 * n_dbps de-punctured positions and exactly n_cbps transmitted bits, so the map repeats from
// symbol to symbol.
#define WR_SRC_PUNCT 0x200u
#define WR_DEC_TAB_STRIDE 216              // steps per OFDM symbol at the highest rate
constexpr int bin_of_carrier(int c)         // data carrier 0..47 -> FFT bin, shifted order (bin 32 = DC)
{
    int i = c + 6;
    if (i >= 11) i++;
    if (i >= 25) i++;
    if (i >= 32) i++;
    if (i >= 39) i++;
    if (i >= 53) i++;
    return i;
}
constexpr uint32_t coded_src(int punct, int n_bpsc, int ci)
{
    const int n_cbps = 48 * n_bpsc;
    const int s = (n_bpsc / 2) < 1 ? 1 : (n_bpsc / 2);
    int pidx = ci;
    if (punct == 1) {                      // 2/3: every 4th bit dropped
        const int r = ci & 3;
        if (r == 3) return WR_SRC_PUNCT;
        pidx = (ci >> 2) * 3 + r;
    } else if (punct == 2) {               // 3/4: bits 3,4 of every 6 dropped
        const int g = ci / 6, r = ci - 6 * g;
        if (r == 3 || r == 4) return WR_SRC_PUNCT;
        pidx = g * 4 + (r < 3 ? r : 3);
    }
    const int k = pidx;                    // < n_cbps: first symbol
    const int i = (n_cbps >> 4) * (k & 15) + (k >> 4);
    const int j = s * (i / s) + (i + n_cbps - (16 * i) / n_cbps) % s;
    const int carrier = j / n_bpsc, bit = j - carrier * n_bpsc;
    const int bin = bin_of_carrier(carrier);
    return (uint32_t)(bin & 31) | ((uint32_t)(2 * bit + (bin >> 5)) << 5);
}
struct SrcTable { uint32_t e[8 * WR_DEC_TAB_STRIDE]; };    // [enc][step of the symbol]: coded bit A in the low half, B in the high half
constexpr SrcTable make_src_table()
{
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    const int punct_tab[8] = { 0, 2, 0, 2, 0, 2, 1, 2 };
    const int nbpsc_tab[8] = { 1, 1, 2, 2, 4, 4, 6, 6 };
    SrcTable t{};
    for (int enc = 0; enc < 8; enc++)
        for (int tt = 0; tt < WR_DEC_TAB_STRIDE; tt++) {
            uint32_t v = WR_SRC_PUNCT | (WR_SRC_PUNCT << 16);
            if (tt < ndbps_tab[enc])
                v = coded_src(punct_tab[enc], nbpsc_tab[enc], 2 * tt) | (coded_src(punct_tab[enc], nbpsc_tab[enc], 2 * tt + 1) << 16);
            t.e[enc * WR_DEC_TAB_STRIDE + tt] = v;
        }
    return t;
}
__constant__ const SrcTable WR_SRC_TABLE = make_src_table();

// the workgroup's copy in LDS (lane-dependent look-ups)
__device__ __forceinline__ void copy_src_table(uint32_t* tab)
{
    for (int e = threadIdx.x; e < 8 * WR_DEC_TAB_STRIDE; e += blockDim.x) tab[e] = WR_SRC_TABLE.e[e];
}

// The received value (0, 1, or 2 = punctured / beyond the tile) of the coded bits A and B of step t of one frame.
// `tile` holds the bit-plane words of the symbols sym0.. of the frame (n_words each, LDS); n_dbps and its reciprocal
// (ceil(2^32 / n_dbps): exact quotients for every t < 2^17) are wave-uniform.
__device__ __forceinline__ void gather_step(const uint32_t* tile, int sym0, int n_words, const uint32_t* __restrict__ tab_enc,
                                            int n_dbps, uint32_t recip, int t, bool in_range, int& ra, int& rb)
{
    ra = 2; rb = 2;
    if (in_range) {
        const int sym = (int)__umulhi((uint32_t)t, recip);
        const int tt = t - sym * n_dbps;
        const uint32_t e = tab_enc[tt];
        const uint32_t ea = e & 0xffffu, eb = e >> 16;
        const uint32_t* sp = tile + (sym - sym0) * n_words;
        if (!(ea & WR_SRC_PUNCT)) ra = (int)((sp[(ea >> 5) & 15u] >> (ea & 31u)) & 1u);
        if (!(eb & WR_SRC_PUNCT)) rb = (int)((sp[(eb >> 5) & 15u] >> (eb & 31u)) & 1u);
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

// ---- pack kernel: `idx` (one byte per data carrier) -> bit planes (wifirx_out.hbits), for callers that hand
//      wifirx_decode_batch decisions without planes.  One wave per frame, lane <-> OFDM symbol: 48 bytes in, 2 n_bpsc
//      words out, both contiguous over the lanes. ----
template <int NB>
__device__ __forceinline__ void pack_symbol(const uint32_t (&by)[12], uint32_t* __restrict__ dst)
{
    uint32_t w[2 * NB];
#pragma unroll
    for (int k = 0; k < 2 * NB; k++) w[k] = 0;
#pragma unroll
    for (int c = 0; c < 48; c++) {
        const int bin = bin_of_carrier(c);
#pragma unroll
        for (int b = 0; b < NB; b++)
            w[2 * b + (bin >> 5)] |= ((by[c >> 2] >> (8 * (c & 3) + b)) & 1u) << (bin & 31);
    }
#pragma unroll
    for (int k = 0; k < 2 * NB; k++) dst[k] = w[k];
}

__global__ __launch_bounds__(256)
void decode_pack_kernel(uint32_t n_slots, uint32_t max_sym, const wifirx_frame* __restrict__ frames,
                        const uint8_t* __restrict__ idx_all, uint32_t psdu_stride, uint32_t* __restrict__ hbits_all,
                        uint32_t n_steps_cap)
{
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    const bool idx4 = ((reinterpret_cast<uintptr_t>(idx_all) | ((size_t)max_sym * 48)) & 3) == 0;
    for (uint32_t slot = blockIdx.x * 4 + wv; slot < n_slots; slot += gridDim.x * 4) {
        const wifirx_frame fr = frames[slot];
        const int enc = fr.encoding & 7;
        const int n_data = frame_steps(fr.flags, enc, fr.psdu_len, psdu_stride, max_sym, n_steps_cap);
        if (n_data == 0) continue;                                    // wave-uniform: one frame per wave
        const int n_sym = n_data / ndbps_tab[enc];
        const int nb = enc < 2 ? 1 : enc < 4 ? 2 : enc < 6 ? 4 : 6;
        const uint8_t* fidx = idx_all + (size_t)slot * max_sym * 48;
        uint32_t* fhb = hbits_all + (size_t)slot * max_sym * 12;
        for (int q = lane; q < n_sym; q += 64) {
            uint32_t by[12];
            if (idx4) {
#pragma unroll
                for (int k = 0; k < 12; k++) by[k] = reinterpret_cast<const uint32_t*>(fidx + (size_t)q * 48)[k];
            } else {
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const uint8_t* p = fidx + (size_t)q * 48 + 4 * k;
                    by[k] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
                }
            }
            uint32_t* dst = fhb + (size_t)q * 2 * nb;
            if (nb == 1)      pack_symbol<1>(by, dst);
            else if (nb == 2) pack_symbol<2>(by, dst);
            else if (nb == 4) pack_symbol<4>(by, dst);
            else              pack_symbol<6>(by, dst);
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
// (stride DBS dwords, two spare words behind the last one).  Four PSDU bytes per iteration: the 32 decoded bits from
// position 16 + 32 k on (a funnel shift of two decoded words), the 32 scrambler bits from the table (the state after
// them is their last seven, reversed), CRC by four table look-ups.  Bytes leave four at a time when the row is
// dword-aligned (wave-uniform `dword_ok`).
template <int DBS = 128>
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
    uint32_t nxt = db[DBS];
    const int n_words = psdu_len >> 2;
    for (int k = 0; k < n_words; k++) {
        const uint32_t nn = db[(size_t)(k + 2) * DBS];                     // spare words behind the last one keep this in range
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

// Six lane-private LDS reads whose rows come from the scalar unit: word[lane] of the rows at byte addresses a[0..5]
// (ds_read_addtid_b32: LDS[M0 + 4 * lane], tools/lds_addtid_probe.hip) -- no address register, no add per read, all six in
// flight together.  One wait state between the write of M0 and the LDS instruction that reads it.
__device__ __forceinline__ void lds_rows6(const uint32_t (&a)[6], uint32_t (&w)[6])
{
    uint32_t m0_kept;                    // M0 is the compiler's: handed back as it was found
    asm volatile("s_mov_b32 %6, m0\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tds_read_addtid_b32 %0\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tds_read_addtid_b32 %1\n\t"
                 "s_mov_b32 m0, %9\n\ts_nop 0\n\tds_read_addtid_b32 %2\n\t"
                 "s_mov_b32 m0, %10\n\ts_nop 0\n\tds_read_addtid_b32 %3\n\t"
                 "s_mov_b32 m0, %11\n\ts_nop 0\n\tds_read_addtid_b32 %4\n\t"
                 "s_mov_b32 m0, %12\n\ts_nop 0\n\tds_read_addtid_b32 %5\n\t"
                 "s_mov_b32 m0, %6\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&s"(m0_kept)
                 : "s"(a[0]), "s"(a[1]), "s"(a[2]), "s"(a[3]), "s"(a[4]), "s"(a[5])
                 : "memory");
}

// MIXED = false: the tasks whose frames all share one rate (the usual case: where a coded bit sits is then the same
// for every lane, and comes through the scalar unit).  MIXED = true: the other tasks (per-lane look-ups).  Both kernels are
// launched over the same tasks, one after the other on the stream, and each leaves the other's tasks alone.
template <bool MIXED>
__global__ __launch_bounds__(256, WR_DEC_WAVES_PER_SIMD)
void decode_kernel(uint32_t n_slots, uint32_t max_sym, wifirx_frame* __restrict__ frames,
                   const uint32_t* __restrict__ hbits_all, uint8_t* __restrict__ psdu_all, uint32_t psdu_stride,
                   uint8_t* __restrict__ scratch, size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves_total,
                   uint32_t frames_per_wave, const uint32_t* __restrict__ perm, uint32_t n_virtual)
{
    // perm != nullptr: the wave's frames are perm[base + ...] (0xffffffff: no frame) -- the decodable frames of a batch
    // with several rates, grouped by rate into runs that start on task boundaries (decode_perm_kernel), so that every
    // task is of one rate; perm == nullptr: frame k is slot k, n_virtual = n_slots.
    // the current OFDM symbols of the wave's frames, lane-private columns.  One rate: row 2 w + g (w = word of the staged
    // block, g = 0, 1) = 16-bit plane g of the word of frame A | that of frame B << 16 (plane p of a symbol = row 2 n_w s + p
    // for the s-th symbol of the block); mixed rates: row 12 h + k = word k of the current symbol of the frame of half h.
    __shared__ uint32_t sym_all[4][(MIXED ? 24 : 32) * 64];
    __shared__ uint32_t src_tab[MIXED ? 8 * WR_DEC_TAB_STRIDE : 1];
    __shared__ FinishTables ft;
    if (MIXED) copy_src_table(src_tab);
    build_finish_tables(ft);
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4 + wv;
    if (wave >= n_waves_total) return;
    uint32_t* symw = sym_all[wv] + lane;
    const uint32_t sym_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)sym_all[wv]);     // LDS byte address of the wave's rows
    const size_t n_data_cap = n_steps_cap;               // trellis steps the scratch slice of a wave holds
    uint32_t* surv = reinterpret_cast<uint32_t*>(scratch + (size_t)wave * scratch_stride);   // [step][lane][4 pieces]
    uint32_t* dbits = surv + n_data_cap * 256;                                               // [word][A/B][lane]
    const uint32_t k1 = 0x00010001u;
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    const uint32_t hb_stride = max_sym * 12;             // words per frame

    // A wave's tasks (frames_per_wave <= 128 frames each, grid-stride); lane l owns frames base + l (l < fA) and
    // base + fA + l (l < fB).
    const uint32_t fA = frames_per_wave < 64 ? frames_per_wave : 64, fB = frames_per_wave - fA;
    const uint32_t n_tasks = (n_virtual + frames_per_wave - 1) / frames_per_wave;
    for (uint32_t task = wave; task < n_tasks; task += n_waves_total) {
        const uint32_t base = task * frames_per_wave;
        // ---- my two frames ----
        int n_data[2], enc[2];
        uint32_t slot_of[2];
        int n_max = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t v = base + (h ? fA : 0u) + lane;
            uint32_t slot = 0xffffffffu;
            if ((uint32_t)lane < (h ? fB : fA) && v < n_virtual) slot = perm ? perm[v] : v;
            n_data[h] = 0;
            enc[h] = 0;
            if (slot < n_slots) {
                enc[h] = frames[slot].encoding & 7;
                n_data[h] = frame_steps(frames[slot].flags, enc[h], frames[slot].psdu_len, psdu_stride, max_sym, n_steps_cap);
            }
            slot_of[h] = slot < n_slots ? slot : 0u;
            n_max = n_data[h] > n_max ? n_data[h] : n_max;
        }
        const uint32_t* rowA = hbits_all + (size_t)slot_of[0] * hb_stride;      // my frames' plane rows
        const uint32_t* rowB = hbits_all + (size_t)slot_of[1] * hb_stride;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            int o = __shfl_xor(n_max, k, 64);
            n_max = o > n_max ? o : n_max;
        }
        n_max = __builtin_amdgcn_readfirstlane(n_max);
        if (n_max == 0) continue;
        // one rate for all frames of the task?
        const uint64_t actA = __ballot(n_data[0] > 0), actB = __ballot(n_data[1] > 0);
        const int enc_u = actA ? __builtin_amdgcn_readlane(enc[0], (int)__builtin_ctzll(actA))
                               : __builtin_amdgcn_readlane(enc[1], (int)__builtin_ctzll(actB));
        const bool uni_rate = __all((n_data[0] == 0 || enc[0] == enc_u) && (n_data[1] == 0 || enc[1] == enc_u));
        if (uni_rate == MIXED) continue;                 // the other kernel's task
        const int nd_u = ndbps_tab[enc_u];
        const int nw_u = enc_u < 2 ? 2 : enc_u < 4 ? 4 : enc_u < 6 ? 8 : 12;        // words per symbol
        const int sym_blk = enc_u < 2 ? 8 : enc_u < 4 ? 4 : enc_u < 6 ? 2 : 1;      // symbols staged together (one rate)
        const int n_ld = enc_u < 6 ? 16 : 12;                                       // = sym_blk * nw_u words

        // ---- phase 2: add-compare-select ----
        uint32_t pm[64];
#pragma unroll
        for (int s = 0; s < 64; s++) pm[s] = (s == 0) ? 0u : WR_DEC_START_PENALTY;
        int best[2] = { 0, 0 };                         // final states, taken when the frames end
        int tt_u = 0, sym_u = 0, since_norm = 0;        // wave-uniform: step within the symbol, symbol (one rate), steps since the minimum left
        uint32_t pos[2] = { 0u, 0u };                   // mixed rates, per lane and half: symbol << 8 | step within the symbol
        for (int tg = 0; tg < n_max; tg += 6) {
            if (since_norm == WR_DEC_NORM_STEPS) {
                // subtract the common minimum of each frame (register phase 0 here; decisions see differences only)
                since_norm = 0;
                uint32_t mn = pm[0];
#pragma unroll
                for (int s = 1; s < 64; s++) mn = pk_min(mn, pm[s]);
#pragma unroll
                for (int s = 0; s < 64; s++) pm[s] = pk_sub(pm[s], mn);
            }
            since_norm += 6;
            // ---- a new OFDM symbol: its bit-plane words, global -> lane-private LDS ----
            if (!MIXED) {
                // 16 words (64 bytes: a whole memory segment) per frame at a time = 8 / 4 / 2 symbols at 1 / 2 / 4 bits per
                // carrier; 12 words = one symbol at 6
                if (tt_u == nd_u) { tt_u = 0; sym_u++; }
                if (tt_u == 0 && (sym_u & (sym_blk - 1)) == 0) {
                    const bool okA = tg < n_data[0], okB = tg < n_data[1];
                    const uint32_t* pa = rowA + (uint32_t)(sym_u * nw_u);
                    const uint32_t* pb = rowB + (uint32_t)(sym_u * nw_u);
                    const uint32_t room = hb_stride - (uint32_t)(sym_u * nw_u);      // words left in a frame's row
#pragma unroll
                    for (int k = 0; k < 16; k += 4) {
                        if (k < n_ld) {
                            uint4 a = make_uint4(0u, 0u, 0u, 0u), b = make_uint4(0u, 0u, 0u, 0u);
                            if (okA && (uint32_t)k + 4 <= room) a = *reinterpret_cast<const uint4*>(pa + k);
                            if (okB && (uint32_t)k + 4 <= room) b = *reinterpret_cast<const uint4*>(pb + k);
                            uint32_t* d = symw + 2 * k * 64;
                            d[0 * 64] = (a.x & 0xffffu) | (b.x << 16);  d[1 * 64] = (a.x >> 16) | (b.x & 0xffff0000u);
                            d[2 * 64] = (a.y & 0xffffu) | (b.y << 16);  d[3 * 64] = (a.y >> 16) | (b.y & 0xffff0000u);
                            d[4 * 64] = (a.z & 0xffffu) | (b.z << 16);  d[5 * 64] = (a.z >> 16) | (b.z & 0xffff0000u);
                            d[6 * 64] = (a.w & 0xffffu) | (b.w << 16);  d[7 * 64] = (a.w >> 16) | (b.w & 0xffff0000u);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int nd = ndbps_tab[enc[h]];
                    if ((int)(pos[h] & 0xffu) == nd) pos[h] = (pos[h] & ~0xffu) + 0x100u;
                    if ((pos[h] & 0xffu) == 0u && tg < n_data[h]) {
                        const int nw = enc[h] < 2 ? 2 : enc[h] < 4 ? 4 : enc[h] < 6 ? 8 : 12;
                        const uint32_t* ps = (h ? rowB : rowA) + (pos[h] >> 8) * (uint32_t)nw;
                        for (int k = 0; k < nw; k++) symw[(12 * h + k) * 64] = ps[k];
                    }
                }
            }
            const bool mine0 = tg < n_data[0], mine1 = tg < n_data[1];
            // one trellis step: the coded bits A, B of both frames (ta, tb: frame A in bit 0, frame B in bit 16) and
            // whether they were transmitted (va, vb)
#define WR_ACS(P, ta, tb, va, vb)                                                                         \
                    {                                                                                     \
                        const uint32_t nv = (va) + (vb);      /* a set bit implies its valid bit */       \
                        uint32_t M[2][2];                                                                 \
                        M[0][0] = (ta) + (tb);                                                            \
                        M[0][1] = (ta) + (vb) - (tb);                                                     \
                        M[1][1] = nv - M[0][0];                                                           \
                        M[1][0] = nv - M[0][1];                                                           \
                        uint32_t acc[4];                                                                  \
                        acs_step<P>(pm, M, acc);                                                          \
                        *reinterpret_cast<uint4*>(surv + ((size_t)(tg + P) * 64 + lane) * 4) =            \
                            make_uint4(acc[0], acc[1], acc[2], acc[3]);                                   \
                    }
            if (!MIXED) {
                // Three steps at a time: their six rows (A and B of each step) are read together, rows and bit positions
                // from the scalar unit; a punctured position reads plane 0 and is masked away (no branches).
                const uint32_t* te = WR_SRC_TABLE.e + (enc_u * WR_DEC_TAB_STRIDE + tt_u);      // wave-uniform: scalar loads
                const uint32_t blk_lds = sym_lds + (uint32_t)((sym_u & (sym_blk - 1)) * nw_u) * 512u;      // this symbol's rows
                uint32_t e[6];
#pragma unroll
                for (int P = 0; P < 6; P++) e[P] = te[P];
#define WR_FETCH3(Q)                                                                                      \
                uint32_t rr##Q[6], ww##Q[6];                                                              \
                _Pragma("unroll") for (int k = 0; k < 3; k++) {                                           \
                    rr##Q[k] = blk_lds + ((e[Q + k] << 4) & 0x1f00u);      /* plane p = bits 4..8 of the half: row p, 256 bytes each */ \
                    rr##Q[3 + k] = blk_lds + ((e[Q + k] >> 12) & 0x1f00u);                                \
                }                                                                                         \
                lds_rows6(rr##Q, ww##Q);
#define WR_ACS_U(P, Q)                                                                                    \
                {                                                                                         \
                    const uint32_t ea = e[P] & 0xffffu, eb = e[P] >> 16;                                  \
                    const uint32_t va = k1 & (((ea >> 9) & 1u) - 1u), vb = k1 & (((eb >> 9) & 1u) - 1u);  \
                    const uint32_t ta = (ww##Q[P - Q] >> (ea & 15u)) & va, tb = (ww##Q[3 + P - Q] >> (eb & 15u)) & vb; \
                    WR_ACS(P, ta, tb, va, vb)                                                             \
                }
                if (mine0 || mine1) {
                    { WR_FETCH3(0)  WR_ACS_U(0, 0) WR_ACS_U(1, 0) WR_ACS_U(2, 0) }
                    { WR_FETCH3(3)  WR_ACS_U(3, 3) WR_ACS_U(4, 3) WR_ACS_U(5, 3) }
                }
#undef WR_ACS_U
#undef WR_FETCH3
            } else {
                // six steps of coded bits: pw[0] / pw[2] = bits A / B (step P of frame A in bit P, of frame B in bit 16 + P),
                // pw[1] / pw[3] = whether they were transmitted
                uint32_t pw[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t* te = src_tab + enc[h] * WR_DEC_TAB_STRIDE + (int)(pos[h] & 0xffu);
                    const uint32_t* sw = symw + 12 * h * 64;
#pragma unroll
                    for (int P = 0; P < 6; P++) {
                        const uint32_t e = te[P];
                        const uint32_t ea = e & 0xffffu, eb = e >> 16;
                        const uint32_t va = ((ea >> 9) & 1u) ^ 1u, vb = ((eb >> 9) & 1u) ^ 1u;
                        pw[0] |= ((sw[((ea >> 5) & 15u) * 64] >> (ea & 31u)) & va) << (16 * h + P);
                        pw[1] |= va << (16 * h + P);
                        pw[2] |= ((sw[((eb >> 5) & 15u) * 64] >> (eb & 31u)) & vb) << (16 * h + P);
                        pw[3] |= vb << (16 * h + P);
                    }
                }
                if (mine0 || mine1) {
#define WR_ACS_M(P) WR_ACS(P, (pw[0] >> P) & k1, (pw[2] >> P) & k1, (pw[1] >> P) & k1, (pw[3] >> P) & k1)
                    WR_ACS_M(0) WR_ACS_M(1) WR_ACS_M(2) WR_ACS_M(3) WR_ACS_M(4) WR_ACS_M(5)
#undef WR_ACS_M
                }
            }
#undef WR_ACS
            {
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
            tt_u += 6;
            if (MIXED) { pos[0] += 6u; pos[1] += 6u; }
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
                const uint32_t slot = slot_of[h];
                finish_frame(dbits + 64 * h + lane, frames[slot].psdu_len, psdu_all + (size_t)slot * psdu_stride,
                             ((reinterpret_cast<uintptr_t>(psdu_all) | psdu_stride) & 3) == 0, frames + slot, frames[slot].flags, ft);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// decode_q_kernel: the throughput decoder with FOUR frames per lane (256 per wave), path metrics a byte each in one
// register ("SWAR").  For batches that fill the GPU with such waves (about a million frames per call); every task is of
// one rate (a batch of several rates comes grouped by rate, decode_perm_kernel).
//
// What makes bytes enough.  After six steps every state has a survivor that started in state 0, and from then on the
// 64 path metrics of a frame lie within 12 of each other: any state is reached from the best state of six steps ago in
// six steps of at most 2 each.  So (i) the first six steps are not compared at all: a state's metric after them is
// the sum of the six branch metrics along its only genuine path (a tree of 126 additions), and its survivor bits there
// are 0 (the other candidate comes from a state that cannot have been reached yet) -- exactly what the oracle's
// "unreachable = 2^28" start yields; (ii) the common minimum leaves the metrics every 48 steps, so a byte stays below
// 12 + 2 x 48 + 2 < 128; (iii) the two candidates of a state differ by at most 14, so the byte 127 - c1 + c0 (ONE v_xad_u32:
// (c1 ^ 0x7f7f7f7f) + c0) lies in 113..141: no carry crosses a byte, bit 7 of it says "c1 < c0" (the strict comparison of
// the oracle's tie rule) and bits 4, 5, 6 all say the opposite.  One v_bfi_b32 therefore drops a decision into bit 7, 6, 5 or
// 4 of the accumulator's bytes without a shift (the last three inverted), four decisions, then the accumulator moves
// down by four; the smaller candidate is picked byte-wise by one v_perm_b32 whose selector is (z >> 5) & 0x04040404 |
// 0x03020100.  A butterfly (two new states of four frames) is 4 + 2 x 5.1 = 14.3 instructions: 3.6 per frame against 5.25 of
// the packed-16-bit form of decode_kernel.
//
// Bytes of a register: byte 0 = frame l (h = 0), byte 1 = frame 128 + l (h = 2), byte 2 = frame 64 + l (h = 1), byte 3 =
// frame 192 + l (h = 3) -- the order in which the 16-bit plane halves of two staged words fall into bytes.
// Survivor words: acc[g] (g = 0..7) holds the decisions of the states 8g .. 8g+7, state 8g + i in bit i ^ 3 of every byte
// ({3,2,1,0,7,6,5,4}[i]), stored inverted except for i = 0 and i = 4.
#define WR_DQ_FRAMES    256
#define WR_DQ_NORM      48
// (WR_DQ_SPEC_WAVES -- waves per SIMD of the speculative-trace-back instances -- is in wr_kernels.h: the host sizes the launch by it)
#define WR_DQ_BYTE(h)   ((h) == 0 ? 0 : (h) == 1 ? 2 : (h) == 2 ? 1 : 3)

__device__ __forceinline__ uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel)
{
    uint32_t r;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "v"(sel));
    return r;
}
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t m, uint32_t o)      // (a & m) | o; m through the scalar unit, o in a register
{
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m), "v"(o));
    return r;
}
// byte-wise min(a, b) for bytes below 128 that differ by less than 16
__device__ __forceinline__ uint32_t min_u8x4(uint32_t a, uint32_t b, uint32_t sel0)
{
    const uint32_t y = (b | 0x80808080u) - a;                  // byte: 128 + b - a; bits 4..6 set iff b < a
    return perm_b32(b, a, and_or(y >> 2, 0x04040404u, sel0));
}

__device__ __forceinline__ uint32_t perm_s(uint32_t hi, uint32_t lo, uint32_t sel)      // selector through the scalar unit
{
    uint32_t r;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "s"(sel));
    return r;
}
// (a ^ m) + b: with m = 0x7f7f7f7f and bytes below 128, byte-wise 127 - a + b
__device__ __forceinline__ uint32_t xad(uint32_t a, uint32_t m, uint32_t b)
{
    uint32_t r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m), "v"(b));
    return r;
}

// The state with the smallest path metric of each of the four frames of a lane, lowest state on ties (register phase 0: pm[s] is
// state s; byte WR_DQ_BYTE(h) of a register = frame h).  Keys (metric << 6 | state) of two frames side by side in the 16-bit
// halves of a register -- bytes 0 / 2 = frames h = 0 / 1, bytes 1 / 3 = h = 2 / 3; a metric is below 128, a key below 2^13 --
// and one packed 16-bit minimum per pair and state: 7 instructions per state for four frames (the byte-by-byte search -- extract,
// compare, two selects per frame -- took 12 and as many wait states).
__device__ __forceinline__ void best_states4(const uint32_t (&pm)[64], int (&bs)[4])
{
    uint32_t k01 = 0xffffffffu, k23 = 0xffffffffu;
#pragma unroll
    for (int s2 = 0; s2 < 64; s2++) {
        const uint32_t st2 = (uint32_t)s2 * 0x00010001u;
        const uint32_t a = ((pm[s2] & 0x00ff00ffu) << 6) | st2;
        const uint32_t b = (((pm[s2] >> 8) & 0x00ff00ffu) << 6) | st2;
        k01 = pk_min(k01, a);
        k23 = pk_min(k23, b);
    }
    bs[0] = (int)(k01 & 63u); bs[1] = (int)((k01 >> 16) & 63u);
    bs[2] = (int)(k23 & 63u); bs[3] = (int)((k23 >> 16) & 63u);
}

// One trellis step at register phase P.  The survivor words leave in two halves (states 0..31 after the butterflies
// 0..15, states 32..63 after 16..31), each transposed to per-frame words right away: four accumulators live at a time.
template <int P>
__device__ __forceinline__ void acs_step_q(uint32_t (&pm)[64], const uint32_t (&M)[2][2], uint32_t sel0, uint32_t* __restrict__ dst)
{
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t acc[4];
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int j = 16 * half + jj;
            const int a = parity_of((j << 1) & 0155), b = parity_of((j << 1) & 0117);
            const uint32_t m = M[a][b], mb = M[a ^ 1][b ^ 1];
            const int r0 = rotr6(j, P), r1 = rotr6(j + 32, P);
            const uint32_t p0 = pm[r0], p1 = pm[r1];
            const uint32_t c00 = p0 + m, c01 = p1 + mb, c10 = p0 + mb, c11 = p1 + m;
            // z = 127 + c0 - c1 per byte (113 .. 141: no carry crosses a byte): bit 7 says "candidate 1 is smaller" (strictly: a
            // tie keeps candidate 0), bits 4, 5, 6 say the opposite
            const uint32_t z0 = xad(c01, 0x7f7f7f7fu, c00), z1 = xad(c11, 0x7f7f7f7fu, c10);
            uint32_t& w = acc[jj >> 2];
            const int i = 2 * (j & 3);                              // decisions i, i + 1 of this word (states 2j, 2j + 1)
            if (i == 4) w >>= 4;                                    // the high nibbles are full: make room
            // i = 0, 4 -> bit 7 and bit 6 (inverted);  i = 2, 6 -> bits 5 and 4 (both inverted)
            if (i == 0)           w = z0 & 0x80808080u;
            else if ((i & 3) == 0) w = bfi(0x80808080u, z0, w);
            else                   w = bfi(0x20202020u, z0, w);
            w = bfi((i & 3) == 0 ? 0x40404040u : 0x10101010u, z1, w);
            pm[r0] = perm_b32(c01, c00, and_or(z0 >> 5, 0x04040404u, sel0));
            pm[r1] = perm_b32(c11, c10, and_or(z1 >> 5, 0x04040404u, sel0));
        }
        // 4 x 4 byte transpose: T[k] = byte k of acc[0..3] = the decisions of the frame with byte index k, states 32 half ..
        // 32 half + 31 (byte g' = states 8g' .. 8g'+7)
        const uint32_t ta = perm_s(acc[1], acc[0], 0x05010400u), tb = perm_s(acc[1], acc[0], 0x07030602u);
        const uint32_t tc = perm_s(acc[3], acc[2], 0x05010400u), td = perm_s(acc[3], acc[2], 0x07030602u);
        // row layout: [half][byte index k] -> one 16-byte store per half
        reinterpret_cast<uint4*>(dst)[half] = make_uint4(perm_s(tc, ta, 0x05040100u), perm_s(tc, ta, 0x07060302u),
                                                         perm_s(td, tb, 0x05040100u), perm_s(td, tb, 0x07060302u));
    }
}

// The trace-back of a task overlapped with the add-compare-select of the wave's NEXT task (template OVL).  Alone, the two
// phases of all waves of a round coincide: nine milliseconds of pure survivor writes, then every wave reads its 4.9 MB
// of survivor bits back at once (19 GB, HBM-bound, ~3.8 ms) with the vector ALUs idle.  Overlapped, a wave keeps the finished
// task as "pending" (its states, its half of the wave's double scratch buffer) and, after every six add-compare-select
// steps of the next task, walks the pending one back by six steps -- the survivor reads are spread over the next task's
// arithmetic; only the last task of a wave is walked back on its own.  Two waves per SIMD then (each wave needs two tasks:
// half as many waves, twice the scratch per wave, 256 registers).  Only tasks whose frames all run the same number of steps
// are deferred; others are walked back on the spot as before.
struct TbPending {
    bool     active;                  // wave-uniform
    int      st[4];
    uint32_t slot[4];                 // 0xffffffff: no frame
    uint32_t aw[4][3];
    int      blk, grp;                // wave-uniform: next block of 96 steps, next group of six in it
    const uint32_t* surv;
    uint32_t* dbits;
};

// Six trace-back steps of a pending task (decode_q_kernel, OVL).  When a block of 96 steps is through, its three decoded
// words per frame leave; when the last block is through, the frames are finished (descrambling, bytes, CRC).
#define WR_DQ_PICKP(ST, LO, HI)                                                                           \
    (__builtin_amdgcn_ubfe((((ST & 32) ? (HI) : (LO)) ^ 0x77777777u), (uint32_t)((ST & 31) ^ 3), 1u))
__device__ __forceinline__ void tb_pending_group(TbPending& pend, int lane, wifirx_frame* __restrict__ frames,
                                                 uint8_t* __restrict__ psdu_all, uint32_t psdu_stride, const FinishTables& ft)
{
    const uint32_t* srow = pend.surv + ((size_t)(pend.blk * 96 + 6 * pend.grp) * 64 + lane) * 8;
    // (one array per word position: a select between two elements of ONE array becomes an indexed access in scratch memory)
    uint32_t l0[6], u0[6], l1[6], u1[6], l2[6], u2[6], l3[6], u3[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const uint4* sp = reinterpret_cast<const uint4*>(srow + (size_t)k * 512);
        const uint4 a = sp[0], b = sp[1];
        l0[k] = a.x; u0[k] = b.x; l1[k] = a.z; u1[k] = b.z; l2[k] = a.y; u2[k] = b.y; l3[k] = a.w; u3[k] = b.w;
    }
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const uint32_t v = __builtin_bitreverse32((uint32_t)pend.st[h]) >> 26;      // u_(t-5) .. u_t, oldest in bit 0
        pend.aw[h][2] = __builtin_amdgcn_alignbit(pend.aw[h][2], pend.aw[h][1], 26);  // the triple moves UP by six ...
        pend.aw[h][1] = __builtin_amdgcn_alignbit(pend.aw[h][1], pend.aw[h][0], 26);
        pend.aw[h][0] = (pend.aw[h][0] << 6) | v;                                      // ... the older steps' bits enter below
    }
#pragma unroll
    for (int q = 5; q >= 0; q--) {
        const uint32_t h0 = WR_DQ_PICKP(pend.st[0], l0[q], u0[q]), h1 = WR_DQ_PICKP(pend.st[1], l1[q], u1[q]);
        const uint32_t h2 = WR_DQ_PICKP(pend.st[2], l2[q], u2[q]), h3 = WR_DQ_PICKP(pend.st[3], l3[q], u3[q]);
        pend.st[0] = (pend.st[0] >> 1) | (int)(h0 << 5);
        pend.st[1] = (pend.st[1] >> 1) | (int)(h1 << 5);
        pend.st[2] = (pend.st[2] >> 1) | (int)(h2 << 5);
        pend.st[3] = (pend.st[3] >> 1) | (int)(h3 << 5);
    }
    if (--pend.grp < 0) {
#pragma unroll
        for (int w = 0; w < 3; w++)
#pragma unroll
            for (int h = 0; h < 4; h++) { pend.dbits[(size_t)(pend.blk * 3 + w) * 256 + 64 * h + lane] = pend.aw[h][w]; pend.aw[h][w] = 0u; }
        pend.grp = 15;
        if (--pend.blk < 0) {
            __threadfence_block();
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const uint32_t slot = pend.slot[h];
                if (slot != 0xffffffffu)
                    finish_frame<256>(pend.dbits + 64 * h + lane, frames[slot].psdu_len, psdu_all + (size_t)slot * psdu_stride,
                                      ((reinterpret_cast<uintptr_t>(psdu_all) | psdu_stride) & 3) == 0, frames + slot, frames[slot].flags, ft);
            }
            pend.active = false;
        }
    }
}

// The trace-back of a task overlapped with ITS OWN add-compare-select (template MODE = 2, round 5).  The overlap above hides the
// trace-back of a task under the arithmetic of the wave's NEXT task -- with two tasks per wave (a million frames = 3 907 tasks over
// 2 048 waves) every second trace-back still ran on its own, all waves at once: 10 GB of survivor reads, 1.9 of 12.1 ms.  Here every
// block of 96 steps is walked back SPECULATIVELY while the next block's add-compare-select runs: when the trellis has reached
// boundary B_(j+1) = 96 (j + 1), a walk starts there in the state with the smallest metric (any state would do for correctness),
// goes back through block j six steps per group of six trellis steps, and leaves the block's 96 decoded bits.  Survivor paths
// merge: the walk from B_(j+2) arrives at B_(j+1) in the state the walk before it STARTED from unless that state was not on the
// final path -- the link of block j + 1 is then marked broken.  After the last trellis step the true trace-back starts in the best
// final state and stops at the first boundary B_j where it stands in the recorded start state S_j of walk j - 1 and no link below
// is broken: from there down its path IS the chain of the speculative walks, whose bits are already in place.  Exact -- a walk
// that was off the final path is walked over again by the final one --, and what runs exposed is one block (plus the steps
// behind the last boundary) instead of the whole trellis.  Tasks whose frames differ in length are walked back on the spot.
struct SpecWalk {
    bool     active;                  // wave-uniform
    int      st[4];
    uint32_t aw[4][3];
    int      blk, grp;                // wave-uniform: the block being walked, the next group of six in it
};

// Six steps of the speculative walk.  When the block is through: its three decoded words per frame leave, and the state the walk
// arrived in at the block's lower boundary is compared with the start state of the walk before (link_ref, a byte per frame): a
// mismatch marks the link of this block broken (fb[h] = lowest broken block).
__device__ __forceinline__ void spec_walk_group(SpecWalk& sw, int lane, const uint32_t* __restrict__ surv, uint32_t* __restrict__ dbits,
                                                uint32_t link_ref, int (&fb)[4])
{
    const uint32_t* srow = surv + ((size_t)(sw.blk * 96 + 6 * sw.grp) * 64 + lane) * 8;
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const uint32_t v = __builtin_bitreverse32((uint32_t)sw.st[h]) >> 26;      // u_(t-5) .. u_t, oldest in bit 0
        sw.aw[h][2] = __builtin_amdgcn_alignbit(sw.aw[h][2], sw.aw[h][1], 26);
        sw.aw[h][1] = __builtin_amdgcn_alignbit(sw.aw[h][1], sw.aw[h][0], 26);
        sw.aw[h][0] = (sw.aw[h][0] << 6) | v;
    }
    // the six rows in two halves of three (24 registers in flight, not 48: the kernel runs at three waves per SIMD -- 168 registers)
#pragma unroll
    for (int half = 1; half >= 0; half--) {
        uint32_t l0[3], u0[3], l1[3], u1[3], l2[3], u2[3], l3[3], u3[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint4* sp = reinterpret_cast<const uint4*>(srow + (size_t)(3 * half + k) * 512);
            const uint4 a = sp[0], b = sp[1];
            l0[k] = a.x; u0[k] = b.x; l1[k] = a.z; u1[k] = b.z; l2[k] = a.y; u2[k] = b.y; l3[k] = a.w; u3[k] = b.w;
        }
#pragma unroll
        for (int q = 2; q >= 0; q--) {
            const uint32_t h0 = WR_DQ_PICKP(sw.st[0], l0[q], u0[q]), h1 = WR_DQ_PICKP(sw.st[1], l1[q], u1[q]);
            const uint32_t h2 = WR_DQ_PICKP(sw.st[2], l2[q], u2[q]), h3 = WR_DQ_PICKP(sw.st[3], l3[q], u3[q]);
            sw.st[0] = (sw.st[0] >> 1) | (int)(h0 << 5);
            sw.st[1] = (sw.st[1] >> 1) | (int)(h1 << 5);
            sw.st[2] = (sw.st[2] >> 1) | (int)(h2 << 5);
            sw.st[3] = (sw.st[3] >> 1) | (int)(h3 << 5);
        }
    }
    if (--sw.grp < 0) {
#pragma unroll
        for (int w = 0; w < 3; w++)
#pragma unroll
            for (int h = 0; h < 4; h++) dbits[(size_t)(sw.blk * 3 + w) * 256 + 64 * h + lane] = sw.aw[h][w];
#pragma unroll
        for (int h = 0; h < 4; h++)
            if (sw.st[h] != (int)((link_ref >> (8 * h)) & 63u) && sw.blk < fb[h]) fb[h] = sw.blk;
        sw.active = false;
    }
}
#undef WR_DQ_PICKP

// MODE: 0 = a task's trace-back behind its add-compare-select; 1 = overlapped with the wave's next task (above); 2 = speculative
// walks overlapped with the task's own add-compare-select (above).
template <int ROWS, int MODE>        // LDS rows per wave: 32 (rates up to 16-QAM: 8 staged words x 4), 48 (64-QAM: one symbol of 12 words)
__global__ __launch_bounds__(256, MODE == 2 ? (ROWS == 32 ? WR_DQ_SPEC_WAVES : 2) : MODE ? 2 : (ROWS == 32 ? 4 : 3))      // (48 rows: 52.5 kB of LDS per workgroup -- two fit a CU anyway)
void decode_q_kernel(uint32_t n_slots, uint32_t max_sym, wifirx_frame* __restrict__ frames,
                     const uint32_t* __restrict__ hbits_all, uint8_t* __restrict__ psdu_all, uint32_t psdu_stride,
                     uint8_t* __restrict__ scratch, size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves_total,
                     uint32_t frames_per_wave, const uint32_t* __restrict__ perm, uint32_t n_virtual)
{
    // the current OFDM symbols of the wave's frames: row 4 w + 2 c + g (w = word of the staged 8-word block, c = 0: frames
    // h = 0, 1; c = 1: frames h = 2, 3; g = 16-bit plane of the word) = plane of the first frame | that of the second << 16
    constexpr bool OVL = MODE == 1, SPEC = MODE == 2;
    __shared__ uint32_t sym_all[4][ROWS * 64];
    __shared__ FinishTables ft;
    build_finish_tables(ft);
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4 + wv;
    if (wave >= n_waves_total) return;
    uint32_t* symw = sym_all[wv] + lane;
    const uint32_t sym_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)sym_all[wv]);
    const size_t n_data_cap = n_steps_cap;
    // scratch of the wave: [step][lane][8 words] of survivor bits + [word][h][lane] of decoded bits; OVL: two such halves
    // (SPEC: + [boundary][lane] the start states of the speculative walks, a byte per frame)
    const size_t half_words = n_data_cap * 512 + (n_data_cap / 32 + 2) * 256;
    uint32_t* const scr0 = reinterpret_cast<uint32_t*>(scratch + (size_t)wave * scratch_stride);
    int buf = 0;
    TbPending pend;
    pend.active = false;
    const uint32_t k1 = 0x01010101u;
    uint32_t sel0 = 0x03020100u;                      // v_perm_b32 selector "every byte from the second source", kept in a register
    asm volatile("" : "+v"(sel0));
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    const uint32_t hb_stride = max_sym * 12;

    const uint32_t n_tasks = (n_virtual + frames_per_wave - 1) / frames_per_wave;
    for (uint32_t task = wave; task < n_tasks; task += n_waves_total) {
        const uint32_t base = task * frames_per_wave;
        int n_data[4], enc_any = 0;
        uint32_t slot_of[4];
        int n_max = 0;
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const uint32_t fi = 64u * h + (uint32_t)lane;          // frame of the task
            uint32_t slot = 0xffffffffu;
            if (fi < frames_per_wave && base + fi < n_virtual) slot = perm ? perm[base + fi] : base + fi;
            n_data[h] = 0;
            if (slot < n_slots) {
                const int e = frames[slot].encoding & 7;
                n_data[h] = frame_steps(frames[slot].flags, e, frames[slot].psdu_len, psdu_stride, max_sym, n_steps_cap);
                if (n_data[h]) enc_any = e;
            }
            slot_of[h] = slot < n_slots ? slot : 0u;
            n_max = n_data[h] > n_max ? n_data[h] : n_max;
        }
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            int o = __shfl_xor(n_max, k, 64);
            n_max = o > n_max ? o : n_max;
        }
        n_max = __builtin_amdgcn_readfirstlane(n_max);
        if (n_max == 0) continue;
        // the task's rate (every frame of a task shares it: the caller groups by rate)
        const uint64_t any = __ballot(n_data[0] > 0 || n_data[1] > 0 || n_data[2] > 0 || n_data[3] > 0);
        const int enc_u = __builtin_amdgcn_readlane(enc_any, (int)__builtin_ctzll(any));
        const int nd_u = ndbps_tab[enc_u];
        const int nw_u = enc_u < 2 ? 2 : enc_u < 4 ? 4 : enc_u < 6 ? 8 : 12;        // words per symbol
        const int blk_w = enc_u < 6 ? 8 : 12;                                       // words staged together: 4 / 2 / 1 / 1 symbols
        if (ROWS < 48 && enc_u >= 6) continue;                                      // (the host launches the 48-row instance when 64-QAM frames exist)
        const int sym_blk = blk_w / nw_u;

        uint32_t* surv = scr0 + (OVL ? (size_t)buf * half_words : 0);
        uint32_t* dbits = surv + n_data_cap * 512;
        uint32_t* rec = dbits + (n_data_cap / 32 + 2) * 256;         // SPEC only (the host sized the slice for it)
        uint32_t pm[64];
        int best[4] = { 0, 0, 0, 0 };
        int tt_u = 0, sym_u = 0, since_norm = 0;
        // fast trace-back (blocks of 96 steps) when all frames of the wave run the full n_max steps
        const bool uni = __all(n_data[0] == n_max && n_data[1] == n_max && n_data[2] == n_max && n_data[3] == n_max);
        const int n_fast = uni ? (n_max / 96) * 96 : 0;
        SpecWalk sw;
        sw.active = false;
        int fb[4] = { 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff };     // SPEC: lowest block whose link to the walk below is broken
        uint32_t cur_start = 0u, link_ref = 0u;                              // SPEC: start states (a byte per frame) of the running walk / of the one before
        // one group of six trellis steps; `first`: the group without comparisons (steps 0..5)
        auto group = [&](auto first_tag, const int tg) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            if (!FIRST && since_norm == WR_DQ_NORM) {
                since_norm = 0;
                uint32_t mn = pm[0];
#pragma unroll
                for (int s = 1; s < 64; s++) mn = min_u8x4(mn, pm[s], sel0);
#pragma unroll
                for (int s = 0; s < 64; s++) pm[s] -= mn;
            }
            since_norm += 6;
            // ---- the next block of bit-plane words, global -> lane-private LDS ----
            if (tt_u == nd_u) { tt_u = 0; sym_u++; }
            if (tt_u == 0 && (sym_u % sym_blk) == 0) {
                const uint32_t w0 = (uint32_t)(sym_u * nw_u);
                const uint32_t room = hb_stride - w0;
#pragma unroll
                for (int k = 0; k < 12; k += 4) {
                    if (k < blk_w) {
                        uint4 v[4];
#pragma unroll
                        for (int h = 0; h < 4; h++) {
                            v[h] = make_uint4(0u, 0u, 0u, 0u);
                            if (tg < n_data[h] && (uint32_t)k + 4 <= room)          // (the row pointer is formed here, every few dozen steps: registers)
                                v[h] = *reinterpret_cast<const uint4*>(hbits_all + (size_t)slot_of[h] * hb_stride + w0 + k);
                        }
                        uint32_t* d = symw + 4 * k * 64;
#define WR_DQ_ST(W, X) \
                        d[(4 * W + 0) * 64] = (v[0].X & 0xffffu) | (v[1].X << 16);  d[(4 * W + 1) * 64] = (v[0].X >> 16) | (v[1].X & 0xffff0000u); \
                        d[(4 * W + 2) * 64] = (v[2].X & 0xffffu) | (v[3].X << 16);  d[(4 * W + 3) * 64] = (v[2].X >> 16) | (v[3].X & 0xffff0000u);
                        WR_DQ_ST(0, x) WR_DQ_ST(1, y) WR_DQ_ST(2, z) WR_DQ_ST(3, w)
#undef WR_DQ_ST
                    }
                }
            }
            const bool mine = tg < n_data[0] || tg < n_data[1] || tg < n_data[2] || tg < n_data[3];
            // six steps: their coded bits A, B of the four frames as bytes (1 = a one was received), and whether the
            // position was transmitted at all (wave-uniform: one rate)
            const uint32_t* te = WR_SRC_TABLE.e + (enc_u * WR_DEC_TAB_STRIDE + tt_u);
            const uint32_t blk_lds = sym_lds + (uint32_t)((sym_u % sym_blk) * nw_u) * 1024u;      // this symbol's rows (4 rows of 256 B per word)
            uint32_t e[6];
#pragma unroll
            for (int P = 0; P < 6; P++) e[P] = te[P];
            if (__any(mine)) {
#pragma unroll
                for (int Q = 0; Q < 6; Q += 3) {
                    uint32_t Ms[3][2][2];
                    {
                        uint32_t ra[6], wa[6], rc[6], wc[6];
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            // 16-bit plane p = bits 4..8 of the entry: word p >> 1, half p & 1 -> row 4 (p >> 1) + (p & 1) (+ 2: frames 2, 3)
                            const uint32_t pa = (e[Q + k] >> 4) & 0x1fu, pb = (e[Q + k] >> 20) & 0x1fu;
                            ra[k] = blk_lds + (4u * (pa >> 1) + (pa & 1u)) * 256u;
                            ra[3 + k] = blk_lds + (4u * (pb >> 1) + (pb & 1u)) * 256u;
                            rc[k] = ra[k] + 512u;
                            rc[3 + k] = ra[3 + k] + 512u;
                        }
                        lds_rows6(ra, wa);
                        lds_rows6(rc, wc);
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const uint32_t ea = e[Q + k] & 0xffffu, eb = e[Q + k] >> 16;
                            const uint32_t va = k1 & (((ea >> 9) & 1u) - 1u), vb = k1 & (((eb >> 9) & 1u) - 1u);
                            const uint32_t ta = (((wa[k] >> (ea & 15u)) & 0x00010001u) | (((wc[k] >> (ea & 15u)) & 0x00010001u) << 8)) & va;
                            const uint32_t tb = (((wa[3 + k] >> (eb & 15u)) & 0x00010001u) | (((wc[3 + k] >> (eb & 15u)) & 0x00010001u) << 8)) & vb;
                            const uint32_t nv = va + vb;
                            Ms[k][0][0] = ta + tb;
                            Ms[k][0][1] = ta + vb - tb;
                            Ms[k][1][1] = nv - Ms[k][0][0];
                            Ms[k][1][0] = nv - Ms[k][0][1];
                        }
                    }
                    if constexpr (FIRST) {
                        // the first six steps without comparisons: the metric of state s = the branch metrics along its one
                        // path from state 0 (input bits = the bits of s, oldest first); survivor bits 0 (0x77 per byte: the six
                        // inverted positions)
                        // (in place in pm: level t reads entries below 2^t and writes entries 2 pp, 2 pp + 1, pp descending)
                        if (Q == 0) pm[0] = 0u;
#pragma unroll
                        for (int t = Q; t < Q + 3; t++) {
#pragma unroll
                            for (int pp = (1 << t) - 1; pp >= 0; pp--) {
                                const int c0 = pp << 1, c1 = (pp << 1) | 1;      // register contents: previous state, input bit
                                const uint32_t base_m = pm[pp];
                                pm[c1] = base_m + Ms[t - Q][parity_of(c1 & 0155)][parity_of(c1 & 0117)];
                                pm[c0] = base_m + Ms[t - Q][parity_of(c0 & 0155)][parity_of(c0 & 0117)];
                            }
                        }
                        if (Q == 3) {
#pragma unroll
                            for (int P = 0; P < 6; P++) {
                                uint4* sp = reinterpret_cast<uint4*>(surv + ((size_t)P * 64 + lane) * 8);
                                sp[0] = make_uint4(0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u);
                                sp[1] = make_uint4(0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u);
                            }
                        }
                    } else {
#define WR_ACS_Q(P, K) acs_step_q<P>(pm, Ms[K], sel0, surv + ((size_t)(tg + P) * 64 + lane) * 8);
                        if (Q == 0) { WR_ACS_Q(0, 0) WR_ACS_Q(1, 1) WR_ACS_Q(2, 2) }
                        else        { WR_ACS_Q(3, 0) WR_ACS_Q(4, 1) WR_ACS_Q(5, 2) }
#undef WR_ACS_Q
                    }
                }
            }
            {
                bool endh[4];
                bool any_end = false;
#pragma unroll
                for (int h = 0; h < 4; h++) { endh[h] = tg < n_data[h] && tg + 6 == n_data[h]; any_end |= endh[h]; }
                if (__any(any_end)) {
                    // a frame just ended (register phase 0 again): smallest metric, lowest state
                    int bs4[4];
                    best_states4(pm, bs4);
#pragma unroll
                    for (int h = 0; h < 4; h++)
                        if (endh[h]) best[h] = bs4[h];
                }
            }
            tt_u += 6;
            if (OVL && pend.active) tb_pending_group(pend, lane, frames, psdu_all, psdu_stride, ft);
            if (SPEC && n_fast > 0) {
                if (sw.active) spec_walk_group(sw, lane, surv, dbits, link_ref, fb);
                const int done = tg + 6;
                if (done % 96 == 0 && done + 96 <= n_max) {
                    // boundary B_j, j = done / 96, with another whole block of trellis steps ahead: the walk through block j - 1
                    // starts in the state with the smallest metric (register phase 0: pm[s] is state s)
                    uint32_t S = 0u;
                    int bs4[4];
                    best_states4(pm, bs4);
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        sw.st[h] = bs4[h];
                        S |= (uint32_t)bs4[h] << (8 * h);
#pragma unroll
                        for (int w = 0; w < 3; w++) sw.aw[h][w] = 0u;
                    }
                    rec[(size_t)(done / 96) * 64 + lane] = S;
                    link_ref = cur_start;
                    cur_start = S;
                    sw.blk = done / 96 - 1;
                    sw.grp = 15;
                    sw.active = true;
                }
            }
        };
        group(std::true_type{}, 0);
        for (int tg = 6; tg < n_max; tg += 6) group(std::false_type{}, tg);
        if (OVL) { while (pend.active) tb_pending_group(pend, lane, frames, psdu_all, psdu_stride, ft); }      // a pending task with a longer trellis: the rest of it
        __threadfence_block();
        // ---- traceback of the four frames of a lane: 32 decoded bits per word, words stored [word][h][lane].  A step of
        //      a frame reads ITS two survivor words of the row (compile-time positions), picks the half by bit 5 of the
        //      state and bit (state & 31) ^ 3 of it.  Fast form when all frames of the wave run the full n_max steps:
        //      blocks of 96 steps = three whole words; before step t the state holds the decoded bits u_t .. u_(t-5), so
        //      they leave six at a time.  General form: one bit per step, every access predicated on the frame's length. ----
        {
            int st[4] = { best[0], best[1], best[2], best[3] };
            uint32_t word[4] = { 0u, 0u, 0u, 0u };
#define WR_DQ_PICK(H, LO, HI)                                                                             \
            (__builtin_amdgcn_ubfe((((st[H] & 32) ? (HI) : (LO)) ^ 0x77777777u), (uint32_t)((st[H] & 31) ^ 3), 1u))
            for (int t1 = n_max - 1; t1 >= n_fast; t1 -= 4) {
                // (one array per word position: a select between two elements of ONE array becomes an indexed access, and the
                // array then lives in scratch memory)
                uint32_t l0[4], u0[4], l1[4], u1[4], l2[4], u2[4], l3[4], u3[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int t = t1 - k;
                    uint4 a = make_uint4(0u, 0u, 0u, 0u), b = make_uint4(0u, 0u, 0u, 0u);
                    if (t >= n_fast && (t < n_data[0] || t < n_data[1] || t < n_data[2] || t < n_data[3])) {
                        const uint4* sp = reinterpret_cast<const uint4*>(surv + ((size_t)t * 64 + lane) * 8);
                        a = sp[0];
                        b = sp[1];
                    }
                    // a = the low words (states 0..31), b = the high words, by byte index; byte index of frame h: 0, 2, 1, 3
                    l0[k] = a.x; u0[k] = b.x; l1[k] = a.z; u1[k] = b.z; l2[k] = a.y; u2[k] = b.y; l3[k] = a.w; u3[k] = b.w;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int t = t1 - k;
                    const uint32_t hb4[4] = { WR_DQ_PICK(0, l0[k], u0[k]), WR_DQ_PICK(1, l1[k], u1[k]),
                                              WR_DQ_PICK(2, l2[k], u2[k]), WR_DQ_PICK(3, l3[k], u3[k]) };
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        if (t >= n_fast && t < n_data[h]) {
                            word[h] |= (uint32_t)(st[h] & 1) << (t & 31);
                            st[h] = (st[h] >> 1) | (int)(hb4[h] << 5);
                            if ((t & 31) == 0) { dbits[(size_t)(t >> 5) * 256 + 64 * h + lane] = word[h]; word[h] = 0; }
                        }
                    }
                }
            }
            if (OVL && n_fast > 0) {
                // the blocks of 96 steps are walked back while the wave's next task runs (or right after the task loop)
                pend.active = true;
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    pend.st[h] = st[h];
                    pend.slot[h] = n_data[h] > 0 ? slot_of[h] : 0xffffffffu;
#pragma unroll
                    for (int w = 0; w < 3; w++) pend.aw[h][w] = 0u;
                }
                pend.blk = n_fast / 96 - 1;
                pend.grp = 15;
                pend.surv = surv;
                pend.dbits = dbits;
                buf ^= 1;
                continue;
            }
            for (int blk = n_fast / 96 - 1; blk >= 0; blk--) {
                // the 96 decoded bits of the block per frame: the groups come from the top step down, so every six steps the
                // register triple moves up by six and the six bits u_(t-5) .. u_t enter at the bottom (a rolled loop: 16 groups)
                uint32_t aw[4][3] = { { 0u, 0u, 0u }, { 0u, 0u, 0u }, { 0u, 0u, 0u }, { 0u, 0u, 0u } };
                const uint32_t* srow = surv + ((size_t)(blk * 96 + 90) * 64 + lane) * 8;       // the rows of the group's six steps
#pragma unroll 1
                for (int grp = 15; grp >= 0; grp--, srow -= 6 * 512) {
                    uint32_t l0[6], u0[6], l1[6], u1[6], l2[6], u2[6], l3[6], u3[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const uint4* sp = reinterpret_cast<const uint4*>(srow + (size_t)k * 512);
                        const uint4 a = sp[0], b = sp[1];
                        l0[k] = a.x; u0[k] = b.x; l1[k] = a.z; u1[k] = b.z; l2[k] = a.y; u2[k] = b.y; l3[k] = a.w; u3[k] = b.w;
                    }
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const uint32_t v = __builtin_bitreverse32((uint32_t)st[h]) >> 26;      // u_(t-5) .. u_t, oldest in bit 0
                        aw[h][2] = __builtin_amdgcn_alignbit(aw[h][2], aw[h][1], 26);      // the triple moves UP by six ...
                        aw[h][1] = __builtin_amdgcn_alignbit(aw[h][1], aw[h][0], 26);
                        aw[h][0] = (aw[h][0] << 6) | v;                                      // ... the older steps' bits enter below
                    }
#pragma unroll
                    for (int q = 5; q >= 0; q--) {
                        const uint32_t h0 = WR_DQ_PICK(0, l0[q], u0[q]), h1 = WR_DQ_PICK(1, l1[q], u1[q]);
                        const uint32_t h2 = WR_DQ_PICK(2, l2[q], u2[q]), h3 = WR_DQ_PICK(3, l3[q], u3[q]);
                        st[0] = (st[0] >> 1) | (int)(h0 << 5);
                        st[1] = (st[1] >> 1) | (int)(h1 << 5);
                        st[2] = (st[2] >> 1) | (int)(h2 << 5);
                        st[3] = (st[3] >> 1) | (int)(h3 << 5);
                    }
                }
#pragma unroll
                for (int w = 0; w < 3; w++)
#pragma unroll
                    for (int h = 0; h < 4; h++) dbits[(size_t)(blk * 3 + w) * 256 + 64 * h + lane] = aw[h][w];
                if (SPEC && blk > 0) {
                    // Boundary B_blk: a speculative walk started here (every boundary below the last whole block has one) in the states
                    // S.  A frame whose true path stands in that state, with every link below intact, has joined the chain of the
                    // walks: what lies below is decoded already.  The wave stops when all its frames have (from then on the
                    // paths coincide, so a frame that joined earlier only rewrote its own bits).
                    const uint32_t S = rec[(size_t)blk * 64 + lane];
                    bool joined = true;
#pragma unroll
                    for (int h = 0; h < 4; h++) joined = joined && st[h] == (int)((S >> (8 * h)) & 63u) && fb[h] >= blk;
                    if (__all(joined)) break;
                }
            }
#undef WR_DQ_PICK
        }
        __threadfence_block();
        // ---- descramble, bytes, CRC-32 ----
#pragma unroll
        for (int h = 0; h < 4; h++) {
            if (n_data[h] > 0) {
                const uint32_t slot = slot_of[h];
                finish_frame<256>(dbits + 64 * h + lane, frames[slot].psdu_len, psdu_all + (size_t)slot * psdu_stride,
                                  ((reinterpret_cast<uintptr_t>(psdu_all) | psdu_stride) & 3) == 0, frames + slot, frames[slot].flags, ft);
            }
        }
    }
    if (OVL) { while (pend.active) tb_pending_group(pend, lane, frames, psdu_all, psdu_stride, ft); }      // the wave's last deferred task: on its own
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
                         const uint32_t* __restrict__ hbits_all, uint8_t* __restrict__ psdu_all, uint32_t psdu_stride,
                         uint8_t* __restrict__ scratch, size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves_total)
{
    __shared__ uint32_t tile_all[4][60 * 12];
    __shared__ uint32_t src_tab[8 * WR_DEC_TAB_STRIDE];
    copy_src_table(src_tab);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4 + wv;
    if (wave >= n_waves_total) return;
    uint32_t* tile = tile_all[wv];
    uint64_t* dec = reinterpret_cast<uint64_t*>(scratch + (size_t)wave * scratch_stride);     // survivor word per step
    uint64_t* words = dec + n_steps_cap;                                                      // decoded bits, 60 per word

    // trellis constants of state `lane`: predecessors p0 = s >> 1 and p1 = p0 | 32, input bit s & 1
    const int s = lane, u = s & 1, p0 = s >> 1, p1 = (s >> 1) | 32;
    const int f0 = (p0 << 1) | u;
    const int a0 = __builtin_popcount(f0 & 0155) & 1, b0 = __builtin_popcount(f0 & 0117) & 1;
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    const uint32_t recip_tab[8] = { WR_RECIP32(24), WR_RECIP32(36), WR_RECIP32(48), WR_RECIP32(72),
                                    WR_RECIP32(96), WR_RECIP32(144), WR_RECIP32(192), WR_RECIP32(216) };

    for (uint32_t slot = wave; slot < n_slots; slot += n_waves_total) {
        const wifirx_frame fr = frames[slot];
        const int enc = fr.encoding & 7, psdu_len = fr.psdu_len;
        const int n_dbps = ndbps_tab[enc];
        const int n_sym = (16 + 8 * psdu_len + 6 + n_dbps - 1) / n_dbps;
        const bool ok = (fr.flags & WIFIRX_F_COMPLETE) && psdu_len <= (int)psdu_stride && psdu_len <= WIFIRX_MAX_PSDU &&
                        n_sym <= WIFIRX_MAX_SYM && n_sym <= (int)max_sym && (uint32_t)(n_sym * n_dbps) <= n_steps_cap;
        if (!ok) continue;                                       // wave-uniform: one frame per wave
        const int n_data = n_sym * n_dbps;
        const int n_words = enc < 2 ? 2 : enc < 4 ? 4 : enc < 6 ? 8 : 12;           // bit-plane words per symbol
        const uint32_t* fhb = hbits_all + (size_t)slot * max_sym * 12;
        const uint32_t recip = recip_tab[enc];
        const uint32_t* tab_enc = src_tab + enc * WR_DEC_TAB_STRIDE;

        // ---- add-compare-select: tiles of 60 symbols, chunks of 60 trellis steps ----
        int pm = (s == 0) ? 0 : (1 << 24);
        for (int sym0 = 0; sym0 < n_sym; sym0 += 60) {
            const int nsy = n_sym - sym0 < 60 ? n_sym - sym0 : 60;
            const int nw = nsy * n_words;
            __builtin_amdgcn_wave_barrier();
            for (int o = lane; o < nw; o += 64) tile[o] = fhb[sym0 * n_words + o];
            __builtin_amdgcn_wave_barrier();
            const int t_hi = (sym0 + nsy) * n_dbps;
            for (int t0 = sym0 * n_dbps; t0 < t_hi; t0 += WR_DEC_CHUNK) {
                const int t = t0 + lane;
                int ra, rb;
                gather_step(tile, sym0, n_words, tab_enc, n_dbps, recip, t, lane < WR_DEC_CHUNK && t < t_hi, ra, rb);
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

// longest trellis (in steps) among the frames decode_kernel would accept -> out[0]; how many of them there are per
// rate -> out[1 + enc]
__global__ __launch_bounds__(256)
void decode_maxsteps_kernel(uint32_t n_slots, uint32_t max_sym, const wifirx_frame* __restrict__ frames,
                            uint32_t psdu_stride, uint32_t* __restrict__ out)
{
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    uint32_t best = 0;
    uint32_t cnt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };          // wave-uniform (counted by ballot)
    const uint32_t n_round = (n_slots + 63u) & ~63u;         // whole waves walk the loop: the ballots need every lane
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        int enc = -1;
        if (i < n_slots) {
            const wifirx_frame fr = frames[i];
            const int nd = ndbps_tab[fr.encoding & 7], len = fr.psdu_len;
            const int n_sym = (16 + 8 * len + 6 + nd - 1) / nd;
            if ((fr.flags & WIFIRX_F_COMPLETE) && len <= (int)psdu_stride && len <= WIFIRX_MAX_PSDU &&
                n_sym <= WIFIRX_MAX_SYM && n_sym <= (int)max_sym) {
                uint32_t v = (uint32_t)(n_sym * nd);
                best = v > best ? v : best;
                enc = fr.encoding & 7;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) cnt[e] += (uint32_t)__builtin_popcountll(__ballot(enc == e));
    }
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        uint32_t o = __shfl_xor(best, k, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) {
        if (best) atomicMax(out, best);
#pragma unroll
        for (int e = 0; e < 8; e++) if (cnt[e]) atomicAdd(out + 1 + e, cnt[e]);
    }
}

// The decodable frames of a batch grouped by rate: perm[start[enc] + k] = slot of the k-th such frame found (the order
// inside a run is whatever the atomics make it -- which frames share a wave does not change any frame's result).  The
// runs start on task boundaries (start[] from the host), so every task of decode_kernel is of one rate; entries between
// the runs stay 0xffffffff (set by the caller).
struct PermStarts { uint32_t s[8]; };
__global__ __launch_bounds__(256)
void decode_perm_kernel(uint32_t n_slots, uint32_t max_sym, const wifirx_frame* __restrict__ frames, uint32_t psdu_stride,
                        PermStarts start, uint32_t* __restrict__ cursor, uint32_t* __restrict__ perm)
{
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    const int lane = threadIdx.x & 63;
    const uint32_t n_round = (n_slots + 63u) & ~63u;         // whole waves walk the loop: the ballots need every lane
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        int enc = -1;
        if (i < n_slots) {
            const wifirx_frame fr = frames[i];
            const int e = fr.encoding & 7, nd = ndbps_tab[e], len = fr.psdu_len;
            const int n_sym = (16 + 8 * len + 6 + nd - 1) / nd;
            if ((fr.flags & WIFIRX_F_COMPLETE) && len <= (int)psdu_stride && len <= WIFIRX_MAX_PSDU &&
                n_sym <= WIFIRX_MAX_SYM && n_sym <= (int)max_sym)
                enc = e;
        }
        // one atomic per wave and rate (a million lanes adding to eight counters would queue up behind each other); the
        // frames of a wave keep their order inside the run
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint64_t m = __ballot(enc == e);
            if (m == 0) continue;                                   // wave-uniform
            uint32_t at = 0;
            if (lane == (int)__builtin_ctzll(m)) at = atomicAdd(cursor + e, (uint32_t)__builtin_popcountll(m));
            at = (uint32_t)__builtin_amdgcn_readlane((int)at, (int)__builtin_ctzll(m));
            if (enc == e) perm[start.s[e] + at + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
        }
    }
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

extern "C" hipError_t wr_launch_decode_pack(hipStream_t st, uint32_t n_slots, uint32_t max_sym, const wifirx_frame* frames,
                                            const uint8_t* idx, uint32_t psdu_stride, uint32_t* hbits, uint32_t n_steps_cap)
{
    if (n_slots == 0) return hipSuccess;
    uint32_t blocks = (n_slots + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(wr::decode_pack_kernel, dim3(blocks), dim3(256), 0, st, n_slots, max_sym, frames, idx, psdu_stride,
                       hbits, n_steps_cap);
    return hipGetLastError();
}

extern "C" hipError_t wr_launch_decode(hipStream_t st, uint32_t n_slots, uint32_t max_sym, wifirx_frame* frames,
                                       const uint32_t* hbits, uint8_t* psdu, uint32_t psdu_stride, uint8_t* scratch,
                                       size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves, uint32_t frames_per_wave,
                                       const uint32_t* perm, uint32_t n_virtual)
{
    if (n_slots == 0 || n_waves == 0) return hipSuccess;
    if (!perm) n_virtual = n_slots;
    // the tasks with one rate, then those with several (each kernel skips the other's; the scratch is shared).  With a
    // permutation every task is of one rate and the second kernel has nothing to do.
    uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL(wr::decode_kernel<false>, dim3(blocks), dim3(256), 0, st, n_slots, max_sym, frames, hbits, psdu,
                       psdu_stride, scratch, scratch_stride, n_steps_cap, n_waves, frames_per_wave, perm, n_virtual);
    if (!perm)
        hipLaunchKernelGGL(wr::decode_kernel<true>, dim3(blocks), dim3(256), 0, st, n_slots, max_sym, frames, hbits, psdu,
                           psdu_stride, scratch, scratch_stride, n_steps_cap, n_waves, frames_per_wave, perm, n_virtual);
    return hipGetLastError();
}

extern "C" hipError_t wr_launch_decode_q(hipStream_t st, uint32_t n_slots, uint32_t max_sym, wifirx_frame* frames,
                                         const uint32_t* hbits, uint8_t* psdu, uint32_t psdu_stride, uint8_t* scratch,
                                         size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves, uint32_t frames_per_wave,
                                         const uint32_t* perm, uint32_t n_virtual, int has_64qam, int overlap)
{
    // overlap: 0 = trace-back behind the task, 1 = under the wave's next task (double scratch slice), 2 = speculative walks under the
    // task's own add-compare-select (slice + (n_steps_cap / 96 + 2) x 256 bytes of start states)
    if (n_slots == 0 || n_waves == 0) return hipSuccess;
    if (!perm) n_virtual = n_slots;
#define WR_LAUNCH_Q(ROWS, MODE) hipLaunchKernelGGL((wr::decode_q_kernel<ROWS, MODE>), dim3((n_waves + 3) / 4), dim3(256), 0, st, n_slots, max_sym, \
                                                   frames, hbits, psdu, psdu_stride, scratch, scratch_stride, n_steps_cap, n_waves, frames_per_wave, perm, n_virtual)
    if (has_64qam) { if (overlap == 2) WR_LAUNCH_Q(48, 2); else if (overlap) WR_LAUNCH_Q(48, 1); else WR_LAUNCH_Q(48, 0); }
    else           { if (overlap == 2) WR_LAUNCH_Q(32, 2); else if (overlap) WR_LAUNCH_Q(32, 1); else WR_LAUNCH_Q(32, 0); }
#undef WR_LAUNCH_Q
    return hipGetLastError();
}

extern "C" hipError_t wr_launch_decode_perm(hipStream_t st, uint32_t n_slots, uint32_t max_sym, const wifirx_frame* frames,
                                            uint32_t psdu_stride, const uint32_t* starts8, uint32_t* cursor8, uint32_t* perm)
{
    if (n_slots == 0) return hipSuccess;
    wr::PermStarts ps;
    for (int e = 0; e < 8; e++) ps.s[e] = starts8[e];
    uint32_t blocks = (n_slots + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wr::decode_perm_kernel, dim3(blocks), dim3(256), 0, st, n_slots, max_sym, frames, psdu_stride, ps, cursor8, perm);
    return hipGetLastError();
}

extern "C" hipError_t wr_launch_decode_small(hipStream_t st, uint32_t n_slots, uint32_t max_sym, wifirx_frame* frames,
                                             const uint32_t* hbits, uint8_t* psdu, uint32_t psdu_stride, uint8_t* scratch,
                                             size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves)
{
    if (n_slots == 0 || n_waves == 0) return hipSuccess;
    hipLaunchKernelGGL(wr::decode_small_kernel, dim3((n_waves + 3) / 4), dim3(256), 0, st, n_slots, max_sym, frames, hbits, psdu,
                       psdu_stride, scratch, scratch_stride, n_steps_cap, n_waves);
    return hipGetLastError();
}
