// wr_decode.hip -- decode_mac on the device: replaces ieee802_11.decode_mac
// (gnu_radio/IRS_AP.py:272,291-292): demap indices to bits, de-interleave, de-puncture, Viterbi
// K=7 (133,171), descramble, CRC-32.
//
// One wavefront decodes 64 frames, lane <-> frame.  Every lane keeps the 64 path metrics of its own
// frame in 64 VGPRs and runs the add-compare-select of all 32 butterflies of a trellis step as
// straight-line code (no cross-lane traffic at all); metrics are updated in place, which rotates the
// state <-> register map by one bit per step, so the code is unrolled over the 6 steps after which the
// map is the identity again (all rates have n_data % 12 == 0).  The 64 survivor bits of a step are
// shifted into two VGPRs by v_addc (carry-in = the compare result) and stored as one coalesced
// 512-byte row per wave and step; traceback, descrambling and the CRC then run per lane as well.
//
// The received coded bits are gathered beforehand by the whole wave for one frame at a time
// (lane <-> trellis step: de-puncture + de-interleave + bit extraction from the hard decisions) into
// per-frame bit masks that go through LDS to the lane that owns the frame.
//
// Results are bit-identical to the oracle's viterbi_decode(): same metrics (Hamming, erasures free),
// same tie rule (the survivor with older bit 0 wins), same final state rule (smallest metric, lowest
// state), start state 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wifirx.h"
#include "wr_kernels.h"

namespace wr {

#ifndef WR_DEC_ASM
#define WR_DEC_ASM 1
#endif
#ifndef WR_DEC_WAVES_PER_SIMD
#define WR_DEC_WAVES_PER_SIMD 4
#endif
#define WR_DEC_CHUNK      60                   // trellis steps per gathered mask word (10 groups of 6)
#define WR_DEC_SEG_CHUNKS 4                    // chunks gathered per segment: 240 steps (8 KB of LDS per wave)
#define WR_DEC_LDS_WORDS  (WR_DEC_SEG_CHUNKS * 4 * 64)   // per wave, 8-byte words: [chunk][A1,AV,B1,BV][frame]

// the coded bit at position `ci` of the de-punctured stream of a frame: 0/1, or 2 when punctured.
// Rate parameters are template constants: every division below is by a compile-time constant.
template <int PUNCT, int N_BPSC>
__device__ __forceinline__ int coded_bit(const uint8_t* __restrict__ idx, int ci)
{
    constexpr int n_cbps = 48 * N_BPSC;
    constexpr int s = (N_BPSC / 2) < 1 ? 1 : (N_BPSC / 2);
    int pidx;
    if (PUNCT == 0) {
        pidx = ci;
    } else if (PUNCT == 1) {               // 2/3: every 4th bit dropped
        int r = ci & 3;
        if (r == 3) return 2;
        pidx = (ci >> 2) * 3 + r;
    } else {                               // 3/4: bits 3,4 of every 6 dropped
        int g = ci / 6, r = ci - 6 * g;
        if (r == 3 || r == 4) return 2;
        pidx = g * 4 + (r < 3 ? r : 3);
    }
    int sym = pidx / n_cbps, k = pidx - sym * n_cbps;
    int i = (n_cbps >> 4) * (k & 15) + (k >> 4);
    int j = s * (i / s) + (i + n_cbps - (16 * i) / n_cbps) % s;
    int carrier = j / N_BPSC, bit = j - carrier * N_BPSC;
    return (idx[sym * 48 + carrier] >> bit) & 1;
}

// gathers one chunk (lane <-> step t0 + lane) of one frame into the four mask words
template <int PUNCT, int N_BPSC>
__device__ __forceinline__ void gather_chunk(const uint8_t* __restrict__ fidx, int t, bool in_range,
                                             uint64_t& A1, uint64_t& AV, uint64_t& B1, uint64_t& BV)
{
    int ra = 2, rb = 2;
    if (in_range) {
        ra = coded_bit<PUNCT, N_BPSC>(fidx, 2 * t);
        rb = coded_bit<PUNCT, N_BPSC>(fidx, 2 * t + 1);
    }
    A1 = __ballot(ra == 1); AV = __ballot(ra != 2);
    B1 = __ballot(rb == 1); BV = __ballot(rb != 2);
}

constexpr __host__ __device__ int rotr6(int s, int p) { return ((s >> p) | (s << (6 - p))) & 63; }
constexpr __host__ __device__ int parity_of(int v) { return __builtin_popcount(v) & 1; }

// new = min(c0, c1), decision = (c1 < c0) shifted into `word` (word = 2*word + decision)
__device__ __forceinline__ int acs_one(int c0, int c1, uint32_t& word)
{
#if WR_DEC_ASM
    uint32_t w = word;
    int m;
    uint64_t cc;
    asm("v_cmp_lt_i32 %2, %4, %5\n\t"
        "v_cndmask_b32 %0, %5, %4, %2\n\t"
        "v_addc_co_u32 %1, %2, %3, %3, %2"
        : "=&v"(m), "=v"(w), "=&s"(cc)
        : "v"(w), "v"(c1), "v"(c0));
    word = w;
    return m;
#else
    const bool d = c1 < c0;
    word = word + word + (uint32_t)d;
    return d ? c1 : c0;
#endif
}

// one trellis step at register phase P: logical state s lives in pm[rotr6(s, P)].
// M[a][b]: branch metric of a transition whose expected coded pair is (a, b).
template <int P>
__device__ __forceinline__ void acs_step(int (&pm)[64], const int (&M)[2][2], uint32_t& dlo, uint32_t& dhi)
{
    uint32_t acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };      // acc[k]: states 8k..8k+7, first state in the top bit
    // logical order 0..63 so that the survivor bit of state s ends at bit (31 - s%32) of dlo (s<32) / dhi
#pragma unroll
    for (int j = 0; j < 32; j++) {
        constexpr int dummy = 0; (void)dummy;
        const int a = parity_of((j << 1) & 0155), b = parity_of((j << 1) & 0117);
        const int m = M[a][b], mb = M[a ^ 1][b ^ 1];
        const int r0 = rotr6(j, P), r1 = rotr6(j + 32, P);
        const int p0 = pm[r0], p1 = pm[r1];
        uint32_t& w = acc[j >> 2];
        // state 2j (input bit 0): from j with m, from j+32 with mb;  state 2j+1: metrics swapped
        const int n0 = acs_one(p0 + m, p1 + mb, w);
        const int n1 = acs_one(p0 + mb, p1 + m, w);
        pm[r0] = n0;      // = register of logical state 2j at phase P+1
        pm[r1] = n1;      // = register of logical state 2j+1 at phase P+1
    }
    dlo = (acc[0] << 24) | (acc[1] << 16) | (acc[2] << 8) | acc[3];
    dhi = (acc[4] << 24) | (acc[5] << 16) | (acc[6] << 8) | acc[7];
}

__device__ __forceinline__ uint32_t crc32_bit(uint32_t c, uint32_t bit)
{
    uint32_t x = (c ^ bit) & 1u;
    return (c >> 1) ^ (0xedb88320u & (0u - x));
}

__global__ __launch_bounds__(256, WR_DEC_WAVES_PER_SIMD)
void decode_kernel(uint32_t n_slots, uint32_t max_sym, wifirx_frame* __restrict__ frames,
                   const uint8_t* __restrict__ idx_all, uint8_t* __restrict__ psdu_all, uint32_t psdu_stride,
                   uint8_t* __restrict__ scratch, size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves_total)
{
    __shared__ uint64_t lds_all[4][WR_DEC_LDS_WORDS];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4 + wv;
    if (wave >= n_waves_total) return;
    uint64_t* lds = lds_all[wv];
    const size_t n_data_cap = n_steps_cap;               // trellis steps the scratch slice of a wave holds
    uint64_t* surv = reinterpret_cast<uint64_t*>(scratch + (size_t)wave * scratch_stride);   // [step][lane]
    uint32_t* dbits = reinterpret_cast<uint32_t*>(surv + n_data_cap * 64);                   // [word][lane]

    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };

    for (uint32_t base = wave * 64; base < n_slots; base += n_waves_total * 64) {
        // ---- my frame ----
        const uint32_t slot = base + lane;
        wifirx_frame fr;
        fr.flags = 0; fr.encoding = 0; fr.psdu_len = 0;
        if (slot < n_slots) fr = frames[slot];
        const int enc = fr.encoding & 7, psdu_len = fr.psdu_len;
        const int n_dbps = ndbps_tab[enc];
        const int n_sym = (16 + 8 * psdu_len + 6 + n_dbps - 1) / n_dbps;
        const bool valid = slot < n_slots && (fr.flags & WIFIRX_F_COMPLETE) && psdu_len <= (int)psdu_stride &&
                           psdu_len <= WIFIRX_MAX_PSDU && n_sym <= WIFIRX_MAX_SYM && n_sym <= (int)max_sym &&
                           (uint32_t)(n_sym * n_dbps) <= n_steps_cap;
        const int n_data = valid ? n_sym * n_dbps : 0;          // multiple of 12
        int n_max = n_data;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            int o = __shfl_xor(n_max, k, 64);
            n_max = o > n_max ? o : n_max;
        }
        if (n_max == 0) continue;
        const uint64_t valid_mask = __ballot(valid);

        // ---- add-compare-select ----
        int pm[64];
#pragma unroll
        for (int s = 0; s < 64; s++) pm[s] = (s == 0) ? 0 : (1 << 24);
        int best = 0;                                   // final state, taken when my frame ends
        const int seg_steps = WR_DEC_CHUNK * WR_DEC_SEG_CHUNKS;
        for (int seg0 = 0; seg0 < n_max; seg0 += seg_steps) {
            // gather the received coded bits of this segment: one frame at a time, lane <-> step
            __builtin_amdgcn_wave_barrier();
            for (int f = 0; f < 64; f++) {
                if (!((valid_mask >> f) & 1)) continue;
                const int f_enc = __builtin_amdgcn_readlane(enc, f);
                const int f_ndata = __builtin_amdgcn_readlane(n_data, f);
                if (seg0 >= f_ndata) continue;
                const uint8_t* fidx = idx_all + (size_t)(base + f) * max_sym * 48;
                for (int c = 0; c < WR_DEC_SEG_CHUNKS; c++) {
                    const int t0 = seg0 + c * WR_DEC_CHUNK;
                    if (t0 >= f_ndata) break;
                    const int t = t0 + lane;
                    const bool in_range = lane < WR_DEC_CHUNK && t < f_ndata;
                    uint64_t A1, AV, B1, BV;
                    switch (f_enc) {
                    case 0:  gather_chunk<0, 1>(fidx, t, in_range, A1, AV, B1, BV); break;
                    case 1:  gather_chunk<2, 1>(fidx, t, in_range, A1, AV, B1, BV); break;
                    case 2:  gather_chunk<0, 2>(fidx, t, in_range, A1, AV, B1, BV); break;
                    case 3:  gather_chunk<2, 2>(fidx, t, in_range, A1, AV, B1, BV); break;
                    case 4:  gather_chunk<0, 4>(fidx, t, in_range, A1, AV, B1, BV); break;
                    case 5:  gather_chunk<2, 4>(fidx, t, in_range, A1, AV, B1, BV); break;
                    case 6:  gather_chunk<1, 6>(fidx, t, in_range, A1, AV, B1, BV); break;
                    default: gather_chunk<2, 6>(fidx, t, in_range, A1, AV, B1, BV); break;
                    }
                    if (lane < 4) {
                        uint64_t wsel = lane == 0 ? A1 : lane == 1 ? AV : lane == 2 ? B1 : BV;
                        lds[(c * 4 + lane) * 64 + f] = wsel;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // my frame's steps of this segment, six at a time
            for (int c = 0; c < WR_DEC_SEG_CHUNKS; c++) {
                const int t0 = seg0 + c * WR_DEC_CHUNK;
                if (t0 >= n_max) break;
                const uint64_t A1 = lds[(c * 4 + 0) * 64 + lane], AV = lds[(c * 4 + 1) * 64 + lane];
                const uint64_t B1 = lds[(c * 4 + 2) * 64 + lane], BV = lds[(c * 4 + 3) * 64 + lane];
                for (int g = 0; g < WR_DEC_CHUNK / 6; g++) {
                    const int tg = t0 + 6 * g;
                    if (tg >= n_max) break;
                    const uint32_t a1 = (uint32_t)(A1 >> (6 * g)), av = (uint32_t)(AV >> (6 * g));
                    const uint32_t b1 = (uint32_t)(B1 >> (6 * g)), bv = (uint32_t)(BV >> (6 * g));
                    const bool mine = tg < n_data;
                    if (mine) {
#define WR_ACS(P)                                                                                         \
                        {                                                                                 \
                            const int va = (av >> P) & 1, vb = (bv >> P) & 1;                             \
                            const int ta = (a1 >> P) & 1 & va, tb = (b1 >> P) & 1 & vb, nv = va + vb;     \
                            int M[2][2];                                                                  \
                            M[0][0] = ta + tb;                                                            \
                            M[0][1] = ta + vb - tb;                                                       \
                            M[1][1] = nv - M[0][0];                                                       \
                            M[1][0] = nv - M[0][1];                                                       \
                            uint32_t dlo = 0, dhi = 0;                                                    \
                            acs_step<P>(pm, M, dlo, dhi);                                                 \
                            surv[(size_t)(tg + P) * 64 + lane] = ((uint64_t)dhi << 32) | dlo;             \
                        }
                        WR_ACS(0) WR_ACS(1) WR_ACS(2) WR_ACS(3) WR_ACS(4) WR_ACS(5)
#undef WR_ACS
                    }
                    if (__any(mine && tg + 6 == n_data)) {
                        // my frame just ended (register phase 0 again): smallest metric, lowest state
                        int bm = pm[0], bs = 0;
#pragma unroll
                        for (int s = 1; s < 64; s++) {
                            if (pm[s] < bm) { bm = pm[s]; bs = s; }
                        }
                        if (mine && tg + 6 == n_data) best = bs;
                    }
                }
            }
        }
        __threadfence_block();
        // ---- traceback: 32 decoded bits per word, words stored [word][lane] ----
        {
            int st = best;
            uint32_t word = 0;
            for (int t = n_max - 1; t >= 0; t--) {
                const bool mine = t < n_data;
                uint64_t sv = 0;
                if (mine) sv = surv[(size_t)t * 64 + lane];
                const uint32_t half = (st < 32) ? (uint32_t)sv : (uint32_t)(sv >> 32);
                const uint32_t h = (half >> (31 - (st & 31))) & 1u;
                if (mine) {
                    word |= (uint32_t)(st & 1) << (t & 31);
                    st = (st >> 1) | (int)(h << 5);
                    if ((t & 31) == 0) {
                        dbits[(size_t)(t >> 5) * 64 + lane] = word;
                        word = 0;
                    }
                }
            }
        }
        __threadfence_block();
        // ---- descramble (x^7+x^4+1, state from the first 7 decoded bits), bytes, CRC-32 ----
        if (valid) {
            uint32_t w0 = dbits[lane];
            int state = 0;
#pragma unroll
            for (int i = 0; i < 7; i++) state |= (int)((w0 >> i) & 1) << (6 - i);
            // positions 7..15 belong to the SERVICE field: advance the scrambler
            for (int i = 7; i < 16; i++) {
                int fb = ((state >> 6) ^ (state >> 3)) & 1;
                state = ((state << 1) & 0x7e) | fb;
            }
            uint8_t* psdu = psdu_all + (size_t)slot * psdu_stride;
            uint32_t crc = 0xffffffffu;
            uint32_t cur = w0;
            int wi = 0;
            for (int b = 0; b < psdu_len; b++) {
                uint32_t byte = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = 16 + 8 * b + k;
                    if ((i >> 5) != wi) { wi = i >> 5; cur = dbits[(size_t)wi * 64 + lane]; }
                    const uint32_t fb = (uint32_t)(((state >> 6) ^ (state >> 3)) & 1);
                    state = ((state << 1) & 0x7e) | (int)fb;
                    const uint32_t d = ((cur >> (i & 31)) & 1u) ^ fb;
                    byte |= d << k;
                    crc = crc32_bit(crc, d);
                }
                psdu[b] = (uint8_t)byte;
            }
            crc = ~crc;
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
                                       size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves)
{
    if (n_slots == 0 || n_waves == 0) return hipSuccess;
    uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL(wr::decode_kernel, dim3(blocks), dim3(256), 0, st, n_slots, max_sym, frames, idx, psdu,
                       psdu_stride, scratch, scratch_stride, n_steps_cap, n_waves);
    return hipGetLastError();
}
