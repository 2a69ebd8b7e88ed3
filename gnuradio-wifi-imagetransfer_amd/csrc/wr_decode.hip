// wr_decode.hip -- decode_mac on the device: replaces ieee802_11.decode_mac
// (gnu_radio/IRS_AP.py:272,291-292): demap indices to bits, de-interleave, de-puncture, Viterbi
// K=7 (133,171), descramble, CRC-32.
//
// One wavefront decodes one frame; lane <-> trellis state (64 states = 64 lanes), so one
// add-compare-select step is two cross-lane reads, a handful of VALU ops and one ballot that
// yields the 64 survivor bits of the step at once.  Waves walk the frames grid-stride; survivor
// words live in a per-wave global scratch (L2 resident), never one per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wifirx.h"
#include "wr_kernels.h"

namespace wr {

__device__ __forceinline__ uint32_t crc32_update(uint32_t c, uint32_t byte)
{
    c ^= byte;
#pragma unroll
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
    return c;
}

// the coded bit at position `ci` of the de-punctured stream of a frame: 0/1, or 2 when punctured
__device__ __forceinline__ int coded_bit(const uint8_t* __restrict__ idx, int ci, int punct, int n_bpsc, int n_cbps)
{
    int pidx;
    if (punct == 0) {
        pidx = ci;
    } else if (punct == 1) {               // 2/3: every 4th bit dropped
        int r = ci & 3;
        if (r == 3) return 2;
        pidx = (ci >> 2) * 3 + r;
    } else {                               // 3/4: bits 3,4 of every 6 dropped
        int g = ci / 6, r = ci - 6 * g;
        if (r == 3 || r == 4) return 2;
        pidx = g * 4 + (r < 3 ? r : 3);
    }
    int sym = pidx / n_cbps, k = pidx - sym * n_cbps;
    int s = n_bpsc >> 1; if (s < 1) s = 1;
    int i = (n_cbps >> 4) * (k & 15) + (k >> 4);
    int j = s * (i / s) + (i + n_cbps - (16 * i) / n_cbps) % s;
    int carrier = j / n_bpsc, bit = j - carrier * n_bpsc;
    return (idx[sym * 48 + carrier] >> bit) & 1;
}

__global__ __launch_bounds__(256)
void decode_kernel(uint32_t n_slots, uint32_t max_sym, wifirx_frame* __restrict__ frames,
                   const uint8_t* __restrict__ idx_all, uint8_t* __restrict__ psdu_all, uint32_t psdu_stride,
                   uint8_t* __restrict__ scratch, size_t scratch_stride, uint32_t n_waves_total)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= n_waves_total) return;
    const size_t n_data_max = (size_t)max_sym * 216;
    uint64_t* dec = reinterpret_cast<uint64_t*>(scratch + (size_t)wave * scratch_stride);
    uint64_t* words = dec + n_data_max;

    // trellis constants of state `lane`
    const int s = lane, u = s & 1, p0 = s >> 1, p1 = (s >> 1) | 32;
    const int f0 = (p0 << 1) | u;
    const int a0 = __builtin_popcount(f0 & 0155) & 1, b0 = __builtin_popcount(f0 & 0117) & 1;
    const int nbpsc_tab[8] = { 1, 1, 2, 2, 4, 4, 6, 6 };
    const int ndbps_tab[8] = { 24, 36, 48, 72, 96, 144, 192, 216 };
    const int punct_tab[8] = { 0, 2, 0, 2, 0, 2, 1, 2 };

    for (uint32_t slot = wave; slot < n_slots; slot += n_waves_total) {
        const wifirx_frame fr = frames[slot];
        if (!(fr.flags & WIFIRX_F_COMPLETE)) continue;
        const int enc = fr.encoding, psdu_len = fr.psdu_len;
        if (psdu_len > (int)psdu_stride || psdu_len > WIFIRX_MAX_PSDU) continue;
        const int n_bpsc = nbpsc_tab[enc], n_dbps = ndbps_tab[enc], punct = punct_tab[enc], n_cbps = 48 * n_bpsc;
        const int n_sym = (16 + 8 * psdu_len + 6 + n_dbps - 1) / n_dbps;
        if (n_sym > WIFIRX_MAX_SYM || n_sym > (int)max_sym) continue;
        const int n_data = n_sym * n_dbps;
        const uint8_t* idx = idx_all + (size_t)slot * max_sym * 48;

        // ---- add-compare-select, 64 trellis steps per chunk ----
        int pm = (s == 0) ? 0 : (1 << 24);
        const int n_chunks = (n_data + 63) >> 6;
        for (int c = 0; c < n_chunks; c++) {
            int t = (c << 6) + lane;
            int ra = 2, rb = 2;
            if (t < n_data) {
                ra = coded_bit(idx, 2 * t, punct, n_bpsc, n_cbps);
                rb = coded_bit(idx, 2 * t + 1, punct, n_bpsc, n_cbps);
            }
            const uint64_t A1 = __ballot(ra == 1), AV = __ballot(ra != 2);
            const uint64_t B1 = __ballot(rb == 1), BV = __ballot(rb != 2);
            const int jn = min(64, n_data - (c << 6));
            uint64_t mydec = 0;
            for (int j = 0; j < jn; j++) {
                int sa = (int)((A1 >> j) & 1), va = (int)((AV >> j) & 1);
                int sb = (int)((B1 >> j) & 1), vb = (int)((BV >> j) & 1);
                int bm0 = (va & (sa ^ a0)) + (vb & (sb ^ b0));
                int bm1 = (va + vb) - bm0;
                int m0 = __shfl(pm, p0, 64) + bm0;
                int m1 = __shfl(pm, p1, 64) + bm1;
                bool sel = m1 < m0;
                pm = sel ? m1 : m0;
                uint64_t d = __ballot(sel);
                if (lane == j) mydec = d;
            }
            if (t < n_data) dec[t] = mydec;
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
        // ---- traceback, one chunk of survivor words in registers at a time ----
        for (int c = n_chunks - 1; c >= 0; c--) {
            int t = (c << 6) + lane;
            uint64_t dw = (t < n_data) ? dec[t] : 0;
            uint32_t lo = (uint32_t)dw, hi = (uint32_t)(dw >> 32);
            const int jn = min(64, n_data - (c << 6));
            uint64_t word = 0;
            for (int j = jn - 1; j >= 0; j--) {
                word |= (uint64_t)(st & 1) << j;
                uint32_t dlo = (uint32_t)__builtin_amdgcn_readlane((int)lo, j);
                uint32_t dhi = (uint32_t)__builtin_amdgcn_readlane((int)hi, j);
                uint32_t h = (st < 32 ? (dlo >> st) : (dhi >> (st - 32))) & 1u;
                st = (st >> 1) | (int)(h << 5);
            }
            if (lane == 0) words[c] = word;
        }
        __threadfence_block();
        // ---- descramble: x^7+x^4+1, state from the first 7 decoded bits ----
        uint64_t w0 = words[0];
        int state = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) state |= (int)((w0 >> i) & 1) << (6 - i);
        uint64_t seq_lo = 0, seq_hi = 0;             // feedback bit for decoded positions 7, 8, ... (period 127)
        for (int i = 0; i < 127; i++) {
            int fb = ((state >> 6) ^ (state >> 3)) & 1;
            if (i < 64) seq_lo |= (uint64_t)fb << i; else seq_hi |= (uint64_t)fb << (i - 64);
            state = ((state << 1) & 0x7e) | fb;
        }
        uint8_t* psdu = psdu_all + (size_t)slot * psdu_stride;
        for (int b = lane; b < psdu_len; b += 64) {
            unsigned byte = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                int i = 16 + 8 * b + k;
                int q = (i - 7) % 127;
                unsigned fb = (unsigned)(((q < 64 ? (seq_lo >> q) : (seq_hi >> (q - 64)))) & 1);
                unsigned d = (unsigned)((words[i >> 6] >> (i & 63)) & 1);
                byte |= (d ^ fb) << k;
            }
            psdu[b] = (uint8_t)byte;
        }
        __threadfence_block();
        // ---- CRC-32 over the PSDU incl. FCS: residue 0x2144DF1C ----
        uint32_t crc = 0xffffffffu;
        for (int b0_ = 0; b0_ < psdu_len; b0_ += 64) {
            int b = b0_ + lane;
            uint32_t mine = (b < psdu_len) ? psdu[b] : 0u;
            const int jn = min(64, psdu_len - b0_);
            for (int j = 0; j < jn; j++)
                crc = crc32_update(crc, (uint32_t)__builtin_amdgcn_readlane((int)mine, j));
        }
        crc = ~crc;
        if (lane == 0) {
            uint32_t fl = fr.flags | WIFIRX_F_DECODED;
            if (psdu_len >= 4 && crc == 558161692u) fl |= WIFIRX_F_CRC_OK; else fl &= ~WIFIRX_F_CRC_OK;
            frames[slot].flags = fl;
        }
    }
}

}  // namespace wr

extern "C" hipError_t wr_launch_decode(hipStream_t st, uint32_t n_slots, uint32_t max_sym, wifirx_frame* frames,
                                       const uint8_t* idx, uint8_t* psdu, uint32_t psdu_stride, uint8_t* scratch,
                                       size_t scratch_stride)
{
    if (n_slots == 0) return hipSuccess;
    uint32_t n_waves = n_slots < WR_DECODE_MAX_WAVES ? n_slots : WR_DECODE_MAX_WAVES;
    uint32_t blocks = (n_waves + 3) / 4;
    hipLaunchKernelGGL(wr::decode_kernel, dim3(blocks), dim3(256), 0, st, n_slots, max_sym, frames, idx, psdu,
                       psdu_stride, scratch, scratch_stride, n_waves);
    return hipGetLastError();
}
