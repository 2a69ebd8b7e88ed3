// wr_kernels.h -- host-visible launch interface of the wifirx HIP kernels (internal, not the C ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wifirx.h"

#ifndef WR_WAVES_PER_BLOCK
#define WR_WAVES_PER_BLOCK 1       // demod kernels: one wave per workgroup (nothing is shared; 0.7 % faster than four)
#endif
#ifndef WR_DEMOD_WAVES_PER_SIMD
#define WR_DEMOD_WAVES_PER_SIMD 4       // register budget of the demod kernels: 512/4 = 128 VGPRs
#endif
#ifndef WR_DEMOD_WAVES_PER_SIMD_STA
#define WR_DEMOD_WAVES_PER_SIMD_STA 4   // (with its window sum free of per-term tests STA fits 128 registers too: 20.3 vs 24.2 ms at 3)
#endif
#define WR_STREAM_SPAN     16       // tiles of 64 samples one wave scans in stream-mode detection
#define WR_DECODE_MAX_WAVES 4096    // waves of the decode kernel (grid-stride; each owns a scratch slice)
#ifndef WR_DQ_SPEC_WAVES
#define WR_DQ_SPEC_WAVES 3          // decode_q_kernel with the speculative trace-back (rates up to 16-QAM): waves per SIMD.  3 = 168 registers,
                                    // 62 spilled but 9 accesses inside the group loop of 6 300 instructions: 12.6 -> 11.45 ms per million frames
                                    // (2: 231 registers, no spill; 4: 128 registers, 202 spilled, 14.7 ms) -- profiles/r05_ab_decode_waves.txt
#endif
#define WR_DECODE_SMALL_MAX 16384      // batches up to this many frames take the wave-per-frame decode kernel
#define WR_DECODE_FRAMES_PER_WAVE 128   // two frames per lane: packed 16-bit path metrics
#define WR_DECODE_Q_FRAMES_PER_WAVE 256 // four frames per lane: byte path metrics (decode_q_kernel)
#define WR_DECODE_Q_MIN_FRAMES 600000  // batches from this many decodable frames on fill the GPU with 256-frame waves
#define WR_DECODE_SCRATCH_BUDGET (24ull << 30)   // bytes of survivor scratch a decode call may hold

namespace wr {

struct DemodParams {
    double   bandwidth;
    double   frequency;
    float    threshold;
    int32_t  min_plateau;
    uint32_t max_sym;
    uint32_t llr_bits;
    int32_t  chan_est;      // WIFIRX_EQ_LS / WIFIRX_EQ_LMS
    int32_t  llr_csi;       // 1: LLRs weighted by |H|^2 (WIFIRX_P_LLR_CSI)
};

// device output rows of the demod kernels (null = not produced); row r of every array belongs to frame r
struct DemodOut {
    wifirx_frame* frames;
    uint8_t*      idx;
    float*        llr;
    float2*       carrier;
    float2*       csi;
    float4*       sym_stats;    // per frame: sum |y|, sum |y|^2, sum |y|^4 over its equalised data symbols, 0
    uint32_t*     hbits;        // the decisions as bit planes, max_sym * 12 words per frame (wifirx_out.hbits)
};

// one detected frame of a continuous stream (stream mode)
struct StreamTrig {
    int64_t pos;        // trigger index in the stream buffer
    int64_t usable;     // L: copied samples that belong to this trigger
    float   cfo;        // coarse CFO
    int32_t pad;
};

}  // namespace wr

// which translation unit's kernels a launch takes (wave-uniform facts the host knows): XK = any output set but decisions + LLRs
static inline bool wr_demod_wants_x(const wr::DemodParams* prm, const wr::DemodOut* out)
{
#if defined(WR_X_LOOPS) && !WR_X_LOOPS
    (void)prm; (void)out;
    return false;
#else
    return out->carrier != nullptr || out->sym_stats != nullptr || out->idx == nullptr || out->llr == nullptr || prm->llr_csi != 0;
#endif
}

extern "C" {
hipError_t wr_launch_demod_batch_x(hipStream_t st, const float2* iq, uint32_t slot_len, uint32_t n_slots,
                                   const wr::DemodParams* prm, const wr::DemodOut* out, const uint64_t* slot_off);
hipError_t wr_launch_demod_stream_x(hipStream_t st, const float2* x, int64_t n_samp, const wr::StreamTrig* trig,
                                    uint32_t n_trig, const wr::DemodParams* prm, const float2* A, const wr::DemodOut* out);
hipError_t wr_launch_demod_batch(hipStream_t st, const float2* iq, uint32_t slot_len, uint32_t n_slots,
                                 const wr::DemodParams* prm, const wr::DemodOut* out, const uint64_t* slot_off);
hipError_t wr_launch_synth(hipStream_t st, const float2* templates, uint32_t n_templates, uint32_t frame_len,
                           float2* slots, uint32_t slot_len, uint32_t n_slots, uint32_t lead, float gain,
                           float noise, float cfo_max, uint64_t seed, float* cfo_out);
hipError_t wr_launch_decode_maxsteps(hipStream_t st, uint32_t n_slots, uint32_t max_sym,
                                     const wifirx_frame* frames, uint32_t psdu_stride, uint32_t* out);
hipError_t wr_launch_decode_pack(hipStream_t st, uint32_t n_slots, uint32_t max_sym, const wifirx_frame* frames,
                                 const uint8_t* idx, uint32_t psdu_stride, uint32_t* hbits, uint32_t n_steps_cap);
hipError_t wr_launch_decode(hipStream_t st, uint32_t n_slots, uint32_t max_sym, wifirx_frame* frames,
                            const uint32_t* hbits, uint8_t* psdu, uint32_t psdu_stride, uint8_t* scratch,
                            size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves, uint32_t frames_per_wave,
                            const uint32_t* perm, uint32_t n_virtual);
hipError_t wr_launch_decode_q(hipStream_t st, uint32_t n_slots, uint32_t max_sym, wifirx_frame* frames,
                              const uint32_t* hbits, uint8_t* psdu, uint32_t psdu_stride, uint8_t* scratch,
                              size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves, uint32_t frames_per_wave,
                              const uint32_t* perm, uint32_t n_virtual, int has_64qam, int overlap);
hipError_t wr_launch_decode_perm(hipStream_t st, uint32_t n_slots, uint32_t max_sym, const wifirx_frame* frames,
                                 uint32_t psdu_stride, const uint32_t* starts8, uint32_t* cursor8, uint32_t* perm);
hipError_t wr_launch_decode_small(hipStream_t st, uint32_t n_slots, uint32_t max_sym, wifirx_frame* frames,
                                  const uint32_t* hbits, uint8_t* psdu, uint32_t psdu_stride, uint8_t* scratch,
                                  size_t scratch_stride, uint32_t n_steps_cap, uint32_t n_waves);
hipError_t wr_launch_stream_detect(hipStream_t st, const float2* x, int64_t n_samp, int64_t tile0,
                                   int64_t n_tiles, float thr, uint64_t* masks, float2* A);
hipError_t wr_launch_demod_stream(hipStream_t st, const float2* x, int64_t n_samp, const wr::StreamTrig* trig,
                                  uint32_t n_trig, const wr::DemodParams* prm, const float2* A, const wr::DemodOut* out);
}
