// wr_kernels.hip -- HIP kernels of the wifirx receive chain for gfx950 (MI355X, CDNA4).
//
// One wavefront (64 lanes) owns four frames.  Preamble phase: the four slots are scanned for the short preamble in lock
// step (row f = slot f, lane <-> sample of a 16-sample block), then pairs of frames go through the LTS search (int8 matrix
// cores, wr_quad.h); symbol phase: row f = lanes 16f..16f+15 owns frame f, lane r of the row holds bins r + 16 j.  The per-frame state of the
// reference's frame_equalizer (d_er, previous pilots, channel estimate H) lives in registers and wave-private LDS for
// the whole frame, so the symbols of a frame are walked in order by the same wave and a million frames run side by
// side.  The whole chain is fused: the samples of a slot are read from HBM once, the only HBM writes are the
// decisions / LLRs / frame record (and, on request, the decisions as bit planes for decode_mac).
//
// Replaces, per the reference's flowgraph (gnu_radio/IRS_AP.py:268-285,294-311):
//   a1  delay(16), conjugate_cc, multiply_vcc, moving_average_cc(48), complex_to_mag_squared,
//       moving_average_ff(64), complex_to_mag, divide_ff            -> detect phase
//   a2  ieee802_11.sync_short(0.56, 2)                              -> detect phase + derotation
//   a3  delay(320) + ieee802_11.sync_long(320)                      -> lts phase + derotation
//   a4  stream_to_vector(64) + fft_vcc(64, forward, rect, shift)    -> fft64
//   a5  ieee802_11.frame_equalizer(LS, freq, bw)                    -> symbol loop
//   a6  constellation decision makers, a7 LLR (north-star addition)
#include "wr_demod.h"

namespace wr {

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

}  // namespace wr

extern "C" hipError_t wr_launch_demod_batch(hipStream_t st, const float2* iq, uint32_t slot_len,
                                            uint32_t n_slots, const wr::DemodParams* prm, const wr::DemodOut* out,
                                            const uint64_t* slot_off)
{
    if (n_slots == 0) return hipSuccess;
    if (wr_demod_wants_x(prm, out)) return wr_launch_demod_batch_x(st, iq, slot_len, n_slots, prm, out, slot_off);
    dim3 grid((n_slots + 4 * WR_WAVES_PER_BLOCK - 1) / (4 * WR_WAVES_PER_BLOCK)), block(64 * WR_WAVES_PER_BLOCK);
#define WR_LAUNCH_BATCH(EQ, HB) hipLaunchKernelGGL((wr::demod_batch_kernel<EQ, HB, false>), grid, block, 0, st, iq, slot_len, n_slots, *prm, *out, slot_off)
#define WR_LAUNCH_BATCH_EQ(EQ) { if (out->hbits) WR_LAUNCH_BATCH(EQ, true); else WR_LAUNCH_BATCH(EQ, false); }
    switch (prm->chan_est) {
    case WIFIRX_EQ_LMS:  WR_LAUNCH_BATCH_EQ(WIFIRX_EQ_LMS) break;
    case WIFIRX_EQ_COMB: WR_LAUNCH_BATCH_EQ(WIFIRX_EQ_COMB) break;
    case WIFIRX_EQ_STA:  WR_LAUNCH_BATCH_EQ(WIFIRX_EQ_STA) break;
    default:             WR_LAUNCH_BATCH_EQ(WIFIRX_EQ_LS) break;
    }
#undef WR_LAUNCH_BATCH_EQ
#undef WR_LAUNCH_BATCH
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
                                             uint32_t n_trig, const wr::DemodParams* prm, const float2* A, const wr::DemodOut* out)
{
    if (n_trig == 0) return hipSuccess;
    if (wr_demod_wants_x(prm, out)) return wr_launch_demod_stream_x(st, x, n_samp, trig, n_trig, prm, A, out);
    dim3 grid((n_trig + 4 * WR_WAVES_PER_BLOCK - 1) / (4 * WR_WAVES_PER_BLOCK)), block(64 * WR_WAVES_PER_BLOCK);
#define WR_LAUNCH_STREAM(EQ, HB) hipLaunchKernelGGL((wr::demod_stream_kernel<EQ, HB, false>), grid, block, 0, st, x, (long)n_samp, trig, n_trig, *prm, A, *out)
#define WR_LAUNCH_STREAM_EQ(EQ) { if (out->hbits) WR_LAUNCH_STREAM(EQ, true); else WR_LAUNCH_STREAM(EQ, false); }
    switch (prm->chan_est) {
    case WIFIRX_EQ_LMS:  WR_LAUNCH_STREAM_EQ(WIFIRX_EQ_LMS) break;
    case WIFIRX_EQ_COMB: WR_LAUNCH_STREAM_EQ(WIFIRX_EQ_COMB) break;
    case WIFIRX_EQ_STA:  WR_LAUNCH_STREAM_EQ(WIFIRX_EQ_STA) break;
    default:             WR_LAUNCH_STREAM_EQ(WIFIRX_EQ_LS) break;
    }
#undef WR_LAUNCH_STREAM_EQ
#undef WR_LAUNCH_STREAM
    return hipGetLastError();
}
