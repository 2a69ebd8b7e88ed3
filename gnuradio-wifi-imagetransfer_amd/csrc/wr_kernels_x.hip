// wr_kernels_x.hip -- the demod kernels for every output set but the contract's usual one (XK = true, wr_demod.h): the
// reference's own set -- the equalised points of the `carrier` port (gnu_radio/IRS_AP.py:293,312-313) --, LLRs weighted by the
// channel state, the SNR probe's moments, the bit planes alone (what the stream path asks for), decisions without LLRs.
#include "wr_demod.h"

extern "C" hipError_t wr_launch_demod_batch_x(hipStream_t st, const float2* iq, uint32_t slot_len,
                                              uint32_t n_slots, const wr::DemodParams* prm, const wr::DemodOut* out,
                                              const uint64_t* slot_off)
{
    if (n_slots == 0) return hipSuccess;
    dim3 grid((n_slots + 4 * WR_WAVES_PER_BLOCK - 1) / (4 * WR_WAVES_PER_BLOCK)), block(64 * WR_WAVES_PER_BLOCK);
#define WR_LAUNCH_BATCH(EQ, HB) hipLaunchKernelGGL((wr::demod_batch_kernel<EQ, HB, true>), grid, block, 0, st, iq, slot_len, n_slots, *prm, *out, slot_off)
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

extern "C" hipError_t wr_launch_demod_stream_x(hipStream_t st, const float2* x, int64_t n_samp, const wr::StreamTrig* trig,
                                               uint32_t n_trig, const wr::DemodParams* prm, const float2* A, const wr::DemodOut* out)
{
    if (n_trig == 0) return hipSuccess;
    dim3 grid((n_trig + 4 * WR_WAVES_PER_BLOCK - 1) / (4 * WR_WAVES_PER_BLOCK)), block(64 * WR_WAVES_PER_BLOCK);
#define WR_LAUNCH_STREAM(EQ, HB) hipLaunchKernelGGL((wr::demod_stream_kernel<EQ, HB, true>), grid, block, 0, st, x, (long)n_samp, trig, n_trig, *prm, A, *out)
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
