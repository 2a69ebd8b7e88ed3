// wr_synth.hip -- synthetic channel on the device: the loop-back test bench of the reference
// (gnu_radio/IRS_tranceiver.py:282-294: signal scaled by sqrt(10^(snr/10)) against unit-variance
// complex Gaussian noise, frequency offset, taps=[1]) for batches too large to build on the host.
// Counter-based RNG (Philox4x32-10): slot contents depend only on (seed, slot, sample index).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wr_kernels.h"

namespace wr {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)x + 0.5f) * 2.3283064365386963e-10f; }

__device__ __forceinline__ float slot_cfo(uint32_t slot, uint64_t seed, float cfo_max)
{
    uint4 r = philox4x32_10(make_uint4(slot, 0u, 0u, 0xC0FFEEu), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5bd1e995u));
    return (2.0f * u01(r.x) - 1.0f) * cfo_max;
}

__global__ __launch_bounds__(256)
void synth_kernel(const float2* __restrict__ templates, uint32_t n_templates, uint32_t frame_len,
                  float2* __restrict__ slots, uint32_t slot_len, uint32_t n_slots, uint32_t lead,
                  float gain, float noise, float cfo_max, uint64_t seed, float* __restrict__ cfo_out)
{
    const uint64_t half = slot_len / 2;
    const uint64_t total = (uint64_t)n_slots * half;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t slot = (uint32_t)(g / half);
        uint32_t j = (uint32_t)(g % half);
        float cfo = slot_cfo(slot, seed, cfo_max);
        if (j == 0 && cfo_out) cfo_out[slot] = cfo;
        uint4 r = philox4x32_10(make_uint4(j, slot, 0u, 0u), key);
        float r1 = sqrtf(-2.0f * logf(u01(r.x))), r2 = sqrtf(-2.0f * logf(u01(r.z)));
        float s1, c1, s2, c2;
        sincosf(6.283185307179586f * u01(r.y), &s1, &c1);
        sincosf(6.283185307179586f * u01(r.w), &s2, &c2);
        const float h = 0.70710678118654752f * noise;       // noise = 1 (unit-variance AWGN) or 0 (noiseless: tests)
        float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (noise != 0.0f)                                   // not 0 * r: a Box-Muller radius of inf (u01 = 0) would give NaN
            o = make_float4(h * r1 * c1, h * r1 * s1, h * r2 * c2, h * r2 * s2);
        const float2* tp = templates + (size_t)(slot % n_templates) * frame_len;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            int64_t m = (int64_t)(2 * j + e) - (int64_t)lead;
            if (m >= 0 && m < (int64_t)frame_len) {
                float2 v = tp[m];
                float sn, cs;
                sincosf(cfo * (float)m, &sn, &cs);
                float re = gain * (v.x * cs - v.y * sn), im = gain * (v.x * sn + v.y * cs);
                if (e == 0) { o.x += re; o.y += im; } else { o.z += re; o.w += im; }
            }
        }
        reinterpret_cast<float4*>(slots + (size_t)slot * slot_len)[j] = o;
    }
}

}  // namespace wr

extern "C" hipError_t wr_launch_synth(hipStream_t st, const float2* templates, uint32_t n_templates,
                                      uint32_t frame_len, float2* slots, uint32_t slot_len, uint32_t n_slots,
                                      uint32_t lead, float gain, float noise, float cfo_max, uint64_t seed, float* cfo_out)
{
    uint64_t total = (uint64_t)n_slots * (slot_len / 2);
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(wr::synth_kernel, dim3((unsigned)blocks), dim3(256), 0, st, templates, n_templates,
                       frame_len, slots, slot_len, n_slots, lead, gain, noise, cfo_max, seed, cfo_out);
    return hipGetLastError();
}
