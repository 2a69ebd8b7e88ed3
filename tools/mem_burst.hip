// Experiment (round 3): does the memory floor of demod_batch_kernel's access pattern (tools/mem_floor.hip) move when a wave
// touches LARGER contiguous pieces per visit?  The kernel reads 512 B and writes 384 + 48 B per frame and symbol; a float4
// copy of the same bytes runs at 6.29 TB/s (MI355X_MICROARCH.md), this pattern at ~4.6.  Variants: B symbols loaded back to
// back, then B symbols stored back to back (B = 1, 2, 4, 8), with the kernel's store shape (byte + float2 per bin) or whole
// 16-byte chunks; and a plain float4 copy of the same byte counts on the same box as the yardstick.
//   hipcc --offload-arch=gfx950 -O3 tools/mem_burst.hip -o tools/mem_burst.bin && tools/mem_burst.bin [n_slots]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int B, int LINES, int WPS>
__global__ __launch_bounds__(256, WPS) void burst(const float2* __restrict__ x, uint32_t n_slots, int slot_len, int n_sym,
                                                  uint8_t* __restrict__ idx, float2* __restrict__ llr, float* sink)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15;
    const uint32_t slot0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    const uint32_t slot = slot0 + row;
    if (slot0 + 3 >= n_slots) return;
    const float2* xs = x + (size_t)slot * slot_len;
    float acc = 0.0f;
    int carrier[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = r + 16 * j;
        const bool data = (i >= 6 && i <= 58 && i != 11 && i != 25 && i != 32 && i != 39 && i != 53);
        carrier[j] = data ? (i - 6 - (i > 11) - (i > 25) - (i > 32) - (i > 39) - (i > 53)) : -1;
    }
    uint8_t* ip = idx + (size_t)slot * n_sym * 48;
    float2* lp = llr + (size_t)slot * n_sym * 48;
    // the three symbols before the data (two LTS symbols + SIGNAL): loads only
    for (int s = 0; s < 3; s++) {
        const int off = 352 + (s < 2 ? 64 * s : 128 + 16);
#pragma unroll
        for (int j = 0; j < 4; j++) { const float2 a = xs[off + r + 16 * j]; acc += a.x + a.y; }
    }
    for (int q0 = 0; q0 < n_sym; q0 += B) {
        float2 v[B][4];
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int off = 352 + 128 + 80 * (q0 + b + 1) + 16;
#pragma unroll
            for (int j = 0; j < 4; j++) v[b][j] = (q0 + b < n_sym) ? xs[off + r + 16 * j] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int q = q0 + b;
            if (q >= n_sym) break;
            if (LINES) {
                for (int c = lane; c < 96; c += 64) {
                    const int rw = c / 24, k = c % 24;
                    float4* dst = reinterpret_cast<float4*>(llr + ((size_t)(slot0 + rw) * n_sym + q) * 48) + k;
                    *dst = make_float4(v[b][0].x + v[b][2].y, v[b][1].y + v[b][3].x, v[b][2].x + v[b][0].y, v[b][3].y + v[b][1].x);   // every loaded component is used

                }
                if (r < 12) reinterpret_cast<uint32_t*>(ip + q * 48)[r] = __float_as_uint(v[b][0].x + v[b][1].x + v[b][2].y + v[b][3].y);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (carrier[j] < 0) { acc += v[b][j].x; continue; }
                    const unsigned o = (unsigned)(q * 48 + carrier[j]);
                    ip[o] = (uint8_t)((v[b][j].x > 0.0f) | ((v[b][j].y > 0.0f) << 1));
                    lp[o] = v[b][j];
                }
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}


// per-symbol loads and 16-byte-chunk stores as burst<1, 1, 4>, plus: every P symbols one dword per 128-byte line of the NEXT P
// symbols of the row's frame is requested (P * 640 B = 5 P lines over 16 lanes), so that DRAM sees P * 640-byte reads per frame
// and the per-symbol loads find their lines on the die
template <int P>
__global__ __launch_bounds__(256, 4) void touch_ahead(const float2* __restrict__ x, uint32_t n_slots, int slot_len, int n_sym,
                                                      uint8_t* __restrict__ idx, float2* __restrict__ llr, float* sink)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15;
    const uint32_t slot0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    const uint32_t slot = slot0 + row;
    if (slot0 + 3 >= n_slots) return;
    const float2* xs = x + (size_t)slot * slot_len;
    const float* xl = reinterpret_cast<const float*>(xs);
    float acc = 0.0f;
    uint8_t* ip = idx + (size_t)slot * n_sym * 48;
    for (int s = 0; s < 3; s++) {
        const int off = 352 + (s < 2 ? 64 * s : 128 + 16);
#pragma unroll
        for (int j = 0; j < 4; j++) { const float2 a = xs[off + r + 16 * j]; acc += a.x + a.y; }
    }
    float t[(5 * P + 15) / 16];
#pragma unroll
    for (int k = 0; k < (5 * P + 15) / 16; k++) t[k] = 0.f;
    for (int q = 0; q < n_sym; q++) {
        float2 v[4];
        const int off = 352 + 128 + 80 * (q + 1) + 16;
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = xs[off + r + 16 * j];
        if (q % P == 0) {
            // lines of symbols q + P .. q + 2 P - 1: byte offset of the first = 8 * (352 + 128 + 80 * (q + P + 1)), 5 P lines
#pragma unroll
            for (int k = 0; k < (5 * P + 15) / 16; k++) {
                const int line = r + 16 * k;
                const size_t sample = (size_t)(352 + 128 + 80 * (q + P + 1)) + 16 * line;
                if (line < 5 * P && q + P < n_sym) { acc += t[k]; t[k] = xl[2 * sample]; }
            }
        }
        for (int c = lane; c < 96; c += 64) {
            const int rw = c / 24, k = c % 24;
            float4* dst = reinterpret_cast<float4*>(llr + ((size_t)(slot0 + rw) * n_sym + q) * 48) + k;
            *dst = make_float4(v[0].x + v[2].y, v[1].y + v[3].x, v[2].x + v[0].y, v[3].y + v[1].x);
        }
        if (r < 12) reinterpret_cast<uint32_t*>(ip + q * 48)[r] = __float_as_uint(v[0].x + v[1].x + v[2].y + v[3].y);
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// one frame per wave: lane <-> sample, 512 B per load instruction from ONE frame; stores 384 B + 48 B of one frame
template <int B>
__global__ __launch_bounds__(256, 4) void wave_per_frame(const float2* __restrict__ x, uint32_t n_slots, int slot_len, int n_sym,
                                                         uint8_t* __restrict__ idx, float2* __restrict__ llr, float* sink)
{
    const int lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= n_slots) return;
    const float2* xs = x + (size_t)slot * slot_len;
    float acc = 0.0f;
    for (int s = 0; s < 3; s++) { const float2 a = xs[352 + (s < 2 ? 64 * s : 144) + lane]; acc += a.x + a.y; }
    for (int q0 = 0; q0 < n_sym; q0 += B) {
        float2 v[B];
#pragma unroll
        for (int b = 0; b < B; b++) v[b] = (q0 + b < n_sym) ? xs[352 + 128 + 80 * (q0 + b + 1) + 16 + lane] : make_float2(0.f, 0.f);
#pragma unroll
        for (int b = 0; b < B; b++) {
            const int q = q0 + b;
            if (q >= n_sym) break;
            if (lane < 24) reinterpret_cast<float4*>(llr + ((size_t)slot * n_sym + q) * 48)[lane] = make_float4(v[b].x, v[b].y, v[b].x, v[b].y);
            if (lane >= 32 && lane < 44) reinterpret_cast<uint32_t*>(idx + ((size_t)slot * n_sym + q) * 48)[lane - 32] = __float_as_uint(v[b].x);
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n_rd, size_t n_wr)
{
    // n_rd float4 read, n_wr float4 written (n_wr <= n_rd): the read / write mix of the pattern
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rd; i += stride) {
        const float4 v = src[i];
        if (i < n_wr) dst[i] = v; else { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) dst[0] = a;
}

template <typename F> static float best_of(F&& launch, int reps = 5)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < reps; it++) {
        (void)hipEventRecord(e0, 0);
        launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return best;
}

int main(int argc, char** argv)
{
    const uint32_t n_slots = argc > 1 ? (uint32_t)atol(argv[1]) : 1000000u;
    const int slot_len = 4608, n_sym = 50;
    float2 *x, *llr; uint8_t* idx; float* o;
    if (hipMalloc(&x, (size_t)n_slots * slot_len * sizeof(float2)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&llr, (size_t)n_slots * n_sym * 48 * sizeof(float2));
    (void)hipMalloc(&idx, (size_t)n_slots * n_sym * 48);
    (void)hipMalloc(&o, 4);
    (void)hipMemset(x, 0x3c, (size_t)n_slots * slot_len * sizeof(float2));
    const double rd = (double)n_slots * 53.0 * 64 * 8, wr = (double)n_slots * n_sym * 48 * 9;
    const dim3 g4((n_slots + 15) / 16), g1((n_slots + 3) / 4), blk(256);
#define RUN(name, kern, grid)                                                                                       \
    {                                                                                                               \
        const float ms = best_of([&] { hipLaunchKernelGGL(kern, grid, blk, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); }); \
        printf("%-58s %8.3f ms  -> %.2f TB/s\n", name, ms, (rd + wr) / ms / 1e9);                                  \
    }
    RUN("4 frames/wave, kernel's store shape, 1 symbol per visit", (burst<1, 0, 4>), g4);
    RUN("4 frames/wave, kernel's store shape, 2 symbols per visit", (burst<2, 0, 4>), g4);
    RUN("4 frames/wave, kernel's store shape, 4 symbols per visit", (burst<4, 0, 4>), g4);
    RUN("4 frames/wave, 16-byte chunks, 1 symbol per visit", (burst<1, 1, 4>), g4);
    RUN("4 frames/wave, 16-byte chunks, 2 symbols per visit", (burst<2, 1, 4>), g4);
    RUN("4 frames/wave, 16-byte chunks, 4 symbols per visit", (burst<4, 1, 4>), g4);
    RUN("4 frames/wave, 16-byte chunks, 8 symbols per visit", (burst<8, 1, 4>), g4);
    RUN("4 frames/wave, 16-byte chunks, 1 symbol, 8 waves/SIMD", (burst<1, 1, 8>), g4);
    RUN("4 frames/wave, 16-byte chunks, 1 symbol, 2 waves/SIMD", (burst<1, 1, 2>), g4);
    RUN("4 frames/wave, 16-byte chunks, 4 symbols, 2 waves/SIMD", (burst<4, 1, 2>), g4);
    RUN("as above + lines of the next 2 symbols touched every 2", (touch_ahead<2>), g4);
    RUN("as above + lines of the next 4 symbols touched every 4", (touch_ahead<4>), g4);
    RUN("as above + lines of the next 8 symbols touched every 8", (touch_ahead<8>), g4);
    RUN("1 frame/wave, 1 symbol per visit", (wave_per_frame<1>), g1);
    RUN("1 frame/wave, 4 symbols per visit", (wave_per_frame<4>), g1);
    RUN("1 frame/wave, 10 symbols per visit", (wave_per_frame<10>), g1);
    {
        const size_t n_rd = (size_t)(rd / 16), n_wr = (size_t)n_slots * n_sym * 48 * 8 / 16;   // the LLR buffer (8/9 of the written bytes)
        const float ms = best_of([&] { hipLaunchKernelGGL(copy4, dim3(256 * 32), blk, 0, 0, reinterpret_cast<const float4*>(x),
                                                          reinterpret_cast<float4*>(llr), n_rd, n_wr); });
        printf("%-58s %8.3f ms  -> %.2f TB/s\n", "float4 stream: same bytes read, 8/9 of the bytes written", ms, (rd + 16.0 * n_wr) / ms / 1e9);
        const float ms2 = best_of([&] { hipLaunchKernelGGL(copy4, dim3(256 * 32), blk, 0, 0, reinterpret_cast<const float4*>(x),
                                                           reinterpret_cast<float4*>(llr), n_wr, n_wr); });
        printf("%-58s %8.3f ms  -> %.2f TB/s\n", "float4 copy of the written bytes (1:1 mix)", ms2, 2 * 16.0 * n_wr / ms2 / 1e9);
        const float ms3 = best_of([&] { hipLaunchKernelGGL(copy4, dim3(256 * 32), blk, 0, 0, reinterpret_cast<const float4*>(x),
                                                           reinterpret_cast<float4*>(llr), n_rd, (size_t)0); });
        printf("%-58s %8.3f ms  -> %.2f TB/s\n", "float4 read of the read bytes", ms3, rd / ms3 / 1e9);
    }
    return 0;
}
