// Memory floor of wr::demod_batch_kernel on the config-3 geometry (64-QAM 3/4, 11 data symbols, slot 1472): the kernel's
// global loads and stores without arithmetic, with the 64-QAM rows leaving (a) as they do now -- per data bin one byte and
// three 8-byte pieces, lanes 24 bytes apart -- and (b) as whole 16-byte pieces (72 per row and symbol + 12 dwords of decisions).
// Says what whole-line stores for 16- / 64-QAM rows could gain before they are built.
//   hipcc --offload-arch=gfx950 -O3 tools/mem_floor64.hip -o tools/mem_floor64.bin && tools/mem_floor64.bin [n_slots]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int do_preamble, int do_loads, int do_stores>     // do_stores: 0 none, 1 the kernel's present shape, 2 whole 16-byte pieces
__global__ __launch_bounds__(256, 4) void pattern(const float2* __restrict__ x, uint32_t n_slots, int slot_len, int n_sym,
                                                  uint8_t* __restrict__ idx, float* __restrict__ llr, float* sink)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15;
    const uint32_t slot = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4) + row;
    if (slot >= n_slots) return;
    const float2* xs = x + (size_t)slot * slot_len;
    float acc = 0.0f;
    if (do_preamble) {
        for (int m = 0; m < 14; m++) { float2 a = xs[16 * m + r]; acc += a.x + a.y; }
        for (int f = 0; f < 4; f++) {
            const float2* xf = x + (size_t)(slot - row + f) * slot_len;
            if (slot - row + f >= n_slots) break;
            for (int p = 0; p < 6; p++) { float2 a = xf[176 + 64 * p + lane]; acc += a.x + a.y; }
        }
    }
    int carrier[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = r + 16 * j;
        const bool data = (i >= 6 && i <= 58 && i != 11 && i != 25 && i != 32 && i != 39 && i != 53);
        carrier[j] = data ? (i - 6 - (i > 11) - (i > 25) - (i > 32) - (i > 39) - (i > 53)) : -1;
    }
    uint8_t* ip = idx + (size_t)slot * n_sym * 48;
    float* lp = llr + (size_t)slot * n_sym * 288;
    for (int s = 0; s < n_sym + 3; s++) {
        const int off = 352 + (s < 2 ? 64 * s : 128 + 80 * (s - 2) + 16);
        float2 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[j] = make_float2((float)s, (float)j);
            if (do_loads) v[j] = xs[off + r + 16 * j];
        }
        if (s >= 3) {
            const int q = s - 3;
            if (do_stores == 2) {
                // a row's 1152 B = 72 pieces of 16 B: lane r writes pieces r, r + 16, r + 32, r + 48 and (r < 8) r + 64
                float4* dst = reinterpret_cast<float4*>(lp + q * 288) + r;
                const float4 w = make_float4(v[0].x + v[2].y, v[1].y + v[3].x, v[2].x + v[0].y, v[3].y + v[1].x);
                dst[0] = w; dst[16] = w; dst[32] = w; dst[48] = w;
                if (r < 8) dst[64] = w;
                if (r < 12) reinterpret_cast<uint32_t*>(ip + q * 48)[r] = __float_as_uint(v[0].x + v[1].x + v[2].y + v[3].y);
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (carrier[j] < 0 || !do_stores) { acc += v[j].x + v[j].y; continue; }
                const unsigned o = (unsigned)(q * 48 + carrier[j]);
                ip[o] = (uint8_t)((v[j].x > 0.0f) | ((v[j].y > 0.0f) << 1));
                float2* l2 = reinterpret_cast<float2*>(lp + o * 6);
                l2[0] = v[j]; l2[1] = make_float2(v[j].y, v[j].x); l2[2] = make_float2(v[j].x + 1.0f, v[j].y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) acc += v[j].x + v[j].y;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv)
{
    const uint32_t n_slots = argc > 1 ? (uint32_t)atol(argv[1]) : 1000000u;
    const int slot_len = 1472, n_sym = 11;
    float2* x; float* llr; uint8_t* idx; float* o;
    if (hipMalloc(&x, (size_t)n_slots * slot_len * sizeof(float2)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&llr, (size_t)n_slots * n_sym * 288 * sizeof(float));
    (void)hipMalloc(&idx, (size_t)n_slots * n_sym * 48);
    (void)hipMalloc(&o, 4);
    (void)hipMemset(x, 0x3c, (size_t)n_slots * slot_len * sizeof(float2));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[6] = { "loads+stores+preamble", "loads+stores", "loads only", "stores only", "loads+line stores+preamble", "line stores only" };
    for (int c = 0; c < 6; c++) {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            (void)hipEventRecord(e0, 0);
            const dim3 g((n_slots + 15) / 16), b(256);
            switch (c) {
            case 0: hipLaunchKernelGGL((pattern<1, 1, 1>), g, b, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); break;
            case 1: hipLaunchKernelGGL((pattern<0, 1, 1>), g, b, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); break;
            case 2: hipLaunchKernelGGL((pattern<0, 1, 0>), g, b, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); break;
            case 3: hipLaunchKernelGGL((pattern<0, 0, 1>), g, b, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); break;
            case 4: hipLaunchKernelGGL((pattern<1, 1, 2>), g, b, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); break;
            default: hipLaunchKernelGGL((pattern<0, 0, 2>), g, b, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); break;
            }
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        printf("%-28s %8.3f ms\n", names[c], best);
    }
    return 0;
}
