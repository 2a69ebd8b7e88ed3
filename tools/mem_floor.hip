// Memory floor of wr::demod_batch_kernel: the same global loads and stores, in the same order, by waves organised
// the same way (four slots per wave, lane r of a row holds bins r + 16 j), with no arithmetic in between.
// The time of this kernel is what the memory system alone needs for config 2's access pattern; the distance of the
// real kernel from it is what arithmetic and imperfect overlap cost.  
//   hipcc --offload-arch=gfx950 -O3 tools/mem_floor.hip -o tools/mem_floor.bin && tools/mem_floor.bin [n_slots]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

// do_preamble / do_loads / do_stores are compile-time: as run-time flags (rounds 1-3) every load sat in a branch of its own,
// followed by its own wait -- the four loads of a symbol went out one after the other and the "floor" was 0.5-0.8 ms too high
template <int do_preamble, int do_loads, int do_stores>
__global__ __launch_bounds__(256, 4) void pattern(const float2* __restrict__ x, uint32_t n_slots, int slot_len, int n_sym,
                                                  uint8_t* __restrict__ idx, float2* __restrict__ llr, float* sink)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15;
    const uint32_t slot = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4) + row;
    if (slot >= n_slots) return;
    const float2* xs = x + (size_t)slot * slot_len;
    float acc = 0.0f;
    if (do_preamble) {
        // detection (round 3): row = slot, blocks of 16 samples (lane r <-> sample 16 m + r) up to the trigger (~ sample 210: 14
        // blocks), every sample once; then, per slot by the whole wave, 383 samples from the trigger - 16 on, as the kernel does.
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
    float2* lp = llr + (size_t)slot * n_sym * 48;
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
            if (do_stores == 3) {      // as 2, but the decisions of eight symbols leave together: 384 bytes per row = 24 pieces of 16 B
                const uint32_t slot0 = slot - row;
                for (int c = lane; c < 96; c += 64) {
                    const int rw = c / 24, k = c % 24;
                    float4* dst = reinterpret_cast<float4*>(llr + ((size_t)(slot0 + rw) * n_sym + q) * 48) + k;
                    *dst = make_float4(v[0].x + v[2].y, v[1].y + v[3].x, v[2].x + v[0].y, v[3].y + v[1].x);
                }
                if ((q & 7) == 7 || q == n_sym - 1) {
                    const int n16 = 3 * ((q & 7) + 1);
                    float4* dst = reinterpret_cast<float4*>(ip + (q & ~7) * 48);
                    const float4 w = make_float4(v[0].x, v[1].x, v[2].y, v[3].y);
                    if (r < n16) dst[r] = w;
                    if (16 + r < n16) dst[16 + r] = w;
                }
                continue;
            }
            if (do_stores == 2) {      // whole lines: 96 chunks of 16 B of LLRs per wave and symbol, 12 dwords of decisions per row
                const uint32_t slot0 = slot - row;
                for (int c = lane; c < 96; c += 64) {
                    const int rw = c / 24, k = c % 24;
                    float4* dst = reinterpret_cast<float4*>(llr + ((size_t)(slot0 + rw) * n_sym + q) * 48) + k;
                    *dst = make_float4(v[0].x + v[2].y, v[1].y + v[3].x, v[2].x + v[0].y, v[3].y + v[1].x);   // every loaded component is used
                }
                if (r < 12) reinterpret_cast<uint32_t*>(ip + q * 48)[r] = __float_as_uint(v[0].x + v[1].x + v[2].y + v[3].y);
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (carrier[j] < 0 || !do_stores) { acc += v[j].x + v[j].y; continue; }
                const unsigned o = (unsigned)(q * 48 + carrier[j]);
                ip[o] = (uint8_t)((v[j].x > 0.0f) | ((v[j].y > 0.0f) << 1));
                lp[o] = v[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) acc += v[j].x + v[j].y;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// Experiment (round 3): the same loads and stores with the NEXT symbol's samples requested before the current symbol's
// stores are issued (eight more registers per lane in flight): does the memory system want more bytes in flight per wave?
template <int depth>
__global__ __launch_bounds__(256, 4) void pattern_prefetch(const float2* __restrict__ x, uint32_t n_slots, int slot_len, int n_sym,
                                                           uint8_t* __restrict__ idx, float2* __restrict__ llr, float* sink)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15;
    const uint32_t slot = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4) + row;
    if (slot >= n_slots) return;
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
    float2 nx[2][4];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
        for (int j = 0; j < 4; j++) nx[d][j] = (d < depth) ? xs[352 + 64 * d + r + 16 * j] : make_float2(0.f, 0.f);
    for (int s = 0; s < n_sym + 3; s++) {
        float2 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = nx[0][j];
        const int sn = s + depth;
        const int off = 352 + (sn < 2 ? 64 * sn : 128 + 80 * (sn - 2) + 16);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (depth == 2) nx[0][j] = nx[1][j];
            if (sn < n_sym + 3) nx[depth - 1][j] = xs[off + r + 16 * j];
        }
        if (s >= 3) {
            const int q = s - 3;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (carrier[j] < 0) { acc += v[j].x; continue; }
                const unsigned o = (unsigned)(q * 48 + carrier[j]);
                ip[o] = (uint8_t)((v[j].x > 0.0f) | ((v[j].y > 0.0f) << 1));
                lp[o] = v[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) acc += v[j].x + v[j].y;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// store-pattern variants: mode 0 = the kernel's pattern (per bin: one byte + one float2, rows interleaved),
// 1 = LLR only in that pattern, 2 = idx only in that pattern, 3 = LLR as full lines (lane l of the wave writes
// 16 B + 8 B of the 1536 contiguous... per-row 384 B: lanes 0..23 of a row-major order), 4 = idx as dwords (12 lanes per row),
// 5 = 3 + 4 together
__global__ __launch_bounds__(256, 4) void store_modes(uint32_t n_slots, int n_sym, uint8_t* __restrict__ idx,
                                                     float2* __restrict__ llr, int mode)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15;
    const uint32_t slot0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    const uint32_t slot = slot0 + row;
    if (slot0 + 3 >= n_slots) return;
    int carrier[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = r + 16 * j;
        const bool data = (i >= 6 && i <= 58 && i != 11 && i != 25 && i != 32 && i != 39 && i != 53);
        carrier[j] = data ? (i - 6 - (i > 11) - (i > 25) - (i > 32) - (i > 39) - (i > 53)) : -1;
    }
    uint8_t* ip = idx + (size_t)slot * n_sym * 48;
    float2* lp = llr + (size_t)slot * n_sym * 48;
    for (int q = 0; q < n_sym; q++) {
        const float2 v = make_float2((float)q, (float)lane);
        if (mode <= 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (carrier[j] < 0) continue;
                const unsigned o = (unsigned)(q * 48 + carrier[j]);
                if (mode != 1) ip[o] = (uint8_t)(q + j);
                if (mode != 2) lp[o] = v;
            }
        }
        if (mode == 3 || mode == 5) {
            // 4 rows x 384 B: 96 chunks of 16 B; lane l writes chunks l and (l < 32 ? 64 + l : none)
            for (int c = lane; c < 96; c += 64) {
                const int rw = c / 24, k = c % 24;
                float4* dst = reinterpret_cast<float4*>(llr + ((size_t)(slot0 + rw) * n_sym + q) * 48) + k;
                *dst = make_float4(v.x, v.y, v.x, v.y);
            }
        }
        if (mode == 4 || mode == 5) {
            if (r < 12) reinterpret_cast<uint32_t*>(idx + ((size_t)slot * n_sym + q) * 48)[r] = 0x01020304u * (unsigned)q;
        }
    }
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
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[8] = { "loads+stores+preamble", "loads+stores", "loads only", "stores only",
                             "loads+line stores+preamble", "loads+line stores", "loads+line stores, idx x8+preamble", "loads+line stores, idx x8" };
    const int cfg[8][3] = { { 1, 1, 1 }, { 0, 1, 1 }, { 0, 1, 0 }, { 0, 0, 1 }, { 1, 1, 2 }, { 0, 1, 2 }, { 1, 1, 3 }, { 0, 1, 3 } };
    float res[8];
    for (int c = 0; c < 8; c++) {
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
            case 6: hipLaunchKernelGGL((pattern<1, 1, 3>), g, b, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); break;
            case 7: hipLaunchKernelGGL((pattern<0, 1, 3>), g, b, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); break;
            default: hipLaunchKernelGGL((pattern<0, 1, 2>), g, b, 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o); break;
            }
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        const double rd = (double)n_slots * ((cfg[c][1] ? 53.0 * 64 * 8 : 0) + (cfg[c][0] ? 640.0 * 8 : 0));
        const double wr = cfg[c][2] ? (double)n_slots * n_sym * 48 * 9 : 0;
        printf("%-28s %8.3f ms   %.1f GB moved by lanes -> %.2f TB/s\n", names[c], best, (rd + wr) / 1e9, (rd + wr) / best / 1e9);
        res[c] = best;
    }
    // one JSON line for profiles/*_mem_floor.json (bench.py quotes combined_ms)
    // (combined_idx_x8_ms: an experiment -- the decisions of eight symbols written together; the kernel did not follow, profiles/r03_ab_idx8.txt)
    // (combined_ms: the rows' LLRs and decisions as whole 16-byte pieces -- what the kernel's BPSK / QPSK loops do since the end
    // of round 3; combined_old_shape_ms: one float2 + one byte per bin, as before and as 16- / 64-QAM rows still leave)
    printf("{\"tool\": \"tools/mem_floor.hip\", \"n_slots\": %u, \"combined_ms\": %.3f, \"symbols_only_ms\": %.3f, "
           "\"combined_old_shape_ms\": %.3f, \"symbols_only_old_shape_ms\": %.3f, "
           "\"loads_only_ms\": %.3f, \"stores_only_ms\": %.3f, \"combined_idx_x8_ms\": %.3f, \"what\": \"the global loads and stores of demod_batch_kernel on "
           "config 2 (same addresses, order and wave organisation), no arithmetic\"}\n",
           n_slots, res[4], res[5], res[0], res[1], res[2], res[3], res[6]);
    for (int depth = 1; depth <= 2; depth++) {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            (void)hipEventRecord(e0, 0);
            if (depth == 1) hipLaunchKernelGGL(pattern_prefetch<1>, dim3((n_slots + 15) / 16), dim3(256), 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o);
            else            hipLaunchKernelGGL(pattern_prefetch<2>, dim3((n_slots + 15) / 16), dim3(256), 0, 0, x, n_slots, slot_len, n_sym, idx, llr, o);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        printf("loads %d symbol(s) ahead + stores (no preamble): %8.3f ms\n", depth, best);
    }
    for (int mode = 0; mode < 6; mode++) {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(store_modes, dim3((n_slots + 15) / 16), dim3(256), 0, 0, n_slots, n_sym, idx, llr, mode);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        const double wr = (double)n_slots * n_sym * 48 * ((mode == 0 || mode == 5) ? 9 : (mode == 1 || mode == 3) ? 8 : 1);
        printf("store mode %d: %8.3f ms  %.1f GB -> %.2f TB/s\n", mode, best, wr / 1e9, wr / best / 1e9);
    }
    return 0;
}
