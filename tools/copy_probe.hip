// Yardstick (round 3): what does a plain float4 stream reach on this box, and which property of the kernel moves it?
// MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; tools/mem_burst.hip's grid-stride copy reached 4.6-4.9.
// Variants: grid-stride vs one chunk per workgroup, 1 / 4 / 8 loads in flight per lane, plain / non-temporal stores and loads,
// read : write mixes 1:1 and 32:22 (the demod kernel's).
//   hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o tools/copy_probe.bin && tools/copy_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float v4f __attribute__((ext_vector_type(4)));

// each workgroup owns a contiguous piece of U * 256 float4; lane i touches i, i + 256, ...  (U loads in flight)
template <int U, int NT_ST, int NT_LD>
__global__ __launch_bounds__(256) void copy_block(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n, int wr_num, int wr_den)
{
    const size_t base = (size_t)blockIdx.x * (U * 256) + threadIdx.x;
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = base + (size_t)u * 256;
        if (i < n) v[u] = NT_LD ? __builtin_nontemporal_load(src + i) : src[i];
    }
    // write only wr_num of every wr_den workgroups' pieces (the mix)
    const bool wr = (int)(blockIdx.x % wr_den) < wr_num;
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = base + (size_t)u * 256;
        if (i >= n) continue;
        if (wr) { if (NT_ST) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
        else a += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (a == 12345.678f) dst[0] = v[0];
}

template <int U>
__global__ __launch_bounds__(256) void copy_stride(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n, int wr_num, int wr_den)
{
    const size_t stride = (size_t)gridDim.x * 256;
    float a = 0.f;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * U) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i0 + u * stride < n) v[u] = src[i0 + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = i0 + u * stride;
            if (i >= n) continue;
            if ((int)((i >> 8) % wr_den) < wr_num) dst[i] = v[u]; else a += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (a == 12345.678f) dst[0] = v4f{a, a, a, a};
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
    return best;
}

int main()
{
    const size_t bytes = (size_t)24 << 30;          // 24 GiB read
    const size_t n = bytes / 16;
    v4f *src, *dst;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 0x3c, bytes);
    (void)hipMemset(dst, 0, bytes);
    const int mixes[3][2] = { { 1, 1 }, { 11, 16 }, { 0, 1 } };        // 1:1, 32:22, read only
    for (int m = 0; m < 3; m++) {
        const int num = mixes[m][0], den = mixes[m][1];
        const double moved = (double)bytes * (1.0 + (double)num / den);
        printf("-- read : write = %d : %d  (%.1f GB moved)\n", den, num, moved / 1e9);
#define BLK(U, S, L)                                                                                                        \
        {                                                                                                                   \
            const float ms = best_of([&] { hipLaunchKernelGGL((copy_block<U, S, L>), dim3((unsigned)((n + U * 256 - 1) / (U * 256))), dim3(256), 0, 0, src, dst, n, num, den); }); \
            printf("one piece per workgroup, %d x 16 B in flight per lane%s%s: %8.3f ms -> %.2f TB/s\n", U, S ? ", nt stores" : "", L ? ", nt loads" : "", ms, moved / ms / 1e9); \
        }
        BLK(1, 0, 0) BLK(2, 0, 0) BLK(4, 0, 0) BLK(8, 0, 0) BLK(4, 1, 0) BLK(4, 1, 1) BLK(1, 1, 0)
#define STR(U, G)                                                                                                           \
        {                                                                                                                   \
            const float ms = best_of([&] { hipLaunchKernelGGL((copy_stride<U>), dim3(G), dim3(256), 0, 0, src, dst, n, num, den); }); \
            printf("grid-stride, %5d workgroups, %d in flight: %8.3f ms -> %.2f TB/s\n", G, U, ms, moved / ms / 1e9);       \
        }
        STR(1, 8192) STR(4, 8192) STR(4, 2048) STR(8, 1024)
    }
    return 0;
}
