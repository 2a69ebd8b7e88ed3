// Calibration of rocprofv3 FETCH_SIZE for the access pattern of wr::demod_batch_kernel: every row of 16
// lanes reads 128 contiguous bytes (8 B per lane, global_load_dwordx2) of its own 36 KB slot, four rows =
// four slots per wave instruction, 64-sample strides of 80 samples (the cyclic prefix is skipped).
// Known byte count -> factor for MI355X_MICROARCH.md's "calibrate on a known byte count" rule.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void read8(const float2* __restrict__ x, size_t n_slots, int slot_len, float* out)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t slot = wave * 4 + row;
    if (slot >= n_slots) return;
    const float2* xs = x + slot * slot_len;
    float acc = 0;
    for (int s = 0; s < 53; s++) {
        int off = 352 + 80 * s;
#pragma unroll
        for (int j = 0; j < 4; j++) { float2 v = xs[off + r + 16 * j]; acc += v.x + v.y; }
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main()
{
    const size_t n_slots = 200000; const int slot_len = 4608;
    float2* x; float* o;
    (void)hipMalloc(&x, n_slots * slot_len * sizeof(float2)); (void)hipMalloc(&o, 4);
    (void)hipMemset(x, 0, n_slots * slot_len * sizeof(float2));
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(read8, dim3((n_slots / 4 + 3) / 4), dim3(256), 0, 0, x, n_slots, slot_len, o);
    (void)hipDeviceSynchronize();
    double used = (double)n_slots * 53 * 64 * 8, touched = (double)n_slots * (53 * 80 * 8);
    printf("bytes_loaded_by_lanes %.0f  bytes_of_lines_touched(approx, CP gaps share lines) %.0f\n", used, touched);
    return 0;
}
