// Does the ORDER of v_mfma_f32_16x16x32_f16 over a wave's accumulators matter?  (tools/f16_probe.hip: one wave, one
// accumulator 17 cycles per MFMA, four accumulators round-robin 23.5.)  Chip-wide throughput, 2 waves per SIMD (the
// occupancy of the split-f16 kernels), 8 accumulators per wave as in convh (hi / lo x 4 fragments), 12 MFMAs per step:
//   order 0: hi0 hi1 hi2 hi3  lo0 lo1 lo2 lo3  lo0 lo1 lo2 lo3      (the kernels' order)
//   order 1: hi0 lo0 lo0  hi1 lo1 lo1  hi2 lo2 lo2  hi3 lo3 lo3      (fragment-major: the two lo updates back to back)
//   order 2: two steps at a time: hi0 hi0 lo0 lo0 lo0 lo0  hi1 hi1 ...                     (runs of 2 and 4)
//   hipcc --offload-arch=gfx950 -O3 tools/f16_order_probe.hip -o tools/f16_order_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0)

template <int ORDER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float* out, int iters, float seed) {
    f32x4 hi[4], lo[4];
    for (int f = 0; f < 4; ++f) hi[f] = lo[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a1, a2, b1[4], b2[4];
    for (int j = 0; j < 8; ++j) {
        a1[j] = (_Float16)(seed + 0.001f * (threadIdx.x & 63) + j);
        a2[j] = (_Float16)(seed * 0.5f + j);
        for (int f = 0; f < 4; ++f) { b1[f][j] = (_Float16)(1.f + 0.01f * j + f); b2[f][j] = (_Float16)(0.5f + 0.02f * j + f); }
    }
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int f = 0; f < 4; ++f) MFMA(hi[f], a1, b1[f]);
#pragma unroll
                for (int f = 0; f < 4; ++f) MFMA(lo[f], a1, b2[f]);
#pragma unroll
                for (int f = 0; f < 4; ++f) MFMA(lo[f], a2, b1[f]);
            }
        } else if (ORDER == 1) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int f = 0; f < 4; ++f) { MFMA(hi[f], a1, b1[f]); MFMA(lo[f], a1, b2[f]); MFMA(lo[f], a2, b1[f]); }
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                MFMA(hi[f], a1, b1[f]); MFMA(hi[f], a2, b2[f]);
                MFMA(lo[f], a1, b2[f]); MFMA(lo[f], a2, b1[f]); MFMA(lo[f], a2, b2[f]); MFMA(lo[f], a1, b1[f]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int f = 0; f < 4; ++f) s += hi[f][0] + lo[f][1];
    if (s == 12345.f) out[0] = s;
}

template <int ORDER>
static void run(float* d, int blocks) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<ORDER>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mf = (double)iters * 24 * blocks * 4;
        printf("order %d blocks %4d: %8.1f us  %.0f TFLOP/s  (%.2f ns per MFMA per SIMD)\n", ORDER, blocks, ms * 1e3,
               mf * 16384 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * 24 * (blocks / 512.0) * 2));
    }
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    for (int blocks : {512, 1024}) { run<0>(d, blocks); run<1>(d, blocks); run<2>(d, blocks); }
    return 0;
}
