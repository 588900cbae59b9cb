// Micro-probe: achievable v_mfma_f32_32x32x2_f32 / 16x16x4 rate per SIMD as a
// function of (independent accumulators per wave, waves per SIMD, operand source).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SRC>   // SRC 0: registers, 1: LDS ds_read_b32 per operand
__global__ __launch_bounds__(256) void probe32(float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const int lane = threadIdx.x & 63;
    float a = lane * 0.01f, b = 1.0f;
    const float* pa = lds + lane;
    const float* pb = lds + 4096 + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if (SRC == 1) {
                    a = pa[(u * NACC + n) * 64 + (it & 1) * 32];
                    b = pb[(u * NACC + n) * 64 + (it & 1) * 32];
                }
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int SRC>
__global__ __launch_bounds__(256) void probe16(float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 4; ++i) acc[a][i] = 0.f;
    const int lane = threadIdx.x & 63;
    float a = lane * 0.01f, b = 1.0f;
    const float* pa = lds + lane;
    const float* pb = lds + 4096 + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if (SRC == 1) {
                    a = pa[(u * NACC + n) * 64 + (it & 1) * 32];
                    b = pb[(u * NACC + n) * 64 + (it & 1) * 32];
                }
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int nacc, double flop_per_mfma, int blocks_per_cu, float* d) {
    const int iters = 2000;
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mfmas = (double)blocks * 4 * iters * 8 * nacc;
    printf("%-34s waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms,
           mfmas * flop_per_mfma / (ms * 1e-3) / 1e12);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 4; w *= 2) {
        run("32x32x2 reg  1 acc", probe32<1, 0>, 1, 4096, w, d);
        run("32x32x2 reg  2 acc", probe32<2, 0>, 2, 4096, w, d);
        run("32x32x2 reg  4 acc", probe32<4, 0>, 4, 4096, w, d);
        run("32x32x2 lds  1 acc", probe32<1, 1>, 1, 4096, w, d);
        run("32x32x2 lds  2 acc", probe32<2, 1>, 2, 4096, w, d);
        run("32x32x2 lds  4 acc", probe32<4, 1>, 4, 4096, w, d);
        run("16x16x4 reg  1 acc", probe16<1, 0>, 1, 2048, w, d);
        run("16x16x4 reg  2 acc", probe16<2, 0>, 2, 2048, w, d);
        run("16x16x4 reg  4 acc", probe16<4, 0>, 4, 2048, w, d);
        run("16x16x4 lds  2 acc", probe16<2, 1>, 2, 2048, w, d);
        run("16x16x4 lds  4 acc", probe16<4, 1>, 4, 2048, w, d);
    }
    return 0;
}
