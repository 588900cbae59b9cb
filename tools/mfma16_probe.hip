// How long does the legacy v_mfma_f32_16x16x16_f16 take on gfx950 next to v_mfma_f32_16x16x32_f16 (half the K)?  And do
// both give the same bits when the x32's second half of K is zeros?  One wave per SIMD (256 threads per block), a chain of
// independent accumulators.   hipcc --offload-arch=gfx950 -O3 tools/mfma16_probe.hip -o tools/mfma16_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int K16>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    f32x4 acc[4];
    for (int f = 0; f < 4; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a8, b8[4];
    f16x4 a4, b4[4];
    for (int j = 0; j < 8; ++j) {
        a8[j] = (_Float16)(0.001f * (threadIdx.x & 63) + j);
        for (int f = 0; f < 4; ++f) b8[f][j] = (_Float16)(1.f + 0.01f * j + f);
    }
    for (int j = 0; j < 4; ++j) {
        a4[j] = a8[j];
        for (int f = 0; f < 4; ++f) b4[f][j] = b8[f][j];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (K16) acc[f] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4[f], acc[f], 0, 0, 0);
                else acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8[f], acc[f], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int f = 0; f < 4; ++f) s += acc[f][0] + acc[f][1] + acc[f][2] + acc[f][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// bit check: C = 16 layout of the split kernels -- lane group g of the x32 holds (tap g >> 1, channels 8 (g & 1) ..), the
// second tap zero; the x16's lane group g holds channels 4 g .. 4 g + 3 of the one tap
__global__ __launch_bounds__(64) void bits(const _Float16* __restrict__ w, const _Float16* __restrict__ x, float* o32, float* o16) {
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    f16x8 a8, b8;
    f16x4 a4, b4;
    for (int j = 0; j < 8; ++j) {
        const int ci = 8 * (g & 1) + j;
        a8[j] = (g >> 1) ? (_Float16)0.f : w[n * 16 + ci];      // row n of the weights, channel ci
        b8[j] = (g >> 1) ? (_Float16)3.5f : x[n * 16 + ci];     // (the zero tap's activations are NOT zero: the weights are)
    }
    for (int j = 0; j < 4; ++j) {
        a4[j] = w[n * 16 + 4 * g + j];
        b4[j] = x[n * 16 + 4 * g + j];
    }
    f32x4 c = {0.25f, -1.5f, 3.f, 0.f};
    const f32x4 r32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
    const f32x4 r16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) {
        o32[lane * 4 + i] = r32[i];
        o16[lane * 4 + i] = r16[i];
    }
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int k16 = 0; k16 < 2; ++k16) {
        for (int rep = 0; rep < 2; ++rep) {
            if (k16) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, iters, cyc);
            else hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, iters, cyc);
            hipDeviceSynchronize();
        }
        unsigned long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%s: %.2f cycles per MFMA (one wave per SIMD, 4 independent accumulators)\n", k16 ? "v_mfma_f32_16x16x16_f16" : "v_mfma_f32_16x16x32_f16",
               (double)c / (iters * 16.0));
    }
    // bit check on random data
    _Float16 hw[256], hx[256];
    srand(1);
    int diff = 0, total = 0;
    _Float16 *dw, *dx;
    float *d32, *d16, h32[256], h16[256];
    hipMalloc(&dw, 512); hipMalloc(&dx, 512); hipMalloc(&d32, 1024); hipMalloc(&d16, 1024);
    for (int t = 0; t < 200; ++t) {
        for (int i = 0; i < 256; ++i) {
            hw[i] = (_Float16)(((rand() % 20001) - 10000) * (t % 3 == 0 ? 1.6f : 0.0007f));
            hx[i] = (_Float16)(((rand() % 20001) - 10000) * (t % 2 == 0 ? 0.0003f : 0.9f));
        }
        hipMemcpy(dw, hw, 512, hipMemcpyHostToDevice);
        hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(bits, dim3(1), dim3(64), 0, 0, dw, dx, d32, d16);
        hipMemcpy(h32, d32, 1024, hipMemcpyDeviceToHost);
        hipMemcpy(h16, d16, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 256; ++i) {
            total++;
            if (memcmp(&h32[i], &h16[i], 4) != 0) diff++;
        }
    }
    printf("x32 with a zero second half against x16: %d of %d results differ\n", diff, total);
    return 0;
}
