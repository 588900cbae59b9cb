// v_mfma_f32_16x16x32_f16 on gfx950: operand layout, f16 subnormal handling, issue rate.
//   hipcc --offload-arch=gfx950 -O3 tools/f16_probe.hip -o tools/f16_probe.bin && tools/f16_probe.bin
// (1) layout: lane l holds A[l%16][(l/16)*8 + j], B[(l/16)*8 + j][l%16], j = 0..7; D[(l/16)*4 + r][l%16], r = 0..3
// (2) subnormal f16 inputs (2^-20) times 1024: flushed -> 0, kept -> 32 * 2^-10
// (3) cycles per MFMA with 1 / 2 / 4 independent accumulators, chip-wide TFLOP/s at 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void layout(const _Float16* A, const _Float16* B, float* D) {
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[(l % 16) * 32 + (l / 16) * 8 + j];
        b[j] = B[((l / 16) * 8 + j) * 16 + (l % 16)];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l / 16) * 4 + r) * 16 + (l % 16)] = c[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate(unsigned long long* out, int iters, float seed) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(seed + 0.001f * (threadIdx.x & 63) + j);
        b[j] = (_Float16)(1.f + 0.01f * j);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        out[w * 2 + 0] = t1 - t0;
        out[w * 2 + 1] = (unsigned long long)(s != 12345.f);
    }
}

template <int NACC>
static void run_rate(unsigned long long* d, unsigned long long* h, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, (size_t)blocks * 4 * 2 * 8, hipMemcpyDeviceToHost);
        double st = 0;
        for (int w = 0; w < blocks * 4; ++w) st += h[w * 2];
        st /= blocks * 4;
        const double mf = (double)iters * 16;
        printf("acc %d blocks %4d rep %d: %8.1f us | ticks per MFMA per wave %.2f | TFLOP/s %.1f\n", NACC, blocks, rep,
               ms * 1e3, st / mf, (double)blocks * 4 * mf * 16384 / (ms * 1e-3) / 1e12);
    }
}

int main() {
    // (1) layout
    _Float16 hA[16 * 32], hB[32 * 16];
    float hD[256], ref[256];
    srand(1);
    for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)((rand() % 15) - 7); hB[i] = (_Float16)((rand() % 15) - 7); }
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            float s = 0;
            for (int k = 0; k < 32; ++k) s += (float)hA[m * 32 + k] * (float)hB[k * 16 + n];
            ref[m * 16 + n] = s;
        }
    _Float16 *dA, *dB;
    float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("layout: %d of 256 outputs differ from A.B with the assumed lane mapping\n", bad);
    // (2) subnormals
    for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)9.5367431640625e-07f; hB[i] = (_Float16)1024.f; }   // 2^-20
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    printf("subnormal inputs: D[0][0] = %.9g (kept: %.9g, flushed: 0)\n", hD[0], 32.0 * 1024.0 * 9.5367431640625e-07);
    // (3) rate
    unsigned long long* d;
    const int maxb = 1024;
    hipMalloc(&d, (size_t)maxb * 4 * 2 * 8);
    unsigned long long* h = (unsigned long long*)malloc((size_t)maxb * 4 * 2 * 8);
    run_rate<1>(d, h, 1);
    run_rate<2>(d, h, 1);
    run_rate<4>(d, h, 1);
    run_rate<4>(d, h, 1024);
    run_rate<2>(d, h, 1024);
    return 0;
}
