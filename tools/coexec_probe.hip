// Do plain VALU instructions co-execute with v_mfma_f32_32x32x2_f32 on gfx950, or
// do they serialise on the same fp32 lanes?  VX extra independent v_fma per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int VX, int KIND>   // KIND 0: f32 32x32x2, 1: bf16 32x32x16
__global__ __launch_bounds__(256) void work(float* out, int iters) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int lane = threadIdx.x & 63;
    float a = lane * 0.001f, b = 1.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3f80 + lane); bb[i] = (short)0x3f80; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
#pragma unroll
            for (int x = 0; x < VX; ++x) v[x & 7] = __builtin_fmaf(v[x & 7], 1.0001f, 0.5f);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int vx, float* d) {
    const int blocks = 1024, iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    double mf = (double)blocks * 4 * iters * 8;
    printf("%-28s extra VALU/MFMA=%2d  %8.1f us   %7.1f ns per MFMA per SIMD-slot\n", name, vx, best * 1e3,
           best * 1e6 / (mf / 1024.0));
}

int main() {
    float* d; hipMalloc(&d, (size_t)1024 * 256 * 4);
    run("f32 32x32x2", work<0, 0>, 0, d);
    run("f32 32x32x2", work<2, 0>, 2, d);
    run("f32 32x32x2", work<4, 0>, 4, d);
    run("f32 32x32x2", work<8, 0>, 8, d);
    run("f32 32x32x2", work<12, 0>, 12, d);
    run("f32 32x32x2", work<16, 0>, 16, d);
    run("bf16 32x32x16", work<0, 1>, 0, d);
    run("bf16 32x32x16", work<2, 1>, 2, d);
    run("bf16 32x32x16", work<4, 1>, 4, d);
    run("bf16 32x32x16", work<8, 1>, 8, d);
    return 0;
}
