// What keeps a 938-block x 4-wave fp32-MFMA kernel from the MFMA roof?
// Variants: UNR (MFMAs per unrolled batch), BAR (barrier per stage), LDSR (operands from LDS)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int UNR, int BAR, int LDSR, int NACC>
__global__ __launch_bounds__(256) void work(float* out, int batches, int stages) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    const int lane = threadIdx.x & 63;
    const float* pa = lds + lane;
    const float* pb = lds + 4096 + lane;
    float a = lane * 0.001f, b = 1.f;
    for (int s = 0; s < stages; ++s) {
        for (int it = 0; it < batches; ++it) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (LDSR) { a = pa[((it & 1) * UNR + u) * 64]; }
#pragma unroll
                for (int n = 0; n < NACC; ++n) {
                    if (LDSR) b = pb[(((it & 1) * UNR + u) * NACC + n) * 32];
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
                }
            }
        }
        if (BAR) __syncthreads();
    }
    float sacc = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) sacc += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = sacc;
}

template <typename K>
void run(const char* name, K kern, int blocks, int lds, int mfma_per_stage, int unr_total, int stages, float* d) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, mfma_per_stage / unr_total, stages);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    double mf = (double)blocks * 4 * mfma_per_stage * stages;
    printf("%-40s blocks=%5d lds=%6d  %7.1f us  %6.1f TF\n", name, blocks, lds, best * 1e3, mf * 4096 / (best * 1e-3) / 1e12);
}

int main() {
    float* d; hipMalloc(&d, (size_t)8192 * 256 * 4);
    const int B = 938;
    run("unr1  bar lds  1acc", work<1, 1, 1, 1>, B, 32768, 44, 1, 4, d);
    run("unr11 bar lds  1acc", work<11, 1, 1, 1>, B, 32768, 44, 11, 4, d);
    run("unr11 nobar lds 1acc", work<11, 0, 1, 1>, B, 32768, 44, 11, 4, d);
    run("unr11 bar reg  1acc", work<11, 1, 0, 1>, B, 32768, 44, 11, 4, d);
    run("unr11 nobar reg 1acc", work<11, 0, 0, 1>, B, 32768, 44, 11, 4, d);
    run("unr11 bar lds  2acc (tile x2)", work<11, 1, 1, 2>, B / 2, 32768, 88, 22, 4, d);
    run("unr11 bar lds  4acc (tile x4)", work<11, 1, 1, 4>, B / 4, 32768, 176, 44, 4, d);
    run("unr11 bar lds 1acc lds16K", work<11, 1, 1, 1>, B, 16384, 44, 11, 4, d);
    run("unr11 bar lds 1acc lds8K", work<11, 1, 1, 1>, B, 8192, 44, 11, 4, d);
    run("unr11 nobar reg 1acc 1024 blocks", work<11, 0, 0, 1>, 1024, 8192, 44, 11, 4, d);
    run("unr11 nobar reg 1acc 2048 blocks", work<11, 0, 0, 1>, 2048, 8192, 44, 11, 4, d);
    run("unr11 nobar reg 1acc 256 blocks x16 stg", work<11, 0, 0, 1>, 256, 8192, 44, 11, 16, d);
    run("unr11 nobar reg 1acc 512 blocks x8 stg", work<11, 0, 0, 1>, 512, 8192, 44, 11, 8, d);
    run("unr11 nobar reg 1acc 1024 blocks x16 stg", work<11, 0, 0, 1>, 1024, 8192, 44, 11, 16, d);
    run("unr11 bar lds 1acc 1024 blocks x16 stg", work<11, 1, 1, 1>, 1024, 32768, 44, 11, 16, d);
    return 0;
}
