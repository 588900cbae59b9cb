// Do the VALU instructions of ONE wave run beside the MFMA stream of ANOTHER wave on the same SIMD (gfx950)?
// A block is 8 waves = 2 per SIMD: waves 0-3 stream matrix instructions (16x16x32 f16, or the same FLOP as 32x32x16 f16),
// waves 4-7 run independent v_fma chains (or ds_read_b128, or nothing).  Times: MFMA waves alone, VALU waves alone, both.
// both ~ max(alone) => the pipes overlap across waves; both ~ sum => they share the issue slot.
//   hipcc --offload-arch=gfx950 -O3 tools/pingpong_probe.hip -o tools/pingpong_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run the side work.  SHAPE 0: 16x16x32, 1: 32x32x16.  SIDE 0: v_fma, 1: ds_read_b128
template <int MODE, int SHAPE, int SIDE, int ROLE = 0, int PRIO = 0>   // ROLE 0: MFMA waves = 0-3; 1: even waves; 2: waves 0, 1, 4, 5; 3: MFMA waves = 4-7 (the YOUNGER half)
// PRIO 1: the side waves at s_setprio 3; 2: the MFMA waves at s_setprio 3
__global__ __launch_bounds__(512) void work(float* out, int iters, int side_per_iter) {
    __shared__ float lds[16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i;
    __syncthreads();
    float s = 0.f;
    const bool mf = ROLE == 0 ? wave < 4 : ROLE == 1 ? (wave & 1) == 0 : ROLE == 2 ? (wave & 2) == 0 : wave >= 4;
    if (PRIO == 1 && !mf) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2 && mf) __builtin_amdgcn_s_setprio(3);
    if (mf) {
        if (MODE & 1) {
            f16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f); b[i] = (_Float16)1.0f; }
            if (SHAPE == 0) {
                f32x4 acc[8];
                for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
                }
                for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
            } else {
                f32x16 acc[4];
                for (int j = 0; j < 4; ++j)
                    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
                }
                for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
            }
        }
    } else if (MODE & 2) {
        if (SIDE == 0) {
            float v[8];
            for (int i = 0; i < 8; ++i) v[i] = lane + i;
            for (int it = 0; it < iters; ++it)
                for (int x = 0; x < side_per_iter; x += 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
                }
            for (int i = 0; i < 8; ++i) s += v[i];
        } else {
            f32x4 v = {0, 0, 0, 0};
            for (int it = 0; it < iters; ++it)
                for (int x = 0; x < side_per_iter; ++x) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&lds[((lane + x * 64 + it) & 4095) * 4]);
                    v += t;
                }
            s = v[0] + v[1] + v[2] + v[3];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K>
float run(K kern, float* d, int iters, int side) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, iters, side);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

template <int SHAPE, int SIDE>
void table(float* d, const char* shape, const char* side_name) {
    const int iters = 2000;
    printf("%s  beside  %s (one block of 8 waves per CU; per iteration: 16 MFMAs of 16x16x32 or 8 of 32x32x16 = 256 matrix cycles)\n", shape, side_name);
    const float m = run(work<1, SHAPE, SIDE>, d, iters, 0);
    printf("   MFMA waves alone: %8.1f us\n", m);
    for (int side : {16, 32, 64, 128}) {
        const float v = run(work<2, SHAPE, SIDE>, d, iters, side);
        const float both = run(work<3, SHAPE, SIDE>, d, iters, side);
        printf("   side %3d per iteration: alone %8.1f us   both %8.1f us   (sum %8.1f, max %8.1f)\n", side, v, both, m + v, m > v ? m : v);
    }
}

template <int ROLE>
void roles(float* d) {
    const int iters = 2000, side = 64;
    const float m = run(work<1, 0, 0, ROLE>, d, iters, 0), v = run(work<2, 0, 0, ROLE>, d, iters, side), b = run(work<3, 0, 0, ROLE>, d, iters, side);
    printf("role map %d (0: MFMA waves 0-3; 1: even waves; 2: waves 0,1,4,5): MFMA alone %.1f, v_fma alone %.1f, both %.1f us\n", ROLE, m, v, b);
}

template <int ROLE, int PRIO, int SIDE>
void prio_row(float* d) {
    const int iters = 2000, side = SIDE == 0 ? 64 : 8;
    const float m = run(work<1, 0, SIDE, ROLE, PRIO>, d, iters, 0), v = run(work<2, 0, SIDE, ROLE, PRIO>, d, iters, side),
                b = run(work<3, 0, SIDE, ROLE, PRIO>, d, iters, side);
    printf("   MFMA waves %s, s_setprio 3 on %s, side = %s: MFMA alone %.1f, side alone %.1f, both %.1f us (sum %.1f, max %.1f)\n",
           ROLE == 0 ? "0-3 (older)" : "4-7 (younger)", PRIO == 0 ? "nobody" : PRIO == 1 ? "the side waves" : "the MFMA waves",
           SIDE == 0 ? "v_pk_fma chains" : "ds_read_b128 + wait", m, v, b, m + v, m > v ? m : v);
}

int main() {
    float* d; hipMalloc(&d, (size_t)256 * 512 * 4);
    roles<0>(d); roles<1>(d); roles<2>(d);
    table<0, 0>(d, "v_mfma_f32_16x16x32_f16", "v_fma_f32 chains");
    table<1, 0>(d, "v_mfma_f32_32x32x16_f16", "v_fma_f32 chains");
    table<0, 1>(d, "v_mfma_f32_16x16x32_f16", "ds_read_b128");
    table<1, 1>(d, "v_mfma_f32_32x32x16_f16", "ds_read_b128");
    printf("who yields to whom on one SIMD (one MFMA wave + one side wave per SIMD)\n");
    prio_row<0, 0, 0>(d); prio_row<0, 1, 0>(d); prio_row<0, 2, 0>(d);
    prio_row<3, 0, 0>(d); prio_row<3, 1, 0>(d); prio_row<3, 2, 0>(d);
    prio_row<0, 0, 1>(d); prio_row<0, 1, 1>(d); prio_row<0, 2, 1>(d);
    prio_row<3, 0, 1>(d); prio_row<3, 1, 1>(d); prio_row<3, 2, 1>(d);
    return 0;
}
