// What does a VALU instruction cost on gfx950 (cycles of its SIMD's issue time), by kind?  The split-f16 epilogues were cut by
// INSTRUCTION COUNT (packed fp32 pairs: v_pk_mul_f32 / v_pk_fma_f32, v_cvt_pk_f16_f32, v_fma_mix*); on a SIMD whose matrix and
// vector instructions serialise (tools/pingpong_probe.hip) what counts is their issue TIME.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe.bin
// One block of W waves per CU (W = 4: one per SIMD; 8: two), each wave a loop of 64 independent instructions of one kind
// (16 registers x 4 rounds: dependency distance 16); cycles per instruction and SIMD from s_memtime of wave 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(512) void work(float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = f32x2{lane + i * 0.5f, lane * 0.25f + i};
    f32x2 c = {1.0001f, 0.9999f}, d = {0.5f, 0.25f};
    asm volatile("" : "+v"(c), "+v"(d));
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define OP(i)                                                                                                              \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(c.x), "v"(d.x));                           \
    else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));                        \
    else if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));                                     \
    else if (KIND == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(c.x));                                    \
    else if (KIND == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(c.x));                                    \
    else if (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[i].x) : "v"(v[i].y), "v"(c.x));                \
    else if (KIND == 6) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(v[i].x) : "v"(v[i].y), "v"(c.x), "v"(d.x)); \
    else if (KIND == 7) asm volatile("v_max3_f32 %0, |%0|, |%1|, |%2|" : "+v"(v[i].x) : "v"(c.x), "v"(d.x));               \
    else if (KIND == 8) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(v[i].x) : "v"(v[i].y));                                 \
    else if (KIND == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));                                     \
    else if (KIND == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(c.x));                                   \
    else if (KIND == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i].x) : "v"(c.x));                          \
    else if (KIND == 12) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(v[i].x) : "v"(v[i].y), "v"(c.x));            \
    else if (KIND == 13) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i].x) : "v"(c.x));                                \
    else if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i].x) : "v"(v[i].y));
            REP16(OP)
#undef OP
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 4000;
    for (int waves : {4, 8}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(work<KIND>, dim3(256), dim3(64 * waves), 0, 0, out, iters, cyc);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_wave = (double)c / (iters * 64.0);
        printf("%-22s %d wave(s) per SIMD: %6.2f clock ticks per instruction and wave, %6.2f per instruction and SIMD (launch %.0f us)\n", name,
               waves / 4, per_wave, per_wave / (waves / 4), best * 1e3);
    }
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<3>("v_mul_f32", out, cyc);
    run<2>("v_pk_mul_f32", out, cyc);
    run<10>("v_add_f32", out, cyc);
    run<9>("v_pk_add_f32", out, cyc);
    run<4>("v_max_f32", out, cyc);
    run<7>("v_max3_f32 |.|", out, cyc);
    run<11>("v_cndmask_b32", out, cyc);
    run<14>("v_mov_b32", out, cyc);
    run<8>("v_cvt_f16_f32", out, cyc);
    run<5>("v_cvt_pk_f16_f32", out, cyc);
    run<12>("v_cvt_pkrtz_f16_f32", out, cyc);
    run<6>("v_fma_mixlo_f16", out, cyc);
    run<13>("v_pk_mul_f16", out, cyc);
    return 0;
}
