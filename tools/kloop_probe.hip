// How fast can a wave-level K loop of the convh shape run when nothing else is in the way?
//   hipcc --offload-arch=gfx950 -O3 tools/kloop_probe.hip -o tools/kloop_probe.bin && tools/kloop_probe.bin
// Per step a wave issues 4 A + 4 B ds_read_b128 (1 KB each: A linear, B = 16 consecutive 16-byte rows x 4 blocks) and
// 12 v_mfma_f32_16x16x32_f16 (2 row sixteenths x 2 fragments x 3 split terms), operands one step ahead, no barriers,
// no weight stream.  Blocks of W waves, one block per CU (the LDS footprint of convh): cycles per step for
// W = 8 (2 waves per SIMD), 12, 16 -- is the loop bound by the matrix pipe (16 cycles per MFMA: 192 x waves per SIMD),
// by LDS bandwidth (8 KB per wave and step at 256 B/clk) or by latency that more waves would hide?
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: the loop alone; 1: + s_waitcnt lgkmcnt(0); s_barrier every two steps (convh's stage barrier);
// 2: + the weight stream: every stage each wave requests its 2 KB of a 16 KB stage by LDS-DMA (buffer_load ... lds) from
// an L2-resident buffer into the A ring, three stages ahead, and waits for the stage it is about to read (vmcnt(4))
template <int AHEAD, int MODE>
__global__ __launch_bounds__(1024) void kloop(float* out, int steps, long long* cycles, const float* wbuf, int wbytes) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 140 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    const int wm = wave & 1, wn = wave >> 1;
    // A: ring-like region, [step & 7][row sixteenth 4][half 2][lane][16 B]; B: image [half][block 16][row 160][16 B]
    const char* aptr = lds + (wm * 2) * 2048 + lane * 16;
    const char* bptr = lds + 65536 + ((lane >> 4) * 160 + (wn & 3) * 32 + (lane & 15)) * 16;
    f32x4 hi[2][2], lo[2][2];
    for (int h = 0; h < 2; ++h)
        for (int f = 0; f < 2; ++f) hi[h][f] = lo[h][f] = f32x4{0, 0, 0, 0};
    f16x8 a[AHEAD + 1][2][2], b[AHEAD + 1][2][2];
    auto fetch = [&](int s, f16x8 (&A)[2][2], f16x8 (&B)[2][2]) {
        const char* ap = aptr + (s & 7) * 8192;
        const char* bp = bptr + ((s & 3) * 4 * 160 + (s % 11)) * 16;
        for (int h = 0; h < 2; ++h) {
            A[h][0] = *reinterpret_cast<const f16x8*>(ap + h * 2048);
            A[h][1] = *reinterpret_cast<const f16x8*>(ap + h * 2048 + 1024);
        }
        for (int e = 0; e < 2; ++e) {
            B[e][0] = *reinterpret_cast<const f16x8*>(bp + e * 256);
            B[e][1] = *reinterpret_cast<const f16x8*>(bp + e * 256 + 40960);
        }
    };
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbuf), 0, wbytes, 0x00020000);
    auto dma = [&](int stage) {
        float* dst = reinterpret_cast<float*>(lds) + (stage & 3) * 4096 + (wave & 7) * 512;
        const unsigned o = (unsigned)(((blockIdx.x * 7 + stage) * 16384) % (wbytes - 16384)) + (unsigned)((wave & 7) * 2048 + lane * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, (int)o, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + 256), 16, (int)(o + 1024u), 0, 0, 0);
    };
    if (MODE == 2) {
        for (int g = 0; g < 3; ++g) dma(g);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    for (int q = 0; q < AHEAD; ++q) fetch(q, a[q], b[q]);
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += AHEAD + 1) {
#pragma unroll
        for (int q = 0; q <= AHEAD; ++q) {
            const int s = s0 + q;
            if (MODE >= 1 && (s & 1) == 0) {
                if (MODE == 2) __builtin_amdgcn_s_waitcnt(0x0F74);        // vmcnt(4): this stage's DMA (issued 3 stages ago) has landed
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (MODE == 2) dma(s / 2 + 3);
            }
            fetch(s + AHEAD, a[(q + AHEAD) % (AHEAD + 1)], b[(q + AHEAD) % (AHEAD + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 2; ++e) hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q][h][0], b[q][e][0], hi[h][e], 0, 0, 0);
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 2; ++e) lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q][h][0], b[q][e][1], lo[h][e], 0, 0, 0);
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 2; ++e) lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q][h][1], b[q][e][0], lo[h][e], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0;
    for (int h = 0; h < 2; ++h)
        for (int f = 0; f < 2; ++f)
            for (int i = 0; i < 4; ++i) acc += hi[h][f][i] + lo[h][f][i];
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int AHEAD, int MODE>
void run(int waves, float* out, long long* cyc, const float* wbuf, int wbytes) {
    const int steps = 6000;
    const size_t lds = 150 * 1024;
    auto kern = kloop<AHEAD, MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), lds, 0, out, steps, cyc, wbuf, wbytes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = 256.0 * waves * steps * 12 * 2.0 * 16 * 16 * 32;
    printf("mode %d, operands %d step(s) ahead, %2d waves per block (%.1f per SIMD): %.2f ms, %.0f ns per step = %.0f shader cycles at 2.4 GHz "
           "(s_memtime delta %lld per step), %.0f TFLOP/s of f16 MFMA\n",
           MODE, AHEAD, waves, waves / 4.0, ms, ms * 1e6 / steps, ms * 1e6 / steps * 2.4, c / steps, flops / (ms * 1e-3) / 1e12);
}

// The same loop on v_mfma_f32_32x32x16_f16 (round 6: VERDICT r5 asked to settle it).  The wave tile is the same 32 x 32 outputs and a
// step the same 32 K: 2 K halves x 3 split terms = 6 MFMAs of 32 cycles instead of 12 of 16, and -- the wave tile decides the operand
// traffic, not the instruction shape -- the same 4 A + 4 B ds_read_b128 per step (A: [32 rows][8 halves] per K quarter, one 1 KB
// read per (K half, split half); B alike).
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int AHEAD>
__global__ __launch_bounds__(1024) void kloop32(float* out, int steps, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 140 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    const int wm = wave & 1, wn = wave >> 1;
    const char* aptr = lds + (wm * 2) * 2048 + lane * 16;
    const char* bptr = lds + 65536 + ((lane >> 5) * 160 + (wn & 3) * 32 + (lane & 31)) * 16;
    f32x16 hi, lo;
    for (int i = 0; i < 16; ++i) hi[i] = lo[i] = 0.f;
    f16x8 a[AHEAD + 1][2][2], b[AHEAD + 1][2][2];     // [K half][split half]
    auto fetch = [&](int s, f16x8 (&A)[2][2], f16x8 (&B)[2][2]) {
        const char* ap = aptr + (s & 7) * 8192;
        const char* bp = bptr + ((s & 3) * 4 * 160 + (s % 11)) * 16;
        for (int h = 0; h < 2; ++h) {
            A[h][0] = *reinterpret_cast<const f16x8*>(ap + h * 2048);
            A[h][1] = *reinterpret_cast<const f16x8*>(ap + h * 2048 + 1024);
        }
        for (int e = 0; e < 2; ++e) {
            B[e][0] = *reinterpret_cast<const f16x8*>(bp + e * 2 * 160 * 16);
            B[e][1] = *reinterpret_cast<const f16x8*>(bp + e * 2 * 160 * 16 + 40960);
        }
    };
    for (int q = 0; q < AHEAD; ++q) fetch(q, a[q], b[q]);
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += AHEAD + 1) {
#pragma unroll
        for (int q = 0; q <= AHEAD; ++q) {
            const int s = s0 + q;
            fetch(s + AHEAD, a[(q + AHEAD) % (AHEAD + 1)], b[(q + AHEAD) % (AHEAD + 1)]);
            __builtin_amdgcn_sched_barrier(0);
            for (int h = 0; h < 2; ++h) hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q][h][0], b[q][h][0], hi, 0, 0, 0);
            for (int h = 0; h < 2; ++h) lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q][h][0], b[q][h][1], lo, 0, 0, 0);
            for (int h = 0; h < 2; ++h) lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q][h][1], b[q][h][0], lo, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0;
    for (int i = 0; i < 16; ++i) acc += hi[i] + lo[i];
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int AHEAD>
void run32(int waves, float* out, long long* cyc) {
    const int steps = 6000;
    const size_t lds = 150 * 1024;
    auto kern = kloop32<AHEAD>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), lds, 0, out, steps, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * waves * steps * 12 * 2.0 * 16 * 16 * 32;
    printf("32x32x16, operands %d step(s) ahead, %2d waves per block (%.1f per SIMD): %.2f ms, %.0f ns per step, %.0f TFLOP/s of f16 MFMA\n",
           AHEAD, waves, waves / 4.0, ms, ms * 1e6 / steps, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 8);
    float* wbuf;
    const int wbytes = 3 << 20;                       // 3 MB: the packed weights of a three-member launch, L2-resident
    hipMalloc(&wbuf, wbytes);
    hipMemset(wbuf, 0, wbytes);
    for (int waves : {8, 12, 16}) run32<1>(waves, out, cyc);
    for (int waves : {8, 12, 16}) run32<2>(waves, out, cyc);
    for (int waves : {8, 12, 16}) run<1, 0>(waves, out, cyc, wbuf, wbytes);
    for (int waves : {8, 12, 16}) run<2, 0>(waves, out, cyc, wbuf, wbytes);
    for (int waves : {8, 16}) run<1, 1>(waves, out, cyc, wbuf, wbytes);
    for (int waves : {8, 16}) run<2, 1>(waves, out, cyc, wbuf, wbytes);
    for (int waves : {8, 16}) run<1, 2>(waves, out, cyc, wbuf, wbytes);
    for (int waves : {8, 16}) run<2, 2>(waves, out, cyc, wbuf, wbytes);
    return 0;
}
