// Can the dependent layers of a stage run inside ONE persistent launch with neighbour flags instead of kernel
// boundaries?  Each block writes a 64 KB slab per round, raises its flag, waits for the flags of its two neighbours
// (round-robin block placement: they sit on other XCDs, behind other L2s) and reads their slabs.
//   hipcc --offload-arch=gfx950 -O3 tools/flag_probe.hip -o tools/flag_probe.bin && tools/flag_probe.bin
// Variants of the data accesses: plain (aux 0), sc1 (agent scope: write-through / L2 bypass per access), sc0 sc1,
// and plain accesses bracketed by agent-scope release / acquire fences (whole-L2 write-back + invalidate).
// Printed: us per round against a dependent launch per round, and the number of stale values read.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

constexpr int kSlab = 16384;          // floats per block and parity (64 KB)
constexpr int kSpin = 1 << 20;

template <int AUX, bool FENCE, int NST = 8>
__device__ __forceinline__ void round_body(float* data, unsigned* flags, unsigned* err, int r, int nblk) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    float* mine = data + ((size_t)(r & 1) * nblk + blk) * kSlab;
    const __amdgpu_buffer_rsrc_t rm = rsrc(mine, kSlab * 4);
    const float v = (float)(r * 1024 + blk);
    const unsigned vb = __float_as_uint(v);
#pragma unroll
    for (int i = 0; i < NST; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(uint4_t{vb, vb, vb, vb}, rm, (tid + i * 512) * 16, 0, AUX);
    __builtin_amdgcn_s_waitcnt(0);                      // every store of this thread acknowledged
    __syncthreads();
    if (tid == 0) {
        if (FENCE) __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_store(flags + blk, (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int nb[2] = {(blk + nblk - 1) % nblk, (blk + 1) % nblk};
    if (tid < 2) {
        int spin = 0;
        while (__hip_atomic_load(flags + nb[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spin > kSpin) {
                atomicAdd(err + 1, 1u);
                break;
            }
        }
        if (FENCE) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    unsigned bad = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const __amdgpu_buffer_rsrc_t rn = rsrc(data + ((size_t)(r & 1) * nblk + nb[s]) * kSlab, kSlab * 4);
        const unsigned want = __float_as_uint((float)(r * 1024 + nb[s]));
#pragma unroll
        for (int i = 0; i < NST / 2; ++i) {
            const uint4_t q = __builtin_amdgcn_raw_buffer_load_b128(rn, (tid + (2 * i + s) * 512) * 16, 0, AUX);
            bad += (q.x != want) + (q.y != want) + (q.z != want) + (q.w != want);
        }
    }
    if (bad) atomicAdd(err, bad);
}

template <int AUX, bool FENCE, int NST = 8>
__global__ __launch_bounds__(512) void persistent(float* data, unsigned* flags, unsigned* err, int rounds) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) smem[0] = 0.f;
    for (int r = 0; r < rounds; ++r) round_body<AUX, FENCE, NST>(data, flags, err, r, gridDim.x);
}

__global__ __launch_bounds__(512) void one_round(float* data, unsigned* flags, unsigned* err, int r) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) smem[0] = 0.f;
    // the same traffic, the neighbours' slabs of the round BEFORE (complete at the kernel boundary)
    const int blk = blockIdx.x, tid = threadIdx.x, nblk = gridDim.x;
    const __amdgpu_buffer_rsrc_t rm = rsrc(data + ((size_t)(r & 1) * nblk + blk) * kSlab, kSlab * 4);
    const unsigned vb = __float_as_uint((float)(r * 1024 + blk));
    unsigned bad = 0;
    if (r > 0) {
        const int nb[2] = {(blk + nblk - 1) % nblk, (blk + 1) % nblk};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const __amdgpu_buffer_rsrc_t rn = rsrc(data + ((size_t)((r - 1) & 1) * nblk + nb[s]) * kSlab, kSlab * 4);
            const unsigned want = __float_as_uint((float)((r - 1) * 1024 + nb[s]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4_t q = __builtin_amdgcn_raw_buffer_load_b128(rn, (tid + (2 * i + s) * 512) * 16, 0, 0);
                bad += (q.x != want) + (q.y != want) + (q.z != want) + (q.w != want);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(uint4_t{vb, vb, vb, vb}, rm, (tid + i * 512) * 16, 0, 0);
    if (bad) atomicAdd(err, bad);
}

template <int AUX, bool FENCE, int NST = 8>
static void run(const char* name, float* data, unsigned* flags, unsigned* err, int blocks, int rounds, size_t lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(persistent<AUX, FENCE, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(flags, 0, blocks * 4);
        hipMemset(err, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL((persistent<AUX, FENCE, NST>), dim3(blocks), dim3(512), lds, 0, data, flags, err, rounds);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned h[2];
        hipMemcpy(h, err, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.2f us per round, stale values %u, spin timeouts %u\n", name, ms * 1e3 / rounds, h[0], h[1]);
    }
}

int main() {
    const int blocks = 256, rounds = 400;
    const size_t lds = 148 * 1024;
    float* data;
    unsigned *flags, *err;
    hipMalloc(&data, (size_t)2 * blocks * kSlab * 4);
    hipMemset(data, 0, (size_t)2 * blocks * kSlab * 4);
    hipMalloc(&flags, blocks * 4);
    hipMalloc(&err, 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(one_round), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(err, 0, 8);
        hipEventRecord(e0);
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(one_round, dim3(blocks), dim3(512), lds, 0, data, flags, err, r);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned h[2];
        hipMemcpy(h, err, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.2f us per round, stale values %u\n", "dependent launches", ms * 1e3 / rounds, h[0]);
    }
    run<0, false>("flags, plain accesses", data, flags, err, blocks, rounds, lds);
    run<16, false>("flags, sc1 accesses", data, flags, err, blocks, rounds, lds);
    run<17, false>("flags, sc0 sc1 accesses", data, flags, err, blocks, rounds, lds);
    run<0, true>("flags, plain + fences", data, flags, err, blocks, rounds, lds);
    run<16, false, 2>("flags, sc1, 16 KB slabs", data, flags, err, blocks, rounds, lds);
    run<0, true, 2>("flags, fences, 16 KB slabs", data, flags, err, blocks, rounds, lds);
    return 0;
}
