// What does a kernel boundary cost against a device-wide barrier inside one persistent kernel?
//   hipcc --offload-arch=gfx950 -O3 tools/gridsync_probe.hip -o tools/gridsync_probe.bin && tools/gridsync_probe.bin
// (a) N dependent launches of a 256-block x 512-thread kernel with 148 KB of dynamic LDS that touches a little memory;
// (c) the same launches replayed from a hipGraph;
// (b) ONE launch of the same grid doing the same work N times with a sense-free counter barrier + agent-scope fences
//     (release: L2 write-back, acquire: L1 / L2 invalidate) between the rounds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);                               // this block's stores are visible device-wide
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

__device__ __forceinline__ void work(float* buf, int round, int n) {
    // every thread reads a value another block wrote in the previous round and writes one for the next
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (i + 4099 * 512) % n;
    buf[(round & 1) * n + i] = buf[((round + 1) & 1) * n + j] + 1.f;
}

__global__ __launch_bounds__(512) void one_round(float* buf, int round, int n) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) smem[0] = 0.f;
    work(buf, round, n);
}

__global__ __launch_bounds__(512) void persistent(float* buf, int rounds, int n, unsigned* ctr) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) smem[0] = 0.f;
    for (int r = 0; r < rounds; ++r) {
        work(buf, r, n);
        grid_barrier(ctr, (unsigned)(r + 1) * gridDim.x);
    }
}

int main() {
    const int blocks = 256, threads = 512, n = blocks * threads, rounds = 200;
    const size_t lds = 148 * 1024;
    float* buf;
    unsigned* ctr;
    hipMalloc(&buf, 2 * n * sizeof(float));
    hipMemset(buf, 0, 2 * n * sizeof(float));
    hipMalloc(&ctr, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(one_round), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(persistent), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(one_round, dim3(blocks), dim3(threads), lds, 0, buf, r, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%d dependent launches: %.2f us each\n", rounds, ms * 1e3 / rounds);
        // (c) the same N dependent launches captured once into a hipGraph and replayed
        static hipGraphExec_t exec = nullptr;
        static hipStream_t cs = nullptr;
        if (!exec) {
            hipStreamCreate(&cs);
            hipGraph_t g;
            hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal);
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(one_round, dim3(blocks), dim3(threads), lds, cs, buf, r, n);
            hipStreamEndCapture(cs, &g);
            hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
            hipGraphLaunch(exec, cs);
            hipStreamSynchronize(cs);
        }
        hipEventRecord(e0, cs);
        hipGraphLaunch(exec, cs);
        hipEventRecord(e1, cs);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%d dependent launches replayed from a hipGraph: %.2f us each\n", rounds, ms * 1e3 / rounds);
        hipMemset(ctr, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(persistent, dim3(blocks), dim3(threads), lds, 0, buf, rounds, n, ctr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("one persistent launch, %d grid barriers: %.2f us per round\n", rounds, ms * 1e3 / rounds);
    }
    float h[4];
    hipMemcpy(h, buf, sizeof h, hipMemcpyDeviceToHost);
    printf("check %.0f %.0f\n", h[0], h[1]);
    return 0;
}
