// What does a dependent launch cost as a function of its kernel-argument block, its dynamic LDS and its block size?
//   hipcc --offload-arch=gfx950 -O3 tools/launch_probe.hip -o tools/launch_probe.bin && tools/launch_probe.bin
// N back-to-back launches on one stream of a kernel whose every thread does one store; time per launch by events.
// With kernels this short the loop runs at the slower of the host's enqueue rate and the device's dispatch rate: the figures
// are an UPPER bound for the device's dispatch-to-dispatch time, and what 2 KB of arguments add may be host time.
// (The fused-pair launches pass their block schedule by value: a 2.3 KB argument block -- is that visible on the device?)
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int WORDS>
struct Args {
    float* buf;
    int n;
    unsigned pad[WORDS];
};

template <int WORDS>
__global__ __launch_bounds__(1024) void k(Args<WORDS> a) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) smem[0] = 0.f;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // (one argument word is read so that the block cannot be dropped)
    a.buf[i % a.n] = (float)a.pad[(WORDS - 1) & blockIdx.x];
}

template <int WORDS>
float run(float* buf, int n, int blocks, int threads, size_t lds, int rounds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<WORDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024));
    Args<WORDS> a;
    a.buf = buf;
    a.n = n;
    for (int i = 0; i < WORDS; ++i) a.pad[i] = i;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k<WORDS>, dim3(blocks), dim3(threads), lds, 0, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f / rounds;
}

int main() {
    const int n = 1 << 20, rounds = 400;
    float* buf;
    hipMalloc(&buf, n * sizeof(float));
    // warm the device (profiles/r06_clock_ramp.txt)
    for (int i = 0; i < 20000; ++i) run<4>(buf, n, 256, 512, 0, 1), i += 999;
    const size_t ldss[2] = {0, 148 * 1024};
    const int thr[3] = {512, 768, 960};
    printf("us per dependent launch (256 blocks), best of 4 x %d launches\n", rounds);
    printf("%10s %10s | %8s %8s %8s %8s\n", "threads", "LDS", "16 B", "528 B", "2064 B", "4080 B");
    for (int t = 0; t < 3; ++t)
        for (int l = 0; l < 2; ++l) {
            const float a = run<1>(buf, n, 256, thr[t], ldss[l], rounds);
            const float b = run<128>(buf, n, 256, thr[t], ldss[l], rounds);
            const float c = run<512>(buf, n, 256, thr[t], ldss[l], rounds);
            const float d = run<1016>(buf, n, 256, thr[t], ldss[l], rounds);
            printf("%10d %10zu | %8.2f %8.2f %8.2f %8.2f\n", thr[t], ldss[l], a, b, c, d);
        }
    // a grid that does not fill the device / one that is 4 blocks per CU
    printf("blocks 64 / 256 / 1024 (512 threads, no LDS, 16 B): %.2f %.2f %.2f\n", run<1>(buf, n, 64, 512, 0, rounds),
           run<1>(buf, n, 256, 512, 0, rounds), run<1>(buf, n, 1024, 512, 0, rounds));
    return 0;
}
