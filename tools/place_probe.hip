// Where and when do the blocks of a 938 x 256-thread, 32 KiB-LDS MFMA kernel run?
// Records per block: XCC id, HW_ID (SE / CU / SIMD), start and end timestamps.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void work(float* out, unsigned long long* rec, int mfmas, int stages) {
    extern __shared__ float lds[];
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long wt0 = wall_clock64();
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int lane = threadIdx.x & 63;
    const float* pa = lds + lane;
    const float* pb = lds + 4096 + lane;
    for (int s = 0; s < stages; ++s) {
        for (int it = 0; it < mfmas; ++it) {
            float a = pa[(it & 31) * 64];
            float b = pb[(it & 31) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    float sacc = 0.f;
    for (int i = 0; i < 16; ++i) sacc += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = sacc;
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long wt1 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[blockIdx.x * 6 + 0] = hwid;
        rec[blockIdx.x * 6 + 1] = xcc;
        rec[blockIdx.x * 6 + 2] = t0;
        rec[blockIdx.x * 6 + 3] = t1;
        rec[blockIdx.x * 6 + 4] = wt0;
        rec[blockIdx.x * 6 + 5] = wt1;
    }
}

int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 938;
    int lds = argc > 2 ? atoi(argv[2]) : 32768;
    int mfmas = argc > 3 ? atoi(argv[3]) : 44;
    int stages = argc > 4 ? atoi(argv[4]) : 4;
    float* d; unsigned long long* r;
    hipMalloc(&d, (size_t)blocks * 256 * 4); hipMalloc(&r, (size_t)blocks * 48);
    hipFuncSetAttribute((const void*)work, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(work, dim3(blocks), dim3(256), lds, 0, d, r, mfmas, stages);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("rep %d: %.1f us\n", rep, ms * 1e3);
    }
    std::vector<unsigned long long> h((size_t)blocks * 6);
    hipMemcpy(h.data(), r, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0, wmin = ~0ull, wmax = 0;
    std::map<unsigned, int> per_cu;
    double life = 0;
    for (int i = 0; i < blocks; ++i) {
        unsigned hw = (unsigned)h[i * 6], xcc = (unsigned)h[i * 6 + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        tmin = std::min(tmin, h[i * 6 + 2]); tmax = std::max(tmax, h[i * 6 + 3]);
        wmin = std::min(wmin, h[i * 6 + 4]); wmax = std::max(wmax, h[i * 6 + 5]);
        life += (double)(h[i * 6 + 3] - h[i * 6 + 2]);
    }
    int mx = 0; for (auto& kv : per_cu) mx = std::max(mx, kv.second);
    printf("blocks=%d lds=%d: distinct CUs used=%zu, max blocks on one CU=%d, mean block life=%.0f cyc, span=%llu cyc, wall span=%llu ticks\n",
           blocks, lds, per_cu.size(), mx, life / blocks, tmax - tmin, wmax - wmin);
    std::map<int, int> hist; for (auto& kv : per_cu) hist[kv.second]++;
    for (auto& kv : hist) printf("  CUs with %d blocks: %d\n", kv.first, kv.second);
    // first 16 blocks' placement
    for (int i = 0; i < 16; ++i) printf("  block %d: xcc=%llu hwid=0x%llx start=%llu\n", i, h[i*6+1] & 0xf, h[i*6], h[i*6+2] - tmin);
    return 0;
}
