// Probe: does raw_buffer_load ... lds (16 B per lane) zero-fill LDS for
// out-of-range lanes, and where do lanes land?  (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const float* src, int nbytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    const int lane = threadIdx.x;
    // lanes 0..39 in range, 40..63 out of range
    unsigned off = lane < 40 ? lane * 16u : 0xFFFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds + 64, 16, off, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 100.f + i;
    float *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 2048);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 4096, o);
    float r[512]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
    for (int i = 56; i < 340; i += 1) { if (i % 16 == 0) printf("\n[%3d] ", i); printf("%6.0f", r[i]); }
    printf("\n");
    return 0;
}
