// fused ResBlock1 pairs, C = 16: 8 waves x two 16-column fragments (256-column tiles)
#include "pair_inst.hpp"
namespace fv {
template int launch_pair_geom<1, 2, 8>(const PairParams&, int, size_t, hipStream_t);

// Self-check of pair_kernels.hpp div_exact on the device: bits [first, first + n) as fp32 values (and their negatives),
// counted where the three-instruction quotient differs from the hardware's IEEE division by d.
__global__ void div_probe_kernel(unsigned first, long long n, float d, unsigned long long* mismatches) {
    const float r = div_rcp(d);
    unsigned long long bad = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = __uint_as_float(first + (unsigned)i);
        if (!(fabsf(v) < __builtin_inff())) continue;
        const float q = v / d;
        if (fabsf(q) < 1.17549435e-38f) continue;                 // a normal quotient is what the kernels promise
        const float a = r != 0.f ? div_exact(v, d, r) : q, b = r != 0.f ? div_exact(-v, d, r) : -q;
        bad += (__float_as_uint(a) != __float_as_uint(q)) + (__float_as_uint(b) != __float_as_uint(-q));
    }
    if (bad) atomicAdd(mismatches, bad);
}

int launch_div_probe(unsigned first, long long n, float d, unsigned long long* mismatches, hipStream_t s) {
    hipLaunchKernelGGL(div_probe_kernel, dim3(2048), dim3(256), 0, s, first, n, d, mismatches);
    FV_HIP(hipGetLastError());
    return 0;
}
}
