// int16 wav sink on the GPU (reference data/audio.py:12-14 encode_16bits):
//   x *= 32767 / max(0.01, max|x|) * rescale_out;  return x.astype(int16)
// Two HBM-bound passes over one waveform (4 B/sample read each, 2 B/sample written):
// a peak reduction, then scale + truncate-toward-zero.  Only the int16 samples then have
// to cross PCIe (or the xGMI gather) -- half the bytes of the fp32 waveform.
#include "fv_internal.h"

namespace fv {

// |x| >= 0, so the IEEE bit pattern orders like the value: atomicMax on the bits.
__global__ __launch_bounds__(256) void peak_abs_kernel(const float* __restrict__ x, int64_t n,
                                                       unsigned* __restrict__ peak_bits) {
    x += (size_t)blockIdx.y * n;          // one waveform (and one peak) per batch row
    peak_bits += blockIdx.y;
    float m = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t i0 = blockIdx.x * 256LL + threadIdx.x;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n / 4 : 0;   // rows may be unaligned
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = i0; i < n4; i += stride) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int64_t i = n4 * 4 + i0; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(peak_bits, __float_as_uint(m));
}

// scale exactly as numpy does for a float32 array: s = fl32(fl32(32767 / peak) * fl32(rescale))
// when peak > 0.01, else the double-precision constant 32767/0.01*rescale rounded to fp32
// (passed in as low_scale); the product is truncated toward zero like ndarray.astype(int16).
__global__ __launch_bounds__(256) void encode16_kernel(float* __restrict__ x, int64_t n,
                                                       const unsigned* __restrict__ peak_bits,
                                                       float rescale, float low_scale,
                                                       short* __restrict__ out, int scale_in_place) {
    x += (size_t)blockIdx.y * n;
    out += (size_t)blockIdx.y * n;
    const float peak = __uint_as_float(peak_bits[blockIdx.y]);
    const float s = (double)peak > 0.01 ? (32767.f / peak) * rescale : low_scale;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i] * s;
        if (scale_in_place) x[i] = v;
        out[i] = (short)(int)v;
    }
}

int launch_encode16(float* x, int B, int64_t n, float rescale, short* out, unsigned* peak_bits,
                    int scale_in_place, hipStream_t s) {
    if (B <= 0) return 0;
    FV_HIP(hipMemsetAsync(peak_bits, 0, sizeof(unsigned) * B, s));
    if (n <= 0) return 0;
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(peak_abs_kernel, dim3((unsigned)blocks, B), dim3(256), 0, s, x, n, peak_bits);
    FV_HIP(hipGetLastError());
    const float low_scale = (float)(32767.0 / 0.01 * (double)rescale);
    hipLaunchKernelGGL(encode16_kernel, dim3((unsigned)blocks, B), dim3(256), 0, s, x, n, peak_bits, rescale,
                       low_scale, out, scale_in_place);
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fv
