// split-f16 ConvTranspose1d 32 -> 16 channels x 2 (HiFi-GAN light's last upsampler): convtn_kernels.hpp
#include "convtn_kernels.hpp"
#include "fv_internal.h"

namespace fv {

// [K step = tap][row sixteenth][split half][lane][8 halves] of the row-prescaled weights, then the 32 inverse prescales:
// row m = co * 2 + phase, tap 0 multiplies x[u - 1] (kernel index 2 + phase), tap 1 x[u] (kernel index phase)
__global__ void pack_convtn_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, const float* __restrict__ inv, int* range_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * 2 * 2 * 64 * 8) return;
    const int j = i & 7, lane = (i >> 3) & 63, half = (i >> 9) & 1, h = (i >> 10) & 1, s = i >> 11;
    const int m = 16 * h + (lane & 15), co = m >> 1, ph = m & 1, ci = 8 * (lane >> 4) + j;
    const float v = w[((size_t)ci * kTnCout + co) * 4 + (s == 0 ? 2 + ph : ph)] * (1.f / inv[m]);
    const _Float16 h1 = (_Float16)v;
    if (range_flag && !(fabsf(v) < 65520.f)) *range_flag = 1;
    wp[i] = half == 0 ? h1 : (_Float16)((v - (float)h1) * 2048.f);
}

int launch_pack_convtn(const float* w, float* packed, const float* inv, int* range_flag, hipStream_t s) {
    hipLaunchKernelGGL(pack_convtn_kernel, dim3(16), dim3(256), 0, s, w, reinterpret_cast<_Float16*>(packed), inv, range_flag);
    FV_HIP(hipGetLastError());
    return 0;
}

int launch_convtn(const PairParams& pp, int Tout, hipStream_t s) {
    const PairMember& mb = pp.m[0];
    if (!mb.x || !mb.w1 || !mb.y) return fail(FV_ERR_INVALID_ARG, "split-f16 transposed conv (32 -> 16): null tensor");
    if (Tout != 2 * pp.T) return fail(FV_ERR_UNSUPPORTED, "split-f16 transposed conv (32 -> 16): Tout = %d (2 T)", Tout);
    if ((double)kTnCin * pp.T * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 transposed conv (32 -> 16): one utterance's tensor exceeds the 1 GiB "
                    "buffer-descriptor range; split the utterance");
    if (mb.add2 && !mb.add1) return fail(FV_ERR_INVALID_ARG, "split-f16 transposed conv: add2 without add1");
    ConvTnParams p = {};
    p.x = mb.x;
    p.add1 = mb.add1;
    p.add2 = mb.add2;
    p.w = mb.w1;
    p.bias = mb.b1;
    p.y = mb.y;
    p.y_act = mb.y_act;
    p.B = pp.B;
    p.T = pp.T;
    p.Tout = Tout;
    p.n_tiles = (pp.T + 1 + kTnCols - 1) / kTnCols;          // columns u = 0 .. T
    p.slope = pp.slope;
    p.act_slope = pp.act_slope;
    p.out_div = mb.add1 ? pp.out_div : 1.f;
    p.guard = pp.guard;
    profile_begin(s);
    hipLaunchKernelGGL(convtn_kernel, dim3((unsigned)(p.n_tiles * p.B)), dim3(256), 0, s, p);
    profile_end(s, FV_KERNEL_CONVT, 2.0 * pp.B * (double)pp.T * kTnCin * kTnCout * 4,
                4.0 * ((double)kTnCin * kTnCout * 4 + (double)pp.B * ((double)kTnCin * pp.T * (mb.add1 ? (mb.add2 ? 3 : 2) : 1) +
                                                                      (double)kTnCout * Tout * (mb.y_act ? 2 : 1))));
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fv
