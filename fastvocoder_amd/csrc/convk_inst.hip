// MelGAN ResidualStack as one split-f16 launch, 32 / 64 / 128 channels: convk_kernel of convk_kernels.hpp
#include "convk_kernels.hpp"
#include "fv_internal.h"

namespace fv {

// [stage][row sixteenth][split half][lane][8 halves] of the row-prescaled weights; lane = (row = lane & 15, K block
// kb = lane >> 4): co = 16 r16 + row, ci = 32 cg + 8 kb + j.  Stages: conv1's (tap * CG + cg: w1 [C][C][3]), then W2's
// (w2 [C][C][1], cg), then the skip layer's (ws [C][C][1], cg) -- the last two share the row prescale inv2
__global__ void pack_convk_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ ws,
                                  _Float16* __restrict__ wp, const float* __restrict__ inv1, const float* __restrict__ inv2,
                                  int C, int* range_flag) {
    const int CG = C / 32, NS1 = 3 * CG, NST = 5 * CG, R16 = C / 16;
    const int total = NST * C * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, lane = (i >> 3) & 63, half = (i >> 9) & 1;
        const int sr = i >> 10, r16 = sr % R16, s = sr / R16;
        const int co = 16 * r16 + (lane & 15), kb = lane >> 4;
        float v;
        if (s < NS1) {
            const int tap = s / CG, ci = 32 * (s % CG) + 8 * kb + j;
            v = w1[((size_t)co * C + ci) * 3 + tap] * (1.f / inv1[co]);
        } else if (s < NS1 + CG) {
            v = w2[(size_t)co * C + 32 * (s - NS1) + 8 * kb + j] * (1.f / inv2[co]);
        } else {
            v = ws[(size_t)co * C + 32 * (s - NS1 - CG) + 8 * kb + j] * (1.f / inv2[co]);
        }
        const _Float16 h1 = (_Float16)v;
        if (range_flag && !(fabsf(v) < 65520.f)) *range_flag = 1;   // f16(v) would be inf (or v is not finite)
        wp[i] = half == 0 ? h1 : (_Float16)((v - (float)h1) * 2048.f);
    }
}

// 256 channels (convk2_kernel): [K step][split half][row sixteenth 16][lane][8 halves]; K steps: conv1's in
// chunks of 128 input channels, tap-major inside a chunk (convs_kernel's order), then W2's eight groups, then the skip layer's
__global__ void pack_convk2_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ ws,
                                   _Float16* __restrict__ wp, const float* __restrict__ inv1, const float* __restrict__ inv2,
                                   int* range_flag) {
    constexpr int C = 256, NK1 = 24, NST = 80;
    const int total = NST * 8192;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, lane = (i >> 3) & 63, r16 = (i >> 9) & 15, st = i >> 13;
        const int ks = st >> 1, half = st & 1;
        const int co = 16 * r16 + (lane & 15), kb = lane >> 4;
        float v;
        if (ks < NK1) {
            const int chunk = ks / 12, tap = (ks % 12) / 4, ci = 32 * (4 * chunk + ks % 4) + 8 * kb + j;
            v = w1[((size_t)co * C + ci) * 3 + tap] * (1.f / inv1[co]);
        } else if (ks < NK1 + 8) {
            v = w2[(size_t)co * C + 32 * (ks - NK1) + 8 * kb + j] * (1.f / inv2[co]);
        } else {
            v = ws[(size_t)co * C + 32 * (ks - NK1 - 8) + 8 * kb + j] * (1.f / inv2[co]);
        }
        const _Float16 h1 = (_Float16)v;
        if (range_flag && !(fabsf(v) < 65520.f)) *range_flag = 1;
        wp[i] = half == 0 ? h1 : (_Float16)((v - (float)h1) * 2048.f);
    }
}

int launch_pack_convk(const float* w1, const float* w2, const float* ws, float* packed, int C, int* range_flag, hipStream_t s) {
    const int image = 5 * (C / 32) * C * 32;             // floats
    const int total = image * 2;
    if (C == 256) {
        hipLaunchKernelGGL(pack_convk2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w1, w2, ws,
                           reinterpret_cast<_Float16*>(packed), packed + image, packed + image + C, range_flag);
        FV_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(pack_convk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w1, w2, ws,
                       reinterpret_cast<_Float16*>(packed), packed + image, packed + image + C, C, range_flag);
    FV_HIP(hipGetLastError());
    return 0;
}

template <int C, int DIL, int NM>
static int launch_k2(const ConvKParams& p, hipStream_t s) {
    typedef ConvK2Geom<C, DIL, NM> G;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convk2_kernel<C, DIL, NM>), (size_t)G::LDS_BYTES)) return rc;
    hipLaunchKernelGGL((convk2_kernel<C, DIL, NM>), dim3(p.nblk), dim3(512), (size_t)G::LDS_BYTES, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
template <int C, int NM>
static int launch_k2_dil(const ConvKParams& p, int dil, hipStream_t s) {
    return dil == 1 ? launch_k2<C, 1, NM>(p, s) : dil == 3 ? launch_k2<C, 3, NM>(p, s) : launch_k2<C, 9, NM>(p, s);
}

int launch_convk(const PairParams& pp, int C, int dil, hipStream_t s) {
    const PairMember& mb = pp.m[0];
    if (!convk_shape(C, 3, dil)) return fail(FV_ERR_UNSUPPORTED, "residual stack: C = %d, dilation %d (32 / 64 / 128 / 256 channels, 3 taps, dilation 1, 3 or 9)", C, dil);
    if (!mb.x || !mb.w1 || !mb.y) return fail(FV_ERR_INVALID_ARG, "residual stack: null tensor");
    if (reinterpret_cast<uintptr_t>(mb.w1) & 15) return fail(FV_ERR_UNSUPPORTED, "residual stack: packed weights must be 16-byte aligned");
    if (pp.B <= 0 || pp.T <= 0) return 0;
    if ((double)C * pp.T * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "residual stack: one utterance's tensor (%d x %d floats) exceeds the 1 GiB "
                    "buffer-descriptor range; split the utterance", C, pp.T);
    if (pp.reflect && dil >= pp.T) return fail(FV_ERR_INVALID_ARG, "residual stack: reflection padding %d needs more than %d samples", dil, pp.T);
    if (pp.slope < 0.f || pp.slope > 1.f || pp.act_slope < 0.f || pp.act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "residual stack: activation slope outside [0, 1]");
    ConvKParams p = {};
    p.x = mb.x;
    p.w = mb.w1;
    p.b1 = mb.b1;
    p.b2 = mb.b2;
    p.y = mb.y;
    p.y_act = mb.y_act;
    p.B = pp.B;
    p.T = pp.T;
    // 256 channels: 64-column tiles (half the weight traffic per column, 24 MFMAs per K step and wave) once there are
    // Tuning::stack_wide tenths of a tile per CU that way; 32-column tiles for the latency-bound runs below that
    const bool wide = C == 256 && (long long)pp.B * ((pp.T + 63) / 64) * 10 >= (long long)tuning().stack_wide * device_cu_count();
    const int nm = wide ? 64 : convk_tile_columns(C);
    p.n_tiles = (pp.T + nm - 1) / nm;
    const long long items = (long long)p.n_tiles * pp.B;
    if (items >= (1LL << 31)) return fail(FV_ERR_UNSUPPORTED, "residual stack: too many tiles");
    p.n_items = (int)items;
    long long nblk = tuning().convh_blocks > 0 ? tuning().convh_blocks : device_cu_count();
    if (nblk > items) nblk = items;
    p.nblk = (int)nblk;
    p.slope = pp.slope;
    p.act_slope = pp.act_slope;
    p.post = pp.post;
    p.sub = pp.sub;
    p.sub_batched = pp.sub_batched;
    p.reflect = pp.reflect;
    p.guard = pp.guard;
    profile_begin(s);
    const int rc = C == 256 ? (wide ? launch_k2_dil<256, 64>(p, dil, s) : launch_k2_dil<256, 32>(p, dil, s))
                 : C == 32 ? launch_k2_dil<32, 256>(p, dil, s) : C == 64 ? launch_k2_dil<64, 128>(p, dil, s) : launch_k2_dil<128, 64>(p, dil, s);
    // conv1 (3 taps) + the two 1x1 convs; x in, y (and its twin) out, the weights once
    profile_end(s, FV_KERNEL_STACK, 2.0 * pp.B * (double)C * C * 5 * pp.T,
                4.0 * (5.0 * C * C + (double)pp.B * C * pp.T * (mb.y_act ? 3 : 2)));
    return rc;
}

}  // namespace fv
