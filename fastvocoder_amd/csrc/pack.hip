// Weight preparation: the one-off kernels that fold weight norm / BatchNorm and lay a layer's weights out for the conv
// kernels (K-major fp32 images, polyphase transposed convs, split-f16 MFMA A-operand images with a power-of-two prescale
// per row), and the C ABI entry points that run them (include/fastvocoder_hip.h, "one-off weight preparation").
// (Until round 5 this lived in api.hip.)
#include <math.h>

#include "fv_internal.h"

namespace fv {

// ---------------------------------------------------------------------------
// weight preparation
// ---------------------------------------------------------------------------

// One block per dim-0 row: ||v||_2 by a wave-shuffle + LDS tree, then scale.
__global__ __launch_bounds__(256) void fold_weight_norm_kernel(const float* __restrict__ v,
                                                               const float* __restrict__ g,
                                                               float* __restrict__ w,
                                                               int64_t inner) {
    __shared__ float part[4];
    const int r = blockIdx.x;
    const float* vr = v + (size_t)r * inner;
    float ss = 0.f;
    for (int64_t i = threadIdx.x; i < inner; i += 256) ss = fmaf(vr[i], vr[i], ss);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf(part[0] + part[1] + part[2] + part[3]);
    const float scale = g[r] / nrm;
    for (int64_t i = threadIdx.x; i < inner; i += 256) w[(size_t)r * inner + i] = vr[i] * scale;
}

// Eval-mode BatchNorm folded into the conv after it; one block per output channel:
// w'[co,ci,j] = w * a[ci], b'[co] = b + sum w * c[ci]  (a = gamma/sqrt(var+eps), c = beta - mean*a).
__global__ __launch_bounds__(256) void fold_batchnorm_conv_kernel(
    const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var,
    float eps, float* __restrict__ w_out, float* __restrict__ b_out, int Cin, int k) {
    __shared__ float part[4];
    const int co = blockIdx.x;
    const int inner = Cin * k;
    float shift = 0.f;
    for (int i = threadIdx.x; i < inner; i += 256) {
        const int ci = i / k;
        const float a = (gamma ? gamma[ci] : 1.f) / sqrtf(var[ci] + eps);
        const float c = (beta ? beta[ci] : 0.f) - mean[ci] * a;
        const float wv = w[(size_t)co * inner + i];
        w_out[(size_t)co * inner + i] = wv * a;
        shift = fmaf(wv, c, shift);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) shift += __shfl_down(shift, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = shift;
    __syncthreads();
    if (threadIdx.x == 0) b_out[co] = (b ? b[co] : 0.f) + ((part[0] + part[1]) + (part[2] + part[3]));
}

// Conv1d weight [Cout, Cin, k] -> Wp[(ci*k + j)][Mpad], zero in the pad rows.
__global__ void pack_conv1d_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                   int Cin, int k, int Mpad) {
    const int64_t total = (int64_t)Cin * k * Mpad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % Mpad);
        const int64_t row = i / Mpad;
        const int j = (int)(row % k), ci = (int)(row / k);
        wp[i] = m < Cout ? w[((size_t)m * Cin + ci) * k + j] : 0.f;
    }
}

// ConvTranspose1d weight [Cin, Cout, k] -> polyphase image
// Wp[(ci*taps + jj)][m = co*s + r] = w[ci, co, a + (c - dmin - jj)*s],
// a = (r+p) % s, c = (r+p) / s, zero where that tap index falls outside [0,k).
__global__ void pack_convT_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin,
                                  int Cout, int k, int s, int p, int dmin, int taps, int Mpad) {
    const int64_t total = (int64_t)Cin * taps * Mpad;
    const int M = Cout * s;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % Mpad);
        const int64_t row = i / Mpad;
        const int jj = (int)(row % taps), ci = (int)(row / taps);
        float val = 0.f;
        if (m < M) {
            const int co = m / s, r = m - co * s;
            const int a = (r + p) % s, c = (r + p) / s;
            const int mi = c - dmin - jj;
            const int j = a + mi * s;
            if (mi >= 0 && j < k) val = w[((size_t)ci * Cout + co) * k + j];
        }
        wp[i] = val;
    }
}

// ConvTranspose1d weight [Cin, Cout, k], k = tp*s -> phase-major image
// Wp[(ci*tp + jj)][m = r*Cout + co] = w[ci, co, (r+p) % s + (tp-1-jj)*s]   (fv_internal.h convt_phase_major)
__global__ void pack_convT_phase_major_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin,
                                              int Cout, int k, int s, int p, int tp, int Mpad) {
    const int64_t total = (int64_t)Cin * tp * Mpad;
    const int M = Cout * s;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % Mpad);
        const int64_t row = i / Mpad;
        const int jj = (int)(row % tp), ci = (int)(row / tp);
        float val = 0.f;
        if (m < M) {
            const int r = m / Cout, co = m - r * Cout;
            const int j = (r + p) % s + (tp - 1 - jj) * s;
            val = w[((size_t)ci * Cout + co) * k + j];
        }
        wp[i] = val;
    }
}

// UpsampleLayer weight [Cout, Cin, k] (nearest-repeat x u, then conv with zero padding p) ->
// phase image Wp[(ci*taps + jj)][m = co*u + r] = sum of w[co, ci, j] over the taps j with
// floor((r + j - p) / u) == dmin + jj (they all read the same input sample).
__global__ void pack_upconv_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                   int Cin, int k, int u, int p, int dmin, int taps, int Mpad) {
    const int64_t total = (int64_t)Cin * taps * Mpad;
    const int M = Cout * u;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % Mpad);
        const int64_t row = i / Mpad;
        const int jj = (int)(row % taps), ci = (int)(row / taps);
        float val = 0.f;
        if (m < M) {
            const int co = m / u, r = m - co * u;
            const int j0 = (dmin + jj) * u - r + p;      // first tap of this delta
            for (int j = max(j0, 0); j < min(j0 + u, k); ++j) val += w[((size_t)co * Cin + ci) * k + j];
        }
        wp[i] = val;
    }
}

// Conv1d weight [C, C, k] -> the A-fragment image of the fused ResBlock pair kernels (pair_kernels.hpp):
// Wl[row half h][step group g][lane][e], step s = 4g + e = (channel group cg = s / k, tap = s % k),
// value w[co = 16h + (lane & 15)][ci = 4cg + (lane >> 4)][tap] -- what lane `lane` feeds to the s-th
// v_mfma_f32_16x16x4_f32 of its row half, so a wave loads 4 steps with one 16-byte LDS read.
__global__ void pack_pair_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int k) {
    const int64_t total = (int64_t)C * C * k;
    const int groups = (C / 16) * k;            // step groups per row half: S / 4, S = C * k / 4
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const int64_t hg = i >> 8;
        const int g = (int)(hg % groups), h = (int)(hg / groups);
        const int s = 4 * g + e, cg = s / k, tap = s - cg * k;
        const int co = 16 * h + (lane & 15), ci = 4 * cg + (lane >> 4);
        wp[i] = w[((size_t)co * C + ci) * k + tap];
    }
}

// Conv1d weight [C, C, k] -> the split-f16 A operands of pairh_kernels.hpp:
// Wh[(K step s * MH + row half h) * 2 + split half][lane][8 halves]; lane = (row m = lane & 15, K block g = lane >> 4),
// (range_flag: an optional device-visible word that is set to 1 when a weight is outside the f16 range -- that layer has
// to run on the fp32 kernels)
constexpr float kSplitLimit = 65520.f;   // the smallest magnitude that rounds to inf in f16

// Power-of-two prescale of the split-f16 weights, per GEMM row (= output channel; transposed conv: (output channel, phase)).
// f16 halves keep 22 bits of a weight only while h1 is a NORMAL f16 (|w| >= 2^-14): a layer whose weights are small --
// and whose activations are correspondingly large, the product being O(1) -- would lose them.  The pack functions therefore
// scale every row by 2^e so that its largest magnitude lands in [2^13, 2^14) (exact in fp32; elements down to 2^-28 of the
// row's maximum keep their 22 bits, smaller ones an absolute 2^-50 of it), and write 2^-e behind the image: the kernels'
// epilogues multiply the accumulated sum by it inside the fused multiply-add that adds the bias -- acc * 2^-e is exact, so
// the result is the one an unscaled weight of unlimited f16 exponent range would give.  Weight overflow cannot happen any
// more (the flag remains for non-finite weights).
// mode 0: Conv1d w [rows][n] (n = Cin k);  1: two 1x1 convs [W1 | W2], w [rows][n], w2 [rows][n];
// mode 2: ConvTranspose1d w [Cin = n][Cout][2 s], rows m = co s + phase (rows >= Cout s: padding, scale 1)
__global__ __launch_bounds__(64) void row_scale_kernel(const float* __restrict__ w, const float* __restrict__ w2,
                                                       float* __restrict__ inv, int rows, int n, int mode, int Cout, int s_) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float m = 0.f;
    bool finite = true;
    auto take = [&](float v) {
        finite = finite && fabsf(v) < __builtin_inff();
        m = fmaxf(m, fabsf(v));
    };
    if (mode == 2) {
        const int co = r / s_, ph = r - co * s_;
        if (co < Cout)
            for (int ci = lane; ci < n; ci += 64) {
                take(w[((size_t)ci * Cout + co) * (2 * s_) + ph]);
                take(w[((size_t)ci * Cout + co) * (2 * s_) + s_ + ph]);
            }
    } else {
        for (int i = lane; i < n; i += 64) {
            take(w[(size_t)r * n + i]);
            if (mode == 1) take(w2[(size_t)r * n + i]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        m = fmaxf(m, __shfl_xor(m, o));
        finite = __shfl_xor((int)finite, o) != 0 && finite;
    }
    if (lane == 0) {
        int e = 0;
        if (finite && m > 0.f) {
            int x;
            frexpf(m, &x);                    // m = f 2^x, f in [0.5, 1)
            e = 14 - x;                       // m 2^e in [2^13, 2^14)
            e = e > 110 ? 110 : e < -110 ? -110 : e;
        }
        inv[r] = ldexpf(1.f, -e);
    }
}
// C = 16: tap = 2s + (g >> 1), channels 8 (g & 1) .. + 7 (an odd tap count is padded with a zero tap);
// C = 32: tap = s, channels 8g .. 8g + 7.  Split: h1 = f16(w), h2 = f16((w - h1) * 2048), round to nearest.
__global__ void pack_pairh_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, const float* __restrict__ inv, int C, int k, int* range_flag) {
    const int MH = C / 16, tps = 32 / C, KS = (k + tps - 1) / tps;
    const int64_t total = (int64_t)KS * MH * 2 * 64 * 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), half = (int)((i >> 9) & 1);
        const int sh = (int)(i >> 10), h = sh % MH, s = sh / MH;
        const int g = lane >> 4, co = 16 * h + (lane & 15);
        const int tap = tps == 2 ? 2 * s + (g >> 1) : s;
        const int ci = tps == 2 ? 8 * (g & 1) + j : 8 * g + j;
        const float v = (tap < k ? w[((size_t)co * C + ci) * k + tap] : 0.f) * (1.f / inv[co]);
        const _Float16 h1 = (_Float16)v;
        if (range_flag && !(fabsf(v) < kSplitLimit)) *range_flag = 1;   // f16(v) would be inf (or v is not finite)
        wp[i] = half == 0 ? h1 : (_Float16)((v - (float)h1) * 2048.f);
    }
}

// Conv1d weight [C, C, k], C = 64 / 128 -> the streamed split-f16 A operands of convh_kernels.hpp:
// Wh[row tile mt][K step s = tap * C/32 + cg][row sixteenth mh][split half][lane][8 halves];
// lane = (row = lane & 15, K block kb = lane >> 4): co = 64 mt + 16 mh + row, ci = 32 cg + 8 kb + j.
__global__ void pack_convh_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, const float* __restrict__ inv, int C, int k, int* range_flag) {
    // above 128 channels the input channels come in chunks of 128: [row tile][chunk][step inside the chunk]...
    const int NCH = C > 128 ? C / 128 : 1, CC = C / NCH, CG = CC / 32, NSTEP = k * CG;
    const int64_t total = (int64_t)(C / 64) * NCH * NSTEP * 4 * 2 * 64 * 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), half = (int)((i >> 9) & 1), mh = (int)((i >> 10) & 3);
        const int ms = (int)(i >> 12), s = ms % NSTEP, mc = ms / NSTEP, chunk = mc % NCH, mt = mc / NCH;
        const int tap = s / CG, cg = s % CG;
        const int co = 64 * mt + 16 * mh + (lane & 15), ci = CC * chunk + 32 * cg + 8 * (lane >> 4) + j;
        const float v = w[((size_t)co * C + ci) * k + tap] * (1.f / inv[co]);
        const _Float16 h1 = (_Float16)v;
        if (range_flag && !(fabsf(v) < kSplitLimit)) *range_flag = 1;   // f16(v) would be inf (or v is not finite)
        wp[i] = half == 0 ? h1 : (_Float16)((v - (float)h1) * 2048.f);
    }
}

// Two 1x1 conv weights as one [C][2 C] matrix [W1 | W2] (w1, w2: [C][C][1]) for convg_kernel: the stage layout of
// pack_convh_kernel with one tap and 2 C / 128 chunks of 128 input channels -- W1's, then W2's
__global__ void pack_convg_kernel(const float* __restrict__ w1, const float* __restrict__ w2, _Float16* __restrict__ wp,
                                  const float* __restrict__ inv, int C, int* range_flag) {
    const int NCH = 2 * C / 128, NSTEP = 4;
    const int64_t total = (int64_t)(C / 64) * NCH * NSTEP * 4 * 2 * 64 * 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), half = (int)((i >> 9) & 1), mh = (int)((i >> 10) & 3);
        const int ms = (int)(i >> 12), s = ms % NSTEP, mc = ms / NSTEP, chunk = mc % NCH, mt = mc / NCH;
        const int co = 64 * mt + 16 * mh + (lane & 15), ci = 128 * chunk + 32 * s + 8 * (lane >> 4) + j;
        const float v = (ci < C ? w1[(size_t)co * C + ci] : w2[(size_t)co * C + (ci - C)]) * (1.f / inv[co]);
        const _Float16 h1 = (_Float16)v;
        if (range_flag && !(fabsf(v) < kSplitLimit)) *range_flag = 1;
        wp[i] = half == 0 ? h1 : (_Float16)((v - (float)h1) * 2048.f);
    }
}

// ConvTranspose1d weights w[Cin][Cout][2 s] for convt_kernel (convh_kernels.hpp): the same stage layout, rows
// m = co * s + phase, K = (tap, ci): tap 0 multiplies x[u - 1] (kernel index s + phase), tap 1 x[u] (kernel index phase)
// (64 input channels: one chunk of 64 = two 32-channel groups per tap; otherwise chunks of 128)
__global__ void pack_convth_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, const float* __restrict__ inv, int Cin, int Cout, int s_, int* range_flag) {
    const int CC = Cin <= 64 ? 64 : 128, NCH = (Cin + CC - 1) / CC, CG = CC / 32, NSTEP = 2 * CG, k = 2 * s_;
    const int64_t total = (int64_t)((Cout * s_ + 63) / 64) * NCH * NSTEP * 4 * 2 * 64 * 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), half = (int)((i >> 9) & 1), mh = (int)((i >> 10) & 3);
        const int ms = (int)(i >> 12), st = ms % NSTEP, mc = ms / NSTEP, chunk = mc % NCH, mt = mc / NCH;
        const int tap = st / CG, cg = st % CG;
        const int m = 64 * mt + 16 * mh + (lane & 15), co = m / s_, ph = m - co * s_;
        const int ci = CC * chunk + 32 * cg + 8 * (lane >> 4) + j;
        const float v = (ci < Cin && co < Cout ? w[((size_t)ci * Cout + co) * k + (tap == 0 ? s_ + ph : ph)] : 0.f) * (1.f / inv[m]);
        const _Float16 h1 = (_Float16)v;
        if (range_flag && !(fabsf(v) < kSplitLimit)) *range_flag = 1;   // f16(v) would be inf (or v is not finite)
        wp[i] = half == 0 ? h1 : (_Float16)((v - (float)h1) * 2048.f);
    }
}


}  // namespace fv

using namespace fv;

extern "C" {

int fv_fold_weight_norm(const float* v, const float* g, float* w, int dim0, int64_t inner,
                        void* stream) {
    if (dim0 <= 0 || inner <= 0) return fail(FV_ERR_INVALID_ARG, "fold: dim0=%d inner=%lld", dim0, (long long)inner);
    hipLaunchKernelGGL(fold_weight_norm_kernel, dim3(dim0), dim3(256), 0, (hipStream_t)stream, v, g,
                       w, inner);
    FV_HIP(hipGetLastError());
    return 0;
}

int fv_fold_batchnorm_conv(const float* w, const float* b, const float* gamma, const float* beta,
                           const float* mean, const float* var, float eps, float* w_out, float* b_out,
                           int Cout, int Cin, int k, void* stream) {
    if (int rc = check_conv_args(Cin, Cout, k, 1)) return rc;
    if (!w || !mean || !var || !w_out || !b_out) return fail(FV_ERR_INVALID_ARG, "fold_batchnorm: null tensor");
    hipLaunchKernelGGL(fold_batchnorm_conv_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, b,
                       gamma, beta, mean, var, eps, w_out, b_out, Cin, k);
    FV_HIP(hipGetLastError());
    return 0;
}

int64_t fv_packed_conv1d_floats(int Cout, int Cin, int k) {
    return (int64_t)Cin * k * pad_rows(Cout);
}

int64_t fv_packed_conv_transpose1d_floats(int Cin, int Cout, int k, int stride, int pad) {
    if (convt_phase_major(Cout, k, stride, pad)) return (int64_t)Cin * (k / stride) * pad_rows(Cout * stride);
    const Polyphase ph = polyphase(k, stride, pad);
    return (int64_t)Cin * ph.taps * pad_rows(Cout * stride);
}

int fv_pack_conv1d_weight(const float* w, float* packed, int Cout, int Cin, int k, void* stream) {
    if (int rc = check_conv_args(Cin, Cout, k, 1)) return rc;
    const int Mpad = pad_rows(Cout);
    const int64_t total = (int64_t)Cin * k * Mpad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv1d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                       packed, Cout, Cin, k, Mpad);
    FV_HIP(hipGetLastError());
    return 0;
}

int fv_pack_conv_transpose1d_weight(const float* w, float* packed, int Cin, int Cout, int k,
                                    int stride, int pad, void* stream) {
    if (int rc = check_conv_args(Cin, Cout, k, 1)) return rc;
    if (stride <= 0 || pad < 0) return fail(FV_ERR_INVALID_ARG, "convT stride=%d pad=%d", stride, pad);
    if (convt_phase_major(Cout, k, stride, pad)) {
        const int tp = k / stride, Mp = pad_rows(Cout * stride);
        const int64_t tot = (int64_t)Cin * tp * Mp;
        const int nb = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
        hipLaunchKernelGGL(pack_convT_phase_major_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w,
                           packed, Cin, Cout, k, stride, pad, tp, Mp);
        FV_HIP(hipGetLastError());
        return 0;
    }
    const Polyphase ph = polyphase(k, stride, pad);
    const int Mpad = pad_rows(Cout * stride);
    const int64_t total = (int64_t)Cin * ph.taps * Mpad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_convT_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                       packed, Cin, Cout, k, stride, pad, ph.dmin, ph.taps, Mpad);
    FV_HIP(hipGetLastError());
    return 0;
}

int64_t fv_packed_upsample_conv1d_floats(int Cout, int Cin, int k, int rate, int pad) {
    const Polyphase ph = upsample_phases(k, rate, pad);
    return (int64_t)Cin * ph.taps * pad_rows(Cout * rate);
}

int fv_pack_upsample_conv1d_weight(const float* w, float* packed, int Cout, int Cin, int k, int rate,
                                   int pad, void* stream) {
    if (int rc = check_conv_args(Cin, Cout, k, 1)) return rc;
    if (rate <= 0 || pad < 0) return fail(FV_ERR_INVALID_ARG, "upsample conv rate=%d pad=%d", rate, pad);
    const Polyphase ph = upsample_phases(k, rate, pad);
    const int Mpad = pad_rows(Cout * rate);
    const int64_t total = (int64_t)Cin * ph.taps * Mpad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_upconv_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed,
                       Cout, Cin, k, rate, pad, ph.dmin, ph.taps, Mpad);
    FV_HIP(hipGetLastError());
    return 0;
}

// ---- BasisSignalLayer + overlap_and_add (reference modules.py:255-267, :34-73) under its own name ----
// frames = weight W^T, out[hop f + j] += frames[f, j]: a ConvTranspose1d with Cin = C, Cout = 1, k = L, stride = hop = L / 2,
// pad = 0 whose weight [C, 1, L] is W^T -- the frame tensor [B, F, L] is never materialised.  fv_pack_basis transposes
// nn.Linear's W [L, C] into the scratch behind the packed image and packs it like any transposed-conv weight.
__global__ void transpose_basis_kernel(const float* __restrict__ W, float* __restrict__ wT, int L, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L * C) wT[(i % C) * L + i / C] = W[i];      // W[j][c] -> wT[c][0][j]
}

int64_t fv_packed_basis_floats(int L, int C) {
    if (L < 2 || L % 2 != 0 || C <= 0) return 0;
    return fv_packed_conv_transpose1d_floats(C, 1, L, L / 2, 0) + (int64_t)L * C;
}

int fv_pack_basis(const float* W, float* packed, int L, int C, void* stream) {
    if (!W || !packed) return fail(FV_ERR_INVALID_ARG, "pack_basis: null tensor");
    if (fv_packed_basis_floats(L, C) <= 0) return fail(FV_ERR_INVALID_ARG, "pack_basis: L=%d (even, >= 2) C=%d", L, C);
    float* wT = packed + fv_packed_conv_transpose1d_floats(C, 1, L, L / 2, 0);
    hipLaunchKernelGGL(transpose_basis_kernel, dim3((unsigned)((L * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, wT, L, C);
    FV_HIP(hipGetLastError());
    return fv_pack_conv_transpose1d_weight(wT, packed, C, 1, L, L / 2, 0, stream);
}

int64_t fv_packed_conv_transpose1d_split_floats(int Cin, int Cout, int k, int stride) {
    if ((Cin != 32 && Cin != 64 && Cin != 128 && Cin != 256 && Cin != 512) || stride < 2 || stride > 16 || k != 2 * stride ||
        Cout <= 0 || Cout * stride < 32)
        return 0;
    if (convtn_shape(Cin, Cout, k, stride)) return kTnPackedFloats;                   // its own kernel and layout (convtn_kernels.hpp)
    const int cc = Cin <= 64 ? 64 : 128;                                             // input channels per chunk (32: half of one)
    const int64_t row_tiles = (Cout * stride + 63) / 64;
    // row tiles x chunks x K steps x 8 KB, then one float per (padded) row: the inverse of its power-of-two prescale
    return row_tiles * ((Cin + cc - 1) / cc) * (2 * cc / 32) * 2048 + row_tiles * 64;
}

int fv_pack_conv_transpose1d_split_f16(const float* w, float* packed, int Cin, int Cout, int k, int stride, int* range_flag,
                                       void* stream) {
    if (!w || !packed) return fail(FV_ERR_INVALID_ARG, "pack_conv_transpose1d_split_f16: null tensor");
    if (convtn_shape(Cin, Cout, k, stride)) {
        hipLaunchKernelGGL(row_scale_kernel, dim3(32), dim3(64), 0, (hipStream_t)stream, w, (const float*)nullptr,
                           packed + (kTnPackedFloats - 32), 32, Cin, 2, Cout, stride);
        return launch_pack_convtn(w, packed, packed + (kTnPackedFloats - 32), range_flag, (hipStream_t)stream);
    }
    if (int rc = check_convt_split_args(Cin, Cout, k, stride, 0, 0)) return rc;
    const int rows = (Cout * stride + 63) / 64 * 64;
    const int64_t image = fv_packed_conv_transpose1d_split_floats(Cin, Cout, k, stride) - rows, total = image * 2;
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, w, (const float*)nullptr,
                       packed + image, rows, Cin, 2, Cout, stride);
    hipLaunchKernelGGL(pack_convth_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<_Float16*>(packed), packed + image, Cin, Cout, stride, range_flag);
    FV_HIP(hipGetLastError());
    return 0;
}

// ---- two-source 1x1 conv with split-f16 operands (convg_kernel) ----
int64_t fv_packed_conv1x1_2src_split_floats(int C) {
    if (C != 128 && C != 256 && C != 512) return 0;
    return (int64_t)(C / 64) * (2 * C / 128) * 4 * 2048 + C;  // row tiles x chunks x 4 K steps x 8 KB, + the rows' inverse prescales
}

int fv_pack_conv1x1_2src_split_f16(const float* w1, const float* w2, float* packed, int C, int* range_flag, void* stream) {
    if (!w1 || !w2 || !packed) return fail(FV_ERR_INVALID_ARG, "pack_conv1x1_2src_split_f16: null tensor");
    if (fv_packed_conv1x1_2src_split_floats(C) <= 0)
        return fail(FV_ERR_UNSUPPORTED, "pack_conv1x1_2src_split_f16: C = %d (128, 256 or 512)", C);
    const int64_t image = fv_packed_conv1x1_2src_split_floats(C) - C, total = image * 2;
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, w1, w2, packed + image, C, C, 1, 0, 0);
    hipLaunchKernelGGL(pack_convg_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w1, w2,
                       reinterpret_cast<_Float16*>(packed), packed + image, C, range_flag);
    FV_HIP(hipGetLastError());
    return 0;
}

// ---- MelGAN ResidualStack as one launch (convk_kernel) ----
int64_t fv_packed_residual_stack_floats(int C, int k) {
    if (!convk_shape(C, k, 1)) return 0;
    return (int64_t)5 * (C / 32) * C * 32 + 2 * C;       // 5 C / 32 stages of C x 128 bytes, + the rows' inverse prescales (conv1's, the 1x1 pair's)
}

int fv_pack_residual_stack_split_f16(const float* w_dilated, const float* w_pointwise, const float* w_skip, float* packed,
                                     int C, int k, int* range_flag, void* stream) {
    if (!w_dilated || !w_pointwise || !w_skip || !packed) return fail(FV_ERR_INVALID_ARG, "pack_residual_stack_split_f16: null tensor");
    const int64_t n = fv_packed_residual_stack_floats(C, k);
    if (n <= 0) return fail(FV_ERR_UNSUPPORTED, "pack_residual_stack_split_f16: C = %d, k = %d (32 / 64 / 128 / 256 channels, 3 taps)", C, k);
    float* inv = packed + (n - 2 * C);
    // the same row prescales as the two-launch form: conv1's rows over their 3 C weights, the 1x1 pair's over [W2 | Ws]
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, w_dilated, (const float*)nullptr,
                       inv, C, 3 * C, 0, 0, 0);
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, w_pointwise, w_skip, inv + C, C, C,
                       1, 0, 0);
    return launch_pack_convk(w_dilated, w_pointwise, w_skip, packed, C, range_flag, (hipStream_t)stream);
}

int64_t fv_packed_pair_floats(int C, int k) { return (int64_t)C * C * k; }

int fv_pack_pair_weight(const float* w, float* packed, int C, int k, void* stream) {
    if (!w || !packed) return fail(FV_ERR_INVALID_ARG, "pack_pair_weight: null tensor");
    if (C <= 0 || C % 16 != 0 || k <= 0) return fail(FV_ERR_INVALID_ARG, "pack_pair_weight: C=%d (multiple of 16) k=%d", C, k);
    const int64_t total = (int64_t)C * C * k;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_pair_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, C, k);
    FV_HIP(hipGetLastError());
    return 0;
}

int64_t fv_packed_pair_floats_ex(int C, int k, int prec) {
    if (prec != FV_PAIR_SPLIT_F16) return fv_packed_pair_floats(C, k);
    // (+ C: one float per row behind the image, the inverse of the row's power-of-two prescale -- row_scale_kernel)
    if (C == 64 || C == 128 || C == 256 || C == 512)
        return (int64_t)(C / 64) * k * (C / 32) * 2048 + C;                   // row tiles x (chunks x) K steps x 8 KB
    if (C != 16 && C != 32) return 0;
    const int tps = 32 / C;
    return (int64_t)((k + tps - 1) / tps) * (C / 16) * 512 + C;   // K steps x row halves x 2 split halves x 64 lanes x 16 bytes
}

int fv_pack_pair_weight_ex(const float* w, float* packed, int C, int k, int prec, int* range_flag, void* stream) {
    if (prec == FV_PAIR_F32) return fv_pack_pair_weight(w, packed, C, k, stream);
    if (prec != FV_PAIR_SPLIT_F16) return fail(FV_ERR_INVALID_ARG, "pack_pair_weight: unknown arithmetic %d", prec);
    if (!w || !packed) return fail(FV_ERR_INVALID_ARG, "pack_pair_weight: null tensor");
    if ((C != 16 && C != 32 && C != 64 && C != 128 && C != 256 && C != 512) || k <= 0)
        return fail(FV_ERR_INVALID_ARG, "pack_pair_weight: C=%d (16 ... 512, a power of two) k=%d", C, k);
    const int64_t image = fv_packed_pair_floats_ex(C, k, prec) - C, total = image * 2;
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, w, (const float*)nullptr,
                       packed + image, C, C * k, 0, 0, 0);
    if (C >= 64)
        hipLaunchKernelGGL(pack_convh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           w, reinterpret_cast<_Float16*>(packed), packed + image, C, k, range_flag);
    else
        hipLaunchKernelGGL(pack_pairh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           w, reinterpret_cast<_Float16*>(packed), packed + image, C, k, range_flag);
    FV_HIP(hipGetLastError());
    return 0;
}

// ---- a whole 16-channel MRF stage as one launch (csrc/mrfh_kernels.hpp) ----------------------------------------------
int64_t fv_packed_mrf_stage_floats(int C, const int* k) {
    const int dil[3] = {1, 3, 5};
    if (!k || !mrf_stage_shape(C, k, dil)) return 0;
    int64_t bytes = 0;
    for (int j = 0; j < 3; ++j) bytes += 3LL * mrf_block_bytes(C, k[j]);
    return bytes / 4;
}

int fv_pack_mrf_stage_split_f16(const float* const* w1, const float* const* w2, const float* const* b1,
                                const float* const* b2, float* packed, int C, const int* k, int* range_flag, void* stream) {
    const int64_t floats = fv_packed_mrf_stage_floats(C, k);
    if (floats == 0) return fail(FV_ERR_UNSUPPORTED, "pack_mrf_stage: shape not built (16 channels, taps 3 / 7 / 11)");
    if (!w1 || !w2 || !packed) return fail(FV_ERR_INVALID_ARG, "pack_mrf_stage: null tensor");
    hipStream_t const s = (hipStream_t)stream;
    FV_HIP(hipMemsetAsync(packed, 0, (size_t)floats * 4, s));          // absent biases and the blocks' padding: zeros
    char* at = reinterpret_cast<char*>(packed);
    for (int j = 0; j < 3; ++j)
        for (int q = 0; q < 3; ++q) {
            const int i = 3 * j + q;
            if (!w1[i] || !w2[i]) return fail(FV_ERR_INVALID_ARG, "pack_mrf_stage: null weight (pair %d)", i);
            const int wb = (mrf_block_bytes(C, k[j]) - 1024) / 2;       // bytes of one conv's image
            float* const tail = reinterpret_cast<float*>(at + 2 * wb);  // [b1 | b2 | s1 | s2]
            const float* const ws[2] = {w1[i], w2[i]};
            for (int c = 0; c < 2; ++c) {
                float* const inv = tail + (2 + c) * C;
                hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)C), dim3(64), 0, s, ws[c], (const float*)nullptr, inv, C,
                                   C * k[j], 0, 0, 0);
                const int64_t total = (int64_t)wb / 2;                  // halves of the image
                hipLaunchKernelGGL(pack_pairh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws[c],
                                   reinterpret_cast<_Float16*>(at + c * wb), inv, C, k[j], range_flag);
            }
            if (b1 && b1[i]) FV_HIP(hipMemcpyAsync(tail, b1[i], (size_t)C * 4, hipMemcpyDeviceToDevice, s));
            if (b2 && b2[i]) FV_HIP(hipMemcpyAsync(tail + C, b2[i], (size_t)C * 4, hipMemcpyDeviceToDevice, s));
            at += mrf_block_bytes(C, k[j]);
        }
    FV_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
