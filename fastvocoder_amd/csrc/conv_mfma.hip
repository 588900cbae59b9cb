// Implicit-GEMM 1-D convolution on the gfx950 fp32 matrix cores.
//
// One kernel family serves every wide convolution on the generator path
// (reference call sites: model/generator/modules.py:223-230 ResBlock1,
// :372-382 ResidualStack, hifigan.py:93-96 conv_pre / ConvTranspose1d,
// melgan.py:66-85, basis_melgan.py:72-97, modules.py:264-267 basis matmul+OLA):
//
//   Y[m, q] = sum_{ci, j} Wp[ci, j, m] * act(X[ci, q + j*dil - pad])
//
// M = Cout rows for Conv1d; for ConvTranspose1d the rows are the Cout*stride
// output phases of its polyphase form (fv_internal.h: polyphase()), so the
// same kernel runs it as a short dense conv and the epilogue interleaves the
// phases back into time ("pixel shuffle").  Arithmetic is exact fp32:
// v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 are bit-for-bit fmaf chains,
// which is what the 1e-4 end-to-end budget over ~80 chained layers needs
// (bf16/fp16 MFMA does not fit it; SURVEY.md section 7 "hard parts").
//
// Block = 256 threads = 4 wave64.  Per input-channel chunk the block stages
//   xs[ci_chunk][xw]      the activated input tile with its dilation halo
//   ws[ci_chunk*k][M_T]   the K-major weight slice (coalesced: Wp is K-major)
// in LDS, then every wave walks K = ci_chunk*k in steps of 2 (32x32x2) or 4
// (16x16x4) reading its A (weights) and B (shifted input window) operands with
// conflict-free ds_read_b32: lanes are consecutive in m for A and consecutive
// in time for B, so a dilated tap is just an address offset into the same row.
// The epilogue fuses bias, residual add, the MRF running sum / mean and
// tanh / ReLU, so no elementwise kernel exists on the path.
#include "fv_internal.h"

namespace fv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.f ? v : v * slope; }

__device__ __forceinline__ int reflect_idx(int i, int T) {
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    // columns staged only for alignment slack or tile overhang may still fall
    // outside; they feed masked outputs only, so clamp instead of faulting
    return min(max(i, 0), T - 1);
}

// Stage the activated input tile for channels [ci0, ci0+ci_chunk) and global
// times [tA, tA + 4*ncol4) into xs (row stride p.xw).
__device__ __forceinline__ void stage_input(const ConvParams& p, float* xs, int b, int ci0,
                                            int tA, int ncol4, int tid) {
    const int total = p.ci_chunk * ncol4;
    for (int idx = tid; idx < total; idx += 256) {
        const int row = idx / ncol4;
        const int c4 = idx - row * ncol4;
        const int ci = ci0 + row;
        const int t = tA + 4 * c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci < p.Cin) {
            const float* xr = p.x + ((size_t)b * p.Cin + ci) * (size_t)p.Tin;
            if (p.vec_ok && t >= 0 && t + 3 < p.Tin) {
                v = *reinterpret_cast<const float4*>(xr + t);
            } else if (p.pad_mode == FV_PAD_REFLECT) {
                v.x = xr[reflect_idx(t, p.Tin)];
                v.y = xr[reflect_idx(t + 1, p.Tin)];
                v.z = xr[reflect_idx(t + 2, p.Tin)];
                v.w = xr[reflect_idx(t + 3, p.Tin)];
            } else {
                if (t >= 0 && t < p.Tin) v.x = xr[t];
                if (t + 1 >= 0 && t + 1 < p.Tin) v.y = xr[t + 1];
                if (t + 2 >= 0 && t + 2 < p.Tin) v.z = xr[t + 2];
                if (t + 3 >= 0 && t + 3 < p.Tin) v.w = xr[t + 3];
            }
            if (p.pre_slope != 1.f) {
                v.x = lrelu(v.x, p.pre_slope);
                v.y = lrelu(v.y, p.pre_slope);
                v.z = lrelu(v.z, p.pre_slope);
                v.w = lrelu(v.w, p.pre_slope);
            }
        }
        *reinterpret_cast<float4*>(xs + row * p.xw + 4 * c4) = v;
    }
}

// Stage the weight slice rows [ci0*k, (ci0+ci_chunk)*k) x columns [m0, m0+M_T).
template <int M_T>
__device__ __forceinline__ void stage_weights(const ConvParams& p, float* ws, int ci0, int m0,
                                              int tid) {
    constexpr int C4 = M_T / 4;
    const int nrow = p.ci_chunk * p.k;
    const int krows = p.Cin * p.k;
    const int row0 = ci0 * p.k;
    for (int idx = tid; idx < nrow * C4; idx += 256) {
        const int row = idx / C4;
        const int c = idx - row * C4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < krows)
            v = *reinterpret_cast<const float4*>(p.wp + (size_t)(row0 + row) * p.Mpad + m0 + 4 * c);
        *reinterpret_cast<float4*>(ws + row * M_T + 4 * c) = v;
    }
}

// One output element through the fused epilogue.
__device__ __forceinline__ void epilogue_store(const ConvParams& p, int b, int m, int q, float v) {
    if (m >= p.M || q >= p.Tq) return;
    int co = m, t = q;
    if (p.ups != 1) {
        co = m / p.ups;
        t = q * p.ups + (m - co * p.ups);
        if (t >= p.Tout) return;
    }
    const size_t o = ((size_t)b * p.Cout + co) * (size_t)p.Tout + t;
    if (p.bias) v += p.bias[co];
    if (p.res) v += p.res[o];
    if (p.acc_in) v = p.acc_in[o] + v;
    if (p.out_div != 1.f) v = v / p.out_div;
    if (p.post == FV_POST_TANH) v = tanhf(v);
    else if (p.post == FV_POST_RELU) v = fmaxf(v, 0.f);
    p.y[o] = v;
}

// XCD-aware tile order: the dispatcher places linear block id b on XCD b % 8,
// so give each XCD a contiguous run of time tiles (neighbouring tiles share
// halo columns and, for Cout > M_T, the same input tile) -- speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, within = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + within;
}

// ---------------------------------------------------------------------------
// 32x32x2 variant: block tile (32*WM) x (32*NR*WN), WM*WN = 4 waves.
// KT > 0 fixes the tap count at compile time (full unroll of the tap loop).
// ---------------------------------------------------------------------------
template <int WM, int WN, int NR, int KT>
__global__ __launch_bounds__(256) void conv_mfma32_kernel(ConvParams p) {
    constexpr int M_T = 32 * WM;
    constexpr int N_T = 32 * NR * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int k = KT > 0 ? KT : p.k;
    float* xs = smem;
    float* ws = smem + p.ci_chunk * p.xw;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wave_m = wave / WN, wave_n = wave % WN;

    const int n_tiles = (p.Tq + N_T - 1) / N_T;
    const int m_tiles = p.Mpad / M_T;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = lin % m_tiles, nt = lin / m_tiles;
    const int b = blockIdx.y;
    const int m0 = mt * M_T, t0 = nt * N_T;
    (void)n_tiles;

    const int halo = (k - 1) * p.dil;
    const int tstart = t0 - p.pad;
    const int aoff = ((tstart % 4) + 4) % 4;
    const int tA = tstart - aoff;
    const int ncol4 = (N_T + halo + aoff + 3) >> 2;

    f32x16 acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;

    const float* wsA = ws + wave_m * 32 + l31;
    const float* xsB = xs + aoff + wave_n * (32 * NR) + l31;

    for (int ci0 = 0; ci0 < p.Cin; ci0 += p.ci_chunk) {
        stage_input(p, xs, b, ci0, tA, ncol4, tid);
        stage_weights<M_T>(p, ws, ci0, m0, tid);
        __syncthreads();
        for (int c2 = 0; c2 < p.ci_chunk; c2 += 2) {
            const int ci = c2 + hi;
            const float* pa = wsA + ci * k * M_T;
            const float* pb = xsB + ci * p.xw;
            if constexpr (KT > 0) {
#pragma unroll
                for (int tap = 0; tap < KT; ++tap) {
                    const float a = pa[tap * M_T];
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[tap * p.dil + r * 32],
                                                                      acc[r], 0, 0, 0);
                }
            } else {
                for (int tap = 0; tap < k; ++tap) {
                    const float a = pa[tap * M_T];
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[tap * p.dil + r * 32],
                                                                      acc[r], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // C/D map of 32x32: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int q = t0 + wave_n * (32 * NR) + r * 32 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + wave_m * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
            epilogue_store(p, b, m, q, acc[r][i]);
        }
    }
}

// ---------------------------------------------------------------------------
// 16x16x4 variant for M <= 16 rows (the C = 16 stage, the basis matmul+OLA):
// block tile 16 x (16*NR*4), four waves side by side in time.
// ---------------------------------------------------------------------------
template <int NR, int KT>
__global__ __launch_bounds__(256) void conv_mfma16_kernel(ConvParams p) {
    constexpr int M_T = 16;
    constexpr int N_T = 16 * NR * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int k = KT > 0 ? KT : p.k;
    float* xs = smem;
    float* ws = smem + p.ci_chunk * p.xw;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;

    const int m_tiles = p.Mpad / M_T;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = lin % m_tiles, nt = lin / m_tiles;
    const int b = blockIdx.y;
    const int m0 = mt * M_T, t0 = nt * N_T;

    const int halo = (k - 1) * p.dil;
    const int tstart = t0 - p.pad;
    const int aoff = ((tstart % 4) + 4) % 4;
    const int tA = tstart - aoff;
    const int ncol4 = (N_T + halo + aoff + 3) >> 2;

    f32x4 acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[r][i] = 0.f;

    const float* wsA = ws + l15;
    const float* xsB = xs + aoff + wave * (16 * NR) + l15;

    for (int ci0 = 0; ci0 < p.Cin; ci0 += p.ci_chunk) {
        stage_input(p, xs, b, ci0, tA, ncol4, tid);
        stage_weights<M_T>(p, ws, ci0, m0, tid);
        __syncthreads();
        for (int c4 = 0; c4 < p.ci_chunk; c4 += 4) {
            const int ci = c4 + kq;
            const float* pa = wsA + ci * k * M_T;
            const float* pb = xsB + ci * p.xw;
            if constexpr (KT > 0) {
#pragma unroll
                for (int tap = 0; tap < KT; ++tap) {
                    const float a = pa[tap * M_T];
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pb[tap * p.dil + r * 16],
                                                                      acc[r], 0, 0, 0);
                }
            } else {
                for (int tap = 0; tap < k; ++tap) {
                    const float a = pa[tap * M_T];
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pb[tap * p.dil + r * 16],
                                                                      acc[r], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // C/D map of 16x16: col = lane & 15, row = 4*(lane >> 4) + reg
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int q = t0 + wave * (16 * NR) + r * 16 + l15;
#pragma unroll
        for (int i = 0; i < 4; ++i) epilogue_store(p, b, m0 + 4 * kq + i, q, acc[r][i]);
    }
}

// ---------------------------------------------------------------------------
// Narrow-output variant (Cout <= 4: conv_post hifigan.py:105 /
// multiband_hifigan.py:114, LastLayer modules.py:85-89).  HBM-bound
// (3.3 FLOP/B): one thread per output time step, weights broadcast from LDS,
// the input tile staged exactly like the MFMA variants.
// ---------------------------------------------------------------------------
template <int MO>
__global__ __launch_bounds__(256) void conv_narrow_kernel(ConvParams p) {
    constexpr int N_T = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + p.ci_chunk * p.xw;  // [ci_chunk*k][16]
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = xcd_remap(blockIdx.x, gridDim.x) * N_T;
    const int k = p.k;
    const int halo = (k - 1) * p.dil;
    const int tstart = t0 - p.pad;
    const int aoff = ((tstart % 4) + 4) % 4;
    const int tA = tstart - aoff;
    const int ncol4 = (N_T + halo + aoff + 3) >> 2;
    float acc[MO];
#pragma unroll
    for (int m = 0; m < MO; ++m) acc[m] = 0.f;
    for (int ci0 = 0; ci0 < p.Cin; ci0 += p.ci_chunk) {
        stage_input(p, xs, b, ci0, tA, ncol4, tid);
        stage_weights<16>(p, ws, ci0, 0, tid);
        __syncthreads();
        const float* pb = xs + aoff + tid;
        for (int ci = 0; ci < p.ci_chunk; ++ci) {
            for (int tap = 0; tap < k; ++tap) {
                const float xv = pb[ci * p.xw + tap * p.dil];
                const float* wr = ws + (ci * k + tap) * 16;
#pragma unroll
                for (int m = 0; m < MO; ++m) acc[m] = fmaf(wr[m], xv, acc[m]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < MO; ++m) epilogue_store(p, b, m, t0 + tid, acc[m]);
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
namespace {

template <typename K>
int launch_one(K kernel, const ConvParams& p, int m_t, int n_t, size_t lds, hipStream_t s) {
    const int n_tiles = (p.Tq + n_t - 1) / n_t;
    const int m_tiles = p.Mpad / m_t;
    dim3 grid(n_tiles * m_tiles, p.B), block(256);
    if (lds > 64 * 1024) {
        FV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}

template <int WM, int WN, int NR>
int launch32(const ConvParams& p, size_t lds, hipStream_t s) {
    constexpr int m_t = 32 * WM, n_t = 32 * NR * WN;
    switch (p.k) {
        case 1: return launch_one(conv_mfma32_kernel<WM, WN, NR, 1>, p, m_t, n_t, lds, s);
        case 3: return launch_one(conv_mfma32_kernel<WM, WN, NR, 3>, p, m_t, n_t, lds, s);
        case 7: return launch_one(conv_mfma32_kernel<WM, WN, NR, 7>, p, m_t, n_t, lds, s);
        case 11: return launch_one(conv_mfma32_kernel<WM, WN, NR, 11>, p, m_t, n_t, lds, s);
        default: return launch_one(conv_mfma32_kernel<WM, WN, NR, 0>, p, m_t, n_t, lds, s);
    }
}

template <int NR>
int launch16(const ConvParams& p, size_t lds, hipStream_t s) {
    constexpr int n_t = 16 * NR * 4;
    switch (p.k) {
        case 3: return launch_one(conv_mfma16_kernel<NR, 3>, p, 16, n_t, lds, s);
        case 7: return launch_one(conv_mfma16_kernel<NR, 7>, p, 16, n_t, lds, s);
        case 11: return launch_one(conv_mfma16_kernel<NR, 11>, p, 16, n_t, lds, s);
        default: return launch_one(conv_mfma16_kernel<NR, 0>, p, 16, n_t, lds, s);
    }
}

// LDS row stride of the input tile: room for the tile, its halo and the
// alignment slack, a multiple of 4 floats (float4 staging writes) and, for the
// 16x16x4 variant, = 16 (mod 32) so the two k-rows a half-wave reads sit on
// disjoint banks.
int row_stride(int n_t, int halo, bool mfma16) {
    int xw = round_up(n_t + halo + 3, 4) + 4;
    if (mfma16) {
        while (xw % 32 != 16) xw += 4;
    }
    return xw;
}

// Input channels staged per LDS round: enough K per barrier pair to amortise
// it (>= ~64 MFMA k-steps), bounded by Cin and by ~48 KiB of LDS.
int pick_ci_chunk(const ConvParams& p, int m_t, int xw, int step) {
    int want = (96 + p.k - 1) / p.k;             // ci_chunk * k ~ 96
    want = round_up(want < step ? step : want, step);
    int cin_pad = round_up(p.Cin, step);
    if (want > cin_pad) want = cin_pad;
    while (want > step && (size_t)want * (xw + p.k * m_t) * 4 > 48 * 1024) want -= step;
    return want;
}

}  // namespace

int launch_conv(ConvParams p, hipStream_t s) {
    if (p.B <= 0 || p.Tq <= 0 || p.Cin <= 0 || p.M <= 0) return 0;
    if (p.k < 1 || p.dil < 1) return fail(FV_ERR_INVALID_ARG, "conv: k=%d dil=%d", p.k, p.dil);
    if (p.pad_mode == FV_PAD_REFLECT && p.pad >= p.Tin)
        return fail(FV_ERR_INVALID_ARG, "reflection pad %d needs an input longer than it (T=%d)",
                    p.pad, p.Tin);
    p.vec_ok = (p.Tin % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
    const int halo = (p.k - 1) * p.dil;
    const double flops = 2.0 * p.B * (double)p.M * p.Tq * p.Cin * p.k;
    const double bytes = 4.0 * ((double)p.B * p.Cin * p.Tin + (double)p.B * p.Cout * p.Tout *
                                (1 + (p.res != nullptr) + (p.acc_in != nullptr)) +
                                (double)p.Cin * p.k * p.M);
    int rc, kind;
    profile_begin(s);
    if (p.ups == 1 && p.M <= 4) {
        kind = FV_KERNEL_CONV_NARROW;
        p.xw = row_stride(256, halo, false);
        p.ci_chunk = pick_ci_chunk(p, 16, p.xw, 1);
        const size_t lds = (size_t)p.ci_chunk * (p.xw + p.k * 16) * 4;
        if (p.M == 1) rc = launch_one(conv_narrow_kernel<1>, p, p.Mpad, 256, lds, s);
        else if (p.M == 2) rc = launch_one(conv_narrow_kernel<2>, p, p.Mpad, 256, lds, s);
        else rc = launch_one(conv_narrow_kernel<4>, p, p.Mpad, 256, lds, s);
    } else if (p.Mpad == 16) {
        kind = FV_KERNEL_CONV_MFMA16;
        const bool big = (long)p.B * ((p.Tq + 255) / 256) >= 512;
        const int n_t = big ? 256 : 128;
        p.xw = row_stride(n_t, halo, true);
        p.ci_chunk = pick_ci_chunk(p, 16, p.xw, 4);
        const size_t lds = (size_t)p.ci_chunk * (p.xw + p.k * 16) * 4;
        rc = big ? launch16<4>(p, lds, s) : launch16<2>(p, lds, s);
    } else if (p.Mpad % 64 == 0) {
        kind = FV_KERNEL_CONV_MFMA32;
        const long blocks128 = (long)p.B * (p.Mpad / 64) * ((p.Tq + 127) / 128);
        const bool big = blocks128 >= 512;
        const int n_t = big ? 128 : 64;
        p.xw = row_stride(n_t, halo, false);
        p.ci_chunk = pick_ci_chunk(p, 64, p.xw, 2);
        const size_t lds = (size_t)p.ci_chunk * (p.xw + p.k * 64) * 4;
        rc = big ? launch32<2, 2, 2>(p, lds, s) : launch32<2, 2, 1>(p, lds, s);
    } else {
        kind = FV_KERNEL_CONV_MFMA32;
        const long blocks256 = (long)p.B * (p.Mpad / 32) * ((p.Tq + 255) / 256);
        const bool big = blocks256 >= 512;
        const int n_t = big ? 256 : 128;
        p.xw = row_stride(n_t, halo, false);
        p.ci_chunk = pick_ci_chunk(p, 32, p.xw, 2);
        const size_t lds = (size_t)p.ci_chunk * (p.xw + p.k * 32) * 4;
        rc = big ? launch32<1, 4, 2>(p, lds, s) : launch32<1, 4, 1>(p, lds, s);
    }
    profile_end(s, kind, flops, bytes);
    return rc;
}

}  // namespace fv
