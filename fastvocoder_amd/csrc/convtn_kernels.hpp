// ConvTranspose1d of the LAST upsampler of HiFi-GAN light -- 32 -> 16 channels, kernel 4, stride 2, padding 1
// (reference model/generator/hifigan.py:39-46, :95-96 with conf/hifigan/light.yaml: upsample_rates[-1] = 2) -- with
// split-f16 operands (pairh_kernels.hpp: v = h1 + h2 / 2048, three v_mfma_f32_16x16x32_f16 terms per product):
//
//     x   = lrelu( ((r0 + r1) + r2) / div, slope )           (r1, r2 optional: the MRF merge of the stage in front,
//                                                              hifigan.py:99-103, formed here -- convh_kernels.hpp merge_window)
//     y[co][2 u + ph - 1] = bias[co] + sum_ci  w[ci][co][2 + ph] x[ci][u - 1]  +  w[ci][co][ph] x[ci][u]
//
// The layer is 30 MB of traffic and 0.5 GFLOP: HBM-bound, and at batch 1 latency-bound.  The general split-f16 transposed conv
// (convt_kernel: 64-row tiles, chunks of 64+ input channels, weights streamed through an LDS ring) pads both M and K to 64
// for it and took 29 us, the fp32-MFMA kernel 18 us.  Here the GEMM is exactly M = 32 rows (output channel, phase) x K = 64
// (2 taps x 32 channels) -- two row sixteenths, two K steps: the whole weight matrix is 32 VGPRs of A operands per lane, read
// once from L2; a block converts a window of 257 input columns into the split image (one barrier) and every wave runs
// 48 MFMAs on its 64 columns.  One tile per block, no ring, no persistent loop.
#pragma once
#include "pairh_kernels.hpp"

namespace fv {

struct ConvTnParams {
    const float* x;       // [B, 32, T]   (r0 when add1 is set)
    const float* add1;    // [B, 32, T] or null
    const float* add2;    // [B, 32, T] or null
    const float* w;       // packed (pack_convtn_kernel): [K step][row sixteenth][split half][lane][8 f16], then 32 inverse row prescales
    const float* bias;    // [16] or null
    float* y;             // [B, 16, Tout]
    float* y_act;         // optional activated twin, or null
    int B, T, Tout, n_tiles;
    float slope, act_slope, out_div;
    int* guard;
};

constexpr int kTnCin = 32, kTnCout = 16, kTnCols = 256, kTnRows = 260, kTnRP = 272;   // window rows u0 - 1 ... u0 + 258
constexpr int kTnHalf = (kTnCin / 8) * kTnRP * 16;                                    // bytes of one split half of the image
constexpr int kTnImageFloats = 2 * 2 * 2 * 64 * 8 / 2;                                // packed weights, in floats (8 KB)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void convtn_kernel(ConvTnParams p) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    __shared__ __attribute__((aligned(16))) char ximg[2 * kTnHalf];
    __shared__ unsigned lowbits[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x / p.n_tiles, tile = (int)blockIdx.x - b * p.n_tiles;
    const int u0 = tile * kTnCols;
    const size_t xoff = (size_t)b * kTnCin * (size_t)p.T;
    const unsigned xbytes = (unsigned)kTnCin * (unsigned)p.T * 4u, t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(p.x + xoff, xbytes);
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(p.add1 ? p.add1 + xoff : p.x, p.add1 ? xbytes : 0u);
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(p.add2 ? p.add2 + xoff : p.x, p.add2 ? xbytes : 0u);
    // ---- window: task = (row, block of 8 channels); 1040 tasks on 256 threads: four full rounds + 16 left over ----
    constexpr int TASKS = kTnRows * (kTnCin / 8), NR = (TASKS + 255) / 256;
    float v[NR][8], a1[NR][8], a2[NR][8];
    const bool merge = p.add1 != nullptr;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int idx = tid + q * 256;
        const int cb = idx / kTnRows, row = idx - cb * kTnRows;
        const int t = u0 - 1 + row;
        const unsigned voff = idx < TASKS && t >= 0 && t < p.T ? (unsigned)(cb * 8 * p.T + t) * 4u : kOutOfRange;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[q][j] = buffer_load1s(r0, voff, (unsigned)j * t4);
        if (merge) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a1[q][j] = buffer_load1s(r1, voff, (unsigned)j * t4);
#pragma unroll
            for (int j = 0; j < 8; ++j) a2[q][j] = buffer_load1s(r2, voff, (unsigned)j * t4);
        }
    }
    pair_wait_vm0();
    float lowm = 0.f;
    const float rcp = div_rcp(p.out_div);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int idx = tid + q * 256;
        const int cb = idx / kTnRows, row = idx - cb * kTnRows;
        if (idx < TASKS) {
            F16x8Parts h1, h2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 raw = {v[q][2 * j], v[q][2 * j + 1]};
                if (merge) {
                    raw = (raw + f32x2{a1[q][2 * j], a1[q][2 * j + 1]}) + f32x2{a2[q][2 * j], a2[q][2 * j + 1]};
                    if (p.out_div != 1.f) {
                        raw.x = rcp != 0.f ? div_exact(raw.x, p.out_div, rcp) : raw.x / p.out_div;
                        raw.y = rcp != 0.f ? div_exact(raw.y, p.out_div, rcp) : raw.y / p.out_div;
                    }
                }
                const f32x2 a = split_act2(raw, p.slope);
                lowm = low_max3(lowm, a.x, a.y);
                split2(a, h1.p[j], h2.p[j]);
            }
            *reinterpret_cast<f16x8*>(ximg + (cb * kTnRP + row) * 16) = __builtin_bit_cast(f16x8, h1);
            *reinterpret_cast<f16x8*>(ximg + (cb * kTnRP + row) * 16 + kTnHalf) = __builtin_bit_cast(f16x8, h2);
        }
    }
    // ---- A operands: the whole weight matrix, 16 bytes per lane and (K step, row sixteenth, split half) ----
    f16x8 A[2][2][2];
    {
        const f16x8* wl = reinterpret_cast<const f16x8*>(p.w) + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 2; ++e) A[s][h][e] = wl[((s * 2 + h) * 2 + e) * 64];
    }
    LowGuard low;
    low_note(low, 0, lowm);
    if (p.guard && lane == 0) lowbits[wave] = low.bits;
    pair_barrier();                                      // the image is complete
    if (p.guard && tid == 0) {
        const unsigned all = lowbits[0] | lowbits[1] | lowbits[2] | lowbits[3];
        if ((all & 2u) && !(all & 1u)) guard_raise_low(p.guard);     // operands not all zero and all below kSplitLow (pairh_kernels.hpp)
    }
    // ---- 64 columns per wave: four fragments x two row sixteenths x two K steps x three split terms ----
    const int n = lane & 15, kb = lane >> 4;
    const int col0 = wave * 64 + n;
    LdsCF* const bb = lds_opaque(reinterpret_cast<const float*>(ximg + (kb * kTnRP + col0) * 16));
    f32x4 hi[2][4], lo[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < 4; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {                        // K step 0: x[u - 1] (window row col), step 1: x[u] (row col + 1)
        f16x8 B1[4], B2[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            B1[f] = *reinterpret_cast<LdsH8*>(bb + (s * 16 + f * 256) / 4);
            B2[f] = *reinterpret_cast<LdsH8*>(bb + (s * 16 + f * 256 + kTnHalf) / 4);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < 4; ++f) hi[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][h][0], B1[f], hi[h][f], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < 4; ++f) lo[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][h][0], B2[f], lo[h][f], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < 4; ++f) lo[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][h][1], B1[f], lo[h][f], 0, 0, 0);
    }
    // ---- epilogue: rows m = 16 h + 4 kb + i = (output channel m / 2, phase m % 2): a lane's rows are two channels x two
    // consecutive samples n = 2 u + phase - 1: two 8-byte stores per fragment
    const size_t yoff = (size_t)b * kTnCout * (size_t)p.Tout;
    const unsigned ybytes = (unsigned)kTnCout * (unsigned)p.Tout * 4u;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + yoff, ybytes);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.y_act ? p.y_act + yoff : p.y, p.y_act ? ybytes : 0u);
    const float* const inv = p.w + kTnImageFloats;
    f32x2 bad2 = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m0 = 16 * h + 4 * kb, co0 = m0 >> 1;
        const f32x2 s01 = {inv[m0], inv[m0 + 1]}, s23 = {inv[m0 + 2], inv[m0 + 3]};
        const float bA = p.bias ? p.bias[co0] : 0.f, bB = p.bias ? p.bias[co0 + 1] : 0.f;
        const f32x2 c = {kSplitInv, kSplitInv};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int u = u0 + col0 + f * 16, nn = 2 * u - 1;
            f32x2 y0 = fma2(fma2(f32x2{lo[h][f][0], lo[h][f][1]}, c, f32x2{hi[h][f][0], hi[h][f][1]}), s01, f32x2{bA, bA});
            f32x2 y1 = fma2(fma2(f32x2{lo[h][f][2], lo[h][f][3]}, c, f32x2{hi[h][f][2], hi[h][f][3]}), s23, f32x2{bB, bB});
            f32x2 z0 = y0, z1 = y1;
            if (p.act_slope != 1.f) {
                z0 = split_act2(y0, p.act_slope);
                z1 = split_act2(y1, p.act_slope);
                if (!p.y_act) {
                    y0 = z0;
                    y1 = z1;
                }
            }
            bad2 = fma2(y0, f32x2{0.f, 0.f}, bad2);
            bad2 = fma2(y1, f32x2{0.f, 0.f}, bad2);
            // samples nn, nn + 1 of channels co0, co0 + 1 (nn = -1 at u = 0: that sample does not exist)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x2 yv = e ? y1 : y0, zv = e ? z1 : z0;
                const unsigned base = (unsigned)((co0 + e) * p.Tout) * 4u;
                const unsigned o0 = nn >= 0 && nn < p.Tout ? base + (unsigned)nn * 4u : kOutOfRange;
                const unsigned o1 = nn + 1 < p.Tout ? base + (unsigned)(nn + 1) * 4u : kOutOfRange;
                buffer_store1(ry, o0, yv.x);
                buffer_store1(ry, o1, yv.y);
                if (p.y_act) {
                    buffer_store1(ra, o0, zv.x);
                    buffer_store1(ra, o1, zv.y);
                }
            }
        }
    }
    if (p.guard) {
        const float bad = bad2.x + bad2.y;
        if (bad != bad) guard_raise_high(p.guard);
    }
}

}  // namespace fv
