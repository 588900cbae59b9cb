// A whole MRF stage of HiFi-GAN at 16 channels in ONE launch, split-f16 operands on the gfx950 matrix cores:
//
//     r_j = ResBlock1_j(x)   (three pairs  x <- x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2,  dilations D0, D1, D2;
//                             reference model/generator/modules.py:223-230),   j = 0, 1, 2  (taps k_j in {3, 7, 11})
//     y   = ((r_0 + r_1) + r_2) / 3                                          (reference model/generator/hifigan.py:97-103)
//     FOLD: out = tanh( conv_post( lrelu(y, 0.01) ) )                         (hifigan.py:104-106) instead of y
//
// Until round 4 a stage was four dependent launches of fused PAIRS (pairh_kernels.hpp): every pair position read its
// 16-channel tensor from memory and wrote it back, converted the raw window into the split image (a transposing pass of its
// own), reloaded the residual and staged its weights again -- 293 MB of external traffic for a stage whose input is 15 MB,
// and ~14 us of fixed cost per launch.  Here a tile carries its columns through all nine pairs:
//   * the running fp32 x of a column lives in the REGISTERS of the lane that owns the column's D fragments -- the column ->
//     lane map is the same for all 18 convs (images are time-aligned: image row FM + c is window column c for x and for the
//     intermediate alike), so the residual add is register + accumulator and the activated, split input of the next conv is
//     written straight from the D-fragment layout (ds_write_b64, pairh_kernels.hpp's image layout): no conversion pass, no
//     residual loads, no stores between the pairs;
//   * the stage input x0 is loaded ONCE per tile, global -> registers in D-fragment order, and kept for the three
//     ResBlocks; the MRF sum accumulates in registers in the reference's association; one store per tile at the end (FOLD:
//     4 bytes per sample instead of 64);
//   * weights: the stage's 18 packed convs (153 KB, fv_pack_mrf_stage_split_f16: per pair [conv1 image | conv2 image |
//     b1 | b2 | 1/prescale of conv1 | conv2]) stream L2 -> LDS by LDS-DMA one pair ahead, double-buffered;
//   * ONE-SIDED halo.  A conv needs (k-1) d / 2 input columns either side.  On the right a tile simply computes W columns of
//     which the last HALO (the sum of the reaches of the longest ResBlock: 60 for 11 taps, dilations 1, 3, 5) are not final;
//     on the left nothing is recomputed: a block walks its columns left to right, and every conv's input image keeps the
//     last (k-1) d / 2 columns in front of the next tile's first column in a small LDS history slot (saved while the image
//     is complete, copied back in front of the image while it is rewritten; 16 bytes per (row, 8-channel block, split
//     half)).  Only the first tile of a block's run starts HALO columns early.  W = 576: 516 final columns per tile
//     (two-sided: 456), and at T = 240 000, batch 1, every one of 256 blocks has exactly two tiles.
// Per pair two barriers, as in pairh_kernel; per tile 18 (FOLD: 20).  Numerics: the operations per output element and
// their order are those of the pair kernels (split_mid4 / combine4 / div_exact), so a stage run here is bit-identical to
// the four-launch form (tests/test_gpu_pairs.py).
// Values outside a tile's final region ("garbage" columns at the right edge, and left of a run's first tile) are computed
// from real, zero-padded data or zeroed margins / history, so they are finite whenever the final values are; they never
// reach a final column except through the zero pad tap of an odd tap count (0 x finite = 0).
#pragma once
#include "pairh_kernels.hpp"

namespace fv {

template <int NF_, int NG_>
struct MrfTile {
    static constexpr int NF = NF_, NG = NG_, C = 16;
    static constexpr int NT = 64 * NG;
    static constexpr int W = 16 * NF * NG;              // window columns of a tile
    static constexpr int FM = 32;                       // margin rows in front of / behind the window
    static constexpr int RP = W + 2 * FM;               // rows of a channel block (a multiple of 16: bank rows)
    static constexpr int HALF = 2 * RP * 16;            // bytes of one split half (two 8-channel blocks)
    static constexpr int IMG = 2 * HALF;
    static constexpr int WSLOT = 2 * 6 * 2048 + 1024;   // bytes of a weight slot: the 11-tap pair + its bias block
    static constexpr int HSLOT = (4 * 25 + 4 * 5) * 16; // history of one pair: conv1's <= 25 rows, conv2's <= 5 rows
    static constexpr int HX = 0, HM = 4 * 25 * 16;
    static constexpr int SBW = 16;                      // FOLD: the activated fp32 tile is [column][16 channels], channel-minor
    static constexpr int OFF_W = 0, OFF_X = 2 * WSLOT, OFF_M = OFF_X + IMG, OFF_H = OFF_M + IMG, OFF_S = OFF_H + 9 * HSLOT;
    static constexpr int OFF_F = OFF_S + 256;           // FOLD: the output conv's weights [16][8] (7 taps + pad) and bias
    static constexpr int LDS = OFF_F + 576;
    static_assert(RP % 16 == 0, "whole bank rows");
    static_assert(W * SBW * 4 <= IMG, "the folded output conv's tile lies over the intermediate image");
    static_assert(LDS <= 160 * 1024, "LDS");
};

// geometry of one conv for pairh_mma (MH, NF, KS, AREG) and the pack layout
template <int NF_, int KT_>
struct MrfGeom {
    static constexpr int MH = 1, NF = NF_, KT = KT_;
    static constexpr int KS = (KT + 1) / 2;             // K steps: two taps x 16 channels each, an odd count padded
    static constexpr int WB = KS * 2048;                // bytes of one conv's packed image
    static constexpr bool AREG = NF_ >= 4;
    static constexpr int P2 = (KT - 1) / 2;
};

__device__ __forceinline__ void mrf_stamp(const MrfParams& p, int nw, int wave, int lane, int it, int ev) {
#ifdef FV_PAIR_TRACE
    if (p.trace && (blockIdx.x & 63) == 0 && blockIdx.x < 256 && it < 3 && lane == 0)
        p.trace[(((size_t)(blockIdx.x >> 6) * nw + wave) * 3 + it) * 64 + ev] = __builtin_amdgcn_s_memtime();
#endif
}

// What one lane needs to know about its place in the block.  Offsets, not pointers: every pair launders them (mrf_pair), so
// that hipcc forms the 18 convs' operand addresses where they are used -- left alone it hoists every one of them (and the
// history copies' addresses) out of the tile loop and spills for it [first build: 135 VGPRs spilled].
template <class TL>
struct MrfLane {
    int tid, lane, wave;
    int colw;                // window column of the lane's fragment 0
    int row0;                // first of its four channels (D fragment)
    int tap16;               // 16 x its tap inside a K step (0 / 16)
    int rdoff;               // B operand: byte offset of row FM + colw of its channel block inside an image
    int wroff;               // D fragment: byte offset of its half block entry of row FM + colw inside an image
    int cpslot;              // history copies: 16 x entry u = tid & 127 (u = 4 row + part, part = 2 split half + channel block)
    int cpimg;               // ... byte offset of (row, part) inside an image, relative to row 0
    char* sm;                // dynamic LDS
};

// the LDS-DMA of one pair's block: pieces of 1 KB, waves round-robin (at most 25 pieces: no loop -- hipcc drains vmcnt in
// front of a loop that holds an LDS-DMA)
template <int NG>
__device__ __forceinline__ void mrf_dma(__amdgpu_buffer_rsrc_t rb, float* dst, unsigned off, int pieces, int wave, int lane) {
    constexpr int R = (25 + NG - 1) / NG;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = wave + r * NG;
        if (i < pieces) dma16(rb, dst + i * 256, off + (unsigned)(i * 1024 + lane * 16));
    }
}

// ---- one conv phase: hi += a1 b1, lo += a1 b2 + a2 b1 over the KS steps (pairh_mma's sums in pairh_mma's order per
// accumulator: identical bits), with the operand queues cut for three fragments at 3 waves per SIMD: per step the a1 b2
// group runs first, so the b2 registers take the NEXT step's b2 while the other two groups run (single-buffered); b1 and
// a1 are double-buffered, a2 (used by the last group only) single: 48 registers where pairh_mma's two-step queue takes 64.
template <class G, int TAPB, int HALF>
__device__ __forceinline__ void mrf_mma(const float* wl, const char* img, f32x4 (&hi)[G::NF], f32x4 (&lo)[G::NF], int lane) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    constexpr int NF = G::NF, KS = G::KS;
    LdsCF* base = lds_opaque(reinterpret_cast<const float*>(img));
    LdsCF* wb = lds_opaque(wl + 4 * lane);
    f16x8 a1[2], a2, b1[2][NF], b2[NF];
    auto fetch_b1 = [&](int s, f16x8 (&d)[NF]) {
#pragma unroll
        for (int f = 0; f < NF; ++f) d[f] = *reinterpret_cast<LdsH8*>(base + (s * TAPB + f * 256) / 4);
    };
    auto fetch_b2 = [&](int s) {
#pragma unroll
        for (int f = 0; f < NF; ++f) b2[f] = *reinterpret_cast<LdsH8*>(base + (s * TAPB + f * 256 + HALF) / 4);
    };
    a1[0] = *reinterpret_cast<LdsH8*>(wb);
    fetch_b2(0);
    fetch_b1(0, b1[0]);
    a2 = *reinterpret_cast<LdsH8*>(wb + 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int f = 0; f < NF; ++f) lo[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s & 1], b2[f], lo[f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < KS) {
            a1[(s + 1) & 1] = *reinterpret_cast<LdsH8*>(wb + ((s + 1) * 2) * 256);
            fetch_b2(s + 1);
            fetch_b1(s + 1, b1[(s + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) hi[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s & 1], b1[s & 1][f], hi[f], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < NF; ++f) lo[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1[s & 1][f], lo[f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < KS) a2 = *reinterpret_cast<LdsH8*>(wb + ((s + 1) * 2 + 1) * 256);
    }
}

// ---- history copies: rows [r0, r0 + P) of an image (both split halves, both channel blocks) <-> a slot, 16 bytes per
// entry u = 4 row + part.  Four copies per pair, each the job of one or two waves (wave-uniform branches: the other waves
// skip them with a scalar jump), entry u = tid & 127 on every site so that the per-thread offsets are computed once per
// kernel: phase 1 -- waves 0, 1 save the x image's rows in front of the next window's column 0, wave 2 fetches the
// intermediate's rows of the previous window; phase 2 -- wave 4 saves the intermediate's, waves 6, 7 fetch the next pair's x
// rows.  The read is issued in front of the epilogue, the write behind it (the LDS latency hides behind the epilogue).
// the activated, split image entry of four consecutive channels of one column, from fp32 values
__device__ __forceinline__ void split_x4(const float (&v)[4], float slope, f16x4& h1, f16x4& h2, float& lowm) {
    const f32x2 a01 = split_act2(f32x2{v[0], v[1]}, slope);
    const f32x2 a23 = split_act2(f32x2{v[2], v[3]}, slope);
    lowm = low_max3(lowm, a01.x, a01.y);
    lowm = low_max3(lowm, a23.x, a23.y);
    f16x2 p0, q0, p1, q1;
    split2(a01, p0, q0);
    split2(a23, p1, q1);
    h1 = f16x4{p0.x, p0.y, p1.x, p1.y};
    h2 = f16x4{q0.x, q0.y, q1.x, q1.y};
}

template <class TL>
__device__ __forceinline__ void mrf_write_x(char* xwr, const float (&v)[TL::NF][4], float slope, LowGuard& low) {
    float lowm = 0.f;
#pragma unroll
    for (int f = 0; f < TL::NF; ++f) {
        f16x4 h1, h2;
        split_x4(v[f], slope, h1, h2, lowm);
        *reinterpret_cast<f16x4*>(xwr + f * 256) = h1;
        *reinterpret_cast<f16x4*>(xwr + f * 256 + TL::HALF) = h2;
    }
    low_note(low, 0, lowm);
}

// A wave inside its K loop outranks one that is in its epilogue (s_setprio): the waves of a SIMD finish a conv phase one
// after the other (the arbiter prefers the oldest), the early ones then run their epilogues under the late ones' MFMAs --
// and have slack until the phase's barrier, which the late ones do not [measured, tools/stage_bench.py: 84 -> 77 us at batch
// 1, 546 -> 513 at batch 8].  prio 2: the last third of the waves (the arbiter's losers) one step higher still.
__device__ __forceinline__ void mrf_prio_up(const MrfParams& p, int wave) {
    if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (p.prio == 2) {
        if (wave >= 8) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
    } else if (p.prio == 3) {
        if (wave >= 8) __builtin_amdgcn_s_setprio(3);
        else if (wave >= 4) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
    }
}

// ---- one pair:  xr <- xr + conv2(lrelu(conv1(x image) + b1)) + b2 ------------------------------------------------------
// On entry the x image holds lrelu(xr) split (complete for every wave, history rows in front), `wl` the pair's block.
// NEXT: what the x image holds when the pair returns (behind its last barrier): 0 -- lrelu(new xr) (the next pair of the
// block); 1 -- lrelu(x0) (the next ResBlock, or the next tile), written only if `write_next`.
// hq: this pair's history slot, hq_next: the slot of the pair that runs next (its conv1's `pnext` rows go in front of the x
// image).
template <class TL, class G, int DIL, int NEXT>
__device__ __forceinline__ void mrf_pair(const MrfParams& p, const MrfLane<TL>& L, const float* wl, float (&xr)[TL::NF][4],
                                         const float (&x0)[TL::NF][4], bool write_next, int tw, bool inside, int adv,
                                         char* hq, char* hq_next, int pnext, LowGuard& low, __amdgpu_buffer_rsrc_t rb,
                                         float* dma_dst, unsigned dma_off, int dma_pieces, int tile_no, int q) {
    constexpr int NF = TL::NF;
    constexpr int P1 = (G::KT - 1) * DIL / 2, P2 = G::P2;
    int rdoff = L.rdoff, wroff = L.wroff, tap16 = L.tap16, colw = L.colw, row0 = L.row0, cpslot = L.cpslot, cpimg = L.cpimg;
    asm volatile("" : "+v"(rdoff), "+v"(wroff), "+v"(tap16), "+v"(colw), "+v"(row0), "+v"(cpslot), "+v"(cpimg));
    char* const ximg = L.sm + TL::OFF_X;
    char* const mimg = L.sm + TL::OFF_M;
    const float* const bl = wl + 2 * G::WB / 4;          // [b1 | b2 | s1 | s2]
    const int wv = L.wave;
    // the block of the pair after this one: its slot was last read before the barrier this pair started behind
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q);
    mrf_dma<TL::NG>(rb, dma_dst, dma_off, dma_pieces, L.wave, L.lane);
    f32x4 hi[NF], lo[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) hi[f] = lo[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    mrf_prio_up(p, wv);
    mrf_mma<G, 2 * DIL * 16, TL::HALF>(wl, ximg + (rdoff + tap16 * DIL - P1 * 16), hi, lo, L.lane);
    if (p.prio) __builtin_amdgcn_s_setprio(0);
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 1);
    {
        f16x8 hv = {};
        if (wv < 2) {
            if (cpslot < 64 * P1) hv = *reinterpret_cast<const f16x8*>(ximg + cpimg + (TL::FM + adv - P1) * 16);
        } else if (wv == 2) {
            if (cpslot < 64 * P2) hv = *reinterpret_cast<const f16x8*>(hq + TL::HM + cpslot);
        }
        char* const mwr = mimg + wroff;
        const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + row0);
        const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 32 + row0);
        const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
        float lowm = 0.f;
        if (inside) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                f16x4 h1, h2;
                split_mid4<false>(hi[f], lo[f], s01, s23, b01, b23, p.slope, true, h1, h2, lowm);
                *reinterpret_cast<f16x4*>(mwr + f * 256) = h1;
                *reinterpret_cast<f16x4*>(mwr + f * 256 + TL::HALF) = h2;
            }
        } else {
            // conv2's zero padding applies to the intermediate: nothing exists outside [0, T)
            int cm = colw;
            asm volatile("" : "+v"(cm));                 // (the compares stay on this side of the branch)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int t = tw + cm + f * 16;
                f16x4 h1, h2;
                split_mid4<true>(hi[f], lo[f], s01, s23, b01, b23, p.slope, t >= 0 && t < p.T, h1, h2, lowm);
                *reinterpret_cast<f16x4*>(mwr + f * 256) = h1;
                *reinterpret_cast<f16x4*>(mwr + f * 256 + TL::HALF) = h2;
            }
        }
        low_note(low, 1, lowm);
        if (wv < 2) {
            if (cpslot < 64 * P1) *reinterpret_cast<f16x8*>(hq + TL::HX + cpslot) = hv;
        } else if (wv == 2) {
            if (cpslot < 64 * P2) *reinterpret_cast<f16x8*>(mimg + cpimg + (TL::FM - P2) * 16) = hv;
        }
    }
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 2);
    pair_barrier();                                      // (C) intermediate complete (history rows included), x image free
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 3);
#pragma unroll
    for (int f = 0; f < NF; ++f) hi[f] = lo[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    mrf_prio_up(p, wv);
    mrf_mma<G, 2 * 16, TL::HALF>(wl + G::WB / 4, mimg + (rdoff + tap16 - P2 * 16), hi, lo, L.lane);
    if (p.prio) __builtin_amdgcn_s_setprio(0);
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 4);
    {
        f16x8 hv = {};
        if (wv == 4) {
            if (cpslot < 64 * P2) hv = *reinterpret_cast<const f16x8*>(mimg + cpimg + (TL::FM + adv - P2) * 16);
        } else if ((wv >> 1) == 3) {
            if (cpslot < 64 * pnext) hv = *reinterpret_cast<const f16x8*>(hq_next + TL::HX + cpslot);
        }
        const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + 16 + row0);
        const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 48 + row0);
        const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
        if (inside) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                combine4(hi[f], lo[f], s01, s23, b01, b23, xr[f]);
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[f][i] = hi[f][i];
            }
        } else {
            // the next conv's zero padding applies to x: nothing exists outside [0, T)
            int cm = colw;
            asm volatile("" : "+v"(cm));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                combine4(hi[f], lo[f], s01, s23, b01, b23, xr[f]);
                const int t = tw + cm + f * 16;
                const bool ok = t >= 0 && t < p.T;
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[f][i] = ok ? hi[f][i] : 0.f;
            }
        }
        if constexpr (NEXT == 0) mrf_write_x<TL>(ximg + wroff, xr, p.slope, low);
        else if (write_next) mrf_write_x<TL>(ximg + wroff, x0, p.slope, low);
        if (wv == 4) {
            if (cpslot < 64 * P2) *reinterpret_cast<f16x8*>(hq + TL::HM + cpslot) = hv;
        } else if ((wv >> 1) == 3) {
            if (cpslot < 64 * pnext) *reinterpret_cast<f16x8*>(ximg + cpimg + (TL::FM - pnext) * 16) = hv;
        }
    }
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 5);
    pair_wait_vm0();                                     // this wave's pieces of the next block have landed
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 6);
    pair_barrier();                                      // (A) next x image complete, intermediate free, next block visible
}

// ---- one ResBlock: three pairs on xr (= x0 on entry).  pnext: conv1 reach of the pair that runs after this block (the next
// ResBlock's first, or the next tile's first): its history rows go in front of the x image at the end.  par: parity of the
// tile (a tile is nine pairs: pair q of tile n sits in weight slot (q + n) & 1)
template <class TL, int KT, int D0, int D1, int D2>
__device__ __forceinline__ void mrf_block_run(const MrfParams& p, const MrfLane<TL>& L, float* wbuf, int j, int par, int pnext,
                                              float (&xr)[TL::NF][4], const float (&x0)[TL::NF][4], bool write_next, int tw,
                                              bool inside, int adv, LowGuard& low, __amdgpu_buffer_rsrc_t rb,
                                              unsigned off_next_block, int pieces_next_block, int tile_no) {
    typedef MrfGeom<TL::NF, KT> G;
    constexpr int PIECES = (2 * G::WB + 1024) / 1024;
    const int q0 = 3 * j;
    float* const w0 = wbuf + ((q0 + par) & 1) * (TL::WSLOT / 4);
    float* const w1 = wbuf + ((q0 + par + 1) & 1) * (TL::WSLOT / 4);
    char* const hist = L.sm + TL::OFF_H;
    char* const h0 = hist + (q0 + 0) * TL::HSLOT;
    char* const h1 = hist + (q0 + 1) * TL::HSLOT;
    char* const h2 = hist + (q0 + 2) * TL::HSLOT;
    char* const hn = hist + (j == 2 ? 0 : q0 + 3) * TL::HSLOT;
    mrf_pair<TL, G, D0, 0>(p, L, w0, xr, x0, true, tw, inside, adv, h0, h1, (KT - 1) * D1 / 2, low, rb, w1, p.blk_off[q0 + 1], PIECES, tile_no, q0);
    mrf_pair<TL, G, D1, 0>(p, L, w1, xr, x0, true, tw, inside, adv, h1, h2, (KT - 1) * D2 / 2, low, rb, w0, p.blk_off[q0 + 2], PIECES, tile_no, q0 + 1);
    mrf_pair<TL, G, D2, 1>(p, L, w0, xr, x0, write_next, tw, inside, adv, h2, hn, pnext, low, rb, w1, off_next_block, pieces_next_block,
                           tile_no, q0 + 2);
}

template <class TL, int D0, int D1, int D2>
__device__ __forceinline__ void mrf_block(const MrfParams& p, const MrfLane<TL>& L, float* wbuf, int j, int par, int k, int knext,
                                          float (&xr)[TL::NF][4], const float (&x0)[TL::NF][4], bool write_next, int tw, bool inside,
                                          int adv, LowGuard& low, __amdgpu_buffer_rsrc_t rb, unsigned off_next, int pieces_next,
                                          int tile_no) {
    const int pnext = (knext - 1) * D0 / 2;
    if (k == 11) mrf_block_run<TL, 11, D0, D1, D2>(p, L, wbuf, j, par, pnext, xr, x0, write_next, tw, inside, adv, low, rb, off_next, pieces_next, tile_no);
    else if (k == 7) mrf_block_run<TL, 7, D0, D1, D2>(p, L, wbuf, j, par, pnext, xr, x0, write_next, tw, inside, adv, low, rb, off_next, pieces_next, tile_no);
    else mrf_block_run<TL, 3, D0, D1, D2>(p, L, wbuf, j, par, pnext, xr, x0, write_next, tw, inside, adv, low, rb, off_next, pieces_next, tile_no);
}

__device__ __forceinline__ int mrf_pieces(int k) { return (2 * ((k + 1) / 2) * 2048 + 1024) / 1024; }

// A block's run of tiles: the global output columns [g, g_hi) (utterances concatenated) in order.  A tile is the window of
// utterance b that starts at time tw and is final for the times [lo, hi).
struct MrfIter {
    int b, tw, lo, hi, run_end;
    bool cold;               // first tile of a run: nothing in the history slots belongs to it
};
__device__ __forceinline__ void mrf_first(MrfIter& it, long long g, long long g_hi, int T, int halo, int ol, int vcols) {
    it.b = (int)(g / T);
    const int a = (int)(g - (long long)it.b * T);
    const long long rest = g_hi - (long long)it.b * T;
    it.run_end = rest < T ? (int)rest : T;
    it.cold = true;
    it.tw = a - halo - ol;
    it.lo = a;
    it.hi = min(it.run_end, it.tw + vcols - ol);
}
// false: the share is done
__device__ __forceinline__ bool mrf_next(MrfIter& it, long long g_hi, int T, int halo, int ol, int vcols, int adv) {
    if (it.hi < it.run_end) {
        it.tw += adv;
        it.lo = it.hi;
        it.hi = min(it.run_end, it.tw + vcols - ol);
        it.cold = false;
        return true;
    }
    const long long g = (long long)it.b * T + it.run_end;
    if (g >= g_hi) return false;
    mrf_first(it, g, g_hi, T, halo, ol, vcols);
    return true;
}

template <int NF, int NG, int D0, int D1, int D2, bool FOLD>
__global__ __launch_bounds__(64 * NG) __attribute__((amdgpu_waves_per_eu((NG + 3) / 4, (NG + 3) / 4))) void mrfh_kernel(MrfParams p) {
    typedef MrfTile<NF, NG> TL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);
    MrfLane<TL> L;
    L.tid = threadIdx.x;
    L.lane = L.tid & 63;
    L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
    L.sm = sm;
    {
        const int n = L.lane & 15, g = L.lane >> 4;
        L.colw = L.wave * (16 * NF) + n;
        L.row0 = 4 * g;
        L.tap16 = 16 * (g >> 1);
        L.rdoff = ((g & 1) * TL::RP + TL::FM + L.colw) * 16;
        L.wroff = ((g >> 1) * TL::RP + TL::FM + L.colw) * 16 + 8 * (g & 1);
        const int u = L.tid & 127, part = u & 3, row = u >> 2;
        L.cpslot = u * 16;
        L.cpimg = (part >> 1) * TL::HALF + (part & 1) * (TL::RP * 16) + row * 16;
    }
    static_assert(NG >= 8, "the history copies are the job of waves 0 ... 7");
    char* const ximg0 = sm + TL::OFF_X;
    char* const mimg0 = sm + TL::OFF_M;
    float* const wbuf = smem + TL::OFF_W / 4;
    float* const scratch = smem + TL::OFF_S / 4;
    const int T = p.T, halo = p.halo, ol = p.ol;
    const int vcols = TL::W - halo;                      // columns of a window that are final for the pairs (given history)
    const int adv = vcols - 2 * ol;                      // window advance inside a run
    const int k0 = p.k[0], k1 = p.k[1], k2 = p.k[2];
    const unsigned off0 = p.blk_off[0], off3 = p.blk_off[3], off6 = p.blk_off[6];
    asm volatile("" ::"s"(T), "s"(halo), "s"(ol), "s"(k0), "s"(k1), "s"(k2), "s"(off0), "s"(off3), "s"(off6));
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.blob, p.blob_bytes);

    // this block's share of the B * T output columns (XCD-aware: neighbouring shares on one XCD)
    const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const long long g_lo = p.total * share / p.nblk, g_hi = p.total * (share + 1) / p.nblk;
    if (g_lo >= g_hi) return;
    MrfIter it;
    mrf_first(it, g_lo, g_hi, T, halo, ol, vcols);

    const unsigned t4 = (unsigned)T * 4u, ubytes = (unsigned)TL::C * (unsigned)T * 4u;
    const size_t ustride = (size_t)TL::C * (size_t)T;
    float x0[NF][4], xr[NF][4], sum[NF][4];
    auto load_x0 = [&](const MrfIter& at) {
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + at.b * ustride, ubytes);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int t = at.tw + L.colw + f * 16;
            const unsigned voff = t >= 0 && t < T ? (unsigned)(L.row0 * T + t) * 4u : kOutOfRange;
#pragma unroll
            for (int i = 0; i < 4; ++i) x0[f][i] = buffer_load1s(rx, voff, (unsigned)i * t4);
        }
    };
    // ---- prologue: margins and history slots zero (finite), first window, first block ----
    load_x0(it);
    mrf_dma<NG>(rb, wbuf, off0, mrf_pieces(k0), L.wave, L.lane);
    for (int idx = L.tid; idx < 2 * 4 * 2 * TL::FM * 4; idx += TL::NT) {
        // (image, part, front / back, row, dword)
        const int dw = idx & 3, row = (idx >> 2) % TL::FM, fb = (idx >> 2) / TL::FM % 2, part = (idx >> 2) / (2 * TL::FM) % 4,
                  im = (idx >> 2) / (8 * TL::FM);
        char* const img = im ? mimg0 : ximg0;
        reinterpret_cast<float*>(img + (part >> 1) * TL::HALF + (part & 1) * (TL::RP * 16) + ((fb ? TL::FM + TL::W : 0) + row) * 16)[dw] = 0.f;
    }
    for (int idx = L.tid; idx < 9 * TL::HSLOT / 4; idx += TL::NT) reinterpret_cast<float*>(sm + TL::OFF_H)[idx] = 0.f;
    if constexpr (FOLD) {
        // the output conv's weights as [4 channels' group q][tap j][4 channels] (the group's seven taps are seven 16-byte
        // entries: one uniform ds_read_b128 each), the bias behind them
        float* const fw = reinterpret_cast<float*>(sm + TL::OFF_F);
        if (L.tid < 112) {
            const int q = L.tid / 28, j = (L.tid % 28) / 4, cc = L.tid & 3;
            fw[L.tid] = p.fold_w[(4 * q + cc) * 7 + j];
        }
        if (L.tid == 112) fw[112] = p.fold_b ? p.fold_b[0] : 0.f;
    }
    LowGuard low;
    float bad = 0.f;
    const float rcp = div_rcp(p.out_div);
    mrf_stamp(p, NG, L.wave, L.lane, 2, 0);              // (stamps of "tile 2": kernel entry, prologue)
    pair_wait_vm0();
    mrf_stamp(p, NG, L.wave, L.lane, 2, 1);
    mrf_write_x<TL>(ximg0 + L.wroff, x0, p.slope, low);
    pair_barrier();
    int tile_no = 0, par = 0;
    for (;;) {
        const MrfIter cur = it;
        const bool more = mrf_next(it, g_hi, T, halo, ol, vcols, adv);
        const bool inside = cur.tw >= 0 && cur.tw + TL::W <= T;
        // ---- the three ResBlocks, one after the other on the same window (a loop, not three copies of the code: a
        // block body is ~12 KB of instructions per tap count) ----
#pragma unroll 1
        for (int j = 0; j < 3; ++j) {
            const int kj = j == 0 ? k0 : j == 1 ? k1 : k2, kn = j == 0 ? k1 : j == 1 ? k2 : k0;
            const unsigned offn = j == 0 ? off3 : j == 1 ? off6 : (more ? off0 : kOutOfRange);
            const int pcn = (j < 2 || more) ? mrf_pieces(kn) : 0;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[f][i] = x0[f][i];
            // x0 is free once the last ResBlock has its copy: the next tile's window travels during that block
            if (j == 2 && more) load_x0(it);
            mrf_block<TL, D0, D1, D2>(p, L, wbuf, j, par, kj, kn, xr, x0, j < 2 || more, cur.tw, inside, adv, low, rb, offn, pcn, tile_no);
            if (j < 2) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum[f][i] = j == 0 ? xr[f][i] : sum[f][i] + xr[f][i];
            }
        }
        // ---- ((r0 + r1) + r2) / 3 and the stores (behind the tile's last barrier: they drain under the next tile) ----
        if (tile_no == 0) mrf_stamp(p, NG, L.wave, L.lane, 2, 2);
        float out[NF][4];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[f][i] = sum[f][i] + xr[f][i];
        // (one wave-uniform branch around all twelve values: with the choice inside the loops hipcc branches per value)
        if (rcp != 0.f) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) out[f][i] = div_exact(out[f][i], p.out_div, rcp);
        } else if (p.out_div != 1.f) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) out[f][i] = out[f][i] / p.out_div;
        }
        if constexpr (FOLD) {
            // The activated tile -> LDS (over the intermediate image, free since the last barrier), zero outside [0, T), as
            // [column][16 channels]: the four channels of a D fragment are ONE 16-byte entry, and one output sample reads a
            // group of four channels of a tap as one ds_read_b128.  The four entries of a column are rotated by column / 4
            // (entry q of column c sits in slot (q + (c >> 2)) & 3): the lane groups a b128 access is served in -- quads
            // of four consecutive columns, 4 or 12 apart -- then cover all sixteen 16-byte slots of a bank row, reads and
            // writes alike.  Then one output sample per thread: conv_narrow_kernel's arithmetic (channel-major FMA chain
            // from zero, bias last), so plans that keep the output conv as a launch of its own give the same bits.
            char* const sb = mimg0;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int col = L.colw + f * 16, t = cur.tw + col;
                const bool ok = t >= 0 && t < T;
                range_note4(bad, out[f][0], out[f][1], out[f][2], out[f][3], t >= cur.lo - ol && t < cur.hi + ol && ok);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ok ? act(out[f][i], p.act_slope) : 0.f;
                *reinterpret_cast<f32x4*>(sb + col * 64 + ((((L.row0 >> 2) + (col >> 2)) & 3) << 4)) = v;
            }
            if (tile_no == 0) mrf_stamp(p, NG, L.wave, L.lane, 2, 3);
            pair_barrier();
            if (tile_no == 0) mrf_stamp(p, NG, L.wave, L.lane, 2, 4);
            const float* const fw = reinterpret_cast<const float*>(sm + TL::OFF_F);
            for (int c0 = L.tid; c0 < vcols - 2 * ol; c0 += TL::NT) {
                // output column: the first final one of a warm tile is `ol`; a cold tile's lie further right (time mask).
                // 112 dependent FMAs in four rounds of 28 (a group of four channels, seven taps).  [measured, tools/mrf_trace.py]
                // the channel-major tile of b32 operands: 6 900 cycles per tile as hipcc scheduled it (a wait behind every
                // operand pair), 5 300 with the next channel's operands requested ahead of this channel's FMAs
                const int col = c0 + ol, t = cur.tw + col;
                float o = 0.f;
                // (four rounds of 14 loads + 28 FMAs, no software pipeline: two operand sets in flight need 112 registers and
                // hipcc spills 180 for them; a round's LDS round trip is ~150 cycles, four of them per tile)
#pragma unroll 1
                for (int q = 0; q < 4; ++q) {
                    f32x4 d[7], w[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const int row = col - 3 + j;
                        d[j] = *reinterpret_cast<const f32x4*>(sb + row * 64 + (((q + (row >> 2)) & 3) << 4));
                        w[j] = *reinterpret_cast<const f32x4*>(fw + (q * 7 + j) * 4);
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int j = 0; j < 7; ++j) o = fmaf(w[j][cc], d[j][cc], o);
                }
                o = o + fw[112];
                if (p.post == FV_POST_TANH) o = tanhf(o);
                else if (p.post == FV_POST_RELU) o = fmaxf(o, 0.f);
                if (t >= cur.lo && t < cur.hi) p.fold_y[(size_t)cur.b * T + t] = o;
            }
            if (tile_no == 0) mrf_stamp(p, NG, L.wave, L.lane, 2, 5);
            pair_barrier();
            if (tile_no == 0) mrf_stamp(p, NG, L.wave, L.lane, 2, 6);
            // The tile lay over the rows BEHIND the intermediate's window too: they feed discarded columns and, through
            // the zero pad tap, final ones -- finite values again (zeros).  (The rows in FRONT of the window are only ever
            // read as far as a pair's history copy has just rewritten them.)  No barrier: nobody else writes these rows,
            // and they are next read behind the next pair's barrier (C), which this wave has yet to reach.
            for (int idx = L.tid; idx < 4 * TL::FM * 4; idx += TL::NT) {
                const int dw = idx & 3, row = (idx >> 2) % TL::FM, part = (idx >> 2) / TL::FM;
                reinterpret_cast<float*>(mimg0 + (part >> 1) * TL::HALF + (part & 1) * (TL::RP * 16) + (TL::FM + TL::W + row) * 16)[dw] = 0.f;
            }
        } else {
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + cur.b * ustride, ubytes);
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.y_act ? p.y_act + cur.b * ustride : p.y, p.y_act ? ubytes : 0u);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int t = cur.tw + L.colw + f * 16;
                const bool ok = t >= cur.lo && t < cur.hi;
                range_note4(bad, out[f][0], out[f][1], out[f][2], out[f][3], ok);
                const unsigned voff = ok ? (unsigned)(L.row0 * T + t) * 4u : kOutOfRange;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = out[f][i];
                    if (p.post == FV_POST_TANH) v[i] = tanhf(v[i]);
                    else if (p.post == FV_POST_RELU) v[i] = fmaxf(v[i], 0.f);
                }
                if (p.y_act) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) buffer_store1s(ry, voff, (unsigned)i * t4, v[i]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) buffer_store1s(ra, voff, (unsigned)i * t4, act(v[i], p.act_slope));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) buffer_store1s(ry, voff, (unsigned)i * t4, p.act_slope != 1.f ? act(v[i], p.act_slope) : v[i]);
                }
            }
        }
        mrf_stamp(p, NG, L.wave, L.lane, tile_no, 63);
        if (!more) break;
        ++tile_no;
        par ^= 1;
    }
    if (p.guard && bad != bad) guard_raise_high(p.guard);
    {
        PairCore pc;
        pc.guard = p.guard;
        low_flag(pc, low, scratch, L.wave, L.lane, NG);
    }
}

}  // namespace fv
