// A whole MRF stage of HiFi-GAN at 32 channels in ONE launch: mrfh_kernels.hpp's scheme (running fp32 x in registers,
// time-aligned split images in LDS, one-sided halo, persistent blocks) where the stage no longer fits the way the
// 16-channel stage does:
//
//     r_j = ResBlock1_j(x)  (reference model/generator/modules.py:223-230),  y = ((r_0 + r_1) + r_2) / 3  (hifigan.py:97-103)
//
//   * M = 32: a wave owns NF fragments of 16 columns x BOTH row halves (D fragments [2][NF]); an A operand feeds NF
//     MFMAs, a B operand two -- 10 ds_read_b128 per 18 MFMAs at NF = 3, so the K loops are matrix-bound where one wave per
//     (row half, fragment) would be LDS-bound;
//   * an image row is 128 bytes (two split halves x four 8-channel blocks): two images of W + 64 rows are 112 KB at
//     W = 384, and W must be at least 360 -- a run's first tile gives W - 2 HALO = W - 120 final columns and at batch 1 the
//     32-channel stage of the headline (T = 61 440) hands every one of 256 blocks 240 columns: one tile per block;
//   * so the WEIGHTS cannot sit in LDS a pair at a time (an 11-tap pair is 90 KB).  They stream in PIECES of at most four
//     taps (16 KB; an 11-tap conv is pieces of 4 + 4 + 3 taps, a 7-tap one 4 + 3) through two slots: the barrier in front
//     of piece i tells everybody that piece i - 1 is consumed and that piece i has landed, and the LDS-DMA of piece i + 1 is
//     issued right behind it.  A conv of n pieces has n - 1 extra barriers: 36 per tile where the 16-channel kernel has 18;
//   * the HISTORY (the (k - 1) d / 2 rows in front of the next tile's first column, per conv) does not fit either: it lives
//     in global memory (MrfParams::hist, 3 840 bytes per pair and block, fv_mrf_stage_workspace_bytes), written and read back
//     by the same threads (program order), one tile apart.  A run's first tile reads zeros instead, and a run's last tile
//     saves nothing -- at batch 1 the history is never touched.
// Arithmetic: pairh_mma's sums in pairh_mma's order per accumulator, split_mid4 / combine4 / div_exact epilogues: bit-identical
// to the stage as four launches of fused pairs (tests/test_gpu_stage.py).
#pragma once
#include "mrfh_kernels.hpp"

namespace fv {

template <int NF_, int NG_>
struct MrfwTile {
    static constexpr int NF = NF_, NG = NG_, C = 32;
    static constexpr int NT = 64 * NG;
    static constexpr int W = 16 * NF * NG;
    static constexpr int FM = 32;
    static constexpr int RP = W + 2 * FM;
    static constexpr int BLK = RP * 16;                 // one 8-channel block of one split half
    static constexpr int HALF = 4 * BLK;
    static constexpr int IMG = 2 * HALF;
    static constexpr int PT = 4;                        // taps of a weight piece
    static constexpr int STEP = 4096;                   // bytes of one tap's A operands: [row half][split half][lane][8 halves]
    static constexpr int PIECE = PT * STEP;
    static constexpr int TAIL = 512;                    // a pair's [b1 | b2 | 1/prescale of conv1 | conv2], 32 floats each
    static constexpr int HROW = 128;                    // history: bytes per image row (2 split halves x 4 blocks x 16)
    static constexpr int HX = 0, HM = 25 * HROW, HSLOT = 30 * HROW;
    static constexpr int OFF_W = 0, OFF_X = 2 * PIECE, OFF_M = OFF_X + IMG, OFF_B = OFF_M + IMG, OFF_S = OFF_B + 9 * TAIL;
    static constexpr int OFF_F = OFF_S + 256;           // FOLD: the output conv's weights [8 groups][7 taps][4] and bias
    static constexpr int LDS = OFF_F + 1024;
    static_assert(W * 128 <= IMG, "FOLD: the activated fp32 tile [column][32 channels] lies over the intermediate image");
    static_assert(RP % 16 == 0, "whole bank rows");
    static_assert(NG >= 5 && NG <= 8, "history copies: waves 0 ... 4; a piece is at most two LDS-DMA instructions per wave");
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <class TL>
struct MrfwLane {
    int tid, lane, wave;
    int colw;                // window column of the lane's fragment 0
    int row0;                // first of its four channels inside a row half (D fragment)
    int rdoff;               // B operand: byte offset of row FM + colw of its channel block (lane group g: channels 8g ...)
    int wroff;               // D fragment of row half 0: byte offset of its half block entry of row FM + colw
    int cpslot;              // history copies: 16 x entry u = tid & 255 (u = 8 row + part, part = 4 split half + block)
    int cpimg;               // ... byte offset of (row, part) inside an image, relative to row 0
    char* sm;
};

// the LDS-DMA of `kb` KB at blob offset `off`: waves round-robin, at most two instructions per wave (no loop: hipcc drains
// vmcnt in front of a loop that holds an LDS-DMA)
template <int NG>
__device__ __forceinline__ void mrfw_dma(__amdgpu_buffer_rsrc_t rb, float* dst, unsigned off, int kb, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = wave + r * NG;
        if (i < kb) dma16(rb, dst + i * 256, off + (unsigned)(i * 1024 + lane * 16));
    }
}

// STEPS taps of one conv: hi += a1 b1, lo += a1 b2 + a2 b1 per tap, in pairh_mma's order per accumulator.  Queues as in
// mrf_mma: the a1 b2 group first, so b2 (and a2, used by the last group) are single-buffered: 60 operand registers.
template <int NF, int STEPS, int TAPB, int HALF>
__device__ __forceinline__ void mrfw_mma(const float* wl, const char* img, f32x4 (&hi)[2][NF], f32x4 (&lo)[2][NF], int lane) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    LdsCF* base = lds_opaque(reinterpret_cast<const float*>(img));
    LdsCF* wb = lds_opaque(wl + 4 * lane);
    f16x8 a1[2][2], a2[2], b1[2][NF], b2[NF];
    auto fetch_a1 = [&](int s, f16x8 (&d)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) d[h] = *reinterpret_cast<LdsH8*>(wb + ((s * 2 + h) * 2) * 256);
    };
    auto fetch_a2 = [&](int s) {
#pragma unroll
        for (int h = 0; h < 2; ++h) a2[h] = *reinterpret_cast<LdsH8*>(wb + ((s * 2 + h) * 2 + 1) * 256);
    };
    auto fetch_b1 = [&](int s, f16x8 (&d)[NF]) {
#pragma unroll
        for (int f = 0; f < NF; ++f) d[f] = *reinterpret_cast<LdsH8*>(base + (s * TAPB + f * 256) / 4);
    };
    auto fetch_b2 = [&](int s) {
#pragma unroll
        for (int f = 0; f < NF; ++f) b2[f] = *reinterpret_cast<LdsH8*>(base + (s * TAPB + f * 256 + HALF) / 4);
    };
    fetch_a1(0, a1[0]);
    fetch_b2(0);
    fetch_b1(0, b1[0]);
    fetch_a2(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < NF; ++f) lo[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s & 1][h], b2[f], lo[h][f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < STEPS) {
            fetch_a1(s + 1, a1[(s + 1) & 1]);
            fetch_b2(s + 1);
            fetch_b1(s + 1, b1[(s + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < NF; ++f) hi[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s & 1][h], b1[s & 1][f], hi[h][f], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < NF; ++f) lo[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[h], b1[s & 1][f], lo[h][f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < STEPS) fetch_a2(s + 1);
    }
}

__host__ __device__ constexpr int mrfw_piece_taps(int k, int r) { return k - 4 * r < 4 ? k - 4 * r : 4; }

// One conv over its pieces.  On entry (behind a barrier) the conv's piece 0 sits in slot `par`; on return the LDS-DMA of the
// conv that runs next (blob offset next_off, next_kb KB; 0: none) is under way into slot `par` (toggled once per piece).
template <class TL, int KT, int TAPB>
__device__ __forceinline__ void mrfw_conv(const MrfParams& p, const MrfwLane<TL>& L, float* wbuf, int& par, const char* img,
                                          f32x4 (&hi)[2][TL::NF], f32x4 (&lo)[2][TL::NF], __amdgpu_buffer_rsrc_t rb,
                                          unsigned conv_off, unsigned next_off, int next_kb) {
    constexpr int NP = (KT + TL::PT - 1) / TL::PT;
    // taps of pieces 0, 1, 2 (pieces a conv does not have: 1, never run)
    constexpr int S0 = mrfw_piece_taps(KT, 0), S1 = NP >= 2 ? mrfw_piece_taps(KT, 1) : 1, S2 = NP >= 3 ? mrfw_piece_taps(KT, 2) : 1;
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        if (r > 0) {
            pair_wait_vm0();                             // this wave's part of piece r has landed
            pair_barrier();                              // ... everybody's; piece r - 1 is consumed
        }
        float* const nxt = wbuf + ((par + 1) & 1) * (TL::PIECE / 4);
        if (r + 1 < NP) mrfw_dma<TL::NG>(rb, nxt, conv_off + (unsigned)((r + 1) * TL::PIECE), 4 * mrfw_piece_taps(KT, r + 1), L.wave, L.lane);
        else mrfw_dma<TL::NG>(rb, nxt, next_off, next_kb, L.wave, L.lane);
        mrf_prio_up(p, L.wave);
        const float* const wl = wbuf + (par & 1) * (TL::PIECE / 4);
        if (r == 0) mrfw_mma<TL::NF, S0, TAPB, TL::HALF>(wl, img, hi, lo, L.lane);
        else if (r == 1) mrfw_mma<TL::NF, S1, TAPB, TL::HALF>(wl, img + 4 * TAPB, hi, lo, L.lane);
        else mrfw_mma<TL::NF, S2, TAPB, TL::HALF>(wl, img + 8 * TAPB, hi, lo, L.lane);
        if (p.prio) __builtin_amdgcn_s_setprio(0);
        par ^= 1;
    }
}

template <class TL>
__device__ __forceinline__ void mrfw_write_x(char* xwr, const float (&v)[2][TL::NF][4], float slope, LowGuard& low) {
    float lowm = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < TL::NF; ++f) {
            f16x4 h1, h2;
            split_x4(v[h][f], slope, h1, h2, lowm);
            *reinterpret_cast<f16x4*>(xwr + h * 2 * TL::BLK + f * 256) = h1;
            *reinterpret_cast<f16x4*>(xwr + h * 2 * TL::BLK + f * 256 + TL::HALF) = h2;
        }
    low_note(low, 0, lowm);
}

// what a pair needs to know about the tile and its neighbours
struct MrfwWhen {
    int tw;                  // time of window column 0
    bool inside;             // the window lies inside [0, T): no masks
    int adv;                 // window advance inside a run
    bool save;               // the run goes on behind this tile: save history
    bool zero_mid;           // this tile is a run's first: the intermediate's history rows are zeros
    bool zero_next;          // ... and so are the x rows of the pair that runs next (the next tile's first pair: !save)
};

// ---- one pair:  xr <- xr + conv2(lrelu(conv1(x image) + b1)) + b2  (mrf_pair's contract) -------------------------------
// hq / hq_next: byte offsets of this pair's / the next-running pair's history slot inside the block's history (rh);
// pair_off: the pair's blob offset; next_off / next_kb: piece 0 of the conv1 that runs after this pair.
template <class TL, int KT, int DIL, int NEXT>
__device__ __forceinline__ void mrfw_pair(const MrfParams& p, const MrfwLane<TL>& L, float* wbuf, int& par, const float* bl,
                                          float (&xr)[2][TL::NF][4], const float (&x0)[2][TL::NF][4], bool write_next,
                                          const MrfwWhen& wh, int hq, int hq_next, int pnext, LowGuard& low,
                                          __amdgpu_buffer_rsrc_t rb, __amdgpu_buffer_rsrc_t rh, unsigned pair_off, unsigned next_off,
                                          int next_kb, int tile_no, int q) {
    constexpr int NF = TL::NF;
    constexpr int P1 = (KT - 1) * DIL / 2, P2 = (KT - 1) / 2;
    constexpr int WB = KT * TL::STEP;
    int rdoff = L.rdoff, wroff = L.wroff, colw = L.colw, row0 = L.row0, cpslot = L.cpslot, cpimg = L.cpimg;
    asm volatile("" : "+v"(rdoff), "+v"(wroff), "+v"(colw), "+v"(row0), "+v"(cpslot), "+v"(cpimg));
    char* const ximg = L.sm + TL::OFF_X;
    char* const mimg = L.sm + TL::OFF_M;
    const int wv = L.wave;
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q);
    f32x4 hi[2][NF], lo[2][NF];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < NF; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    mrfw_conv<TL, KT, DIL * 16>(p, L, wbuf, par, ximg + (rdoff - P1 * 16), hi, lo, rb, pair_off, pair_off + (unsigned)WB,
                                4 * mrfw_piece_taps(KT, 0));
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 1);
    {
        // history, phase 1: waves 0 ... 3 save the x image's rows in front of the next window's column 0; wave 4 fetches the
        // intermediate's rows of the previous window
        u32x4 hv = {0u, 0u, 0u, 0u};
        if (wv < 4) {
            if (wh.save && cpslot < TL::HROW * P1) hv = *reinterpret_cast<const u32x4*>(ximg + cpimg + (TL::FM + wh.adv - P1) * 16);
        } else if (wv == 4) {
            if (!wh.zero_mid && cpslot < TL::HROW * P2) hv = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rh, hq + TL::HM + cpslot, 0, 0));
        }
        char* const mwr = mimg + wroff;
        float lowm = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + 16 * h + row0);
            const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 64 + 16 * h + row0);
            const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
            if (wh.inside) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    f16x4 h1, h2;
                    split_mid4<false>(hi[h][f], lo[h][f], s01, s23, b01, b23, p.slope, true, h1, h2, lowm);
                    *reinterpret_cast<f16x4*>(mwr + h * 2 * TL::BLK + f * 256) = h1;
                    *reinterpret_cast<f16x4*>(mwr + h * 2 * TL::BLK + f * 256 + TL::HALF) = h2;
                }
            } else {
                // conv2's zero padding applies to the intermediate: nothing exists outside [0, T)
                int cm = colw;
                asm volatile("" : "+v"(cm));
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int t = wh.tw + cm + f * 16;
                    f16x4 h1, h2;
                    split_mid4<true>(hi[h][f], lo[h][f], s01, s23, b01, b23, p.slope, t >= 0 && t < p.T, h1, h2, lowm);
                    *reinterpret_cast<f16x4*>(mwr + h * 2 * TL::BLK + f * 256) = h1;
                    *reinterpret_cast<f16x4*>(mwr + h * 2 * TL::BLK + f * 256 + TL::HALF) = h2;
                }
            }
            // (the store behind the first row half: vmcnt counts stores too, and the barrier below waits for vmcnt(0))
            if (h == 0 && wv < 4) {
                if (wh.save && cpslot < TL::HROW * P1) __builtin_amdgcn_raw_buffer_store_b128(hv, rh, hq + TL::HX + cpslot, 0, 0);
            }
        }
        low_note(low, 1, lowm);
        if (wv == 4) {
            if (cpslot < TL::HROW * P2) *reinterpret_cast<u32x4*>(mimg + cpimg + (TL::FM - P2) * 16) = hv;
        }
    }
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 2);
    pair_wait_vm0();                                     // conv2's piece 0
    pair_barrier();                                      // (C) intermediate complete (history rows included), x image free
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 3);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int f = 0; f < NF; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    mrfw_conv<TL, KT, 16>(p, L, wbuf, par, mimg + (rdoff - P2 * 16), hi, lo, rb, pair_off + (unsigned)WB, next_off, next_kb);
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 4);
    {
        // history, phase 2: wave 4 saves the intermediate's rows, waves 0 ... 3 fetch the x rows of the pair that runs next
        u32x4 hv = {0u, 0u, 0u, 0u};
        if (wv == 4) {
            if (wh.save && cpslot < TL::HROW * P2) hv = *reinterpret_cast<const u32x4*>(mimg + cpimg + (TL::FM + wh.adv - P2) * 16);
        } else if (wv < 4) {
            if (!wh.zero_next && cpslot < TL::HROW * pnext) hv = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rh, hq_next + TL::HX + cpslot, 0, 0));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + 32 + 16 * h + row0);
            const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 96 + 16 * h + row0);
            const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
            if (wh.inside) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    combine4(hi[h][f], lo[h][f], s01, s23, b01, b23, xr[h][f]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) xr[h][f][i] = hi[h][f][i];
                }
            } else {
                // the next conv's zero padding applies to x: nothing exists outside [0, T)
                int cm = colw;
                asm volatile("" : "+v"(cm));
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    combine4(hi[h][f], lo[h][f], s01, s23, b01, b23, xr[h][f]);
                    const int t = wh.tw + cm + f * 16;
                    const bool ok = t >= 0 && t < p.T;
#pragma unroll
                    for (int i = 0; i < 4; ++i) xr[h][f][i] = ok ? hi[h][f][i] : 0.f;
                }
            }
        }
        if (wv == 4) {
            if (wh.save && cpslot < TL::HROW * P2) __builtin_amdgcn_raw_buffer_store_b128(hv, rh, hq + TL::HM + cpslot, 0, 0);
        }
        if constexpr (NEXT == 0) mrfw_write_x<TL>(ximg + wroff, xr, p.slope, low);
        else if (write_next) mrfw_write_x<TL>(ximg + wroff, x0, p.slope, low);
        if (wv < 4) {
            if (cpslot < TL::HROW * pnext) *reinterpret_cast<u32x4*>(ximg + cpimg + (TL::FM - pnext) * 16) = hv;
        }
    }
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 5);
    pair_wait_vm0();                                     // the next conv's piece 0
    mrf_stamp(p, TL::NG, L.wave, L.lane, tile_no, 7 * q + 6);
    pair_barrier();                                      // (A) next x image complete, intermediate free
}

// ---- one ResBlock: three pairs on xr (= x0 on entry).  The pair that runs after the block: conv1 reach pnext, blob offset
// next_off, piece 0 of next_kb KB.  zero_after: what MrfwWhen::zero_next is for the block's last pair.
template <class TL, int KT, int D0, int D1, int D2>
__device__ __forceinline__ void mrfw_block_run(const MrfParams& p, const MrfwLane<TL>& L, float* wbuf, int& par, int j, int pnext,
                                               float (&xr)[2][TL::NF][4], const float (&x0)[2][TL::NF][4], bool write_next,
                                               MrfwWhen wh, bool zero_after, LowGuard& low, __amdgpu_buffer_rsrc_t rb,
                                               __amdgpu_buffer_rsrc_t rh, unsigned next_off, int next_kb, int tile_no) {
    constexpr int KB0 = 4 * mrfw_piece_taps(KT, 0);
    const int q0 = 3 * j;
    const float* const bl = reinterpret_cast<const float*>(L.sm + TL::OFF_B) + q0 * (TL::TAIL / 4);
    const int h0 = q0 * TL::HSLOT, hn = (j == 2 ? 0 : q0 + 3) * TL::HSLOT;
    const unsigned o0 = p.blk_off[q0], o1 = p.blk_off[q0 + 1], o2 = p.blk_off[q0 + 2];
    mrfw_pair<TL, KT, D0, 0>(p, L, wbuf, par, bl, xr, x0, true, wh, h0, h0 + TL::HSLOT, (KT - 1) * D1 / 2, low, rb, rh, o0, o1, KB0,
                             tile_no, q0);
    mrfw_pair<TL, KT, D1, 0>(p, L, wbuf, par, bl + TL::TAIL / 4, xr, x0, true, wh, h0 + TL::HSLOT, h0 + 2 * TL::HSLOT,
                             (KT - 1) * D2 / 2, low, rb, rh, o1, o2, KB0, tile_no, q0 + 1);
    wh.zero_next = zero_after;
    mrfw_pair<TL, KT, D2, 1>(p, L, wbuf, par, bl + 2 * (TL::TAIL / 4), xr, x0, write_next, wh, h0 + 2 * TL::HSLOT, hn, pnext, low, rb,
                             rh, o2, next_off, next_kb, tile_no, q0 + 2);
}

template <int NF, int NG, int D0, int D1, int D2, bool FOLD>
__global__ __launch_bounds__(64 * NG) __attribute__((amdgpu_waves_per_eu((NG + 3) / 4, (NG + 3) / 4))) void mrfw_kernel(MrfParams p) {
    typedef MrfwTile<NF, NG> TL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);
    MrfwLane<TL> L;
    L.tid = threadIdx.x;
    L.lane = L.tid & 63;
    L.wave = __builtin_amdgcn_readfirstlane(L.tid >> 6);
    L.sm = sm;
    {
        const int n = L.lane & 15, g = L.lane >> 4;
        L.colw = L.wave * (16 * NF) + n;
        L.row0 = 4 * g;
        L.rdoff = (g * TL::RP + TL::FM + L.colw) * 16;
        L.wroff = ((g >> 1) * TL::RP + TL::FM + L.colw) * 16 + 8 * (g & 1);
        const int u = L.tid & 255, part = u & 7, row = u >> 3;
        L.cpslot = u * 16;
        L.cpimg = (part >> 2) * TL::HALF + (part & 3) * TL::BLK + row * 16;
    }
    char* const ximg0 = sm + TL::OFF_X;
    char* const mimg0 = sm + TL::OFF_M;
    float* const wbuf = smem + TL::OFF_W / 4;
    float* const scratch = smem + TL::OFF_S / 4;
    const int T = p.T, halo = p.halo, ol = p.ol;
    const int vcols = TL::W - halo;                      // columns of a window that are final for the pairs (given history)
    const int adv = vcols - 2 * ol;                      // window advance inside a run (FOLD: the output conv's reach either side)
    const int k0 = p.k[0], k1 = p.k[1], k2 = p.k[2];
    const unsigned off0 = p.blk_off[0], off3 = p.blk_off[3], off6 = p.blk_off[6];
    asm volatile("" ::"s"(T), "s"(halo), "s"(ol), "s"(k0), "s"(k1), "s"(k2), "s"(off0), "s"(off3), "s"(off6));
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.blob, p.blob_bytes);
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(p.hist + (size_t)blockIdx.x * (9 * TL::HSLOT / 4), p.hist ? 9 * TL::HSLOT : 0);

    const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const long long g_lo = p.total * share / p.nblk, g_hi = p.total * (share + 1) / p.nblk;
    if (g_lo >= g_hi) return;
    MrfIter it;
    mrf_first(it, g_lo, g_hi, T, halo, ol, vcols);

    const unsigned t4 = (unsigned)T * 4u, ubytes = (unsigned)TL::C * (unsigned)T * 4u;
    const size_t ustride = (size_t)TL::C * (size_t)T;
    float x0[2][NF][4], xr[2][NF][4], sum[2][NF][4];
    auto load_x0 = [&](const MrfIter& at) {
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + at.b * ustride, ubytes);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int t = at.tw + L.colw + f * 16;
                const unsigned voff = t >= 0 && t < T ? (unsigned)((L.row0 + 16 * h) * T + t) * 4u : kOutOfRange;
#pragma unroll
                for (int i = 0; i < 4; ++i) x0[h][f][i] = buffer_load1s(rx, voff, (unsigned)i * t4);
            }
    };
    // ---- prologue: margins zero (finite), the nine pairs' bias blocks, first window, first piece ----
    // L2 warm-up.  Every block walks the same 36 pieces in step, one piece ahead of its K loops (~1 us): inside a forward the
    // packed stage (306 KB) is in nobody's L2 when the launch starts, and a piece that every block of an XCD asks for at the
    // same moment is a miss for all of them.  So the blocks of an XCD (blockIdx % 8) read one slice of the stage each, up
    // front, into registers that nobody uses: by the time the third piece is due the whole stage sits in the XCD's L2.
    // [Measured inside the step: no difference -- the misses it removes were not what the launch waited for; it costs two
    // loads per thread and stays.]
    u32x4 warm0, warm1;
    {
        const unsigned per_xcd = gridDim.x >= 8 ? gridDim.x >> 3 : 1u;
        const unsigned slice = ((p.blob_bytes + per_xcd - 1) / per_xcd + 15u) & ~15u;
        const unsigned lo = ((blockIdx.x >> 3) % per_xcd) * slice;
        const unsigned o0 = (unsigned)L.tid * 16u, o1 = o0 + TL::NT * 16u;
        warm0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(o0 < slice ? lo + o0 : kOutOfRange), 0, 0));
        warm1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(o1 < slice ? lo + o1 : kOutOfRange), 0, 0));
        for (unsigned o = o1 + TL::NT * 16u; o < slice; o += TL::NT * 16u) {   // (grids of a few blocks: the rest, one by one)
            const u32x4 w = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(lo + o), 0, 0));
            asm volatile("" ::"v"(w));
        }
    }
    load_x0(it);
    mrfw_dma<NG>(rb, wbuf, off0, 4 * (k0 < 4 ? k0 : 4), L.wave, L.lane);
    for (int idx = L.tid; idx < 2 * 8 * 2 * TL::FM * 4; idx += TL::NT) {
        // (image, part, front / back, row, dword)
        const int dw = idx & 3, row = (idx >> 2) % TL::FM, fb = (idx >> 2) / TL::FM % 2, part = (idx >> 2) / (2 * TL::FM) % 8,
                  im = (idx >> 2) / (16 * TL::FM);
        char* const img = im ? mimg0 : ximg0;
        reinterpret_cast<float*>(img + (part >> 2) * TL::HALF + (part & 3) * TL::BLK + ((fb ? TL::FM + TL::W : 0) + row) * 16)[dw] = 0.f;
    }
    for (int idx = L.tid; idx < 9 * (TL::TAIL / 4); idx += TL::NT) {
        const int q = idx >> 7, e = idx & 127, j = q / 3;
        const int kj = j == 0 ? k0 : j == 1 ? k1 : k2;
        const unsigned base = (j == 0 ? off0 : j == 1 ? off3 : off6) + (unsigned)((q - 3 * j) * (2 * kj * TL::STEP + 1024));
        reinterpret_cast<float*>(sm + TL::OFF_B)[idx] = p.blob[(base + 2u * (unsigned)kj * TL::STEP) / 4 + e];
    }
    if constexpr (FOLD) {
        // the output conv's weights as [4 channels' group q][tap j][4 channels] (a group's seven taps are seven 16-byte entries:
        // one uniform ds_read_b128 each), the bias behind them
        float* const fw = reinterpret_cast<float*>(sm + TL::OFF_F);
        if (L.tid < 224) {
            const int q = L.tid / 28, j = (L.tid % 28) / 4, cc = L.tid & 3;
            fw[L.tid] = p.fold_w[(4 * q + cc) * 7 + j];
        }
        if (L.tid == 224) fw[224] = p.fold_b ? p.fold_b[0] : 0.f;
    }
    LowGuard low;
    float bad = 0.f;
    const float rcp = div_rcp(p.out_div);
    mrf_stamp(p, NG, L.wave, L.lane, 2, 0);
    pair_wait_vm0();
    asm volatile("" ::"v"(warm0), "v"(warm1));
    mrf_stamp(p, NG, L.wave, L.lane, 2, 1);
    mrfw_write_x<TL>(ximg0 + L.wroff, x0, p.slope, low);
    pair_barrier();
    int tile_no = 0, par = 0;
    for (;;) {
        const MrfIter cur = it;
        const bool more = mrf_next(it, g_hi, T, halo, ol, vcols, adv);
        MrfwWhen wh;
        wh.tw = cur.tw;
        wh.inside = cur.tw >= 0 && cur.tw + TL::W <= T;
        wh.adv = adv;
        wh.save = more && !it.cold;
        wh.zero_mid = wh.zero_next = cur.cold;
#pragma unroll 1
        for (int j = 0; j < 3; ++j) {
            const int kj = j == 0 ? k0 : j == 1 ? k1 : k2, kn = j == 0 ? k1 : j == 1 ? k2 : k0;
            const unsigned offn = j == 0 ? off3 : j == 1 ? off6 : off0;
            const int kbn = (j < 2 || more) ? 4 * (kn < 4 ? kn : 4) : 0;
            const int pnext = (kn - 1) * D0 / 2;
            const bool zero_after = j == 2 ? !wh.save : cur.cold;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) xr[h][f][i] = x0[h][f][i];
            if (j == 2 && more) load_x0(it);
            const bool wn = j < 2 || more;
            if (kj == 11) mrfw_block_run<TL, 11, D0, D1, D2>(p, L, wbuf, par, j, pnext, xr, x0, wn, wh, zero_after, low, rb, rh, offn, kbn, tile_no);
            else if (kj == 7) mrfw_block_run<TL, 7, D0, D1, D2>(p, L, wbuf, par, j, pnext, xr, x0, wn, wh, zero_after, low, rb, rh, offn, kbn, tile_no);
            else mrfw_block_run<TL, 3, D0, D1, D2>(p, L, wbuf, par, j, pnext, xr, x0, wn, wh, zero_after, low, rb, rh, offn, kbn, tile_no);
            if (j < 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum[h][f][i] = j == 0 ? xr[h][f][i] : sum[h][f][i] + xr[h][f][i];
            }
        }
        // ---- ((r0 + r1) + r2) / 3 and the stores (behind the tile's last barrier: they drain under the next tile) ----
        if (tile_no == 0) mrf_stamp(p, NG, L.wave, L.lane, 2, 2);
        if constexpr (FOLD) {
            // mrfh_kernel's fold at 32 channels.  The activated tile -> LDS over the intermediate image (free since the last
            // barrier), zero outside [0, T), as [column][32 channels]: the four channels of a D fragment are ONE 16-byte entry
            // (entry e = channels 4 e ...: e = 4 h + g), a column's eight entries rotated by column / 2 -- sixteen consecutive
            // columns then cover all sixteen 16-byte slots of a bank row, reads and writes alike.  Then one output sample per
            // thread: conv_narrow_kernel's arithmetic (channel-major FMA chain from zero, bias last), so plans that keep the
            // output conv as a launch of its own give the same bits.
            char* const sb = mimg0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float out[NF][4];
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[f][i] = sum[h][f][i] + xr[h][f][i];
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
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int col = L.colw + f * 16, t = cur.tw + col;
                    const bool ok = t >= 0 && t < T;
                    range_note4(bad, out[f][0], out[f][1], out[f][2], out[f][3], t >= cur.lo - ol && t < cur.hi + ol && ok);
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = ok ? act(out[f][i], p.act_slope) : 0.f;
                    *reinterpret_cast<f32x4*>(sb + col * 128 + (((4 * h + (L.row0 >> 2)) + (col >> 1)) & 7) * 16) = v;
                }
            }
            pair_barrier();
            const float* const fw = reinterpret_cast<const float*>(sm + TL::OFF_F);
            for (int c0 = L.tid; c0 < vcols - 2 * ol; c0 += TL::NT) {
                const int col = c0 + ol, t = cur.tw + col;
                float o = 0.f;
#pragma unroll 1
                for (int q = 0; q < 8; ++q) {
                    f32x4 d[7], w[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const int row = col - 3 + j;
                        d[j] = *reinterpret_cast<const f32x4*>(sb + row * 128 + ((q + (row >> 1)) & 7) * 16);
                        w[j] = *reinterpret_cast<const f32x4*>(fw + (q * 7 + j) * 4);
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int j = 0; j < 7; ++j) o = fmaf(w[j][cc], d[j][cc], o);
                }
                o = o + fw[224];
                if (p.post == FV_POST_TANH) o = tanhf(o);
                else if (p.post == FV_POST_RELU) o = fmaxf(o, 0.f);
                if (t >= cur.lo && t < cur.hi) p.fold_y[(size_t)cur.b * T + t] = o;
            }
            pair_barrier();
            // the tile lay over rows BEHIND the intermediate's window too: finite values again (mrfh_kernel)
            for (int idx = L.tid; idx < 8 * TL::FM * 4; idx += TL::NT) {
                const int dw = idx & 3, row = (idx >> 2) % TL::FM, part = (idx >> 2) / TL::FM;
                reinterpret_cast<float*>(mimg0 + (part >> 2) * TL::HALF + (part & 3) * TL::BLK + (TL::FM + TL::W + row) * 16)[dw] = 0.f;
            }
        } else {
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + cur.b * ustride, ubytes);
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.y_act ? p.y_act + cur.b * ustride : p.y, p.y_act ? ubytes : 0u);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float out[NF][4];
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[f][i] = sum[h][f][i] + xr[h][f][i];
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
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int t = cur.tw + L.colw + f * 16;
                    const bool ok = t >= cur.lo && t < cur.hi;
                    range_note4(bad, out[f][0], out[f][1], out[f][2], out[f][3], ok);
                    const unsigned voff = ok ? (unsigned)((L.row0 + 16 * h) * T + t) * 4u : kOutOfRange;
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
        }
        mrf_stamp(p, NG, L.wave, L.lane, tile_no, 63);
        if (!more) break;
        ++tile_no;
    }
    if (p.guard && bad != bad) guard_raise_high(p.guard);
    {
        PairCore pc;
        pc.guard = p.guard;
        low_flag(pc, low, scratch, L.wave, L.lane, NG);
    }
}

}  // namespace fv
