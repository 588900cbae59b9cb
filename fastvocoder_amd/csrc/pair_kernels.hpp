// Fused ResBlock1 pair on the gfx950 fp32 matrix cores:
//
//     x' = x + conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2          (reference model/generator/modules.py:223-230)
//
// conv1: KT taps, dilation DIL, 'same' zero padding; conv2: KT taps, undilated; C -> C channels,
// C = 16 * MH.  One launch runs this pair for the three ResBlocks of an MRF stage (hifigan.py:97-103,
// taps 11 / 7 / 3), and in SUM mode the LAST pairs of the three blocks plus the MRF mean
//     y = act( ( sum_j x'_j ) / 3 )
// in one accumulator.  Compared with two conv launches per pair (conv_kernels.hpp) the intermediate
// tensor never leaves the CU, x is read once (raw: the activation is applied on the LDS image, the
// residual is kept in registers), x' is written once, and there is no activated twin tensor.
//
// Structure, shaped by what round 1 measured: on gfx950 every non-MFMA VALU
// instruction costs fp32-matrix time, and at 16-32 channels a (tile, conv) unit holds too few MFMAs to
// hide per-tile setup.  So:
//   * blocks are PERSISTENT: each takes a contiguous, cost-balanced run of (member, utterance, tile)
//     items (static partition, no atomics), so weights are staged once per block, not once per tile;
//   * the weights of both convs live in LDS in A-FRAGMENT order ([row half][step/4][lane][4], packed by
//     fv_pack_pair_weight): a wave reloads its A operands for a phase with S/4 ds_read_b128 and keeps
//     them in VGPRs for every column fragment it owns -- the MFMA loop is v_mfma_f32_16x16x4_f32 +
//     one ds_read_b32 (the B operand, immediate offset from ONE per-block base register), no VALU;
//   * a tile = NM columns of the intermediate (16 * NF * NG), NOUT = NM - (KT-1) rounded down to a
//     multiple of 4 output columns; per tile: x window -> LDS by LDS-DMA (issued during the PREVIOUS
//     conv2 phase, so its latency hides behind matrix work), residual rows to registers, lrelu on the
//     LDS image (one pass: 2 VALU per element instead of 2 per operand read), conv1 -> + b1, lrelu,
//     zero outside [0, T) -> LDS, conv2 -> + b2 + residual -> HBM.  Three barriers per tile;
//   * waves = MH row halves x NG column groups; each wave owns NF 16-column fragments (NF = 2 at C = 16:
//     two independent accumulators cover the 40-cycle dependent latency of the 32-cycle MFMA).
//
// Per output element the K loop runs in a fixed order (channel group, tap), independent of the tile and
// batch decomposition: results do not depend on B or on how a batch is sharded over GPUs.
#pragma once
#include "conv_kernels.hpp"

namespace fv {

// smallest row stride (floats) >= n that is a multiple of 4 and = 16 (mod 32): the two channel rows a
// half-wave reads for a 16x16x4 B operand then sit on disjoint LDS banks
constexpr int pair_stride(int n) {
    int r = (n + 3) / 4 * 4;
    while (r % 32 != 16) r += 4;
    return r;
}

template <int MH_, int NF_, int NG_, int KT_, int DIL_>
struct PairGeom {
    static constexpr int MH = MH_, NF = NF_, NG = NG_, KT = KT_, DIL = DIL_;
    static constexpr int C = 16 * MH;
    static constexpr int NW = MH * NG, NT = 64 * NW;
    static constexpr int NM = 16 * NF * NG;             // intermediate columns per tile
    static constexpr int S = 4 * MH * KT;               // MFMA K steps per conv (K = C * KT = 4 * S)
    static constexpr int P1 = (KT - 1) * DIL / 2, P2 = (KT - 1) / 2;
    static constexpr int AOFF = (4 - (P1 + P2) % 4) % 4;   // the window starts AOFF columns early: multiple of 4
    static constexpr int XWIN = NM + (KT - 1) * DIL + AOFF;   // columns of x a tile reads
    static constexpr int NCOL4 = (XWIN + 3) / 4;
    static constexpr int XS = pair_stride(4 * NCOL4);   // LDS row stride of the x image
    static constexpr int XF4 = C * XS / 4;              // float4s of the x image
    static constexpr int NXI = (XF4 + 63) / 64;         // DMA instructions per x image
    static constexpr int NSLOT = (NXI + NW - 1) / NW;   // ... per wave
    static constexpr int MS = NM + 16;                  // LDS row stride of the intermediate (>= NM + KT - 1)
    static constexpr int WF = C * C * KT;               // floats per packed conv weight
    static constexpr int NOUT = (NM - (KT - 1)) / 4 * 4;   // output columns per tile
    static_assert(KT % 2 == 1 && KT - 1 <= 16, "odd tap counts up to 17");
    static_assert(4 * (C / 4 - 1) * XS * 4 + ((KT - 1) * DIL + 16 * (NF - 1)) * 4 < 65536, "ds_read immediate range");
};

__device__ __forceinline__ void pair_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void pair_barrier() { lds_barrier(); }

// s_waitcnt vmcnt(0) the compiler can see (simm16: vmcnt 0, expcnt 7, lgkmcnt 15): after it hipcc knows
// that no global store is in flight and does not guard later register re-use with waits of its own --
// waits that would also cover the LDS-DMA issued in between
__device__ __forceinline__ void pair_drain_vm() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// tuning aid, compiled in with -DFV_PAIR_TRACE only (tools/pair_trace.py): time stamp `ev` of this wave's
// `it`-th tile (every 64th block, tiles 0..7 only)
__device__ __forceinline__ void pair_stamp(const PairCore& p, int nw, int wave, int lane, int it, int ev) {
#ifdef FV_PAIR_TRACE
    if (p.trace && (blockIdx.x & 63) == 0 && blockIdx.x < 512 && it < 8 && lane == 0)
        p.trace[(((size_t)(blockIdx.x >> 6) * nw + wave) * 8 + it) * 16 + ev] = __builtin_amdgcn_s_memtime();
#endif
}

// ---- the weights of one member: global -> LDS, linear copy by LDS-DMA (both convs) -----------------
template <class G>
__device__ __forceinline__ void pair_stage_weights(const PairMember& mb, float* wl, int wave, int lane) {
    constexpr int NI = G::WF / 256;   // 64-lane x 16-byte instructions per conv
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(mb.w1, (unsigned)G::WF * 4u);
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(mb.w2, (unsigned)G::WF * 4u);
    for (int j = wave; j < NI; j += G::NW) {
        dma16(r1, wl + j * 256, (unsigned)(j * 1024 + lane * 16));
        dma16(r2, wl + G::WF + j * 256, (unsigned)(j * 1024 + lane * 16));
    }
}

// ---- x window of one tile: global -> LDS image [C][XS] ------------------------------------------------
template <class G>
struct PairDma {
    unsigned off[G::NSLOT];   // byte offset of the lane's float4 inside the (row 0, tA) window, or kOutOfRange
};

template <class G>
__device__ __forceinline__ void pair_dma_plan(PairDma<G>& d, int T, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < G::NSLOT; ++i) {
        const int idx = (wave + i * G::NW) * 64 + lane;
        const int row = idx / (G::XS / 4), c4 = idx % (G::XS / 4);
        d.off[i] = (idx < G::XF4 && c4 < G::NCOL4) ? (unsigned)(row * T + 4 * c4) * 4u : kOutOfRange;
    }
}

// tA: first column of the window (a multiple of 4; negative or past T at the sequence ends, where the
// zero padding comes from out-of-range offsets: rows are 16-byte aligned and T % 4 == 0, so every float4
// lies entirely inside or entirely outside [0, T))
template <class G>
__device__ __forceinline__ void pair_dma_x(const PairDma<G>& d, const float* xb, int T, float* xs, int tA,
                                           int wave, int lane) {
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)G::C * (unsigned)T * 4u);
    const bool interior = tA >= 0 && tA + 4 * G::NCOL4 <= T;
    if (interior) {
        const unsigned base = (unsigned)tA * 4u;
#pragma unroll
        for (int i = 0; i < G::NSLOT; ++i)
            if (wave + i * G::NW < G::NXI) dma16(rx, xs + (wave + i * G::NW) * 256, d.off[i] + base);
    } else {
#pragma unroll
        for (int i = 0; i < G::NSLOT; ++i) {
            if (wave + i * G::NW < G::NXI) {
                const int idx = (wave + i * G::NW) * 64 + lane;
                const int row = idx / (G::XS / 4), c4 = idx % (G::XS / 4);
                const int t = tA + 4 * c4;
                const bool ok = idx < G::XF4 && c4 < G::NCOL4 && t >= 0 && t < T;
                dma16(rx, xs + (wave + i * G::NW) * 256, ok ? (unsigned)(row * T + t) * 4u : kOutOfRange);
            }
        }
    }
}

// ---- one conv phase: NF accumulators += W (registers) x image (LDS) ---------------------------------
// wl: this wave's A fragments [S/4][64][4]; img: the lane's image base (channel row kq, column n of the
// wave's first fragment, tap 0); STRIDE / TAPSTEP: row stride and column step per tap of the image.
// K order: step s = (channel group cg = s / KT, tap = s % KT), channels 4*cg + kq inside the MFMA.
template <class G, int STRIDE, int TAPSTEP>
__device__ __forceinline__ void pair_mma(const float* wl, const float* img, f32x4 (&acc)[G::NF], int lane) {
    // The K loop runs in NH passes of SH steps (one pass at C = 16; two at C = 32, where a whole phase of
    // A operands -- 88 registers at 11 taps -- would not leave room for anything else).  Per pass:
    //   * A operands up front, SH/4 16-byte reads.  (f32x4, not HIP's float4 struct: a struct-typed LDS
    //     load makes hipcc wait for every LDS-DMA in flight before it -- the next tile's image, issued just
    //     before the conv2 phase -- which would serialise the DMA latency with the matrix work.  Landing
    //     is ordered by the explicit waits + barriers.)
    //   * B operands through a queue PF steps deep, refilled two steps at a time BEFORE the MFMAs of the
    //     steps that free the slots.  Left to itself hipcc's scheduler sinks every LDS read to just in front
    //     of its MFMA (least register pressure) and the wave then exposes the whole LDS latency every other
    //     MFMA; sched_barrier pins the order, the waitcnt pass still derives the exact lgkmcnt per use.
    //   * one base register per channel group (kept opaque so that it is not re-derived with an add in
    //     front of every read): inside a group every (tap, fragment) offset is an instruction immediate.
    constexpr int NH = G::S > 48 ? 2 : 1;
    constexpr int SH = G::S / NH;
    constexpr int PF = G::NF == 1 ? 8 : 4;
    static_assert(G::S % (4 * NH) == 0 && PF % 2 == 0 && PF <= SH, "steps are processed in pairs");
    LdsCF* cgb[G::C / 4];
#pragma unroll
    for (int c = 0; c < G::C / 4; ++c) cgb[c] = lds_opaque(img + 4 * c * STRIDE);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(wl) + lane;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        float A[SH];
#pragma unroll
        for (int g = 0; g < SH / 4; ++g) {
            const f32x4 v = w4[(h * (SH / 4) + g) * 64];
            A[4 * g] = v.x;
            A[4 * g + 1] = v.y;
            A[4 * g + 2] = v.z;
            A[4 * g + 3] = v.w;
        }
        float bq[PF][G::NF];
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int f = 0; f < G::NF; ++f) {
                const int s = h * SH + q;
                bq[q][f] = cgb[s / G::KT][(s % G::KT) * TAPSTEP + f * 16];
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < SH; s2 += 2) {
            float bv[2][G::NF];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int f = 0; f < G::NF; ++f) bv[u][f] = bq[(s2 + u) % PF][f];
            if (s2 + PF < SH) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int f = 0; f < G::NF; ++f) {
                        const int s = h * SH + s2 + u + PF;
                        bq[(s2 + u) % PF][f] = cgb[s / G::KT][(s % G::KT) * TAPSTEP + f * 16];
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int f = 0; f < G::NF; ++f)
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s2 + u], bv[u][f], acc[f], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// per-wave constants of a member's geometry
template <class G>
struct PairLane {
    const float* wa1;     // A fragments of this wave's row half, conv1 (conv2: + WF)
    const float* ximg;    // B base into the x image
    const float* mimg;    // B base into the intermediate
    float* mid_w;         // where this lane's conv1 results go: + i * MS + f * 16
    int row0;             // first of the 4 output rows of the lane (C/D layout: row = 4 * (lane >> 4) + i)
    int col0;             // column of the lane in the tile's fragment 0 of this wave
};

template <class G>
__device__ __forceinline__ PairLane<G> pair_lane(float* wl, float* xs, float* mid, int wave, int lane) {
    const int mh = wave / G::NG, ng = wave % G::NG;
    const int n = lane & 15, kq = lane >> 4;
    PairLane<G> L;
    L.wa1 = wl + mh * (G::S / 4) * 256;
    L.row0 = mh * 16 + 4 * kq;
    L.col0 = ng * (16 * G::NF) + n;
    L.ximg = xs + kq * G::XS + L.col0 + G::AOFF;
    L.mimg = mid + kq * G::MS + L.col0;
    L.mid_w = mid + L.row0 * G::MS + L.col0;
    return L;
}

// residual rows of the lane's outputs: re-read from global memory (L2 hits: the block has just staged
// the same window).  Reading them from the LDS image would need one more barrier per tile, between the
// reads and the in-place activation pass.
template <class G>
__device__ __forceinline__ void pair_residual(const PairLane<G>& L, const float* xb, int T, int t0, int nout,
                                              float (&res)[G::NF][4]) {
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)G::C * (unsigned)T * 4u);
    const unsigned t4 = (unsigned)T * 4u;
#pragma unroll
    for (int f = 0; f < G::NF; ++f) {
        const int col = L.col0 + f * 16, t = t0 + col;
        const unsigned voff = (col < nout && t < T) ? (unsigned)(L.row0 * T + t) * 4u : kOutOfRange;
#pragma unroll
        for (int i = 0; i < 4; ++i) res[f][i] = buffer_load1s(rx, voff, (unsigned)i * t4);
    }
}

// lrelu over the whole x image, in place
template <class G>
__device__ __forceinline__ void pair_activate(float* xs, float slope, int tid) {
    f32x4* x4 = reinterpret_cast<f32x4*>(xs);
    for (int idx = tid; idx < G::XF4; idx += G::NT) {
        f32x4 v = x4[idx];
        v.x = act(v.x, slope);
        v.y = act(v.y, slope);
        v.z = act(v.z, slope);
        v.w = act(v.w, slope);
        x4[idx] = v;
    }
}

// biases of a member -> LDS [b1[C] | b2[C]] (zeros when the layer has none); read back per phase with
// ds_read: a global load per tile would sit in vmcnt behind the previous tile's stores
template <class G>
__device__ __forceinline__ void pair_stage_bias(const PairMember& mb, float* bl, int tid) {
    if (tid < G::C) {
        bl[tid] = mb.b1 ? mb.b1[tid] : 0.f;
        bl[G::C + tid] = mb.b2 ? mb.b2[tid] : 0.f;
    }
}

// conv1 of one tile: x image -> intermediate image (+ b1, lrelu, zero outside the sequence)
template <class G>
__device__ __forceinline__ void pair_conv1(const PairLane<G>& L, const float* bl, float slope, int t0, int T,
                                           int lane) {
    float bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = bl[L.row0 + i];
    f32x4 acc[G::NF];
#pragma unroll
    for (int f = 0; f < G::NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    pair_mma<G, G::XS, G::DIL>(L.wa1, L.ximg, acc, lane);
    // intermediate column u of the tile is time t0 - P2 + u; conv2's zero padding applies to the
    // intermediate, so columns outside [0, T) are zero, not conv1 of the padded input
    const int tm = t0 - G::P2;
    const bool inside = tm >= 0 && tm + G::NM <= T;
    if (inside) {
#pragma unroll
        for (int f = 0; f < G::NF; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) L.mid_w[i * G::MS + f * 16] = act(acc[f][i] + bv[i], slope);
    } else {
#pragma unroll
        for (int f = 0; f < G::NF; ++f) {
            const int t = tm + L.col0 + f * 16;
            const bool ok = t >= 0 && t < T;
#pragma unroll
            for (int i = 0; i < 4; ++i) L.mid_w[i * G::MS + f * 16] = ok ? act(acc[f][i] + bv[i], slope) : 0.f;
        }
    }
}

// v / d for the divisors of the MRF mean (hifigan.py:103: xs / num_kernels, a small integer) in three instructions instead
// of the ~11 of an IEEE division: q0 = v r, e = fma(-d, q0, v) (the exact residual), q = fma(e, r, q0) with r = RN(1 / d)
// is the correctly rounded quotient for every finite v whose quotient is a normal number (Markstein; checked exhaustively
// over all 2^24 significands for the integers up to 15 on the CPU, tests/test_split_precision.py, and on the GPU against the
// hardware's own division, fv_div_probe / tests/test_gpu_parity.py).  Non-finite v: the range guard has fired anyway.
// div_rcp: 1 / d where that holds, else 0 (-> IEEE division).
__device__ __forceinline__ float div_rcp(float d) { return (d >= 1.f && d <= 16.f && d == truncf(d)) ? 1.f / d : 0.f; }
__device__ __forceinline__ float div_exact(float v, float d, float r) {
    const float q0 = v * r;
    return fmaf(fmaf(-d, q0, v), r, q0);
}

// final stores of a tile: v = post(v / out_div), y (and the activated twin)
// (rcp: div_rcp(p.out_div) computed once per run, or 0)
template <int AUX = 0>
__device__ __forceinline__ void pair_store(const PairCore& p, float* y, float* y_act, int C, int b, int row0, int t,
                                           bool ok, float (&v)[4], bool finish, float rcp = 0.f) {
    const size_t boff = (size_t)b * C * (size_t)p.T;
    const unsigned bytes = (unsigned)C * (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + boff, bytes);
    const unsigned voff = ok ? (unsigned)(row0 * p.T + t) * 4u : kOutOfRange;
    const unsigned t4 = (unsigned)p.T * 4u;
    if (finish) {
        if (p.out_div != 1.f) {
            if (rcp != 0.f) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = div_exact(v[i], p.out_div, rcp);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] / p.out_div;
            }
        }
        if (p.post == FV_POST_TANH) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = tanhf(v[i]);
        } else if (p.post == FV_POST_RELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
    }
    if (y_act) {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(y_act + boff, bytes);
#pragma unroll
        for (int i = 0; i < 4; ++i) buffer_store1s_aux<AUX>(ry, voff, (unsigned)i * t4, v[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) buffer_store1s_aux<AUX>(ra, voff, (unsigned)i * t4, act(v[i], p.act_slope));
    } else {
        if (p.act_slope != 1.f) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = act(v[i], p.act_slope);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) buffer_store1s_aux<AUX>(ry, voff, (unsigned)i * t4, v[i]);
    }
}

// ---------------------------------------------------------------------------------------------------
// plain mode: items [lo, hi) of ONE member (item = utterance * n_tiles + tile)
// ---------------------------------------------------------------------------------------------------
// items [item0, hi) of ONE member, in order (item = utterance * n_tiles + tile).
// (Tried and measured slower on MI355X, tools/pair_bench.py: drawing tiles from a global atomic queue instead
// of the static partition -- the draw sits in vmcnt behind the previous tile's stores; a start-up stagger or a
// clock-driven s_setprio flip between the two blocks of a CU; one 16-wave block per CU at C = 16.)
template <class G>
__device__ __forceinline__ void pair_run_member(const PairParams& p, const PairMember& mb, int item0, int hi,
                                                float* smem, int wave, int lane_in, int tid_in, bool first) {
    // per-lane constants are derived from an opaque copy of the lane id: otherwise the three tap-count
    // variants' sets are all computed (and kept live, or spilled) ahead of the branch that picks one
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    (void)tid_in;
    float* const wl = smem + mb.w_off;
    float* const xs = smem + p.x_off;
    float* const mid = smem + p.mid_off;
    const PairLane<G> L = pair_lane<G>(wl, xs, mid, wave, lane);
    PairDma<G> dp;
    pair_dma_plan<G>(dp, p.T, wave, lane);
    const size_t ustride = (size_t)G::C * (size_t)p.T;
    constexpr int HEAD = G::P1 + G::P2 + G::AOFF;   // the window starts HEAD columns before the tile's first output
    int item = item0;
    int b = item / mb.n_tiles, tile = item - b * mb.n_tiles;
    // a block that comes from another member: everybody is done with the LDS before it is overwritten
    if (!first) pair_barrier();
    float* const bl = smem + p.bias_off;
    pair_stage_weights<G>(mb, wl, wave, lane);
    pair_dma_x<G>(dp, mb.x + b * ustride, p.T, xs, tile * G::NOUT - HEAD, wave, lane);
    pair_stage_bias<G>(mb, bl, tid);
    pair_wait_vm0();
    pair_stamp(p, G::NW, wave, lane, 7, 13);              // weights + first image staged
    for (int it = 0;; ++it) {
        const int t0 = tile * G::NOUT;
        pair_stamp(p, G::NW, wave, lane, it, 0);
        pair_barrier();                                      // (A) the x image has landed for every wave
        pair_stamp(p, G::NW, wave, lane, it, 1);
        if (!(p.dbg & 2)) pair_activate<G>(xs, p.slope, tid);
        pair_stamp(p, G::NW, wave, lane, it, 2);
        pair_barrier();                                      // (B) activated image complete
        pair_stamp(p, G::NW, wave, lane, it, 3);
        if (!(p.dbg & 4)) pair_conv1<G>(L, bl, p.slope, t0, p.T, lane);
        pair_stamp(p, G::NW, wave, lane, it, 4);
        pair_barrier();                                      // (C) intermediate complete, x image free
        pair_stamp(p, G::NW, wave, lane, it, 5);
        const int nitem = item + 1;
        const bool more = nitem < hi;
        int nb = b, ntile = tile + 1;
        if (ntile == mb.n_tiles) {
            ntile = 0;
            ++nb;
        }
        pair_drain_vm();                                      // the previous tile's stores (issued two phases ago)
        pair_stamp(p, G::NW, wave, lane, it, 6);
        if (more && !(p.dbg & 1))
            pair_dma_x<G>(dp, mb.x + nb * ustride, p.T, xs, ntile * G::NOUT - HEAD, wave, lane);
        float bv[4], res[G::NF][4];
        pair_residual<G>(L, mb.x + b * ustride, p.T, t0, (p.dbg & 16) ? 0 : G::NOUT, res);
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = bl[G::C + L.row0 + i];
        f32x4 acc[G::NF];
#pragma unroll
        for (int f = 0; f < G::NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        pair_stamp(p, G::NW, wave, lane, it, 7);
        if (!(p.dbg & 4)) pair_mma<G, G::MS, 1>(L.wa1 + G::WF, L.mimg, acc, lane);
        pair_stamp(p, G::NW, wave, lane, it, 8);
        // the next image (issued a whole conv phase ago) before any store: loads and stores share vmcnt
        // but retire out of order with respect to each other
        pair_wait_vm0();
        pair_stamp(p, G::NW, wave, lane, it, 9);
#pragma unroll
        for (int f = 0; f < G::NF; ++f) {
            const int col = L.col0 + f * 16;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (acc[f][i] + bv[i]) + res[f][i];
            pair_store(p, mb.y, mb.y_act, G::C, b, L.row0, t0 + col,
                       col < G::NOUT && t0 + col < p.T && !(p.dbg & 8), v, false);
        }
        pair_stamp(p, G::NW, wave, lane, it, 10);
        if (!more) break;
        item = nitem;
        b = nb;
        tile = ntile;
    }
}

template <int MH, int NF, int NG, int DIL>
__device__ __forceinline__ void pair_run_any(const PairParams& p, int m, int item0, int hi, float* smem, int wave,
                                             int lane, int tid, bool first) {
    const PairMember& mb = p.m[m];
    if (mb.k == 11) pair_run_member<PairGeom<MH, NF, NG, 11, DIL>>(p, mb, item0, hi, smem, wave, lane, tid, first);
    else if (mb.k == 7) pair_run_member<PairGeom<MH, NF, NG, 7, DIL>>(p, mb, item0, hi, smem, wave, lane, tid, first);
    else pair_run_member<PairGeom<MH, NF, NG, 3, DIL>>(p, mb, item0, hi, smem, wave, lane, tid, first);
}

// contiguous, cost-balanced share of block `blk`: items of member m whose start cost
// base_m + j * cost_m falls in [blk, blk + 1) * total / nblk  (32-bit division whenever the numbers fit:
// a 64-bit division is a few hundred instructions, and every block does six of them before its first tile)
__device__ __forceinline__ int pair_share(long long blk, long long total, long long base, int cost, int n, int nblk) {
    const long long num = blk * total - base * nblk;
    if (num <= 0) return 0;
    const long long den = (long long)cost * nblk;
    const long long up = num + den - 1;
    const long long j = (up >> 32) == 0 ? (long long)((unsigned)up / (unsigned)den) : up / den;
    return j > n ? n : (int)j;
}

// equal items: block blk takes [blk n / nblk, (blk + 1) n / nblk) -- in 32 bits whenever the product fits (a 64-bit division
// is a few hundred scalar instructions between kernel entry and the first load)
__device__ __forceinline__ int equal_share(int blk, int n, int nblk) {
    return n < (1 << 21) ? (int)((unsigned)blk * (unsigned)n / (unsigned)nblk) : (int)((long long)blk * n / nblk);
}

// 16 waves per CU (4 per SIMD: <= 128 VGPRs) whatever the block size
#define FV_PAIR_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))

template <int MH, int NF, int NG, int DIL>
__global__ __launch_bounds__(64 * MH * NG) FV_PAIR_WAVES void pair_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    pair_stamp(p, MH * NG, wave, lane, 7, 15);            // kernel entry
#ifdef FV_PAIR_TRACE
    if (p.trace && tid == 0) {                             // per block: entry time, hardware ids (tuning aid)
        unsigned long long* t2 = p.trace + 8 * (MH * NG) * 8 * 16 + (size_t)blockIdx.x * 4;
        t2[0] = __builtin_amdgcn_s_memtime();
        t2[2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        t2[3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    }
#endif
    long long total = 0;
    for (int m = 0; m < p.n_members; ++m) total += (long long)p.m[m].n_tiles * p.B * p.m[m].cost;
    long long base = 0;
    bool first = true;
    for (int m = 0; m < p.n_members; ++m) {
        const int n = p.m[m].n_tiles * p.B;
        const int lo = pair_share(xcd_remap((int)blockIdx.x, (int)gridDim.x), total, base, p.m[m].cost, n, p.nblk);
        const int hi = pair_share(xcd_remap((int)blockIdx.x, (int)gridDim.x) + 1, total, base, p.m[m].cost, n, p.nblk);
        base += (long long)n * p.m[m].cost;
        if (lo >= hi) continue;
        pair_run_any<MH, NF, NG, DIL>(p, m, lo, hi, smem, wave, lane, tid, first);
        first = false;
    }
    pair_stamp(p, MH * NG, wave, lane, 7, 14);            // kernel exit
#ifdef FV_PAIR_TRACE
    if (p.trace && tid == 0) p.trace[8 * (MH * NG) * 8 * 16 + (size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
#endif
}

// ---------------------------------------------------------------------------------------------------
// sum mode: every item is one output tile; the three members' pairs (11, 7, 3 taps) accumulate into it
// ---------------------------------------------------------------------------------------------------
// One member's share of a tile: its residual and bias go straight into the accumulators (one more term
// of the same fp32 sum), the conv2 products follow.  `next`: issues the DMA of the x image that is needed
// next (the following member's, or the next tile's first) once this member's image is free.  Per-lane
// constants are re-derived here from an opaque copy of the lane id so that the three members' sets are
// not all kept live across the tile loop (the register budget is 128).
template <class G, class Next>
__device__ __forceinline__ void pair_sum_member(const PairParams& p, const PairMember& mb, float* smem,
                                                const float* bl, const float* xb, int t0, int nout, int wave,
                                                int lane_in, int tid, f32x4 (&acc)[G::NF], Next next) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    float* const xs = smem + p.x_off;
    const PairLane<G> L = pair_lane<G>(smem + mb.w_off, xs, smem + p.mid_off, wave, lane);
    pair_barrier();                                          // (A) this member's x image has landed
    pair_activate<G>(xs, p.slope, tid);
    pair_barrier();                                          // (B)
    pair_conv1<G>(L, bl, p.slope, t0, p.T, lane);
    pair_barrier();                                          // (C) intermediate complete, x image free
    pair_drain_vm();                                          // the previous tile's stores
    next();
    float rj[G::NF][4], bv[4];
    pair_residual<G>(L, xb, p.T, t0, nout, rj);
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = bl[G::C + L.row0 + i];
    pair_mma<G, G::MS, 1>(L.wa1 + G::WF, L.mimg, acc, lane);
    pair_wait_vm0();
#pragma unroll
    for (int f = 0; f < G::NF; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[f][i] += rj[f][i] + bv[i];
}

template <class G>
__device__ __forceinline__ void pair_dma_x_now(const float* xb, int T, float* xs, int tA, int wave, int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    PairDma<G> d;
    pair_dma_plan<G>(d, T, wave, lane);
    pair_dma_x<G>(d, xb, T, xs, tA, wave, lane);
}

template <int MH, int NF, int NG, int DIL>
__global__ __launch_bounds__(64 * MH * NG) FV_PAIR_WAVES void pair_sum_kernel(PairParams p) {
    typedef PairGeom<MH, NF, NG, 11, DIL> G0;
    typedef PairGeom<MH, NF, NG, 7, DIL> G1;
    typedef PairGeom<MH, NF, NG, 3, DIL> G2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const xs = smem + p.x_off;
    const int n = p.m[0].n_tiles * p.B;
    int item = (int)((long long)xcd_remap((int)blockIdx.x, (int)gridDim.x) * n / p.nblk);
    const int hi = (int)((long long)(xcd_remap((int)blockIdx.x, (int)gridDim.x) + 1) * n / p.nblk);
    if (item >= hi) return;
    const size_t ustride = (size_t)G0::C * (size_t)p.T;
    const int nout = p.n_out_sum;
    constexpr int H0 = G0::P1 + G0::P2 + G0::AOFF, H1 = G1::P1 + G1::P2 + G1::AOFF, H2 = G2::P1 + G2::P2 + G2::AOFF;
    int b = item / p.m[0].n_tiles, tile = item - b * p.m[0].n_tiles;
    pair_stage_weights<G0>(p.m[0], smem + p.m[0].w_off, wave, lane);
    pair_stage_weights<G1>(p.m[1], smem + p.m[1].w_off, wave, lane);
    pair_stage_weights<G2>(p.m[2], smem + p.m[2].w_off, wave, lane);
    float* const bl = smem + p.bias_off;
    pair_stage_bias<G0>(p.m[0], bl, tid);
    pair_stage_bias<G0>(p.m[1], bl + 2 * G0::C, tid);
    pair_stage_bias<G0>(p.m[2], bl + 4 * G0::C, tid);
    pair_dma_x_now<G0>(p.m[0].x + b * ustride, p.T, xs, tile * nout - H0, wave, lane);
    pair_wait_vm0();
    for (;;) {
        const int t0 = tile * nout;
        int nitem = item + 1, nb = b, ntile = tile + 1;
        bool more = false;
        f32x4 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        pair_sum_member<G0>(p, p.m[0], smem, bl, p.m[0].x + b * ustride, t0, nout, wave, lane, tid, acc, [&]() {
            pair_dma_x_now<G1>(p.m[1].x + b * ustride, p.T, xs, t0 - H1, wave, lane);
        });
        pair_sum_member<G1>(p, p.m[1], smem, bl + 2 * G0::C, p.m[1].x + b * ustride, t0, nout, wave, lane, tid, acc,
                            [&]() { pair_dma_x_now<G2>(p.m[2].x + b * ustride, p.T, xs, t0 - H2, wave, lane); });
        pair_sum_member<G2>(p, p.m[2], smem, bl + 4 * G0::C, p.m[2].x + b * ustride, t0, nout, wave, lane, tid, acc,
                            [&]() {
            if (ntile == p.m[0].n_tiles) {
                ntile = 0;
                ++nb;
            }
            more = nitem < hi;
            if (more) pair_dma_x_now<G0>(p.m[0].x + nb * ustride, p.T, xs, ntile * nout - H0, wave, lane);
        });
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const int row0 = (wave / NG) * 16 + 4 * (lane_s >> 4);
        const int colb = (wave % NG) * (16 * NF) + (lane_s & 15);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int col = colb + f * 16;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[f][i];
            pair_store(p, p.m[0].y, p.m[0].y_act, G0::C, b, row0, t0 + col, col < nout && t0 + col < p.T, v, true);
        }
        if (!more) break;
        item = nitem;
        b = nb;
        tile = ntile;
    }
}

}  // namespace fv
