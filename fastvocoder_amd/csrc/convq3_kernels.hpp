// Fused ResBlock1 pair at 64 channels as TWO wave groups per block, one conv phase apart ("ping-pong"):
//
//     x' = x + conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2          (reference model/generator/modules.py:223-230)
//
// convq2_kernel (convq2_kernels.hpp) walks a block's tiles with all eight waves in lock step: K loop of conv1 (matrix pipe
// only), epilogue (VALU + LDS only), barrier, K loop of conv2, barrier, epilogue + stores + conversion of the next window
// (VALU, LDS, memory) -- 44 % of a tile lies outside the K loops [profiles/r04_convq2_tile_timeline.txt], and while it runs
// the matrix pipe idles: MFMA-busy 0.49 at saturation, 0.3 at batch 1 [profiles/r05_*].  Here a block is two independent
// pipelines:
//   * group g = waves 4 g ... 4 g + 3, ONE wave per SIMD and group.  A group owns a whole tile -- all 64 rows x 64
//     intermediate columns (4 row slabs of 16 rows x one column group: convq2's 16 x 64 wave tile, A operands L2 ->
//     registers, no ring) -- with its own x image, intermediate image and bias block in LDS, and walks its own contiguous
//     share of the launch's items (the host cuts 2 nblk shares);
//   * a tile is four slots -- K1 | E1 | K2 | E2W (epilogue 2, stores, conversion of the next window) -- with a block-wide
//     s_barrier behind each, and group 1 starts ONE barrier late: whenever group 0 is in a K loop group 1 is in an epilogue
//     and vice versa, so every SIMD always holds one wave that issues MFMAs and one that issues VALU / LDS / memory
//     instructions (the two pipes run concurrently for waves of one SIMD: /opt/skills/guides/MI355X_MICROARCH.md,
//     "Wave scheduling").  A slot is as long as the longer of the two phases instead of their sum.
//   * the barriers are the block's only coupling: three of the four a group needs anyway (window image complete /
//     intermediate complete / intermediate free), the fourth (K1 | E1) only aligns the slots.  A group that runs out of
//     items leaves; s_barrier counts the waves that are still alive.
// Arithmetic per output element: convq2_kernel's (same K order, same epilogues) -- identical bits to the 128-column form
// and to the two-launch form (tests/test_gpu_pairs.py).
//
// [Measured, round 6, MI355X -- profiles/r06_pingpong.txt.]  The premise does not hold on this hardware.  A group's K loop
// alone runs at 18 cycles per MFMA (89 % of the pipe from ONE wave per SIMD), but an epilogue that takes ~2 000 cycles by
// itself takes ~6 000 beside the other group's K loop, and that K loop 6 000-7 500 instead of 4 800: on one SIMD the MFMA
// stream of one wave and the VALU / LDS instructions of another share the issue slot -- tools/pingpong_probe.hip: MFMA waves
// 271 us alone, v_fma waves 382 us alone, both together 607 us (the sum) when they share SIMDs, 490 us (the maximum) when the
// two roles sit on different SIMDs; the same for v_mfma_f32_32x32x16_f16; s_setprio either way changes nothing, and neither does
// which of the two waves is the older one (second collection of the probe: even a side wave that only issues eight
// ds_read_b128 + waits per iteration takes the sum).  A slot is the
// SUM of the two phases, and the 64-column tiles convert more halo per output column: the three-member launch of HiFi-GAN
// light's 64-channel stage takes 61-66 us against convq2_kernel's 53-57 at batch 1, 403-413 against 348-384 at batch 8.
// Not the default (Tuning::convp_pp = 0); kept, with its bit-identity test, as the measured form of the experiment.
#pragma once
#include "convq2_kernels.hpp"

namespace fv {

// window loader / converter geometry of a group's tile (convh_load_raw / convh_convert): 256 threads
template <int KT_, int DIL_>
struct ConvQ3Img {
    static constexpr int C = 64, CB = 8, NT = 256;
    static constexpr int XROWS = (64 + (KT_ - 1) * DIL_ + 3) / 4 * 4;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;   // image: [split half][8-channel block][XRP rows][8 halves]
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;
    static constexpr int NRAW = XR * 8;
};

template <int KT_, int DIL_>
struct ConvQ3Run {
    typedef ConvQ3Img<KT_, DIL_> IMG;
    static constexpr int KT = KT_, DIL = DIL_, C = 64, CG = 2, CB = 8, NT = 256;
    static constexpr int NM = 64, NOUT = NM - (KT - 1);  // intermediate / output columns per tile
    static constexpr int P1 = (KT - 1) * DIL / 2, P2 = (KT - 1) / 2;
    static constexpr int NSTEP = KT * CG, NSEQ = 2 * NSTEP;
    static constexpr int XRP = IMG::XRP, XHALF = IMG::XHALF, NRAW = IMG::NRAW;
    static constexpr int MRP = NM + 16, MHALF = CB * MRP * 16;
    static constexpr int WTILE = NSTEP * 8192, WBYTES = WTILE;
    static constexpr int NFW = 4, NSLAB = 4;
    // A operands QD K steps ahead (convq2_kernels.hpp ConvQ2Run::QD at 64 channels: QD + 1 divides the tile's step count)
    static constexpr int QD = KT_ == 3 ? 5 : KT_ == 7 ? 6 : 3;
    static constexpr int NA = 2, NRES = 4 * NFW;
    static constexpr int RAWK = NSEQ - QD;               // K step at which the next tile's window is requested
    static_assert(NSEQ % (QD + 1) == 0, "the A queue runs on from tile to tile: slot = K step % (QD + 1)");
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * 3) * 16 + 16 < 65536, "ds_read immediate range");
};

// LDS of one group at dilation DIL (bytes): x image of the widest member (11 taps), intermediate image, bias block
template <int DIL>
struct ConvQ3Lds {
    static constexpr int XIMG = 2 * ConvQ3Img<11, DIL>::XHALF;
    static constexpr int MIMG = 2 * 8 * (64 + 16) * 16;
    static constexpr int BIAS = (4 * 64 + 16) * 4;       // [b1 | b2 | inverse row prescales of conv1 | conv2 | guard scratch]
    static constexpr int OFF_X = 0, OFF_M = XIMG, OFF_B = XIMG + MIMG;
    static constexpr int GROUP = (OFF_B + BIAS + 255) / 256 * 256;
    static constexpr int TOTAL = 2 * GROUP;
    static_assert(TOTAL <= 160 * 1024, "LDS");
};

// items [item0, hi_item) of ONE member, run by ONE group (4 waves: `wave` 0 ... 3, `tid` 0 ... 255; `sm`: the group's LDS)
template <class G, class LD>
__device__ __forceinline__ void convq3_run_member(const PairParams& p, const PairMember& mb, int item0, int hi_item,
                                                  char* sm, int wave, int lane_in, int gwave) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    typedef typename G::IMG IMG;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    char* const ximg = sm + LD::OFF_X;
    char* const mimg = sm + LD::OFF_M;
    float* const bl = reinterpret_cast<float*>(sm + LD::OFF_B);
    const int n = lane & 15, kb = lane >> 4;
    const int ws = wave;                                 // row slab of 16 rows
    const int col0 = n;
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;
    const char* const mptr = mimg + (kb * G::MRP + col0) * 16;
    const int row0 = 16 * ws + 4 * kb;                   // + i
    // D fragment -> intermediate image: channels row0 + i = half of the 8-channel block 2 ws + (kb >> 1)
    char* const mw = mimg + ((2 * ws + (kb >> 1)) * G::MRP + col0) * 16 + 8 * (kb & 1);

    const size_t ustride = (size_t)G::C * (size_t)p.T;
    const unsigned ubytes = (unsigned)G::C * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(mb.w1, (unsigned)G::WBYTES);
    const __amdgpu_buffer_rsrc_t rw2 = make_rsrc(mb.w2, (unsigned)G::WBYTES);
    // packed image (fv_pack_pair_weight_ex): [K step][row sixteenth 4][split half][lane][8 halves]
    const unsigned aoff = (unsigned)(ws * 2048 + lane * 16);
    auto load_a = [&](auto SC, f16x8 (&dst)[2]) {
        constexpr int S = decltype(SC)::value % G::NSEQ;
        constexpr int step = S % G::NSTEP;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            dst[e] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(S < G::NSTEP ? rw1 : rw2, (int)aoff,
                                                                                    step * 8192 + e * 1024, 0));
    };
    const int b_last = (hi_item - 1) / mb.n_tiles;
    const int c_last = min(((hi_item - 1) - b_last * mb.n_tiles + 1) * G::NOUT, p.T);     // end of the run in the last utterance
    int b = item0 / mb.n_tiles;
    int tout = (item0 - b * mb.n_tiles) * G::NOUT;       // first output column of the tile
    int c_end = b == b_last ? c_last : p.T;
    bool warm = false;                                   // (cold / warm tiles of a run: convq2_kernels.hpp)
    LowGuard low;
    f32x2 bad2 = {0.f, 0.f};
    const float rcp = div_rcp(p.out_div);
    ConvHRaw<IMG> raw;
    convh_load_raw<IMG>(raw, mb.x + b * ustride, p.T, tout - G::P1 - G::P2, tid, true);
    f16x8 aq[G::QD + 1][2];                              // K step S sits in aq[S % (QD + 1)]
    static_for<0, G::QD>([&](auto QC) { load_a(QC, aq[decltype(QC)::value]); });
    if (tid < G::C) {
        bl[tid] = mb.b1 ? mb.b1[tid] : 0.f;
        bl[G::C + tid] = mb.b2 ? mb.b2[tid] : 0.f;
        bl[2 * G::C + tid] = mb.w1[G::WBYTES / 4 + tid];      // the rows' inverse weight prescales: behind the packed images
        bl[3 * G::C + tid] = mb.w2[G::WBYTES / 4 + tid];
    }
    pair_wait_vm0();
    if (!(p.dbg & 2)) convh_convert<IMG>(raw, ximg, p.slope, tid, low);
    for (int it = 0;; ++it) {
        const int t0 = tout;
        const int r0 = warm ? G::KT - 1 : 0;             // image row of the first NEW intermediate column
        const int n_out = warm ? G::NM : G::NOUT;
        const bool cont = t0 + n_out < c_end;            // the run goes on
        const bool nwarm = FV_WARM_TILES && cont;
        const bool more = cont || b < b_last;
        const int nb = cont ? b : b + 1;
        const int ntout = cont ? t0 + n_out : 0;
        const int nwin = ntout - G::P2 - G::P1 + (nwarm ? G::KT - 1 : 0);
        f32x4 hi[G::NFW], lo[G::NFW];
        f16x8 bbuf[2][G::NFW][2];

        LdsCF* const bb = lds_opaque(reinterpret_cast<const float*>(bptr));
        LdsCF* const bb2 = lds_opaque(reinterpret_cast<const float*>(bptr + G::XHALF));
        LdsCF* const mb1 = lds_opaque(reinterpret_cast<const float*>(mptr));
        LdsCF* const mb2 = lds_opaque(reinterpret_cast<const float*>(mptr + G::MHALF));
        // B operands of K step S: conv1 from the x image (tap stride DIL), conv2 from the intermediate (stride 1)
        auto fetch_b = [&](auto SC, f16x8 (&dst)[G::NFW][2]) {
            constexpr int S = decltype(SC)::value, step = S % G::NSTEP;
            constexpr int tap = step / G::CG, cg = step % G::CG;
#pragma unroll
            for (int e = 0; e < G::NFW; ++e) {
                if constexpr (S < G::NSTEP) {
                    constexpr int off = (cg * 4 * G::XRP + tap * G::DIL) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(bb + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(bb2 + off + e * 64);
                } else {
                    constexpr int off = (cg * 4 * G::MRP + tap) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(mb1 + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(mb2 + off + e * 64);
                }
            }
        };
        // K steps [S0, S1) of the tile's sequence (convq2_kernels.hpp: loads return in order -- step S's operands have landed
        // once at most the loads issued after them are outstanding)
        auto run = [&](auto S0C, auto S1C) {
            constexpr int S0 = decltype(S0C)::value, S1 = decltype(S1C)::value;
#pragma unroll
            for (int f = 0; f < G::NFW; ++f) hi[f] = lo[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            fetch_b(IntC<S0>{}, bbuf[S0 & 1]);
            static_for<S0, S1>([&](auto SC) {
                constexpr int S = decltype(SC)::value;
                if constexpr (S == G::RAWK)
                    convh_load_raw<IMG>(raw, mb.x + nb * ustride, p.T, nwin, tid, more && !(p.dbg & 1));
                constexpr bool raw_after = G::RAWK > S - G::QD && G::RAWK <= S;     // requested after step S's loads were
                // (the early residual sits between the A operands of conv2's first QD steps -- issued during conv1 -- and the rest)
                constexpr int res_after = S >= G::NSTEP && S < G::NSTEP + G::QD ? G::NRES : 0;
                if constexpr (S >= G::QD) wait_vm<G::NA * (G::QD - 1) + (raw_after ? G::NRAW : 0) + res_after>();
                __builtin_amdgcn_sched_barrier(0);
                load_a(IntC<S + G::QD>{}, aq[(S + G::QD) % (G::QD + 1)]);
                if constexpr (S + 1 < S1) fetch_b(IntC<S + 1>{}, bbuf[(S + 1) & 1]);
                f16x8 (&a)[2] = aq[S % (G::QD + 1)];
#pragma unroll
                for (int e = 0; e < G::NFW; ++e) hi[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bbuf[S & 1][e][0], hi[e], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < G::NFW; ++e) lo[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bbuf[S & 1][e][1], lo[e], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < G::NFW; ++e) lo[e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], bbuf[S & 1][e][0], lo[e], 0, 0, 0);
                // one load per MFMA gap instead of a burst between two steps (convq2_kernels.hpp ILV)
#pragma unroll
                for (int i = 0; i < G::NA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // buffer load
                }
                if constexpr (S + 1 < S1) {
#pragma unroll
                    for (int i = 0; i < 2 * G::NFW; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // LDS read
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * G::NFW - G::NA - 2 * G::NFW, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * G::NFW - G::NA, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        pair_stamp(p, 8, gwave, lane, it, 0);
        if (p.dbg & 64) __builtin_amdgcn_s_setprio(0);
        pair_barrier();                                  // [1] the window image is complete
        pair_stamp(p, 8, gwave, lane, it, 1);
        if (p.dbg & 128) __builtin_amdgcn_s_setprio(3);
        run(IntC<0>{}, IntC<G::NSTEP>{});
        if (p.dbg & 128) __builtin_amdgcn_s_setprio(0);
        if (p.dbg & 64) __builtin_amdgcn_s_setprio(3);
        pair_stamp(p, 8, gwave, lane, it, 2);
        pair_barrier();                                  // [2] slot boundary K1 | E1 (alignment with the other group only)
        pair_stamp(p, 8, gwave, lane, it, 3);
        {
            // conv1 -> intermediate image: row r is time t0 - P2 + r; conv2's zero padding applies to the intermediate
            const int tm = t0 - G::P2 + r0;              // time of the first new column
            const bool inside = tm >= 0 && tm + G::NM <= p.T;
            char* const mwr = mw + r0 * 16;
            float lowm = 0.f;
            const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + row0);
            const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 2 * G::C + row0);
            const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
            for (int f = 0; f < G::NFW; ++f) {
                const int t = tm + col0 + f * 16;
                f16x4 h1, h2;
                if (inside) split_mid4<false>(hi[f], lo[f], s01, s23, b01, b23, p.slope, true, h1, h2, lowm);
                else split_mid4<true>(hi[f], lo[f], s01, s23, b01, b23, p.slope, t >= 0 && t < p.T, h1, h2, lowm);
                *reinterpret_cast<f16x4*>(mwr + f * 256) = h1;
                *reinterpret_cast<f16x4*>(mwr + f * 256 + G::MHALF) = h2;
            }
            low_note(low, 1, lowm);
        }
        float res[G::NFW][4];
        unsigned voff[G::NFW];
        {
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(mb.x + b * ustride, ubytes);     // the residual is x itself
#pragma unroll
            for (int f = 0; f < G::NFW; ++f) {
                const int col = col0 + f * 16, t = t0 + col;
                voff[f] = col < n_out && t < c_end ? (unsigned)(row0 * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                for (int i = 0; i < 4; ++i) res[f][i] = buffer_load1s(rr, voff[f], (unsigned)i * t4);
            }
        }
        pair_stamp(p, 8, gwave, lane, it, 4);
        pair_barrier();                                  // [3] the intermediate is complete (and nobody reads the x image any more)
        pair_stamp(p, 8, gwave, lane, it, 5);
        if (p.dbg & 64) __builtin_amdgcn_s_setprio(0);
        if (p.dbg & 128) __builtin_amdgcn_s_setprio(3);
        run(IntC<G::NSTEP>{}, IntC<G::NSEQ>{});
        if (p.dbg & 128) __builtin_amdgcn_s_setprio(0);
        if (p.dbg & 64) __builtin_amdgcn_s_setprio(3);
        pair_stamp(p, 8, gwave, lane, it, 6);
        pair_barrier();                                  // [4] every wave of the group is done with the intermediate
        pair_stamp(p, 8, gwave, lane, it, 7);
        if (nwarm) {
            // the last KT - 1 valid columns -> the front of the image (convq2_kernels.hpp)
            constexpr int NC = 2 * G::CB * (G::KT - 1);
            if (tid < NC) {
                const int row = tid % (G::KT - 1), hb = tid / (G::KT - 1);
                char* const base = mimg + (hb / G::CB) * G::MHALF + ((hb % G::CB) * G::MRP) * 16;
                *reinterpret_cast<f16x8*>(base + row * 16) =
                    *reinterpret_cast<const f16x8*>(base + (r0 + G::NM - (G::KT - 1) + row) * 16);
            }
        }
        pair_wait_vm0();                                 // the next window, the residual, the next tile's first A operands
        pair_stamp(p, 8, gwave, lane, it, 8);
        const bool fin = mb.add1 != nullptr;
        {
            const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + G::C + row0);
            const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 3 * G::C + row0);
            const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
            for (int f = 0; f < G::NFW; ++f) combine4(hi[f], lo[f], s01, s23, b01, b23, res[f]);
        }
        if (fin) {
            const __amdgpu_buffer_rsrc_t r1 = make_rsrc(mb.add1 + b * ustride, ubytes);
            const __amdgpu_buffer_rsrc_t r2 = make_rsrc(mb.add2 ? mb.add2 + b * ustride : mb.add1, mb.add2 ? ubytes : 0u);
#pragma unroll
            for (int f = 0; f < G::NFW; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[f][i] = buffer_load1s(r1, voff[f], (unsigned)i * t4);
                    res[f][i] = buffer_load1s(r2, voff[f], (unsigned)i * t4);
                }
            pair_wait_vm0();
#pragma unroll
            for (int f = 0; f < G::NFW; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) hi[f][i] = (hi[f][i] + lo[f][i]) + res[f][i];
        }
#pragma unroll
        for (int f = 0; f < G::NFW; ++f) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = hi[f][i];
            const int col = col0 + f * 16;
            range_note4p(bad2, hi[f]);
            pair_store(p, mb.y, mb.y_act, G::C, b, row0, t0 + col, col < n_out && t0 + col < c_end && !(p.dbg & 8), v, fin, rcp);
        }
        pair_stamp(p, 8, gwave, lane, it, 9);
        if (more && !(p.dbg & 2)) convh_convert<IMG>(raw, ximg, p.slope, tid, low);
        pair_stamp(p, 8, gwave, lane, it, 10);
        if (!more) break;
        if (!cont) c_end = nb == b_last ? c_last : p.T;
        b = nb;
        tout = ntout;
        warm = nwarm;
    }
    pair_wait_vm0();
    range_flag(p, bad2.x + bad2.y);
    // the group's verdict on the low side, and the end of the member's run: ALWAYS two barriers (an even count keeps the
    // groups' slots opposite); behind the first every wave of the group is done with the bias block and the images -- the next
    // member may stage its own behind the second
    {
        unsigned* const su = reinterpret_cast<unsigned*>(bl + 4 * G::C);
        if (p.guard && lane == 0) su[wave] = low.bits;
        pair_barrier();
        if (p.guard && wave == 0 && lane == 0) {
            const unsigned all = su[0] | su[1] | su[2] | su[3];
            if (((all & 2u) && !(all & 1u)) || ((all & 8u) && !(all & 4u))) guard_raise_low(p.guard);
        }
        pair_barrier();
    }
}

// one 8-wave block per CU: two groups of four waves (one wave per SIMD and group)
template <int DIL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convq3_kernel(PairParams p) {
    typedef ConvQ3Lds<DIL> LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3;
    char* const sm = reinterpret_cast<char*>(smem) + grp * LD::GROUP;
    PairParams q;
    q.n_members = p.n_members; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = p.out_div;
    q.act_slope = p.act_slope; q.post = p.post; q.dbg = p.dbg; q.trace = p.trace; q.guard = p.guard;
    int n_items[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) n_items[m] = p.m[m].n_items;
    asm volatile("" ::"s"(q.n_members), "s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.out_div), "s"(q.act_slope),
                 "s"(q.post), "s"(q.dbg), "s"(q.trace), "s"(n_items[0]), "s"(n_items[1]), "s"(n_items[2]), "s"(q.guard));
    // the group's share of the items (members concatenated): share 2 block + group of 2 nblk (host: pair_cut_schedule)
    const int share = 2 * xcd_remap((int)blockIdx.x, (int)gridDim.x) + grp;
    int g_lo = (int)p.sched[share];
    int g_hi = share + 1 < 2 * q.nblk ? (int)p.sched[share + 1]
                                      : n_items[0] + (q.n_members > 1 ? n_items[1] : 0) + (q.n_members > 2 ? n_items[2] : 0);
    asm volatile("" ::"s"(g_lo), "s"(g_hi));
    // rows [NM, MRP) of the intermediate feed only discarded columns: finite values once (no tile writes them)
    {
        char* const mimg = sm + LD::OFF_M;
        for (int idx = tid & 255; idx < 2 * 8 * 64; idx += 256)
            reinterpret_cast<float*>(mimg + ((idx >> 6) * 80 + 64) * 16)[idx & 63] = 0.f;
    }
    if (g_lo >= g_hi) return;
    if (grp == 1) pair_barrier();                        // one slot behind group 0
    int off = 0;
    for (int m = 0; m < q.n_members; ++m) {
        const int n = m == 0 ? n_items[0] : m == 1 ? n_items[1] : n_items[2];
        const int lo = min(max(g_lo - off, 0), n), hi = min(max(g_hi - off, 0), n);
        off += n;
        if (lo >= hi) continue;
        PairMember mb;
        mb.x = p.m[m].x; mb.w1 = p.m[m].w1; mb.w2 = p.m[m].w2; mb.b1 = p.m[m].b1; mb.b2 = p.m[m].b2; mb.add1 = p.m[m].add1;
        mb.add2 = p.m[m].add2; mb.y = p.m[m].y; mb.y_act = p.m[m].y_act; mb.k = p.m[m].k; mb.n_tiles = p.m[m].n_tiles;
        asm volatile("" ::"s"(mb.x), "s"(mb.w1), "s"(mb.w2), "s"(mb.b1), "s"(mb.b2), "s"(mb.add1), "s"(mb.add2), "s"(mb.y),
                     "s"(mb.y_act), "s"(mb.k), "s"(mb.n_tiles));
        if (mb.k == 11) convq3_run_member<ConvQ3Run<11, DIL>, LD>(q, mb, lo, hi, sm, gw, lane, wave);
        else if (mb.k == 7) convq3_run_member<ConvQ3Run<7, DIL>, LD>(q, mb, lo, hi, sm, gw, lane, wave);
        else convq3_run_member<ConvQ3Run<3, DIL>, LD>(q, mb, lo, hi, sm, gw, lane, wave);
    }
}

// The same group pipeline as a block of its OWN: four waves (one per SIMD), 54 KB of LDS, two blocks per CU -- the two tile
// streams of a CU coupled by nothing (no shared barrier: a wave that waits, for memory or at its own block's barrier, leaves
// its SIMD to the other block's wave).  2 nblk blocks, share = block.
template <int DIL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void convq4_kernel(PairParams p) {
    typedef ConvQ3Lds<DIL> LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const sm = reinterpret_cast<char*>(smem);
    PairParams q;
    q.n_members = p.n_members; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = p.out_div;
    q.act_slope = p.act_slope; q.post = p.post; q.dbg = p.dbg; q.trace = p.trace; q.guard = p.guard;
    int n_items[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) n_items[m] = p.m[m].n_items;
    asm volatile("" ::"s"(q.n_members), "s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.out_div), "s"(q.act_slope),
                 "s"(q.post), "s"(q.dbg), "s"(q.trace), "s"(n_items[0]), "s"(n_items[1]), "s"(n_items[2]), "s"(q.guard));
    const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int g_lo = (int)p.sched[share];
    int g_hi = share + 1 < 2 * q.nblk ? (int)p.sched[share + 1]
                                      : n_items[0] + (q.n_members > 1 ? n_items[1] : 0) + (q.n_members > 2 ? n_items[2] : 0);
    asm volatile("" ::"s"(g_lo), "s"(g_hi));
    {
        char* const mimg = sm + LD::OFF_M;
        for (int idx = tid; idx < 2 * 8 * 64; idx += 256)
            reinterpret_cast<float*>(mimg + ((idx >> 6) * 80 + 64) * 16)[idx & 63] = 0.f;
    }
    if (g_lo >= g_hi) return;
    int off = 0;
    for (int m = 0; m < q.n_members; ++m) {
        const int n = m == 0 ? n_items[0] : m == 1 ? n_items[1] : n_items[2];
        const int lo = min(max(g_lo - off, 0), n), hi = min(max(g_hi - off, 0), n);
        off += n;
        if (lo >= hi) continue;
        PairMember mb;
        mb.x = p.m[m].x; mb.w1 = p.m[m].w1; mb.w2 = p.m[m].w2; mb.b1 = p.m[m].b1; mb.b2 = p.m[m].b2; mb.add1 = p.m[m].add1;
        mb.add2 = p.m[m].add2; mb.y = p.m[m].y; mb.y_act = p.m[m].y_act; mb.k = p.m[m].k; mb.n_tiles = p.m[m].n_tiles;
        asm volatile("" ::"s"(mb.x), "s"(mb.w1), "s"(mb.w2), "s"(mb.b1), "s"(mb.b2), "s"(mb.add1), "s"(mb.add2), "s"(mb.y),
                     "s"(mb.y_act), "s"(mb.k), "s"(mb.n_tiles));
        if (mb.k == 11) convq3_run_member<ConvQ3Run<11, DIL>, LD>(q, mb, lo, hi, sm, wave, lane, wave);
        else if (mb.k == 7) convq3_run_member<ConvQ3Run<7, DIL>, LD>(q, mb, lo, hi, sm, wave, lane, wave);
        else convq3_run_member<ConvQ3Run<3, DIL>, LD>(q, mb, lo, hi, sm, wave, lane, wave);
    }
}

}  // namespace fv
