// Fused ResBlock1 pair at 128 channels, split-f16 operands, streamed weights:
//
//     x' = x + conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2          (reference model/generator/modules.py:223-230)
//
// convp_kernels.hpp (64 channels) one size up.  What changes is the tile: a block owns ALL 128 rows of a 64-column tile
// (8 waves = 4 row slabs of 32 x 2 column groups of 32; per wave and K step the same 4 A + 4 B ds_read_b128 for 12
// MFMAs as convh / convp), because conv2 needs every channel of the intermediate and two 128-column images of 128
// channels do not fit in LDS: x image (64 + (KT - 1) DIL columns) 64 KB + intermediate (64 + 16 columns) 40 KB + a ring
// of THREE 16 KB weight stages -- a stage is ONE K step of all 128 rows here (two 8 KB pieces, one per 64-row tile of
// the packed image fv_pack_pair_weight_ex lays out for the conv kernel) -- and the staged biases: 153 KB.
// Against the two convh launches per pair it replaces: one launch, prologue, pipeline fill and tail instead of two
// (at batch 1 a 128-channel launch is bound by its longest tile and its fixed costs, not by throughput: DESIGN.md), no
// HBM round trip of the intermediate, one conversion of the x window; price: KT - 1 of a tile's 64 intermediate columns
// are recomputed by the neighbouring tile (16 % at 11 taps).  Same K order and the same split of the intermediate as the
// two-launch form: identical bits.
#pragma once
#include "convh_kernels.hpp"
#ifndef FV_WARM_TILES
#define FV_WARM_TILES 1             // 0: every tile cold (A/B builds, tools/build_variant.py)
#endif

namespace fv {

template <int KT_, int DIL_>
struct ConvQGeom {
    static constexpr int KT = KT_, DIL = DIL_, C = 128, CG = 4, CB = 16, NFW = 2, NT = 512;
    static constexpr int NM = 64;                        // intermediate columns per tile
    static constexpr int NOUT = NM - (KT - 1);           // output columns per tile
    static constexpr int P1 = (KT - 1) * DIL / 2, P2 = (KT - 1) / 2;
    static constexpr int NSTEP = KT * CG;                // K steps of 32 per conv = weight stages per conv
    static constexpr int NST = 2 * NSTEP;                // stages per tile: conv1's, then conv2's
    static constexpr int XROWS = (NM + (KT - 1) * DIL + 3) / 4 * 4;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;   // image: [split half][8-channel block][XRP rows][8 halves]
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT, XRM = XR;
    static constexpr int MRP = NM + 16;                  // rows of the intermediate image (>= NM + KT - 1: a warm tile's, multiple of 16)
    static constexpr int MHALF = CB * MRP * 16;
    // (a ring of four stages fits at dilation 1 and 3 -- smaller x image -- and was measured: no difference)
    static constexpr int STAGE_BYTES = 16384, RING = 3, AHEAD = RING - 1;
    static constexpr int WTILE = NSTEP * 8192;           // packed bytes of one 64-row tile of a conv ([tile][step][8 KB])
    static constexpr int RAWST = NST - 8;                // stage at which the next tile's raw window is requested
    // (the residual is NOT prefetched during the last stages as in convh / convp: its 16 registers would be live together
    // with the raw window, both operand queues and the accumulators -- the kernel is at the 256-register limit -- so it
    // is loaded in the epilogue, an L2 round trip per 22-27 us tile)
#ifndef FV_CONVQ_BDEPTH
#define FV_CONVQ_BDEPTH 2
#endif
    static constexpr int BD = FV_CONVQ_BDEPTH;           // B operands BD - 1 steps ahead of their MFMAs (register budget)
    static constexpr int NRAW = XRM * 8;
    static_assert(KT - 1 <= 16 && NSTEP >= 8, "taps");
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * (NFW - 1)) * 16 + 16 < 65536, "ds_read immediate range");
};

// K step `step` of a conv's packed image -> ring slot: [row tile][row sixteenth][split half][lane][8 halves], the two
// 8 KB pieces from the two 64-row tiles of the image (2 DMA instructions per wave); step_off: byte offset of the step
// inside a tile (step * 8192) or kOutOfRange
template <class G>
__device__ __forceinline__ void convq_dma_stage(__amdgpu_buffer_rsrc_t rw, float* ring, int slot, unsigned step_off, int wave,
                                                int lane) {
    float* dst = ring + slot * (G::STAGE_BYTES / 4) + wave * 512;
    const unsigned o = step_off == kOutOfRange ? kOutOfRange
                                               : step_off + (unsigned)((wave >> 2) * G::WTILE + (wave & 3) * 2048 + lane * 16);
    dma16(rw, dst, o);
    dma16(rw, dst + 256, o == kOutOfRange ? kOutOfRange : o + 1024u);
}

template <class G>
__device__ __forceinline__ void convq_run_member(const PairParams& p, const PairMember& mb, int item0, int hi_item,
                                                 float* smem, int wave, int lane_in, bool first) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    float* const ring = smem + p.x_off;
    char* const ximg = reinterpret_cast<char*>(smem + p.img_off);
    char* const mimg = reinterpret_cast<char*>(smem + p.mid_off);
    float* const bl = smem + p.bias_off;                 // [b1[128] | b2[128]]
    const int n = lane & 15, kb = lane >> 4;
    const int ws = wave >> 1, wn = wave & 1;             // row slab of 32, column group of 32
    const int col0 = wn * (16 * G::NFW) + n;
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;
    const char* const mptr = mimg + (kb * G::MRP + col0) * 16;
    // A: slot * 4096 + ((row tile * 4 + row sixteenth) * 2 + split half) * 256 floats; this wave: sixteenths 2 ws, 2 ws + 1
    const float* const aptr = ring + (2 * ws) * 512 + lane * 4;
    const int row0 = 32 * ws + 4 * kb;                   // + 16 h + i
    // D fragment -> intermediate image: channels row0 + 16 h + i = half of the 8-channel block 4 ws + 2 h + (kb >> 1)
    char* const mw = mimg + ((4 * ws + (kb >> 1)) * G::MRP + col0) * 16 + 8 * (kb & 1);

    const size_t ustride = (size_t)G::C * (size_t)p.T;
    const unsigned ubytes = (unsigned)G::C * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(mb.w1, (unsigned)(2 * G::WTILE));
    const __amdgpu_buffer_rsrc_t rw2 = make_rsrc(mb.w2, (unsigned)(2 * G::WTILE));
    int item = item0;
    int g0 = 0;                                          // ring slot of the tile's stage 0 (stage g sits in slot (g0 + g) % RING)
    auto slot_of = [&](int g) {                          // g: compile-time stage number inside the tile (or a little beyond)
        int s = g0 + g % G::RING;
        return s >= G::RING ? s - G::RING : s;
    };
    // A block's items are consecutive tiles: inside one utterance they form a RUN of columns [tout, c_end).  Its first tile
    // is "cold" -- conv1 produces the 64 intermediate columns conv2 needs for NOUT = 64 - (KT - 1) outputs, the first KT - 1
    // of them a recomputation of what the tile before (another block's) had.  Every further tile of the run is "warm": the
    // last KT - 1 intermediate columns of the tile before are still in LDS -- they are moved to the front of the image
    // (5 KB) -- conv1 produces 64 NEW columns behind them and conv2 64 outputs: no recomputation inside a run (18 % more
    // outputs per tile at 11 taps).  Every output is formed exactly as before: same bits whatever the tiling.
    const int b_last = (hi_item - 1) / mb.n_tiles;
    const int c_last = min(((hi_item - 1) - b_last * mb.n_tiles + 1) * G::NOUT, p.T);     // end of the run in the last utterance
    int b = item / mb.n_tiles;
    int tout = (item - b * mb.n_tiles) * G::NOUT;        // first output column of the tile
    int c_end = b == b_last ? c_last : p.T;
    bool warm = false;
    if (!first) pair_barrier();
    LowGuard low;                                        // low side of the range guard (pairh_kernels.hpp)
    f32x2 bad2 = {0.f, 0.f};                             // range guard (pairh_kernels.hpp range_note4p)
    const float rcp = div_rcp(p.out_div);                // the MRF mean's divisor (pair_kernels.hpp div_exact)
    ConvHRaw<G> raw;
    convh_load_raw<G>(raw, mb.x + b * ustride, p.T, tout - G::P1 - G::P2, tid, true);
#pragma unroll
    for (int st = 0; st < G::AHEAD; ++st) convq_dma_stage<G>(rw1, ring, st, (unsigned)(st * 8192), wave, lane);
    if (tid < G::C) {
        bl[tid] = mb.b1 ? mb.b1[tid] : 0.f;
        bl[G::C + tid] = mb.b2 ? mb.b2[tid] : 0.f;
        bl[2 * G::C + tid] = mb.w1[(2 * G::WTILE) / 4 + tid];      // the rows' inverse weight prescales: behind the packed images
        bl[3 * G::C + tid] = mb.w2[(2 * G::WTILE) / 4 + tid];
    }
    // rows [NM, MRP) of the intermediate feed only discarded columns: finite values once
    for (int idx = tid; idx < 2 * G::CB * 64; idx += G::NT)
        reinterpret_cast<float*>(mimg + ((idx >> 6) * G::MRP + G::NM) * 16)[idx & 63] = 0.f;
    pair_wait_vm0();
    if (!(p.dbg & 2)) convh_convert<G>(raw, ximg, p.slope, tid, low);
    for (;;) {
        const int t0 = tout;
        const int r0 = warm ? G::KT - 1 : 0;             // image row of the first NEW intermediate column (row r = time t0 - P2 + r)
        const int n_out = warm ? G::NM : G::NOUT;        // outputs of this tile
        // the next tile: warm in this run, or cold at the start of the next utterance's
        const bool cont = t0 + n_out < c_end;            // the run goes on
        const bool nwarm = FV_WARM_TILES && cont;
        const bool more = cont || b < b_last;
        const int nb = cont ? b : b + 1;
        const int ntout = cont ? t0 + n_out : 0;
        const int nwin = ntout - G::P2 - G::P1 + (nwarm ? G::KT - 1 : 0);      // first window row of the next tile
        f32x4 hi[2][G::NFW], lo[2][G::NFW];
        float res[2][G::NFW][4];
        unsigned voff[G::NFW];
        f16x8 abuf[2][2][2], bbuf[G::BD][2][2];

        // ---- stage entry: the stage's weights are in its ring slot for every wave; every wave holds the A operands of
        // the stage before in registers, so that slot is free: request the stage AHEAD further on into it
        auto entry = [&](auto GC) {
            constexpr int GS = decltype(GC)::value;
            {
                // this stage's DMA was issued AHEAD entries ago; loads return in order: it has landed once at most as many
                // loads are outstanding as were issued after it -- the DMAs of the entries in between, plus the raw window
                // / the residual if they were requested at one of those entries (for GS < AHEAD the DMA was issued in the
                // previous tile, whose raw / residual requests were waited for in its epilogue)
                constexpr bool raw_between = GS >= G::AHEAD && G::RAWST >= GS - G::AHEAD && G::RAWST <= GS - 1;
                // (GS < AHEAD: landed before that epilogue issued its stores -- no count, it would wait for the stores)
                if constexpr (GS >= G::AHEAD) wait_vm<2 * (G::AHEAD - 1) + (raw_between ? G::NRAW : 0)>();
            }
            pair_barrier();
            constexpr int NS = GS + G::AHEAD;            // this tile's stage NS, or the next tile's NS - NST
            if constexpr (NS < G::NSTEP)
                convq_dma_stage<G>(rw1, ring, slot_of(NS), (unsigned)(NS * 8192), wave, lane);
            else if constexpr (NS < G::NST)
                convq_dma_stage<G>(rw2, ring, slot_of(NS), (unsigned)((NS - G::NSTEP) * 8192), wave, lane);
            else
                convq_dma_stage<G>(rw1, ring, slot_of(NS), more ? (unsigned)((NS - G::NST) * 8192) : kOutOfRange, wave, lane);
            if constexpr (GS == G::RAWST)
                convh_load_raw<G>(raw, mb.x + nb * ustride, p.T, nwin, tid, more && !(p.dbg & 1));
        };
        auto fetch_a = [&](auto SC, f16x8 (&dst)[2][2]) {        // SC: stage of the tile (= step of its 2 x NSTEP sequence)
            constexpr int S = decltype(SC)::value;
            LdsCF* a = lds_opaque(aptr + slot_of(S) * (G::STAGE_BYTES / 4));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                dst[h][0] = *reinterpret_cast<LdsH8*>(a + h * 512);
                dst[h][1] = *reinterpret_cast<LdsH8*>(a + h * 512 + 256);
            }
        };
        LdsCF* const bb = lds_opaque(reinterpret_cast<const float*>(bptr));
        LdsCF* const bb2 = lds_opaque(reinterpret_cast<const float*>(bptr + G::XHALF));
        LdsCF* const mb1 = lds_opaque(reinterpret_cast<const float*>(mptr));
        LdsCF* const mb2 = lds_opaque(reinterpret_cast<const float*>(mptr + G::MHALF));
        // B operands of step S of conv CV (0: from the x image, tap stride DIL; 1: from the intermediate, stride 1)
        auto fetch_b = [&](auto CVC, auto SC, f16x8 (&dst)[2][2]) {
            constexpr int CV = decltype(CVC)::value, S = decltype(SC)::value;
            constexpr int tap = S / G::CG, cg = S % G::CG;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if constexpr (CV == 0) {
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
        // one conv: NSTEP steps; CV selects the B image, the stage numbers continue across the two convs
        auto conv = [&](auto CVC) {
            constexpr int CV = decltype(CVC)::value;
            constexpr int S0 = CV * G::NSTEP;                // first stage of this conv in the tile's sequence
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (CV == 0) entry(IntC<0>{});         // conv2's first stage was entered during conv1's last step
            fetch_a(IntC<S0>{}, abuf[S0 & 1]);
            fetch_b(CVC, IntC<0>{}, bbuf[0]);
            if constexpr (G::BD > 2) fetch_b(CVC, IntC<1>{}, bbuf[1]);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, G::NSTEP>([&](auto UC) {
                constexpr int U = decltype(UC)::value;       // step inside this conv
                constexpr int S = S0 + U, SN = S + 1;
                if constexpr (SN < G::NST) {
                    // the next step is a new stage (also across the conv1 -> conv2 boundary): take its barrier now, then
                    // prefetch its A operands (conv2's first: after the intermediate is complete, below)
                    entry(IntC<SN>{});
                    if constexpr (U + 1 < G::NSTEP) fetch_a(IntC<SN>{}, abuf[SN & 1]);
                }
                if constexpr (U + G::BD - 1 < G::NSTEP) fetch_b(CVC, IntC<U + G::BD - 1>{}, bbuf[(U + G::BD - 1) % G::BD]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][0], bbuf[U % G::BD][e][0], hi[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][0], bbuf[U % G::BD][e][1], lo[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][1], bbuf[U % G::BD][e][0], lo[h][e], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        conv(IntC<0>{});
        {
            // conv1 -> intermediate image: column u of the tile is time t0 - P2 + u; conv2's zero padding applies to
            // the intermediate: columns outside [0, T) are zero, not conv1 of the padded input
            const int tm = t0 - G::P2 + r0;              // time of the first new column
            const bool inside = tm >= 0 && tm + G::NM <= p.T;     // (uniform) no column of this tile needs the mask
            char* const mwr = mw + r0 * 16;
            float lowm = 0.f;                            // largest magnitude of this tile's intermediate in this lane
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + row0 + 16 * h);
                const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 2 * G::C + row0 + 16 * h);
                const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    const int t = tm + col0 + f * 16;
                    f16x4 h1, h2;
                    if (inside) split_mid4<false>(hi[h][f], lo[h][f], s01, s23, b01, b23, p.slope, true, h1, h2, lowm);
                    else split_mid4<true>(hi[h][f], lo[h][f], s01, s23, b01, b23, p.slope, t >= 0 && t < p.T, h1, h2, lowm);
                    *reinterpret_cast<f16x4*>(mwr + f * 256 + h * (2 * G::MRP * 16)) = h1;
                    *reinterpret_cast<f16x4*>(mwr + f * 256 + h * (2 * G::MRP * 16) + G::MHALF) = h2;
                }
            }
            low_note(low, 1, lowm);
        }
        pair_barrier();                                  // the intermediate is complete (and nobody reads the x image any more)
        conv(IntC<1>{});
        // ---- epilogue: outputs, then the image of the next window ----------------------------------------------
        pair_barrier();                                  // every wave is done with the intermediate
        if (nwarm) {
            // the last KT - 1 valid columns -> the front of the image (rows [r0 + NM - (KT - 1), r0 + NM) -> [0, KT - 1));
            // the next tile's conv1 writes rows [KT - 1, KT - 1 + NM) many barriers from here, conv2 reads after its own
            constexpr int NC = 2 * G::CB * (G::KT - 1);
            if (tid < NC) {
                const int row = tid % (G::KT - 1), hb = tid / (G::KT - 1);      // hb: (split half, 8-channel block)
                char* const base = mimg + (hb / G::CB) * G::MHALF + ((hb % G::CB) * G::MRP) * 16;
                *reinterpret_cast<f16x8*>(base + row * 16) =
                    *reinterpret_cast<const f16x8*>(base + (r0 + G::NM - (G::KT - 1) + row) * 16);
            }
        }
        {
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(mb.x + b * ustride, ubytes);     // the residual is x itself
#pragma unroll
            for (int f = 0; f < G::NFW; ++f) {
                const int col = col0 + f * 16, t = t0 + col;
                voff[f] = col < n_out && t < c_end ? (unsigned)(row0 * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) res[h][f][i] = buffer_load1s(rr, voff[f], (unsigned)(16 * h + i) * t4);
            }
        }
        pair_wait_vm0();                                 // raw window, residual (and the last stage requests)
        const bool fin = mb.add1 != nullptr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + G::C + row0 + 16 * h);
            const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 3 * G::C + row0 + 16 * h);
            const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
            for (int f = 0; f < G::NFW; ++f) combine4(hi[h][f], lo[h][f], s01, s23, b01, b23, res[h][f]);
        }
        if (fin) {
            const __amdgpu_buffer_rsrc_t r1 = make_rsrc(mb.add1 + b * ustride, ubytes);
            const __amdgpu_buffer_rsrc_t r2 = make_rsrc(mb.add2 ? mb.add2 + b * ustride : mb.add1, mb.add2 ? ubytes : 0u);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        lo[h][f][i] = buffer_load1s(r1, voff[f], (unsigned)(16 * h + i) * t4);
                        res[h][f][i] = buffer_load1s(r2, voff[f], (unsigned)(16 * h + i) * t4);
                    }
            pair_wait_vm0();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hi[h][f][i] = (hi[h][f][i] + lo[h][f][i]) + res[h][f][i];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int f = 0; f < G::NFW; ++f) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = hi[h][f][i];
                const int col = col0 + f * 16;
                range_note4p(bad2, hi[h][f]);         // (every column is computed from real, zero-padded data)
                pair_store(p, mb.y, mb.y_act, G::C, b, row0 + 16 * h, t0 + col,
                           col < n_out && t0 + col < c_end && !(p.dbg & 8), v, fin, rcp);
            }
        if (more && !(p.dbg & 2)) convh_convert<G>(raw, ximg, p.slope, tid, low);
        if (!more) break;
        g0 = slot_of(G::NST);
        if (!cont) c_end = nb == b_last ? c_last : p.T;
        b = nb;
        tout = ntout;
        warm = nwarm;
    }
    pair_wait_vm0();
    range_flag(p, bad2.x + bad2.y);
    low_flag(p, low, bl + 4 * G::C, wave, lane, 8);
}

// one 8-wave block per CU (153 KB of LDS), 2 waves per SIMD
template <int DIL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convq_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the launch's scalars in one batch of kernarg loads (see convh_kernel)
    PairParams q;
    q.n_members = p.n_members; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = p.out_div;
    q.act_slope = p.act_slope; q.post = p.post; q.x_off = p.x_off; q.img_off = p.img_off; q.mid_off = p.mid_off;
    q.bias_off = p.bias_off; q.dbg = p.dbg; q.trace = p.trace; q.guard = p.guard;
    int n_items[3], cost[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) { n_items[m] = p.m[m].n_items; cost[m] = p.m[m].cost; }
    asm volatile("" ::"s"(q.n_members), "s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.out_div), "s"(q.act_slope),
                 "s"(q.post), "s"(q.x_off), "s"(q.img_off), "s"(q.mid_off), "s"(q.bias_off), "s"(q.dbg), "s"(q.trace),
                 "s"(n_items[0]), "s"(n_items[1]), "s"(n_items[2]), "s"(cost[0]), "s"(cost[1]), "s"(cost[2]), "s"(q.guard));
    const bool sched = p.sched_on == 1, cut = p.sched_on == 2;      // 2: the contiguous cut as a table (pair_cut_schedule)
    int slo[3] = {0, 0, 0}, shi[3] = {0, 0, 0};
    int g_lo = 0, g_hi = 0;
    if (cut) {
        const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
        g_lo = (int)p.sched[share];
        g_hi = share + 1 < q.nblk ? (int)p.sched[share + 1] : n_items[0] + (q.n_members > 1 ? n_items[1] : 0) + (q.n_members > 2 ? n_items[2] : 0);
        asm volatile("" ::"s"(g_lo), "s"(g_hi));
    }
    if (sched) {
        // two words of the kernel arguments per block: (lo : 11, count : 5) of member 0 | member 1 << 16, member 2
        const unsigned w0 = p.sched[2 * xcd_remap((int)blockIdx.x, (int)gridDim.x)], w1 = p.sched[2 * xcd_remap((int)blockIdx.x, (int)gridDim.x) + 1];
        slo[0] = (int)(w0 & 2047u);         shi[0] = slo[0] + (int)((w0 >> 11) & 31u);
        slo[1] = (int)((w0 >> 16) & 2047u); shi[1] = slo[1] + (int)(w0 >> 27);
        slo[2] = (int)(w1 & 2047u);         shi[2] = slo[2] + (int)((w1 >> 11) & 31u);
        asm volatile("" ::"s"(slo[0]), "s"(shi[0]), "s"(slo[1]), "s"(shi[1]), "s"(slo[2]), "s"(shi[2]));
    }
    long long total = 0;
    if (!sched && !cut) {
#pragma unroll
        for (int m = 0; m < 3; ++m) total += m < q.n_members ? (long long)n_items[m] * cost[m] : 0;
    }
    long long base = 0;
    int off = 0;
    bool first = true;
    for (int m = 0; m < q.n_members; ++m) {
        const int n = m == 0 ? n_items[0] : m == 1 ? n_items[1] : n_items[2];
        const int cm = m == 0 ? cost[0] : m == 1 ? cost[1] : cost[2];
        int lo, hi;
        if (sched) {
            lo = m == 0 ? slo[0] : m == 1 ? slo[1] : slo[2];
            hi = m == 0 ? shi[0] : m == 1 ? shi[1] : shi[2];
        } else if (cut) {
            lo = min(max(g_lo - off, 0), n);
            hi = min(max(g_hi - off, 0), n);
            off += n;
        } else {
            lo = pair_share(xcd_remap((int)blockIdx.x, (int)gridDim.x), total, base, cm, n, q.nblk);
            hi = pair_share(xcd_remap((int)blockIdx.x, (int)gridDim.x) + 1, total, base, cm, n, q.nblk);
            base += (long long)n * cm;
        }
        if (lo >= hi) continue;
        PairMember mb;
        mb.x = p.m[m].x; mb.w1 = p.m[m].w1; mb.w2 = p.m[m].w2; mb.b1 = p.m[m].b1; mb.b2 = p.m[m].b2; mb.add1 = p.m[m].add1;
        mb.add2 = p.m[m].add2; mb.y = p.m[m].y; mb.y_act = p.m[m].y_act; mb.k = p.m[m].k; mb.n_tiles = p.m[m].n_tiles;
        asm volatile("" ::"s"(mb.x), "s"(mb.w1), "s"(mb.w2), "s"(mb.b1), "s"(mb.b2), "s"(mb.add1), "s"(mb.add2), "s"(mb.y),
                     "s"(mb.y_act), "s"(mb.k), "s"(mb.n_tiles));
        if (mb.k == 11) convq_run_member<ConvQGeom<11, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else if (mb.k == 7) convq_run_member<ConvQGeom<7, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else convq_run_member<ConvQGeom<3, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        first = false;
    }
}

}  // namespace fv
