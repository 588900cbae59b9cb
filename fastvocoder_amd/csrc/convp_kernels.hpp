// Fused ResBlock1 pair at 64 channels, split-f16 operands, streamed weights:
//
//     x' = x + conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2          (reference model/generator/modules.py:223-230)
//
// convh_kernels.hpp run twice without leaving the CU: a tile walks ONE linear sequence of 2 KT weight stages (conv1's,
// then conv2's) through the same 4-slot LDS ring; after conv1's last stage the accumulators become the split image of
// the intermediate (+ b1, lrelu, zero outside [0, T)) in LDS, conv2 reads its B operands from there.  Against two
// convh launches per pair: one prologue / pipeline fill / tail instead of two, the intermediate never touches HBM,
// the x window is converted once; price: KT - 1 of the tile's 128 intermediate columns are recomputed by the
// neighbouring tile (8 % at 11 taps).  LDS: ring 64 KB + x image 48 KB + intermediate image 36 KB.  At 128 channels
// the two images alone would be 172 KB: those pairs stay two launches.
#pragma once
#include "convh_kernels.hpp"
#ifndef FV_WARM_TILES
#define FV_WARM_TILES 1             // 0: every tile cold (A/B builds, tools/build_variant.py)
#endif

namespace fv {

template <int KT_, int DIL_>
struct ConvPGeom {
    typedef ConvHGeom<2, 2, KT_, DIL_> H;                // the conv1 side: same image, ring, wave layout
    static constexpr int KT = KT_, DIL = DIL_, C = 64, NFW = 2;
    static constexpr int NM = H::NTC;                    // 128 intermediate columns per tile
    static constexpr int NOUT = NM - (KT - 1);           // output columns per tile
    static constexpr int P1 = (KT - 1) * DIL / 2, P2 = (KT - 1) / 2;
    static constexpr int NST1 = H::NST, NST = 2 * NST1;  // weight stages per tile: conv1's, then conv2's
    static constexpr int NUNIT1 = H::NUNIT;              // MFMA groups per conv
    static constexpr int MRP = NM + 16;                  // rows of the intermediate image (>= NM + KT - 1, multiple of 16)
    static constexpr int MHALF = (C / 8) * MRP * 16;
    static constexpr int RAWST = NST - 4, RESST = NST - 2;
    static constexpr int NRAW = H::NRAW, NRES = 8 * NFW;
    static_assert(KT - 1 <= 16 && NST1 >= 3, "taps");
};

// CHAIN: a phase of a chained launch (fv_internal.h PairChain): activations move with agent-scope accesses, an item
// waits for the flags of the tiles it reads (`dep`: the member's dependency descriptors, in the kernel arguments) and
// raises its own once its stores are acknowledged.
template <class G, bool CHAIN = false>
__device__ __forceinline__ void convp_run_member(const PairCore& p, const PairMember& mb, int item0, int hi_item,
                                                 float* smem, int wave, int lane_in, bool first,
                                                 const PairMember::Dep* dep = nullptr, const ChainCtx* cc = nullptr,
                                                 int dir_in = 1) {
    // items item0, item0 + dir, ... up to (not including) hi_item; dir = -1 in chained launches only (the block
    // schedule alternates the direction from phase to phase: fv_internal.h PairChain)
    const int dir = CHAIN ? dir_in : 1;
    typedef typename G::H H;
#ifdef FV_CHAIN_PLAIN
    constexpr int AUX = 0;                               // (timing experiment: results are not coherent)
#else
    constexpr int AUX = CHAIN ? kAuxAgent : 0;
#endif
    // stage entries at which the tile BEFORE is signalled (its stores were issued a window conversion ago: vmcnt(0) is
    // nearly free there) and at which the next item's flags are requested (three entries ahead of their use)
    constexpr int SIGST = 0, FLST = G::RAWST >= 3 ? G::RAWST - 3 : 0;
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    float* const ring = smem + p.x_off;
    char* const ximg = reinterpret_cast<char*>(smem + p.img_off);
    char* const mimg = reinterpret_cast<char*>(smem + p.mid_off);
    float* const bl = smem + p.bias_off;                 // [b1[64] | b2[64]]
    const int n = lane & 15, kb = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int col0 = wn * (16 * G::NFW) + n;
    const char* const bptr = ximg + (kb * H::XRP + col0) * 16;
    const char* const mptr = mimg + (kb * G::MRP + col0) * 16;
    const float* const aptr = ring + (wm * 2) * 512 + lane * 4;
    const int row0 = 16 * (2 * wm) + 4 * kb;             // + 16 h + i
    // D fragment -> intermediate image: channels row0 + 16 h + i = half of the 8-channel block 2 (2 wm + h) + (kb >> 1)
    char* const mw = mimg + ((2 * (2 * wm) + (kb >> 1)) * G::MRP + col0) * 16 + 8 * (kb & 1);

    const size_t ustride = (size_t)G::C * (size_t)p.T;
    const unsigned ubytes = (unsigned)G::C * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(mb.w1, (unsigned)H::WTILE);
    const __amdgpu_buffer_rsrc_t rw2 = make_rsrc(mb.w2, (unsigned)H::WTILE);
    int item = item0;
    int g0 = 0;
    int b = item / mb.n_tiles, tile = item - b * mb.n_tiles;
    // Runs of consecutive tiles inside one utterance: the first tile is cold (NOUT outputs, KT - 1 intermediate columns
    // recomputed), the others warm -- the last KT - 1 intermediate columns of the tile before move to the front of the image,
    // conv1 produces NM new ones, conv2 NM outputs (convq_kernels.hpp; not in chained launches, whose flags count cold tiles)
    constexpr bool WARM = !CHAIN;                        // (the run logic; FV_WARM_TILES = 0 keeps every tile of a run cold)
    const int b_last = (hi_item - 1) / mb.n_tiles;
    const int c_last = min(((hi_item - 1) - b_last * mb.n_tiles + 1) * G::NOUT, p.T);
    int tout = tile * G::NOUT;
    int c_end = !WARM || b != b_last ? p.T : c_last;
    bool warm = false;
    if (!first) pair_barrier();
    pair_stamp(p, 8, wave, lane, 7, 12);                 // (tuning aid, -DFV_PAIR_TRACE: tools/convp_trace.py) run start
    LowGuard low;                                        // low side of the range guard (pairh_kernels.hpp)
    f32x2 bad2 = {0.f, 0.f};                             // range guard (pairh_kernels.hpp range_note4p)
    const float rcp = div_rcp(p.out_div);                // the MRF mean's divisor (pair_kernels.hpp div_exact)
    ConvHRaw<H> raw;
    unsigned fo = kOutOfRange, fv = 0;                   // chained: this lane's flag of the next item, its value
    if constexpr (CHAIN) {
        fo = chain_dep_offset(dep, b, tile * G::NOUT, G::NOUT, p.T, lane);
        fv = chain_load(*cc, fo);
        if (!(p.dbg & 64) && !chain_ready(*cc, fo, fv)) chain_spin(*cc, fo, lane);
    }
    convh_load_raw<H, AUX>(raw, mb.x + b * ustride, p.T, tile * G::NOUT - G::P1 - G::P2, tid, true);
#pragma unroll
    for (int st = 0; st < 3; ++st) convh_dma_stage<H>(rw1, ring, st, (unsigned)(st * H::STAGE_BYTES), wave, lane);
    if (tid < G::C) {
        bl[tid] = mb.b1 ? mb.b1[tid] : 0.f;
        bl[G::C + tid] = mb.b2 ? mb.b2[tid] : 0.f;
        bl[2 * G::C + tid] = mb.w1[(H::WTILE) / 4 + tid];      // the rows' inverse weight prescales: behind the packed images
        bl[3 * G::C + tid] = mb.w2[(H::WTILE) / 4 + tid];
    }
    // rows [NM, MRP) of the intermediate feed only discarded columns: finite values once
    for (int idx = tid; idx < 2 * (G::C / 8) * 64; idx += 512)
        reinterpret_cast<float*>(mimg + ((idx >> 6) * G::MRP + G::NM) * 16)[idx & 63] = 0.f;
    pair_wait_vm0();
    pair_stamp(p, 8, wave, lane, 7, 10);
    if (!(p.dbg & 2)) convh_convert<H>(raw, ximg, p.slope, tid, low);
    pair_stamp(p, 8, wave, lane, 7, 13);
    for (int it = 0;; ++it) {
        pair_stamp(p, 8, wave, lane, it, 0);
        const int t0 = WARM ? tout : tile * G::NOUT;
        const int nitem = item + dir;
        int nb = b, ntile = tile + dir;
        if (ntile == mb.n_tiles) {
            ntile = 0;
            ++nb;
        }
        if constexpr (CHAIN) {
            if (ntile < 0) {
                ntile = mb.n_tiles - 1;
                --nb;
            }
        }
        const int r0 = WARM && warm ? G::KT - 1 : 0;     // image row of the first NEW intermediate column
        const int n_out = WARM && warm ? G::NM : G::NOUT;
        const bool cont = WARM && t0 + n_out < c_end;    // the next tile continues this run
        const bool nwarm = FV_WARM_TILES && cont;
        const bool more = WARM ? (cont || b < b_last) : nitem != hi_item;
        if constexpr (WARM) nb = cont ? b : b + 1;
        const int ntout = cont ? t0 + n_out : 0;
        const int nwin = WARM ? ntout - G::P1 - G::P2 + (nwarm ? G::KT - 1 : 0) : ntile * G::NOUT - G::P1 - G::P2;
        f32x4 hi[2][G::NFW], lo[2][G::NFW];
        float res[2][G::NFW][4];
        unsigned voff[G::NFW];
        f16x8 abuf[2][2][2], bbuf[3][2][2];

        auto entry = [&](auto GC) {
            constexpr int GS = decltype(GC)::value;
            {
                constexpr bool raw_between = GS >= 3 && G::RAWST >= GS - 3 && G::RAWST <= GS - 1;
                constexpr bool res_between = GS >= 3 && G::RESST >= GS - 3 && G::RESST <= GS - 1;
                constexpr bool flag_between = CHAIN && FLST >= GS - 3 && FLST <= GS - 1;
                // (the first three stages of a tile landed before the tile in front issued its stores -- its epilogue
                // waits for them: a count here would wait for those stores, which are younger and retire late)
                if constexpr (CHAIN && GS == SIGST) wait_vm<0>();       // ... and the stores of the tile before
                else if constexpr (GS >= 3)
                    wait_vm<4 + (raw_between ? G::NRAW : 0) + (res_between ? G::NRES : 0) + (flag_between ? 1 : 0)>();
            }
            pair_barrier();
            if constexpr (CHAIN && GS == SIGST) {
                if (tid == 0 && item != item0) chain_signal(*cc, mb.flag_off + item - dir);
            }
            constexpr int NS = GS + 3;                   // this tile's stage NS, or the next tile's NS - NST
            if constexpr (NS < G::NST1)
                convh_dma_stage<H>(rw1, ring, (g0 + NS) & 3, (unsigned)(NS * H::STAGE_BYTES), wave, lane);
            else if constexpr (NS < G::NST)
                convh_dma_stage<H>(rw2, ring, (g0 + NS) & 3, (unsigned)((NS - G::NST1) * H::STAGE_BYTES), wave, lane);
            else
                convh_dma_stage<H>(rw1, ring, (g0 + NS) & 3,
                                   more ? (unsigned)((NS - G::NST) * H::STAGE_BYTES) : kOutOfRange, wave, lane);
            if constexpr (CHAIN && GS == FLST) {
                fo = more ? chain_dep_offset(dep, nb, ntile * G::NOUT, G::NOUT, p.T, lane) : kOutOfRange;
                fv = chain_load(*cc, fo);
            }
            if constexpr (GS == G::RAWST) {
                if constexpr (CHAIN) {
                    if (!(p.dbg & 64) && !chain_ready(*cc, fo, fv)) chain_spin(*cc, fo, lane);
                }
                convh_load_raw<H, AUX>(raw, mb.x + nb * ustride, p.T, nwin, tid, more && !(p.dbg & 1));
            }
            if constexpr (GS == G::RESST) {
                const __amdgpu_buffer_rsrc_t rr = make_rsrc(mb.x + b * ustride, ubytes);     // the residual is x itself
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    const int col = col0 + f * 16, t = t0 + col;
                    voff[f] = col < n_out && t < c_end ? (unsigned)(row0 * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) res[h][f][i] = buffer_load1s_aux<AUX>(rr, voff[f], (unsigned)(16 * h + i) * t4);
                }
            }
        };
        auto fetch_a = [&](auto SC, f16x8 (&dst)[2][2]) {        // SC: step of the tile's 2 x NSTEP sequence
            constexpr int S = decltype(SC)::value;
            LdsCF* a = lds_opaque(aptr + ((g0 + S / 2) & 3) * (H::STAGE_BYTES / 4));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                dst[h][0] = *reinterpret_cast<LdsH8*>(a + ((S % 2) * 4 + h) * 512);
                dst[h][1] = *reinterpret_cast<LdsH8*>(a + ((S % 2) * 4 + h) * 512 + 256);
            }
        };
        LdsCF* const bb = lds_opaque(reinterpret_cast<const float*>(bptr));
        LdsCF* const bb2 = lds_opaque(reinterpret_cast<const float*>(bptr + H::XHALF));
        LdsCF* const mb1 = lds_opaque(reinterpret_cast<const float*>(mptr));
        LdsCF* const mb2 = lds_opaque(reinterpret_cast<const float*>(mptr + G::MHALF));
        // B operands of step S of conv CV (0: from the x image, tap stride DIL; 1: from the intermediate, stride 1)
        auto fetch_b = [&](auto CVC, auto SC, f16x8 (&dst)[2][2]) {
            constexpr int CV = decltype(CVC)::value, S = decltype(SC)::value;
            constexpr int tap = S / 2, cg = S % 2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if constexpr (CV == 0) {
                    constexpr int off = (cg * 4 * H::XRP + tap * G::DIL) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(bb + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(bb2 + off + e * 64);
                } else {
                    constexpr int off = (cg * 4 * G::MRP + tap) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(mb1 + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(mb2 + off + e * 64);
                }
            }
        };
        // one conv: NUNIT1 groups (NP = 1: group = step); CV selects the B image, the stage numbers continue
        auto conv = [&](auto CVC) {
            constexpr int CV = decltype(CVC)::value;
            constexpr int S0 = CV * H::NSTEP;                // first step of this conv in the tile's sequence
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (CV == 0) entry(IntC<0>{});         // conv2's first stage was entered during conv1's last group
            fetch_a(IntC<S0>{}, abuf[S0 & 1]);
            fetch_b(CVC, IntC<0>{}, bbuf[0]);
            fetch_b(CVC, IntC<1>{}, bbuf[1]);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, G::NUNIT1>([&](auto UC) {
                constexpr int U = decltype(UC)::value;       // group = step inside this conv
                constexpr int S = S0 + U, SN = S + 1;
                if constexpr (SN < 2 * H::NSTEP) {
                    // the next step starts a new stage (also across the conv1 -> conv2 boundary: the ring does not care)
                    if constexpr (SN % 2 == 0) entry(IntC<SN / 2>{});
                    if constexpr (U + 1 < G::NUNIT1) fetch_a(IntC<SN>{}, abuf[SN & 1]);
                }
                if constexpr (U + 2 < G::NUNIT1) fetch_b(CVC, IntC<U + 2>{}, bbuf[(U + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][0], bbuf[U % 3][e][0], hi[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][0], bbuf[U % 3][e][1], lo[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][1], bbuf[U % 3][e][0], lo[h][e], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        conv(IntC<0>{});
        pair_stamp(p, 8, wave, lane, it, 1);
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
        pair_stamp(p, 8, wave, lane, it, 2);
        // fetch_a of conv2's first step: its stage was entered (barrier, DMA landed) during conv1's last group
        conv(IntC<1>{});
        pair_stamp(p, 8, wave, lane, it, 3);
        // ---- epilogue: outputs, then the image of the next window ----------------------------------------------
        pair_barrier();                                  // every wave is done with the intermediate
        if (nwarm) {
            // the last KT - 1 valid columns -> the front of the image, for the warm tile that follows
            constexpr int NC = 2 * (G::C / 8) * (G::KT - 1);
            if (tid < NC) {
                const int row = tid % (G::KT - 1), hb = tid / (G::KT - 1);      // hb: (split half, 8-channel block)
                char* const base = mimg + (hb / (G::C / 8)) * G::MHALF + ((hb % (G::C / 8)) * G::MRP) * 16;
                *reinterpret_cast<f16x8*>(base + row * 16) =
                    *reinterpret_cast<const f16x8*>(base + (r0 + G::NM - (G::KT - 1) + row) * 16);
            }
        }
        wait_vm<0>();                                    // raw window, residual, the next tile's first three weight stages
        pair_stamp(p, 8, wave, lane, it, 4);
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
                        lo[h][f][i] = buffer_load1s_aux<AUX>(r1, voff[f], (unsigned)(16 * h + i) * t4);
                        res[h][f][i] = buffer_load1s_aux<AUX>(r2, voff[f], (unsigned)(16 * h + i) * t4);
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
                pair_store<AUX>(p, mb.y, mb.y_act, G::C, b, row0 + 16 * h, t0 + col,
                           col < n_out && t0 + col < c_end && !(p.dbg & 8), v, fin, rcp);
            }
        pair_stamp(p, 8, wave, lane, it, 5);
        if (more && !(p.dbg & 2)) convh_convert<H>(raw, ximg, p.slope, tid, low);
        pair_stamp(p, 8, wave, lane, it, 6);
        if (!more) break;
        g0 += G::NST;
        if (WARM && !cont) c_end = nb == b_last ? c_last : p.T;
        item = nitem;
        b = nb;
        tile = ntile;
        tout = ntout;
        warm = nwarm;
    }
    pair_wait_vm0();
    if constexpr (CHAIN) {
        pair_barrier();                                  // every wave's stores of the last tile are acknowledged
        if (tid == 0) chain_signal(*cc, mb.flag_off + item);
    }
    range_flag(p, bad2.x + bad2.y);
    low_flag(p, low, bl + 4 * G::C, wave, lane, 8);
}

// one 8-wave block per CU (150 KB of LDS), 2 waves per SIMD
template <int DIL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convp_kernel(PairParams p) {
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
    // this block's items of each member: from the host's schedule (pair_schedule: few, unequal items per block), or its
    // contiguous share of the cost-weighted item sequence
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
        if (mb.k == 11) convp_run_member<ConvPGeom<11, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else if (mb.k == 7) convp_run_member<ConvPGeom<7, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else convp_run_member<ConvPGeom<3, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        first = false;
    }
}

}  // namespace fv
