// Fused ResBlock1 pair at 128 / 64 channels, split-f16 operands, weights from L2 straight into registers:
//
//     x' = x + conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2          (reference model/generator/modules.py:223-230)
//
// (Rounds 3 / 4 streamed the weights through an LDS ring -- convq_kernel at 128 channels, convp_kernel at 64; this form
// measured faster or equal everywhere and bit-identical, and the ring forms were removed in round 5: docs/history/.)
//
// Same tile (all 128 rows x 64 intermediate columns), same images, same K order, same warm tiles -- a different wave
// layout: 8 waves = 8 row slabs of SIXTEEN rows x ONE column group of 64 columns (a 16 x 64 wave tile: per K step
// 1 row sixteenth x 4 fragments x 3 split terms = the same 12 MFMAs, 8 B-operand ds_read_b128 -- what the 32 x 32 tile
// read as 4 A + 4 B).  With one column group no two waves share a row of weights, so every wave loads the A operands of
// its own 16 rows straight from L2 into registers (2 x 16 bytes per lane and K step, three K steps ahead in a queue that
// runs on from conv1 to conv2 to the next tile) -- the L2 traffic of the ring form, no LDS-DMA, no ring slot to hand
// over: the K loops have NO barrier (the ring form: one per K step, 88 per 11-tap tile); three per tile remain (window
// image complete, intermediate complete, intermediate free).  Identical bits: the same MFMA order per output.
#pragma once
#include "convh_kernels.hpp"
#ifndef FV_WARM_TILES
#define FV_WARM_TILES 1             // 0: every tile cold (A/B builds, tools/build_variant.py)
#endif

namespace fv {

// ---- tile geometry at 128 channels: a block owns ALL 128 rows of a 64-column tile (conv2 needs every channel of the
// intermediate); x image 64 + (KT - 1) DIL columns, intermediate 64 + 16 columns
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
    // (the residual: ConvQ2Run::RES_EARLY -- requested behind conv1's epilogue on the 16 x 64 wave tiles, in the epilogue on the
    // 32 x 64 ones, which are at the 256-register limit)
#ifndef FV_CONVQ_BDEPTH
#define FV_CONVQ_BDEPTH 2
#endif
    static constexpr int BD = FV_CONVQ_BDEPTH;           // B operands BD - 1 steps ahead of their MFMAs (register budget)
    static constexpr int NRAW = XRM * 8;
    static_assert(KT - 1 <= 16 && NSTEP >= 8, "taps");
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * (NFW - 1)) * 16 + 16 < 65536, "ds_read immediate range");
};

// ---- tile geometry at 64 channels: 128 intermediate columns per tile, the image / wave layout of convh_kernel<2, 2>
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


// C = 128: the geometry of convq_kernel (64 intermediate columns); C = 64: that of convp_kernel (128 columns: two column
// groups of 64 -- two waves do share a row sixteenth there and load it twice, 5.6 KB of weights per column instead of
// 2.8: the 64-channel weights are a quarter of the 128-channel ones and the tile is twice as wide)
// The kernel's second template argument: a channel count on its ordinary tile (128 / 64), or one of the wide forms below
constexpr int kPair64Wide = 65;      // 64 channels on 256-column tiles
constexpr int kPair128Wide = 129;    // 128 channels on 128-column tiles (dilation 1 and 3)
template <int KT_, int DIL_, int C_>
struct ConvQ2Geom;
template <int KT_, int DIL_>
struct ConvQ2Geom<KT_, DIL_, 128> : ConvQGeom<KT_, DIL_> {
    typedef ConvQGeom<KT_, DIL_> IMG;                    // window loader / converter geometry
    static constexpr int WN = 1;                         // column groups of 64
    static constexpr int WBYTES = 2 * IMG::WTILE;        // packed bytes of one conv
};
template <int KT_, int DIL_>
struct ConvQ2Geom<KT_, DIL_, 64> : ConvPGeom<KT_, DIL_> {
    typedef typename ConvPGeom<KT_, DIL_>::H IMG;
    static constexpr int WN = 2;
    static constexpr int CG = 2, CB = 8, NT = 512;
    static constexpr int NSTEP = IMG::NSTEP, XRP = IMG::XRP, XHALF = IMG::XHALF, WTILE = IMG::WTILE, NRAW = IMG::NRAW;
    static constexpr int WBYTES = IMG::WTILE;
};
// kPair64Wide: 64 channels on 256-column tiles -- 8 waves = 2 row slabs of 32 x 4 column groups of
// 64 (a 32 x 64 wave tile: 24 MFMAs per K step for 8 B reads and 4 A loads; the LDS operand traffic per MFMA of the 32 x 32
// tile halved, k - 1 of 256 intermediate columns recomputed by a cold tile instead of k - 1 of 128).  Without a ring the
// two images fit: (256 + 50) x 64 channels = 80 KB + 70 KB.  For launches with tiles to spare (the launcher decides).
template <int KT_, int DIL_>
struct ConvWideImg {                                     // window loader / converter geometry of the 256-column tile
    static constexpr int C = 64, CB = 8, NT = 512;
    static constexpr int XROWS = (256 + (KT_ - 1) * DIL_ + 3) / 4 * 4;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;
    static constexpr int NRAW = XR * 8;
};
template <int KT_, int DIL_>
struct ConvQ2Geom<KT_, DIL_, kPair64Wide> {
    typedef ConvWideImg<KT_, DIL_> IMG;
    static constexpr int KT = KT_, DIL = DIL_, C = 64, CG = 2, CB = 8, NT = 512, WN = 4;
    static constexpr int NM = 256, NOUT = NM - (KT - 1);
    static constexpr int P1 = (KT - 1) * DIL / 2, P2 = (KT - 1) / 2;
    static constexpr int NSTEP = KT * CG;
    static constexpr int XRP = IMG::XRP, XHALF = IMG::XHALF, NRAW = IMG::NRAW;
    static constexpr int MRP = NM + 16, MHALF = CB * MRP * 16;
    static constexpr int WTILE = NSTEP * 8192, WBYTES = WTILE;
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * 3) * 16 + 16 < 65536, "ds_read immediate range");
};
// kPair128Wide: 128 channels on 128-column tiles, 8 waves = 4 row slabs of 32 x 2 column groups of 64 -- the two images fit
// without the ring at dilation 1 and 3 (80 + 72 KB at 11 taps x dilation 3), not at 5
template <int KT_, int DIL_>
struct ConvWideImg128 {
    static constexpr int C = 128, CB = 16, NT = 512;
    static constexpr int XROWS = (128 + (KT_ - 1) * DIL_ + 3) / 4 * 4;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;
    static constexpr int NRAW = XR * 8;
};
template <int KT_, int DIL_>
struct ConvQ2Geom<KT_, DIL_, kPair128Wide> {
    typedef ConvWideImg128<KT_, DIL_> IMG;
    static constexpr int KT = KT_, DIL = DIL_, C = 128, CG = 4, CB = 16, NT = 512, WN = 2;
    static constexpr int NM = 128, NOUT = NM - (KT - 1);
    static constexpr int P1 = (KT - 1) * DIL / 2, P2 = (KT - 1) / 2;
    static constexpr int NSTEP = KT * CG;
    static constexpr int XRP = IMG::XRP, XHALF = IMG::XHALF, NRAW = IMG::NRAW;
    static constexpr int MRP = NM + 16, MHALF = CB * MRP * 16;
    static constexpr int WTILE = NSTEP * 8192, WBYTES = 2 * WTILE;
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * 3) * 16 + 16 < 65536, "ds_read immediate range");
    static_assert(2 * XHALF + 2 * MHALF + (4 * C + 16) * 4 <= 160 * 1024, "LDS: dilation 1 and 3 only");
};
#ifndef FV_Q2_ILV
#define FV_Q2_ILV 1
#endif
template <int KT_, int DIL_, int C_>
struct ConvQ2Run : ConvQ2Geom<KT_, DIL_, C_> {
    typedef ConvQ2Geom<KT_, DIL_, C_> B;
    static constexpr int NFW = 4;                        // fragments per wave: 64 columns
    static constexpr int NH = C_ == kPair64Wide || C_ == kPair128Wide ? 2 : 1;     // row sixteenths per wave
    static constexpr int NSLAB = B::C / (16 * NH);       // row slabs = waves per column group
#ifndef FV_Q2_DEEP
#define FV_Q2_DEEP 1
#endif
    // A operands this many K steps ahead (queue of QD + 1 slots).  The 32 x 64 tiles: one (a K step is 768 matrix cycles per
    // SIMD).  The 16 x 64 tiles: three was enough for the operands themselves (L2 hits), but the next tile's WINDOW can only be
    // requested where no operand of this tile queues behind it -- vmcnt completes in order -- i.e. QD steps before the tile ends:
    // at three steps (0.6 us) plus the epilogue a window that comes from another XCD's L2 or from HBM (~2 us) was waited for at
    // every tile end [measured, tools/convq2_trace.py: ~1 us per tile].  There are 50 spare registers: five / six steps at 64
    // channels with 3 / 7 taps (QD + 1 has to divide the tile's step count: the queue runs on from tile to tile; 44 steps at 11
    // taps leave three) [measured, tools/forward_ab.py, batch 1: the 64-channel stage 157.0 -> 154.4 us].  At 128 channels a
    // block has one or two tiles per launch at batch 1 -- hardly a next window to wait for -- and seven steps cost more in the
    // run's prologue than they save (140.3 -> 144.5 us): three.
    static constexpr int QD = NH == 2 ? 1 : !FV_Q2_DEEP || B::C == 128 ? 3 : KT_ == 3 ? 5 : KT_ == 7 ? 6 : 3;
    static constexpr int NA = 2 * NH;                    // loads per wave and K step
    static constexpr bool ILV = FV_Q2_ILV && NH == 1;    // the step's loads one per MFMA gap (below)
    static constexpr int NSEQ = 2 * B::NSTEP;            // K steps per tile: conv1's, then conv2's
    // The residual (x itself: 4 NFW NH dwords per lane) is requested behind conv1's epilogue on the 16 x 64 wave tiles (180-190
    // VGPRs: room for it) and lands during conv2 -- requested where it is consumed it was a round trip per tile (~1 us of the
    // 8-25 us a tile takes at batch 1).  The 32 x 64 tiles (250 VGPRs) keep the late request.
    static constexpr bool RES_EARLY = NH == 1;
    static constexpr int NRES = 4 * NFW * NH;
    static constexpr int RAWK = NSEQ - QD;               // K step at which the next tile's window is requested: no A operand of
                                                         // THIS tile is issued after it, so nothing here waits for it
    static_assert(NSEQ % (QD + 1) == 0, "the A queue runs on from tile to tile: slot = K step % (QD + 1)");
    static_assert(B::NM == 64 * B::WN && NSLAB * B::WN == 8, "8 waves: row slabs x column groups of 64");
};

template <class G>
__device__ __forceinline__ void convq2_run_member(const PairParams& p, const PairMember& mb, int item0, int hi_item,
                                                  float* smem, int wave, int lane_in, bool first) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    char* const ximg = reinterpret_cast<char*>(smem + p.img_off);
    char* const mimg = reinterpret_cast<char*>(smem + p.mid_off);
    float* const bl = smem + p.bias_off;                 // [b1[128] | b2[128] | inverse row prescales 1 | 2 | guard scratch]
    const int n = lane & 15, kb = lane >> 4;
    const int ws = wave % G::NSLAB;                      // row slab of 16 NH rows: sixteenths NH ws ...
    const int col0 = (wave / G::NSLAB) * 64 + n;
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;
    const char* const mptr = mimg + (kb * G::MRP + col0) * 16;
    const int row0 = 16 * G::NH * ws + 4 * kb;           // + 16 h + i
    // D fragment -> intermediate image: channels row0 + 16 h + i = half of the 8-channel block 2 (NH ws + h) + (kb >> 1)
    char* const mw = mimg + ((2 * G::NH * ws + (kb >> 1)) * G::MRP + col0) * 16 + 8 * (kb & 1);

    const size_t ustride = (size_t)G::C * (size_t)p.T;
    const unsigned ubytes = (unsigned)G::C * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(mb.w1, (unsigned)G::WBYTES);
    const __amdgpu_buffer_rsrc_t rw2 = make_rsrc(mb.w2, (unsigned)G::WBYTES);
    // packed image (fv_pack_pair_weight_ex): [64-row tile][K step][row sixteenth 4][split half][lane][8 halves]
    const int s16 = G::NH * ws;                          // first row sixteenth of this wave (64 channels: one row tile)
    const unsigned aoff = (unsigned)((s16 >> 2) * G::WTILE + (s16 & 3) * 2048 + lane * 16);
    // A operands of K step S of the tile's sequence (compile time; beyond the tile: the next tile's): [split half]
    auto load_a = [&](auto SC, f16x8 (&dst)[G::NH][2]) {
        constexpr int S = decltype(SC)::value % G::NSEQ;
        constexpr int step = S % G::NSTEP;
#pragma unroll
        for (int h = 0; h < G::NH; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                dst[h][e] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    S < G::NSTEP ? rw1 : rw2, (int)aoff, step * 8192 + h * 2048 + e * 1024, 0));
    };
    const int b_last = (hi_item - 1) / mb.n_tiles;
    const int c_last = min(((hi_item - 1) - b_last * mb.n_tiles + 1) * G::NOUT, p.T);     // end of the run in the last utterance
    int b = item0 / mb.n_tiles;
    int tout = (item0 - b * mb.n_tiles) * G::NOUT;       // first output column of the tile
    int c_end = b == b_last ? c_last : p.T;
    bool warm = false;                                   // (cold / warm tiles of a run: convq_kernels.hpp)
    if (!first) pair_barrier();
    pair_stamp(p, 8, wave, lane, 7, 12);                 // (tuning aid, -DFV_PAIR_TRACE: tools/convq2_trace.py) run start
    LowGuard low;
    f32x2 bad2 = {0.f, 0.f};
    const float rcp = div_rcp(p.out_div);
    typedef typename G::IMG IMG;
    ConvHRaw<IMG> raw;
    convh_load_raw<IMG>(raw, mb.x + b * ustride, p.T, tout - G::P1 - G::P2, tid, true);
    f16x8 aq[G::QD + 1][G::NH][2];                       // K step S sits in aq[S % (QD + 1)]
    static_for<0, G::QD>([&](auto QC) { load_a(QC, aq[decltype(QC)::value]); });
    if (tid < G::C) {
        bl[tid] = mb.b1 ? mb.b1[tid] : 0.f;
        bl[G::C + tid] = mb.b2 ? mb.b2[tid] : 0.f;
        bl[2 * G::C + tid] = mb.w1[G::WBYTES / 4 + tid];      // the rows' inverse weight prescales: behind the packed images
        bl[3 * G::C + tid] = mb.w2[G::WBYTES / 4 + tid];
    }
    // rows [NM, MRP) of the intermediate feed only discarded columns: finite values once
    for (int idx = tid; idx < 2 * G::CB * 64; idx += G::NT)
        reinterpret_cast<float*>(mimg + ((idx >> 6) * G::MRP + G::NM) * 16)[idx & 63] = 0.f;
    pair_wait_vm0();
    pair_stamp(p, 8, wave, lane, 7, 10);
    if (!(p.dbg & 2)) convh_convert<IMG>(raw, ximg, p.slope, tid, low);
    pair_stamp(p, 8, wave, lane, 7, 13);
    for (int it = 0;; ++it) {
        pair_stamp(p, 8, wave, lane, it, 0);
        const int t0 = tout;
        const int r0 = warm ? G::KT - 1 : 0;             // image row of the first NEW intermediate column
        const int n_out = warm ? G::NM : G::NOUT;
        const bool cont = t0 + n_out < c_end;            // the run goes on
        const bool nwarm = FV_WARM_TILES && cont;
        const bool more = cont || b < b_last;
        const int nb = cont ? b : b + 1;
        const int ntout = cont ? t0 + n_out : 0;
        const int nwin = ntout - G::P2 - G::P1 + (nwarm ? G::KT - 1 : 0);
        f32x4 hi[G::NH][G::NFW], lo[G::NH][G::NFW];
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
        // K steps [S0, S1) of the tile's sequence.  Loads return in order: step S's operands have landed once at most the
        // loads issued after them are outstanding -- QD steps' worth, plus the next window where it was requested in between
        auto run = [&](auto S0C, auto S1C) {
            constexpr int S0 = decltype(S0C)::value, S1 = decltype(S1C)::value;
#pragma unroll
            for (int h = 0; h < G::NH; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            fetch_b(IntC<S0>{}, bbuf[S0 & 1]);
            static_for<S0, S1>([&](auto SC) {
                constexpr int S = decltype(SC)::value;
                if constexpr (S == G::RAWK)
                    convh_load_raw<IMG>(raw, mb.x + nb * ustride, p.T, nwin, tid, more && !(p.dbg & 1));
                constexpr bool raw_after = G::RAWK > S - G::QD && G::RAWK <= S;     // requested after step S's loads were
                // (the early residual sits between the A operands of conv2's first QD steps -- issued during conv1 -- and the rest)
                constexpr int res_after = G::RES_EARLY && S >= G::NSTEP && S < G::NSTEP + G::QD ? G::NRES : 0;
                // (a tile's first QD steps were waited for in the epilogue of the tile before)
                if constexpr (G::ILV) {
                    if constexpr (S >= G::QD) wait_vm<G::NA * (G::QD - 1) + (raw_after ? G::NRAW : 0) + res_after>();
                    __builtin_amdgcn_sched_barrier(0);
                    load_a(IntC<S + G::QD>{}, aq[(S + G::QD) % (G::QD + 1)]);
                    if constexpr (S + 1 < S1) fetch_b(IntC<S + 1>{}, bbuf[(S + 1) & 1]);
                } else {
                    load_a(IntC<S + G::QD>{}, aq[(S + G::QD) % (G::QD + 1)]);
                    if constexpr (S + 1 < S1) fetch_b(IntC<S + 1>{}, bbuf[(S + 1) & 1]);
                    if constexpr (S >= G::QD) wait_vm<G::NA * G::QD + (raw_after ? G::NRAW : 0) + res_after>();
                    __builtin_amdgcn_sched_barrier(0);
                }
                f16x8 (&a)[G::NH][2] = aq[S % (G::QD + 1)];
#pragma unroll
                for (int h = 0; h < G::NH; ++h)
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e)
                        hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bbuf[S & 1][e][0], hi[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < G::NH; ++h)
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bbuf[S & 1][e][1], lo[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < G::NH; ++h)
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][1], bbuf[S & 1][e][0], lo[h][e], 0, 0, 0);
                if constexpr (G::ILV) {
                    // the 16 x 64 wave tiles: one load per MFMA gap instead of a burst of NA + 2 NFW between two steps -- the
                    // wave's own stream keeps the matrix pipe fed while it issues them [measured, batch 1: 47.6 -> 46.0 us
                    // at 128 channels, 54.5 -> 52.8 at 64; nothing on the 32 x 64 tiles, whose steps are twice as long]
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
                        __builtin_amdgcn_sched_group_barrier(0x008, 3 * G::NH * G::NFW - G::NA - 2 * G::NFW, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 3 * G::NH * G::NFW - G::NA, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        pair_barrier();                                  // the window image is complete
        run(IntC<0>{}, IntC<G::NSTEP>{});
        pair_stamp(p, 8, wave, lane, it, 1);
        {
            // conv1 -> intermediate image: row r is time t0 - P2 + r; conv2's zero padding applies to the intermediate
            const int tm = t0 - G::P2 + r0;              // time of the first new column
            const bool inside = tm >= 0 && tm + G::NM <= p.T;
            char* const mwr = mw + r0 * 16;
            float lowm = 0.f;
#pragma unroll
            for (int h = 0; h < G::NH; ++h) {
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
        float res[G::NH][G::NFW][4];
        unsigned voff[G::NFW];
        auto load_res = [&]() {
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(mb.x + b * ustride, ubytes);     // the residual is x itself
#pragma unroll
            for (int f = 0; f < G::NFW; ++f) {
                const int col = col0 + f * 16, t = t0 + col;
                voff[f] = col < n_out && t < c_end ? (unsigned)(row0 * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                for (int h = 0; h < G::NH; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) res[h][f][i] = buffer_load1s(rr, voff[f], (unsigned)(16 * h + i) * t4);
            }
        };
        if constexpr (G::RES_EARLY) load_res();
        pair_barrier();                                  // the intermediate is complete (and nobody reads the x image any more)
        pair_stamp(p, 8, wave, lane, it, 2);
        run(IntC<G::NSTEP>{}, IntC<G::NSEQ>{});
        pair_stamp(p, 8, wave, lane, it, 3);
        pair_barrier();                                  // every wave is done with the intermediate
        if (nwarm) {
            // the last KT - 1 valid columns -> the front of the image (convq_kernels.hpp)
            constexpr int NC = 2 * G::CB * (G::KT - 1);
            if (tid < NC) {
                const int row = tid % (G::KT - 1), hb = tid / (G::KT - 1);
                char* const base = mimg + (hb / G::CB) * G::MHALF + ((hb % G::CB) * G::MRP) * 16;
                *reinterpret_cast<f16x8*>(base + row * 16) =
                    *reinterpret_cast<const f16x8*>(base + (r0 + G::NM - (G::KT - 1) + row) * 16);
            }
        }
        if constexpr (!G::RES_EARLY) load_res();
        pair_wait_vm0();                                 // the next window, the residual, the next tile's first A operands
        pair_stamp(p, 8, wave, lane, it, 4);
        const bool fin = mb.add1 != nullptr;
#pragma unroll
        for (int h = 0; h < G::NH; ++h) {
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
            for (int h = 0; h < G::NH; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        lo[h][f][i] = buffer_load1s(r1, voff[f], (unsigned)(16 * h + i) * t4);
                        res[h][f][i] = buffer_load1s(r2, voff[f], (unsigned)(16 * h + i) * t4);
                    }
            pair_wait_vm0();
#pragma unroll
            for (int h = 0; h < G::NH; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hi[h][f][i] = (hi[h][f][i] + lo[h][f][i]) + res[h][f][i];
        }
#pragma unroll
        for (int h = 0; h < G::NH; ++h)
#pragma unroll
            for (int f = 0; f < G::NFW; ++f) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = hi[h][f][i];
                const int col = col0 + f * 16;
                range_note4p(bad2, hi[h][f]);
                pair_store(p, mb.y, mb.y_act, G::C, b, row0 + 16 * h, t0 + col, col < n_out && t0 + col < c_end && !(p.dbg & 8), v, fin,
                           rcp);
            }
        pair_stamp(p, 8, wave, lane, it, 5);
        if (more && !(p.dbg & 2)) convh_convert<IMG>(raw, ximg, p.slope, tid, low);
        pair_stamp(p, 8, wave, lane, it, 6);
        if (!more) break;
        if (!cont) c_end = nb == b_last ? c_last : p.T;
        b = nb;
        tout = ntout;
        warm = nwarm;
    }
    pair_wait_vm0();
    range_flag(p, bad2.x + bad2.y);
    low_flag(p, low, bl + 4 * G::C, wave, lane, 8);
}

// one 8-wave block per CU, 2 waves per SIMD
// C: 128, 64, kPair64Wide or kPair128Wide
template <int DIL, int C>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convq2_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const bool sched = p.sched_on == 1, cut = p.sched_on == 2;
    int slo[3] = {0, 0, 0}, shi[3] = {0, 0, 0};
    int g_lo = 0, g_hi = 0;
    if (cut) {
        const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
        g_lo = (int)p.sched[share];
        g_hi = share + 1 < q.nblk ? (int)p.sched[share + 1] : n_items[0] + (q.n_members > 1 ? n_items[1] : 0) + (q.n_members > 2 ? n_items[2] : 0);
        asm volatile("" ::"s"(g_lo), "s"(g_hi));
    }
    if (sched) {
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
        if (mb.k == 11) convq2_run_member<ConvQ2Run<11, DIL, C>>(q, mb, lo, hi, smem, wave, lane, first);
        else if (mb.k == 7) convq2_run_member<ConvQ2Run<7, DIL, C>>(q, mb, lo, hi, smem, wave, lane, first);
        else convq2_run_member<ConvQ2Run<3, DIL, C>>(q, mb, lo, hi, smem, wave, lane, first);
        first = false;
    }
}

}  // namespace fv
