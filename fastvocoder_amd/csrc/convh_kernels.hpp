// Conv1d with SPLIT-F16 operands for the wide ResBlock convs (C = 64, 128; 'same' zero padding):
//
//     y = post( ( conv1d( lrelu(x, slope); w, KT taps, dilation DIL ) + bias + res + add1 + add2 ) / out_div )
//
// Same arithmetic as pairh_kernels.hpp (every fp32 operand v = h1 + h2/2048, h1 / h2 f16; a product is three
// v_mfma_f32_16x16x32_f16 terms a1 b1 + (a1 b2 + a2 b1)/2048 accumulated in fp32: fp32-class accuracy at 3/16 of
// the fp32-MFMA cycles), same channels-last split image of the activated input in LDS (built in the kernel
// from the raw fp32 tensor, which is prefetched global -> registers half a tile ahead).  What is new is that
// the weights no longer fit on chip: C x C x KT x 4 bytes is 180 KB at C = 64 and 720 KB at C = 128.
//
//   * a block (8 waves = 2 row groups x 4 column groups) owns a 64-row x NTC-column output tile and walks the
//     K range (tap-major: step = 32 input channels of one tap) in STAGES of two steps;
//   * the packed weights of a stage (16 KB: [step][row sixteenth][split half][lane][8 f16], fv_pack_pair_weight_ex)
//     stream global(L2) -> LDS by LDS-DMA through a ring of 4 stage slots, three stages ahead of their use,
//     across tile boundaries (a block's tiles of one member form ONE linear stage sequence);
//   * one barrier per stage (it proves the stage's DMA parts of all waves have landed and frees the slot of
//     the stage before); the barrier of stage g+1 is taken one MFMA group early so that the operand prefetch
//     never drains at a stage boundary;
//   * a wave owns 2 row sixteenths x NFW column fragments: per step 4 A reads + 2 NFW B reads (ds_read_b128)
//     feed 6 NFW MFMAs; A operands are fetched one group (two fragments) ahead, B operands two, order pinned by
//     sched_barrier;
//   * vmcnt bookkeeping is static: every tile issues the same loads in the same order (out-of-range offsets
//     where there is nothing to load), so each stage waits with the exact count of younger loads.
//
// Tiles of one member are ordered (utterance, column tile, row tile): the two row tiles of a C = 128 layer share
// the column tile's image, which is converted once.  256 / 512 channels (HiFi-GAN large): the image holds 128 input
// channels at a time; an output tile walks its 2 / 4 channel chunks one after the other (convert chunk, K loop over
// the chunk's stages) and keeps its accumulators in registers across them.
#pragma once
#include "pairh_kernels.hpp"

namespace fv {

template <int I>
struct IntC {
    static constexpr int value = I;
};
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<I + 1, N>(f);
    }
}

// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
template <int N>
__device__ __forceinline__ void wait_vm() {
    constexpr int n = N > 63 ? 63 : N;
    __builtin_amdgcn_s_waitcnt((n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14));
}

// TR: the transposed conv of convt_kernel below (KT = 2 taps x[u-1], x[u]; rows are (output channel, phase))
// TWO: the two-source 1x1 conv of convg_kernel below (KT = 1; the K range is [lrelu(x); x2], chunks of 128 channels)
template <int CG_, int NFW_, int KT_, int DIL_, bool TR_ = false, bool TWO_ = false>
struct ConvHGeom {
    static constexpr int CG = CG_, NFW = NFW_, KT = KT_, DIL = DIL_;
    static constexpr bool TR = TR_, TWO = TWO_;
    static constexpr int C = 32 * CG;
    static constexpr int WM = 2, WN = 4, NW = 8, NT = 512;
    static constexpr int NTC = 16 * NFW * WN;            // output columns per tile
    static constexpr int NSTEP = KT * CG;                // K steps of 32
    static constexpr int NST = NSTEP / 2;                // stages of two steps
    static constexpr int NP = NFW / 2;                   // MFMA groups (two fragments) per step
    static constexpr int NUNIT = NSTEP * NP;
    static constexpr int P = TR ? 1 : (KT - 1) * DIL / 2;
    static constexpr int XROWS = (NTC + (KT - 1) * DIL + 3) / 4 * 4;
    static constexpr int CB = C / 8;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;   // image: [split half][8-channel block][XRP rows][8 halves]
    static constexpr int XHALF = CB * XRP * 16;          // (layout and why: PairHGeom)
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;   // (row, 8-channel block) conversion tasks per thread
    static constexpr int STAGE_BYTES = 16384, RING = 4;
    static constexpr int WTILE = NSTEP * 8192;           // packed bytes of one 64-row tile
    static constexpr int NMT = C / 64;
    static constexpr int RAWST = NST >= 4 ? NST - 4 : 0; // stage at which the next tile's raw window is requested
    static constexpr int NRAW = XR * 8;
    static constexpr int RESST = NST - 2;                // ... this tile's bias and residual
    static constexpr int NRES = TR ? 16 : 16 + 8 * NFW;  // loads of a tile's bias, inverse row prescales (+ residual)
    static_assert(NSTEP % 2 == 0 && NST >= 2 && NFW % 2 == 0, "stages of two steps, at least two per chunk");
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * (NFW - 1)) * 16 + 16 < 65536, "ds_read immediate range");
};

template <class G>
struct ConvHRaw {
    float v[G::XR][8];
};

// raw window rows [tA, tA + XROWS) of all C channels -> registers; rows outside [0, T) (or `live` false) read as zero,
// or (reflect: MelGAN's ReflectionPad1d, pad < T) as the row mirrored at the first / last sample
template <class G, int AUX = 0>
__device__ __forceinline__ void convh_load_raw(ConvHRaw<G>& r, const float* xb, int T, int tA, int tid, bool live,
                                               bool reflect = false, int cvalid = G::C) {
    // (cvalid < G::C: the transposed conv of a 64-channel input -- the missing channels read as zero)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)cvalid * (unsigned)T * 4u);
    const unsigned t4 = (unsigned)T * 4u;
#pragma unroll
    for (int q = 0; q < G::XR; ++q) {
        const int idx = tid + q * G::NT;
        const int cb = idx / G::XROWS, row = idx - cb * G::XROWS;
        int t = tA + row;
        if (reflect) {
            t = t < 0 ? -t : t;
            t = t >= T ? 2 * (T - 1) - t : t;
        }
        const bool ok = live && idx < G::XROWS * G::CB && t >= 0 && t < T;
        const unsigned voff = ok ? (unsigned)(cb * 8 * T + t) * 4u : kOutOfRange;
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[q][j] = buffer_load1s_aux<AUX>(rx, voff, (unsigned)j * t4);
    }
}

// (low: the low side of the range guard, pairh_kernels.hpp LowGuard; which: the operand tensor the window belongs to)
template <class G>
__device__ __forceinline__ void convh_convert(const ConvHRaw<G>& r, char* ximg, float slope, int tid, LowGuard& low, int which = 0) {
    float lowm = 0.f;
#pragma unroll
    for (int q = 0; q < G::XR; ++q) {
        const int idx = tid + q * G::NT;
        const int cb = idx / G::XROWS, row = idx - cb * G::XROWS;
        if (idx < G::XROWS * G::CB) {
            F16x8Parts h1, h2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 a = split_act2(f32x2{r.v[q][2 * j], r.v[q][2 * j + 1]}, slope);
                lowm = low_max3(lowm, a.x, a.y);
                split2(a, h1.p[j], h2.p[j]);
            }
            *reinterpret_cast<f16x8*>(ximg + (cb * G::XRP + row) * 16) = __builtin_bit_cast(f16x8, h1);
            *reinterpret_cast<f16x8*>(ximg + (cb * G::XRP + row) * 16 + G::XHALF) = __builtin_bit_cast(f16x8, h2);
        }
    }
    low_note(low, which, lowm);
}

// one stage of packed weights -> ring slot (2 DMA instructions per wave); byte_off: offset of the stage inside the
// member's packed image, or kOutOfRange (nothing to load: the slot is written with zeros and never read)
template <class G>
__device__ __forceinline__ void convh_dma_stage(__amdgpu_buffer_rsrc_t rw, float* ring, int slot, unsigned byte_off,
                                                int wave, int lane) {
    float* dst = ring + slot * (G::STAGE_BYTES / 4) + wave * 512;
    const unsigned o = byte_off + (unsigned)(wave * 2048 + lane * 16);
    dma16(rw, dst, o);
    dma16(rw, dst + 256, o + 1024u);
}

// items [item0, hi_item) of ONE member; item = (utterance * n_tiles + column tile) * p.nmt + row tile
template <class G>
__device__ __forceinline__ void convh_run_member(const PairParams& p, const PairMember& mb, int item0, int hi_item,
                                                 float* smem, int wave, int lane_in, bool first) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    float* const ring = smem + p.x_off;
    char* const ximg = reinterpret_cast<char*>(smem + p.img_off);
    const int n = lane & 15, kb = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int col0 = wn * (16 * G::NFW) + n;
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;              // B: + (4 cg XRP + tap DIL + 16 f) 16 (+ XHALF)
    const float* const aptr = ring + (wm * 2) * 512 + lane * 4;             // A: + slot*4096 + (i*4 + h)*512 + half*256
    const int row0 = 16 * (2 * wm) + 4 * kb;                                // + 16 h + i: row inside the 64-row tile

    // p.ctot channels in and out = p.nch chunks of G::C input channels, p.nmt row tiles of 64 (one chunk up to 128 channels)
    const int nch = p.nch, nmt = p.nmt;
    const size_t ustride = (size_t)p.ctot * (size_t)p.T;
    const size_t cstride = (size_t)G::C * (size_t)p.T;                      // one chunk of input channels
    const unsigned ubytes = (unsigned)p.ctot * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(mb.w1, (unsigned)(nmt * nch * G::WTILE));
    // transposed conv: rows are (output channel, phase): m = co * ups + phase, p.cout channels (rows beyond them, in
    // the last row tile, fall outside the output's buffer descriptor: their stores are dropped)
    const int cout = G::TR ? p.cout : 0;
    auto row_phase = [&](int m, int& co, int& ph) {
        co = (int)((unsigned)m / (unsigned)p.ups);
        ph = m - co * p.ups;
    };
    int item = item0, chunk = 0;
    int g0 = 0;                                                             // stage counter of the run (ring slot = g & 3)
    auto decode = [&](int it, int& b, int& nt, int& mt) {
        mt = it % nmt;
        const int q = it / nmt;
        b = q / mb.n_tiles;
        nt = q - b * mb.n_tiles;
    };
    int b, ntile, mtile;
    decode(item, b, ntile, mtile);
    if (!first) pair_barrier();                         // everybody is done with the previous member's LDS
    pair_stamp(p, 8, wave, lane, 7, 12);
    float bad = 0.f;                                    // range guard (pairh_kernels.hpp range_note, LowGuard)
    LowGuard low;
    ConvHRaw<G> raw;
    auto chunk_channels = [&](int c) { return G::TR ? min(G::C, p.ctot - c * G::C) : G::C; };
    // the input channels of chunk c: the tensor they come from, and the slope of their on-chip activation
    // (TWO: the first half of the chunks is lrelu(x, slope), the second half x2 as it is -- ResidualStack's skip branch)
    auto chunk_src = [&](int c, int bb) -> const float* {
        if constexpr (G::TWO) return (c < nch / 2 ? mb.x : mb.x2) + bb * ustride + (c < nch / 2 ? c : c - nch / 2) * cstride;
        else return mb.x + bb * ustride + c * cstride;
    };
    auto chunk_slope = [&](int c) { return G::TWO && c >= nch / 2 ? 1.f : p.slope; };
    // Transposed conv with a MERGED input (round 4): the upsampler behind an MRF stage reads the three ResBlocks' results
    // r0 (member x), r1 (add1), r2 (add2) and forms x = ((r0 + r1) + r2) / out_div itself -- hifigan.py:99-103 in the
    // reference's order -- so that the stage ends in ONE three-member pair launch instead of two launches (the second a
    // single cheap member that waited for the other two: 18-24 us for ~9 us of work at batch 1).  The window of r0 is
    // prefetched as before; r1 and r2 are fetched where the window is consumed (one L2 round trip per window, both in
    // flight together: they were written by the launch before).  Every value of x is formed exactly as the stage's last
    // launch formed it: same bits.
    const bool merge = G::TR && mb.add1 != nullptr;
    const float mrcp = div_rcp(p.out_div);
    auto merge_window = [&](int c, int bb, int nt) {
        ConvHRaw<G> t, u;
        const size_t off = (size_t)bb * ustride + (size_t)c * cstride;
        convh_load_raw<G>(t, mb.add1 + off, p.T, nt * G::NTC - G::P, tid, true, false, chunk_channels(c));
        convh_load_raw<G>(u, mb.add2 ? mb.add2 + off : mb.add1 + off, p.T, nt * G::NTC - G::P, tid, mb.add2 != nullptr, false,
                          chunk_channels(c));
        pair_wait_vm0();
#pragma unroll
        for (int q = 0; q < G::XR; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = (raw.v[q][j] + t.v[q][j]) + u.v[q][j];       // (no add2: u is zeros, and v + 0 = v)
                if (p.out_div != 1.f) v = mrcp != 0.f ? div_exact(v, p.out_div, mrcp) : v / p.out_div;
                raw.v[q][j] = v;
            }
    };
    // Two stages per chunk (the two-source 1x1 conv; the transposed conv of 64 input channels): three stages ahead can be
    // two chunks -- or, with one chunk per item, two ITEMS -- on.  The stages of an item (all chunks of one row tile) are
    // contiguous in the packed image: stage number `lin` counted from the first stage of item `it`, wherever it falls
    auto stage_off = [&](int it, int lin) -> unsigned {
        const int spi = nch * G::NST;
        const int it2 = it + lin / spi, st = lin % spi;
        return it2 < hi_item ? (unsigned)((it2 % nmt) * nch * G::WTILE + st * G::STAGE_BYTES) : kOutOfRange;
    };
    convh_load_raw<G>(raw, chunk_src(0, b), p.T, ntile * G::NTC - G::P, tid, true, p.reflect != 0, chunk_channels(0));
#pragma unroll
    for (int st = 0; st < 3; ++st)
        convh_dma_stage<G>(rw, ring, st, G::NST < 3 ? stage_off(item, st) : (unsigned)(mtile * nch * G::WTILE + st * G::STAGE_BYTES),
                           wave, lane);
    pair_stamp(p, 8, wave, lane, 7, 11);
    pair_wait_vm0();
    pair_stamp(p, 8, wave, lane, 7, 10);
    if constexpr (G::TR) {
        if (merge) merge_window(0, b, ntile);
    }
    if (!(p.dbg & 2)) convh_convert<G>(raw, ximg, chunk_slope(0), tid, low, 0);
    pair_stamp(p, 8, wave, lane, 7, 13);                 // (tuning aid, -DFV_PAIR_TRACE) prologue done
    f32x4 hi[2][G::NFW], lo[2][G::NFW];                // live across the channel chunks of an item
    for (int it = 0;; ++it) {
        const int t0 = ntile * G::NTC;
        pair_stamp(p, 8, wave, lane, it, 0);
        // the next (item, chunk): the next chunk of this output tile, or chunk 0 of the next item
        int nchunk = chunk + 1, nitem = item;
        if (nchunk == nch) {
            nchunk = 0;
            nitem = item + 1;
        }
        const bool last = nchunk == 0;                   // this is the tile's last chunk: the epilogue runs
        const bool more = nitem < hi_item;
        int nb = b, nnt = ntile, nmt_ = mtile;
        if (more && last) decode(nitem, nb, nnt, nmt_);
        const bool new_win = more && (nb != b || nnt != ntile || nchunk != chunk);
        const unsigned wnext = (unsigned)((nmt_ * nch + nchunk) * G::WTILE), wcur = (unsigned)((mtile * nch + chunk) * G::WTILE);
        if (chunk == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float res[2][G::NFW][4], bv[2][4], sv[2][4];     // sv: the rows' inverse weight prescales (behind the packed image)
        unsigned voff[G::NFW];
        f16x8 abuf[2][2][2], bbuf[3][2][2];         // A one group ahead, B two (from the image: no ring slot involved)

        // ---- stage entry: the stage's weights are in its ring slot for every wave; the slot of the stage
        // before is free: request the stage three ahead into it (this tile's, or the next item's first stages)
        auto entry = [&](auto GC) {
            constexpr int GS = decltype(GC)::value;
            {
                // This stage's DMA was issued three entries ago (for the first three stages: in the previous tile, or
                // the run's prologue).  Loads return in order, so it has landed once at most as many loads are
                // outstanding as were issued AFTER it: the DMAs of the two entries in between (4), plus the raw
                // window / the residual if they were requested at one of the three entries in between.  (Stores in
                // flight only make the count larger: conservative.)
                // For GS < 3 the DMA was issued in the previous tile, whose epilogue waited for it BEFORE it issued its
                // stores: no wait here -- a count would wait for those stores, which are younger and retire late
                // [measured, 64 channels: ~0.8 us per tile].
                constexpr bool raw_between = G::RAWST >= GS - 3 && G::RAWST <= GS - 1;
                constexpr bool res_between = G::RESST >= GS - 3 && G::RESST <= GS - 1;
                if constexpr (GS >= 3) wait_vm<4 + (raw_between ? G::NRAW : 0) + (res_between ? G::NRES : 0)>();
            }
#ifdef FV_CONVH_EXP
            if (!(p.dbg & 32))
#endif
            pair_barrier();
            constexpr int NS = GS + 3;
            unsigned off;
            if constexpr (G::NST < 3) off = stage_off(item, chunk * G::NST + NS);
            else if constexpr (NS < G::NST) off = wcur + (unsigned)(NS * G::STAGE_BYTES);
            else off = more ? wnext + (unsigned)((NS - G::NST) * G::STAGE_BYTES) : kOutOfRange;
            convh_dma_stage<G>(rw, ring, (g0 + NS) & 3, off, wave, lane);
            if constexpr (GS == G::RAWST)
                convh_load_raw<G>(raw, chunk_src(nchunk, nb), p.T, nnt * G::NTC - G::P, tid,
                                  new_win && !(p.dbg & 1), p.reflect != 0, chunk_channels(nchunk));
            if constexpr (GS == G::RESST) {
                // bias, inverse row prescales and residual of THIS tile: in flight during the last two stages
                // (before the tile's last chunk the same loads are issued out of range: the wait counts stay static)
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(mb.w1 + (size_t)nmt * nch * (G::WTILE / 4), last ? (unsigned)(nmt * 64) * 4u : 0u);
                if constexpr (G::TR) {
                    const __amdgpu_buffer_rsrc_t rb = make_rsrc(mb.b1 ? mb.b1 : mb.w1, mb.b1 && last ? (unsigned)cout * 4u : 0u);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        int co, ph;
                        row_phase(64 * mtile + row0 + 16 * h, co, ph);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            bv[h][i] = buffer_load1(rb, (unsigned)co * 4u);
                            if (++ph == p.ups) { ph = 0; ++co; }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) sv[h][i] = buffer_load1(rs, (unsigned)(64 * mtile + row0 + 16 * h + i) * 4u);
                    }
                } else {
                const __amdgpu_buffer_rsrc_t rb = make_rsrc(mb.b1 ? mb.b1 : mb.w1, mb.b1 && last ? (unsigned)p.ctot * 4u : 0u);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        bv[h][i] = buffer_load1(rb, (unsigned)(64 * mtile + row0 + 16 * h + i) * 4u);
                        sv[h][i] = buffer_load1(rs, (unsigned)(64 * mtile + row0 + 16 * h + i) * 4u);
                    }
                const __amdgpu_buffer_rsrc_t rr = make_rsrc(mb.res ? mb.res + b * ustride : mb.w1, mb.res ? ubytes : 0u);
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    const int t = t0 + col0 + f * 16;
                    voff[f] = t < p.T && last ? (unsigned)((64 * mtile + row0) * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) res[h][f][i] = buffer_load1s(rr, voff[f], (unsigned)(16 * h + i) * t4);
                }
                }
            }
        };
        auto fetch_a = [&](auto SC, f16x8 (&dst)[2][2]) {
            constexpr int S = decltype(SC)::value;
            LdsCF* a = lds_opaque(aptr + ((g0 + S / 2) & 3) * (G::STAGE_BYTES / 4));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                dst[h][0] = *reinterpret_cast<LdsH8*>(a + ((S % 2) * 4 + h) * 512);
                dst[h][1] = *reinterpret_cast<LdsH8*>(a + ((S % 2) * 4 + h) * 512 + 256);
            }
        };
        LdsCF* const bb = lds_opaque(reinterpret_cast<const float*>(bptr));
        LdsCF* const bb2 = lds_opaque(reinterpret_cast<const float*>(bptr + G::XHALF));
        auto fetch_b = [&](auto SC, auto PC, f16x8 (&dst)[2][2]) {
            constexpr int S = decltype(SC)::value, PP = decltype(PC)::value;
            constexpr int tap = S / G::CG, cg = S % G::CG;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                constexpr int off = (cg * 4 * G::XRP + tap * G::DIL + (2 * PP) * 16) * 4;   // in floats
                dst[e][0] = *reinterpret_cast<LdsH8*>(bb + off + e * 64);
                dst[e][1] = *reinterpret_cast<LdsH8*>(bb2 + off + e * 64);
            }
        };

        entry(IntC<0>{});
        pair_stamp(p, 8, wave, lane, it, 1);
        fetch_a(IntC<0>{}, abuf[0]);
        fetch_b(IntC<0>{}, IntC<0>{}, bbuf[0]);
        fetch_b(IntC<1 / G::NP>{}, IntC<1 % G::NP>{}, bbuf[1]);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, G::NUNIT>([&](auto UC) {
            constexpr int U = decltype(UC)::value;
            constexpr int S = U / G::NP, PP = U % G::NP;
            constexpr int UN = U + 1, SN = UN / G::NP, PN = UN % G::NP;
            if constexpr (UN < G::NUNIT) {
                // the next group starts a new stage: take its barrier now, then prefetch from its slot
                if constexpr (PN == 0 && SN % 2 == 0) entry(IntC<SN / 2>{});
                if constexpr (PN == 0) fetch_a(IntC<SN>{}, abuf[SN & 1]);
            }
            if constexpr (U + 2 < G::NUNIT) fetch_b(IntC<(U + 2) / G::NP>{}, IntC<(U + 2) % G::NP>{}, bbuf[(U + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#ifdef FV_CONVH_EXP
            if (!(p.dbg & 4))
#endif
            {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        hi[h][2 * PP + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][0], bbuf[U % 3][e][0],
                                                                                   hi[h][2 * PP + e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        lo[h][2 * PP + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][0], bbuf[U % 3][e][1],
                                                                                   lo[h][2 * PP + e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        lo[h][2 * PP + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][1], bbuf[U % 3][e][0],
                                                                                   lo[h][2 * PP + e], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- epilogue: outputs, then the image of the next window ------------------------------------------------------
        pair_stamp(p, 8, wave, lane, it, 2);
        pair_barrier();                                  // every wave is done with the image (and with the last ring reads)
        pair_stamp(p, 8, wave, lane, it, 3);
        wait_vm<0>();                                    // raw window, residual, the weight stages requested so far
        pair_stamp(p, 8, wave, lane, it, 4);
        if constexpr (G::TR) {
            // (before this tile's stores are issued: a wait behind them would take their whole round trip)
            if (merge && new_win) merge_window(nchunk, nb, nnt);
        }
        if constexpr (G::TR) {
            if (last) {
                // y[co][ups u + phase - pad]: a lane's four rows are four consecutive phases -- inside one output channel
                // four consecutive samples, one 16-byte store (8-phase upsamplers: always)
                const size_t yoff = (size_t)b * (size_t)cout * (size_t)p.Tout;
                const unsigned ybytes = (unsigned)cout * (unsigned)p.Tout * 4u;
                const __amdgpu_buffer_rsrc_t ry = make_rsrc(mb.y + yoff, ybytes);
                const __amdgpu_buffer_rsrc_t ra = make_rsrc(mb.y_act ? mb.y_act + yoff : mb.y, mb.y_act ? ybytes : 0u);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int co0, ph0;
                    row_phase(64 * mtile + row0 + 16 * h, co0, ph0);
                    const bool one_row = ph0 + 3 < p.ups;
#pragma unroll
                    for (int f = 0; f < G::NFW; ++f) {
                        const int n0 = (t0 + col0 + f * 16) * p.ups - p.pad_t;
                        float v[4], a[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] = fmaf(fmaf(lo[h][f][i], kSplitInv, hi[h][f][i]), sv[h][i], bv[h][i]);
                            a[i] = act(v[i], p.act_slope);
                            if (!mb.y_act) v[i] = a[i];              // no twin: y itself is stored activated
                        }
                        // (columns past the input and rows past the output channels multiply zeros: finite)
                        range_note4(bad, v[0], v[1], v[2], v[3], true);
                        if (one_row && n0 + ph0 >= 0 && n0 + ph0 + 3 < p.Tout && !(p.dbg & 8)) {
                            const unsigned off = (unsigned)(co0 * p.Tout + n0 + ph0) * 4u;
                            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                                                         __float_as_uint(v[3])}, ry, (int)off, 0, 0);
                            if (mb.y_act)
                                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(a[0]), __float_as_uint(a[1]),
                                                                             __float_as_uint(a[2]), __float_as_uint(a[3])}, ra, (int)off, 0, 0);
                        } else {
                            int co = co0, ph = ph0;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int n = n0 + ph;
                                const unsigned off = n >= 0 && n < p.Tout && !(p.dbg & 8) ? (unsigned)(co * p.Tout + n) * 4u : kOutOfRange;
                                buffer_store1(ry, off, v[i]);
                                if (mb.y_act) buffer_store1(ra, off, a[i]);
                                if (++ph == p.ups) { ph = 0; ++co; }
                            }
                        }
                    }
                }
            }
        } else
        if (last) {
            const bool fin = G::TWO || mb.add1 != nullptr;      // (TWO: the op's post applies to its one member)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        hi[h][f][i] = fmaf(fmaf(lo[h][f][i], kSplitInv, hi[h][f][i]), sv[h][i], bv[h][i]) + res[h][f][i];
            if (mb.add1 != nullptr) {
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
            if (G::TWO && p.sub != nullptr) {
                // the op carries an output offset (bias removal, basis_melgan.py:147-159): y2 = act(post(y)) - sub, or y
                // itself when there is no second output -- conv_kernels.hpp's epilogue rule
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.sub + (p.sub_batched ? b * ustride : 0), ubytes);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int f = 0; f < G::NFW; ++f)
#pragma unroll
                        for (int i = 0; i < 4; ++i) res[h][f][i] = buffer_load1s(rs, voff[f], (unsigned)(16 * h + i) * t4);
                pair_wait_vm0();
                const __amdgpu_buffer_rsrc_t ry = make_rsrc(mb.y + b * ustride, ubytes);
                const __amdgpu_buffer_rsrc_t ra = make_rsrc(mb.y_act ? mb.y_act + b * ustride : mb.y, mb.y_act ? ubytes : 0u);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int f = 0; f < G::NFW; ++f) {
                        const int t = t0 + col0 + f * 16;
                        range_note4(bad, hi[h][f][0], hi[h][f][1], hi[h][f][2], hi[h][f][3], t < p.T);
                        const unsigned vo = (p.dbg & 8) ? kOutOfRange : voff[f];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v = hi[h][f][i];
                            if (p.post == FV_POST_TANH) v = tanhf(v);
                            else if (p.post == FV_POST_RELU) v = fmaxf(v, 0.f);
                            const float a = (p.act_slope != 1.f ? act(v, p.act_slope) : v) - res[h][f][i];
                            if (mb.y_act) {
                                buffer_store1s(ry, vo, (unsigned)(16 * h + i) * t4, v);
                                buffer_store1s(ra, vo, (unsigned)(16 * h + i) * t4, a);
                            } else {
                                buffer_store1s(ry, vo, (unsigned)(16 * h + i) * t4, a);
                            }
                        }
                    }
            } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = hi[h][f][i];
                    const int t = t0 + col0 + f * 16;
                    range_note4(bad, v[0], v[1], v[2], v[3], t < p.T);
                    pair_store(p, mb.y, mb.y_act, p.ctot, b, 64 * mtile + row0 + 16 * h, t, t < p.T && !(p.dbg & 8), v, fin);
                }
            }
        }
        pair_stamp(p, 8, wave, lane, it, 5);
        // the stores first, the conversion of the next window after them: a vmcnt wait cannot tell stores from loads,
        // the next tile's first stage waits would otherwise sit behind the stores' round trip
        if (new_win && !(p.dbg & 2)) convh_convert<G>(raw, ximg, chunk_slope(nchunk), tid, low, G::TWO && nchunk >= nch / 2 ? 1 : 0);
        pair_stamp(p, 8, wave, lane, it, 6);
        if (!more) break;
        g0 += G::NST;
        item = nitem;
        chunk = nchunk;
        b = nb;
        ntile = nnt;
        mtile = nmt_;
    }
    // the DMAs requested for a next item that does not exist wrote zeros; nothing is in flight past this point
    pair_wait_vm0();
    range_flag(p, bad);
    pair_barrier();                                      // every wave's ring DMAs have landed: the ring is scratch now
    low_flag(p, low, ring, wave, lane, G::NW);
}

// 8 waves per block, one block per CU (150-160 KB of LDS): 2 waves per SIMD, 256 VGPRs
template <int CG, int NFW, int DIL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convh_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    pair_stamp(p, 8, wave, lane, 7, 15);
    // Every scalar of the launch in ONE batch of kernarg loads.  Left to itself hipcc loads each field right before its
    // first use: nine dependent s_load round trips (~500 cycles each) stood between kernel entry and the first tile
    // (tools/convh_trace.py: 4.4-6.3k of a launch's ~60k cycles).
    PairParams q;
    q.n_members = p.n_members; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = p.out_div;
    q.act_slope = p.act_slope; q.post = p.post; q.x_off = p.x_off; q.img_off = p.img_off; q.dbg = p.dbg; q.trace = p.trace;
    q.ctot = p.ctot; q.nch = p.nch; q.nmt = p.nmt; q.reflect = p.reflect; q.guard = p.guard;
    int n_items[3], cost[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) { n_items[m] = p.m[m].n_items; cost[m] = p.m[m].cost; }
    asm volatile("" ::"s"(q.n_members), "s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.out_div), "s"(q.act_slope),
                 "s"(q.post), "s"(q.x_off), "s"(q.img_off), "s"(q.dbg), "s"(q.trace), "s"(n_items[0]), "s"(n_items[1]),
                 "s"(n_items[2]), "s"(cost[0]), "s"(cost[1]), "s"(cost[2]), "s"(q.ctot), "s"(q.nch), "s"(q.nmt), "s"(q.reflect),
                 "s"(q.guard));
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
        // ... and this member's pointers and sizes in one more
        PairMember mb;
        mb.x = p.m[m].x; mb.w1 = p.m[m].w1; mb.b1 = p.m[m].b1; mb.res = p.m[m].res; mb.add1 = p.m[m].add1;
        mb.add2 = p.m[m].add2; mb.y = p.m[m].y; mb.y_act = p.m[m].y_act; mb.k = p.m[m].k; mb.n_tiles = p.m[m].n_tiles;
        asm volatile("" ::"s"(mb.x), "s"(mb.w1), "s"(mb.b1), "s"(mb.res), "s"(mb.add1), "s"(mb.add2), "s"(mb.y), "s"(mb.y_act),
                     "s"(mb.k), "s"(mb.n_tiles));
        if constexpr (DIL > 5)                            // (dilation 9 is MelGAN's third ResidualStack layer: 3 taps only)
            convh_run_member<ConvHGeom<CG, NFW, 3, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else if (mb.k == 11) convh_run_member<ConvHGeom<CG, NFW, 11, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else if (mb.k == 7) convh_run_member<ConvHGeom<CG, NFW, 7, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else convh_run_member<ConvHGeom<CG, NFW, 3, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        first = false;
    }
    pair_stamp(q, 8, wave, lane, 7, 14);
}

// ConvTranspose1d, kernel = 2 x stride (every shipped upsampler: hifigan.py:45-46, melgan.py:37-39), as the same
// streamed-weight GEMM: with n' = n + pad = ups u + phase every output sample has exactly the two taps x[u] (kernel index
// phase) and x[u - 1] (kernel index ups + phase), so all phases share ONE window and the rows of the GEMM are
// (output channel, phase) -- M = Cout ups rows, K = 2 taps x Cin (chunks of 128 channels), N = Tin + 1 columns u.
// Rows are channel-major (m = co ups + phase): the four rows a lane holds are consecutive output samples.
template <int CG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convt_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PairParams q;
    // (out_div, add1, add2: the INPUT merge of an upsampler behind an MRF stage -- convh_run_member merge_window)
    q.n_members = 1; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = p.out_div;
    q.act_slope = p.act_slope; q.post = 0; q.x_off = p.x_off; q.img_off = p.img_off; q.dbg = p.dbg; q.trace = p.trace;
    q.ctot = p.ctot; q.nch = p.nch; q.nmt = p.nmt; q.reflect = 0; q.ups = p.ups; q.pad_t = p.pad_t; q.Tout = p.Tout; q.cout = p.cout;
    q.guard = p.guard;
    PairMember mb;
    mb.x = p.m[0].x; mb.w1 = p.m[0].w1; mb.b1 = p.m[0].b1; mb.res = nullptr; mb.add1 = p.m[0].add1; mb.add2 = p.m[0].add2;
    mb.y = p.m[0].y; mb.y_act = p.m[0].y_act; mb.k = 2; mb.n_tiles = p.m[0].n_tiles;
    const int n_items = p.m[0].n_items;
    asm volatile("" ::"s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.act_slope), "s"(q.x_off), "s"(q.img_off), "s"(q.dbg),
                 "s"(q.trace), "s"(q.ctot), "s"(q.nch), "s"(q.nmt), "s"(q.ups), "s"(q.pad_t), "s"(q.Tout), "s"(q.cout), "s"(mb.x), "s"(mb.w1),
                 "s"(mb.b1), "s"(mb.y), "s"(mb.y_act), "s"(mb.n_tiles), "s"(n_items), "s"(q.guard), "s"(q.out_div), "s"(mb.add1), "s"(mb.add2));
    // equal items: block b takes [b n / nblk, (b + 1) n / nblk)
    const int lo = equal_share(xcd_remap((int)blockIdx.x, (int)gridDim.x), n_items, q.nblk), hi = equal_share(xcd_remap((int)blockIdx.x, (int)gridDim.x) + 1, n_items, q.nblk);
    if (lo < hi) convh_run_member<ConvHGeom<CG, 2, 2, 1, true>>(q, mb, lo, hi, smem, wave, lane, true);
}

// Two-source 1x1 conv: y = post(W1 lrelu(x, slope) + W2 x2 + bias + res) -- the tail of MelGAN's ResidualStack
// (modules.py:362-366,382: stack[4](act(h)) + skip_layer(c)) as ONE GEMM over the concatenated K range [lrelu(x); x2], on
// the same streamed-weight pipeline: an item (utterance, column tile, row tile of 64) walks 2 C / 128 chunks of 128 input
// channels, the first half from x (activated while it is split), the second half from x2 as it is; a chunk is 4 K steps =
// two weight stages, the accumulators stay in registers across the chunks.  C = 128, 256 or 512 channels in and out.
template <int CG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convg_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PairParams q;
    q.n_members = 1; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = 1.f;
    q.act_slope = p.act_slope; q.post = p.post; q.x_off = p.x_off; q.img_off = p.img_off; q.dbg = p.dbg; q.trace = p.trace;
    q.ctot = p.ctot; q.nch = p.nch; q.nmt = p.nmt; q.reflect = 0; q.guard = p.guard; q.sub = p.sub; q.sub_batched = p.sub_batched;
    PairMember mb;
    mb.x = p.m[0].x; mb.x2 = p.m[0].x2; mb.w1 = p.m[0].w1; mb.b1 = p.m[0].b1; mb.res = p.m[0].res; mb.add1 = nullptr;
    mb.add2 = nullptr; mb.y = p.m[0].y; mb.y_act = p.m[0].y_act; mb.k = 1; mb.n_tiles = p.m[0].n_tiles;
    const int n_items = p.m[0].n_items;
    asm volatile("" ::"s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.act_slope), "s"(q.post), "s"(q.x_off), "s"(q.img_off),
                 "s"(q.dbg), "s"(q.trace), "s"(q.ctot), "s"(q.nch), "s"(q.nmt), "s"(q.guard), "s"(q.sub), "s"(q.sub_batched), "s"(mb.x), "s"(mb.x2), "s"(mb.w1),
                 "s"(mb.b1), "s"(mb.res), "s"(mb.y), "s"(mb.y_act), "s"(mb.n_tiles), "s"(n_items));
    // equal items: block b takes [b n / nblk, (b + 1) n / nblk) -- row tiles of one column tile stay together
    const int lo = equal_share(xcd_remap((int)blockIdx.x, (int)gridDim.x), n_items, q.nblk), hi = equal_share(xcd_remap((int)blockIdx.x, (int)gridDim.x) + 1, n_items, q.nblk);
    if (lo < hi) convh_run_member<ConvHGeom<CG, 2, 1, 1, false, true>>(q, mb, lo, hi, smem, wave, lane, true);
}

}  // namespace fv
