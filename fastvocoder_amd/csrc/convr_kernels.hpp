// Two-source 1x1 conv with split-f16 operands, 128 output rows per tile:
//
//     y = post( W1 lrelu(x, slope) + W2 x2 + bias + res )              (ResidualStack's tail, reference modules.py:351-359:
//                                                                        conv1x1(lrelu(dilated conv)) + skip1x1(c))
//
// convg_kernel (convh_kernels.hpp, 64 rows x 128 columns per tile) one size up.  The GEMM has K = 2 C and no taps: every
// 128-channel chunk of the inputs is loaded, activated, split and written to the LDS image for FOUR K steps only, and
// that conversion is VALU work the matrix cores wait for (0.9 us per chunk against 0.64 us of MFMAs on a 64-row tile).
// Here a block owns 128 rows x 128 columns (8 waves = 4 row slabs of 32 x 2 column groups of 64; per wave and K step
// 4 A + 8 B ds_read_b128 for 24 MFMAs), so a converted chunk feeds twice the MFMAs.  A weight stage is ONE K step of all
// 128 rows (two 8 KB pieces, one per 64-row tile of the image fv_pack_conv1x1_2src_split_f16 lays out), ring of four:
// LDS = image 64 KB + ring 64 KB.  Same K order per output as convg_kernel: identical bits.
#pragma once
#include "convh_kernels.hpp"

namespace fv {

// TWO: the two-source 1x1 conv (KT = 1; K range [lrelu(x); x2]); else a 'same' conv with KT taps of dilation DIL --
// convh_kernel's convs on 128-row tiles: the (halo'd) window of a 128-channel chunk is converted once for KT * 4 K steps
// of BOTH 64-row tiles (ResidualStack's dilated conv, reflection-padded; HiFi-GAN large's 256 / 512-channel ResBlocks)
// TR: the transposed conv of convt_kernel (KT = 2 taps x[u-1], x[u]; rows are (output channel, phase): convh_kernels.hpp)
template <int KT_, int DIL_, bool TWO_ = false, bool TR_ = false>
struct ConvRGeom {
    static constexpr int KT = KT_, DIL = DIL_;
    static constexpr bool TWO = TWO_, TR = TR_;
    static constexpr int C = 128, CG = 4, CB = 16, NFW = 4, NT = 512;
    static constexpr int NTC = 128;                      // output columns per tile
    static constexpr int NSTEP = KT * CG;                // K steps of 32 per chunk = weight stages per chunk (tap-major)
    static constexpr int P = TR ? 1 : (KT - 1) * DIL / 2;
    static constexpr int XROWS = (NTC + (KT - 1) * DIL + 3) / 4 * 4;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;   // image: [split half][8-channel block][XRP rows][8 halves]
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;   // (row, 8-channel block) conversion tasks per thread
    static constexpr int NRAW = XR * 8;
    static constexpr int STAGE_BYTES = 16384, RING = 4, AHEAD = 3;
    static constexpr int WTILE = NSTEP * 8192;           // packed bytes of one (64-row tile, chunk): [step][8 KB]
    static constexpr int RAWST = NSTEP >= 8 ? NSTEP - 6 : 0;   // stage entry at which the next window is requested
    static_assert(NSTEP % RING == 0, "ring slot = step & 3");
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * (NFW - 1)) * 16 + 16 < 65536, "ds_read immediate range");
};

// items [item0, hi_item) of the launch's one member: item = ((utterance * n_tiles) + column tile) * nrt + row pair
template <class G>
__device__ __forceinline__ void convr_run(const PairParams& p, const PairMember& mb, int item0, int hi_item, float* smem,
                                          int wave, int lane_in, bool first) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    float* const ring = smem + p.x_off;
    char* const ximg = reinterpret_cast<char*>(smem + p.img_off);
    const int n = lane & 15, kb = lane >> 4;
    const int ws = wave >> 1, wn = wave & 1;             // row slab of 32, column group of 64
    const int col0 = wn * (16 * G::NFW) + n;
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;
    // A: slot * 4096 + ((row tile * 4 + row sixteenth) * 2 + split half) * 256 floats; this wave: sixteenths 2 ws, 2 ws + 1
    const float* const aptr = ring + (2 * ws) * 512 + lane * 4;
    const int row0 = 32 * ws + 4 * kb;                   // + 16 h + i: row inside the 128-row tile

    const int nch = p.nch, nrt = G::TR ? (p.nmt + 1) / 2 : p.nmt / 2;   // chunks of 128 input channels (TWO: x's, then x2's); row pairs
                                                         // (TR: an odd tile count -- the last pair's second tile lies beyond
                                                         // the packed image: its DMAs read zeros, its stores are dropped)

    const int spi = nch * G::NSTEP;                      // stages per item
    const size_t ustride = (size_t)p.ctot * (size_t)p.T;
    const size_t cstride = (size_t)G::C * (size_t)p.T;
    const unsigned ubytes = (unsigned)p.ctot * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const unsigned pair_stride = (unsigned)(nch * G::WTILE);        // from a 64-row tile's image to the next one's
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(mb.w1, (unsigned)(p.nmt * nch * G::WTILE));
    auto decode = [&](int it, int& b, int& nt, int& rt) {
        rt = it % nrt;
        const int q = it / nrt;
        b = q / mb.n_tiles;
        nt = q - b * mb.n_tiles;
    };
    // stage `lin` counted from the first stage of item `it` (it may lie in a later item): byte offset of its piece of the
    // first 64-row tile of the pair, or kOutOfRange past the block's last item
    auto stage_off = [&](int it, int lin) -> unsigned {
        const int it2 = it + lin / spi, st = lin % spi;
        return it2 < hi_item ? (unsigned)((it2 % nrt) * 2) * pair_stride + (unsigned)((st / G::NSTEP) * G::WTILE + (st % G::NSTEP) * 8192)
                             : kOutOfRange;
    };
    auto dma_stage = [&](int slot, unsigned off) {
        float* dst = ring + slot * (G::STAGE_BYTES / 4) + wave * 512;
        const unsigned o = off == kOutOfRange ? kOutOfRange : off + (unsigned)(wave >> 2) * pair_stride + (unsigned)((wave & 3) * 2048 + lane * 16);
        dma16(rw, dst, o);
        dma16(rw, dst + 256, o == kOutOfRange ? kOutOfRange : o + 1024u);
    };
    // chunk c of the K range: the tensor it comes from and the slope of its on-chip activation
    auto chunk_src = [&](int c, int bb) -> const float* {
        if constexpr (G::TWO) return (c < nch / 2 ? mb.x : mb.x2) + bb * ustride + (c < nch / 2 ? c : c - nch / 2) * cstride;
        else return mb.x + bb * ustride + c * cstride;
    };
    auto chunk_slope = [&](int c) { return G::TWO && c >= nch / 2 ? 1.f : p.slope; };

    int item = item0, chunk = 0;
    int b, ntile, rt;
    decode(item, b, ntile, rt);
    if (!first) pair_barrier();                          // everybody is done with the previous member's LDS
    LowGuard low;                                        // low side of the range guard (pairh_kernels.hpp)
    float bad = 0.f;                                     // range guard (pairh_kernels.hpp range_note4)
    ConvHRaw<G> raw;
    convh_load_raw<G>(raw, chunk_src(0, b), p.T, ntile * G::NTC - G::P, tid, true, p.reflect != 0);
#pragma unroll
    for (int st = 0; st < G::AHEAD; ++st) dma_stage(st, stage_off(item, st));
    pair_wait_vm0();
    if (!(p.dbg & 2)) convh_convert<G>(raw, ximg, chunk_slope(0), tid, low, 0);
    f32x4 hi[2][G::NFW], lo[2][G::NFW];                  // live across the chunks of an item
    for (int it = 0;; ++it) {
        pair_stamp(p, 8, wave, lane, it, 0);             // (tuning aid, -DFV_PAIR_TRACE: tools/convr_trace.py)
        const int t0 = ntile * G::NTC;
        int nchunk = chunk + 1, nitem = item;
        if (nchunk == nch) {
            nchunk = 0;
            nitem = item + 1;
        }
        const bool last = nchunk == 0;                   // the item's last chunk: the epilogue runs
        const bool more = nitem < hi_item;               // there is a next (item, chunk)
        int nb = b, nnt = ntile, nrt_ = rt;
        if (more && last) decode(nitem, nb, nnt, nrt_);
        if (chunk == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f16x8 abuf[2][2][2], bbuf[2][G::NFW][2];
        float bv[2][4], qv[2][4];                        // the item's bias and inverse row prescales (behind the packed image):
                                                         // requested in front of the last step's MFMAs (both operand queues'
                                                         // other halves are dead there: no register cost)

        // ---- stage entry (stage = K step GS of this chunk, ring slot GS): its weights are in the slot for every wave;
        // every wave holds the A operands of the stage before in registers, so that slot is free: request the stage three
        // further on into it
        auto entry = [&](auto GC) {
            constexpr int GS = decltype(GC)::value;
            {
                // this stage's DMA was issued three entries ago; loads return in order: it has landed once at most as
                // many loads are outstanding as were issued after it -- the DMAs of the two entries in between, plus
                // the raw window when it was requested at one of them (entry 0 of this chunk)
                // (an entry of the chunk before does not count: its window was converted before this chunk began)
                constexpr bool raw_between = G::RAWST <= GS - 1 && G::RAWST >= (GS >= 3 ? GS - 3 : 0);
                wait_vm<4 + (raw_between ? G::NRAW : 0)>();
            }
            pair_barrier();
            dma_stage((GS + 3) & 3, stage_off(item, chunk * G::NSTEP + GS + 3));
            if constexpr (GS == G::RAWST)
                convh_load_raw<G>(raw, chunk_src(nchunk, nb), p.T, nnt * G::NTC - G::P, tid, more && !(p.dbg & 1), p.reflect != 0);
        };
        auto fetch_a = [&](auto SC, f16x8 (&dst)[2][2]) {
            constexpr int S = decltype(SC)::value;
            LdsCF* a = lds_opaque(aptr + (S & 3) * (G::STAGE_BYTES / 4));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                dst[h][0] = *reinterpret_cast<LdsH8*>(a + h * 512);
                dst[h][1] = *reinterpret_cast<LdsH8*>(a + h * 512 + 256);
            }
        };
        LdsCF* const bb = lds_opaque(reinterpret_cast<const float*>(bptr));
        LdsCF* const bb2 = lds_opaque(reinterpret_cast<const float*>(bptr + G::XHALF));
        auto fetch_b = [&](auto SC, f16x8 (&dst)[G::NFW][2]) {
            constexpr int S = decltype(SC)::value;
            constexpr int tap = S / G::CG, cg = S % G::CG;
            constexpr int off = (cg * 4 * G::XRP + tap * G::DIL) * 4;
#pragma unroll
            for (int e = 0; e < G::NFW; ++e) {
                dst[e][0] = *reinterpret_cast<LdsH8*>(bb + off + e * 64);
                dst[e][1] = *reinterpret_cast<LdsH8*>(bb2 + off + e * 64);
            }
        };
        entry(IntC<0>{});
        pair_stamp(p, 8, wave, lane, it, 1);
        fetch_a(IntC<0>{}, abuf[0]);
        fetch_b(IntC<0>{}, bbuf[0]);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, G::NSTEP>([&](auto UC) {
            constexpr int U = decltype(UC)::value;
            if constexpr (U + 1 < G::NSTEP) {
                entry(IntC<U + 1>{});
                fetch_a(IntC<U + 1>{}, abuf[(U + 1) & 1]);
                fetch_b(IntC<U + 1>{}, bbuf[(U + 1) & 1]);
            } else {
                // (consumed in the epilogue below, before the next stage entry: the entries' wait counts do not see it)
                if constexpr (G::TR) {
                    const __amdgpu_buffer_rsrc_t rb = make_rsrc(mb.b1 ? mb.b1 : mb.w1, mb.b1 && last ? (unsigned)p.cout * 4u : 0u);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int m = 128 * rt + row0 + 16 * h;
                        int co = (int)((unsigned)m / (unsigned)p.ups), ph = m - co * p.ups;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            bv[h][i] = buffer_load1(rb, (unsigned)co * 4u);
                            if (++ph == p.ups) { ph = 0; ++co; }
                        }
                    }
                } else {
                    const __amdgpu_buffer_rsrc_t rb = make_rsrc(mb.b1 ? mb.b1 : mb.w1, mb.b1 && last ? (unsigned)p.ctot * 4u : 0u);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) bv[h][i] = buffer_load1(rb, (unsigned)(128 * rt + row0 + 16 * h + i) * 4u);
                }
                {
                    // (TR: an odd tile count -- the rows of the last pair's second tile lie beyond the tail: they read zeros)
                    const __amdgpu_buffer_rsrc_t rs = make_rsrc(mb.w1 + (size_t)p.nmt * nch * (G::WTILE / 4), last ? (unsigned)(p.nmt * 64) * 4u : 0u);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) qv[h][i] = buffer_load1(rs, (unsigned)(128 * rt + row0 + 16 * h + i) * 4u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < G::NFW; ++e)
                    hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[U & 1][h][0], bbuf[U & 1][e][0], hi[h][e], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < G::NFW; ++e)
                    lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[U & 1][h][0], bbuf[U & 1][e][1], lo[h][e], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < G::NFW; ++e)
                    lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[U & 1][h][1], bbuf[U & 1][e][0], lo[h][e], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- end of the chunk: (the item's outputs,) then the image of the next window --------------------------
        pair_stamp(p, 8, wave, lane, it, 2);
        pair_barrier();                                  // every wave is done with the image
        pair_stamp(p, 8, wave, lane, it, 3);
        if constexpr (G::TR) {
            if (last) {
                // y[co][ups u + phase - pad]: a lane's four rows are four consecutive phases -- inside one output channel
                // four consecutive samples, one 16-byte store (convh_run_member's epilogue, rows 128 rt + ...)
                const size_t yoff = (size_t)b * (size_t)p.cout * (size_t)p.Tout;
                const unsigned ybytes = (unsigned)p.cout * (unsigned)p.Tout * 4u;
                const __amdgpu_buffer_rsrc_t ry = make_rsrc(mb.y + yoff, ybytes);
                const __amdgpu_buffer_rsrc_t ra = make_rsrc(mb.y_act ? mb.y_act + yoff : mb.y, mb.y_act ? ybytes : 0u);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m0 = 128 * rt + row0 + 16 * h;
                    const int co0 = (int)((unsigned)m0 / (unsigned)p.ups), ph0 = m0 - co0 * p.ups;
                    const bool one_row = ph0 + 3 < p.ups;
#pragma unroll
                    for (int f = 0; f < G::NFW; ++f) {
                        const int n0 = (t0 + col0 + f * 16) * p.ups - p.pad_t;
                        float v[4], a[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] = fmaf(fmaf(lo[h][f][i], kSplitInv, hi[h][f][i]), qv[h][i], bv[h][i]);
                            a[i] = act(v[i], p.act_slope);
                            if (!mb.y_act) v[i] = a[i];              // no twin: y itself is stored activated
                        }
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
            const int rowt = 128 * rt + row0;
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(mb.res ? mb.res + b * ustride : mb.w1, mb.res ? ubytes : 0u);
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.sub ? p.sub + (p.sub_batched ? b * ustride : 0) : mb.w1, p.sub ? ubytes : 0u);
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(mb.y + b * ustride, ubytes);
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(mb.y_act ? mb.y_act + b * ustride : mb.y, mb.y_act ? ubytes : 0u);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    const int t = t0 + col0 + f * 16;
                    const unsigned voff = t < p.T ? (unsigned)((rowt + 16 * h) * p.T + t) * 4u : kOutOfRange;
                    float rv[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {0.f, 0.f, 0.f, 0.f}, v[4];
                    // (uniform branches: a launch without residual / offset does not wait for loads that return zeros)
                    if (mb.res != nullptr) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) rv[i] = buffer_load1s(rr, voff, (unsigned)i * t4);
                    }
                    if (G::TWO && p.sub != nullptr) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) sv[i] = buffer_load1s(rs, voff, (unsigned)i * t4);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaf(fmaf(lo[h][f][i], kSplitInv, hi[h][f][i]), qv[h][i], bv[h][i]) + rv[i];
                    if (!G::TWO && mb.add1 != nullptr) {
                        // the last launch of an MRF stage: ((own + add1) + add2), the reference's order (convh_run_member)
                        const __amdgpu_buffer_rsrc_t r1 = make_rsrc(mb.add1 + b * ustride, ubytes);
                        const __amdgpu_buffer_rsrc_t r2 = make_rsrc(mb.add2 ? mb.add2 + b * ustride : mb.add1, mb.add2 ? ubytes : 0u);
                        float a1[4], a2[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            a1[i] = buffer_load1s(r1, voff, (unsigned)i * t4);
                            a2[i] = buffer_load1s(r2, voff, (unsigned)i * t4);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = (v[i] + a1[i]) + a2[i];
                    }
                    range_note4(bad, v[0], v[1], v[2], v[3], t < p.T);
                    const unsigned vo = (p.dbg & 8) ? kOutOfRange : voff;
                    if (G::TWO && p.sub != nullptr) {
                        // the op carries an output offset (bias removal, basis_melgan.py:147-159): y2 = act(post(y)) - sub,
                        // or y itself when there is no second output -- conv_kernels.hpp's epilogue rule
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float w = v[i];
                            if (p.post == FV_POST_TANH) w = tanhf(w);
                            else if (p.post == FV_POST_RELU) w = fmaxf(w, 0.f);
                            const float a = (p.act_slope != 1.f ? act(w, p.act_slope) : w) - sv[i];
                            if (mb.y_act) {
                                buffer_store1s(ry, vo, (unsigned)i * t4, w);
                                buffer_store1s(ra, vo, (unsigned)i * t4, a);
                            } else {
                                buffer_store1s(ry, vo, (unsigned)i * t4, a);
                            }
                        }
                    } else {
                        pair_store(p, mb.y, mb.y_act, p.ctot, b, rowt + 16 * h, t, t < p.T && !(p.dbg & 8), v,
                                   G::TWO || mb.add1 != nullptr);
                    }
                }
            }
        }
        pair_stamp(p, 8, wave, lane, it, 4);
        // the stores first, the conversion of the next window after them (convh_run_member)
        if (more && !(p.dbg & 2)) {
#ifdef FV_PAIR_TRACE
            pair_wait_vm0();                             // (the raw window: its own stamp)
            pair_stamp(p, 8, wave, lane, it, 5);
#endif
            convh_convert<G>(raw, ximg, chunk_slope(nchunk), tid, low, G::TWO && nchunk >= nch / 2 ? 1 : 0);
        }
        pair_stamp(p, 8, wave, lane, it, 6);
        if (!more) break;
        item = nitem;
        chunk = nchunk;
        b = nb;
        ntile = nnt;
        rt = nrt_;
    }
    // the DMAs requested for a next item that does not exist wrote zeros; nothing is in flight past this point
    pair_wait_vm0();
    range_flag(p, bad);
    pair_barrier();                                      // every wave's ring DMAs have landed: the ring is scratch now
    low_flag(p, low, ring, wave, lane, 8);
}

// one 8-wave block per CU (128 KB of LDS), 2 waves per SIMD
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convr_kernel(PairParams p) {
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
    // equal items: block b takes [b n / nblk, (b + 1) n / nblk) -- the row pairs of one column tile stay together
    const int lo = equal_share(xcd_remap((int)blockIdx.x, (int)gridDim.x), n_items, q.nblk), hi = equal_share(xcd_remap((int)blockIdx.x, (int)gridDim.x) + 1, n_items, q.nblk);
    if (lo < hi) convr_run<ConvRGeom<1, 1, true>>(q, mb, lo, hi, smem, wave, lane, true);
}

// 'same' convs with KT taps (3 / 7 / 11) of dilation DIL, C = 128, 256 or 512 channels in and out, on 128-row tiles: the
// members of a launch as in convh_kernel (contiguous cost-weighted shares, or the host's block schedule)
template <int DIL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convs_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PairParams q;
    q.n_members = p.n_members; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = p.out_div;
    q.act_slope = p.act_slope; q.post = p.post; q.x_off = p.x_off; q.img_off = p.img_off; q.dbg = p.dbg; q.trace = p.trace;
    q.ctot = p.ctot; q.nch = p.nch; q.nmt = p.nmt; q.reflect = p.reflect; q.guard = p.guard; q.sub = nullptr; q.sub_batched = 0;
    int n_items[3], cost[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) { n_items[m] = p.m[m].n_items; cost[m] = p.m[m].cost; }
    asm volatile("" ::"s"(q.n_members), "s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.out_div), "s"(q.act_slope),
                 "s"(q.post), "s"(q.x_off), "s"(q.img_off), "s"(q.dbg), "s"(q.trace), "s"(n_items[0]), "s"(n_items[1]),
                 "s"(n_items[2]), "s"(cost[0]), "s"(cost[1]), "s"(cost[2]), "s"(q.ctot), "s"(q.nch), "s"(q.nmt), "s"(q.reflect),
                 "s"(q.guard));
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
        mb.x = p.m[m].x; mb.x2 = nullptr; mb.w1 = p.m[m].w1; mb.b1 = p.m[m].b1; mb.res = p.m[m].res; mb.add1 = p.m[m].add1;
        mb.add2 = p.m[m].add2; mb.y = p.m[m].y; mb.y_act = p.m[m].y_act; mb.k = p.m[m].k; mb.n_tiles = p.m[m].n_tiles;
        asm volatile("" ::"s"(mb.x), "s"(mb.w1), "s"(mb.b1), "s"(mb.res), "s"(mb.add1), "s"(mb.add2), "s"(mb.y), "s"(mb.y_act),
                     "s"(mb.k), "s"(mb.n_tiles));
        if constexpr (DIL > 5)                            // (dilation 9 is MelGAN's third ResidualStack layer: 3 taps only)
            convr_run<ConvRGeom<3, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else if (mb.k == 11) convr_run<ConvRGeom<11, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else if (mb.k == 7) convr_run<ConvRGeom<7, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        else convr_run<ConvRGeom<3, DIL>>(q, mb, lo, hi, smem, wave, lane, first);
        first = false;
    }
}

// ConvTranspose1d, kernel = 2 x stride, 128 or more input channels, on 128-row tiles (rows = (output channel, phase)):
// convt_kernel's GEMM with a chunk's window converted once for two 64-row tiles
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convu_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PairParams q;
    q.n_members = 1; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = 1.f;
    q.act_slope = p.act_slope; q.post = 0; q.x_off = p.x_off; q.img_off = p.img_off; q.dbg = p.dbg; q.trace = p.trace;
    q.ctot = p.ctot; q.nch = p.nch; q.nmt = p.nmt; q.reflect = 0; q.ups = p.ups; q.pad_t = p.pad_t; q.Tout = p.Tout; q.cout = p.cout;
    q.guard = p.guard; q.sub = nullptr; q.sub_batched = 0;
    PairMember mb;
    mb.x = p.m[0].x; mb.x2 = nullptr; mb.w1 = p.m[0].w1; mb.b1 = p.m[0].b1; mb.res = nullptr; mb.add1 = nullptr; mb.add2 = nullptr;
    mb.y = p.m[0].y; mb.y_act = p.m[0].y_act; mb.k = 2; mb.n_tiles = p.m[0].n_tiles;
    const int n_items = p.m[0].n_items;
    asm volatile("" ::"s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.act_slope), "s"(q.x_off), "s"(q.img_off), "s"(q.dbg),
                 "s"(q.trace), "s"(q.ctot), "s"(q.nch), "s"(q.nmt), "s"(q.ups), "s"(q.pad_t), "s"(q.Tout), "s"(q.cout), "s"(mb.x), "s"(mb.w1),
                 "s"(mb.b1), "s"(mb.y), "s"(mb.y_act), "s"(mb.n_tiles), "s"(n_items), "s"(q.guard));
    const int blk = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int lo = equal_share(blk, n_items, q.nblk), hi = equal_share(blk + 1, n_items, q.nblk);
    if (lo < hi) convr_run<ConvRGeom<2, 1, false, true>>(q, mb, lo, hi, smem, wave, lane, true);
}

}  // namespace fv
