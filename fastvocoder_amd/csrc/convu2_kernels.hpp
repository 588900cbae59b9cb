// ConvTranspose1d (kernel = 2 x stride) with split-f16 operands, 128 or 256 input channels, for launches with MANY items --
// the upsamplers of Basis-MelGAN / MelGAN / HiFi-GAN large at batch (reference model/generator/basis_melgan.py:60-75,
// melgan.py:75-85, hifigan.py:93-96):
//
//     y = conv_transpose1d( lrelu(x, pre_slope); w [Cin, Cout, 2 s], stride s, pad, out_pad ) + bias
//
// Same GEMM, packed weights and sums per output as convt_kernel / convu_kernel (rows m = co s + phase, K = 2 taps x Cin in chunks
// of 128 channels, columns u = input sample): identical bits.  convu_kernel's item is (column tile, pair of 64-row tiles): with
// Cout s = 1024 rows every 128-column window is loaded, activated, split and written to LDS EIGHT times, each time for sixteen K
// steps behind a barrier each -- conversion and barriers are 40 % of an item, MFMA-busy 0.28 (config 4).  Here an item is a column
// tile with ALL its rows:
//   * the images of all chunks stay resident (two chunks of 128 channels x 132 rows = 147 KB: no weight ring to make room for),
//     converted once per item;
//   * the row tiles of 128 run one after the other on them; every wave loads the A operands of its own 32 rows straight from L2
//     into registers one K step ahead (4 x 16 bytes per lane and step), the queue running on from row tile to row tile and item
//     to item: no LDS-DMA, no barrier inside an item except the two that publish the images;
//   * the second chunk's window travels while the first row tile works on the first chunk's image, the next item's first window
//     during the last row tile.
// 8 waves = 4 row slabs of 32 x 2 column groups of 64 (a 32 x 64 wave tile: 24 MFMAs per K step for 4 A loads + 8 B reads).
#pragma once
#include "convh_kernels.hpp"

namespace fv {

template <int NCH_>
struct ConvU2Geom {
    static constexpr int NCH = NCH_;                     // chunks of 128 input channels, all resident
    static constexpr int C = 128, CG = 4, CB = 16, NFW = 4, NH = 2, NT = 512;
    static constexpr int NTC = 128;                      // columns (input samples u) per tile
    static constexpr int P = 1;                          // the window starts one sample early: tap 0 multiplies x[u - 1]
    static constexpr int NSTEP = 2 * CG;                 // K steps of 32 per chunk: tap-major, then 32-channel group
    static constexpr int XROWS = (NTC + 1 + 3) / 4 * 4;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;   // image: [split half][8-channel block][XRP rows][8 halves]
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;
    static constexpr int NRAW = XR * 8;
    static constexpr int WTILE = NSTEP * 8192;           // packed bytes of one (64-row tile, chunk)
    static constexpr int MAXROWS = 1024;                 // rows (Cout x stride, padded to 64) whose bias / prescale table sits in LDS
    static constexpr int LDS = NCH * 2 * XHALF + 256 + 2 * MAXROWS * 4;
    static_assert(LDS <= 160 * 1024, "LDS");
};

// items [item0, hi_item): item = utterance * n_tiles + column tile
template <class G>
__device__ __forceinline__ void convu2_run(const PairParams& p, const PairMember& mb, int item0, int hi_item, float* smem,
                                           int wave, int lane_in) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    char* const ximg = reinterpret_cast<char*>(smem);
    float* const scratch = smem + G::NCH * 2 * G::XHALF / 4;
    float* const tab = scratch + 64;                     // [bias of row m | inverse prescale of row m], MAXROWS each
    const int n = lane & 15, kb = lane >> 4;
    const int ws = wave & 3, wn = wave >> 2;             // row slab of 32, column group of 64
    const int col0 = wn * 64 + n;                        // + 16 f
    const int row0 = 32 * ws + 4 * kb;                   // + 16 h + i: row inside the 128-row tile
    const int nmt = p.nmt, nrt = (nmt + 1) / 2, cout = p.cout;
    const size_t ustride = (size_t)p.ctot * (size_t)p.T;
    const size_t cstride = (size_t)G::C * (size_t)p.T;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(mb.w1, (unsigned)(nmt * G::NCH * G::WTILE));
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(mb.w1 + (size_t)nmt * G::NCH * (G::WTILE / 4), (unsigned)(nmt * 64) * 4u);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(mb.b1 ? mb.b1 : mb.w1, mb.b1 ? (unsigned)cout * 4u : 0u);
    // packed image: [64-row tile][chunk][K step][row sixteenth 4][split half][lane][8 halves]; this wave: sixteenths 2 ws + h of 8
    const unsigned aoff = (unsigned)(((2 * ws) & 3) * 2048 + lane * 16);
    auto wbase = [&](int rt, int c) -> unsigned {
        const int mt = 2 * rt + (ws >> 1);               // (an odd tile count: the last pair's second tile reads zeros)
        return mt < nmt ? (unsigned)((mt * G::NCH + c) * G::WTILE) : kOutOfRange;
    };
    auto load_a_at = [&](unsigned bs, int step, f16x8 (&dst)[2][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                dst[h][e] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    rw, (int)(bs == kOutOfRange ? kOutOfRange : bs + aoff), step * 8192 + h * 2048 + e * 1024, 0));
    };
    // every row's bias and inverse prescale -> LDS, once (held in registers across a row tile's K loops they cost 16 VGPRs the
    // kernel does not have; requested in its epilogue they cost an L2 round trip per row tile)
    for (int m = tid; m < nmt * 64; m += G::NT) {
        tab[m] = buffer_load1(rb, (unsigned)(m / p.ups) * 4u);
        tab[G::MAXROWS + m] = buffer_load1(rs, (unsigned)m * 4u);
    }
    float bad = 0.f;
    LowGuard low;
    ConvHRaw<G> raw;
    int item = item0;
    int b = item / mb.n_tiles, ntile = item - b * mb.n_tiles;
    convh_load_raw<G>(raw, mb.x + b * ustride, p.T, ntile * G::NTC - G::P, tid, true, false);
    f16x8 aq[2][2][2];                                   // K step S (counted over the item's row tiles and chunks) sits in aq[S & 1]
    load_a_at(wbase(0, 0), 0, aq[0]);
    for (;;) {
        const int nitem = item + 1;
        const bool more = nitem < hi_item;
        const int nb = more ? nitem / mb.n_tiles : b, nnt = more ? nitem - nb * mb.n_tiles : ntile;
        // ---- the first chunk's image; the second chunk's window travels during the first row tile's first K loop
        convh_convert<G>(raw, ximg, p.slope, tid, low, 0);
        pair_barrier();                                  // image 0 complete
        if constexpr (G::NCH == 2)
            convh_load_raw<G>(raw, mb.x + b * ustride + cstride, p.T, ntile * G::NTC - G::P, tid, true, false);
#pragma unroll 1
        for (int rt = 0; rt < nrt; ++rt) {
            f32x4 hi[2][G::NFW], lo[2][G::NFW];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bool last_rt = rt + 1 == nrt;
#pragma unroll
            for (int c = 0; c < G::NCH; ++c) {
                if (G::NCH == 2 && c == 1 && rt == 0) {
                    convh_convert<G>(raw, ximg + 2 * G::XHALF, p.slope, tid, low, 0);
                    pair_barrier();                      // image 1 complete
                }
                // the next item's first window: requested in front of the item's last K loop
                if (c == G::NCH - 1 && last_rt)
                    convh_load_raw<G>(raw, mb.x + nb * ustride, p.T, nnt * G::NTC - G::P, tid, more, false);
                // A operands of the K loop that runs after this one: the next chunk, the next row tile, the next item's first
                const unsigned nbase = c + 1 < G::NCH ? wbase(rt, c + 1) : !last_rt ? wbase(rt + 1, 0) : more ? wbase(0, 0) : kOutOfRange;
                const unsigned base = wbase(rt, c);
                const char* const bptr = ximg + c * 2 * G::XHALF + (kb * G::XRP + col0) * 16;
                LdsCF* const bb = lds_opaque(reinterpret_cast<const float*>(bptr));
                LdsCF* const bb2 = lds_opaque(reinterpret_cast<const float*>(bptr + G::XHALF));
                // B operands: the a1 b2 group runs first, so its registers take the NEXT step's second halves while the other two
                // groups run (mrf_mma's queue: b1 double-buffered, b2 single -- 16 registers the kernel does not have otherwise)
                f16x8 b1[2][G::NFW], b2[G::NFW];
                auto fetch_b1 = [&](auto SC, f16x8 (&dst)[G::NFW]) {
                    constexpr int S = decltype(SC)::value;
                    constexpr int off = ((S % G::CG) * 4 * G::XRP + S / G::CG) * 4;
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e) dst[e] = *reinterpret_cast<LdsH8*>(bb + off + e * 64);
                };
                auto fetch_b2 = [&](auto SC) {
                    constexpr int S = decltype(SC)::value;
                    constexpr int off = ((S % G::CG) * 4 * G::XRP + S / G::CG) * 4;
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e) b2[e] = *reinterpret_cast<LdsH8*>(bb2 + off + e * 64);
                };
                fetch_b2(IntC<0>{});
                fetch_b1(IntC<0>{}, b1[0]);
                static_for<0, G::NSTEP>([&](auto SC) {
                    constexpr int S = decltype(SC)::value;
                    f16x8 (&a)[2][2] = aq[S & 1];
                    // (the next step's A operands first: a whole step of MFMAs for their L2 round trip)
                    if constexpr (S + 1 < G::NSTEP) load_a_at(base, S + 1, aq[(S + 1) & 1]);
                    else load_a_at(nbase, 0, aq[0]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < G::NFW; ++e)
                            lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], b2[e], lo[h][e], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (S + 1 < G::NSTEP) {
                        fetch_b2(IntC<S + 1>{});
                        fetch_b1(IntC<S + 1>{}, b1[(S + 1) & 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < G::NFW; ++e)
                            hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], b1[S & 1][e], hi[h][e], 0, 0, 0);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < G::NFW; ++e)
                            lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][1], b1[S & 1][e], lo[h][e], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // ---- the row tile's outputs: y[co][ups u + phase - pad] -- a lane's four rows are four consecutive phases: inside one
            // output channel four consecutive samples, one 16-byte store (convr_run's epilogue)
            const size_t yoff = (size_t)b * (size_t)cout * (size_t)p.Tout;
            const unsigned ybytes = (unsigned)cout * (unsigned)p.Tout * 4u;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(mb.y + yoff, ybytes);
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(mb.y_act ? mb.y_act + yoff : mb.y, mb.y_act ? ybytes : 0u);
            const int t0 = ntile * G::NTC;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m0 = 128 * rt + row0 + 16 * h;
                const int co0 = (int)((unsigned)m0 / (unsigned)p.ups), ph0 = m0 - co0 * p.ups;
                const bool one_row = ph0 + 3 < p.ups;
                const f32x4 bv4 = *reinterpret_cast<const f32x4*>(tab + (m0 < nmt * 64 ? m0 : 0));
                const f32x4 qv4 = *reinterpret_cast<const f32x4*>(tab + G::MAXROWS + (m0 < nmt * 64 ? m0 : 0));
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    const int n0 = (t0 + col0 + f * 16) * p.ups - p.pad_t;
                    float v[4], a[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = fmaf(fmaf(lo[h][f][i], kSplitInv, hi[h][f][i]), qv4[i], bv4[i]);
                        a[i] = act(v[i], p.act_slope);
                        if (!mb.y_act) v[i] = a[i];              // no twin: y itself is stored activated
                    }
                    range_note4(bad, v[0], v[1], v[2], v[3], co0 < cout);
                    if (one_row && co0 < cout && n0 + ph0 >= 0 && n0 + ph0 + 3 < p.Tout) {
                        const unsigned off = (unsigned)(co0 * p.Tout + n0 + ph0) * 4u;
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                                                     __float_as_uint(v[3])}, ry, (int)off, 0, 0);
                        if (mb.y_act)
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]),
                                                                         __float_as_uint(a[3])}, ra, (int)off, 0, 0);
                    } else {
                        int co = co0, ph = ph0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int nn = n0 + ph;
                            const unsigned off = co < cout && nn >= 0 && nn < p.Tout ? (unsigned)(co * p.Tout + nn) * 4u : kOutOfRange;
                            buffer_store1(ry, off, v[i]);
                            if (mb.y_act) buffer_store1(ra, off, a[i]);
                            if (++ph == p.ups) { ph = 0; ++co; }
                        }
                    }
                }
            }
        }
        if (!more) break;
        pair_barrier();                                  // every wave is done with the images
        item = nitem;
        b = nb;
        ntile = nnt;
    }
    range_flag(p, bad);
    pair_barrier();
    low_flag(p, low, scratch, wave, lane, 8);
}

template <int NCH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convu2_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PairParams q;
    q.n_members = 1; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = 1.f;
    q.act_slope = p.act_slope; q.post = 0; q.dbg = 0; q.trace = p.trace;
    q.ctot = p.ctot; q.nch = p.nch; q.nmt = p.nmt; q.reflect = 0; q.ups = p.ups; q.pad_t = p.pad_t; q.Tout = p.Tout; q.cout = p.cout;
    q.guard = p.guard;
    PairMember mb;
    mb.x = p.m[0].x; mb.w1 = p.m[0].w1; mb.b1 = p.m[0].b1; mb.res = nullptr; mb.add1 = nullptr; mb.add2 = nullptr;
    mb.y = p.m[0].y; mb.y_act = p.m[0].y_act; mb.k = 2; mb.n_tiles = p.m[0].n_tiles;
    const int n_items = p.m[0].n_items;
    asm volatile("" ::"s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.act_slope), "s"(q.ctot), "s"(q.nmt), "s"(q.ups),
                 "s"(q.pad_t), "s"(q.Tout), "s"(q.cout), "s"(mb.x), "s"(mb.w1), "s"(mb.b1), "s"(mb.y), "s"(mb.y_act), "s"(mb.n_tiles),
                 "s"(n_items), "s"(q.guard));
    const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int lo = equal_share(share, n_items, q.nblk), hi = equal_share(share + 1, n_items, q.nblk);
    if (lo < hi) convu2_run<ConvU2Geom<NCH>>(q, mb, lo, hi, smem, wave, lane);
}

}  // namespace fv
