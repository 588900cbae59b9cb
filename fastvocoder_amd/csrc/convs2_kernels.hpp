// 'Same' conv with split-f16 operands at 256 / 512 channels WITHOUT the weight ring (round 5; HiFi-GAN large's 256-channel
// ResBlocks, reference model/generator/modules.py:223-230 at conf/hifigan/large.yaml):
//
//     y = post( ( conv1d( lrelu(x, slope); w, KT taps, dilation DIL ) + bias + res + add1 + add2 ) / out_div )
//
// convs_kernel (convr_kernels.hpp) streams its weights through an LDS ring shared by the block's waves: a barrier per K step
// (12 ... 44 per 128-channel chunk) and MFMA-busy 0.38 at saturation.  This is the transformation that took the 128-channel
// pairs from the ring form to convq2_kernel: a block owns 128 rows x 64 columns, 8 waves = 8 row slabs of SIXTEEN rows x one
// column group -- no two waves share a row of weights, so every wave loads the A operands of its own 16 rows straight from
// L2 into registers (2 x 16 bytes per lane and K step, QD = 3 steps ahead in a queue that runs on from chunk to chunk and from
// item to item) -- no LDS-DMA, no ring, and NO barrier inside a chunk's K loop; two per chunk remain (image free / image
// complete).  The window of a 128-channel chunk (64 + (KT - 1) DIL rows) is loaded a chunk ahead and converted once per
// (item, chunk).  Same packed weights (fv_pack_conv1d_split_f16), same K order per output (chunks in order, tap-major steps),
// same MFMA order per accumulator, the epilogue's arithmetic: identical bits to convs_kernel (tests/test_gpu_pairs.py).
#pragma once
#include "convh_kernels.hpp"

namespace fv {

// NH_ = 1: 8 row slabs of 16 x one column group (64-column tiles); NH_ = 2: 4 row slabs of 32 x two column groups (128-column
// tiles: half the A loads, LDS reads, conversions and barriers per MFMA -- for launches with tiles to spare; convq2's wide form)
template <int KT_, int DIL_, int NH_>
struct ConvS2Geom {
    static constexpr int KT = KT_, DIL = DIL_, NH = NH_;
    static constexpr int C = 128, CG = 4, CB = 16, NFW = 4, NT = 512;      // per chunk: 128 input channels
    static constexpr int NTC = 64 * NH;                  // output columns per tile
    static constexpr int NSTEP = KT * CG;                // K steps of 32 per chunk, tap-major
    static constexpr int P = (KT - 1) * DIL / 2;
    static constexpr int XROWS = (NTC + (KT - 1) * DIL + 3) / 4 * 4;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;   // image: [split half][8-channel block][XRP rows][8 halves]
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;
    static constexpr int NRAW = XR * 8;
    static constexpr int WTILE = NSTEP * 8192;           // packed bytes of one (64-row tile, chunk): [step][8 KB]
    static constexpr int QD = NH == 2 ? 1 : 3, NA = 2 * NH;   // A operands QD steps ahead; loads per wave and K step
    static constexpr int RAWK = NSTEP - QD - 1;          // K step at which the next window is requested
    static constexpr int LDS = 2 * XHALF + 256;
    static_assert(NSTEP > 2 * QD, "taps");
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * (NFW - 1)) * 16 + 16 < 65536, "ds_read immediate range");
};

// items [item0, hi_item) of one member: item = ((utterance * n_tiles) + column tile) * nrt + row tile of 128 (fastest)
template <class G>
__device__ __forceinline__ void convs2_run(const PairParams& p, const PairMember& mb, int item0, int hi_item, float* smem,
                                           int wave, int lane_in, bool first) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    char* const ximg = reinterpret_cast<char*>(smem);
    float* const scratch = smem + 2 * G::XHALF / 4;
    const int n = lane & 15, kb = lane >> 4;
    const int ws = wave % (8 / G::NH), wn = wave / (8 / G::NH);            // row slab of 16 NH rows, column group of 64
    const int col0 = wn * 64 + n;                        // + 16 f
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;
    const int row0 = 16 * G::NH * ws + 4 * kb;           // + 16 h + i: row inside the 128-row tile
    const int nch = p.nch, nrt = p.nmt / 2;
    const size_t ustride = (size_t)p.ctot * (size_t)p.T;
    const size_t cstride = (size_t)G::C * (size_t)p.T;
    const unsigned ubytes = (unsigned)p.ctot * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(mb.w1, (unsigned)(p.nmt * nch * G::WTILE));
    // packed image: [64-row tile][chunk][K step][row sixteenth 4][split half][lane][8 halves]; this wave: sixteenths NH ws + h of 8
    const int s16 = G::NH * ws;
    const unsigned aoff = (unsigned)((s16 & 3) * 2048 + lane * 16);
    auto wbase = [&](int rt, int c) -> unsigned { return (unsigned)(((2 * rt + (s16 >> 2)) * nch + c) * G::WTILE); };
    auto decode = [&](int it, int& b, int& nt, int& rt) {
        rt = it % nrt;
        const int q = it / nrt;
        b = q / mb.n_tiles;
        nt = q - b * mb.n_tiles;
    };
    int item = item0, chunk = 0, b, ntile, rt;
    decode(item, b, ntile, rt);
    if (!first) pair_barrier();                          // everybody is done with the previous member's LDS
    LowGuard low;
    float bad = 0.f;
    ConvHRaw<G> raw;
    convh_load_raw<G>(raw, mb.x + b * ustride, p.T, ntile * G::NTC - G::P, tid, true, false);
    f16x8 aq[G::QD + 1][G::NH][2];                       // K step S of a chunk sits in aq[S % (QD + 1)] (NSTEP % (QD + 1) == 0)
    static_assert(G::NSTEP % (G::QD + 1) == 0, "the A queue runs on from chunk to chunk");
    unsigned base = wbase(rt, 0);
    auto load_a_at = [&](unsigned bs, int step, f16x8 (&dst)[G::NH][2]) {
#pragma unroll
        for (int h = 0; h < G::NH; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                dst[h][e] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    rw, (int)(bs == kOutOfRange ? kOutOfRange : bs + aoff), step * 8192 + h * 2048 + e * 1024, 0));
    };
    static_for<0, G::QD>([&](auto QC) { load_a_at(base, decltype(QC)::value, aq[decltype(QC)::value]); });
    pair_wait_vm0();
    convh_convert<G>(raw, ximg, p.slope, tid, low, 0);
    pair_barrier();                                      // image complete
    f32x4 hi[G::NH][G::NFW], lo[G::NH][G::NFW];          // live across the chunks of an item
    float bv[G::NH][4], qv[G::NH][4];
    for (;;) {
        int nchunk = chunk + 1, nitem = item;
        if (nchunk == nch) {
            nchunk = 0;
            nitem = item + 1;
        }
        const bool last = nchunk == 0, more = nitem < hi_item;
        int nb = b, nnt = ntile, nrt_ = rt;
        if (more && last) decode(nitem, nb, nnt, nrt_);
        const unsigned nbase = (more || !last) ? wbase(last ? nrt_ : rt, nchunk) : kOutOfRange;
        if (chunk == 0) {
#pragma unroll
            for (int h = 0; h < G::NH; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (last) {
            // the item's bias and inverse row prescales (behind the packed image): older than every A operand of this chunk,
            // so the K loop's wait counts do not see them; consumed in the epilogue
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(mb.b1 ? mb.b1 : mb.w1, mb.b1 ? (unsigned)p.ctot * 4u : 0u);
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(mb.w1 + (size_t)p.nmt * nch * (G::WTILE / 4), (unsigned)(p.nmt * 64) * 4u);
#pragma unroll
            for (int h = 0; h < G::NH; ++h) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[h][i] = buffer_load1(rb, (unsigned)(128 * rt + row0 + 16 * h + i) * 4u);
#pragma unroll
                for (int i = 0; i < 4; ++i) qv[h][i] = buffer_load1(rs, (unsigned)(128 * rt + row0 + 16 * h + i) * 4u);
            }
        }
        f16x8 bbuf[2][G::NFW][2];
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
        // ---- the chunk's K loop: no barrier.  Loads return in order: step S's operands have landed once at most the loads
        // issued after them are outstanding -- QD steps' worth, plus the next window where it was requested in between.  (The
        // first QD steps of a chunk were waited for at the chunk boundary: vmcnt(0) in front of the conversion.)
        fetch_b(IntC<0>{}, bbuf[0]);
        static_for<0, G::NSTEP>([&](auto SC) {
            constexpr int S = decltype(SC)::value;
            if constexpr (S == G::RAWK)
                convh_load_raw<G>(raw, mb.x + (last ? nb : b) * ustride + (last ? 0 : nchunk) * cstride, p.T,
                                  (last ? nnt : ntile) * G::NTC - G::P, tid, more || !last, false);
            constexpr bool raw_after = G::RAWK > S - G::QD && G::RAWK <= S;
            if constexpr (S + G::QD < G::NSTEP) load_a_at(base, S + G::QD, aq[(S + G::QD) % (G::QD + 1)]);
            else load_a_at(nbase, S + G::QD - G::NSTEP, aq[(S + G::QD) % (G::QD + 1)]);
            if constexpr (S + 1 < G::NSTEP) fetch_b(IntC<S + 1>{}, bbuf[(S + 1) & 1]);
            if constexpr (S >= G::QD) wait_vm<G::NA * G::QD + (raw_after ? G::NRAW : 0)>();
            __builtin_amdgcn_sched_barrier(0);
            f16x8 (&a)[G::NH][2] = aq[S % (G::QD + 1)];
#pragma unroll
            for (int h = 0; h < G::NH; ++h)
#pragma unroll
                for (int e = 0; e < G::NFW; ++e) hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bbuf[S & 1][e][0], hi[h][e], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < G::NH; ++h)
#pragma unroll
                for (int e = 0; e < G::NFW; ++e) lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bbuf[S & 1][e][1], lo[h][e], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < G::NH; ++h)
#pragma unroll
                for (int e = 0; e < G::NFW; ++e) lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][1], bbuf[S & 1][e][0], lo[h][e], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        pair_barrier();                                  // every wave is done with the image
        // ---- the next window first (requested at step RAWK), the item's stores behind it: vmcnt counts stores too, and a
        // conversion behind the stores would wait for their acknowledgement [measured: config 5 102.0 -> see DESIGN].  The K
        // loop's wait counts stay valid with stores in flight: loads return in order among themselves, so "at most NA QD
        // operations outstanding" still means that every load older than the newest NA QD has landed.
        const bool cont = more || !last;
        if (cont) {
            convh_convert<G>(raw, ximg, p.slope, tid, low, 0);
            pair_barrier();                              // image complete
        }
        // ---- the item's outputs (convr_run's epilogue) ----
        if (last) {
            const int t0 = ntile * G::NTC;
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(mb.res ? mb.res + b * ustride : mb.w1, mb.res ? ubytes : 0u);
#pragma unroll
            for (int h = 0; h < G::NH; ++h) {
                const int rowt = 128 * rt + row0 + 16 * h;
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    const int t = t0 + col0 + f * 16;
                    const unsigned voff = t < p.T ? (unsigned)(rowt * p.T + t) * 4u : kOutOfRange;
                    float rv[4] = {0.f, 0.f, 0.f, 0.f}, v[4];
                    if (mb.res != nullptr) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) rv[i] = buffer_load1s(rr, voff, (unsigned)i * t4);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaf(fmaf(lo[h][f][i], kSplitInv, hi[h][f][i]), qv[h][i], bv[h][i]) + rv[i];
                    if (mb.add1 != nullptr) {
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
                    pair_store(p, mb.y, mb.y_act, p.ctot, b, rowt, t, t < p.T, v, mb.add1 != nullptr);
                }
            }
        }
        if (!cont) break;
        base = nbase;
        item = nitem;
        chunk = nchunk;
        b = nb;
        ntile = nnt;
        rt = nrt_;
    }
    pair_wait_vm0();
    range_flag(p, bad);
    pair_barrier();
    low_flag(p, low, scratch, wave, lane, 8);
}

// members of a launch as in convs_kernel (contiguous cost-weighted shares, or the host's block schedule)
template <int DIL, int NH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convs2_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PairParams q;
    q.n_members = p.n_members; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = p.out_div;
    q.act_slope = p.act_slope; q.post = p.post; q.dbg = 0; q.trace = p.trace;
    q.ctot = p.ctot; q.nch = p.nch; q.nmt = p.nmt; q.reflect = 0; q.guard = p.guard; q.sub = nullptr; q.sub_batched = 0;
    int n_items[3], cost[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) { n_items[m] = p.m[m].n_items; cost[m] = p.m[m].cost; }
    asm volatile("" ::"s"(q.n_members), "s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.out_div), "s"(q.act_slope),
                 "s"(q.post), "s"(q.trace), "s"(n_items[0]), "s"(n_items[1]), "s"(n_items[2]), "s"(cost[0]), "s"(cost[1]),
                 "s"(cost[2]), "s"(q.ctot), "s"(q.nch), "s"(q.nmt), "s"(q.guard));
    const bool cut = p.sched_on == 2;                    // the contiguous cut as a table (pair_cut_schedule)
    int g_lo = 0, g_hi = 0;
    if (cut) {
        const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
        g_lo = (int)p.sched[share];
        g_hi = share + 1 < q.nblk ? (int)p.sched[share + 1] : n_items[0] + (q.n_members > 1 ? n_items[1] : 0) + (q.n_members > 2 ? n_items[2] : 0);
        asm volatile("" ::"s"(g_lo), "s"(g_hi));
    }
    long long total = 0;
    if (!cut) {
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
        if (cut) {
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
        if (mb.k == 11) convs2_run<ConvS2Geom<11, DIL, NH>>(q, mb, lo, hi, smem, wave, lane, first);
        else if (mb.k == 7) convs2_run<ConvS2Geom<7, DIL, NH>>(q, mb, lo, hi, smem, wave, lane, first);
        else convs2_run<ConvS2Geom<3, DIL, NH>>(q, mb, lo, hi, smem, wave, lane, first);
        first = false;
    }
}

}  // namespace fv
