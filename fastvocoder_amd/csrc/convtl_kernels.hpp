// ConvTranspose1d (kernel = 2 x stride) with split-f16 operands for launches with FEW items -- the upsamplers at small
// batch (reference model/generator/hifigan.py:93-96, melgan.py:75-85):
//
//     y = conv_transpose1d( lrelu(x, pre_slope); w [Cin, Cout, 2 s], stride s, pad, out_pad ) + bias
//
// Same GEMM, same packed weights (fv_pack_conv_transpose1d_split_f16) and the same sums per output as convt_kernel
// (convh_kernels.hpp: rows m = co s + phase, K = 2 taps x Cin in chunks of 128 / 64 channels, columns u = input sample) --
// identical bits -- on a pipeline cut for launch latency instead of throughput.  convt_kernel streams its weights through an
// LDS ring shared by a block's waves (a barrier per weight stage, a ring to fill before the first MFMA, 128-column tiles):
// at batch 1 an upsampler launch is 1-5 such tiles per CU and spends 21-25 us on 3 GFLOP -- prologue 5 800 cycles, first
// stage entry 3 400, 12 000 per chunk around 3 000 of MFMA issue (docs/history, tools/convt_trace.py).  Here:
//   * a tile is 64 rows x 64 columns (twice the items: 256 for 256 -> 128 x 8 at 1000 frames, one per CU);
//   * 8 waves = 4 row sixteenths x 2 column groups of 32: each wave loads the A operands of its OWN sixteen rows straight from
//     L2 into registers -- all K steps of a chunk up front (64 registers), in flight while the chunk's window is converted --
//     so the K loop has no barrier, no ring and no LDS-DMA: B operands from the LDS image one step ahead, six MFMAs per step;
//   * per chunk: window global -> registers (the NEXT chunk's -- all three tensors of a merged window -- requested behind
//     this chunk's A operands), merge (an upsampler behind an MRF stage forms ((r0 + r1) + r2) / 3 itself: convt_kernel's
//     merge_window), lrelu + split -> LDS image, barrier, K loop, barrier; consecutive row tiles of one window skip all of
//     that but the K loop.  One block per CU, 2 waves per SIMD (<= 256 VGPRs).
// The launcher (launch_convt) picks this kernel when the launch has few 64-column items per CU (Tuning::convt_lean).
#pragma once
#include "convh_kernels.hpp"

namespace fv {

template <int CG_>
struct ConvTLGeom {
    static constexpr int CG = CG_, C = 32 * CG, NT = 512, NW = 8;
    static constexpr int NTC = 64;                       // columns (input samples u) per tile
    static constexpr int P = 1;                          // the window starts one sample early: tap 0 multiplies x[u - 1]
    static constexpr int NSTEP = 2 * CG;                 // K steps of 32 per chunk: tap-major, then 32-channel group
    static constexpr int XROWS = (NTC + 1 + 3) / 4 * 4;
    static constexpr int CB = C / 8;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;   // image: [split half][8-channel block][XRP rows][8 halves]
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;
    static constexpr int WTILE = NSTEP * 8192;           // packed bytes of one (64-row tile, chunk)
    static constexpr int LDS = 2 * XHALF + 256;
};

// items [item0, hi_item); item = (utterance * n_tiles + column tile) * p.nmt + row tile (row tile fastest: a block's
// consecutive items share their window's cache lines)
template <class G>
__device__ __forceinline__ void convtl_run(const PairParams& p, const PairMember& mb, int item0, int hi_item, float* smem,
                                           int wave, int lane_in) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    char* const ximg = reinterpret_cast<char*>(smem);
    float* const scratch = smem + 2 * G::XHALF / 4;
    const int n = lane & 15, kb = lane >> 4;
    const int mh = wave & 3, wn = wave >> 2;
    const int col0 = wn * 32 + n;                                           // + 16 f
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;              // B: + (4 cg XRP + tap + 16 f) 16 (+ XHALF)
    const int row0 = 16 * mh + 4 * kb;                                      // + i: row inside the 64-row tile
    const int nch = p.nch, nmt = p.nmt, cout = p.cout;
    const size_t ustride = (size_t)p.ctot * (size_t)p.T;
    const size_t cstride = (size_t)G::C * (size_t)p.T;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(mb.w1, (unsigned)(nmt * nch * G::WTILE));
    auto decode = [&](int it, int& b, int& nt, int& mt) {
        mt = it % nmt;
        const int q = it / nmt;
        b = q / mb.n_tiles;
        nt = q - b * mb.n_tiles;
    };
    auto chunk_channels = [&](int c) { return min(G::C, p.ctot - c * G::C); };
    const bool merge = mb.add1 != nullptr;
    const float mrcp = div_rcp(p.out_div);
    float bad = 0.f;
    LowGuard low;
    // The window of a chunk is up to three tensors (an upsampler behind an MRF stage forms ((r0 + r1) + r2) / 3 itself): all of
    // them are requested a chunk ahead, behind the barrier that completes the current image, and land during the K loop, the
    // epilogue and the stores [round 5, first form: only x travelled ahead; the two addends were requested where they were
    // needed, a full memory latency per item].
    ConvHRaw<G> raw, rt, ru;
    auto load_window = [&](int bb, int cc, int nt_) {
        const size_t off = (size_t)bb * ustride + (size_t)cc * cstride;
        convh_load_raw<G>(raw, mb.x + off, p.T, nt_ * G::NTC - G::P, tid, true, false, chunk_channels(cc));
        if (merge) {
            convh_load_raw<G>(rt, mb.add1 + off, p.T, nt_ * G::NTC - G::P, tid, true, false, chunk_channels(cc));
            convh_load_raw<G>(ru, mb.add2 ? mb.add2 + off : mb.add1 + off, p.T, nt_ * G::NTC - G::P, tid, mb.add2 != nullptr, false,
                              chunk_channels(cc));
        }
    };
    int item = item0, chunk = 0, b, ntile, mtile;
    decode(item, b, ntile, mtile);
    pair_stamp(p, 8, wave, lane, 7, 12);                 // (tuning aid, -DFV_PAIR_TRACE: tools/convtl_trace.py) run start
    load_window(b, 0, ntile);
    pair_stamp(p, 8, wave, lane, 7, 13);
    int traced = 0;                                      // chunks stamped so far (the first seven of a block)
    bool fresh = true;                                   // the image has to be built from the window in the registers
    f32x4 hi[2], lo[2];
    float bv[4], sv[4];
    for (;;) {
        int nchunk = chunk + 1, nitem = item;
        if (nchunk == nch) {
            nchunk = 0;
            nitem = item + 1;
        }
        const bool last = nchunk == 0, more = nitem < hi_item;
        int nb = b, nnt = ntile, nmt_ = mtile;
        if (more && last) decode(nitem, nb, nnt, nmt_);
        pair_stamp(p, 8, wave, lane, traced, 0);
        // ---- this chunk's A operands: the wave's sixteen rows, every K step, L2 -> registers (in flight during the conversion)
        f16x8 A[G::NSTEP][2];
        {
            const unsigned wbase = (unsigned)((mtile * nch + chunk) * G::WTILE + mh * 2048 + lane * 16);
#pragma unroll
            for (int st = 0; st < G::NSTEP; ++st) {
                A[st][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(wbase + st * 8192), 0, 0));
                A[st][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(wbase + st * 8192 + 1024), 0, 0));
            }
        }
        if (chunk == 0) {
            hi[0] = hi[1] = lo[0] = lo[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            // bias and inverse row prescales of this item's rows (consumed in the epilogue)
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(mb.w1 + (size_t)nmt * nch * (G::WTILE / 4), (unsigned)(nmt * 64) * 4u);
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(mb.b1 ? mb.b1 : mb.w1, mb.b1 ? (unsigned)cout * 4u : 0u);
            int co = (int)((unsigned)(64 * mtile + row0) / (unsigned)p.ups), ph = 64 * mtile + row0 - co * p.ups;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bv[i] = buffer_load1(rb, (unsigned)co * 4u);
                if (++ph == p.ups) { ph = 0; ++co; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[i] = buffer_load1(rs, (unsigned)(64 * mtile + row0 + i) * 4u);
        }
        // ---- the window: merge (an upsampler behind an MRF stage), lrelu, split -> image.  Consecutive items of a block are
        // row tiles of the SAME window more often than not (row tile fastest in the item order): a one-chunk window then stays
        // where it is -- no loads, no conversion, no barriers.
        if (fresh) {
            if (merge) {
#pragma unroll
                for (int q = 0; q < G::XR; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float v = (raw.v[q][j] + rt.v[q][j]) + ru.v[q][j];   // (no add2: ru is zeros, and v + 0 = v)
                        if (p.out_div != 1.f) v = mrcp != 0.f ? div_exact(v, p.out_div, mrcp) : v / p.out_div;
                        raw.v[q][j] = v;
                    }
            }
            convh_convert<G>(raw, ximg, p.slope, tid, low, 0);
            pair_stamp(p, 8, wave, lane, traced, 1);
            pair_barrier();                              // image complete
        }
        pair_stamp(p, 8, wave, lane, traced, 2);
        const bool reuse = more && last && nch == 1 && nb == b && nnt == ntile;
        if ((more || !last) && !reuse) load_window(last ? nb : b, last ? 0 : nchunk, last ? nnt : ntile);
        // ---- K loop: no barrier, B operands one step ahead
        {
            LdsCF* const bb = lds_opaque(reinterpret_cast<const float*>(bptr));
            LdsCF* const bb2 = lds_opaque(reinterpret_cast<const float*>(bptr + G::XHALF));
            f16x8 B[2][2][2];
            auto fetch_b = [&](int st, f16x8 (&dst)[2][2]) {
                const int tap = st / G::CG, cg = st % G::CG;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int off = (cg * 4 * G::XRP + tap + f * 16) * 4;   // in floats
                    dst[f][0] = *reinterpret_cast<LdsH8*>(bb + off);
                    dst[f][1] = *reinterpret_cast<LdsH8*>(bb2 + off);
                }
            };
            fetch_b(0, B[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < G::NSTEP; ++st) {
                if (st + 1 < G::NSTEP) fetch_b(st + 1, B[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < 2; ++f) hi[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[st][0], B[st & 1][f][0], hi[f], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < 2; ++f) lo[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[st][0], B[st & 1][f][1], lo[f], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < 2; ++f) lo[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[st][1], B[st & 1][f][0], lo[f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        pair_stamp(p, 8, wave, lane, traced, 3);
        // ---- epilogue of the item's last chunk: y[co][ups u + phase - pad] -- a lane's four rows are four consecutive phases:
        // inside one output channel four consecutive samples, one 16-byte store (convt_kernel's epilogue)
        if (last) {
            const size_t yoff = (size_t)b * (size_t)cout * (size_t)p.Tout;
            const unsigned ybytes = (unsigned)cout * (unsigned)p.Tout * 4u;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(mb.y + yoff, ybytes);
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(mb.y_act ? mb.y_act + yoff : mb.y, mb.y_act ? ybytes : 0u);
            const int co0 = (int)((unsigned)(64 * mtile + row0) / (unsigned)p.ups), ph0 = 64 * mtile + row0 - co0 * p.ups;
            const bool one_row = ph0 + 3 < p.ups;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int n0 = (ntile * G::NTC + col0 + f * 16) * p.ups - p.pad_t;
                float v[4], a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = fmaf(fmaf(lo[f][i], kSplitInv, hi[f][i]), sv[i], bv[i]);
                    a[i] = act(v[i], p.act_slope);
                    if (!mb.y_act) v[i] = a[i];              // no twin: y itself is stored activated
                }
                range_note4(bad, v[0], v[1], v[2], v[3], true);
                if (one_row && n0 + ph0 >= 0 && n0 + ph0 + 3 < p.Tout) {
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
                        const unsigned off = nn >= 0 && nn < p.Tout ? (unsigned)(co * p.Tout + nn) * 4u : kOutOfRange;
                        buffer_store1(ry, off, v[i]);
                        if (mb.y_act) buffer_store1(ra, off, a[i]);
                        if (++ph == p.ups) { ph = 0; ++co; }
                    }
                }
            }
        }
        pair_stamp(p, 8, wave, lane, traced, 4);
        if (!more && last) break;
        if (!reuse) pair_barrier();                      // every wave is done with the image: the next window may overwrite it
        pair_stamp(p, 8, wave, lane, traced, 5);
        ++traced;
        fresh = !reuse;
        item = nitem;
        chunk = nchunk;
        b = nb;
        ntile = nnt;
        mtile = nmt_;
    }
#ifdef FV_PAIR_TRACE
    pair_wait_vm0();
    pair_stamp(p, 8, wave, lane, 7, 14);                 // the stores have drained
#endif
    range_flag(p, bad);
    pair_barrier();
    low_flag(p, low, scratch, wave, lane, G::NW);
}

template <int CG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convtl_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PairParams q;
    q.n_members = 1; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = p.out_div;
    q.act_slope = p.act_slope; q.post = 0; q.dbg = p.dbg; q.trace = p.trace;
    q.ctot = p.ctot; q.nch = p.nch; q.nmt = p.nmt; q.reflect = 0; q.ups = p.ups; q.pad_t = p.pad_t; q.Tout = p.Tout; q.cout = p.cout;
    q.guard = p.guard;
    PairMember mb;
    mb.x = p.m[0].x; mb.w1 = p.m[0].w1; mb.b1 = p.m[0].b1; mb.res = nullptr; mb.add1 = p.m[0].add1; mb.add2 = p.m[0].add2;
    mb.y = p.m[0].y; mb.y_act = p.m[0].y_act; mb.k = 2; mb.n_tiles = p.m[0].n_tiles;
    const int n_items = p.m[0].n_items;
    asm volatile("" ::"s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.act_slope), "s"(q.ctot), "s"(q.nch), "s"(q.nmt), "s"(q.ups),
                 "s"(q.pad_t), "s"(q.Tout), "s"(q.cout), "s"(mb.x), "s"(mb.w1), "s"(mb.b1), "s"(mb.y), "s"(mb.y_act), "s"(mb.n_tiles),
                 "s"(n_items), "s"(q.guard), "s"(q.out_div), "s"(mb.add1), "s"(mb.add2));
    const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int lo = equal_share(share, n_items, q.nblk), hi = equal_share(share + 1, n_items, q.nblk);
    if (lo < hi) convtl_run<ConvTLGeom<CG>>(q, mb, lo, hi, smem, wave, lane);
}

}  // namespace fv
