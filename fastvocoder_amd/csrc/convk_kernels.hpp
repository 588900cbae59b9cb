// MelGAN ResidualStack as ONE launch, 32 / 64 / 128 channels, split-f16 operands:
//
//     y = W2 lrelu( W1 (*) lrelu( pad(x) ) + b1 ) + Ws x + (b2 + bs)        (reference model/generator/modules.py:351-382:
//                                                                             stack = [act, pad, conv k=3 dilated, act, conv 1x1],
//                                                                             skip_layer = conv 1x1 of the raw input)
//
// The two-launch form (convh_kernel for the dilated conv, convg_kernel for the K-concatenated 1x1 pair; 64 channels and
// fewer: the fp32 kernels) writes the hidden tensor to HBM and reads it and x back; at MelGAN's sizes (T = 200 frames,
// batch 1: 1 600 ... 51 200 columns) a launch is a single tile per block and its fixed costs, twice.  Here a block owns
// ALL C rows of a column tile, as convq_kernel does for the ResBlock pair (convq_kernels.hpp): 8 waves = C / 32 row slabs
// x 256 / C column groups of 32 columns -- 64-column tiles at 128 channels, 128 at 64, 256 at 32: the same 32 KB of
// intermediate whatever C.  A tile walks ONE sequence of 5 C / 32 weight stages through a ring of three:
//
//     stages [0, 3 CG)        conv1: taps x 32-channel groups, B operands from the activated window's split image
//     -- accumulators (+ b1, lrelu) -> split image of the hidden tile in LDS; the RAW window (still in the registers
//        it was prefetched into) -> split image of the tile's centre columns, over the activated image, which is dead
//     stages [3 CG, 4 CG)     W2 x hidden tile
//     stages [4 CG, 5 CG)     Ws x raw centre                          (the K order of convg_kernel: hidden, then x)
//
// Same K order, same splits, same epilogue arithmetic as convh_kernel + convg_kernel at 128 channels: identical bits
// (tests/test_gpu_stack.py).  LDS: ring 3 x C x 128 B (48 KB at 128 channels) + window image (<= 48 KB) + hidden image
// 32 KB + biases and row scales: <= 131 KB, one block per CU.
#pragma once
#include "convh_kernels.hpp"

namespace fv {

struct ConvKParams {
    const float* x;      // [B, C, T]
    const float* w;      // packed (pack_convk_kernel): [stage][row sixteenth][split half][lane][8 f16], then the rows'
                         // inverse prescales of conv1 [C] and of [W2 | Ws] [C]
    const float* b1;     // [C] or null
    const float* b2;     // [C] or null: b2 + bs
    float* y;            // [B, C, T]
    float* y_act;        // optional activated twin lrelu(y, act_slope), or null (then act_slope != 1: y is stored activated)
    int B, T, n_tiles, n_items, nblk;
    float slope, act_slope;
    int post;            // FV_POST_* applied to y (the graph's last stack: Basis-MelGAN's final ReLU)
    const float* sub;    // optional output offset (fv_plan_set_output_offset): y2 = act(post(y)) - sub, or y itself when there
    int sub_batched;     //   is no second output (convh_kernels.hpp's rule); [C, T] or (sub_batched) [B, C, T]
    int reflect;         // rows outside [0, T): mirrored samples (ReflectionPad1d) instead of zeros
    int* guard;
};

template <int CG_, int DIL_>
struct ConvKGeom {
    static constexpr int CG = CG_, DIL = DIL_, KT = 3, C = 32 * CG, CB = C / 8, NFW = 2, NT = 512;
    static constexpr int WN = 8 / CG;                    // column groups of 32
    static constexpr int NM = 32 * WN;                   // columns per tile
    static constexpr int P = DIL;                        // 'same' padding of the 3-tap conv
    static constexpr int XROWS = (NM + 2 * DIL + 3) / 4 * 4;
    static constexpr int XRP = (XROWS + 15) / 16 * 16;   // image: [split half][8-channel block][XRP rows][8 halves]
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;
    static constexpr int MRP = NM, MHALF = CB * MRP * 16;   // hidden tile / raw centre: [split half][block][NM rows][8 halves]
    static constexpr int NS1 = KT * CG, NSH = CG, NST = NS1 + 2 * CG;
    static constexpr int STAGE_BYTES = C * 128, RING = 3, AHEAD = RING - 1;
    static constexpr int NDMA = STAGE_BYTES >= 16384 ? 2 : 1;      // LDS-DMA instructions (1 KB each) per wave and stage
    static constexpr int DW = STAGE_BYTES / (1024 * NDMA);         // waves that carry a stage (4 at 32 channels)
    static constexpr int NRAW = XR * 8;
    static constexpr int WBYTES = NST * STAGE_BYTES;
    static constexpr int RING_BYTES = RING * STAGE_BYTES;
    static constexpr int LDS_BYTES = RING_BYTES + 2 * XHALF + 2 * MHALF + (4 * C + 16) * 4;
    static_assert(2 * MHALF <= 2 * XHALF, "the raw centre lies over the window image");
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * (NFW - 1)) * 16 + 16 < 65536, "ds_read immediate range");
    static_assert(NST > AHEAD && DW <= 8, "stages");
};

// the centre columns [P, P + NM) of the raw window -> their split image (no activation: the skip branch reads x itself)
template <class G>
__device__ __forceinline__ void convk_convert_centre(const ConvHRaw<G>& r, char* rimg, int tid) {
#pragma unroll
    for (int q = 0; q < G::XR; ++q) {
        const int idx = tid + q * G::NT;
        const int cb = idx / G::XROWS, row = idx - cb * G::XROWS;
        if (idx < G::XROWS * G::CB && row >= G::P && row < G::P + G::NM) {
            F16x8Parts h1, h2;
#pragma unroll
            for (int j = 0; j < 4; ++j) split2(split_act2(f32x2{r.v[q][2 * j], r.v[q][2 * j + 1]}, 1.f), h1.p[j], h2.p[j]);
            *reinterpret_cast<f16x8*>(rimg + (cb * G::MRP + row - G::P) * 16) = __builtin_bit_cast(f16x8, h1);
            *reinterpret_cast<f16x8*>(rimg + (cb * G::MRP + row - G::P) * 16 + G::MHALF) = __builtin_bit_cast(f16x8, h2);
        }
    }
}

// one 8-wave block per CU, 2 waves per SIMD; blocks own contiguous runs of (utterance, column tile) items
template <int CG, int DIL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convk_kernel(ConvKParams p) {
    typedef ConvKGeom<CG, DIL> G;
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int item = equal_share(share, p.n_items, p.nblk);
    const int hi_item = equal_share(share + 1, p.n_items, p.nblk);
    if (item >= hi_item) return;

    float* const ring = smem;
    char* const ximg = reinterpret_cast<char*>(smem) + G::RING_BYTES;
    char* const mimg = ximg + 2 * G::XHALF;
    float* const bl = reinterpret_cast<float*>(mimg + 2 * G::MHALF);      // [b1 | b2 | inverse row prescales 1 | 2 | guard scratch]
    const int n = lane & 15, kb = lane >> 4;
    const int ws = wave / G::WN, wn = wave % G::WN;      // row slab of 32, column group of 32
    const int col0 = wn * 32 + n;
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;
    const char* const mptr = mimg + (kb * G::MRP + col0) * 16;
    const char* const rptr = ximg + (kb * G::MRP + col0) * 16;
    const float* const aptr = ring + (2 * ws) * 512 + lane * 4;   // + slot stage + (h * 2 + split half) * 256 floats
    const int row0 = 32 * ws + 4 * kb;                   // + 16 h + i
    // D fragment -> hidden image: channels row0 + 16 h + i = half of the 8-channel block 4 ws + 2 h + (kb >> 1)
    char* const mw = mimg + ((4 * ws + (kb >> 1)) * G::MRP + col0) * 16 + 8 * (kb & 1);

    const size_t ustride = (size_t)G::C * (size_t)p.T;
    const unsigned ubytes = (unsigned)G::C * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (unsigned)G::WBYTES);
    int g0 = 0;                                          // ring slot of the tile's stage 0
    auto slot_of = [&](int g) {                          // g: compile-time stage number inside the tile (or a little beyond)
        int s = g0 + g % G::RING;
        return s >= G::RING ? s - G::RING : s;
    };
    auto dma_stage = [&](int slot, unsigned stage_off) {
        if (wave < G::DW) {
            float* dst = ring + slot * (G::STAGE_BYTES / 4) + wave * (256 * G::NDMA);
            const unsigned o = stage_off == kOutOfRange ? kOutOfRange : stage_off + (unsigned)(wave * (1024 * G::NDMA) + lane * 16);
            dma16(rw, dst, o);
            if constexpr (G::NDMA == 2) dma16(rw, dst + 256, o == kOutOfRange ? kOutOfRange : o + 1024u);
        }
    };
    int b = item / p.n_tiles, tile = item - b * p.n_tiles;
    LowGuard low;                                        // low side of the range guard (pairh_kernels.hpp)
    f32x2 bad2 = {0.f, 0.f};                             // range guard (pairh_kernels.hpp range_note4p)
    ConvHRaw<G> raw;
    convh_load_raw<G>(raw, p.x + b * ustride, p.T, tile * G::NM - G::P, tid, true, p.reflect != 0);
#pragma unroll
    for (int st = 0; st < G::AHEAD; ++st) dma_stage(st, (unsigned)(st * G::STAGE_BYTES));
    if (tid < G::C) {
        bl[tid] = p.b1 ? p.b1[tid] : 0.f;
        bl[G::C + tid] = p.b2 ? p.b2[tid] : 0.f;
        bl[2 * G::C + tid] = p.w[G::WBYTES / 4 + tid];
        bl[3 * G::C + tid] = p.w[G::WBYTES / 4 + G::C + tid];
    }
    pair_wait_vm0();
    convh_convert<G>(raw, ximg, p.slope, tid, low, 0);
    for (;;) {
        const int t0 = tile * G::NM;
        const int nitem = item + 1;
        const bool more = nitem < hi_item;
        int nb = b, ntile = tile + 1;
        if (ntile == p.n_tiles) {
            ntile = 0;
            ++nb;
        }
        f32x4 hi[2][G::NFW], lo[2][G::NFW];
        f16x8 abuf[2][2][2], bbuf[2][2][2];

        // ---- stage entry: the stage's weights are in its ring slot for every wave; every wave holds the A operands of the
        // stage before in registers, so that slot is free: request the stage AHEAD further on into it
        auto entry = [&](auto GC) {
            constexpr int GS = decltype(GC)::value;
            // this stage's DMA was issued AHEAD entries ago; loads return in order: it has landed once at most as many loads
            // are outstanding as were issued after it -- the DMAs of the entries in between, plus the next tile's raw window
            // where that was requested in between (after entry NS1: below).  GS < AHEAD: landed before the previous tile's
            // epilogue issued its stores (no count: it would wait for those stores)
            constexpr bool raw_between = GS >= G::NS1 + 1 && GS <= G::NS1 + G::AHEAD;
            if constexpr (GS >= G::AHEAD) wait_vm<G::NDMA * (G::AHEAD - 1) + (raw_between ? G::NRAW : 0)>();
            pair_barrier();
            constexpr int NS = GS + G::AHEAD;            // this tile's stage NS, or the next tile's NS - NST
            if constexpr (NS < G::NST) dma_stage(slot_of(NS), (unsigned)(NS * G::STAGE_BYTES));
            else dma_stage(slot_of(NS), more ? (unsigned)((NS - G::NST) * G::STAGE_BYTES) : kOutOfRange);
        };
        auto fetch_a = [&](auto SC, f16x8 (&dst)[2][2]) {
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
        LdsCF* const rb1 = lds_opaque(reinterpret_cast<const float*>(rptr));
        LdsCF* const rb2 = lds_opaque(reinterpret_cast<const float*>(rptr + G::MHALF));
        // B operands of stage S: the window image (tap stride DIL), the hidden tile, the raw centre
        auto fetch_b = [&](auto SC, f16x8 (&dst)[2][2]) {
            constexpr int S = decltype(SC)::value;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if constexpr (S < G::NS1) {
                    constexpr int tap = S / G::CG, cg = S % G::CG;
                    constexpr int off = (cg * 4 * G::XRP + tap * G::DIL) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(bb + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(bb2 + off + e * 64);
                } else if constexpr (S < G::NS1 + G::NSH) {
                    constexpr int off = ((S - G::NS1) * 4 * G::MRP) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(mb1 + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(mb2 + off + e * 64);
                } else {
                    constexpr int off = ((S - G::NS1 - G::NSH) * 4 * G::MRP) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(rb1 + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(rb2 + off + e * 64);
                }
            }
        };
        // stages [S0, S1) into the accumulators: operands one stage ahead of their MFMAs
        auto run = [&](auto S0C, auto S1C) {
            constexpr int S0 = decltype(S0C)::value, S1 = decltype(S1C)::value;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            fetch_a(IntC<S0>{}, abuf[S0 & 1]);
            fetch_b(IntC<S0>{}, bbuf[S0 & 1]);
            __builtin_amdgcn_sched_barrier(0);
            static_for<S0, S1>([&](auto SC) {
                constexpr int S = decltype(SC)::value, SN = S + 1;
                if constexpr (SN < G::NST) entry(IntC<SN>{});
                if constexpr (SN < S1) {
                    fetch_a(IntC<SN>{}, abuf[SN & 1]);
                    fetch_b(IntC<SN>{}, bbuf[SN & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][0], bbuf[S & 1][e][0], hi[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][0], bbuf[S & 1][e][1], lo[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[S & 1][h][1], bbuf[S & 1][e][0], lo[h][e], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        entry(IntC<0>{});                                // the window image is complete, stage 0 is in place
        run(IntC<0>{}, IntC<G::NS1>{});
        {
            // conv1 -> hidden image (columns beyond T feed only columns beyond T: no mask)
            float lowm = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + row0 + 16 * h);
                const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 2 * G::C + row0 + 16 * h);
                const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    f16x4 h1, h2;
                    split_mid4<false>(hi[h][f], lo[h][f], s01, s23, b01, b23, p.slope, true, h1, h2, lowm);
                    *reinterpret_cast<f16x4*>(mw + f * 256 + h * (2 * G::MRP * 16)) = h1;
                    *reinterpret_cast<f16x4*>(mw + f * 256 + h * (2 * G::MRP * 16) + G::MHALF) = h2;
                }
            }
            low_note(low, 1, lowm);
        }
        pair_barrier();                                  // the hidden tile is complete, nobody reads the window image any more
        convk_convert_centre<G>(raw, ximg, tid);         // (visible to every wave after the next stage entry's barrier)
        convh_load_raw<G>(raw, p.x + nb * ustride, p.T, ntile * G::NM - G::P, tid, more, p.reflect != 0);
        run(IntC<G::NS1>{}, IntC<G::NST>{});
        // ---- epilogue: outputs, then the image of the next window ----------------------------------------------
        pair_barrier();                                  // every wave is done with the hidden tile and the raw centre
        pair_wait_vm0();                                 // the next window, the next tile's first stages
        {
            const size_t boff = (size_t)b * ustride;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + boff, ubytes);
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.y_act ? p.y_act + boff : p.y, p.y_act ? ubytes : 0u);
            const float zero[4] = {0.f, 0.f, 0.f, 0.f};
            float off[2][2][4] = {};                     // the output offset of the bias-removal flows (the graph's last op only)
            if (p.sub) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.sub + (p.sub_batched ? boff : 0), ubytes);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const int t = t0 + col0 + f * 16;
                        const unsigned vo = t < p.T ? (unsigned)((row0 + 16 * h) * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                        for (int i = 0; i < 4; ++i) off[h][f][i] = buffer_load1s(rs, vo, (unsigned)i * t4);
                    }
                pair_wait_vm0();
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + G::C + row0 + 16 * h);
                const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 3 * G::C + row0 + 16 * h);
                const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    combine4(hi[h][f], lo[h][f], s01, s23, b01, b23, zero);
                    range_note4p(bad2, hi[h][f]);
                    const int t = t0 + col0 + f * 16;
                    const unsigned voff = t < p.T ? (unsigned)((row0 + 16 * h) * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = hi[h][f][i];
                        if (p.post == FV_POST_TANH) v = tanhf(v);
                        else if (p.post == FV_POST_RELU) v = fmaxf(v, 0.f);
                        const float a = (p.act_slope != 1.f ? act(v, p.act_slope) : v) - off[h][f][i];
                        buffer_store1s(ry, voff, (unsigned)i * t4, p.y_act ? v : a);
                        if (p.y_act) buffer_store1s(ra, voff, (unsigned)i * t4, a);
                    }
                }
            }
        }
        if (!more) break;
        convh_convert<G>(raw, ximg, p.slope, tid, low, 0);
        g0 = slot_of(G::NST);
        item = nitem;
        b = nb;
        tile = ntile;
    }
    pair_wait_vm0();
    if (p.guard) {
        const float bad = bad2.x + bad2.y;
        if (bad != bad) *p.guard = 1;
        // the block's verdict on the low side (pairh_kernels.hpp low_flag), through the scratch words behind the biases
        unsigned* const su = reinterpret_cast<unsigned*>(bl + 4 * G::C);
        if (lane == 0) su[wave] = low.bits;
        pair_barrier();
        if (tid == 0) {
            unsigned all = 0;
            for (int w = 0; w < 8; ++w) all |= su[w];
            if (((all & 2u) && !(all & 1u)) || ((all & 8u) && !(all & 4u))) *p.guard = 4;
        }
    }
}

// ---- 256 channels ---------------------------------------------------------------------------------------------------
// A block holds all 256 rows of a 32-COLUMN tile: 8 waves = 8 row slabs of 32, ONE column group -- the same 32 x 32 wave
// tile, window image <= 52 KB, hidden tile 32 KB.  With one column group no two waves share a row of weights, so there is
// nothing for an LDS ring to share: every wave loads the A operands of ITS 32 rows straight from L2 into registers
// (4 x 16 bytes per lane and K step, three K steps ahead) and the K loops have NO barrier -- four per tile instead of one
// per weight stage.  (First built with a ring of 16 KB half-K-step stages, 80 stage entries per tile: 1350 cycles per K
// step against 384 of MFMA issue; the packed image [K step][split half][row sixteenth][lane][8 halves] is that form's.)
// The MFMA order of every split kernel -- hi += a1 b1, lo += a1 b2, lo += a2 b1 -- and convs_kernel's K order (chunks of
// 128 input channels, tap-major inside a chunk; then the hidden tile's eight groups, then the raw centre's): the bits of
// convs_kernel + convr_kernel, the two-launch form at 256 channels.
// NM_: columns per tile -- 32 (50 tiles for MelGAN's first stage at batch 1: the latency-bound form), or 64 for runs with
// tiles to spare: a 32 x 64 wave tile (24 MFMAs per 4 A loads and 8 B reads), half the weight traffic per column -- at 32
// columns every tile pulls the stack's 1.3 MB of weights through L2, 20 TB/s over the chip at batch 64 -- and 150 KB of LDS
template <int DIL_, int NM_>
struct ConvK2Geom {
    static constexpr int DIL = DIL_, KT = 3, C = 256, CG = 8, CB = 32, NM = NM_, NFW = NM / 16, NT = 512;
    static constexpr int P = DIL;
    static constexpr int XROWS = (NM + 2 * DIL + 3) / 4 * 4;
    static constexpr int XRP = XROWS;                    // (B reads stay below row NM - 1 + 2 DIL: no padding rows needed)
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;
    static constexpr int MRP = NM, MHALF = CB * MRP * 16;
    static constexpr int NK1 = KT * CG, NK2 = 2 * CG, NK = NK1 + NK2;     // K steps of conv1 / of the 1x1 pair
    static constexpr int STEP_BYTES = 2 * 16384;         // packed: [K step][split half: 16 KB]
    static constexpr int QD = NM >= 64 ? 1 : 3;          // A operands this many K steps ahead of their MFMAs (queue of QD + 1 slots;
                                                         // a K step of the wide tile is 768 matrix cycles per SIMD: one ahead is enough)
    static constexpr int NA = 4;                         // loads per wave and K step
    static constexpr int NRAW = XR * 8;
    static constexpr int WBYTES = NK * STEP_BYTES;
    static constexpr int LDS_BYTES = 2 * XHALF + 2 * MHALF + (4 * C + 16) * 4;
    static_assert(2 * MHALF <= 2 * XHALF && LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * (NFW - 1)) * 16 + 16 < 65536, "ds_read immediate range");
    static_assert(NK % (QD + 1) == 0, "the A queue runs on from tile to tile: slot = K step % (QD + 1)");
};

template <int DIL, int NM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convk2_kernel(ConvKParams p) {
    typedef ConvK2Geom<DIL, NM> G;
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int item = equal_share(share, p.n_items, p.nblk);
    const int hi_item = equal_share(share + 1, p.n_items, p.nblk);
    if (item >= hi_item) return;

    char* const ximg = reinterpret_cast<char*>(smem);
    char* const mimg = ximg + 2 * G::XHALF;
    float* const bl = reinterpret_cast<float*>(mimg + 2 * G::MHALF);
    const int n = lane & 15, kb = lane >> 4;
    const int ws = wave;                                 // row slab of 32; one column group of NM
    const int col0 = n;
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;
    const char* const mptr = mimg + (kb * G::MRP + col0) * 16;
    const char* const rptr = ximg + (kb * G::MRP + col0) * 16;
    const int row0 = 32 * ws + 4 * kb;
    char* const mw = mimg + ((4 * ws + (kb >> 1)) * G::MRP + col0) * 16 + 8 * (kb & 1);

    const size_t ustride = (size_t)G::C * (size_t)p.T;
    const unsigned ubytes = (unsigned)G::C * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (unsigned)G::WBYTES);
    const unsigned aoff = (unsigned)((2 * ws) * 1024 + lane * 16);        // this wave's two row sixteenths inside a 16 KB half
    // A operands of K step KS (compile time, modulo the tile's NK: the queue runs on into the next tile): [h][split half]
    auto load_a = [&](auto KC, f16x8 (&dst)[2][2]) {
        constexpr int KS = decltype(KC)::value % G::NK;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                dst[h][e] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    rw, (int)(aoff + (unsigned)(h * 1024)), KS * G::STEP_BYTES + e * 16384, 0));
    };
    int b = item / p.n_tiles, tile = item - b * p.n_tiles;
    LowGuard low;
    f32x2 bad2 = {0.f, 0.f};
    ConvHRaw<G> raw;
    convh_load_raw<G>(raw, p.x + b * ustride, p.T, tile * G::NM - G::P, tid, true, p.reflect != 0);
    if (tid < G::C) {
        bl[tid] = p.b1 ? p.b1[tid] : 0.f;
        bl[G::C + tid] = p.b2 ? p.b2[tid] : 0.f;
        bl[2 * G::C + tid] = p.w[G::WBYTES / 4 + tid];
        bl[3 * G::C + tid] = p.w[G::WBYTES / 4 + G::C + tid];
    }
    f16x8 aq[G::QD + 1][2][2];                           // K step KS sits in aq[KS % (QD + 1)]
    static_for<0, G::QD>([&](auto QC) { load_a(QC, aq[decltype(QC)::value]); });
    pair_wait_vm0();
    convh_convert<G>(raw, ximg, p.slope, tid, low, 0);
    for (;;) {
        const int t0 = tile * G::NM;
        const int nitem = item + 1;
        const bool more = nitem < hi_item;
        int nb = b, ntile = tile + 1;
        if (ntile == p.n_tiles) {
            ntile = 0;
            ++nb;
        }
        f32x4 hi[2][G::NFW], lo[2][G::NFW];
        f16x8 bbuf[2][G::NFW][2];

        LdsCF* const bb = lds_opaque(reinterpret_cast<const float*>(bptr));
        LdsCF* const bb2 = lds_opaque(reinterpret_cast<const float*>(bptr + G::XHALF));
        LdsCF* const mb1 = lds_opaque(reinterpret_cast<const float*>(mptr));
        LdsCF* const mb2 = lds_opaque(reinterpret_cast<const float*>(mptr + G::MHALF));
        LdsCF* const rb1 = lds_opaque(reinterpret_cast<const float*>(rptr));
        LdsCF* const rb2 = lds_opaque(reinterpret_cast<const float*>(rptr + G::MHALF));
        auto fetch_b = [&](auto KC, f16x8 (&dst)[G::NFW][2]) {
            constexpr int KS = decltype(KC)::value;
#pragma unroll
            for (int e = 0; e < G::NFW; ++e) {
                if constexpr (KS < G::NK1) {
                    constexpr int chunk = KS / 12, tap = (KS % 12) / 4, cg = 4 * chunk + KS % 4;
                    constexpr int off = (cg * 4 * G::XRP + tap * G::DIL) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(bb + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(bb2 + off + e * 64);
                } else if constexpr (KS < G::NK1 + G::CG) {
                    constexpr int off = ((KS - G::NK1) * 4 * G::MRP) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(mb1 + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(mb2 + off + e * 64);
                } else {
                    constexpr int off = ((KS - G::NK1 - G::CG) * 4 * G::MRP) * 4;
                    dst[e][0] = *reinterpret_cast<LdsH8*>(rb1 + off + e * 64);
                    dst[e][1] = *reinterpret_cast<LdsH8*>(rb2 + off + e * 64);
                }
            }
        };
        // K steps [K0, K1): the A queue runs QD steps ahead (loads return in order: K step KS has landed once at most the
        // loads issued after it are outstanding -- QD steps' worth, plus the next tile's raw window where that was requested
        // in between: RAWK = first K step issued after it); B operands one step ahead, from the images
        auto run = [&](auto K0C, auto K1C) {
            constexpr int K0 = decltype(K0C)::value, K1 = decltype(K1C)::value;
            // the next tile's window is requested QD steps before the tile ends: no A operand of THIS tile is issued after it,
            // so no count here waits for it (requested at the start of the second GEMM it had one to three K steps to land)
            constexpr int RAWK = G::NK - G::QD;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            fetch_b(IntC<K0>{}, bbuf[K0 & 1]);
            static_for<K0, K1>([&](auto KC) {
                constexpr int KS = decltype(KC)::value;
                if constexpr (KS == RAWK)
                    convh_load_raw<G>(raw, p.x + nb * ustride, p.T, ntile * G::NM - G::P, tid, more, p.reflect != 0);
                load_a(IntC<KS + G::QD>{}, aq[(KS + G::QD) % (G::QD + 1)]);
                if constexpr (KS + 1 < K1) fetch_b(IntC<KS + 1>{}, bbuf[(KS + 1) & 1]);
                // outstanding after K step KS's loads: steps KS + 1 .. KS + QD, and the window if it was requested after them
                constexpr bool raw_after = RAWK > KS - G::QD && RAWK <= KS;
                // (a tile's first QD steps were waited for in the epilogue of the tile before, ahead of its stores: a count
                // here would wait for those stores)
                if constexpr (KS >= G::QD) wait_vm<G::NA * G::QD + (raw_after ? G::NRAW : 0)>();
                __builtin_amdgcn_sched_barrier(0);
                f16x8 (&a)[2][2] = aq[KS % (G::QD + 1)];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e)
                        hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bbuf[KS & 1][e][0], hi[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bbuf[KS & 1][e][1], lo[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][1], bbuf[KS & 1][e][0], lo[h][e], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        pair_barrier();                                  // the window image is complete
        run(IntC<0>{}, IntC<G::NK1>{});
        {
            float lowm = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + row0 + 16 * h);
                const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 2 * G::C + row0 + 16 * h);
                const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    f16x4 h1, h2;
                    split_mid4<false>(hi[h][f], lo[h][f], s01, s23, b01, b23, p.slope, true, h1, h2, lowm);
                    *reinterpret_cast<f16x4*>(mw + f * 256 + h * (2 * G::MRP * 16)) = h1;
                    *reinterpret_cast<f16x4*>(mw + f * 256 + h * (2 * G::MRP * 16) + G::MHALF) = h2;
                }
            }
            low_note(low, 1, lowm);
        }
        pair_barrier();                                  // the hidden tile is complete, nobody reads the window image any more
        convk_convert_centre<G>(raw, ximg, tid);
        pair_barrier();                                  // the raw centre's image is complete
        run(IntC<G::NK1>{}, IntC<G::NK>{});
        pair_barrier();                                  // every wave is done with the hidden tile and the raw centre
        {
            const size_t boff = (size_t)b * ustride;
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + boff, ubytes);
            const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.y_act ? p.y_act + boff : p.y, p.y_act ? ubytes : 0u);
            const float zero[4] = {0.f, 0.f, 0.f, 0.f};
            float off[2][G::NFW][4] = {};                // the output offset of the bias-removal flows (the graph's last op only)
            if (p.sub) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.sub + (p.sub_batched ? boff : 0), ubytes);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int f = 0; f < G::NFW; ++f) {
                        const int t = t0 + col0 + f * 16;
                        const unsigned vo = t < p.T ? (unsigned)((row0 + 16 * h) * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                        for (int i = 0; i < 4; ++i) off[h][f][i] = buffer_load1s(rs, vo, (unsigned)i * t4);
                    }
                pair_wait_vm0();
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + G::C + row0 + 16 * h);
                const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 3 * G::C + row0 + 16 * h);
                const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) {
                    combine4(hi[h][f], lo[h][f], s01, s23, b01, b23, zero);
                    range_note4p(bad2, hi[h][f]);
                    const int t = t0 + col0 + f * 16;
                    const unsigned voff = t < p.T ? (unsigned)((row0 + 16 * h) * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = hi[h][f][i];
                        if (p.post == FV_POST_TANH) v = tanhf(v);
                        else if (p.post == FV_POST_RELU) v = fmaxf(v, 0.f);
                        const float a = (p.act_slope != 1.f ? act(v, p.act_slope) : v) - off[h][f][i];
                        buffer_store1s(ry, voff, (unsigned)i * t4, p.y_act ? v : a);
                        if (p.y_act) buffer_store1s(ra, voff, (unsigned)i * t4, a);
                    }
                }
            }
        }
        if (!more) break;
        // the window (requested before conv2) and the A operands of the next tile's first steps are all older than this
        // tile's stores (8 NFW per lane, twice that with the activated twin): wait for the loads only
        if (p.y_act) wait_vm<16 * G::NFW>();
        else wait_vm<8 * G::NFW>();
        convh_convert<G>(raw, ximg, p.slope, tid, low, 0);
        item = nitem;
        b = nb;
        tile = ntile;
    }
    pair_wait_vm0();
    if (p.guard) {
        const float bad = bad2.x + bad2.y;
        if (bad != bad) *p.guard = 1;
        unsigned* const su = reinterpret_cast<unsigned*>(bl + 4 * G::C);
        if (lane == 0) su[wave] = low.bits;
        pair_barrier();
        if (tid == 0) {
            unsigned all = 0;
            for (int w = 0; w < 8; ++w) all |= su[w];
            if (((all & 2u) && !(all & 1u)) || ((all & 8u) && !(all & 4u))) *p.guard = 4;
        }
    }
}

}  // namespace fv
