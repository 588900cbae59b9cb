// MelGAN ResidualStack as ONE launch, 32 ... 256 channels, split-f16 operands:
//
//     y = W2 lrelu( W1 (*) lrelu( pad(x) ) + b1 ) + Ws x + (b2 + bs)        (reference model/generator/modules.py:351-382:
//                                                                             stack = [act, pad, conv k=3 dilated, act, conv 1x1],
//                                                                             skip_layer = conv 1x1 of the raw input)
//
// The two-launch form (convh_kernel / convs_kernel for the dilated conv, convg_kernel / convr_kernel for the K-concatenated
// 1x1 pair; 64 channels and fewer: the fp32 kernels) writes the hidden tensor to HBM and reads it and x back; at MelGAN's
// sizes (T = 200 frames, batch 1: 1 600 ... 51 200 columns) a launch is a single tile per block and its fixed costs, twice.
// Here a block owns ALL C rows of a column tile and walks ONE sequence of 5 C / 32 K steps:
//
//     K steps [0, 3 CG)        conv1: taps x 32-channel groups, B operands from the activated window's split image
//     -- accumulators (+ b1, lrelu) -> split image of the hidden tile in LDS; the RAW window (still in the registers
//        it was prefetched into) -> split image of the tile's centre columns, over the activated image, which is dead
//     K steps [3 CG, 4 CG)     W2 x hidden tile
//     K steps [4 CG, 5 CG)     Ws x raw centre                          (the K order of convg_kernel: hidden, then x)
//
// Same K order, same splits, same epilogue arithmetic as the two-launch form at 128 and 256 channels: identical bits
// (tests/test_gpu_stack.py).  Weights come from L2 straight into registers (no LDS ring: convk2_kernel below).
#pragma once
#include "convh_kernels.hpp"

namespace fv {

struct ConvKParams {
    const float* x;      // [B, C, T]
    const float* w;      // packed (pack_convk_kernel / pack_convk2_kernel): the K steps' A operands, then the rows' inverse
                         // prescales of conv1 [C] and of [W2 | Ws] [C]
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

// ---- the kernel --------------------------------------------------------------------------------------------------------
// 256 channels: a block holds all 256 rows of a 32-COLUMN tile: 8 waves = 8 row slabs of 32, ONE column group -- the same 32 x 32 wave
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
// C_ <= 128: the same kernel with 16-row slabs (convq2_kernels.hpp's wave layout): 8 waves = C / 16 row slabs x 128 / C column
// groups of 64 columns (64-column tiles at 128 channels, 128 at 64, 256 at 32: the hidden tile is 32 KB whatever C); two / four
// waves then load the same (small) weights.  Packed images: [K step][row sixteenth][split half] (pack_convk_kernel), K order
// tap-major.  (Its first form streamed the weights through an LDS ring of three K-step stages, one barrier per stage: equal
// within 1-2 % at every batch size -- MelGAN batch 1 / 8 / 64: 0.254 vs 0.256, 0.797 vs 0.813, 5.85 vs 5.79 ms -- and gone.)  At 32 and 64 channels the queue holds the whole stack's A operands (5 / 10 K steps of 8 registers):
// they are loaded once per tile and would not have to be -- L2 hits either way.
#ifndef FV_K2_ILV
#define FV_K2_ILV 1
#endif
template <int C_, int DIL_, int NM_>
struct ConvK2Geom {
    static constexpr int DIL = DIL_, KT = 3, C = C_, CG = C / 32, CB = C / 8, NM = NM_, NT = 512;
    static constexpr int NH = C == 256 ? 2 : 1;          // row sixteenths per wave
    static constexpr int NSLAB = C / (16 * NH), WN = 8 / NSLAB;         // row slabs x column groups = 8 waves
    static constexpr int NMW = NM / WN, NFW = NMW / 16;  // columns / fragments per wave
    static constexpr int P = DIL;
    static constexpr int XROWS = (NM + 2 * DIL + 3) / 4 * 4;
    static constexpr int XRP = XROWS;                    // (B reads stay below row NM - 1 + 2 DIL: no padding rows needed)
    static constexpr int XHALF = CB * XRP * 16;
    static constexpr int XR = (XROWS * CB + NT - 1) / NT;
    static constexpr int MRP = NM, MHALF = CB * MRP * 16;
    static constexpr int NK1 = KT * CG, NK2 = 2 * CG, NK = NK1 + NK2;     // K steps of conv1 / of the 1x1 pair
    static constexpr int STEP_BYTES = C * 128;           // packed bytes of a K step: all rows, both split halves
    // byte offset of (row sixteenth r16, split half e) inside a K step: 256 channels [half][sixteenth], else [sixteenth][half]
    static constexpr int A_R16 = C == 256 ? 1024 : 2048, A_HALF = C == 256 ? 16384 : 1024;
    // A operands this many K steps ahead of their MFMAs (queue of QD + 1 slots, which must divide NK: the queue runs on from
    // tile to tile).  A K step of the wide 256-channel tile is 768 matrix cycles per SIMD: one ahead is enough
    static constexpr int QD = C == 256 ? (NM >= 64 ? 1 : 3) : C == 128 ? 3 : 4;
    static constexpr bool RES = QD + 1 == NK;            // the queue holds every K step (32 channels): loaded once per block
    static constexpr int NA = 2 * NH;                    // loads per wave and K step
    static constexpr int NRAW = XR * 8;
    static constexpr int WBYTES = NK * STEP_BYTES;
    static constexpr int LDS_BYTES = 2 * XHALF + 2 * MHALF + (4 * C + 16) * 4;
    static_assert(NSLAB * WN == 8 && NM == NMW * WN && NFW * 16 == NMW, "8 waves");
    static_assert(2 * MHALF <= 2 * XHALF && LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(((CG - 1) * 4 * XRP + (KT - 1) * DIL + 16 * (NFW - 1)) * 16 + 16 < 65536, "ds_read immediate range");
    static_assert(NK % (QD + 1) == 0, "the A queue runs on from tile to tile: slot = K step % (QD + 1)");
};

template <int C, int DIL, int NM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convk2_kernel(ConvKParams p) {
    typedef ConvK2Geom<C, DIL, NM> G;
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
    const int ws = wave % G::NSLAB;                      // row slab of 16 NH rows, column group of NMW columns
    const int col0 = (wave / G::NSLAB) * G::NMW + n;
    const char* const bptr = ximg + (kb * G::XRP + col0) * 16;
    const char* const mptr = mimg + (kb * G::MRP + col0) * 16;
    const char* const rptr = ximg + (kb * G::MRP + col0) * 16;
    const int row0 = 16 * G::NH * ws + 4 * kb;           // + 16 h + i
    char* const mw = mimg + ((2 * G::NH * ws + (kb >> 1)) * G::MRP + col0) * 16 + 8 * (kb & 1);

    const size_t ustride = (size_t)G::C * (size_t)p.T;
    const unsigned ubytes = (unsigned)G::C * (unsigned)p.T * 4u;
    const unsigned t4 = (unsigned)p.T * 4u;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (unsigned)G::WBYTES);
    const unsigned aoff = (unsigned)((G::NH * ws) * G::A_R16 + lane * 16);     // this wave's row sixteenths inside a K step
    // A operands of K step KS (compile time, modulo the tile's NK: the queue runs on into the next tile): [h][split half]
    auto load_a = [&](auto KC, f16x8 (&dst)[G::NH][2]) {
        constexpr int KS = decltype(KC)::value % G::NK;
#pragma unroll
        for (int h = 0; h < G::NH; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                dst[h][e] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    rw, (int)(aoff + (unsigned)(h * G::A_R16)), KS * G::STEP_BYTES + e * G::A_HALF, 0));
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
    f16x8 aq[G::QD + 1][G::NH][2];                       // K step KS sits in aq[KS % (QD + 1)]
    static_for<0, G::RES ? G::QD + 1 : G::QD>([&](auto QC) { load_a(QC, aq[decltype(QC)::value]); });
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
        f32x4 hi[G::NH][G::NFW], lo[G::NH][G::NFW];
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
                    // (256 channels: chunks of 128 input channels, tap-major inside a chunk; else tap-major over all groups)
                    constexpr int chunk = G::C == 256 ? KS / 12 : 0, tap = G::C == 256 ? (KS % 12) / 4 : KS / G::CG;
                    constexpr int cg = G::C == 256 ? 4 * chunk + KS % 4 : KS % G::CG;
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
            constexpr int RAWK = G::RES ? G::NK1 : G::NK - G::QD;
            constexpr bool ILV = FV_K2_ILV && G::NH == 1;
            static_assert(RAWK >= G::NK1, "the window registers are read after conv1 (the raw centre)");
#pragma unroll
            for (int h = 0; h < G::NH; ++h)
#pragma unroll
                for (int f = 0; f < G::NFW; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            fetch_b(IntC<K0>{}, bbuf[K0 & 1]);
            static_for<K0, K1>([&](auto KC) {
                constexpr int KS = decltype(KC)::value;
                if constexpr (KS == RAWK)
                    convh_load_raw<G>(raw, p.x + nb * ustride, p.T, ntile * G::NM - G::P, tid, more, p.reflect != 0);
                // outstanding after K step KS's loads: steps KS + 1 .. KS + QD, and the window if it was requested after them
                constexpr bool raw_after = RAWK > KS - G::QD && RAWK <= KS;
                // (a tile's first QD steps were waited for in the epilogue of the tile before, ahead of its stores: a count
                // here would wait for those stores)
                if constexpr (ILV) {                     // the loads of this step go one per MFMA gap (below): wait first
                    if constexpr (!G::RES && KS >= G::QD) wait_vm<G::NA * (G::QD - 1) + (raw_after ? G::NRAW : 0)>();
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!G::RES) load_a(IntC<KS + G::QD>{}, aq[(KS + G::QD) % (G::QD + 1)]);
                    if constexpr (KS + 1 < K1) fetch_b(IntC<KS + 1>{}, bbuf[(KS + 1) & 1]);
                } else {
                    if constexpr (!G::RES) load_a(IntC<KS + G::QD>{}, aq[(KS + G::QD) % (G::QD + 1)]);
                    if constexpr (KS + 1 < K1) fetch_b(IntC<KS + 1>{}, bbuf[(KS + 1) & 1]);
                    if constexpr (!G::RES && KS >= G::QD) wait_vm<G::NA * G::QD + (raw_after ? G::NRAW : 0)>();
                    __builtin_amdgcn_sched_barrier(0);
                }
                f16x8 (&a)[G::NH][2] = aq[KS % (G::QD + 1)];
#pragma unroll
                for (int h = 0; h < G::NH; ++h)
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e)
                        hi[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bbuf[KS & 1][e][0], hi[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < G::NH; ++h)
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bbuf[KS & 1][e][1], lo[h][e], 0, 0, 0);
#pragma unroll
                for (int h = 0; h < G::NH; ++h)
#pragma unroll
                    for (int e = 0; e < G::NFW; ++e)
                        lo[h][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][1], bbuf[KS & 1][e][0], lo[h][e], 0, 0, 0);
                if constexpr (ILV) {                     // (convq2_kernels.hpp: one load per MFMA gap on the 16 x 64 wave tiles)
                    constexpr int NV = G::RES ? 0 : G::NA, ND = KS + 1 < K1 ? 2 * G::NFW : 0;
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
#pragma unroll
                    for (int i = 0; i < ND; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * G::NH * G::NFW - NV - ND, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        pair_barrier();                                  // the window image is complete
        run(IntC<0>{}, IntC<G::NK1>{});
        {
            float lowm = 0.f;
#pragma unroll
            for (int h = 0; h < G::NH; ++h) {
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
            float off[G::NH][G::NFW][4] = {};                // the output offset of the bias-removal flows (the graph's last op only)
            if (p.sub) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.sub + (p.sub_batched ? boff : 0), ubytes);
#pragma unroll
                for (int h = 0; h < G::NH; ++h)
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
            for (int h = 0; h < G::NH; ++h) {
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
        // tile's stores (4 NH NFW per lane, twice that with the activated twin): wait for the loads only
        if (p.y_act) wait_vm<8 * G::NH * G::NFW>();
        else wait_vm<4 * G::NH * G::NFW>();
        convh_convert<G>(raw, ximg, p.slope, tid, low, 0);
        item = nitem;
        b = nb;
        tile = ntile;
    }
    pair_wait_vm0();
    if (p.guard) {
        const float bad = bad2.x + bad2.y;
        if (bad != bad) guard_raise_high(p.guard);
        unsigned* const su = reinterpret_cast<unsigned*>(bl + 4 * G::C);
        if (lane == 0) su[wave] = low.bits;
        pair_barrier();
        if (tid == 0) {
            unsigned all = 0;
            for (int w = 0; w < 8; ++w) all |= su[w];
            if (((all & 2u) && !(all & 1u)) || ((all & 8u) && !(all & 4u))) guard_raise_low(p.guard);
        }
    }
}

}  // namespace fv
