// Fused ResBlock1 pair with SPLIT-F16 operands on the gfx950 matrix cores:
//
//     x' = x + conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2          (reference model/generator/modules.py:223-230)
//
// Same operator, tiling and persistent-block structure as pair_kernels.hpp; what changes is the arithmetic of
// the two convolutions.  Every fp32 operand v (activation or weight) is written as
//         v  =  h1 + h2 / 2048 + e,     h1 = f16(v),  h2 = f16((v - h1) * 2048),  |e| <= 2^-22 |v|
// (v - h1 is exact in fp32; the 2^11 scale keeps h2 a normal f16 number whenever h1 is one), and a product
// becomes three v_mfma_f32_16x16x32_f16 terms accumulated in fp32:
//         a * b  ~  a1 b1  +  (a1 b2 + a2 b1) / 2048                    (dropped: a2 b2 / 2^22 and the e terms)
// f16 x f16 products are exact in the fp32 accumulator, so the only new error is the dropped 2^-22 terms: per
// layer the result is as close to the exact sum as the fp32 FMA chain is (tests/test_split_precision.py,
// DESIGN.md section 3.7).  Three f16 MFMAs cover 32 K values in 3 x 16 cycles where the fp32 MFMA needs
// 8 x 32: at 16-32 channels the fp32 pair kernel is matrix-core bound; this one is not -- what is left is the HBM
// traffic of x and x' (where SURVEY section 8(d) puts these layers), LDS operand traffic and, at batch 1, the
// per-tile conversion / epilogue work and latency (profiles/r02_*: 0.12-0.22 MFMA-busy).
//
// Layout: the activated, split input lives in LDS as [split half][block of 8 channels][time row][8 halves]:
// the B operand of a K step (lane = column n, K block = 8 consecutive channels of one tap) is ONE ds_read_b128
// per half, and consecutive columns sit in consecutive 16-byte slots, which is conflict-free for the lane
// groups ds_read_b128 is serviced in (PairHGeom).  The D fragment (lane = column, 4 consecutive channels) is
// half a block entry: conv1's epilogue splits and stores the intermediate with ds_write_b64.
// K order: step s covers 32 / C taps (C = 16: taps 2s, 2s+1 x 16 channels; an odd tap count is padded with a
// zero tap), channels inside a tap.  The packed weights of a member's two convs sit in LDS and stream through a
// two-step register queue like the B operands (AREG: with four fragments per wave at C = 16 a phase's A operands are
// loaded up front instead, 48 registers at 11 taps -- the 2-waves-per-SIMD shape).
// Occupancy is what these kernels live on at batch 1 (a tile is a chain of short phases: loads, conv1, barrier, conv2,
// epilogue, convert, barrier; there is no matrix work to hide the latencies behind): 116-119 VGPRs, 4 waves per SIMD.
// C = 32: ONE block of 15 waves per CU (240-column tiles; the two 11-tap weight images are 88 KB, a 16th wave's columns
// would need 162 KB of LDS): 49 us per three-member launch at T = 120 000 against 59 us with 8 waves (128-column
// tiles) and 53 with 12.  C = 16: two 8-wave blocks per CU, two fragments per wave (256-column tiles).
//
// Per tile: the NEXT tile's raw fp32 x window is loaded global -> registers at the top of the tile (in flight for
// the whole tile, no LDS landing buffer); conv1 -> + b1, lrelu, zero outside [0, T), split -> intermediate image;
// barrier; conv2 -> + b2 + residual (fp32, from global memory); convert pass for the next tile (lrelu, split,
// transpose registers -> x image); stores -> HBM; barrier.  Two barriers per tile.
// Limits: |activation| and |weight| < 65520 (f16 range); beyond that the range guard (range_note / range_flag below) fires
// and the host repeats the call on the fp32 kernels.
#pragma once
#include "pair_kernels.hpp"

namespace fv {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr float kSplitScale = 2048.f, kSplitInv = 1.f / 2048.f;

// The split runs once per operand element per tile and its VALU instructions cost matrix time, so:
// leaky ReLU as mul + max without the canonicalising v_max hipcc puts in front of fmaxf, and the scaled remainder
// (v - h1) * 2048 as ONE fused multiply-add on the f16 value (v_fma_mix_f32: the f16 -> f32 conversion is part of
// the instruction).  Both are exact rewrites: v - h1 and the power-of-two scalings are exact in fp32.
__device__ __forceinline__ float split_act(float v, float slope) {
    const float t = v * slope;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(t));
    return r;
}
__device__ __forceinline__ _Float16 split_rem(float v, _Float16 h1) {
    return (_Float16)fmaf((float)h1, -kSplitScale, v * kSplitScale);
}

// ---- the same arithmetic on PAIRS of elements, with the instructions spelled out (round 4) ---------------------------
// Per element the split is lrelu (v_pk_mul_f32 for two elements + v_max_f32), h1 = f16(a) (v_cvt_pk_f16_f32 for two),
// a * 2048 (v_pk_mul_f32 for two) and h2 = f16(fma(h1, -2048, a * 2048)) -- v_fma_mixlo_f16 / v_fma_mixhi_f16: the f16
// operand is read straight from its half of the packed register, the fp32 result is rounded to f16 into one half of the
// destination: conversion and packing are part of the instruction.  3.5 VALU instructions per element where hipcc's own
// selection of the scalar code above took 5.6 (it converted h1 back to fp32 for three of four pairs).  Bit for bit the
// same values: (a - h1) * 2048 is exact in fp32 either way.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 split_act2(f32x2 v, float slope) {
    const f32x2 t = v * f32x2{slope, slope};
    f32x2 r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r.x) : "v"(v.x), "v"(t.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(r.y) : "v"(v.y), "v"(t.y));
    return r;
}
// a (activated, fp32) -> h1 = f16(a), h2 = f16((a - h1) * 2048), two elements
__device__ __forceinline__ void split2(f32x2 a, f16x2& h1, f16x2& h2) {
    h1 = __builtin_convertvector(a, f16x2);
    const f32x2 a2 = a * f32x2{kSplitScale, kSplitScale};
    const float neg = -kSplitScale;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(h2) : "v"(h1), "s"(neg), "v"(a2.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h2) : "v"(h1), "s"(neg), "v"(a2.y));
}
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
struct F16x8Parts {
    f16x2 p[4];
};

// Range guard of the split-f16 kernels.  An operand beyond the f16 range (|v| >= 65520: f16(v) = inf) turns every output
// it feeds into inf or NaN -- where the fp32 reference stays finite.  Every split kernel therefore folds its final values
// into one register per thread (v * 0 is NaN exactly when v is inf or NaN: one VALU instruction per stored element) and a
// thread that saw a non-finite value raises the launch's guard word (PairParams::guard, fv_plan_set_guard): the host then
// repeats the call on the exact-fp32 kernels (fv_plan_check_range, engine.py NativeModule._guarded).
// (only values that are actually stored count: the columns a tile computes beyond its valid outputs read LDS rows that
// hold whatever the previous phase left there -- finite or not, they are discarded)
__device__ __forceinline__ void range_note4(float& bad, float v0, float v1, float v2, float v3, bool stored) {
    float t = fmaf(v0, 0.f, bad);
    t = fmaf(v1, 0.f, t);
    t = fmaf(v2, 0.f, t);
    t = fmaf(v3, 0.f, t);
    bad = stored ? t : bad;
}
// The guard word's two sides are separate BYTES (FV_GUARD_HIGH = byte 0, FV_GUARD_LOW = byte 1): any block of any launch may
// raise either with a plain store, and a store to one byte cannot take back what another block wrote to the other -- with
// one 32-bit value per side the last writer won, and "overflow, then a quiet block" read as "low" (ADVICE r5).
__device__ __forceinline__ void guard_raise_high(int* g) { reinterpret_cast<volatile unsigned char*>(g)[0] = 1; }
__device__ __forceinline__ void guard_raise_low(int* g) { reinterpret_cast<volatile unsigned char*>(g)[1] = 1; }
__device__ __forceinline__ void range_flag(const PairCore& p, float bad) {
    if (p.guard && bad != bad) guard_raise_high(p.guard);
}

// The LOW side of the domain.  h1 = f16(v) is a normal f16 number -- and the pair keeps 22 bits of v -- only for
// |v| >= 2^-14; below that the pair keeps an ABSOLUTE 2^-36.  That is harmless for the small elements of a tensor whose
// scale is ordinary (2^-36 is 2^-26 of a 2^-10 maximum: below the fp32 rounding of its large elements), and it is a loss
// for a tensor that is small as a whole -- whose consumer's weights are correspondingly large.  Weights are taken care of
// at pack time (a power-of-two prescale per row: api.hip row_scale_kernel); for activations every split kernel watches
// the magnitudes of the operands it actually splits (the window conversion, the intermediate of a fused pair: ~0.6 VALU
// instructions per element, v_max3_f32 with |.| modifiers and two compares per group), and a block one of whose operand
// tensors was not all zero and all below kSplitLow raises the guard word (FV_GUARD_LOW): the host repeats the call on the
// exact-fp32 kernels, as it does for the high side.  The test is per BLOCK (every channel of hundreds of samples), never per element: silence inside an
// ordinary signal stays on the split kernels, and a false alarm costs time, not accuracy.
constexpr float kSplitLow = 0x1p-10f;
// Per wave, in ONE scalar register (no VGPR is spent on the guard): has any lane split an operand of
// magnitude >= kSplitLow (big), any non-zero operand (nz) -- separately for the two operand tensors a kernel splits
// ([0] the input window; [1] the intermediate of a fused pair / the second source of the two-source 1x1 conv): a small
// input in front of an ordinary intermediate is still a small input.
struct LowGuard {
    unsigned bits = 0;      // 1: [0] big, 2: [0] nz, 4: [1] big, 8: [1] nz -- one scalar register
};
__device__ __forceinline__ float low_max3(float a, float b, float c) {
    float t;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t) : "v"(a), "v"(b), "v"(c));
    return t;
}
// t: the largest magnitude of a group of operands this lane has just split
__device__ __forceinline__ void low_note(LowGuard& g, int which, float t) {
    const unsigned b = (__builtin_amdgcn_ballot_w64(t >= kSplitLow) != 0 ? 1u : 0u) | (__builtin_amdgcn_ballot_w64(t > 0.f) != 0 ? 2u : 0u);
    g.bits |= b << (2 * which);
}
__device__ __forceinline__ float low_max8(const float (&v)[8]) {
    float t = low_max3(v[0], v[1], v[2]);
    t = low_max3(t, v[3], v[4]);
    t = low_max3(t, v[5], v[6]);
    return low_max3(t, v[7], v[7]);
}
// end of a member's run: the block's verdict through `scratch` (>= 16 words of LDS no one else uses), one flag per block.
// Every wave of the block calls it (two barriers).
__device__ __forceinline__ void low_flag(const PairCore& p, const LowGuard& g, float* scratch, int wave, int lane, int nwaves) {
    if (!p.guard) return;
    const unsigned bits = g.bits;
    unsigned* const su = reinterpret_cast<unsigned*>(scratch);
    if (lane == 0) su[wave] = bits;
    pair_barrier();
    if (wave == 0 && lane == 0) {
        unsigned all = 0;
        for (int w = 0; w < nwaves; ++w) all |= su[w];
        if (((all & 2u) && !(all & 1u)) || ((all & 8u) && !(all & 4u))) guard_raise_low(p.guard);
    }
    pair_barrier();                                      // scratch may be written again (the block's next member)
}

// ---- epilogue pieces of the fused pairs, on pairs of rows (v_pk_fma_f32 / v_pk_mul_f32) --------------------------------
// A D fragment holds four consecutive rows of one column; s / b: the rows' inverse weight prescales and biases as two
// register pairs each (rows 0-1, 2-3).
// conv1 -> the intermediate's image entry: m = lrelu((hi + lo / 2048) s + b, slope), zero outside the sequence (MASK: only
// tiles that touch a sequence end pay for the select), split into h1 / h2; lowm: the lane's largest magnitude (LowGuard)
template <bool MASK>
__device__ __forceinline__ void split_mid4(const f32x4& hi, const f32x4& lo, f32x2 s01, f32x2 s23, f32x2 b01, f32x2 b23,
                                           float slope, bool ok, f16x4& h1, f16x4& h2, float& lowm) {
    const f32x2 c = {kSplitInv, kSplitInv};
    f32x2 a01 = split_act2(fma2(fma2(f32x2{lo[0], lo[1]}, c, f32x2{hi[0], hi[1]}), s01, b01), slope);
    f32x2 a23 = split_act2(fma2(fma2(f32x2{lo[2], lo[3]}, c, f32x2{hi[2], hi[3]}), s23, b23), slope);
    if constexpr (MASK) {
        a01.x = ok ? a01.x : 0.f;
        a01.y = ok ? a01.y : 0.f;
        a23.x = ok ? a23.x : 0.f;
        a23.y = ok ? a23.y : 0.f;
    }
    lowm = low_max3(lowm, a01.x, a01.y);
    lowm = low_max3(lowm, a23.x, a23.y);
    f16x2 p0, q0, p1, q1;
    split2(a01, p0, q0);
    split2(a23, p1, q1);
    h1 = f16x4{p0.x, p0.y, p1.x, p1.y};
    h2 = f16x4{q0.x, q0.y, q1.x, q1.y};
}
// conv2: (hi + lo / 2048) s + b + res -> hi (res: four floats, rows 0 .. 3)
__device__ __forceinline__ void combine4(f32x4& hi, const f32x4& lo, f32x2 s01, f32x2 s23, f32x2 b01, f32x2 b23,
                                         const float (&res)[4]) {
    const f32x2 c = {kSplitInv, kSplitInv};
    const f32x2 v01 = fma2(fma2(f32x2{lo[0], lo[1]}, c, f32x2{hi[0], hi[1]}), s01, b01) + f32x2{res[0], res[1]};
    const f32x2 v23 = fma2(fma2(f32x2{lo[2], lo[3]}, c, f32x2{hi[2], hi[3]}), s23, b23) + f32x2{res[2], res[3]};
    hi = f32x4{v01.x, v01.y, v23.x, v23.y};
}
// the range guard on a fragment, two rows per instruction (bad2: NaN in either half once a value was not finite)
__device__ __forceinline__ void range_note4p(f32x2& bad2, const f32x4& v) {
    const f32x2 z = {0.f, 0.f};
    bad2 = fma2(f32x2{v[0], v[1]}, z, bad2);
    bad2 = fma2(f32x2{v[2], v[3]}, z, bad2);
}

template <int MH_, int NF_, int NG_, int KT_, int DIL_>
struct PairHGeom {
    static constexpr int MH = MH_, NF = NF_, NG = NG_, KT = KT_, DIL = DIL_;
    static constexpr int C = 16 * MH;
    static constexpr int NW = NG, NT = 64 * NW;
    static constexpr int NM = 16 * NF * NG;             // intermediate columns per tile
    static constexpr int TPS = 32 / C;                  // taps per K step
    static constexpr int KS = (KT + TPS - 1) / TPS;     // K steps per conv
    static constexpr int KTP = KS * TPS;                // taps incl. the zero tap that pads an odd count
    static constexpr int P1 = (KT - 1) * DIL / 2, P2 = (KT - 1) / 2;
    static constexpr int AOFF = (4 - (P1 + P2) % 4) % 4;
    static constexpr int XWIN = NM + (KTP - 1) * DIL + AOFF;   // columns of x a tile reads (pad tap included)
    static constexpr int NCOL4 = (XWIN + 3) / 4;
    static constexpr int XROWS = 4 * NCOL4;             // rows of the x image
    static constexpr int WB = KS * MH * 2 * 1024;       // bytes of one conv's packed weights
    static constexpr bool AREG = MH == 1 && NF >= 4;    // a phase's A operands fit in registers (next to four fragments' accumulators: 2 waves per SIMD)
    static constexpr int MROWS = NM + 16;               // rows of the intermediate image (>= NM + KTP - 1)
    // image layout [split half][8-channel block][row][8 halves]: a block is RP rows of 16 bytes, RP a multiple of
    // 16, so the sixteen lanes of a ds_read_b128 lane group (all sixteen columns, two channel blocks) fall on the
    // sixteen 16-byte slots of the 256-byte bank row whatever the tap offset (no conflicts; a row-major
    // [row][channels] image with a padded stride measured 37 % conflict cycles)
    static constexpr int XRP = (XROWS + 15) / 16 * 16, MRP = MROWS;
    static constexpr int XHALF = (C / 8) * XRP * 16, MHALF = (C / 8) * MRP * 16;   // bytes of one split half
    static constexpr int NOUT = (NM - (KT - 1)) / 4 * 4;
    static_assert(C == 16 || C == 32, "16 or 32 channels");
    static_assert(KT % 2 == 1 && KTP - 1 <= 16, "odd tap counts up to 15");
    static_assert(MROWS % 16 == 0, "whole bank rows");
    static_assert(XHALF + ((KS - 1) * TPS * DIL + 16 * (NF - 1)) * 16 + 16 < 65536, "ds_read immediate range");
    static_assert(AREG || KS * MH * 2 * 1024 < 65536, "ds_read immediate range (weights)");
};

// ---- raw x window: global -> registers (a whole tile ahead) -> image [XROWS][h1 | h2], activated -------------
// task = (row t of the window, block of 8 channels), lanes along t (coalesced rows).  The tasks are FULL whole rounds of
// the block's NT threads plus REM left-over tasks.  Round 3 gave the left-over tasks to threads 0 .. REM - 1: at C = 16
// (632 tasks, 512 threads) waves 0 and 1 converted twice what the other six did and the whole block waited for them at the
// tile's last barrier [measured, tools/pairh_trace.py: convert 2000 against 1000-1200 cycles].  Now a left-over task is cut
// into Q parts of 8 / Q channels (Q = 4 when 4 REM threads exist, else 2, else 1) that go to Q x REM threads: the longest
// conversion of a tile is 1 + 1/Q rounds instead of 2.
template <class G>
struct PairHRaw {
    static constexpr int CB = G::C / 8;
    static constexpr int TASKS = G::XROWS * CB;
    static constexpr int FULL = TASKS / G::NT, REM = TASKS - FULL * G::NT;
    static constexpr int Q = REM == 0 ? 1 : 4 * REM <= G::NT ? 4 : 2 * REM <= G::NT ? 2 : 1;
    static constexpr int PART = 8 / Q;                 // channels of a left-over part
    float v[FULL > 0 ? FULL : 1][8];
    float w[PART];
};

// tA: time of window row 0; rows outside [0, T) read as zero (per-lane offset: the descriptor only bounds the tensor)
template <class G>
__device__ __forceinline__ void pairh_load_raw(PairHRaw<G>& r, const float* xb, int T, int tA, int tid) {
    typedef PairHRaw<G> R;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)G::C * (unsigned)T * 4u);
    const unsigned t4 = (unsigned)T * 4u;
#pragma unroll
    for (int q = 0; q < R::FULL; ++q) {
        const int idx = tid + q * G::NT;
        const int cb = idx / G::XROWS, row = idx - cb * G::XROWS;
        const int t = tA + row;
        const unsigned voff = t >= 0 && t < T ? (unsigned)(cb * 8 * T + t) * 4u : kOutOfRange;
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[q][j] = buffer_load1s(rx, voff, (unsigned)j * t4);
    }
    if constexpr (R::REM > 0) {
        const int idx = R::FULL * G::NT + tid / R::Q, part = tid % R::Q;
        const int cb = idx / G::XROWS, row = idx - cb * G::XROWS;
        const int t = tA + row;
        const bool ok = tid < R::Q * R::REM && t >= 0 && t < T;
        const unsigned voff = ok ? (unsigned)((cb * 8 + part * R::PART) * T + t) * 4u : kOutOfRange;
#pragma unroll
        for (int j = 0; j < R::PART; ++j) r.w[j] = buffer_load1s(rx, voff, (unsigned)j * t4);
    }
}

template <class G>
__device__ __forceinline__ void pairh_convert(const PairHRaw<G>& r, char* ximg, float slope, int tid, LowGuard& low) {
    typedef PairHRaw<G> R;
    float lowm = 0.f;
#pragma unroll
    for (int q = 0; q < R::FULL; ++q) {
        const int idx = tid + q * G::NT;
        const int cb = idx / G::XROWS, row = idx - cb * G::XROWS;
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
    if constexpr (R::REM > 0) {
        if (tid < R::Q * R::REM) {
            const int idx = R::FULL * G::NT + tid / R::Q, part = tid % R::Q;
            const int cb = idx / G::XROWS, row = idx - cb * G::XROWS;
            char* const dst = ximg + (cb * G::XRP + row) * 16 + part * (2 * R::PART);
            f16x2 h1[R::PART / 2], h2[R::PART / 2];
#pragma unroll
            for (int j = 0; j < R::PART / 2; ++j) {
                const f32x2 a = split_act2(f32x2{r.w[2 * j], r.w[2 * j + 1]}, slope);
                lowm = low_max3(lowm, a.x, a.y);
                split2(a, h1[j], h2[j]);
            }
            if constexpr (R::PART == 2) {
                *reinterpret_cast<f16x2*>(dst) = h1[0];
                *reinterpret_cast<f16x2*>(dst + G::XHALF) = h2[0];
            } else if constexpr (R::PART == 4) {
                *reinterpret_cast<f16x4*>(dst) = f16x4{h1[0].x, h1[0].y, h1[1].x, h1[1].y};
                *reinterpret_cast<f16x4*>(dst + G::XHALF) = f16x4{h2[0].x, h2[0].y, h2[1].x, h2[1].y};
            } else {
                *reinterpret_cast<f16x8*>(dst) = f16x8{h1[0].x, h1[0].y, h1[1].x, h1[1].y, h1[2].x, h1[2].y, h1[3].x, h1[3].y};
                *reinterpret_cast<f16x8*>(dst + G::XHALF) = f16x8{h2[0].x, h2[0].y, h2[1].x, h2[1].y, h2[2].x, h2[2].y, h2[3].x, h2[3].y};
            }
        }
    }
    low_note(low, 0, lowm);
}

// the two weight images of a member: global -> LDS by LDS-DMA ([conv1 | conv2], G::WB bytes each)
template <class G>
__device__ __forceinline__ void pairh_stage_weights(const PairMember& mb, float* wl, int wave, int lane) {
    constexpr int NI = G::WB / 1024;   // 64-lane x 16-byte instructions per conv
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(mb.w1, (unsigned)G::WB);
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(mb.w2, (unsigned)G::WB);
    for (int j = wave; j < NI; j += G::NW) {
        dma16(r1, wl + j * 256, (unsigned)(j * 1024 + lane * 16));
        dma16(r2, wl + G::WB / 4 + j * 256, (unsigned)(j * 1024 + lane * 16));
    }
}

// biases and the rows' inverse weight prescales (behind the packed images: api.hip row_scale_kernel) of a member ->
// LDS [b1[C] | b2[C] | s1[C] | s2[C]]
template <class G>
__device__ __forceinline__ void pairh_stage_bias(const PairMember& mb, float* bl, int tid) {
    if (tid < G::C) {
        bl[tid] = mb.b1 ? mb.b1[tid] : 0.f;
        bl[G::C + tid] = mb.b2 ? mb.b2[tid] : 0.f;
        bl[2 * G::C + tid] = mb.w1[G::WB / 4 + tid];
        bl[3 * G::C + tid] = mb.w2[G::WB / 4 + tid];
    }
}

// A operands of one conv, LDS -> registers (per phase): packed by fv_pack_pair_weight_ex as
// [(step * MH + row half) * 2 + split half][lane][8 halves]
template <class G>
__device__ __forceinline__ void pairh_load_a(const float* wl, f16x8 (&A)[G::KS][G::MH][2], int lane) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    LdsCF* base = lds_opaque(wl + 4 * lane);
#pragma unroll
    for (int s = 0; s < G::KS; ++s)
#pragma unroll
        for (int h = 0; h < G::MH; ++h) {
            A[s][h][0] = *reinterpret_cast<LdsH8*>(base + ((s * G::MH + h) * 2 + 0) * 256);
            A[s][h][1] = *reinterpret_cast<LdsH8*>(base + ((s * G::MH + h) * 2 + 1) * 256);
        }
}

// ---- one conv phase: hi += a1 b1, lo += a1 b2 + a2 b1 over the KS steps --------------------------------
// img: the lane's image base (row = its column of fragment 0 at the lane group's first tap, byte offset of
// its channel block); TAPB: bytes between consecutive K steps of the image.  B operands are read one step
// ahead of the MFMAs that use them (sched_barrier pins the order; the waitcnt pass derives the counts).
template <class G, int TAPB, int HALF>
__device__ __forceinline__ void pairh_mma(const float* wl, const char* img, f32x4 (&hi)[G::MH][G::NF],
                                          f32x4 (&lo)[G::MH][G::NF], int lane) {
    typedef __attribute__((address_space(3))) const f16x8 LdsH8;
    LdsCF* base = lds_opaque(reinterpret_cast<const float*>(img));
    f16x8 bq[2][G::NF][2];
    auto fetch = [&](int s, f16x8 (&dst)[G::NF][2]) {
#pragma unroll
        for (int f = 0; f < G::NF; ++f) {
            dst[f][0] = *reinterpret_cast<LdsH8*>(base + (s * TAPB + f * 256) / 4);
            dst[f][1] = *reinterpret_cast<LdsH8*>(base + (s * TAPB + f * 256 + HALF) / 4);
        }
    };
    auto mma = [&](const f16x8 (&a)[G::MH][2], const f16x8 (&bv)[G::NF][2]) {
#pragma unroll
        for (int h = 0; h < G::MH; ++h)
#pragma unroll
            for (int f = 0; f < G::NF; ++f)
                hi[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bv[f][0], hi[h][f], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < G::MH; ++h)
#pragma unroll
            for (int f = 0; f < G::NF; ++f)
                lo[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][0], bv[f][1], lo[h][f], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < G::MH; ++h)
#pragma unroll
            for (int f = 0; f < G::NF; ++f)
                lo[h][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h][1], bv[f][0], lo[h][f], 0, 0, 0);
    };
    if constexpr (G::AREG) {
        // the phase's A operands up front (C = 16: 48 registers at 11 taps)
        f16x8 A[G::KS][G::MH][2];
        pairh_load_a<G>(wl, A, lane);
        fetch(0, bq[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            if (s + 1 < G::KS) fetch(s + 1, bq[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            mma(A[s], bq[s & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // C = 32: a phase's A operands are 176 registers at 11 taps -- they stream through a two-step queue like B
        LdsCF* wb = lds_opaque(wl + 4 * lane);
        f16x8 aq[2][G::MH][2];
        auto fetch_a = [&](int s, f16x8 (&dst)[G::MH][2]) {
#pragma unroll
            for (int h = 0; h < G::MH; ++h) {
                dst[h][0] = *reinterpret_cast<LdsH8*>(wb + ((s * G::MH + h) * 2 + 0) * 256);
                dst[h][1] = *reinterpret_cast<LdsH8*>(wb + ((s * G::MH + h) * 2 + 1) * 256);
            }
        };
        fetch_a(0, aq[0]);
        fetch(0, bq[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            if (s + 1 < G::KS) {
                fetch_a(s + 1, aq[(s + 1) & 1]);
                fetch(s + 1, bq[(s + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(aq[s & 1], bq[s & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// items [item0, hi) of ONE member, in order (item = utterance * n_tiles + tile).
// Per tile: [loads of the NEXT tile's raw window and of this tile's residual are issued] conv1 -> intermediate,
// barrier, conv2, convert the next window into the x image (free since the barrier), stores, barrier.
// FOLD (the last pair of a HiFi-GAN, C = 16): the pair's output never goes to memory -- conv_post (16 -> 1 channels, 7
// taps, hifigan.py:104-106) runs on the activated tile in LDS: y_out = post( conv7( lrelu(x', act_slope) ) + b ).  Tiles
// then advance by NOUT - 6 samples and start 3 early (the 3-sample halo either side is recomputed).
template <class G, bool FOLD = false>
__device__ __forceinline__ void pairh_run_member(const PairParams& p, const PairMember& mb, int item0, int hi_item,
                                                 float* smem, int wave, int lane_in, bool first) {
    constexpr int TSTRIDE = FOLD ? G::NOUT - 6 : G::NOUT, TSHIFT = FOLD ? 3 : 0;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    float* const wl = smem + p.x_off;                  // [conv1 | conv2] packed weights
    char* const ximg = reinterpret_cast<char*>(smem + p.img_off);
    char* const mimg = reinterpret_cast<char*>(smem + p.mid_off);
    float* const bl = smem + p.bias_off;
    const int n = lane & 15, g = lane >> 4;
    const int col0 = wave * (16 * G::NF) + n;          // the lane's column in the wave's fragment 0
    const int tapg = G::TPS == 2 ? (g >> 1) : 0;       // the lane group's tap inside a K step
    const int cb = G::TPS == 2 ? (g & 1) : g;          // ... and its block of 8 channels
    const char* const xb = ximg + (cb * G::XRP + col0 + G::AOFF + tapg * G::DIL) * 16;
    const char* const mbase = mimg + (cb * G::MRP + col0 + tapg) * 16;
    // D fragment: channels 4g .. 4g+3 (+ 16 per row half) of column n = half of the 8-channel block g >> 1
    char* const mw = mimg + ((g >> 1) * G::MRP + col0) * 16 + 8 * (g & 1);
    const int row0 = 4 * g;

    const size_t ustride = (size_t)G::C * (size_t)p.T;
    constexpr int HEAD = G::P1 + G::P2 + G::AOFF;
    int item = item0;
    int b = item / mb.n_tiles, tile = item - b * mb.n_tiles;
    if (!first) pair_barrier();                         // everybody is done with the previous member's LDS
    pair_stamp(p, G::NW, wave, lane, 7, 12);
    float bad = 0.f;                                    // range guard: NaN once a final value was not finite
    f32x2 bad2 = {0.f, 0.f};                            //   (two rows per instruction outside the FOLD variant)
    const float rcp = div_rcp(p.out_div);               // the MRF mean's divisor (pair_kernels.hpp div_exact)
    LowGuard low;                                       // ... and its low side: were the operands small as a whole
    PairHRaw<G> raw;
    pairh_load_raw<G>(raw, mb.x + b * ustride, p.T, tile * TSTRIDE - TSHIFT - HEAD, tid);
    pairh_stage_weights<G>(mb, wl, wave, lane);
    pairh_stage_bias<G>(mb, bl, tid);
    // rows [NM, MROWS) of the intermediate feed only discarded columns / the zero tap: finite values once
    for (int idx = tid; idx < 2 * (G::C / 8) * 64; idx += G::NT)
        reinterpret_cast<float*>(mimg + ((idx >> 6) * G::MRP + G::NM) * 16)[idx & 63] = 0.f;
    pair_wait_vm0();
    pairh_convert<G>(raw, ximg, p.slope, tid, low);
    pair_barrier();
    pair_stamp(p, G::NW, wave, lane, 7, 13);
    for (;;) {
        const int t0 = tile * TSTRIDE - TSHIFT;
        const int nitem = item + 1;
        pair_stamp(p, G::NW, wave, lane, item - item0, 0);   // (tuning aid, -DFV_PAIR_TRACE: tools/pairh_trace.py)
        const bool more = nitem < hi_item;
        int nb = b, ntile = tile + 1;
        if (ntile == mb.n_tiles) {
            ntile = 0;
            ++nb;
        }
        // the next tile's raw window: in flight for the whole tile; then this tile's residual (fp32, L2 hits)
        // (requesting the window of the tile AFTER the next right behind the conversion instead -- the registers are free
        // from there -- was measured slower: 383 against 338 us per three-member launch at C = 32, B = 8: the loads then
        // compete with the tile's own stores and residual reads; round 4, tools/build_variant.py)
        if (more && !(p.dbg & 1)) pairh_load_raw<G>(raw, mb.x + nb * ustride, p.T, ntile * TSTRIDE - TSHIFT - HEAD, tid);
        float res[G::MH][G::NF][4];
        const unsigned ubytes = (unsigned)G::C * (unsigned)p.T * 4u;
        const unsigned t4 = (unsigned)p.T * 4u;
        unsigned voff[G::MH][G::NF];
        {
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(mb.x + b * ustride, ubytes);
#pragma unroll
            for (int h = 0; h < G::MH; ++h)
#pragma unroll
                for (int f = 0; f < G::NF; ++f) {
                    const int col = col0 + f * 16, t = t0 + col;
                    const bool ok = col < G::NOUT && t >= 0 && t < p.T && !(p.dbg & 16);
                    voff[h][f] = ok ? (unsigned)((16 * h + row0) * p.T + t) * 4u : kOutOfRange;
#pragma unroll
                    for (int i = 0; i < 4; ++i) res[h][f][i] = buffer_load1s(rx, voff[h][f], (unsigned)i * t4);
                }
        }
        f32x4 hi[G::MH][G::NF], lo[G::MH][G::NF];
#pragma unroll
        for (int h = 0; h < G::MH; ++h)
#pragma unroll
            for (int f = 0; f < G::NF; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(p.dbg & 4)) pairh_mma<G, G::TPS * G::DIL * 16, G::XHALF>(wl, xb, hi, lo, lane);
        pair_stamp(p, G::NW, wave, lane, item - item0, 1);
        {
            // intermediate column u of the tile is time t0 - P2 + u; conv2's zero padding applies to the
            // intermediate: columns outside [0, T) are zero, not conv1 of the padded input
            const int tm = t0 - G::P2;
            const bool inside = tm >= 0 && tm + G::NM <= p.T;     // (uniform) no column of this tile needs the mask
            float lowm = 0.f;                            // largest magnitude of this tile's intermediate in this lane
#pragma unroll
            for (int h = 0; h < G::MH; ++h) {
                const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + 16 * h + row0);
                const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 2 * G::C + 16 * h + row0);
                const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
                for (int f = 0; f < G::NF; ++f) {
                    const int t = tm + col0 + f * 16;
                    f16x4 h1, h2;
                    if (inside) split_mid4<false>(hi[h][f], lo[h][f], s01, s23, b01, b23, p.slope, true, h1, h2, lowm);
                    else split_mid4<true>(hi[h][f], lo[h][f], s01, s23, b01, b23, p.slope, t >= 0 && t < p.T, h1, h2, lowm);
                    *reinterpret_cast<f16x4*>(mw + f * 256 + h * (2 * G::MRP * 16)) = h1;
                    *reinterpret_cast<f16x4*>(mw + f * 256 + h * (2 * G::MRP * 16) + G::MHALF) = h2;
                }
            }
            low_note(low, 1, lowm);
        }
        pair_stamp(p, G::NW, wave, lane, item - item0, 2);
        pair_barrier();                                  // (C) intermediate complete, x image free
        pair_stamp(p, G::NW, wave, lane, item - item0, 3);
#pragma unroll
        for (int h = 0; h < G::MH; ++h)
#pragma unroll
            for (int f = 0; f < G::NF; ++f) hi[h][f] = lo[h][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(p.dbg & 4)) pairh_mma<G, G::TPS * 16, G::MHALF>(wl + G::WB / 4, mbase, hi, lo, lane);
        pair_stamp(p, G::NW, wave, lane, item - item0, 4);
        // raw window and residual have been in flight for two conv phases; no store is outstanding here
        // (the previous tile's were issued a tile ago and are drained with the same wait)
        pair_wait_vm0();
        pair_stamp(p, G::NW, wave, lane, item - item0, 5);
        if (more && !(p.dbg & 2)) pairh_convert<G>(raw, ximg, p.slope, tid, low);
        pair_stamp(p, G::NW, wave, lane, item - item0, 6);
        const bool fin = mb.add1 != nullptr;
#pragma unroll
        for (int h = 0; h < G::MH; ++h) {
            const f32x2* const b2 = reinterpret_cast<const f32x2*>(bl + G::C + 16 * h + row0);
            const f32x2* const s2 = reinterpret_cast<const f32x2*>(bl + 3 * G::C + 16 * h + row0);
            const f32x2 b01 = b2[0], b23 = b2[1], s01 = s2[0], s23 = s2[1];
#pragma unroll
            for (int f = 0; f < G::NF; ++f) combine4(hi[h][f], lo[h][f], s01, s23, b01, b23, res[h][f]);
        }
        if (fin) {
            // last launch of an MRF stage: (r0 + r1) + r2 in the reference's order (hifigan.py:99-103)
            const __amdgpu_buffer_rsrc_t r1 = make_rsrc(mb.add1 + b * ustride, ubytes);
            const __amdgpu_buffer_rsrc_t r2 = make_rsrc(mb.add2 ? mb.add2 + b * ustride : mb.add1, mb.add2 ? ubytes : 0u);
#pragma unroll
            for (int h = 0; h < G::MH; ++h)
#pragma unroll
                for (int f = 0; f < G::NF; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        lo[h][f][i] = buffer_load1s(r1, voff[h][f], (unsigned)i * t4);
                        res[h][f][i] = buffer_load1s(r2, voff[h][f], (unsigned)i * t4);
                    }
            pair_wait_vm0();
#pragma unroll
            for (int h = 0; h < G::MH; ++h)
#pragma unroll
                for (int f = 0; f < G::NF; ++f)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hi[h][f][i] = (hi[h][f][i] + lo[h][f][i]) + res[h][f][i];
        }
        pair_stamp(p, G::NW, wave, lane, item - item0, 7);
        // (FOLD: the activated tile below overwrites the zeroed rows behind the intermediate, so discarded columns can read
        // any bit pattern there: only stored values count.  Otherwise every column is computed from real, zero-padded data.)
#pragma unroll
        for (int h = 0; h < G::MH; ++h)
#pragma unroll
            for (int f = 0; f < G::NF; ++f) {
                if constexpr (FOLD) {
                    const int col = col0 + f * 16, t = t0 + col;
                    range_note4(bad, hi[h][f][0], hi[h][f][1], hi[h][f][2], hi[h][f][3], col < G::NOUT && t >= 0 && t < p.T);
                } else {
                    range_note4p(bad2, hi[h][f]);
                }
            }
        if constexpr (FOLD) {
            // the activated tile -> LDS (the intermediate image's space: every wave is past its conv2 reads after the
            // barrier), zero outside [0, T) and beyond the tile's valid columns; then one output sample per thread
            float* const sb = reinterpret_cast<float*>(mimg);          // [C][NM]
            pair_barrier();
#pragma unroll
            for (int h = 0; h < G::MH; ++h)
#pragma unroll
                for (int f = 0; f < G::NF; ++f) {
                    const int col = col0 + f * 16, t = t0 + col;
                    const bool ok = col < G::NOUT && t >= 0 && t < p.T;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = hi[h][f][i];
                        if (fin) v = rcp != 0.f ? div_exact(v, p.out_div, rcp) : v / p.out_div;
                        sb[(16 * h + row0 + i) * G::NM + col] = ok ? act(v, p.act_slope) : 0.f;
                    }
                }
            pair_barrier();
            if (tid < TSTRIDE) {
                const int t = t0 + 3 + tid;
                // (channel-major FMA chain from zero, bias last: conv_narrow_kernel's arithmetic -- the plans that keep
                // conv_post as a launch of its own, e.g. inference_minus, give the same bits)
                float o = 0.f;
#pragma unroll 2
                for (int c = 0; c < G::C; ++c)
#pragma unroll
                    for (int j = 0; j < 7; ++j) o = fmaf(p.fold_w[c * 7 + j], sb[c * G::NM + tid + j], o);
                o = o + (p.fold_b ? p.fold_b[0] : 0.f);
                if (p.post == FV_POST_TANH) o = tanhf(o);
                else if (p.post == FV_POST_RELU) o = fmaxf(o, 0.f);
                if (t < p.T && !(p.dbg & 8)) p.fold_y[(size_t)b * p.T + t] = o;
            }
        } else {
#pragma unroll
        for (int h = 0; h < G::MH; ++h)
#pragma unroll
            for (int f = 0; f < G::NF; ++f) {
                const int col = col0 + f * 16;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = hi[h][f][i];
                pair_store(p, mb.y, mb.y_act, G::C, b, 16 * h + row0, t0 + col,
                           col < G::NOUT && t0 + col < p.T && !(p.dbg & 8), v, fin, rcp);
            }
        }
        pair_stamp(p, G::NW, wave, lane, item - item0, 8);
        if (!more) break;
        pair_barrier();                                  // (A) the next x image is complete; the intermediate is free
        pair_stamp(p, G::NW, wave, lane, item - item0, 9);
        item = nitem;
        b = nb;
        tile = ntile;
    }
    range_flag(p, bad + (bad2.x + bad2.y));
    low_flag(p, low, bl + 4 * G::C, wave, lane, G::NW);
}

template <int MH, int NF, int NG, int DIL, bool FOLD>
__device__ __forceinline__ void pairh_run_any(const PairParams& p, const PairMember& mb, int item0, int hi, float* smem,
                                              int wave, int lane, bool first) {
    if (mb.k == 11) pairh_run_member<PairHGeom<MH, NF, NG, 11, DIL>, FOLD>(p, mb, item0, hi, smem, wave, lane, first);
    else if (mb.k == 7) pairh_run_member<PairHGeom<MH, NF, NG, 7, DIL>, FOLD>(p, mb, item0, hi, smem, wave, lane, first);
    else pairh_run_member<PairHGeom<MH, NF, NG, 3, DIL>, FOLD>(p, mb, item0, hi, smem, wave, lane, first);
}

// waves per SIMD the register budget is cut for: C = 32 -- one block of NG waves per CU (8: 2, 12: 3, 13-16: 4);
// C = 16 -- two blocks per CU: 4-wave blocks with four fragments per wave: 2 (256 VGPRs), 8-wave blocks with two: 4
constexpr int pairh_waves_per_simd(int MH, int NF, int NG) {
    return MH == 1 ? (NG >= 8 ? 4 : 2) : (NG > 12 ? 4 : NG > 8 ? 3 : 2);
}

template <int MH, int NF, int NG, int DIL, bool FOLD = false>
__global__ __launch_bounds__(64 * NG) __attribute__((amdgpu_waves_per_eu(pairh_waves_per_simd(MH, NF, NG), pairh_waves_per_simd(MH, NF, NG)))) void pairh_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    pair_stamp(p, NG, wave, lane, 7, 15);
    // the launch's scalars in one batch of kernarg loads (see convh_kernel: left to itself hipcc loads every field
    // right before its first use, a chain of dependent s_load round trips between kernel entry and the first tile)
    PairParams q;
    q.n_members = p.n_members; q.B = p.B; q.T = p.T; q.nblk = p.nblk; q.slope = p.slope; q.out_div = p.out_div;
    q.act_slope = p.act_slope; q.post = p.post; q.x_off = p.x_off; q.img_off = p.img_off; q.mid_off = p.mid_off;
    q.bias_off = p.bias_off; q.dbg = p.dbg; q.trace = p.trace; q.guard = p.guard;
    q.fold_w = p.fold_w; q.fold_b = p.fold_b; q.fold_y = p.fold_y;
    int n_tiles[3], cost[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) { n_tiles[m] = p.m[m].n_tiles; cost[m] = p.m[m].cost; }
    asm volatile("" ::"s"(q.n_members), "s"(q.B), "s"(q.T), "s"(q.nblk), "s"(q.slope), "s"(q.out_div), "s"(q.act_slope),
                 "s"(q.post), "s"(q.x_off), "s"(q.img_off), "s"(q.mid_off), "s"(q.bias_off), "s"(q.dbg), "s"(q.trace),
                 "s"(n_tiles[0]), "s"(n_tiles[1]), "s"(n_tiles[2]), "s"(cost[0]), "s"(cost[1]), "s"(cost[2]), "s"(q.fold_w),
                 "s"(q.fold_b), "s"(q.fold_y), "s"(q.guard));
    // This block's share of the cost-weighted item sequence (members one after the other): from the host's table when
    // there is one (p.sched_on == 2, pair_cut_schedule: the global item number every share starts at -- two scalar loads
    // instead of ~600 scalar instructions of 64-bit arithmetic between kernel entry and the first load, 1.8 us per launch
    // [measured, tools/pairh_trace.py]), else computed here (pair_share).
    const int share = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const bool cut = p.sched_on == 2;
    int g_lo = 0, g_hi = 0;
    if (cut) {
        g_lo = (int)p.sched[share];                     // (nblk entries; the last share ends with the last item)
        g_hi = share + 1 < q.nblk ? (int)p.sched[share + 1] : (n_tiles[0] + (q.n_members > 1 ? n_tiles[1] : 0) + (q.n_members > 2 ? n_tiles[2] : 0)) * q.B;
        asm volatile("" ::"s"(g_lo), "s"(g_hi));
    }
    long long total = 0;
    if (!cut) {
#pragma unroll
        for (int m = 0; m < 3; ++m) total += m < q.n_members ? (long long)n_tiles[m] * q.B * cost[m] : 0;
    }
    long long base = 0;
    int off = 0;
    bool first = true;
    for (int m = 0; m < q.n_members; ++m) {
        const int nt = m == 0 ? n_tiles[0] : m == 1 ? n_tiles[1] : n_tiles[2];
        const int cm = m == 0 ? cost[0] : m == 1 ? cost[1] : cost[2];
        const int n = nt * q.B;
        int lo, hi;
        if (cut) {
            lo = min(max(g_lo - off, 0), n);
            hi = min(max(g_hi - off, 0), n);
            off += n;
        } else {
            lo = pair_share(share, total, base, cm, n, q.nblk);
            hi = pair_share(share + 1, total, base, cm, n, q.nblk);
            base += (long long)n * cm;
        }
        if (lo >= hi) continue;
        // ... and this member's pointers and sizes in one more
        PairMember mb;
        mb.x = p.m[m].x; mb.w1 = p.m[m].w1; mb.w2 = p.m[m].w2; mb.b1 = p.m[m].b1; mb.b2 = p.m[m].b2; mb.add1 = p.m[m].add1;
        mb.add2 = p.m[m].add2; mb.y = p.m[m].y; mb.y_act = p.m[m].y_act; mb.k = p.m[m].k; mb.n_tiles = nt;
        asm volatile("" ::"s"(mb.x), "s"(mb.w1), "s"(mb.w2), "s"(mb.b1), "s"(mb.b2), "s"(mb.add1), "s"(mb.add2), "s"(mb.y),
                     "s"(mb.y_act), "s"(mb.k));
        pairh_run_any<MH, NF, NG, DIL, FOLD>(q, mb, lo, hi, smem, wave, lane, first);
        first = false;
    }
}

}  // namespace fv
