// Implicit-GEMM 1-D convolution on the gfx950 fp32 matrix cores.
//
// One kernel family serves every convolution on the generator path
// (reference call sites: model/generator/modules.py:223-230 ResBlock1,
// :372-382 ResidualStack, hifigan.py:93-96 conv_pre / ConvTranspose1d,
// melgan.py:66-85, basis_melgan.py:72-97, modules.py:264-267 basis matmul+OLA):
//
//   Y[m, q] = sum_{ci, j} Wp[ci, j, m] * act(X[ci, q + j*dil - pad])
//
// M = Cout rows for Conv1d; for ConvTranspose1d the rows are the Cout*stride
// output phases of its polyphase form (fv_internal.h: polyphase()), so the
// same kernel runs it as a short dense conv and the epilogue interleaves the
// phases back into time ("pixel shuffle").  Arithmetic is exact fp32:
// v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 are bit-for-bit fmaf chains,
// which is what the 1e-4 end-to-end budget over ~80 chained layers needs
// (bf16/fp16 MFMA does not fit it; SURVEY.md section 7 "hard parts").
//
// Structure (per workgroup of WM x WN x WK wave64), shaped by measured facts
// (tools/*_probe.hip, profiles/): memory latency under load is ~2 us while one
// (tile, channel-chunk) stage holds 0.5-2 us of matrix work; and on gfx950 plain
// VALU instructions do NOT co-execute with v_mfma_f32_* -- every VALU costs ~3
// cycles of matrix time -- so the hot loop must be MFMA + ds_read and nothing else:
//   * a block handles one time tile (a run of tiles only when the grid would exceed
//     FV_GRID_CAP blocks); per (tile, channel-chunk) STAGE
//         xs[ci_chunk][xw]      input window with its dilation halo
//         ws[ci_chunk*k][M_T]   K-major weight slice (dense 2-D block of Wp)
//     are brought in by LDS-DMA (buffer_load_dwordx4 ... lds): no staging VGPRs,
//     no ds_write pass, bounds-checked by the buffer descriptor (rows past Cin
//     read as 0).  Stages form one linear pipeline across chunk and tile
//     boundaries: the DMA of stage s+1 is issued before the MFMA loop of stage s
//     into the other LDS buffer; one barrier per stage;
//   * the input activation is NOT in the hot loop: plans feed every conv a
//     tensor that already holds act(x) (the producing epilogue writes it, next
//     to the raw tensor when a residual also needs that), see engine.py.  A
//     read-time activation (max(x, slope*x)) exists only for the stand-alone
//     operator entry points (ACT = true variants);
//   * tap count and dilation are template parameters for the hot shapes, so
//     every A/B operand is a ds_read with an immediate offset (paired into
//     ds_read2_b32 by the compiler): two address adds per K step, nothing per MFMA;
//   * everything AROUND the hot loop is written for instruction count too (a tile
//     has only 100-350 MFMAs per wave): per-lane DMA offsets once per block and
//     an issue sequence LLVM cannot hoist into spilled SGPRs (opaque_uniform),
//     branch-free row setup through bounds-checked descriptors, affine epilogue
//     addressing (one vector offset + a scalar per row) for plain convs;
//   * 118-124 VGPRs => 4 waves per SIMD, 39 KiB of LDS per block => 4 blocks per CU;
//   * WK > 1 splits the K range of every stage over WK wave groups that share
//     the staged tiles and reduce through LDS at the end of the tile;
//   * tiles that touch the sequence ends with reflection padding, and unaligned
//     tensors, take a synchronous register path (stage_x_edge, SLOW variants);
//     zero-padded edges of aligned tensors are still DMA (masked lanes);
//   * the epilogue fuses bias, residual add, the MRF running sum / mean,
//     tanh / ReLU and the optional activated twin output through bounds-checked
//     buffer loads/stores.
#pragma once
#include <stdlib.h>

#include "fv_internal.h"

namespace fv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Out-of-range marker for buffer offsets.  Descriptors never cover more than 1 GiB
// (host-checked), so any offset with bit 30 or 31 set is out of range (loads give 0,
// stores are dropped).  The marker is ADDITIVE: marker + valid offset and marker +
// marker stay out of range, so masked lanes need neither a compare nor a select --
// hoisted per-lane predicates would otherwise pile up as 64-bit SGPR masks and spill.
constexpr unsigned kOutOfRange = 0x40000000u;

// leaky-ReLU / ReLU / identity for 0 <= slope <= 1 as max(x, slope*x): bitwise
// equal to x >= 0 ? x : slope*x, branch-free (slope is wave-uniform)
__device__ __forceinline__ float act(float v, float slope) { return fmaxf(v, v * slope); }

// Workgroup barrier for LDS hand-offs WITHOUT __syncthreads()'s fences: the release fence makes hipcc wait
// for every global store in flight (vmcnt) before the barrier, so a block that walks several tiles pays
// the ~2 us drain of the previous tile's stores at the next stage boundary.  LDS traffic only needs this
// wave's LDS operations retired (lgkmcnt); LDS-DMA landing is ordered by the explicit vmcnt waits.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A wave-uniform value the optimiser must treat as unknown at this point.  The DMA issue code
// runs once per stage inside the tile/chunk loops; left alone, LLVM hoists every per-instruction
// predicate (as a 64-bit lane mask) and LDS address out of those loops, runs out of SGPRs and
// spills them to VGPR lanes -- ~3 v_readlane per DMA slot per stage, i.e. more issue slots than
// the staging itself.  Laundering the two scalars they derive from keeps them as one compare
// with an immediate and one s_add with a literal at the point of use.
__device__ __forceinline__ int opaque_uniform(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

__device__ __forceinline__ int reflect_idx(int i, int T) {
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    // columns staged only for alignment slack or tile overhang may still fall
    // outside; they feed masked outputs only, so clamp instead of faulting
    return min(max(i, 0), T - 1);
}

// interior <=> every column the tile reads is a real sample and rows are 16-byte
// aligned, so the window can be copied verbatim by the DMA engine
__device__ __forceinline__ bool interior(const ConvParams& p, int tA) {
    return p.vec_ok && tA >= 0 && tA + 4 * p.ncol4 <= p.Tin;
}

// Buffer descriptor over [base, base+bytes): accesses through it are
// bounds-checked by the hardware (out of range: loads give 0, stores are
// dropped), so masked lanes need no branch -- they get an out-of-range offset.
// Built from kernel arguments and blockIdx only => provably wave-uniform.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buffer_load1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
// the same with a wave-uniform byte offset in the instruction's SGPR operand (no VALU add); the
// hardware bounds check covers the VGPR part only, so the scalar part must stay inside the buffer
__device__ __forceinline__ float buffer_load1s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned s_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)s_off, 0));
}
__device__ __forceinline__ void buffer_store1s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned s_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, (int)s_off, 0);
}
__device__ __forceinline__ void buffer_store1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, 0, 0);
}
// ... with cache-policy bits (AUX: 0 plain, kAuxAgent = sc1: the access is coherent device-wide -- stores write through
// the XCD's L2, loads do not hit in it; what tensors handed between blocks INSIDE a launch would need)
constexpr int kAuxAgent = 16;
template <int AUX>
__device__ __forceinline__ float buffer_load1s_aux(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned s_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)s_off, AUX));
}
template <int AUX>
__device__ __forceinline__ void buffer_store1s_aux(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned s_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, (int)s_off, AUX);
}
// 16 bytes per lane straight into LDS: lane l lands at lds + 16*l (wave-uniform base)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* lds, unsigned byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, (int)byte_off, 0, 0, 0);
}

// --- asynchronous staging (LDS-DMA) --------------------------------------------
// The x image is [ci_chunk][ncol4c] float4, the w image [ci_chunk*k][M_T/4]
// float4, both linear in their float4 index: DMA instruction j of the block
// covers float4s [64j, 64j+64); waves take instructions round-robin.  The
// per-lane source offsets do not depend on the stage except for a uniform
// base, so they are computed once (DmaPlan) and each stage costs one add per
// instruction.

// (rows past Cin / past Cin*k need no test: their offsets fall outside the
// buffer descriptors, which cover exactly this utterance's input / the weights)
struct DmaPlan {
    unsigned xoff[kMaxDmaX];   // byte offset inside the (ci0, tA) window, or kOutOfRange
    unsigned woff[kMaxDmaW];   // byte offset inside the chunk's weight rows, or kOutOfRange
};

template <int NW, int M_T>
__device__ __forceinline__ void dma_plan(const ConvParams& p, DmaPlan& d, int m0, int wave, int lane) {
    constexpr int C4 = M_T / 4;
    const int xtotal = p.ci_chunk * p.ncol4c;
    const int wtotal = p.ci_chunk * p.k * C4;
#pragma unroll
    for (int i = 0; i < kMaxDmaX; ++i) {
        d.xoff[i] = kOutOfRange;
        if (wave + i * NW < p.nx_inst) {   // wave-uniform: unused slots cost nothing
            const int idx = (wave + i * NW) * 64 + lane;
            const int row = (int)__umulhi((unsigned)idx, p.ncol4c_magic);
            const int c4 = idx - row * p.ncol4c;
            if (idx < xtotal) d.xoff[i] = (unsigned)(row * p.Tin + 4 * c4) * 4u;
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxDmaW; ++i) {
        d.woff[i] = kOutOfRange;
        if (wave + i * NW < p.nw_inst) {
            const int idx = (wave + i * NW) * 64 + lane;
            const int row = idx / C4, c = idx % C4;
            if (idx < wtotal) d.woff[i] = (unsigned)(row * p.Mpad + m0 + 4 * c) * 4u;
        }
    }
}

// number of DMA instructions of a stage image with n_inst instructions that fall to this wave
template <int NW>
__device__ __forceinline__ int wave_share(int n_inst, int wave) {
    return n_inst > wave ? (n_inst - wave + NW - 1) / NW : 0;
}

template <int NW>
__device__ __forceinline__ void dma_x(const ConvParams& p, const DmaPlan& d, __amdgpu_buffer_rsrc_t rx,
                                      float* xs, int ci0, int tA, int wave) {
    const unsigned base = (unsigned)(ci0 * p.Tin + tA) * 4u;
    const int n = opaque_uniform(wave_share<NW>(p.nx_inst, wave));
    float* const xw = xs + opaque_uniform(wave * 256);
#pragma unroll
    for (int i = 0; i < kMaxDmaX; ++i)
        if (i < n) dma16(rx, xw + i * (NW * 256), d.xoff[i] + base);
}

// Zero-padded tiles at the sequence ends, still by DMA: with 16-byte aligned rows
// (Tin % 4 == 0) and tA a multiple of 4, every float4 lies entirely inside or
// entirely outside [0, Tin), so padding is just one more out-of-range case.
template <int NW>
__device__ __forceinline__ void dma_x_zero_edge(const ConvParams& p, __amdgpu_buffer_rsrc_t rx,
                                                float* xs, int cin_src, int ci0, int tA, int wave,
                                                int lane) {
    for (int j = wave; j < p.nx_inst; j += NW) {
        const int idx = j * 64 + lane;
        const int row = (int)__umulhi((unsigned)idx, p.ncol4c_magic);
        const int t = tA + 4 * (idx - row * p.ncol4c);
        const bool ok = row < p.ci_chunk && ci0 + row < cin_src && t >= 0 && t < p.Tin;
        dma16(rx, xs + j * 256, ok ? (unsigned)((ci0 + row) * p.Tin + t) * 4u : kOutOfRange);
    }
}

template <int NW, int M_T>
__device__ __forceinline__ void dma_w(const ConvParams& p, const DmaPlan& d, __amdgpu_buffer_rsrc_t rw,
                                      float* ws, int ci0, int wave) {
    const unsigned base = (unsigned)(ci0 * p.k * p.Mpad) * 4u;
    const int n = opaque_uniform(wave_share<NW>(p.nw_inst, wave));
    float* const ww = ws + opaque_uniform(wave * 256);
#pragma unroll
    for (int i = 0; i < kMaxDmaW; ++i)
        if (i < n) dma16(rw, ww + i * (NW * 256), d.woff[i] + base);
}

// Synchronous path for tiles that touch the sequence ends or unaligned tensors:
// zero / reflection padding resolved per element, raw values written to LDS.
template <int NT>
__device__ __forceinline__ void stage_x_edge(const ConvParams& p, float* xs, const float* xb, int cin_src,
                                             int ci0, int tA, int tid) {
    const int total = p.ci_chunk * p.ncol4c;
    for (int idx = tid; idx < total; idx += NT) {
        const int row = (int)__umulhi((unsigned)idx, p.ncol4c_magic);
        const int c4 = idx - row * p.ncol4c;
        const int ci = ci0 + row;
        const int t = tA + 4 * c4;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (ci < cin_src) {
            const float* xr = xb + (size_t)ci * (size_t)p.Tin;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tj = t + j;
                if (p.pad_mode == FV_PAD_REFLECT) e[j] = xr[reflect_idx(tj, p.Tin)];
                else if (tj >= 0 && tj < p.Tin) e[j] = xr[tj];
            }
        }
        *reinterpret_cast<float4*>(xs + idx * 4) = make_float4(e[0], e[1], e[2], e[3]);
    }
}

// Stage one input window: DMA for interior tiles, masked DMA for zero-padded edge
// tiles of aligned tensors; SLOW variants (reflection padding, unaligned rows) fall
// back to the synchronous per-element path.  Returns true when the data is in flight
// asynchronously (false: it was written synchronously by stage_x_edge).
// (xb, cin_src, rx) describe the source tensor of this stage for batch item b -- p.x, or
// p.x2 for the rows past Cin1 of a two-source conv; ci0 counts channels inside that source.
template <int NW, int NT, bool SLOW>
__device__ __forceinline__ void stage_x(const ConvParams& p, const DmaPlan& d, __amdgpu_buffer_rsrc_t rx,
                                        float* xs, const float* xb, int cin_src, int ci0, int tA, int wave,
                                        int lane, int tid) {
    if (interior(p, tA)) {
        dma_x<NW>(p, d, rx, xs, ci0, tA, wave);
    } else {
        if constexpr (SLOW) {
            if (p.vec_ok && p.pad_mode == FV_PAD_ZERO) dma_x_zero_edge<NW>(p, rx, xs, cin_src, ci0, tA, wave, lane);
            else stage_x_edge<NT>(p, xs, xb, cin_src, ci0, tA, tid);
        } else {
            dma_x_zero_edge<NW>(p, rx, xs, cin_src, ci0, tA, wave, lane);
        }
    }
}

// --- fused epilogue --------------------------------------------------------------
// Descriptors of the epilogue tensors of batch item b: a masked element simply
// gets an out-of-range offset (loads 0, store dropped), so the loads of a tile
// issue back to back instead of one s_waitcnt vmcnt(0) per element.
struct EpilogueRsrc {
    __amdgpu_buffer_rsrc_t y, y2, res, acc, acc2, sub;
};

__device__ __forceinline__ EpilogueRsrc epilogue_rsrc(const ConvParams& p, int b) {
    const size_t boff = (size_t)b * p.Cout * (size_t)p.Tout;
    const unsigned bytes = (unsigned)p.Cout * (unsigned)p.Tout * 4u;
    EpilogueRsrc e;
    e.y = make_rsrc(p.y + boff, bytes);
    e.y2 = make_rsrc(p.y_act ? p.y_act + boff : p.y + boff, bytes);
    e.res = make_rsrc(p.res ? p.res + boff : p.y + boff, bytes);
    e.acc = make_rsrc(p.acc_in ? p.acc_in + boff : p.y + boff, bytes);
    e.acc2 = make_rsrc(p.acc_in2 ? p.acc_in2 + boff : p.y + boff, bytes);
    e.sub = make_rsrc(p.sub ? p.sub + (p.sub_batched ? boff : 0) : p.y + boff, bytes);
    return e;
}

// Per-thread constants of the N GEMM rows it owns: byte offset of (row, t = 0) in
// the output tensor (or kOutOfRange for padded rows) and the row's bias.
template <int N>
struct RowInfo {
    unsigned off[N];
    float bias[N];
    unsigned short_mask;   // transposed convs: bit i set = row i has no sample in the last column
};

// Branch-free: the bias comes through a bounds-checked descriptor over exactly Cout floats
// (zero records when the layer has no bias), so padded rows and bias-less layers read 0
// without a predicate; the row offset of a padded row gets the additive out-of-range marker.
template <int N>
__device__ __forceinline__ void row_info(const ConvParams& p, const int (&m)[N], RowInfo<N>& ri) {
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.bias ? p.bias : p.wp, p.bias ? (unsigned)p.Cout * 4u : 0u);
    ri.short_mask = 0u;
    if (p.ups == 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            ri.off[i] = (unsigned)(m[i] * p.Tout) * 4u + (m[i] < p.M ? 0u : kOutOfRange);
            ri.bias[i] = buffer_load1(rb, (unsigned)m[i] * 4u);
        }
    } else {
        const int rem = p.Tout - (p.Tq - 1) * p.ups;   // phases present in the last column
#pragma unroll
        for (int i = 0; i < N; ++i) {
            int co = m[i] / p.ups, ph = m[i] - co * p.ups;         // rows co-major: m = co*ups + phase
            if (p.phase_major) {                                    // rows phase-major: m = phase*Cout + co
                ph = m[i] / p.Cout;
                co = m[i] - ph * p.Cout;
            }
            ri.off[i] = (unsigned)(co * p.Tout + ph) * 4u + (m[i] < p.M ? 0u : kOutOfRange);
            ri.short_mask |= (ph >= rem ? 1u : 0u) << i;   // see epilogue_offsets
            ri.bias[i] = buffer_load1(rb, (unsigned)co * 4u);
        }
    }
}

// N output elements of one thread at GEMM column q:
//   y = post( ( (acc_in + acc_in2) + ( (v + bias) + res ) ) / out_div );   y_act = act(y, act_slope)
// all uniform switches are hoisted; loads are issued as one batch.  (Issuing the residual read
// before the tile's last MFMAs was tried: the 16 extra live registers cost more than the ~2 us
// of latency they hid.)
template <int N>
__device__ __forceinline__ void epilogue_offsets(const ConvParams& p, const RowInfo<N>& ri,
                                                 const int (&m)[N], int q, unsigned (&off)[N]) {
    // additive masking (see kOutOfRange): no per-element predicate
    const unsigned qoff = q < p.Tq ? (unsigned)(q * p.ups) * 4u : kOutOfRange;
#pragma unroll
    for (int i = 0; i < N; ++i) off[i] = ri.off[i] + qoff;
    if (p.ups != 1) {
        // A transposed conv's last column can run past Tout (Tout need not be a multiple of ups):
        // row_info recorded which rows have no sample there (short_mask), and those get the
        // out-of-range marker when q is the last column.  The mask goes through an empty asm so
        // that nothing derived from it can be hoisted out of the tile loop: LLVM hoists any
        // loop-invariant arithmetic above this (uniform) branch -- with the original test
        // q*ups + m % ups >= Tout that was N modulo computations, ~280 VALU instructions per
        // tile for EVERY conv, transposed or not.
        unsigned sm = ri.short_mask;
        asm volatile("" : "+v"(sm));
        const unsigned sel = q == p.Tq - 1 ? sm : 0u;
#pragma unroll
        for (int i = 0; i < N; ++i) off[i] += ((sel >> i) & 1u) << 30;
    }
}

// The tensor reads: res for every conv2 of a ResBlock; acc_in / acc_in2 only on the last conv
// of an MRF stage -- those live in a branch so that the common path neither zero-fills nor
// adds 2 x N registers.  All loads of a batch are issued before the first use.
// Element i lives at byte offset off[i] + so[i], so[i] being wave-uniform (0 in the general case).
template <int N>
__device__ __forceinline__ void epilogue_finish(const ConvParams& p, const EpilogueRsrc& e,
                                                const float (&bias)[N], const unsigned (&off)[N],
                                                const unsigned (&so)[N], float (&v)[N]) {
    if (p.acc_in) {
        // (acc_in + acc_in2) first, like xs = r0; xs += r1; then + this block's output (hifigan.py:99-102)
        float rv[N], av[N], a2[N];
#pragma unroll
        for (int i = 0; i < N; ++i) rv[i] = p.res ? buffer_load1s(e.res, off[i], so[i]) : 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) av[i] = buffer_load1s(e.acc, off[i], so[i]);
#pragma unroll
        for (int i = 0; i < N; ++i) a2[i] = p.acc_in2 ? buffer_load1s(e.acc2, off[i], so[i]) : 0.f;
        // either association reproduces xs = r0; xs += r1; xs += r2 exactly, depending on which
        // block's conv carries the sum: the last one (own = r2) or the first (own = r0)
        if (p.own_first) {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = (((v[i] + bias[i]) + rv[i]) + av[i]) + a2[i];
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = (av[i] + a2[i]) + ((v[i] + bias[i]) + rv[i]);
        }
    } else if (p.res) {
        float rv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) rv[i] = buffer_load1s(e.res, off[i], so[i]);
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = (v[i] + bias[i]) + rv[i];
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = v[i] + bias[i];
    }
    if (p.out_div != 1.f) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = v[i] / p.out_div;
    }
    if (p.post == FV_POST_TANH) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = tanhf(v[i]);
    } else if (p.post == FV_POST_RELU) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (p.sub) {
        // bias-removal flows (last op of a graph only): y raw, y2 = act(y) - sub; or y = act(y) - sub
        float sv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) sv[i] = buffer_load1s(e.sub, off[i], so[i]);
        if (p.y_act) {
#pragma unroll
            for (int i = 0; i < N; ++i) buffer_store1s(e.y, off[i], so[i], v[i]);
#pragma unroll
            for (int i = 0; i < N; ++i) buffer_store1s(e.y2, off[i], so[i], act(v[i], p.act_slope) - sv[i]);
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) buffer_store1s(e.y, off[i], so[i], act(v[i], p.act_slope) - sv[i]);
        }
    } else if (p.y_act) {
        // raw tensor for residual consumers + activated twin for conv consumers
#pragma unroll
        for (int i = 0; i < N; ++i) buffer_store1s(e.y, off[i], so[i], v[i]);
#pragma unroll
        for (int i = 0; i < N; ++i) buffer_store1s(e.y2, off[i], so[i], act(v[i], p.act_slope));
    } else {
        if (p.act_slope != 1.f) {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = act(v[i], p.act_slope);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) buffer_store1s(e.y, off[i], so[i], v[i]);
    }
}

template <int N>
__device__ __forceinline__ void epilogue_store(const ConvParams& p, const EpilogueRsrc& e,
                                               const RowInfo<N>& ri, const int (&m)[N], int q,
                                               float (&v)[N]) {
    unsigned off[N], so[N];
    epilogue_offsets<N>(p, ri, m, q, off);
#pragma unroll
    for (int i = 0; i < N; ++i) so[i] = 0u;
    epilogue_finish<N>(p, e, ri.bias, off, so, v);
}

// Plain conv (ups == 1) whose row tile lies entirely inside the output: row r of the thread is
// mlane + rc[r] with rc compile-time, so every element of the batch shares ONE vector offset
// ((mlane*Tout + q)*4, or the out-of-range marker past the last column) and differs only in
// the wave-uniform rc*Tout*4, which rides in the instruction's scalar offset: no per-row
// offset registers, no per-row adds.  (Tout through opaque_uniform so that the scalar products
// are formed here, not hoisted and then spilled.)
template <int N, typename F>
__device__ __forceinline__ void epilogue_store_affine(const ConvParams& p, const EpilogueRsrc& e,
                                                      const float (&bias)[N], int mlane, int reg0,
                                                      int q, float (&v)[N]) {
    const unsigned t4 = (unsigned)opaque_uniform(p.Tout) * 4u;
    const unsigned voff = q < p.Tq ? (unsigned)(mlane * p.Tout + q) * 4u : kOutOfRange;
    unsigned off[N], so[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        off[i] = voff;
        so[i] = (unsigned)F::row(reg0 + i, 0) * t4;
    }
    epilogue_finish<N>(p, e, bias, off, so, v);
}

// XCD-aware block order: the dispatcher places linear block id b on XCD b % 8,
// so give each XCD a contiguous run of work (blocks that share an input tile
// because Cout > M_T, and neighbours in time that share halo columns, then
// meet in one L2) -- speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, within = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + within;
}

// an LDS address the optimiser must treat as a fresh value (a 32-bit address-space-3 pointer, so loads through
// it stay ds_read with immediate offsets; laundering a generic pointer would turn them into flat loads)
typedef __attribute__((address_space(3))) const float LdsCF;
__device__ __forceinline__ LdsCF* lds_opaque(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned v = __builtin_bit_cast(unsigned, (LdsCF*)p);
    asm volatile("" : "+v"(v));
    return __builtin_bit_cast(LdsCF*, v);
#else
    return (LdsCF*)p;   // host pass: never executed
#endif
}

// MFMA shape traits: MF = 32 -> v_mfma_f32_32x32x2_f32 (K step 2, 16 acc regs),
//                    MF = 16 -> v_mfma_f32_16x16x4_f32 (K step 4,  4 acc regs).
template <int MF>
struct Frag;
template <>
struct Frag<32> {
    typedef f32x16 acc_t;
    static constexpr int KS = 2, REGS = 16;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    static __device__ __forceinline__ int row(int reg, int lane) {
        return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    }
};
template <>
struct Frag<16> {
    typedef f32x4 acc_t;
    static constexpr int KS = 4, REGS = 4;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D map: col = lane & 15, row = 4*(lane >> 4) + reg
    static __device__ __forceinline__ int row(int reg, int lane) { return 4 * (lane >> 4) + reg; }
};

// Software-pipelined K loop of the conv kernels: n_it iterations of (one MFMA K step: KS channels x KT taps),
// A operands at pa[tap * tap_a], B operands at pb[tap * tap_b + r * MF]; pa / pb advance by step_a / step_b.
// An iteration is cut in two halves of taps; the operands of a half are read from LDS while the MFMAs of the
// PREVIOUS half run -- the second half of this iteration behind its first, the first half of the next
// iteration behind this second.  Left alone, hipcc issues all reads of an iteration at its top and the wave
// exposes one LDS latency per iteration (every 3-11 MFMAs); sched_barrier pins the order, the waitcnt pass
// still derives the exact lgkmcnt per use.  Same register count as the plain loop.  The running pointers are
// opaque (one register each, one add per iteration): otherwise their lane part and running part are kept
// apart and re-added in front of every group of reads.
template <int MF, int KT, int NR>
__device__ __forceinline__ void mma_pipelined(const float* pa0, const float* pb0, int tap_a, int tap_b, int step_a,
                                              int step_b, int n_it, typename Frag<MF>::acc_t (&acc)[NR]) {
    typedef Frag<MF> F;
    constexpr int H1 = (KT + 1) / 2, H2 = KT - H1;
    if (n_it <= 0) return;
    LdsCF* pa = lds_opaque(pa0);
    LdsCF* pb = lds_opaque(pb0);
    float a1[H1], b1[H1][NR], a2[H2 > 0 ? H2 : 1], b2[H2 > 0 ? H2 : 1][NR];
    auto load1 = [&]() {
#pragma unroll
        for (int t = 0; t < H1; ++t) {
            a1[t] = pa[t * tap_a];
#pragma unroll
            for (int r = 0; r < NR; ++r) b1[t][r] = pb[t * tap_b + r * MF];
        }
    };
    auto load2 = [&]() {
#pragma unroll
        for (int t = 0; t < H2; ++t) {
            a2[t] = pa[(H1 + t) * tap_a];
#pragma unroll
            for (int r = 0; r < NR; ++r) b2[t][r] = pb[(H1 + t) * tap_b + r * MF];
        }
    };
    auto mma1 = [&]() {
#pragma unroll
        for (int t = 0; t < H1; ++t)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = F::mfma(a1[t], b1[t][r], acc[r]);
    };
    auto mma2 = [&]() {
#pragma unroll
        for (int t = 0; t < H2; ++t)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] = F::mfma(a2[t], b2[t][r], acc[r]);
    };
    load1();
    for (int it = 0; it + 1 < n_it; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        load2();
        __builtin_amdgcn_sched_barrier(0);
        mma1();
        __builtin_amdgcn_sched_barrier(0);
        pa += step_a;
        pb += step_b;
        load1();
        __builtin_amdgcn_sched_barrier(0);
        mma2();
    }
    __builtin_amdgcn_sched_barrier(0);
    load2();
    __builtin_amdgcn_sched_barrier(0);
    mma1();
    mma2();
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate);
// n <= 2 * (kMaxDmaX + kMaxDmaW).  Larger or unexpected values wait for everything.
__device__ __forceinline__ void wait_vmcnt(int n) {
#define FV_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        FV_W(1) FV_W(2) FV_W(3) FV_W(4) FV_W(5) FV_W(6) FV_W(7) FV_W(8) FV_W(9) FV_W(10) FV_W(11) FV_W(12)
        FV_W(13) FV_W(14) FV_W(15) FV_W(16) FV_W(17) FV_W(18) FV_W(19) FV_W(20) FV_W(21) FV_W(22) FV_W(23)
        FV_W(24) FV_W(25) FV_W(26) FV_W(27) FV_W(28)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef FV_W
}

// ---------------------------------------------------------------------------
// The kernel.  Block tile (MF*WM) x (MF*NR*WN); WK wave groups split K.
// KT > 0 / DIL > 0 fix the tap count / dilation at compile time; ACT enables the
// read-time input activation (stand-alone operator calls only).
// ---------------------------------------------------------------------------
template <int MF, int WM, int WN, int WK, int NR, int KT, int DIL, bool ACT, bool SLOW>
__device__ __forceinline__ void conv_body(const ConvParams& p, const int block_x, const int grid_x,
                                          const int b) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr int NW = WM * WN * WK;
    constexpr int NT = 64 * NW;
    constexpr int M_T = MF * WM;
    constexpr int N_T = MF * NR * WN;
    constexpr int EN = F::REGS < 8 ? F::REGS : 8;   // epilogue batch
    constexpr int EH = F::REGS / EN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int k = KT > 0 ? KT : p.k;
    const int dil = DIL > 0 ? DIL : p.dil;
    float* const xs0 = smem;                      // NS x p.xbuf floats, then NS x p.wbuf (see the ring below)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int lm = lane & (MF - 1), kq = lane / MF;   // position inside the MFMA operand
    const int wk = wave / (WM * WN);
    const int wave_m = (wave / WN) % WM, wave_n = wave % WN;

    // this block's run of time tiles [tile_lo, tile_hi) for its m tile
    const int m_tiles = p.m_tiles;
    const int lin = xcd_remap(block_x, grid_x);
    const int mt = lin % m_tiles, run = lin / m_tiles;
    const int tile_lo = run * p.tiles_per_run;
    const int tile_hi = min(tile_lo + p.tiles_per_run, p.n_tiles);
    const int m0 = mt * M_T;
    const int nchunks = p.nchunks;
    if (tile_hi <= tile_lo) return;

    // Two-source convs (p.x2: the K range is the concatenation of two tensors) exist for 1-tap
    // kernels only, which run in the KT <= 1 instantiations; every other instantiation is
    // compiled without the second descriptor.
    constexpr bool TWO = KT <= 1;
    const int cin_a = TWO ? p.Cin1 : p.Cin;
    const float* const xb = p.x + (size_t)b * cin_a * (size_t)p.Tin;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)cin_a * (unsigned)p.Tin * 4u);
    const float* const xb2 = (TWO && p.x2) ? p.x2 + (size_t)b * (p.Cin - cin_a) * (size_t)p.Tin : xb;
    const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(xb2, (unsigned)(TWO && p.x2 ? p.Cin - cin_a : cin_a) * (unsigned)p.Tin * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.wp, (unsigned)p.Cin * (unsigned)p.k * (unsigned)p.Mpad * 4u);
    DmaPlan dp;
    dma_plan<NW, M_T>(p, dp, m0, wave, lane);
    // stage the input window of channels [ci0, ci0 + ci_chunk) (a chunk never straddles Cin1: host-checked)
    auto stage = [&](float* xs, int ci0, int tA) {
        if constexpr (TWO) {
            if (ci0 >= cin_a) {
                stage_x<NW, NT, SLOW>(p, dp, rx2, xs, xb2, p.Cin - cin_a, ci0 - cin_a, tA, wave, lane, tid);
                return;
            }
        }
        stage_x<NW, NT, SLOW>(p, dp, rx, xs, xb, cin_a, ci0, tA, wave, lane, tid);
    };
    // epilogue addressing: affine (see epilogue_store_affine) unless this is a transposed conv
    // or the row tile sticks out of the output rows
    const bool affine = p.ups == 1 && m0 + M_T <= p.M && !(p.dbg & 8);   // FV_DBG=8: general path (A/B)
    const int mlane = m0 + wave_m * MF + F::row(0, lane);
    RowInfo<EN> ri[EH];
    if (affine) {
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.bias ? p.bias : p.wp, p.bias ? (unsigned)p.Cout * 4u : 0u);
#pragma unroll
        for (int h = 0; h < EH; ++h) {
            ri[h].short_mask = 0u;
#pragma unroll
            for (int i = 0; i < EN; ++i) {
                ri[h].off[i] = 0u;
                ri[h].bias[i] = buffer_load1s(rb, (unsigned)mlane * 4u, (unsigned)F::row(h * EN + i, 0) * 4u);
            }
        }
    } else {
#pragma unroll
        for (int h = 0; h < EH; ++h) {
            int mm[EN];
#pragma unroll
            for (int i = 0; i < EN; ++i) mm[i] = m0 + wave_m * MF + F::row(h * EN + i, lane);
            row_info<EN>(p, mm, ri[h]);
        }
    }
    // the tile's window start is t0 - pad; with N_T a multiple of 4 only pad sets the phase.
    // A phase-major transposed conv shifts the window per row tile (= per output phase).
    const int pad_t = p.phase_major ? p.k - 1 - ((m0 / p.Cout + p.pad_orig) / p.ups) : p.pad;
    const int aoff = (((-pad_t) % 4) + 4) % 4;

    // ---- software pipeline over this block's stages s = (tile, channel chunk) -----------------
    // LDS holds NS stage buffers; stage s lives in buffer s % NS and its DMA is issued NS-1
    // stages ahead (across tile boundaries too).  Per stage: wait for OWN DMA instructions of
    // stage s (counted vmcnt: the later stages' stay in flight), barrier (everyone's landed,
    // and everyone is done reading buffer (s-1) % NS), issue stage s+NS-1 into that buffer,
    // then the matrix work of stage s.  NS = FV_RING (2: one stage ahead; deeper rings were
    // measured slower, see FV_RING).  The SLOW / ACT variants stage some tiles synchronously (no
    // fixed instruction count per stage): they always use NS = 2 and full waits.
    constexpr int NS = kRingStages(SLOW, ACT);
    const int total = (tile_hi - tile_lo) * nchunks;
    // this wave's DMA instructions per stage
    const int n_inst = wave_share<NW>(p.nx_inst, wave) + (nchunks > 1 ? wave_share<NW>(p.nw_inst, wave) : 0);
    float* const ws_base = smem + NS * p.xbuf;
    int it = tile_lo, ic = 0, ibuf = 0, issued = 0;   // next stage to issue: (tile, chunk), its buffer
    const int issue_limit = (p.dbg & 2) ? min(total, NS - 1) : total;
    auto issue = [&]() {
        const int tA = it * N_T - pad_t - aoff;
        stage(xs0 + ibuf * p.xbuf, ic * p.ci_chunk, tA);
        if (nchunks > 1) dma_w<NW, M_T>(p, dp, rw, ws_base + ibuf * p.wbuf, ic * p.ci_chunk, wave);
        if (++ic == nchunks) { ic = 0; ++it; }
        if (++ibuf == NS) ibuf = 0;
        ++issued;
    };
    // with a single chunk the weights never change: staged once, into buffer 0
    if (nchunks == 1) dma_w<NW, M_T>(p, dp, rw, ws_base, 0, wave);
    for (int i = 0; i < NS - 1 && issued < issue_limit; ++i) issue();

    int tile = tile_lo, chunk = 0, cur = 0;
    // Stores and loads share vmcnt but retire out of order with respect to each other, so no
    // counted wait is valid while a tile's stores are in flight -- and draining them costs ~2 us.
    // A block that goes on to another tile therefore makes sure of its NEXT stage before it
    // stores (that DMA was issued a whole stage of matrix work earlier), and skips the wait at
    // the top of that stage.
    bool landed = false;
    acc_t acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < F::REGS; ++i) acc[r][i] = 0.f;
    for (int s = 0; s < total; ++s) {
        // ---- stage s has landed (own wave: vmcnt; others: barrier) ----
        if (!landed) {
            if (NS == 2) wait_vmcnt(0);
            else wait_vmcnt((issued - s - 1) * n_inst);
        }
        landed = false;
        lds_barrier();
        if (issued < issue_limit) issue();
        const bool last_chunk = chunk == nchunks - 1;
        // ---- matrix work on buffer cur ----
        {
            const float* wsA = ws_base + (nchunks > 1 ? cur * p.wbuf : 0) + wave_m * MF + lm + kq * (k * M_T);
            const float* xsB = xs0 + cur * p.xbuf + aoff + wave_n * (MF * NR) + lm + kq * p.xw;
            const float slope = p.pre_slope;
            const int cend = (p.dbg & 4) ? 0 : p.ci_chunk;
            for (int c = wk * F::KS; c < cend; c += F::KS * WK) {
                const float* pa = wsA + c * (k * M_T);
                const float* pb = xsB + c * p.xw;
                if constexpr (KT > 0) {
#pragma unroll
                    for (int tap = 0; tap < KT; ++tap) {
                        const float a = pa[tap * M_T];
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            float bv = pb[tap * dil + r * MF];
                            if constexpr (ACT) bv = act(bv, slope);
                            acc[r] = F::mfma(a, bv, acc[r]);
                        }
                    }
                } else {
                    for (int tap = 0; tap < k; ++tap) {
                        const float a = pa[tap * M_T];
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            float bv = pb[tap * dil + r * MF];
                            if constexpr (ACT) bv = act(bv, slope);
                            acc[r] = F::mfma(a, bv, acc[r]);
                        }
                    }
                }
            }
        }
        if (last_chunk) {
            // ---- tile finished: (split-K reduce and) fused epilogue ----
            if constexpr (WK > 1) {
                float* red = smem + p.red_off;
                if (wk > 0) {
                    float* dst = red + ((wk - 1) * (WM * WN) + wave_m * WN + wave_n) *
                                           (NR * F::REGS * 64) + lane;
#pragma unroll
                    for (int r = 0; r < NR; ++r)
#pragma unroll
                        for (int i = 0; i < F::REGS; ++i) dst[(r * F::REGS + i) * 64] = acc[r][i];
                }
                lds_barrier();
                if (wk == 0) {
#pragma unroll
                    for (int g = 1; g < WK; ++g) {
                        const float* src = red + ((g - 1) * (WM * WN) + wave_m * WN + wave_n) *
                                                     (NR * F::REGS * 64) + lane;
#pragma unroll
                        for (int r = 0; r < NR; ++r)
#pragma unroll
                            for (int i = 0; i < F::REGS; ++i) acc[r][i] += src[(r * F::REGS + i) * 64];
                    }
                }
            }
            if (s + 1 < total) {
                wait_vmcnt(0);   // the next stage's DMA; nothing else of this wave is in flight
                landed = true;
            }
            if ((WK == 1 || wk == 0) && !(p.dbg & 1)) {
                const EpilogueRsrc ersrc = epilogue_rsrc(p, b);   // built here: no SGPRs held across the MFMA loop
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int q = tile * N_T + wave_n * (MF * NR) + r * MF + lm;
#pragma unroll
                    for (int h = 0; h < EH; ++h) {
                        float vv[EN];
#pragma unroll
                        for (int i = 0; i < EN; ++i) vv[i] = acc[r][h * EN + i];
                        if (affine) {
                            epilogue_store_affine<EN, F>(p, ersrc, ri[h].bias, mlane, h * EN, q, vv);
                        } else {
                            int mm[EN];
#pragma unroll
                            for (int i = 0; i < EN; ++i) mm[i] = m0 + wave_m * MF + F::row(h * EN + i, lane);
                            epilogue_store<EN>(p, ersrc, ri[h], mm, q, vv);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int i = 0; i < F::REGS; ++i) acc[r][i] = 0.f;
            chunk = 0;
            ++tile;
        } else {
            ++chunk;
        }
        if (++cur == NS) cur = 0;
    }
}

// At least FV_MIN_WAVES waves per SIMD: caps the VGPR budget (512 / waves per SIMD) so that as
// many blocks as the LDS budget allows stay resident on a CU.  The plain shapes need 123-124
// VGPRs (4 waves); the split-K shapes would spill there and keep 3, the two-accumulator
// 32x32 shapes (> 168 VGPRs) keep 2.
#ifndef FV_MIN_WAVES
#define FV_MIN_WAVES 4
#endif
#define FV_SUM3_WAVES __attribute__((amdgpu_waves_per_eu(FV_MIN_WAVES)))
#define FV_WAVES_ATTR \
    __attribute__((amdgpu_waves_per_eu((MF == 32 && NR == 2) ? 2 : ((WK > 1 && FV_MIN_WAVES > 3) ? 3 : FV_MIN_WAVES))))

template <int MF, int WM, int WN, int WK, int NR, int KT, int DIL, bool ACT, bool SLOW>
__global__ __launch_bounds__(64 * WM * WN * WK) FV_WAVES_ATTR void conv_mfma_kernel(ConvParams p) {
    conv_body<MF, WM, WN, WK, NR, KT, DIL, ACT, SLOW>(p, blockIdx.x, gridDim.x, blockIdx.y);
}

// Grouped launch: the convolutions at the same position of the three ResBlocks of
// an MRF stage (kernel sizes 11 / 7 / 3, everything else equal: hifigan.py:97-103,
// modules.py:223-230) are independent, and at batch 1 none of them fills 256 CUs
// alone.  One launch runs all three -- blockIdx.z picks the problem, largest
// kernel first -- so the block scheduler balances them, with one kernel boundary
// instead of three and no cross-stream hand-offs.
template <int MF, int WM, int WN, int WK, int NR, int DIL>
__global__ __launch_bounds__(64 * WM * WN * WK) FV_WAVES_ATTR void conv_group3_kernel(GroupParams gp) {
    const int g = blockIdx.z;
    if ((int)blockIdx.x >= gp.grid_x[g]) return;
    if (g == 0) conv_body<MF, WM, WN, WK, NR, 11, DIL, false, false>(gp.p[0], blockIdx.x, gp.grid_x[0], blockIdx.y);
    else if (g == 1) conv_body<MF, WM, WN, WK, NR, 7, DIL, false, false>(gp.p[1], blockIdx.x, gp.grid_x[1], blockIdx.y);
    else conv_body<MF, WM, WN, WK, NR, 3, DIL, false, false>(gp.p[2], blockIdx.x, gp.grid_x[2], blockIdx.y);
}

// ---------------------------------------------------------------------------
// MRF merge in one launch: the LAST convs of the three ResBlocks of a stage (11 / 7 / 3 taps,
// undilated, modules.py:226-229) all add into the same output tile,
//     y = act( ( sum_j ( conv_j(mid_j) + b_j + cur_j ) ) / 3 )            (hifigan.py:97-103)
// so one block runs the three K loops back to back into ONE accumulator and stores once:
// no r_1 / r_2 tensors (4 of the 11 tensor passes of the separate form) and no third launch.
// The sum is then formed inside the fp32 accumulator instead of as ((r0 + r1) + r2): same
// value up to fp32 rounding of the additions (<= 2e-7 relative; FV_MRF_FINAL=carrier keeps
// the reference's association exactly).  Stages of the three members form one linear DMA
// pipeline (two LDS buffers, one stage ahead, across member boundaries); the staging is the
// SLOW-capable one, so unaligned sequence lengths need no other variant.
// sp.p[0] carries y / y_act / out_div / act_slope / post and the SUMMED bias; p[j].res are
// the three residual inputs.
// ---------------------------------------------------------------------------
template <int MF, int KT, int NR>
__device__ __forceinline__ void sum3_mma(const float* wsA, const float* xsB, int cend, int xw,
                                         typename Frag<MF>::acc_t (&acc)[NR]) {
    typedef Frag<MF> F;
    constexpr int M_T = MF;   // WM == 1 in the shapes this kernel is built for
    mma_pipelined<MF, KT, NR>(wsA, xsB, M_T, 1, F::KS * (KT * M_T), F::KS * xw, (cend + F::KS - 1) / F::KS, acc);
}

template <int MF, int WN, int NR>
__global__ __launch_bounds__(64 * WN) FV_SUM3_WAVES void conv_sum3_kernel(Sum3Params sp) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr int NW = WN, NT = 64 * NW, M_T = MF, N_T = MF * NR * WN;
    constexpr int EN = F::REGS < 8 ? F::REGS : 8, EH = F::REGS / EN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvParams& p0 = sp.p[0];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & (MF - 1), kq = lane / MF;
    const int wave_n = wave;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = lin % p0.m_tiles, run = lin / p0.m_tiles;
    const int tile_lo = run * p0.tiles_per_run;
    const int tile_hi = min(tile_lo + p0.tiles_per_run, p0.n_tiles);
    const int m0 = mt * M_T;
    if (tile_hi <= tile_lo) return;
    float* const xs0 = smem;
    float* const ws0 = smem + 2 * sp.xbuf_max;

    // epilogue addressing: always the affine form (one vector offset per batch, the per-row part
    // in the scalar offset operand) -- the host only selects this kernel when the row tiles lie
    // entirely inside the output rows (Cout a multiple of the tile height)
    const int mlane = m0 + F::row(0, lane);
    float bias[EH][EN];
    {
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(p0.bias ? p0.bias : p0.wp, p0.bias ? (unsigned)p0.Cout * 4u : 0u);
#pragma unroll
        for (int h = 0; h < EH; ++h)
#pragma unroll
            for (int i = 0; i < EN; ++i)
                bias[h][i] = buffer_load1s(rb, (unsigned)mlane * 4u, (unsigned)F::row(h * EN + i, 0) * 4u);
    }

    const int total = sp.p[0].nchunks + sp.p[1].nchunks + sp.p[2].nchunks;   // stages per tile
    DmaPlan dp;
    __amdgpu_buffer_rsrc_t rx = make_rsrc(p0.x, 0u), rw = make_rsrc(p0.wp, 0u);
    const float* xb = p0.x;
    int im = 0, ic = 0, ibuf = 0;   // next stage to issue: member, chunk, LDS buffer
    auto issue = [&](int tile) {
        const ConvParams& q = sp.p[im];
        if (ic == 0) {   // entering a member: its DMA offsets and descriptors
            dma_plan<NW, M_T>(q, dp, m0, wave, lane);
            xb = q.x + (size_t)b * q.Cin * (size_t)q.Tin;
            rx = make_rsrc(xb, (unsigned)q.Cin * (unsigned)q.Tin * 4u);
            rw = make_rsrc(q.wp, (unsigned)q.Cin * (unsigned)q.k * (unsigned)q.Mpad * 4u);
        }
        const int aoff = (((-q.pad) % 4) + 4) % 4;
        const int tA = tile * N_T - q.pad - aoff;
        stage_x<NW, NT, true>(q, dp, rx, xs0 + ibuf * sp.xbuf_max, xb, q.Cin, ic * q.ci_chunk, tA, wave, lane, tid);
        dma_w<NW, M_T>(q, dp, rw, ws0 + ibuf * sp.wbuf_max, ic * q.ci_chunk, wave);
        if (++ic == q.nchunks) {
            ic = 0;
            if (++im == 3) im = 0;
        }
        ibuf ^= 1;
    };

    int cur = 0;
    bool landed = false;
    issue(tile_lo);
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        acc_t acc[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int i = 0; i < F::REGS; ++i) acc[r][i] = 0.f;
        int cm = 0, cc = 0;
        for (int s = 0; s < total; ++s) {
            if (!landed) wait_vmcnt(0);   // stage s landed (own wave); the barrier: everyone's, and buffer cur^1 is free
            landed = false;
            lds_barrier();
            const bool more = s + 1 < total || tile + 1 < tile_hi;
            if (more) issue(s + 1 < total ? tile : tile + 1);
            const ConvParams& q = sp.p[cm];
            const int aoff = (((-q.pad) % 4) + 4) % 4;
            const float* wsA = ws0 + cur * sp.wbuf_max + lm + kq * (q.k * M_T);
            const float* xsB = xs0 + cur * sp.xbuf_max + aoff + wave_n * (MF * NR) + lm + kq * q.xw;
            if (cm == 0) sum3_mma<MF, 11, NR>(wsA, xsB, q.ci_chunk, q.xw, acc);
            else if (cm == 1) sum3_mma<MF, 7, NR>(wsA, xsB, q.ci_chunk, q.xw, acc);
            else sum3_mma<MF, 3, NR>(wsA, xsB, q.ci_chunk, q.xw, acc);
            if (++cc == q.nchunks) {
                cc = 0;
                ++cm;
            }
            cur ^= 1;
        }
        // ---- epilogue: + the other two residuals, then the standard fused one on p[0] ----
        if (tile + 1 < tile_hi) {   // next tile's first stage before any store (see conv_body)
            wait_vmcnt(0);
            landed = true;
        }
        const EpilogueRsrc ersrc = epilogue_rsrc(p0, b);
        const size_t boff = (size_t)b * p0.Cout * (size_t)p0.Tout;
        const unsigned bytes = (unsigned)p0.Cout * (unsigned)p0.Tout * 4u;
        const __amdgpu_buffer_rsrc_t r1 = make_rsrc(sp.p[1].res + boff, bytes), r2 = make_rsrc(sp.p[2].res + boff, bytes);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int q = tile * N_T + wave_n * (MF * NR) + r * MF + lm;
#pragma unroll
            for (int h = 0; h < EH; ++h) {
                float vv[EN];
                unsigned off[EN], so[EN];
#pragma unroll
                for (int i = 0; i < EN; ++i) vv[i] = acc[r][h * EN + i];
                {
                    const unsigned t4 = (unsigned)opaque_uniform(p0.Tout) * 4u;
                    const unsigned voff = q < p0.Tq ? (unsigned)(mlane * p0.Tout + q) * 4u : kOutOfRange;
#pragma unroll
                    for (int i = 0; i < EN; ++i) {
                        off[i] = voff;
                        so[i] = (unsigned)F::row(h * EN + i, 0) * t4;
                    }
                }
                float e1[EN], e2[EN];
#pragma unroll
                for (int i = 0; i < EN; ++i) e1[i] = buffer_load1s(r1, off[i], so[i]);
#pragma unroll
                for (int i = 0; i < EN; ++i) e2[i] = buffer_load1s(r2, off[i], so[i]);
#pragma unroll
                for (int i = 0; i < EN; ++i) vv[i] += e1[i] + e2[i];
                epilogue_finish<EN>(p0, ersrc, bias[h], off, so, vv);
            }
        }
    }
}

template <int MF, int WN, int NR>
int launch_sum3_geom(const Sum3Params& sp, size_t lds, int grid_x, hipStream_t s) {
    auto kern = conv_sum3_kernel<MF, WN, NR>;
    if (lds > 64 * 1024)
        FV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid_x, sp.p[0].B), dim3(64 * WN), lds, s, sp);
    FV_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Narrow-output variant (Cout <= 4: conv_post hifigan.py:105 /
// multiband_hifigan.py:114, LastLayer modules.py:85-89).  HBM-bound
// (3.3 FLOP/B): one thread per output time step, weights broadcast from LDS.
// ---------------------------------------------------------------------------
// One staged chunk of input channels into the accumulators: acc[m] += w[ci][tap][m] * act(x[ci][t + tap dil]), channel-
// major, tap-minor.  K > 0: the tap count at compile time (7: conv_post, LastLayer) -- the K window reads and K weight
// reads of a channel are in flight together; with a run-time tap count every iteration waited out an LDS round trip
// (224 of them for MelGAN's 32-channel last layer: 9 us of a 23 us launch).  Same FMA order either way: same bits.
template <int MO, int K>
__device__ __forceinline__ void narrow_accumulate(float (&acc)[MO], const float* pb, const float* ws, int nci, int xw, int dil,
                                                  float slope, int k_rt = K) {
    if constexpr (K > 0) {
        for (int ci = 0; ci < nci; ++ci) {
            float xv[K];
#pragma unroll
            for (int tap = 0; tap < K; ++tap) xv[tap] = pb[ci * xw + tap * dil];
            const float* wrow = ws + ci * K * 16;
            float wv[K][MO];
#pragma unroll
            for (int tap = 0; tap < K; ++tap)
#pragma unroll
                for (int m = 0; m < MO; ++m) wv[tap][m] = wrow[tap * 16 + m];
#pragma unroll
            for (int tap = 0; tap < K; ++tap) {
                const float a = act(xv[tap], slope);
#pragma unroll
                for (int m = 0; m < MO; ++m) acc[m] = fmaf(wv[tap][m], a, acc[m]);
            }
        }
    } else {
        for (int ci = 0; ci < nci; ++ci) {
            for (int tap = 0; tap < k_rt; ++tap) {
                const float xv = act(pb[ci * xw + tap * dil], slope);
                const float* wrow = ws + (ci * k_rt + tap) * 16;
#pragma unroll
                for (int m = 0; m < MO; ++m) acc[m] = fmaf(wrow[m], xv, acc[m]);
            }
        }
    }
}

template <int MO>
__global__ __launch_bounds__(256) void conv_narrow_kernel(ConvParams p) {
    constexpr int N_T = 256, NT = 256, NW = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + p.xbuf;  // [ci_chunk*k][16]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int t0 = xcd_remap(blockIdx.x, gridDim.x) * N_T;
    const int k = p.k;
    const int aoff = (((-p.pad) % 4) + 4) % 4;
    const int tA = t0 - p.pad - aoff;
    float acc[MO];
#pragma unroll
    for (int m = 0; m < MO; ++m) acc[m] = 0.f;
    const __amdgpu_buffer_rsrc_t rx =
        make_rsrc(p.x + (size_t)b * p.Cin * (size_t)p.Tin, (unsigned)p.Cin * (unsigned)p.Tin * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.wp, (unsigned)p.Cin * (unsigned)p.k * (unsigned)p.Mpad * 4u);
    DmaPlan dp;
    dma_plan<NW, 16>(p, dp, 0, wave, lane);
    const float slope = p.pre_slope;
    for (int ci0 = 0; ci0 < p.Cin; ci0 += p.ci_chunk) {
        dma_w<NW, 16>(p, dp, rw, ws, ci0, wave);
        stage_x<NW, NT, true>(p, dp, rx, xs, p.x + (size_t)b * p.Cin * (size_t)p.Tin, p.Cin, ci0, tA, wave, lane, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* pb = xs + aoff + tid;
        if (k == 7) narrow_accumulate<MO, 7>(acc, pb, ws, p.ci_chunk, p.xw, p.dil, slope);
        else narrow_accumulate<MO, 0>(acc, pb, ws, p.ci_chunk, p.xw, p.dil, slope, k);
        __syncthreads();
    }
    const EpilogueRsrc ersrc = epilogue_rsrc(p, b);
    int mm[MO];
#pragma unroll
    for (int m = 0; m < MO; ++m) mm[m] = m;
    RowInfo<MO> ri;
    row_info<MO>(p, mm, ri);
    epilogue_store<MO>(p, ersrc, ri, mm, t0 + tid, acc);
}

// ---------------------------------------------------------------------------
// conv_post + tanh + PQMF synthesis in one launch (Multiband-HiFi-GAN's inference tail, multiband_hifigan.py:114-115,136
// with pqmf.py:121-135): the narrow conv above for MO = S sub-bands, whose activated tile goes to LDS instead of HBM,
// followed by the polyphase synthesis of pqmf.hip on that tile.  A block computes 256 sub-band samples and emits the
// S * 240 full-band samples whose 16 live taps per band lie inside them (8 sub-band samples of halo either side are
// recomputed by the neighbours): the [B, S, T'] sub-band tensor never exists.  Same FMA order as the two kernels it
// replaces: bit-identical to conv_narrow_kernel followed by pqmf_synthesis_kernel.
// ---------------------------------------------------------------------------

template <int MO>
__global__ __launch_bounds__(256) void conv_post_pqmf_kernel(ConvParams p, PqmfTail q) {
    constexpr int NT = 256, NW = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + p.xbuf;  // [ci_chunk*k][16]
    float* sb = smem + q.tail_off;          // [MO][256]: tanh(conv_post) of the block's sub-band window
    float* hs = sb + MO * 256;              // [MO][ntaps], scaled by MO (pqmf.hip)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int bx = xcd_remap(blockIdx.x, gridDim.x);
    const int t0 = bx * kPqmfAdvance - kPqmfHalo;        // first sub-band sample of the window (a multiple of 4)
    const int k = p.k;
    const int aoff = (((-p.pad) % 4) + 4) % 4;
    const int tA = t0 - p.pad - aoff;
    float acc[MO];
#pragma unroll
    for (int m = 0; m < MO; ++m) acc[m] = 0.f;
    const __amdgpu_buffer_rsrc_t rx =
        make_rsrc(p.x + (size_t)b * p.Cin * (size_t)p.Tin, (unsigned)p.Cin * (unsigned)p.Tin * 4u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.wp, (unsigned)p.Cin * (unsigned)p.k * (unsigned)p.Mpad * 4u);
    DmaPlan dp;
    dma_plan<NW, 16>(p, dp, 0, wave, lane);
    const float slope = p.pre_slope;
    for (int i = tid; i < MO * q.ntaps; i += NT) hs[i] = q.h[i] * (float)MO;
    for (int ci0 = 0; ci0 < p.Cin; ci0 += p.ci_chunk) {
        dma_w<NW, 16>(p, dp, rw, ws, ci0, wave);
        stage_x<NW, NT, true>(p, dp, rx, xs, p.x + (size_t)b * p.Cin * (size_t)p.Tin, p.Cin, ci0, tA, wave, lane, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* pb = xs + aoff + tid;
        if (k == 7) narrow_accumulate<MO, 7>(acc, pb, ws, p.ci_chunk, p.xw, p.dil, slope);
        else narrow_accumulate<MO, 0>(acc, pb, ws, p.ci_chunk, p.xw, p.dil, slope, k);
        __syncthreads();
    }
    {
        // the sub-bands of this window: + bias, post; zero outside [0, Tsub) (the synthesis filter sees a zero-padded signal)
        const int t = t0 + tid;
        const bool ok = t >= 0 && t < p.Tq;
#pragma unroll
        for (int m = 0; m < MO; ++m) {
            float v = acc[m] + (p.bias ? p.bias[m] : 0.f);
            if (p.post == FV_POST_TANH) v = tanhf(v);
            else if (p.post == FV_POST_RELU) v = fmaxf(v, 0.f);
            sb[m * 256 + tid] = ok ? v : 0.f;
        }
    }
    __syncthreads();
    const int Tsub = p.Tq, ntaps = q.ntaps, half = (ntaps - 1) / 2;
    const long long T = (long long)MO * Tsub;
#pragma unroll
    for (int r = 0; r < MO; ++r) {
        const int local = tid + NT * r;
        const long long n = (long long)MO * bx * kPqmfAdvance + local;
        if (local >= MO * kPqmfAdvance || n >= T) continue;
        int j0 = (int)((half - n) % MO);
        if (j0 < 0) j0 += MO;
        const int mbase = (int)((n + j0 - half) / MO);  // exact: divisible by construction
        float a = 0.f;
        for (int kb = 0; kb < MO; ++kb) {
            const float* xr = sb + kb * 256 - t0;
            const float* hr = hs + kb * ntaps;
            for (int j = j0, i = 0; j < ntaps; j += MO, ++i) {
                const int m = mbase + i;
                if (m >= 0 && m < Tsub) a = fmaf(hr[j], xr[m], a);
            }
        }
        const size_t o = (size_t)b * (size_t)T + (size_t)n;
        if (q.sub) {
            const float d = a - q.sub[(q.sub_batched ? (size_t)b * (size_t)T : 0) + (size_t)n];
            if (q.y2) {
                q.y[o] = a;
                q.y2[o] = d;
            } else {
                q.y[o] = d;
            }
        } else {
            q.y[o] = a;
        }
    }
}

// ---------------------------------------------------------------------------
// launch of one tile shape: picks the kernel variant (tap count, dilation, SLOW / ACT) --
// instantiated once per shape in conv_inst_*.hip so that the shapes compile in parallel
// ---------------------------------------------------------------------------
template <int MF, int WM, int WN, int WK, int NR>
int launch_geom(const ConvParams& p, size_t lds, int grid_x, hipStream_t s) {
    dim3 grid(grid_x, p.B), block(64 * WM * WN * WK);
#define FV_LAUNCH(KT, DIL, ACT, SLOW)                                                         \
    do {                                                                                      \
        auto kern = conv_mfma_kernel<MF, WM, WN, WK, NR, KT, DIL, ACT, SLOW>;                 \
        if (lds > 64 * 1024)                                                                  \
            FV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, p);                                     \
    } while (0)
#define FV_LAUNCH_DIL(KT)                                  \
    switch (p.dil) {                                       \
        case 1: FV_LAUNCH(KT, 1, false, false); break;     \
        case 3: FV_LAUNCH(KT, 3, false, false); break;     \
        case 5: FV_LAUNCH(KT, 5, false, false); break;     \
        default: FV_LAUNCH(KT, 0, false, false); break;    \
    }
    // SLOW variants carry the per-element staging path (reflection padding, rows
    // that are not 16-byte aligned); read-time activation exists only there too
    const bool slow = p.pad_mode == FV_PAD_REFLECT || !p.vec_ok;
    if (p.pre_slope != 1.f) {
        FV_LAUNCH(0, 0, true, true);
    } else if (slow) {
        // MelGAN / Basis-MelGAN: reflect-padded 3-tap convs with dilation 3^j (modules.py:351-357,
        // melgan.py:97-108) carry 60 % of their FLOPs; first / last layer: 7 taps, undilated
        switch (p.k) {
            case 3:
                switch (p.dil) {
                    case 1: FV_LAUNCH(3, 1, false, true); break;
                    case 3: FV_LAUNCH(3, 3, false, true); break;
                    case 9: FV_LAUNCH(3, 9, false, true); break;
                    default: FV_LAUNCH(3, 0, false, true); break;
                }
                break;
            case 7:
                if (p.dil == 1) FV_LAUNCH(7, 1, false, true);
                else FV_LAUNCH(7, 0, false, true);
                break;
            default: FV_LAUNCH(0, 0, false, true); break;
        }
    } else {
        switch (p.k) {
            case 1: FV_LAUNCH(1, 1, false, false); break;
            case 2:   // the phase-major transposed convs (k = 2*stride)
                if (p.dil == 1) FV_LAUNCH(2, 1, false, false);
                else FV_LAUNCH(0, 0, false, false);
                break;
            case 3: FV_LAUNCH_DIL(3); break;
            case 7: FV_LAUNCH_DIL(7); break;
            case 11: FV_LAUNCH_DIL(11); break;
            default: FV_LAUNCH(0, 0, false, false); break;
        }
    }
#undef FV_LAUNCH_DIL
#undef FV_LAUNCH
    FV_HIP(hipGetLastError());
    return 0;
}


template <int MF, int WM, int WN, int WK, int NR>
int launch_group_geom(const GroupParams& gp, size_t lds, int grid_x, int B, int dil, hipStream_t s) {
    dim3 grid(grid_x, B, 3), block(64 * WM * WN * WK);
#define FV_GROUP(DIL)                                                                          \
    do {                                                                                       \
        auto kern = conv_group3_kernel<MF, WM, WN, WK, NR, DIL>;                               \
        if (lds > 64 * 1024)                                                                   \
            FV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, gp);                                     \
    } while (0)
    switch (dil) {
        case 1: FV_GROUP(1); break;
        case 3: FV_GROUP(3); break;
        default: FV_GROUP(5); break;
    }
#undef FV_GROUP
    FV_HIP(hipGetLastError());
    return 0;
}


// Cout <= 4
int launch_narrow(const ConvParams& p, size_t lds, int grid_x, hipStream_t s);

}  // namespace fv
