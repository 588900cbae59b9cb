// Host side of the conv kernels: validation, tile-shape choice, staging geometry, launch.
// The device code lives in conv_kernels.hpp and is instantiated per tile shape in
// conv_inst_*.hip (compiled in parallel); this file only sees declarations.
#include <stdlib.h>

#include "fv_internal.h"

namespace fv {

template <int MF, int WM, int WN, int WK, int NR>
int launch_geom(const ConvParams& p, size_t lds, int grid_x, hipStream_t s);
template <int MF, int WM, int WN, int WK, int NR>
int launch_group_geom(const GroupParams& gp, size_t lds, int grid_x, int B, int dil, hipStream_t s);
int launch_narrow(const ConvParams& p, size_t lds, int grid_x, hipStream_t s);
int launch_post_pqmf_kernel(const ConvParams& p, const PqmfTail& q, size_t lds, int grid_x, hipStream_t s);
template <int MF, int WN, int NR>
int launch_sum3_geom(const Sum3Params& sp, size_t lds, int grid_x, hipStream_t s);
extern template int launch_sum3_geom<32, 4, 1>(const Sum3Params&, size_t, int, hipStream_t);
extern template int launch_sum3_geom<16, 4, 2>(const Sum3Params&, size_t, int, hipStream_t);
#define FV_SHAPE_DECL(MF, WM, WN, WK, NR)                                                            \
    extern template int launch_geom<MF, WM, WN, WK, NR>(const ConvParams&, size_t, int, hipStream_t); \
    extern template int launch_group_geom<MF, WM, WN, WK, NR>(const GroupParams&, size_t, int, int, int, hipStream_t);
FV_SHAPE_DECL(16, 1, 4, 1, 2)
FV_SHAPE_DECL(16, 1, 2, 2, 2)
FV_SHAPE_DECL(32, 1, 4, 1, 1)
FV_SHAPE_DECL(32, 1, 2, 2, 1)
FV_SHAPE_DECL(32, 1, 1, 4, 1)
FV_SHAPE_DECL(32, 2, 2, 1, 1)
#undef FV_SHAPE_DECL

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
namespace {

struct Geometry {
    int mf, wm, wn, wk, nr;
    int m_t() const { return mf * wm; }
    int n_t() const { return mf * nr * wn; }
    int threads() const { return 64 * wm * wn * wk; }
};

// LDS image of the input tile: ncol4c float4 columns per channel row.  For the
// 16x16x4 MFMA the row stride is made = 16 (mod 32) floats so the two k-rows a
// half-wave reads sit on disjoint banks.
void plan_x_image(ConvParams& p, int n_t, bool mfma16) {
    const int halo = (p.k - 1) * p.dil;
    p.ncol4 = (n_t + halo + 3 + 3) / 4;
    int c = p.ncol4;
    if (mfma16) {
        while (c % 8 != 4) ++c;
    }
    p.ncol4c = c;
    p.xw = 4 * c;
    p.ncol4c_magic = (unsigned)(((1ull << 32) + c - 1) / c);
}

// Fill in the staging geometry for tile shape g; returns the dynamic LDS bytes
// or 0 when the shape cannot be staged within the LDS budget.
size_t plan_staging(ConvParams& p, const Geometry& g, int k_rows_target) {
    plan_x_image(p, g.n_t(), g.mf == 16);
    const int ks = (g.mf == 32 ? 2 : 4) * g.wk;     // ci granularity of one MFMA step x split
    const int cin_pad = round_up(p.Cin, ks);
    const int ns = kRingStages(p.pad_mode == FV_PAD_REFLECT || !p.vec_ok, p.pre_slope != 1.f);
    // stage size: about k_rows_target MFMA K-rows (ci_chunk*k), all ns ring buffers <= 39 KiB
    // so that 4 blocks stay resident per CU (160 KiB LDS; <= 128 VGPRs at 4 waves/SIMD); prefer chunks dividing Cin
    // (defaults from end-to-end sweeps on MI355X, tools/bench_sweep.sh)
    int best = 0;
    for (int c = ks; c <= cin_pad; c += ks) {
        const size_t per_buf = (size_t)round_up(c * p.ncol4c, 64) * 16 + (size_t)round_up(c * p.k * g.m_t() / 4, 64) * 16;
        const int nw = g.wm * g.wn * g.wk;
        const bool dma_ok = round_up(c * p.ncol4c, 64) / 64 <= kMaxDmaX * nw &&
                            round_up(c * p.k * g.m_t() / 4, 64) / 64 <= kMaxDmaW * nw;
        if (best && (!dma_ok || ns * per_buf > (size_t)tuning().lds_budget * 1024)) break;
        if (!dma_ok) return 0;
        // a stage of a two-source conv must not straddle the boundary between its tensors
        const bool src_ok = !p.x2 || p.Cin1 % c == 0;
        if (src_ok && (cin_pad % c == 0 || !best)) best = c;
        if (best == c && c * p.k >= k_rows_target && cin_pad % c == 0) break;
    }
    if (!best) return 0;
    p.ci_chunk = best;
    p.nchunks = (p.Cin + best - 1) / best;
    p.nx_inst = round_up(best * p.ncol4c, 64) / 64;
    p.nw_inst = round_up(best * p.k * g.m_t() / 4, 64) / 64;
    p.xbuf = round_up(best * p.ncol4c, 64) * 4;
    p.wbuf = round_up(best * p.k * g.m_t() / 4, 64) * 4;
    size_t floats = (size_t)ns * (p.xbuf + p.wbuf);
    p.red_off = (int)floats;
    if (g.wk > 1) floats += (size_t)(g.wk - 1) * g.wm * g.wn * g.nr * (g.mf == 32 ? 16 : 4) * 64;
    return floats * 4;
}

// The tile shapes built into the library, by id.
const Geometry kShapes[] = {
    {16, 1, 4, 1, 2},   // 0: 16 x 128
    {16, 1, 2, 2, 2},   // 1: 16 x 64,   K split 2
    {32, 1, 4, 1, 1},   // 2: 32 x 128
    {32, 1, 2, 2, 1},   // 3: 32 x 64,   K split 2
    {32, 1, 1, 4, 1},   // 4: 32 x 32,   K split 4
    {32, 2, 2, 1, 1},   // 5: 64 x 64
};   // (tried and dropped: 64x32 split-K, 32x128 with two accumulators per wave, 16x256 -- none faster)
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

int launch_shape(int id, const ConvParams& p, size_t lds, int grid_x, hipStream_t s) {
    switch (id) {
        case 0: return launch_geom<16, 1, 4, 1, 2>(p, lds, grid_x, s);
        case 1: return launch_geom<16, 1, 2, 2, 2>(p, lds, grid_x, s);
        case 2: return launch_geom<32, 1, 4, 1, 1>(p, lds, grid_x, s);
        case 3: return launch_geom<32, 1, 2, 2, 1>(p, lds, grid_x, s);
        case 4: return launch_geom<32, 1, 1, 4, 1>(p, lds, grid_x, s);
        default: return launch_geom<32, 2, 2, 1, 1>(p, lds, grid_x, s);
    }
}

}  // namespace

namespace {

struct LaunchInfo {
    int kind, shape, grid_x;
    size_t lds;
    double flops, bytes;
    bool narrow;
};

// Validate one conv, pick its tile shape and fill in the staging geometry.
int prepare_conv(ConvParams& p, LaunchInfo& li, int members = 1, int force_shape = -1) {
    if (p.k < 1 || p.dil < 1) return fail(FV_ERR_INVALID_ARG, "conv: k=%d dil=%d", p.k, p.dil);
    if (p.pad_mode == FV_PAD_REFLECT && p.pad >= p.Tin)
        return fail(FV_ERR_INVALID_ARG, "reflection pad %d needs an input longer than it (T=%d)",
                    p.pad, p.Tin);
    if (p.pre_slope < 0.f || p.pre_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "conv: input activation slope %g outside [0, 1]", p.pre_slope);
    if (p.x2 && (p.k != 1 || p.ups != 1 || p.pre_slope != 1.f || p.Cin1 <= 0 || p.Cin1 >= p.Cin || p.M <= 4))
        return fail(FV_ERR_UNSUPPORTED, "two-source conv: needs k = 1, no input activation, Cout > 4 "
                    "(k=%d Cin1=%d Cin=%d Cout=%d)", p.k, p.Cin1, p.Cin, p.Cout);
    if (!p.x2) p.Cin1 = p.Cin;
    if ((double)p.Cin * p.Tin * 4.0 >= 1073741824.0 || (double)p.Cin * p.k * p.Mpad * 4.0 >= 1073741824.0 ||
        (double)p.Cout * p.Tout * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "conv: one utterance's tensor (%d x %d or %d x %d floats) exceeds the "
                    "1 GiB buffer-descriptor range of the kernels; split the utterance", p.Cin, p.Tin,
                    p.Cout, p.Tout);
    p.vec_ok = (p.Tin % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) &&
               ((reinterpret_cast<uintptr_t>(p.x2) & 15) == 0);
    li.flops = p.alg_flops > 0 ? p.alg_flops : 2.0 * p.B * (double)p.M * p.Tq * p.Cin * p.k;
    li.bytes = 4.0 * ((double)p.B * p.Cin * p.Tin + (double)p.B * p.Cout * p.Tout *
                      (1 + (p.res != nullptr) + (p.acc_in != nullptr) + (p.acc_in2 != nullptr) +
                       (p.y_act != nullptr)) +
                      (double)p.Cin * p.k * p.M);
    p.dbg = tuning().conv_dbg;
    li.narrow = p.ups == 1 && p.M <= 4;
    if (li.narrow) {
        li.kind = FV_KERNEL_CONV_NARROW;
        plan_x_image(p, 256, false);
        int c = 1;
        while (c + 1 <= p.Cin && (size_t)(round_up((c + 1) * p.ncol4c, 64) + round_up((c + 1) * p.k * 4, 64)) * 16 <= 48 * 1024 &&
               round_up((c + 1) * p.ncol4c, 64) / 64 <= kMaxDmaX * 4 && round_up((c + 1) * p.k * 4, 64) / 64 <= kMaxDmaW * 4)
            ++c;
        p.nchunks = (p.Cin + c - 1) / c;
        p.nx_inst = round_up(c * p.ncol4c, 64) / 64;
        p.nw_inst = round_up(c * p.k * 4, 64) / 64;
        p.ci_chunk = c;
        p.xbuf = round_up(c * p.ncol4c, 64) * 4;
        p.wbuf = round_up(c * p.k * 4, 64) * 4;
        li.lds = (size_t)(p.xbuf + p.wbuf) * 4;
        li.grid_x = (p.Tq + 255) / 256;
        li.shape = -1;
        return 0;
    }
    // ---- tile shape: wide tiles when the utterance alone yields enough work
    // units, otherwise narrower tiles with the K range split over more waves.
    // The choice depends on per-utterance sizes only (never on B), so a
    // batch sharded over GPUs reproduces the single-GPU result bit for bit.
    const bool m16 = p.Mpad == 16;
    const bool m64 = !m16 && p.Mpad % 64 == 0;
    li.kind = m16 ? FV_KERNEL_CONV_MFMA16 : FV_KERNEL_CONV_MFMA32;
    auto units = [&](int id) {
        const Geometry& gg = kShapes[id];
        return (long)(p.Mpad / gg.m_t()) * ((p.Tq + gg.n_t() - 1) / gg.n_t());
    };
    // thresholds from per-layer and end-to-end sweeps on MI355X (tools/conv_bench.py, bench_sweep.sh).
    // ``members`` = convs sharing the launch (3 in a grouped MRF launch): that many times the
    // units, so the wide 32x128 tile -- least weight re-streaming per MFMA (64 B vs 128 B of
    // L2->LDS traffic per instruction for the 64-column shapes) -- already pays at C = 128, T = 8000.
    const int want = tuning().units;
    int shape;
    if (m16) shape = units(0) >= want ? 0 : 1;
    else if (units(2) * members >= want) shape = 2;
    else if (m64 && units(5) * members >= want) shape = 5;
    else if (units(3) * members >= 400) shape = 3;
    else shape = 4;
    // a phase-major transposed conv needs row tiles that hold one phase: m_t must divide Cout
    if (p.phase_major && p.Cout % kShapes[shape].m_t() != 0) shape = units(3) * members >= 400 ? 3 : 4;
    int force = m16 ? tuning().shape16 : (m64 ? tuning().shape64 : tuning().shape32);
    if (force_shape >= 0) force = force_shape;
    if (force >= 0 && force < kNumShapes && kShapes[force].mf == (m16 ? 16 : 32) &&
        p.Mpad % kShapes[force].m_t() == 0 && !(p.phase_major && p.Cout % kShapes[force].m_t() != 0))
        shape = force;
    const Geometry g = kShapes[shape];
    li.shape = shape;
    li.lds = plan_staging(p, g, tuning().krows);
    if (!li.lds) return fail(FV_ERR_UNSUPPORTED, "conv: k=%d dil=%d cannot be staged (window too wide)", p.k, p.dil);
    p.n_tiles = (p.Tq + g.n_t() - 1) / g.n_t();
    const int m_tiles = p.Mpad / g.m_t();
    // runs of consecutive time tiles per block: cap the grid
    const int cap = tuning().grid_cap;
    int runs = p.n_tiles;
    const long per_batch_cap = cap / ((long)p.B * m_tiles) > 0 ? cap / ((long)p.B * m_tiles) : 1;
    if (runs > per_batch_cap) runs = (int)per_batch_cap;
    p.m_tiles = m_tiles;
    p.tiles_per_run = (p.n_tiles + runs - 1) / runs;
    runs = (p.n_tiles + p.tiles_per_run - 1) / p.tiles_per_run;   // no empty runs
    li.grid_x = runs * m_tiles;
    return 0;
}

}  // namespace

int launch_conv(ConvParams p, hipStream_t s) {
    if (p.B <= 0 || p.Tq <= 0 || p.Cin <= 0 || p.M <= 0) return 0;
    LaunchInfo li;
    if (int rc = prepare_conv(p, li)) return rc;
    int rc = 0;
    profile_begin(s);
    if (li.narrow) {
        rc = launch_narrow(p, li.lds, li.grid_x, s);
    } else {
        rc = launch_shape(li.shape, p, li.lds, li.grid_x, s);
    }
    profile_end(s, li.kind, li.flops, li.bytes);
    return rc;
}

int launch_conv_post_pqmf(ConvParams p, const float* h, int ntaps, float* y, float* y2, const float* sub, int sub_batched,
                          hipStream_t s) {
    if (p.B <= 0 || p.Tq <= 0 || p.Cin <= 0) return 0;
    if (p.M != 4 || p.ups != 1 || ntaps != 63 || p.res || p.acc_in || p.y_act)
        return fail(FV_ERR_UNSUPPORTED, "conv_post + pqmf: 4 sub-bands, 63 taps, a plain conv in front (Cout=%d ntaps=%d)", p.M, ntaps);
    if ((double)p.M * p.Tq * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "conv_post + pqmf: the utterance exceeds the 1 GiB range of the kernels; split it");
    LaunchInfo li;
    if (int rc = prepare_conv(p, li)) return rc;
    if (!li.narrow) return fail(FV_ERR_UNSUPPORTED, "conv_post + pqmf: not a narrow conv");
    PqmfTail q = {h, y, y2, sub, sub_batched, ntaps, (int)(li.lds / 4)};
    const size_t lds = li.lds + (size_t)(4 * 256 + round_up(4 * ntaps, 4)) * 4;
    const int grid_x = (p.Tq + kPqmfAdvance - 1) / kPqmfAdvance;
    profile_begin(s);
    const int rc = launch_post_pqmf_kernel(p, q, lds, grid_x, s);
    profile_end(s, li.kind, li.flops + 2.0 * p.B * 4.0 * p.Tq * ntaps, li.bytes - 4.0 * p.B * 4.0 * p.Tq + 4.0 * p.B * 4.0 * p.Tq);
    return rc;
}

// The three last convs of an MRF stage summed in one accumulator (conv_sum3_kernel).
int launch_conv_sum3(ConvParams* ps, hipStream_t s) {
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i)       // sort by taps, largest first
        for (int j = i + 1; j < 3; ++j)
            if (ps[order[j]].k > ps[order[i]].k) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    Sum3Params sp;
    LaunchInfo li[3];
    const bool m16 = pad_rows(ps[0].M) == 16;
    for (int i = 0; i < 3; ++i) {
        ConvParams& q = ps[order[i]];
        const ConvParams& a = ps[order[0]];
        static const int want_k[3] = {11, 7, 3};
        if (q.k != want_k[i] || q.dil != 1 || q.ups != 1 || q.pad != (q.k - 1) / 2 || q.pad_mode != FV_PAD_ZERO ||
            q.pre_slope != 1.f || q.x2 || !q.res || q.Cin != a.Cin || q.M != a.M || q.Tin != a.Tin || q.B != a.B ||
            q.M <= 4 || (!m16 && pad_rows(q.M) % 32 != 0))
            return fail(FV_ERR_UNSUPPORTED, "conv_sum3: needs the 11/7/3-tap undilated same-shape trio with residuals "
                        "(member %d: k=%d dil=%d pad=%d Cin=%d Cout=%d)", i, q.k, q.dil, q.pad, q.Cin, q.M);
        const int want_shape = m16 ? 0 : 2;   // 16 x 128 / 32 x 128 (32 x 64 two-wave tiles were measured slower)
        if (int rc = prepare_conv(q, li[i], 3, want_shape)) return rc;
        if (li[i].narrow || li[i].shape != want_shape) return fail(FV_ERR_UNSUPPORTED, "conv_sum3: tile shape");
        sp.p[i] = q;
    }
    {   // the summed bias rides on whichever member the caller attached it to: move it to p[0]
        const float* bias = nullptr;
        for (int i = 0; i < 3; ++i)
            if (sp.p[i].bias) bias = sp.p[i].bias;
        for (int i = 0; i < 3; ++i) sp.p[i].bias = i == 0 ? bias : nullptr;
    }
    sp.xbuf_max = sp.wbuf_max = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < 3; ++i) {
        if (sp.p[i].xbuf > sp.xbuf_max) sp.xbuf_max = sp.p[i].xbuf;
        if (sp.p[i].wbuf > sp.wbuf_max) sp.wbuf_max = sp.p[i].wbuf;
        flops += li[i].flops;
        bytes += li[i].bytes;
    }
    // the three members stored no outputs of their own: only ONE output (+ twin) is written
    bytes -= 2.0 * 4.0 * sp.p[0].B * (double)sp.p[0].Cout * sp.p[0].Tout * (sp.p[0].y_act ? 2 : 1);
    const size_t lds = (size_t)2 * (sp.xbuf_max + sp.wbuf_max) * 4;
    profile_begin(s);
    const int rc = m16 ? launch_sum3_geom<16, 4, 2>(sp, lds, li[0].grid_x, s)
                       : launch_sum3_geom<32, 4, 1>(sp, lds, li[0].grid_x, s);
    profile_end(s, li[0].kind, flops, bytes);
    return rc;
}

// Three mutually independent convs in one launch when they are the (11, 7, 3)-tap
// members of an MRF position and agree on everything else; otherwise three launches.
int launch_conv_group(ConvParams* ps, int n, hipStream_t s) {
    // three members (11, 7, 3 taps) or the two large ones (11, 7): the kernel is the same, the
    // third problem just has an empty grid
    bool ok = (n == 3 || n == 2) && !tuning().no_group;
    LaunchInfo li[3];
    int order[3] = {0, 1, 2};
    if (ok) {
        for (int i = 0; i < n && ok; ++i) {
            if (ps[i].B <= 0 || ps[i].Tq <= 0) ok = false;
            else if (prepare_conv(ps[i], li[i], n)) ok = false;
        }
    }
    if (ok) {
        for (int i = 0; i < n; ++i)       // sort by taps, largest first
            for (int j = i + 1; j < n; ++j)
                if (ps[order[j]].k > ps[order[i]].k) { int t = order[i]; order[i] = order[j]; order[j] = t; }
        const ConvParams& a = ps[order[0]];
        ok = a.k == 11 && ps[order[1]].k == 7 && (n == 2 || ps[order[2]].k == 3) && !li[order[0]].narrow &&
             (a.dil == 1 || a.dil == 3 || a.dil == 5) && a.ups == 1;
        for (int i = 0; i < n && ok; ++i) {
            const ConvParams& q = ps[order[i]];
            const LaunchInfo& l = li[order[i]];
            ok = !l.narrow && l.shape == li[order[0]].shape && q.dil == a.dil && q.B == a.B &&
                 q.Cin == a.Cin && q.M == a.M && q.Tq == a.Tq && q.ups == 1 && q.pre_slope == 1.f &&
                 q.pad_mode == FV_PAD_ZERO && q.vec_ok && !q.x2;
        }
    }
    if (!ok) {
        for (int i = 0; i < n; ++i)
            if (int rc = launch_conv(ps[i], s)) return rc;
        return 0;
    }
    GroupParams gp;
    size_t lds = 0;
    int grid_x = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
        gp.p[i] = ps[order[i]];
        gp.grid_x[i] = li[order[i]].grid_x;
        if (li[order[i]].lds > lds) lds = li[order[i]].lds;
        if (gp.grid_x[i] > grid_x) grid_x = gp.grid_x[i];
        flops += li[order[i]].flops;
        bytes += li[order[i]].bytes;
    }
    for (int i = n; i < 3; ++i) {   // absent member: its blocks return at once
        gp.p[i] = gp.p[0];
        gp.grid_x[i] = 0;
    }
    profile_begin(s);
    int rc;
    const int B = gp.p[0].B, dil = gp.p[0].dil;
    switch (li[order[0]].shape) {
        case 0: rc = launch_group_geom<16, 1, 4, 1, 2>(gp, lds, grid_x, B, dil, s); break;
        case 1: rc = launch_group_geom<16, 1, 2, 2, 2>(gp, lds, grid_x, B, dil, s); break;
        case 2: rc = launch_group_geom<32, 1, 4, 1, 1>(gp, lds, grid_x, B, dil, s); break;
        case 3: rc = launch_group_geom<32, 1, 2, 2, 1>(gp, lds, grid_x, B, dil, s); break;
        case 4: rc = launch_group_geom<32, 1, 1, 4, 1>(gp, lds, grid_x, B, dil, s); break;
        default: rc = launch_group_geom<32, 2, 2, 1, 1>(gp, lds, grid_x, B, dil, s); break;
    }
    profile_end(s, li[order[0]].kind, flops, bytes);
    return rc;
}

}  // namespace fv
