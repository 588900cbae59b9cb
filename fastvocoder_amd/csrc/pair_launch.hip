// Host side of the fused ResBlock1-pair kernels (pair_kernels.hpp): validation, tile geometry, LDS
// layout, persistent grid size, launch.  The device code is instantiated in pair_inst_c16.hip /
// pair_inst_c32.hip (compiled in parallel); this file only sees declarations.
#include <stdlib.h>

#include "fv_internal.h"

namespace fv {

extern template int launch_pair_geom<1, 2, 8>(const PairParams&, int, size_t, hipStream_t);
extern template int launch_pair_geom<2, 1, 8>(const PairParams&, int, size_t, hipStream_t);
extern template int launch_pairh_geom<1, FV_PAIRH16_NF, FV_PAIRH16_NG>(const PairParams&, int, size_t, hipStream_t);
extern template int launch_pairh_geom<2, 1, FV_PAIRH32_NG>(const PairParams&, int, size_t, hipStream_t);

namespace {

int pair_row_stride(int n) {
    int r = (n + 3) / 4 * 4;
    while (r % 32 != 16) r += 4;
    return r;
}

int num_cus() {
    static int cus = 0;   // one device model per process: the CU count of the current device
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    return cus;
}

}  // namespace

// C = 16: 8 waves, two 16-column fragments each (256-column tiles, 2 blocks per CU);
// C = 32: 16 waves = 2 row halves x 8 column groups, one fragment each (128-column tiles, 1 block per CU:
// both 11-tap weight images are 90 KB of LDS)
PairShape pair_shape(int C, int k, int dil) {
    PairShape g = {};
    g.MH = C / 16;
    g.NF = C == 16 ? 2 : 1;
    g.NG = 8;

    g.NW = g.MH * g.NG;
    g.NM = 16 * g.NF * g.NG;
    const int p1 = (k - 1) * dil / 2, p2 = (k - 1) / 2;
    const int aoff = (4 - (p1 + p2) % 4) % 4;
    const int xwin = g.NM + (k - 1) * dil + aoff;
    g.XS = pair_row_stride((xwin + 3) / 4 * 4);
    g.NXI = (C * g.XS / 4 + 63) / 64;
    g.MS = g.NM + 16;
    g.WF = C * C * k;
    g.NOUT = (g.NM - (k - 1)) / 4 * 4;
    return g;
}

// ---- split-f16 pairs (pairh_kernels.hpp): run-time mirror of PairHGeom<> ----------------------------------
PairHShape pairh_shape(int C, int k, int dil) {
    PairHShape g = {};
    // C = 16: 8 waves x two fragments (256-column tiles, two blocks per CU: 4 waves per SIMD);
    // C = 32: 15 waves x one fragment, both row halves in the wave (240-column tiles, one block per CU: the two
    // 11-tap weight images are 88 KB; 16 waves would need 162 KB of LDS) -- fv_internal.h FV_PAIRH32_NG / FV_PAIRH16_*
    g.MH = C / 16;
    g.NF = C == 16 ? FV_PAIRH16_NF : 1;
    g.NG = C == 16 ? FV_PAIRH16_NG : FV_PAIRH32_NG;
    g.NM = 16 * g.NF * g.NG;
    const int tps = 32 / C;
    g.KS = (k + tps - 1) / tps;
    const int ktp = g.KS * tps;
    const int p1 = (k - 1) * dil / 2, p2 = (k - 1) / 2;
    const int aoff = (4 - (p1 + p2) % 4) % 4;
    const int xwin = g.NM + (ktp - 1) * dil + aoff;
    g.XROWS = (xwin + 3) / 4 * 4;
    g.WB = g.KS * g.MH * 2 * 1024;
    g.MROWS = g.NM + 16;
    g.XIMG = 4 * C * ((g.XROWS + 15) / 16 * 16);
    g.MIMG = 4 * C * g.MROWS;
    g.NOUT = (g.NM - (k - 1)) / 4 * 4;
    return g;
}

static int launch_pairs_split(PairParams p, int C, int dil, hipStream_t s) {
    if (p.sum) return fail(FV_ERR_UNSUPPORTED, "mrf sum: fp32 kernels only (split-f16 stages end in a pair with add1 / add2)");
    if (p.fold_w && (C != 16 || p.n_members != 1 || !p.fold_y || p.m[0].y_act))
        return fail(FV_ERR_UNSUPPORTED, "resblock pair with a folded output conv: one 16-channel member, no activated twin");
    int w_bytes = 0, img_bytes = 0, mid_bytes = 0;
    double flops = 0, bytes = 0;
    long long items = 0;
    for (int i = 0; i < p.n_members; ++i) {
        PairMember& mb = p.m[i];
        if (mb.k != 11 && mb.k != 7 && mb.k != 3) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %d taps (3, 7 or 11)", mb.k);
        if (!mb.x || !mb.w1 || !mb.w2 || (!mb.y && !p.fold_w)) return fail(FV_ERR_INVALID_ARG, "resblock pair: null tensor (member %d)", i);
        if (mb.add2 && !mb.add1) return fail(FV_ERR_INVALID_ARG, "resblock pair: add2 without add1 (member %d)", i);
        if ((reinterpret_cast<uintptr_t>(mb.x) | reinterpret_cast<uintptr_t>(mb.w1) | reinterpret_cast<uintptr_t>(mb.w2)) & 15)
            return fail(FV_ERR_UNSUPPORTED, "resblock pair: x / packed weights must be 16-byte aligned");
        const PairHShape g = pairh_shape(C, mb.k, dil);
        const int tstride = p.fold_w ? g.NOUT - 6 : g.NOUT;      // folded 7-tap output conv: 3 samples of halo either side
        mb.n_tiles = (p.T + tstride - 1) / tstride;
        // a tile costs its K steps (two convs, LDS-bandwidth bound) plus a part that does not depend on the taps
        // (loads, convert pass, epilogues, stores, barriers): measured per member alone (tools/pair_bench.py)
        // 0.8 us per step + 8 steps' worth at C = 16, 1.6 us per step + 6 steps' worth at C = 32 (240-column tiles)
        mb.cost = g.KS + (tuning().pairh_skel >= 0 ? tuning().pairh_skel : (C == 16 ? 8 : 6));
        mb.w_off = 0;
        if (2 * g.WB > w_bytes) w_bytes = 2 * g.WB;
        if (g.XIMG > img_bytes) img_bytes = g.XIMG;
        if (g.MIMG > mid_bytes) mid_bytes = g.MIMG;
        items += (long long)mb.n_tiles * p.B;
        flops += 2.0 * 2.0 * p.B * (double)C * C * mb.k * p.T;
        bytes += 4.0 * (2.0 * C * C * mb.k + (double)p.B * C * p.T * ((mb.y_act ? 3 : 2) + (mb.add1 ? 1 : 0) + (mb.add2 ? 1 : 0)));
        if (p.fold_w) {                // + the 16 -> 1, 7-tap conv; its output instead of the pair's
            flops += 2.0 * p.B * (double)C * 7 * p.T;
            bytes += 4.0 * ((double)p.B * p.T - (double)p.B * C * p.T);
        }
    }
    size_t floats = 0;
    p.x_off = 0;                       // [conv1 | conv2] packed weights of the member a block is working on
    floats += (size_t)w_bytes / 4;
    p.img_off = (int)floats;
    floats += (size_t)img_bytes / 4;
    p.mid_off = (int)floats;
    floats += (size_t)mid_bytes / 4;
    p.bias_off = (int)floats;
    floats += 4 * (size_t)C + 16;      // [b1 | b2 | inverse row prescales of conv1 | conv2 | scratch of the low-range guard]
    const size_t lds = floats * 4;
    if (lds > (C == 16 ? 80 : 160) * 1024) return fail(FV_ERR_UNSUPPORTED, "resblock pair, split-f16: %zu bytes of LDS", lds);
    // 15-16 waves per CU (4 per SIMD): two 8-wave blocks at C = 16, one 15-wave block at C = 32
    long long nblk = tuning().pair_blocks > 0 ? tuning().pair_blocks : (C == 16 ? 2LL : 1LL) * num_cus();
    if (nblk > items) nblk = items;
    p.nblk = (int)nblk;
    p.sched_on = 0;
    {
        long long n[3] = {0, 0, 0};
        for (int i = 0; i < p.n_members; ++i) n[i] = (long long)p.m[i].n_tiles * p.B;
        pair_cut_schedule(p, p.nblk, n);           // the blocks' shares as a table in the kernel arguments
    }
    p.dbg = tuning().pair_dbg;
    p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);
    profile_begin(s);
    const int rc = C == 16 ? launch_pairh_geom<1, FV_PAIRH16_NF, FV_PAIRH16_NG>(p, dil, lds, s) : launch_pairh_geom<2, 1, FV_PAIRH32_NG>(p, dil, lds, s);
    profile_end(s, C == 16 ? FV_KERNEL_PAIRH16 : FV_KERNEL_PAIRH32, flops, bytes);
    return rc;
}

int launch_pairs(PairParams p, int C, int dil, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0) return 0;
    if (C != 16 && C != 32) return fail(FV_ERR_UNSUPPORTED, "resblock pair: C = %d (16 or 32)", C);
    if (dil != 1 && dil != 3 && dil != 5) return fail(FV_ERR_UNSUPPORTED, "resblock pair: dilation %d (1, 3 or 5)", dil);
    if (p.n_members < 1 || p.n_members > 3) return fail(FV_ERR_INVALID_ARG, "resblock pair: %d members", p.n_members);
    if (p.T % 4 != 0)
        return fail(FV_ERR_UNSUPPORTED, "resblock pair: T = %d is not a multiple of 4 (rows must be 16-byte aligned); "
                    "use the conv1d ops", p.T);
    if ((double)C * p.T * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "resblock pair: one utterance's tensor (%d x %d floats) exceeds the 1 GiB "
                    "buffer-descriptor range; split the utterance", C, p.T);
    if (p.slope < 0.f || p.slope > 1.f || p.act_slope < 0.f || p.act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "resblock pair: activation slope outside [0, 1]");
    if (p.prec == FV_PAIR_SPLIT_F16) return launch_pairs_split(p, C, dil, s);
    if (p.prec != FV_PAIR_F32) return fail(FV_ERR_INVALID_ARG, "resblock pair: unknown arithmetic %d", p.prec);
    if (p.fold_w) return fail(FV_ERR_UNSUPPORTED, "resblock pair: a folded output conv exists with FV_PAIR_SPLIT_F16 only");
    for (int i = 0; i < p.n_members; ++i)
        if (p.m[i].add1 || p.m[i].add2)
            return fail(FV_ERR_UNSUPPORTED, "resblock pair: add1 / add2 exist with FV_PAIR_SPLIT_F16 only (fp32: fv_mrf_stage)");
    if (p.sum) {
        if (p.n_members != 3) return fail(FV_ERR_UNSUPPORTED, "mrf sum: needs the three members");
        for (int i = 0; i < 3; ++i)          // sort by taps, largest first: the kernel's member order
            for (int j = i + 1; j < 3; ++j)
                if (p.m[j].k > p.m[i].k) { PairMember t = p.m[i]; p.m[i] = p.m[j]; p.m[j] = t; }
        if (p.m[0].k != 11 || p.m[1].k != 7 || p.m[2].k != 3)
            return fail(FV_ERR_UNSUPPORTED, "mrf sum: needs the 11 / 7 / 3-tap trio (got %d, %d, %d)", p.m[0].k, p.m[1].k,
                        p.m[2].k);
        if (C != 16) return fail(FV_ERR_UNSUPPORTED, "mrf sum: C = %d (the three weight sets must fit in LDS: C = 16)", C);
    } else if (p.out_div != 1.f || p.post != FV_POST_NONE) {
        return fail(FV_ERR_INVALID_ARG, "resblock pair: out_div / post only exist in sum mode");
    }
    size_t floats = 0;
    int xs_floats = 0, mid_floats = 0, wmax = 0;
    double flops = 0, bytes = 0;
    long long items = 0;
    for (int i = 0; i < p.n_members; ++i) {
        PairMember& mb = p.m[i];
        if (mb.k != 11 && mb.k != 7 && mb.k != 3) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %d taps (3, 7 or 11)", mb.k);
        if (!mb.x || !mb.w1 || !mb.w2 || (!p.sum && !mb.y) || (p.sum && i == 0 && !mb.y))
            return fail(FV_ERR_INVALID_ARG, "resblock pair: null tensor (member %d)", i);
        if ((reinterpret_cast<uintptr_t>(mb.x) | reinterpret_cast<uintptr_t>(mb.w1) | reinterpret_cast<uintptr_t>(mb.w2)) & 15)
            return fail(FV_ERR_UNSUPPORTED, "resblock pair: x / packed weights must be 16-byte aligned");
        const PairShape g = pair_shape(C, mb.k, dil);
        const int nout = p.sum ? pair_shape(C, 11, dil).NOUT : g.NOUT;
        mb.n_tiles = (p.T + nout - 1) / nout;
        // a tile costs its MFMA time (proportional to the taps) plus a per-tile part that does not depend on
        // them (staging, activation pass, barriers, epilogue): measured 4.2k + 1.1k * taps cycles at C = 16,
        // 6k + 2.0k * taps at C = 32 (tools/pair_trace.py) -- in units of one tap's time
        mb.cost = mb.k + (tuning().pair_skel >= 0 ? tuning().pair_skel : (C == 16 ? 4 : 3));
        if (p.sum) {
            mb.w_off = (int)floats;
            floats += 2 * (size_t)g.WF;
        } else {
            mb.w_off = 0;
            if (2 * g.WF > wmax) wmax = 2 * g.WF;
        }
        if (g.NXI * 256 > xs_floats) xs_floats = g.NXI * 256;
        if (C * g.MS > mid_floats) mid_floats = C * g.MS;
        items += (long long)mb.n_tiles * p.B;
        flops += 2.0 * 2.0 * p.B * (double)C * C * mb.k * p.T;
        bytes += 4.0 * (2.0 * C * C * mb.k + (double)p.B * C * p.T * (p.sum ? 1 : (mb.y_act ? 3 : 2)));
    }
    if (p.sum) {
        p.n_out_sum = pair_shape(C, 11, dil).NOUT;
        items = (long long)p.m[0].n_tiles * p.B;
        bytes += 4.0 * p.B * (double)C * p.T * (p.m[0].y_act ? 2 : 1);
    } else {
        floats = (size_t)wmax;
    }
    p.x_off = (int)floats;
    floats += (size_t)xs_floats;
    p.mid_off = (int)floats;
    floats += (size_t)mid_floats;
    p.bias_off = (int)floats;
    floats += (size_t)(p.sum ? 6 : 2) * C;

    const size_t lds = floats * 4;
    if (lds > 160 * 1024) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %zu bytes of LDS", lds);
    const PairShape g0 = pair_shape(C, 11, dil);
    int per_cu = (int)(160 * 1024 / lds);
    if (per_cu * g0.NW > 16) per_cu = 16 / g0.NW;      // <= 128 VGPRs: 4 waves per SIMD
    if (per_cu < 1) per_cu = 1;
    long long nblk = tuning().pair_blocks > 0 ? tuning().pair_blocks : (long long)per_cu * num_cus();
    if (nblk > items) nblk = items;
    p.nblk = (int)nblk;
    p.sched_on = 0;
    p.guard = nullptr;                 // (exact fp32 products: no operand range to guard)
    p.dbg = tuning().pair_dbg;
    p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);

    profile_begin(s);
    int rc;
    if (C == 16) rc = launch_pair_geom<1, 2, 8>(p, dil, lds, s);
    else rc = launch_pair_geom<2, 1, 8>(p, dil, lds, s);
    profile_end(s, C == 16 ? FV_KERNEL_PAIR16 : FV_KERNEL_PAIR32, flops, bytes);
    return rc;
}

}  // namespace fv
