// Host side of the split-f16 wide-channel conv kernels (convh_kernels.hpp): validation, tile geometry, LDS layout,
// persistent grid, launch.  Device code: convh_inst_c64.hip / convh_inst_c128.hip.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <vector>

#include "fv_internal.h"

namespace fv {

extern template int launch_convh_geom<4, 2>(const PairParams&, int, size_t, hipStream_t);
extern template int launch_convh_geom<2, 2>(const PairParams&, int, size_t, hipStream_t);


// ---- block schedule for launches with few items per block ----------------------------------------------------
// The kernels' own partition cuts the cost-weighted item sequence into nblk contiguous shares: with 1-3 items per
// block and items of three different costs, a share ends up anywhere between one cheap item and two expensive ones
// (HiFi-GAN light at batch 1, 128 channels: 126 items each of cost 24 / 16 / 8 on 256 blocks -- makespan 32 against a
// mean of 23.6).  Items of one member are interchangeable, so a greedy longest-processing-time-first assignment
// (most expensive member first, every item to the block that finishes it earliest; taking up a second member costs
// a pipeline refill) is computed once per shape on the host and handed to the kernel as per-block item ranges.
// [measured, MI355X, B = 1] the two-member launches at the end of a stage (11 + 7 taps: every block ends up with ONE
// member) 56.7 -> 47.7 us at 128 channels, 49.7 -> 47.0 at 64; three-member launches gain nothing (128 channels:
// a 7-tap + 3-tap block pays a member switch, a pipeline drain and refill, for what it saves) or lose (64 channels,
// 58 vs 55 us: more switches than the contiguous cut has), so they keep the contiguous cut (FV_SCHED=2: all).
// (three_members: the kernel wants its three-member launches scheduled too -- the fused 128-channel pairs, whose tiles
// are long enough for the balance to show: 191 -> 170 us over the four launches of a stage; Tuning::sched = 2: all)
void pair_schedule(PairParams& p, int nblk, bool three_members) {
    typedef std::array<unsigned, 2 * kSchedBlocks> Table;
    static std::mutex mu;
    static std::map<std::array<int, 9>, Table> cache;           // host memory only; a process sees a handful of shapes
    p.sched_on = 0;
    const Tuning& tn = tuning();
    if (tn.sched == 0) return;
    long long items = 0;
    for (int m = 0; m < p.n_members; ++m) items += p.m[m].n_items;
    if (p.n_members < 2 || nblk < 2 || nblk > kSchedBlocks || items > 6LL * nblk) return;
    if (p.n_members != 2 && tn.sched != 2 && !three_members) return;
    for (int m = 0; m < p.n_members; ++m)
        if (p.m[m].n_items > 2047) return;                       // 11-bit item numbers
    const int sw = tn.sched_switch;
    std::array<int, 9> key = {nblk, p.n_members, sw, 0, 0, 0, 0, 0, 0};
    for (int m = 0; m < p.n_members; ++m) {
        key[3 + 2 * m] = p.m[m].n_items;
        key[4 + 2 * m] = p.m[m].cost;
    }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (cache.size() >= 256) return;
        int order[3] = {0, 1, 2};
        for (int i = 0; i < p.n_members; ++i)
            for (int j = i + 1; j < p.n_members; ++j)
                if (p.m[order[j]].cost > p.m[order[i]].cost) std::swap(order[i], order[j]);
        std::vector<long long> load(nblk, 0);
        std::vector<int> cnt((size_t)nblk * 3, 0);
        for (int oi = 0; oi < p.n_members; ++oi) {
            const int m = order[oi], c = p.m[m].cost;
            for (int i = 0; i < p.m[m].n_items; ++i) {
                int best = 0;
                long long best_end = -1;
                for (int b = 0; b < nblk; ++b) {
                    const long long end = load[b] + c + (cnt[(size_t)b * 3 + m] == 0 && load[b] > 0 ? sw : 0);
                    if (best_end < 0 || end < best_end) {
                        best_end = end;
                        best = b;
                    }
                }
                load[best] = best_end;
                ++cnt[(size_t)best * 3 + m];
            }
        }
        Table t = {};
        bool fits = true;
        int at[3] = {0, 0, 0};
        for (int b = 0; b < nblk; ++b) {
            unsigned e[3] = {0, 0, 0};
            for (int m = 0; m < p.n_members; ++m) {
                const int c = cnt[(size_t)b * 3 + m];
                if (c > 31) fits = false;                        // 5-bit counts
                e[m] = (unsigned)at[m] | ((unsigned)c << 11);
                at[m] += c;
            }
            t[2 * b] = e[0] | (e[1] << 16);
            t[2 * b + 1] = e[2];
        }
        if (!fits) return;
        it = cache.emplace(key, t).first;
    }
    memcpy(p.sched, it->second.data(), sizeof(unsigned) * 2 * (size_t)nblk);
    p.sched_on = 1;
}

// pair_kernels.hpp pair_share on the host (the same integers)
static long long host_share(long long blk, long long total, long long base, int cost, long long n, int nblk) {
    const long long num = blk * total - base * nblk;
    if (num <= 0) return 0;
    const long long den = (long long)cost * nblk;
    const long long j = (num + den - 1) / den;
    return j > n ? n : j;
}

void pair_cut_schedule(PairParams& p, int nblk, const long long* n) {
    if (nblk < 1 || nblk > 2 * kSchedBlocks) return;
    long long total = 0, items = 0;
    for (int m = 0; m < p.n_members; ++m) {
        total += n[m] * p.m[m].cost;
        items += n[m];
    }
    if (items >= (1LL << 31)) return;
    // The table depends on (shares, item counts, costs) only: a forward asks for the same handful of tables call after call
    // (nblk x members divisions each, ~5 us of host time per launch) -- kept per shape, as pair_schedule's are.
    typedef std::array<long long, 8> Key;
    typedef std::array<unsigned, 2 * kSchedBlocks> Table;
    static std::mutex mu;
    static std::map<Key, Table> cache;
    Key key = {nblk, p.n_members, 0, 0, 0, 0, 0, 0};
    for (int m = 0; m < p.n_members; ++m) {
        key[2 + 2 * m] = n[m];
        key[3 + 2 * m] = p.m[m].cost;
    }
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(key);
        if (it == cache.end()) {
            Table t = {};
            for (int i = 0; i < nblk; ++i) {
                long long g = 0, base = 0;
                for (int m = 0; m < p.n_members; ++m) {
                    g += host_share(i, total, base, p.m[m].cost, n[m], nblk);
                    base += n[m] * p.m[m].cost;
                }
                t[i] = (unsigned)g;
            }
            if (cache.size() >= 512) cache.clear();
            it = cache.emplace(key, t).first;
        }
        memcpy(p.sched, it->second.data(), sizeof(unsigned) * (size_t)nblk);
    }
    p.sched_on = 2;
}

// run-time mirror of ConvHGeom<>
ConvHShape convh_shape(int C, int k, int dil) {
    ConvHShape g = {};
    g.NCH = C > 128 ? C / 128 : 1;     // the image holds at most 128 input channels at a time
    g.NMT = C / 64;
    C = C > 128 ? 128 : C;             // ... everything below is per chunk
    g.CG = C / 32;
    g.NFW = 2;
    g.NTC = 16 * g.NFW * 4;
    g.NSTEP = k * g.CG;
    g.NST = g.NSTEP / 2;
    g.XROWS = (g.NTC + (k - 1) * dil + 3) / 4 * 4;
    g.XIMG = 4 * C * ((g.XROWS + 15) / 16 * 16);
    return g;
}

int launch_convh(PairParams p, int C, int dil, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0) return 0;
    if (C != 64 && C != 128 && C != 256 && C != 512)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 conv: C = %d (64, 128, 256 or 512)", C);
    if (dil != 1 && dil != 3 && dil != 5 && dil != 9)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 conv: dilation %d (1, 3, 5; 9 with 3 taps)", dil);
    if (p.n_members < 1 || p.n_members > 3) return fail(FV_ERR_INVALID_ARG, "split-f16 conv: %d members", p.n_members);
    if ((double)C * p.T * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 conv: one utterance's tensor (%d x %d floats) exceeds the 1 GiB "
                    "buffer-descriptor range; split the utterance", C, p.T);
    if (p.slope < 0.f || p.slope > 1.f || p.act_slope < 0.f || p.act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "split-f16 conv: activation slope outside [0, 1]");
    int img_bytes = 0;
    double flops = 0, bytes = 0;
    long long items = 0;
    const int cus = device_cu_count();
    // 128 channels and more: 128-row tiles (convs_kernel, convr_kernels.hpp -- a chunk's window is converted once for
    // both 64-row tiles) when the launch has items enough for the chip that way (as launch_convg); Tuning::convh_rows64
    bool wide = false;
    if (C >= 128) {
        long long w = 0;
        for (int i = 0; i < p.n_members; ++i) w += (long long)((p.T + 127) / 128) * p.B * (C / 128);
        wide = tuning().convh_rows64 < 0 ? w * 10 >= 7LL * cus : !tuning().convh_rows64;
    }
    // 256 / 512 channels with items enough: the ring-free form (convs2_kernels.hpp: 128 rows x 64 columns, every wave loads the A
    // operands of its own sixteen rows L2 -> registers; no reflection padding, dilations 1 / 3 / 5) -- Tuning::convs_ringfree
    const bool ringfree = wide && C >= 256 && !p.reflect && dil <= 5 && tuning().convs_ringfree != 0;
    int rf_nh = 1;                     // 128-column tiles (32 x 64 wave tiles) when they still give every CU four items
    if (ringfree) {
        long long w = 0;
        for (int i = 0; i < p.n_members; ++i) w += (long long)((p.T + 127) / 128) * p.B * (C / 128);
        rf_nh = tuning().convs_ringfree > 0 ? tuning().convs_ringfree : (w >= 4LL * cus ? 2 : 1);
        if (rf_nh != 1 && rf_nh != 2) rf_nh = 1;
    }
    const int rf_cols = 64 * rf_nh;
    for (int i = 0; i < p.n_members; ++i) {
        PairMember& mb = p.m[i];
        if (mb.k != 11 && mb.k != 7 && mb.k != 3) return fail(FV_ERR_UNSUPPORTED, "split-f16 conv: %d taps (3, 7 or 11)", mb.k);
        if (dil == 9 && mb.k != 3) return fail(FV_ERR_UNSUPPORTED, "split-f16 conv: dilation 9 with %d taps (3 only)", mb.k);
        if (p.reflect && (mb.k - 1) / 2 * dil >= p.T)
            return fail(FV_ERR_INVALID_ARG, "split-f16 conv: reflection padding %d needs more than %d samples", (mb.k - 1) / 2 * dil, p.T);
        if (!mb.x || !mb.w1 || !mb.y) return fail(FV_ERR_INVALID_ARG, "split-f16 conv: null tensor (member %d)", i);
        if (mb.add2 && !mb.add1) return fail(FV_ERR_INVALID_ARG, "split-f16 conv: add2 without add1 (member %d)", i);
        if (reinterpret_cast<uintptr_t>(mb.w1) & 15)
            return fail(FV_ERR_UNSUPPORTED, "split-f16 conv: packed weights must be 16-byte aligned");
        const ConvHShape g = convh_shape(C, mb.k, dil);
        mb.n_tiles = ringfree ? (p.T + rf_cols - 1) / rf_cols : (p.T + g.NTC - 1) / g.NTC;
        mb.n_items = mb.n_tiles * p.B * (wide ? g.NMT / 2 : g.NMT);
        p.ctot = C;
        p.nch = g.NCH;
        p.nmt = g.NMT;
        // a tile costs its stages (LDS / matrix time) plus loads, conversion, epilogue
        mb.cost = g.NST * g.NCH + (tuning().convh_skel >= 0 ? tuning().convh_skel : (C == 64 ? 5 : 2));
        if (ringfree) {
            const int xrows = (rf_cols + (mb.k - 1) * dil + 3) / 4 * 4;
            const int ximg = 4 * 128 * ((xrows + 15) / 16 * 16);
            if (ximg > img_bytes) img_bytes = ximg;
        } else if (g.XIMG > img_bytes) img_bytes = g.XIMG;
        items += mb.n_items;
        flops += 2.0 * p.B * (double)C * C * mb.k * p.T;
        bytes += 4.0 * ((double)C * C * mb.k + (double)p.B * C * p.T *
                        (2 + (mb.res ? 1 : 0) + (mb.add1 ? 1 : 0) + (mb.add2 ? 1 : 0) + (mb.y_act ? 1 : 0)));
    }
    size_t floats = 0;
    p.x_off = 0;                       // ring of 4 weight stages
    if (!ringfree) floats += 4 * 16384 / 4;
    p.img_off = (int)floats;
    floats += (size_t)img_bytes / 4;
    p.bias_off = 0;                    // (biases are read from global memory in the epilogue)
    if (ringfree) floats += 64;        // the low-side guard's scratch
    const size_t lds = floats * 4;
    if (lds > 160 * 1024) return fail(FV_ERR_UNSUPPORTED, "split-f16 conv: %zu bytes of LDS", lds);
    long long nblk = tuning().convh_blocks > 0 ? tuning().convh_blocks : cus;     // one 8-wave block per CU
    if (nblk > items) nblk = items;
    p.nblk = (int)nblk;
    if (ringfree) p.sched_on = 0;
    else pair_schedule(p, p.nblk);
    if (!p.sched_on) {                 // no per-block schedule: the kernels' contiguous cut, as a table instead of arithmetic
        long long n[3] = {0, 0, 0};
        for (int i = 0; i < p.n_members; ++i) n[i] = p.m[i].n_items;
        pair_cut_schedule(p, p.nblk, n);
    }
    p.dbg = tuning().pair_dbg;
    p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);
    profile_begin(s);
    const int rc = ringfree ? launch_convs2_geom(p, dil, rf_nh, lds, s) : wide ? launch_convs_geom(p, dil, lds, s)
                 : C >= 128 ? launch_convh_geom<4, 2>(p, dil, lds, s) : launch_convh_geom<2, 2>(p, dil, lds, s);
    profile_end(s, C == 64 ? FV_KERNEL_CONVH64 : FV_KERNEL_CONVH128, flops, bytes);
    return rc;
}

// ---- ConvTranspose1d, kernel = 2 x stride (convt_kernel) -----------------------------------------------------------
int launch_convt(PairParams p, int Cin, int Cout, int stride, int pad, int Tout, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0 || Tout <= 0) return 0;
    if (Cin != 32 && Cin != 64 && Cin != 128 && Cin != 256 && Cin != 512)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 transposed conv: Cin = %d (32, 64, 128, 256 or 512)", Cin);
    if (stride < 2 || stride > 16 || Cout <= 0 || Cout * stride < 32)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 transposed conv: stride %d (2..16), Cout * stride = %d (32 or more)",
                    stride, Cout * stride);
    if (pad < 0 || pad > stride) return fail(FV_ERR_UNSUPPORTED, "split-f16 transposed conv: padding %d (0..stride)", pad);
    if ((double)Cin * p.T * 4.0 >= 1073741824.0 || (double)Cout * Tout * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 transposed conv: one utterance's tensor exceeds the 1 GiB "
                    "buffer-descriptor range; split the utterance");
    if (p.slope < 0.f || p.slope > 1.f || p.act_slope < 0.f || p.act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "split-f16 transposed conv: activation slope outside [0, 1]");
    PairMember& mb = p.m[0];
    if (!mb.x || !mb.w1 || !mb.y) return fail(FV_ERR_INVALID_ARG, "split-f16 transposed conv: null tensor");
    if (reinterpret_cast<uintptr_t>(mb.w1) & 15)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 transposed conv: packed weights must be 16-byte aligned");
    p.n_members = 1;
    p.ctot = Cin;
    const int cc = Cin <= 64 ? 64 : 128;      // input channels per chunk (64 input channels: one chunk of 64, two K steps per tap;
                                              // 32: half of that chunk -- the missing channels read as zeros)
    p.nch = (Cin + cc - 1) / cc;
    p.nmt = (Cout * stride + 63) / 64;        // rows beyond Cout * stride: zero weights, stores dropped
    p.cout = Cout;
    p.ups = stride;
    p.pad_t = pad;
    p.Tout = Tout;
    p.reflect = 0;
    const int ncols = (Tout + pad + stride - 1) / stride;          // u = (n + pad) / stride of the last sample, + 1
    mb.k = 2;
    // Few items: the lean kernel (convtl_kernels.hpp: 64-column tiles, A operands L2 -> registers, no ring) -- a launch of a
    // handful of tiles per CU is all prologue on the ring pipeline below.  Tuning::convt_lean: (64 x 64 item, chunk) units per CU
    // (in tenths) up to which it runs; 0: never (A/B, bit-identity tests).  [measured, tools/convt_lean_bench.py, hot loop]
    // 256 -> 128 x 8 at 1000 frames (2 units per CU): 10.6 against 15.6 us; 512 -> 256 x 8 at 200 frames (2): 15.2 / 23.6;
    // 64 -> 32 x 3 at 40 000 (4.9): 17.5 / 19.6; 128 -> 64 x 5 at 8000 (2.5): 15.9 / 17.0; at 3.3 units the two are equal, from
    // ~8 up the ring pipeline wins (batch 16: 74 against 107 us) -- its weights reach all eight waves through LDS once.
    {
        const long long lean_items = (long long)((ncols + 63) / 64) * p.B * p.nmt;
        if (tuning().convt_lean > 0 && lean_items * p.nch * 10 <= (long long)tuning().convt_lean * device_cu_count()) {
            if (!mb.add1) p.out_div = 1.f;
            if (mb.add2 && !mb.add1) return fail(FV_ERR_INVALID_ARG, "split-f16 transposed conv: add2 without add1");
            mb.n_tiles = (ncols + 63) / 64;
            mb.n_items = (int)lean_items;
            mb.cost = 1;
            long long nb = tuning().convh_blocks > 0 ? tuning().convh_blocks : device_cu_count();
            if (nb > lean_items) nb = lean_items;
            p.nblk = (int)nb;
            p.sched_on = 0;
            p.dbg = 0;
            p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);
            profile_begin(s);
            const int rc = launch_convtl_geom(p, cc / 32, s);
            profile_end(s, FV_KERNEL_CONVT, 2.0 * p.B * (double)p.T * Cin * Cout * 2 * stride,
                        4.0 * ((double)Cin * Cout * 2 * stride + (double)p.B * ((double)Cin * p.T + (double)Cout * Tout * (mb.y_act ? 2 : 1))));
            return rc;
        }
    }
    const int NTC = 128;
    mb.n_tiles = (ncols + NTC - 1) / NTC;
    // Many items, 128 / 256 input channels, no merged input: an item is a column tile with ALL its rows on resident images
    // (convu2_kernels.hpp) -- convu_kernel below converts every window once per PAIR of 64-row tiles.  Tuning::convu_resident
    {
        const long long items = (long long)mb.n_tiles * p.B;
        const int mode = tuning().convu_resident;
        if (cc == 128 && p.nch <= 2 && p.nmt * 64 <= 1024 && !mb.add1 && mode != 0 && (mode == 2 || items >= 2LL * device_cu_count())) {
            p.out_div = 1.f;
            mb.n_items = (int)items;
            mb.cost = 1;
            long long nb = tuning().convh_blocks > 0 ? tuning().convh_blocks : device_cu_count();
            if (nb > items) nb = items;
            p.nblk = (int)nb;
            p.sched_on = 0;
            p.dbg = 0;
            p.trace = nullptr;
            profile_begin(s);
            const int rc = launch_convu2_geom(p, p.nch, s);
            profile_end(s, FV_KERNEL_CONVT, 2.0 * p.B * (double)p.T * Cin * Cout * 2 * stride,
                        4.0 * ((double)Cin * Cout * 2 * stride + (double)p.B * ((double)Cin * p.T + (double)Cout * Tout * (mb.y_act ? 2 : 1))));
            return rc;
        }
    }
    // 128 and more input channels: 128-row tiles (convu_kernel, convr_kernels.hpp) when that still gives the chip items
    // enough, as launch_convg / launch_convh; Tuning::convt_rows64
    const long long wide_items = (long long)mb.n_tiles * p.B * ((p.nmt + 1) / 2);
    // (a merged input -- mb.add1: the upsampler behind an MRF stage forms ((x + add1) + add2) / out_div in its window loader --
    // exists on the 64-row kernel only: the 128-row one has no registers left for the second window)
    const bool wide = cc == 128 && !mb.add1 &&
                      (tuning().convt_rows64 < 0 ? wide_items * 10 >= 7LL * device_cu_count() : !tuning().convt_rows64);
    if (!mb.add1) p.out_div = 1.f;
    if (mb.add2 && !mb.add1) return fail(FV_ERR_INVALID_ARG, "split-f16 transposed conv: add2 without add1");
    mb.n_items = wide ? (int)wide_items : mb.n_tiles * p.B * p.nmt;
    mb.cost = 1;
    const int xrows = (NTC + 1 + 3) / 4 * 4;
    const int img_bytes = 4 * cc * ((xrows + 15) / 16 * 16);
    p.x_off = 0;                       // ring of 4 weight stages
    p.img_off = 4 * 16384 / 4;
    const size_t lds = 4 * 16384 + (size_t)img_bytes;
    long long nblk = tuning().convh_blocks > 0 ? tuning().convh_blocks : device_cu_count();
    if (nblk > mb.n_items) nblk = mb.n_items;
    p.nblk = (int)nblk;
    p.sched_on = 0;
    p.dbg = tuning().pair_dbg;
    p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);
    profile_begin(s);
    const int rc = wide ? launch_convu_geom(p, lds, s) : launch_convt_geom(p, cc / 32, lds, s);
    // MACs of a ConvTranspose1d = Tin * Cin * Cout * k
    profile_end(s, FV_KERNEL_CONVT, 2.0 * p.B * (double)p.T * Cin * Cout * 2 * stride,
                4.0 * ((double)Cin * Cout * 2 * stride + (double)p.B * ((double)Cin * p.T + (double)Cout * Tout * (mb.y_act ? 2 : 1))));
    return rc;
}

// Long runs of the fused 64- / 128-channel pairs (convp / convq: many tiles per block) are mostly WARM tiles -- NM outputs
// for the work a cold tile spends on NM - (k - 1) (convq_kernels.hpp) -- so an ITEM (NM - (k - 1) columns) of an 11-tap
// member costs 84 % of a tile there, one of a 3-tap member 97 %: the members' partition weights follow, or the blocks that
// walk the 3-tap member finish last [measured, 128 channels, 16 x 64 000 columns, three members: 4.76 ms with equal
// tile costs whatever the tiling, against 3.50 vs 3.87 ms for the two-member launch]
#ifndef FV_WARM_TILES
#define FV_WARM_TILES 1
#endif
static void warm_run_costs(PairParams& p, long long items, int nm) {
    if (!FV_WARM_TILES || items < 8LL * p.nblk) return;
    for (int i = 0; i < p.n_members; ++i) p.m[i].cost *= nm - (p.m[i].k - 1);
}

// ---- two-source 1x1 conv (convg_kernel) -----------------------------------------------------------------------------
int launch_convg(PairParams p, int C, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0) return 0;
    if (C != 128 && C != 256 && C != 512)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 two-source 1x1 conv: C = %d (128, 256 or 512)", C);
    if ((double)C * p.T * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 two-source 1x1 conv: one utterance's tensor (%d x %d floats) exceeds the "
                    "1 GiB buffer-descriptor range; split the utterance", C, p.T);
    if (p.slope < 0.f || p.slope > 1.f || p.act_slope < 0.f || p.act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "split-f16 two-source 1x1 conv: activation slope outside [0, 1]");
    PairMember& mb = p.m[0];
    if (!mb.x || !mb.x2 || !mb.w1 || !mb.y) return fail(FV_ERR_INVALID_ARG, "split-f16 two-source 1x1 conv: null tensor");
    if (reinterpret_cast<uintptr_t>(mb.w1) & 15)
        return fail(FV_ERR_UNSUPPORTED, "split-f16 two-source 1x1 conv: packed weights must be 16-byte aligned");
    p.n_members = 1;
    p.ctot = C;                                // channels of each input and of the output
    p.nch = 2 * C / 128;                       // chunks of 128 input channels: x's, then x2's
    p.nmt = C / 64;
    p.reflect = 0;
    p.out_div = 1.f;
    const int NTC = 128;
    mb.k = 1;
    mb.n_tiles = (p.T + NTC - 1) / NTC;
    // 128 output rows per item (convr_kernels.hpp: a converted chunk of the inputs feeds two 64-row tiles), or 64
    // (convg_kernel; Tuning::convg_rows64, A/B and bit-identity tests)
    // [measured, MI355X] Basis-MelGAN light, 64 utterances: 16.3 -> 14.0 ms per step on 128-row tiles; MelGAN, one
    // utterance of 200 frames (13 - 100 column tiles per launch): 0.44 -> 0.47 ms -- with fewer items than CUs the
    // 64-row tiles keep more of the chip busy.  -1: by the item count.
    const int cus = device_cu_count();
    const bool wide = tuning().convg_rows64 < 0 ? (long long)mb.n_tiles * p.B * (p.nmt / 2) * 10 >= 7LL * cus : !tuning().convg_rows64;
    mb.n_items = mb.n_tiles * p.B * (wide ? p.nmt / 2 : p.nmt);
    mb.cost = 1;
    p.x_off = 0;                               // ring of 4 weight stages
    p.img_off = 4 * 16384 / 4;
    const size_t lds = 4 * 16384 + (size_t)(4 * 128 * 128);      // + image: 128 channels x 128 rows x (2 + 2) bytes
    long long nblk = tuning().convh_blocks > 0 ? tuning().convh_blocks : device_cu_count();
    if (nblk > mb.n_items) nblk = mb.n_items;
    p.nblk = (int)nblk;
    p.sched_on = 0;
    p.dbg = tuning().pair_dbg;
    p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);
    profile_begin(s);
    const int rc = wide ? launch_convr_geom(p, lds, s) : launch_convg_geom(p, lds, s);
    profile_end(s, FV_KERNEL_CONVG, 2.0 * p.B * (double)C * 2 * C * p.T,
                4.0 * (2.0 * C * C + (double)p.B * C * p.T * (3 + (mb.res ? 1 : 0) + (mb.y_act ? 1 : 0))));
    return rc;
}

// ---- fused pair at C = 64 (convq2_kernels.hpp) -------------------------------------------------------------------
constexpr int kPair64Wide = 65, kPair128Wide = 129;     // (convq2_kernels.hpp: the wide forms of the ring-free pair kernel)
template <int DIL, int C>
int launch_convq2_dil(const PairParams& p, size_t lds, hipStream_t s);      // convq2_inst.hip: the pairs without the weight ring
extern template int launch_convq2_dil<1, 128>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<3, 128>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<5, 128>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<1, 64>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<3, 64>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<5, 64>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<1, kPair64Wide>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<3, kPair64Wide>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<5, kPair64Wide>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<1, kPair128Wide>(const PairParams&, size_t, hipStream_t);
extern template int launch_convq2_dil<3, kPair128Wide>(const PairParams&, size_t, hipStream_t);
template <int DIL>
int launch_convq3_dil(const PairParams& p, hipStream_t s);                  // convq3_inst.hip: two wave groups one conv phase apart
extern template int launch_convq3_dil<1>(const PairParams&, hipStream_t);
extern template int launch_convq3_dil<3>(const PairParams&, hipStream_t);
extern template int launch_convq3_dil<5>(const PairParams&, hipStream_t);
template <int DIL>
int launch_convq4_dil(const PairParams& p, hipStream_t s);                  // ... the same pipelines as blocks of their own, two per CU
extern template int launch_convq4_dil<1>(const PairParams&, hipStream_t);
extern template int launch_convq4_dil<3>(const PairParams&, hipStream_t);
extern template int launch_convq4_dil<5>(const PairParams&, hipStream_t);

// everything launch_convp does in front of the launch: validation, member order, tile counts, LDS layout, block schedule
// (form: 1 convq2_kernel<DIL, 64> -- 128-column tiles; 2 convq2_kernel<DIL, 65> -- 256-column tiles; 3 convq3_kernel<DIL> -- two
// wave groups per block, each on 64-column tiles of its own share: 2 nblk shares, LDS laid out by the kernel)
static int prepare_convp(PairParams& p, int dil, size_t& lds_out, double& flops, double& bytes, int form) {
    const int C = 64;
    const int NMc = form == 2 ? 256 : form == 3 ? 64 : 128;           // intermediate columns per tile
    if (dil != 1 && dil != 3 && dil != 5) return fail(FV_ERR_UNSUPPORTED, "resblock pair: dilation %d (1, 3 or 5)", dil);
    if (p.n_members < 1 || p.n_members > 3) return fail(FV_ERR_INVALID_ARG, "resblock pair: %d members", p.n_members);
    if ((double)C * p.T * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "resblock pair: one utterance's tensor (%d x %d floats) exceeds the 1 GiB "
                    "buffer-descriptor range; split the utterance", C, p.T);
    if (p.slope < 0.f || p.slope > 1.f || p.act_slope < 0.f || p.act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "resblock pair: activation slope outside [0, 1]");
    for (int i = 0; i < p.n_members; ++i)        // largest taps first (cost order of the contiguous partition)
        for (int j = i + 1; j < p.n_members; ++j)
            if (p.m[j].k > p.m[i].k) { PairMember t = p.m[i]; p.m[i] = p.m[j]; p.m[j] = t; }
    int img_bytes = 0;
    flops = bytes = 0;
    long long items = 0;
    for (int i = 0; i < p.n_members; ++i) {
        PairMember& mb = p.m[i];
        if (mb.k != 11 && mb.k != 7 && mb.k != 3) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %d taps (3, 7 or 11)", mb.k);
        if (!mb.x || !mb.w1 || !mb.w2 || !mb.y) return fail(FV_ERR_INVALID_ARG, "resblock pair: null tensor (member %d)", i);
        if (mb.add2 && !mb.add1) return fail(FV_ERR_INVALID_ARG, "resblock pair: add2 without add1 (member %d)", i);
        if ((reinterpret_cast<uintptr_t>(mb.w1) | reinterpret_cast<uintptr_t>(mb.w2)) & 15)
            return fail(FV_ERR_UNSUPPORTED, "resblock pair: packed weights must be 16-byte aligned");
        const ConvHShape g = convh_shape(C, mb.k, dil);
        const int nout = NMc - (mb.k - 1);
        mb.n_tiles = (p.T + nout - 1) / nout;
        mb.n_items = mb.n_tiles * p.B;
        mb.cost = 2 * g.NST + tuning().convp_skel;
        const int xrows = (NMc + (mb.k - 1) * dil + 3) / 4 * 4;
        const int ximg = form == 2 ? 2 * (C / 8) * ((xrows + 15) / 16 * 16) * 16 : g.XIMG;
        if (ximg > img_bytes) img_bytes = ximg;
        items += mb.n_items;
        flops += 2.0 * 2.0 * p.B * (double)C * C * mb.k * p.T;
        bytes += 4.0 * (2.0 * C * C * mb.k + (double)p.B * C * p.T * ((mb.y_act ? 3 : 2) + (mb.add1 ? 1 : 0) + (mb.add2 ? 1 : 0)));
    }
    size_t floats = 0;
    p.x_off = 0;
    p.img_off = (int)floats;
    floats += (size_t)img_bytes / 4;
    p.mid_off = (int)floats;
    floats += (size_t)4 * C * (NMc + 16) / 4;
    p.bias_off = (int)floats;
    floats += 4 * (size_t)C + 16;      // [b1 | b2 | inverse row prescales of conv1 | conv2 | scratch of the low-range guard]
    const size_t lds = floats * 4;
    if (lds > 160 * 1024) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %zu bytes of LDS", lds);
    long long nblk = tuning().convh_blocks > 0 ? tuning().convh_blocks : device_cu_count();
    if (form == 3) {
        // one share per wave group: 2 nblk contiguous, cost-balanced shares of the concatenated items
        if (nblk > kSchedBlocks) nblk = kSchedBlocks;
        if (2 * nblk > items) nblk = (items + 1) / 2;
        p.nblk = (int)nblk;
        p.sched_on = 0;
        warm_run_costs(p, items, NMc);
        long long n[3] = {0, 0, 0};
        for (int i = 0; i < p.n_members; ++i) n[i] = p.m[i].n_items;
        pair_cut_schedule(p, 2 * p.nblk, n);
        if (p.sched_on != 2) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %lld items do not fit the share table", items);
        p.dbg = tuning().pair_dbg;
        p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);
        lds_out = 0;
        return 0;
    }
    if (nblk > items) nblk = items;
    p.nblk = (int)nblk;
    pair_schedule(p, p.nblk);
    if (!p.sched_on) {                 // no per-block schedule: the kernels' contiguous cut, as a table instead of arithmetic
        warm_run_costs(p, items, NMc);
        long long n[3] = {0, 0, 0};
        for (int i = 0; i < p.n_members; ++i) n[i] = p.m[i].n_items;
        pair_cut_schedule(p, p.nblk, n);
    }
    p.dbg = tuning().pair_dbg;
    p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);
    lds_out = lds;
    return 0;
}

int launch_convp(PairParams p, int dil, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0) return 0;
    size_t lds;
    double flops, bytes;
    // convq2_kernels.hpp at 64 channels (A operands from L2 into registers) on 128-column tiles, and on 256-column tiles
    // (32 x 64 wave tiles) from Tuning::convp_wide tenths of such a tile per CU up
    long long wide_items = 0;
    for (int i = 0; i < p.n_members; ++i) wide_items += (long long)p.B * ((p.T + 256 - p.m[i].k) / (257 - p.m[i].k));
    const int form = tuning().convp_pp ? 3 : wide_items * 10 >= (long long)tuning().convp_wide * device_cu_count() ? 2 : 1;
    if (int rc = prepare_convp(p, dil, lds, flops, bytes, form)) return rc;
    profile_begin(s);
    const int rc = form == 3 && tuning().convp_pp == 2 ? (dil == 1 ? launch_convq4_dil<1>(p, s) : dil == 3 ? launch_convq4_dil<3>(p, s) : launch_convq4_dil<5>(p, s))
                   : form == 3 ? (dil == 1 ? launch_convq3_dil<1>(p, s) : dil == 3 ? launch_convq3_dil<3>(p, s) : launch_convq3_dil<5>(p, s))
                   : form == 2 ? (dil == 1 ? launch_convq2_dil<1, kPair64Wide>(p, lds, s) : dil == 3 ? launch_convq2_dil<3, kPair64Wide>(p, lds, s) : launch_convq2_dil<5, kPair64Wide>(p, lds, s))
                             : (dil == 1 ? launch_convq2_dil<1, 64>(p, lds, s) : dil == 3 ? launch_convq2_dil<3, 64>(p, lds, s) : launch_convq2_dil<5, 64>(p, lds, s));
    profile_end(s, FV_KERNEL_CONVH64, flops, bytes);
    return rc;
}

// ---- fused pair at C = 128 (convq2_kernels.hpp) ------------------------------------------------------------------

int launch_convq(PairParams p, int dil, hipStream_t s) {
    const int C = 128;
    // 128-column tiles (convq2_kernel<DIL, 129>: 32 x 64 wave tiles) where both images fit without the ring -- dilation 1 and
    // 3 -- from Tuning::convq_wide tenths of such a tile per CU up
    long long wide_items = 0;
    for (int i = 0; i < p.n_members && i < 3; ++i) wide_items += (long long)p.B * ((p.T + 128 - p.m[i].k) / (129 - p.m[i].k));
    const bool wide = dil <= 3 && wide_items * 10 >= (long long)tuning().convq_wide * device_cu_count();
    const int NM = wide ? 128 : 64;
    if (p.B <= 0 || p.T <= 0) return 0;
    if (dil != 1 && dil != 3 && dil != 5) return fail(FV_ERR_UNSUPPORTED, "resblock pair: dilation %d (1, 3 or 5)", dil);
    if (p.n_members < 1 || p.n_members > 3) return fail(FV_ERR_INVALID_ARG, "resblock pair: %d members", p.n_members);
    if ((double)C * p.T * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "resblock pair: one utterance's tensor (%d x %d floats) exceeds the 1 GiB "
                    "buffer-descriptor range; split the utterance", C, p.T);
    if (p.slope < 0.f || p.slope > 1.f || p.act_slope < 0.f || p.act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "resblock pair: activation slope outside [0, 1]");
    for (int i = 0; i < p.n_members; ++i)        // largest taps first (cost order of the contiguous partition)
        for (int j = i + 1; j < p.n_members; ++j)
            if (p.m[j].k > p.m[i].k) { PairMember t = p.m[i]; p.m[i] = p.m[j]; p.m[j] = t; }
    int img_bytes = 0;
    double flops = 0, bytes = 0;
    long long items = 0;
    for (int i = 0; i < p.n_members; ++i) {
        PairMember& mb = p.m[i];
        if (mb.k != 11 && mb.k != 7 && mb.k != 3) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %d taps (3, 7 or 11)", mb.k);
        if (!mb.x || !mb.w1 || !mb.w2 || !mb.y) return fail(FV_ERR_INVALID_ARG, "resblock pair: null tensor (member %d)", i);
        if (mb.add2 && !mb.add1) return fail(FV_ERR_INVALID_ARG, "resblock pair: add2 without add1 (member %d)", i);
        if ((reinterpret_cast<uintptr_t>(mb.w1) | reinterpret_cast<uintptr_t>(mb.w2)) & 15)
            return fail(FV_ERR_UNSUPPORTED, "resblock pair: packed weights must be 16-byte aligned");
        const int nout = NM - (mb.k - 1);
        mb.n_tiles = (p.T + nout - 1) / nout;
        mb.n_items = mb.n_tiles * p.B;
        mb.cost = 4 * mb.k + (tuning().convq_skel >= 0 ? tuning().convq_skel : 5);     // K steps of one conv + per-tile work
        const int xrows = (NM + (mb.k - 1) * dil + 3) / 4 * 4;
        const int img = 4 * C * ((xrows + 15) / 16 * 16);
        if (img > img_bytes) img_bytes = img;
        items += mb.n_items;
        flops += 2.0 * 2.0 * p.B * (double)C * C * mb.k * p.T;
        bytes += 4.0 * (2.0 * C * C * mb.k + (double)p.B * C * p.T * ((mb.y_act ? 3 : 2) + (mb.add1 ? 1 : 0) + (mb.add2 ? 1 : 0)));
    }
    size_t floats = 0;
    p.x_off = 0;
    p.img_off = (int)floats;
    floats += (size_t)img_bytes / 4;
    p.mid_off = (int)floats;
    floats += (size_t)4 * C * (NM + 16) / 4;
    p.bias_off = (int)floats;
    floats += 4 * (size_t)C + 16;      // [b1 | b2 | inverse row prescales of conv1 | conv2 | scratch of the low-range guard]
    const size_t lds = floats * 4;
    if (lds > 160 * 1024) return fail(FV_ERR_UNSUPPORTED, "resblock pair: %zu bytes of LDS", lds);
    long long nblk = tuning().convh_blocks > 0 ? tuning().convh_blocks : device_cu_count();
    if (nblk > items) nblk = items;
    p.nblk = (int)nblk;
    pair_schedule(p, p.nblk, true);
    if (!p.sched_on) {                 // no per-block schedule: the kernels' contiguous cut, as a table instead of arithmetic
        warm_run_costs(p, items, NM);
        long long n[3] = {0, 0, 0};
        for (int i = 0; i < p.n_members; ++i) n[i] = p.m[i].n_items;
        pair_cut_schedule(p, p.nblk, n);
    }
    p.dbg = tuning().pair_dbg;
    p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);
    profile_begin(s);
    const int rc = wide ? (dil == 1 ? launch_convq2_dil<1, kPair128Wide>(p, lds, s) : launch_convq2_dil<3, kPair128Wide>(p, lds, s))
                      : (dil == 1 ? launch_convq2_dil<1, 128>(p, lds, s) : dil == 3 ? launch_convq2_dil<3, 128>(p, lds, s) : launch_convq2_dil<5, 128>(p, lds, s));
    profile_end(s, FV_KERNEL_CONVH128, flops, bytes);
    return rc;
}

}  // namespace fv
