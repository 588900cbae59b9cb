// The pair launches of a 64-channel MRF stage as ONE chained launch (fv_internal.h PairChain): the phases are the
// launches convp_kernel would have been, the same tiles with the same arithmetic; see convp_run_member<G, true>.
#pragma once
#include "convp_kernels.hpp"

namespace fv {

// ---- the launches of an MRF stage at 64 channels as one chained launch (fv_internal.h PairChain) -------------------
template <int DIL>
__device__ __forceinline__ void convp_chain_member(const PairCore& q, const PairMember& mb, int first_item, int end_item,
                                                   int dir, float* smem, int wave, int lane, bool first,
                                                   const PairMember::Dep* dep, const ChainCtx& cc) {
    if (mb.k == 11) convp_run_member<ConvPGeom<11, DIL>, true>(q, mb, first_item, end_item, smem, wave, lane, first, dep, &cc, dir);
    else if (mb.k == 7) convp_run_member<ConvPGeom<7, DIL>, true>(q, mb, first_item, end_item, smem, wave, lane, first, dep, &cc, dir);
    else convp_run_member<ConvPGeom<3, DIL>, true>(q, mb, first_item, end_item, smem, wave, lane, first, dep, &cc, dir);
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void convp_chain_kernel(PairChain c) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ChainCtx cc;
    cc.rf = make_rsrc(reinterpret_cast<const float*>(c.ph[0].flags), c.ph[0].flag_bytes);
    cc.epoch = c.ph[0].epoch;
    cc.spin_limit = c.spin_limit;
    cc.guard = c.ph[0].guard;
    cc.stall = nullptr;
    bool first = true;
    for (int ph = 0; ph < c.n_phases; ++ph) {
        const PairCore& p = c.ph[ph];
        PairCore q;
        q.n_members = p.n_members; q.B = p.B; q.T = p.T; q.slope = p.slope; q.out_div = p.out_div;
        q.act_slope = p.act_slope; q.post = p.post; q.x_off = p.x_off; q.img_off = p.img_off; q.mid_off = p.mid_off;
        q.bias_off = p.bias_off; q.dbg = p.dbg; q.trace = p.trace; q.guard = p.guard;
        const int dil = c.dil[ph];
        // this block's share of the phase (chain_schedule, convh_launch.hip): per member lo | count << 20, word 3: bit 0 =
        // the phase runs from the last item down
        const unsigned* tab = c.sched + ((size_t)ph * gridDim.x + xcd_remap((int)blockIdx.x, (int)gridDim.x)) * 4;
        const unsigned w0 = tab[0], w1 = tab[1], w2 = tab[2], w3 = tab[3];
        const int dir = (w3 & 1u) ? -1 : 1;
#ifdef FV_PAIR_TRACE
        // tuning aid (tools/chain_trace.py): per block and phase [start, end, ticks in chain_spin (all waves), spins,
        // the four schedule words]
        unsigned long long* const tr = c.ph[0].trace ? c.ph[0].trace + ((size_t)blockIdx.x * kChainPhases + ph) * 8 : nullptr;
        cc.stall = tr ? tr + 2 : nullptr;
        if (tr && tid == 0) {
            tr[0] = __builtin_amdgcn_s_memtime();
            tr[4] = w0; tr[5] = w1; tr[6] = w2; tr[7] = w3;
        }
#endif
        asm volatile("" ::"s"(q.n_members), "s"(q.B), "s"(q.T), "s"(q.slope), "s"(q.out_div), "s"(q.act_slope),
                     "s"(q.post), "s"(q.x_off), "s"(q.img_off), "s"(q.mid_off), "s"(q.bias_off), "s"(q.dbg), "s"(q.trace),
                     "s"(q.guard), "s"(dil), "s"(w0), "s"(w1), "s"(w2), "s"(dir));
        for (int mi = 0; mi < q.n_members; ++mi) {
            const int m = dir > 0 ? mi : q.n_members - 1 - mi;
            const unsigned w = m == 0 ? w0 : m == 1 ? w1 : w2;
            const int lo = (int)(w & 0xFFFFFu), cnt = (int)(w >> 20);
            if (cnt == 0) continue;
            PairMember mb;
            mb.x = p.m[m].x; mb.w1 = p.m[m].w1; mb.w2 = p.m[m].w2; mb.b1 = p.m[m].b1; mb.b2 = p.m[m].b2; mb.add1 = p.m[m].add1;
            mb.add2 = p.m[m].add2; mb.y = p.m[m].y; mb.y_act = p.m[m].y_act; mb.k = p.m[m].k; mb.n_tiles = p.m[m].n_tiles;
            mb.flag_off = p.m[m].flag_off;
            asm volatile("" ::"s"(mb.x), "s"(mb.w1), "s"(mb.w2), "s"(mb.b1), "s"(mb.b2), "s"(mb.add1), "s"(mb.add2), "s"(mb.y),
                         "s"(mb.y_act), "s"(mb.k), "s"(mb.n_tiles), "s"(mb.flag_off));
            const PairMember::Dep* dep = p.m[m].dep;
            const int i0 = dir > 0 ? lo : lo + cnt - 1, i1 = dir > 0 ? lo + cnt : lo - 1;
            if (dil == 1) convp_chain_member<1>(q, mb, i0, i1, dir, smem, wave, lane, first, dep, cc);
            else if (dil == 3) convp_chain_member<3>(q, mb, i0, i1, dir, smem, wave, lane, first, dep, cc);
            else convp_chain_member<5>(q, mb, i0, i1, dir, smem, wave, lane, first, dep, cc);
            first = false;
        }
#ifdef FV_PAIR_TRACE
        if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memtime();
#endif
    }
}

}  // namespace fv
