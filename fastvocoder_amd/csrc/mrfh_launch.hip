// Host side of the one-launch MRF stage (mrfh_kernels.hpp): validation, blob offsets, window geometry, grid.
#include "fv_internal.h"

namespace fv {

extern template int launch_mrfh_geom<3, 12>(const MrfParams&, hipStream_t);
extern template int launch_mrfh_geom<2, 16>(const MrfParams&, hipStream_t);

// history of the 32-channel kernel: one slot set per block; a launch has at most max(CUs, Tuning::mrf_blocks) blocks
static long long mrfw_max_blocks() {
    const long long cus = device_cu_count(), asked = tuning().mrf_blocks;
    return asked > cus ? asked : cus;
}
long long mrf_workspace_bytes(int C) { return C == 32 ? mrfw_max_blocks() * kMrfwHistBytes : 0; }

int launch_mrfh(MrfParams p, int C, const int* dil, hipStream_t s) {
    if (p.B <= 0 || p.T <= 0) return 0;
    if (!mrf_stage_shape(C, p.k, dil))
        return fail(FV_ERR_UNSUPPORTED, "mrf stage: C = %d, taps (%d, %d, %d), dilations (%d, %d, %d): built for 16 / 32 channels, taps "
                    "3 / 7 / 11, dilations (1, 3, 5)", C, p.k[0], p.k[1], p.k[2], dil ? dil[0] : 0, dil ? dil[1] : 0, dil ? dil[2] : 0);
    const bool fold = p.fold_w != nullptr;
    if (!p.x || !p.blob || (!fold && !p.y) || (fold && (!p.fold_y || p.y || p.y_act)))
        return fail(FV_ERR_INVALID_ARG, "mrf stage: null tensor (or both an output tensor and a folded output conv)");
    // every block reads x columns (a run's first tile starts `halo` early, plus the right-hand recompute margin) that its
    // neighbours write as y: an output that is the input would be read half-updated (the pair entries refuse it too)
    if (p.y == p.x || p.y_act == p.x || (p.y_act && p.y_act == p.y) || p.fold_y == p.x)
        return fail(FV_ERR_INVALID_ARG, "mrf stage: an output tensor aliases the input (or y_act aliases y)");
    if ((reinterpret_cast<uintptr_t>(p.blob) & 15) != 0) return fail(FV_ERR_UNSUPPORTED, "mrf stage: the packed stage must be 16-byte aligned");
    if ((double)C * p.T * 4.0 >= 1073741824.0)
        return fail(FV_ERR_UNSUPPORTED, "mrf stage: one utterance's tensor (%d x %d floats) exceeds the 1 GiB buffer-descriptor "
                    "range; split the utterance", C, p.T);
    if ((double)p.B * p.T >= 2147483647.0) return fail(FV_ERR_UNSUPPORTED, "mrf stage: B x T = %d x %d columns exceed 2^31", p.B, p.T);
    if (p.slope < 0.f || p.slope > 1.f || p.act_slope < 0.f || p.act_slope > 1.f)
        return fail(FV_ERR_INVALID_ARG, "mrf stage: activation slope outside [0, 1]");
    unsigned off = 0;
    double flops = 0;
    for (int j = 0; j < 3; ++j)
        for (int q = 0; q < 3; ++q) {
            p.blk_off[3 * j + q] = off;
            off += (unsigned)mrf_block_bytes(C, p.k[j]);
            flops += 2.0 * 2.0 * p.B * (double)C * C * p.k[j] * p.T;
        }
    p.blob_bytes = off;
    p.halo = mrf_halo(p.k, dil);
    p.ol = fold ? 3 : 0;
    p.total = (long long)p.B * p.T;
    const int shape = tuning().mrf_shape;
    const int W = C == 32 ? 384 : shape == 1 ? 512 : 576;
    const int vcols = W - p.halo, adv = vcols - 2 * p.ol;
    if (adv < 64) return fail(FV_ERR_UNSUPPORTED, "mrf stage: a %d-column window leaves %d final columns", W, adv);
    // One block per CU (the weight slots, two images and the history are 147 KB of LDS); a share below ~a quarter window
    // is all run-in (a run's first tile starts `halo` columns early).
    long long nblk = tuning().mrf_blocks > 0 ? tuning().mrf_blocks : device_cu_count();
    const long long most = (p.total + 127) / 128;
    if (nblk > most) nblk = most;
    if (nblk < 1) nblk = 1;
    p.nblk = (int)nblk;
    if (C == 32) {
        if (!p.hist || p.hist_bytes < nblk * kMrfwHistBytes || (reinterpret_cast<uintptr_t>(p.hist) & 15) != 0)
            return fail(FV_ERR_WORKSPACE, "mrf stage: 32 channels need a 16-byte aligned workspace of fv_mrf_stage_workspace_bytes "
                        "(%lld bytes for %lld blocks; given %lld)", nblk * kMrfwHistBytes, nblk, p.hist ? p.hist_bytes : 0LL);
    } else {
        p.hist = nullptr;
    }
    p.prio = tuning().mrf_prio;
    p.trace = reinterpret_cast<unsigned long long*>(tuning().trace_ptr);
    double bytes = 4.0 * ((double)p.B * C * p.T * (fold ? 1.0 : (p.y_act ? 3.0 : 2.0)) + (fold ? (double)p.B * p.T : 0.0)) + off;
    if (fold) flops += 2.0 * p.B * (double)C * 7 * p.T;   // (the folded output conv: C -> 1 channels, 7 taps)
    profile_begin(s);
    const int rc = C == 32 ? launch_mrfw_geom(p, s) : shape == 1 ? launch_mrfh_geom<2, 16>(p, s) : launch_mrfh_geom<3, 12>(p, s);
    profile_end(s, C == 32 ? FV_KERNEL_MRF32 : FV_KERNEL_MRF16, flops, bytes);
    return rc;
}

}  // namespace fv
