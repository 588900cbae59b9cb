// launch of one split-f16 pair-kernel geometry: picks the dilation instantiation
#pragma once
#include "pairh_kernels.hpp"

namespace fv {

template <int MH, int NF, int NG>
int launch_pairh_geom(const PairParams& p, int dil, size_t lds, hipStream_t s) {
#define FV_PAIRH(DIL)                                                                          \
    do {                                                                                       \
        auto kern = p.fold_w && MH == 1 ? pairh_kernel<MH, NF, NG, DIL, MH == 1> : pairh_kernel<MH, NF, NG, DIL, false>; \
        if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc; \
        hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(64 * NG), lds, s, p);                      \
    } while (0)
    switch (dil) {
        case 1: FV_PAIRH(1); break;
        case 3: FV_PAIRH(3); break;
        default: FV_PAIRH(5); break;
    }
#undef FV_PAIRH
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fv
