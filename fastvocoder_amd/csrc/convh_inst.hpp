// launch of one split-f16 wide-channel conv geometry: picks the dilation instantiation
#pragma once
#include "convh_kernels.hpp"

namespace fv {

template <int CG, int NFW>
int launch_convh_geom(const PairParams& p, int dil, size_t lds, hipStream_t s) {
#define FV_CONVH(DIL)                                                                          \
    do {                                                                                       \
        auto kern = convh_kernel<CG, NFW, DIL>;                                                \
        if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc; \
        hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(512), lds, s, p);                          \
    } while (0)
    switch (dil) {
        case 1: FV_CONVH(1); break;
        case 3: FV_CONVH(3); break;
        case 5: FV_CONVH(5); break;
        default: FV_CONVH(9); break;
    }
#undef FV_CONVH
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fv
