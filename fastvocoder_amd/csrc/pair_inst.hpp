// launch of one pair-kernel geometry: picks the dilation instantiation and the mode
#pragma once
#include "pair_kernels.hpp"

namespace fv {

template <int MH, int NF, int NG>
int launch_pair_geom(const PairParams& p, int dil, size_t lds, hipStream_t s) {
#define FV_PAIR(DIL)                                                                            \
    do {                                                                                        \
        if (p.sum) {                                                                            \
            auto kern = pair_sum_kernel<MH, NF, NG, DIL>;                                       \
            if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc; \
            hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(64 * MH * NG), lds, s, p);              \
        } else {                                                                                \
            auto kern = pair_kernel<MH, NF, NG, DIL>;                                           \
            if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc; \
            hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(64 * MH * NG), lds, s, p);              \
        }                                                                                       \
    } while (0)
    switch (dil) {
        case 1: FV_PAIR(1); break;
        case 3: FV_PAIR(3); break;
        default: FV_PAIR(5); break;
    }
#undef FV_PAIR
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fv
