// launch of one geometry of the one-launch MRF stage kernel (mrfh_kernels.hpp)
#pragma once
#include "mrfh_kernels.hpp"

namespace fv {

template <int NF, int NG>
int launch_mrfh_geom(const MrfParams& p, hipStream_t s) {
    typedef MrfTile<NF, NG> TL;
    auto kern = p.fold_w ? mrfh_kernel<NF, NG, 1, 3, 5, true> : mrfh_kernel<NF, NG, 1, 3, 5, false>;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), TL::LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(64 * NG), TL::LDS, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fv
