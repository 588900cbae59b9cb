// one-launch MRF stage, 32 channels: 8 waves x 3 fragments x 2 row halves (384-column windows, 2 waves per SIMD)
#include "mrfw_kernels.hpp"

namespace fv {

int launch_mrfw_geom(const MrfParams& p, hipStream_t s) {
    typedef MrfwTile<3, 8> TL;
    static_assert(9 * TL::HSLOT == kMrfwHistBytes, "history bytes per block");
    auto kern = p.fold_w ? mrfw_kernel<3, 8, 1, 3, 5, true> : mrfw_kernel<3, 8, 1, 3, 5, false>;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), TL::LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(TL::NT), TL::LDS, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fv
