// split-f16 ConvTranspose1d (kernel = 2 x stride): convt_kernel of convh_kernels.hpp
#include "convh_kernels.hpp"
namespace fv {
// cg: 32-channel groups of an input-channel chunk -- 4 (chunks of 128 channels) or 2 (64 input channels: one chunk of 64)
int launch_convt_geom(const PairParams& p, int cg, size_t lds, hipStream_t s) {
    if (cg == 2) {
        if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convt_kernel<2>), lds)) return rc;
        hipLaunchKernelGGL(convt_kernel<2>, dim3(p.nblk), dim3(512), lds, s, p);
    } else {
        if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convt_kernel<4>), lds)) return rc;
        hipLaunchKernelGGL(convt_kernel<4>, dim3(p.nblk), dim3(512), lds, s, p);
    }
    FV_HIP(hipGetLastError());
    return 0;
}
}  // namespace fv
