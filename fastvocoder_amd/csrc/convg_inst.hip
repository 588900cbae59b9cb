// split-f16 two-source 1x1 conv (ResidualStack tail): convg_kernel of convh_kernels.hpp
#include "convh_kernels.hpp"
namespace fv {
int launch_convg_geom(const PairParams& p, size_t lds, hipStream_t s) {
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convg_kernel<4>), lds)) return rc;
    hipLaunchKernelGGL(convg_kernel<4>, dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
}  // namespace fv
