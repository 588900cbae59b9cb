// split-f16 two-source 1x1 conv on 128-row tiles (ResidualStack tail): convr_kernel of convr_kernels.hpp
#include "convr_kernels.hpp"
namespace fv {
int launch_convr_geom(const PairParams& p, size_t lds, hipStream_t s) {
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convr_kernel), lds)) return rc;
    hipLaunchKernelGGL(convr_kernel, dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
}  // namespace fv
