// the transposed conv for launches with few items (convtl_kernels.hpp): chunks of 128 / 64 input channels
#include "convtl_kernels.hpp"
#include "fv_internal.h"
namespace fv {
int launch_convtl_geom(const PairParams& p, int cg, hipStream_t s) {
    if (cg == 2) {
        if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convtl_kernel<2>), ConvTLGeom<2>::LDS)) return rc;
        hipLaunchKernelGGL(convtl_kernel<2>, dim3(p.nblk), dim3(512), ConvTLGeom<2>::LDS, s, p);
    } else {
        if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convtl_kernel<4>), ConvTLGeom<4>::LDS)) return rc;
        hipLaunchKernelGGL(convtl_kernel<4>, dim3(p.nblk), dim3(512), ConvTLGeom<4>::LDS, s, p);
    }
    FV_HIP(hipGetLastError());
    return 0;
}
}  // namespace fv
