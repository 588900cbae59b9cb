// the 128-row-tile kernels of convr_kernels.hpp: convr_kernel (two-source 1x1 conv, ResidualStack's tail) and convs_kernel
// (convs with 3 / 7 / 11 taps, C = 128, 256, 512) -- one translation unit: convr_kernel is not a template
#include "convr_kernels.hpp"
namespace fv {
int launch_convr_geom(const PairParams& p, size_t lds, hipStream_t s) {
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convr_kernel), lds)) return rc;
    hipLaunchKernelGGL(convr_kernel, dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
int launch_convu_geom(const PairParams& p, size_t lds, hipStream_t s) {
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convu_kernel), lds)) return rc;
    hipLaunchKernelGGL(convu_kernel, dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
template <int DIL>
static int launch_convs_dil(const PairParams& p, size_t lds, hipStream_t s) {
    auto kern = convs_kernel<DIL>;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
int launch_convs_geom(const PairParams& p, int dil, size_t lds, hipStream_t s) {
    return dil == 1 ? launch_convs_dil<1>(p, lds, s) : dil == 3 ? launch_convs_dil<3>(p, lds, s)
         : dil == 5 ? launch_convs_dil<5>(p, lds, s) : launch_convs_dil<9>(p, lds, s);
}
}  // namespace fv
