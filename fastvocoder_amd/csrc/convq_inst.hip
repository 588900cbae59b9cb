// fused split-f16 ResBlock pair at 128 channels: convq_kernel of convq_kernels.hpp
#include "convq_kernels.hpp"
namespace fv {
template <int DIL>
int launch_convq_dil(const PairParams& p, size_t lds, hipStream_t s) {
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convq_kernel<DIL>), lds)) return rc;
    hipLaunchKernelGGL(convq_kernel<DIL>, dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
template int launch_convq_dil<1>(const PairParams&, size_t, hipStream_t);
template int launch_convq_dil<3>(const PairParams&, size_t, hipStream_t);
template int launch_convq_dil<5>(const PairParams&, size_t, hipStream_t);
}  // namespace fv
