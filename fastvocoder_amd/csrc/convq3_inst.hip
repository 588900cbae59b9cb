// fused split-f16 ResBlock pairs at 64 channels as two wave groups one conv phase apart: convq3_kernel of convq3_kernels.hpp
#include "convq3_kernels.hpp"
namespace fv {
template <int DIL>
int launch_convq3_dil(const PairParams& p, hipStream_t s) {
    constexpr size_t lds = ConvQ3Lds<DIL>::TOTAL;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convq3_kernel<DIL>), lds)) return rc;
    hipLaunchKernelGGL((convq3_kernel<DIL>), dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
template <int DIL>
int launch_convq4_dil(const PairParams& p, hipStream_t s) {
    constexpr size_t lds = ConvQ3Lds<DIL>::GROUP;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convq4_kernel<DIL>), lds)) return rc;
    hipLaunchKernelGGL((convq4_kernel<DIL>), dim3(2 * p.nblk), dim3(256), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
template int launch_convq4_dil<1>(const PairParams&, hipStream_t);
template int launch_convq4_dil<3>(const PairParams&, hipStream_t);
template int launch_convq4_dil<5>(const PairParams&, hipStream_t);
template int launch_convq3_dil<1>(const PairParams&, hipStream_t);
template int launch_convq3_dil<3>(const PairParams&, hipStream_t);
template int launch_convq3_dil<5>(const PairParams&, hipStream_t);
}  // namespace fv
