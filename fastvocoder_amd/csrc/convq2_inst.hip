// fused split-f16 ResBlock pairs at 128 / 64 channels, weights from L2 straight into registers: convq2_kernel of convq2_kernels.hpp
#include "convq2_kernels.hpp"
namespace fv {
template <int DIL, int C>
int launch_convq2_dil(const PairParams& p, size_t lds, hipStream_t s) {
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(convq2_kernel<DIL, C>), lds)) return rc;
    hipLaunchKernelGGL((convq2_kernel<DIL, C>), dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
template int launch_convq2_dil<1, 128>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<3, 128>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<5, 128>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<1, 64>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<3, 64>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<5, 64>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<1, kPair64Wide>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<3, kPair64Wide>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<5, kPair64Wide>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<1, kPair128Wide>(const PairParams&, size_t, hipStream_t);
template int launch_convq2_dil<3, kPair128Wide>(const PairParams&, size_t, hipStream_t);
}  // namespace fv
