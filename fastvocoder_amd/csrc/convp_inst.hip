// fused ResBlock1 pairs at 64 channels (split-f16 operands, streamed weights): one kernel per dilation
#include "convp_kernels.hpp"

namespace fv {

template <int DIL>
int launch_convp_dil(const PairParams& p, size_t lds, hipStream_t s) {
    auto kern = convp_kernel<DIL>;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}

template int launch_convp_dil<1>(const PairParams&, size_t, hipStream_t);
template int launch_convp_dil<3>(const PairParams&, size_t, hipStream_t);
template int launch_convp_dil<5>(const PairParams&, size_t, hipStream_t);

}  // namespace fv
