// the pair launches of a 64-channel MRF stage as one chained launch (convp_kernels.hpp convp_chain_kernel)
#include "convp_chain.hpp"

namespace fv {

int launch_convp_chain_kernel(const PairChain& c, int nblk, size_t lds, hipStream_t s) {
    auto kern = convp_chain_kernel;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, s, c);
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fv
