// the transposed conv on resident images (convu2_kernels.hpp): one or two chunks of 128 input channels
#include "convu2_kernels.hpp"
namespace fv {
template <int NCH>
static int launch_convu2_nch(const PairParams& p, hipStream_t s) {
    typedef ConvU2Geom<NCH> G;
    auto kern = convu2_kernel<NCH>;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), G::LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(512), G::LDS, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
int launch_convu2_geom(const PairParams& p, int nch, hipStream_t s) {
    return nch == 2 ? launch_convu2_nch<2>(p, s) : launch_convu2_nch<1>(p, s);
}
}  // namespace fv
