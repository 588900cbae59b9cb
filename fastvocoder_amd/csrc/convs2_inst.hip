// the ring-free 256 / 512-channel conv (convs2_kernels.hpp): per dilation, 64- and 128-column tiles
#include "convs2_kernels.hpp"
namespace fv {
template <int DIL, int NH>
static int launch_convs2_dil(const PairParams& p, size_t lds, hipStream_t s) {
    auto kern = convs2_kernel<DIL, NH>;
    if (int rc = allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(512), lds, s, p);
    FV_HIP(hipGetLastError());
    return 0;
}
int launch_convs2_geom(const PairParams& p, int dil, int nh, size_t lds, hipStream_t s) {
    if (nh == 2)
        return dil == 1 ? launch_convs2_dil<1, 2>(p, lds, s) : dil == 3 ? launch_convs2_dil<3, 2>(p, lds, s) : launch_convs2_dil<5, 2>(p, lds, s);
    return dil == 1 ? launch_convs2_dil<1, 1>(p, lds, s) : dil == 3 ? launch_convs2_dil<3, 1>(p, lds, s) : launch_convs2_dil<5, 1>(p, lds, s);
}
}  // namespace fv
