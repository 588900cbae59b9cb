// The narrow-output conv kernels (Cout <= 4: conv_post, LastLayer).
#include "conv_kernels.hpp"

namespace fv {

int launch_narrow(const ConvParams& p, size_t lds, int grid_x, hipStream_t s) {
    dim3 grid(grid_x, p.B), block(256);
    if (p.M == 1) hipLaunchKernelGGL(conv_narrow_kernel<1>, grid, block, lds, s, p);
    else if (p.M == 2) hipLaunchKernelGGL(conv_narrow_kernel<2>, grid, block, lds, s, p);
    else hipLaunchKernelGGL(conv_narrow_kernel<4>, grid, block, lds, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "narrow conv launch: %s", hipGetErrorString(e));
    return 0;
}

int launch_post_pqmf_kernel(const ConvParams& p, const PqmfTail& q, size_t lds, int grid_x, hipStream_t s) {
    dim3 grid(grid_x, p.B), block(256);
    hipLaunchKernelGGL(conv_post_pqmf_kernel<4>, grid, block, lds, s, p, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "conv_post + pqmf launch: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace fv
