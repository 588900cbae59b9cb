// Tile shape 0 of conv_mfma.hip's kShapes[]: every kernel variant of this shape.
#include "conv_kernels.hpp"

namespace fv {
template int launch_geom<16, 1, 4, 1, 2>(const ConvParams&, size_t, int, hipStream_t);
template int launch_group_geom<16, 1, 4, 1, 2>(const GroupParams&, size_t, int, int, int, hipStream_t);
template int launch_sum3_geom<16, 4, 2>(const Sum3Params&, size_t, int, hipStream_t);
}  // namespace fv
