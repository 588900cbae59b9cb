// Tile shape 2 of conv_mfma.hip's kShapes[]: every kernel variant of this shape.
#include "conv_kernels.hpp"

namespace fv {
template int launch_geom<32, 1, 4, 1, 1>(const ConvParams&, size_t, int, hipStream_t);
template int launch_group_geom<32, 1, 4, 1, 1>(const GroupParams&, size_t, int, int, int, hipStream_t);
template int launch_sum3_geom<32, 4, 1>(const Sum3Params&, size_t, int, hipStream_t);
}  // namespace fv
