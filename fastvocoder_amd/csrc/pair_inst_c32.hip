// fused ResBlock1 pairs, C = 32: 2 row halves x 8 column groups, one 16-column fragment each (128-column tiles)
#include "pair_inst.hpp"
namespace fv {
template int launch_pair_geom<2, 1, 8>(const PairParams&, int, size_t, hipStream_t);
}
