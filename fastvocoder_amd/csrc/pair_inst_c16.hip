// fused ResBlock1 pairs, C = 16: 8 waves x two 16-column fragments (256-column tiles)
#include "pair_inst.hpp"
namespace fv {
template int launch_pair_geom<1, 2, 8>(const PairParams&, int, size_t, hipStream_t);
}
