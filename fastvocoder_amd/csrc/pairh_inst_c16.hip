// fused ResBlock1 pairs with split-f16 operands, C = 16: 4 waves x four 16-column fragments (256-column tiles)
#include "pairh_inst.hpp"
namespace fv {
template int launch_pairh_geom<1, 4, 4>(const PairParams&, int, size_t, hipStream_t);
}
