// fused ResBlock1 pairs with split-f16 operands, C = 16: FV_PAIRH16_NG (8) waves x FV_PAIRH16_NF (2) 16-column fragments
// (256-column tiles, two blocks per CU)
#include "pairh_inst.hpp"
namespace fv {
template int launch_pairh_geom<1, FV_PAIRH16_NF, FV_PAIRH16_NG>(const PairParams&, int, size_t, hipStream_t);
}
