// fused ResBlock1 pairs with split-f16 operands, C = 32: FV_PAIRH32_NG (15) waves x one 16-column fragment, both row
// halves per wave (240-column tiles; the two 11-tap weight images are 88 KB of LDS: one block per CU)
#include "pairh_inst.hpp"
namespace fv {
template int launch_pairh_geom<2, 1, FV_PAIRH32_NG>(const PairParams&, int, size_t, hipStream_t);
}
