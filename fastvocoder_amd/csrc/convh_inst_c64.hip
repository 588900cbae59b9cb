// split-f16 conv1d, C = 64: 64-row x 128-column tiles (a wave: 2 row sixteenths x 2 fragments).  (256-column tiles --
// 4 fragments per wave -- were measured slower at T = 40 000, B = 1: 72 vs 65 us per ResBlock pair.)
#include "convh_inst.hpp"
namespace fv {
template int launch_convh_geom<2, 2>(const PairParams&, int, size_t, hipStream_t);
}
