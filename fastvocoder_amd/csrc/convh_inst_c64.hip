// split-f16 conv1d, C = 64: 64-row x 256-column tiles (a wave: 2 row sixteenths x 4 fragments)
#include "convh_inst.hpp"
namespace fv {
template int launch_convh_geom<2, 4>(const PairParams&, int, size_t, hipStream_t);
template int launch_convh_geom<2, 2>(const PairParams&, int, size_t, hipStream_t);   // 128-column tiles (A/B: FV_CONVH_NFW=2)
}
