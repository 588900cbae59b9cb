// split-f16 conv1d, C = 128: 64-row x 128-column tiles (a wave: 2 row sixteenths x 2 fragments), two row tiles
#include "convh_inst.hpp"
namespace fv {
template int launch_convh_geom<4, 2>(const PairParams&, int, size_t, hipStream_t);
}
