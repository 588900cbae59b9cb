// one-launch MRF stage, 16 channels: 16 waves x 2 fragments (512-column windows, 4 waves per SIMD)
#include "mrfh_inst.hpp"
namespace fv {
template int launch_mrfh_geom<2, 16>(const MrfParams&, hipStream_t);
}
