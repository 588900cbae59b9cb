// one-launch MRF stage, 16 channels: 12 waves x 3 fragments (576-column windows, 3 waves per SIMD)
#include "mrfh_inst.hpp"
namespace fv {
template int launch_mrfh_geom<3, 12>(const MrfParams&, hipStream_t);
}
