// WIRE2D with 129..256 complex hidden features (network_width 256 is not reduced, wire2d.py:76): 512 interleaved rows.
// 64-coordinate tiles, two waves per group of 32 coordinates (inr_mlp_wide_impl.h): 2 x 512 rows x 36 floats = 147 KB.
#define INR_DW_ATTR __noinline__
#include "inr_mlp_wide_impl.h"
#include "inr_aux.h"

namespace inr {

hipError_t launch_wire2d_nb16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st) {
  if (nd.input != IN_X || nd.NB != 16 || nd.NW != 2 || nd.hact != ACT_GABOR2D || a.save == nullptr)
    return hipErrorInvalidValue;
  switch (mode) {
    case MODE_FWD: return launch_mlp_wide<16, IN_X, ACT_GABOR2D, MODE_FWD>(nd, ld, a, grid, st);
    case MODE_BWD: return launch_mlp_wide<16, IN_X, ACT_GABOR2D, MODE_BWD>(nd, ld, a, grid, st);
    default: return launch_mlp_wide<16, IN_X, ACT_GABOR2D, MODE_FUSED>(nd, ld, a, grid, st);
  }
}

}  // namespace inr
