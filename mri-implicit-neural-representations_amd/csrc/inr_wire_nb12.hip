// WIRE width 256 -> int(256/sqrt(2)) = 181 complex hidden features = 362 interleaved real rows, padded to 384 =
// 12 blocks.  64-coordinate tiles, two waves per group of 32 coordinates (inr_mlp_wide_impl.h): 2 x 384 rows x 36
// floats = 110 KB of LDS, all four SIMDs busy, and 25 000 coordinates = 391 half-length tiles instead of 261.
#define INR_DW_ATTR __noinline__
#include "inr_mlp_wide_impl.h"
#include "inr_aux.h"

namespace inr {

hipError_t launch_wire_nb12(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st) {
  if (nd.input != IN_X || nd.NB != 12 || nd.NW != 2 || nd.hact != ACT_GABOR) return hipErrorInvalidValue;
  switch (mode) {
    case MODE_FWD: return launch_mlp_wide<12, IN_X, ACT_GABOR, MODE_FWD>(nd, ld, a, grid, st);
    case MODE_BWD: return launch_mlp_wide<12, IN_X, ACT_GABOR, MODE_BWD>(nd, ld, a, grid, st);
    default: return launch_mlp_wide<12, IN_X, ACT_GABOR, MODE_FUSED>(nd, ld, a, grid, st);
  }
}

}  // namespace inr
