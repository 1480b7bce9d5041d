// hidden widths 257..512 (the reference's shipped SIREN config is 8 x 512).  64-coordinate tiles, two waves per
// group of 32 coordinates (inr_mlp_wide_impl.h): 2 x 512 rows x 36 floats = 147 KB of LDS, four SIMDs busy.
#include "inr_mlp_wide_impl.h"
#include "inr_aux.h"

namespace inr {

template <int INMODE, int HACT>
static hipError_t dispatch(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st) {
  switch (mode) {
    case MODE_FWD: return launch_mlp_wide<16, INMODE, HACT, MODE_FWD>(nd, ld, a, grid, st);
    case MODE_BWD: return launch_mlp_wide<16, INMODE, HACT, MODE_BWD>(nd, ld, a, grid, st);
    default: return launch_mlp_wide<16, INMODE, HACT, MODE_FUSED>(nd, ld, a, grid, st);
  }
}

hipError_t launch_mlp_nb16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st) {
  if (nd.NB != 16 || nd.NW != 2) return hipErrorInvalidValue;
  if (nd.input == IN_GAUSS) {
    if (nd.hact == ACT_SIN) return dispatch<IN_GAUSS, ACT_SIN>(nd, ld, a, mode, grid, st);
    return dispatch<IN_GAUSS, ACT_RELU>(nd, ld, a, mode, grid, st);
  }
  if (nd.hact == ACT_SIN) return dispatch<IN_X, ACT_SIN>(nd, ld, a, mode, grid, st);
  return dispatch<IN_X, ACT_RELU>(nd, ld, a, mode, grid, st);
}

}  // namespace inr
