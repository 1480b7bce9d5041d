// inr_mlp_inst.h -- instantiates the fused MLP kernel for one hidden width (INR_NB blocks of 32)
// and exposes a mode / input / activation dispatcher.  Included by inr_mlp_nb*.hip.
#include "inr_mlp_impl.h"
#include "inr_aux.h"

namespace inr {

template <int NB, int INMODE, int HACT>
static hipError_t dispatch_mode(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid,
                                hipStream_t st) {
  switch (mode) {
    case MODE_FWD: return launch_mlp<NB, INMODE, HACT, MODE_FWD>(nd, ld, a, grid, st);
    case MODE_BWD: return launch_mlp<NB, INMODE, HACT, MODE_BWD>(nd, ld, a, grid, st);
    default: return launch_mlp<NB, INMODE, HACT, MODE_FUSED>(nd, ld, a, grid, st);
  }
}

hipError_t INR_LAUNCH_NAME(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid,
                           hipStream_t st) {
  if (nd.input == IN_GAUSS) {
    if (nd.hact == ACT_SIN) return dispatch_mode<INR_NB, IN_GAUSS, ACT_SIN>(nd, ld, a, mode, grid, st);
    return dispatch_mode<INR_NB, IN_GAUSS, ACT_RELU>(nd, ld, a, mode, grid, st);
  }
  if (nd.hact == ACT_SIN) return dispatch_mode<INR_NB, IN_X, ACT_SIN>(nd, ld, a, mode, grid, st);
  return dispatch_mode<INR_NB, IN_X, ACT_RELU>(nd, ld, a, mode, grid, st);
}

}  // namespace inr
