// inr_mlp_rs_inst.h -- instantiates the row-split fused step for one tile width (INR_RS_NCB column blocks of 16
// coordinates) and both real hidden activations.  Included by inr_mlp_rs_n*.hip (one translation unit each).
#include "inr_mlp_rs_impl.h"
#include "inr_aux.h"

#define INR_RS_CAT2(a, b) a##b
#define INR_RS_CAT(a, b) INR_RS_CAT2(a, b)

namespace inr {

hipError_t INR_RS_CAT(launch_mlp_rs_n, INR_RS_NCB)(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid,
                                                   hipStream_t st) {
  if (nd.NB != 8 || nd.input != IN_GAUSS || nd.L[0].rf_off < 0) return hipErrorInvalidValue;
  if (nd.hact == ACT_SIN) return launch_mlp_rs<INR_RS_NCB, ACT_SIN>(nd, ld, a, grid, st);
  if (nd.hact == ACT_RELU) return launch_mlp_rs<INR_RS_NCB, ACT_RELU>(nd, ld, a, grid, st);
  return hipErrorInvalidValue;
}

}  // namespace inr
