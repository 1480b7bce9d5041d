// inr_mfn_inst.h -- instantiates the MFN kernel for one width (INR_NB blocks) / workgroup shape.
#define INR_DW_ATTR __noinline__  // head dW passes as real functions, like dwf_pass_impl
#include "inr_mfn_impl.h"
#include "inr_aux.h"

namespace inr {

hipError_t INR_LAUNCH_NAME(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid,
                           hipStream_t st) {
  if (nd.input != IN_GAUSS && nd.input != IN_X) return hipErrorInvalidValue;
  if (nd.gabor) {
    switch (mode) {
      case MODE_FWD: return launch_mfn<INR_NB, INR_NW, MODE_FWD, true>(nd, ld, a, grid, st);
      case MODE_BWD: return launch_mfn<INR_NB, INR_NW, MODE_BWD, true>(nd, ld, a, grid, st);
      default: return launch_mfn<INR_NB, INR_NW, MODE_FUSED, true>(nd, ld, a, grid, st);
    }
  }
  switch (mode) {
    case MODE_FWD: return launch_mfn<INR_NB, INR_NW, MODE_FWD, false>(nd, ld, a, grid, st);
    case MODE_BWD: return launch_mfn<INR_NB, INR_NW, MODE_BWD, false>(nd, ld, a, grid, st);
    default: return launch_mfn<INR_NB, INR_NW, MODE_FUSED, false>(nd, ld, a, grid, st);
  }
}

}  // namespace inr
