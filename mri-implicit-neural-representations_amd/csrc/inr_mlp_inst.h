// inr_mlp_inst.h -- instantiates the fused MLP kernel for one hidden block count (INR_NB blocks of
// 32 rows), one workgroup shape (INR_NW waves) and one family, and exposes a mode / input
// dispatcher.  Included by inr_mlp_nb*.hip / inr_wire_nb*.hip (one translation unit each).
#if defined(INR_FAMILY_WIRE) || defined(INR_FAMILY_WIRE2D)
#define INR_DW_ATTR __noinline__  // WIRE kernels are register-bound: keep the dW pass out of their allocation
#endif
#include "inr_mlp_impl.h"
#include "inr_aux.h"

namespace inr {

template <int NB, int NW, int INMODE, int HACT>
static hipError_t dispatch_mode(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid,
                                hipStream_t st) {
  switch (mode) {
    case MODE_FWD: return launch_mlp<NB, NW, INMODE, HACT, MODE_FWD>(nd, ld, a, grid, st);
    case MODE_BWD: return launch_mlp<NB, NW, INMODE, HACT, MODE_BWD>(nd, ld, a, grid, st);
    default: return launch_mlp<NB, NW, INMODE, HACT, MODE_FUSED>(nd, ld, a, grid, st);
  }
}

hipError_t INR_LAUNCH_NAME(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid,
                           hipStream_t st) {
#ifdef INR_FAMILY_WIRE2D
  if (nd.input != IN_X) return hipErrorInvalidValue;
  return dispatch_mode<INR_NB, INR_NW, IN_X, ACT_GABOR2D>(nd, ld, a, mode, grid, st);
#elif defined(INR_FAMILY_WIRE)
  // WIRE takes raw coordinates (encoder.embedding: none in the reference's WIRE configs)
  if (nd.input != IN_X) return hipErrorInvalidValue;
  return dispatch_mode<INR_NB, INR_NW, IN_X, ACT_GABOR>(nd, ld, a, mode, grid, st);
#else
  if (nd.input == IN_GAUSS) {
    if (nd.hact == ACT_SIN) return dispatch_mode<INR_NB, INR_NW, IN_GAUSS, ACT_SIN>(nd, ld, a, mode, grid, st);
    return dispatch_mode<INR_NB, INR_NW, IN_GAUSS, ACT_RELU>(nd, ld, a, mode, grid, st);
  }
  if (nd.hact == ACT_SIN) return dispatch_mode<INR_NB, INR_NW, IN_X, ACT_SIN>(nd, ld, a, mode, grid, st);
  return dispatch_mode<INR_NB, INR_NW, IN_X, ACT_RELU>(nd, ld, a, mode, grid, st);
#endif
}

}  // namespace inr
