// bf16-MFMA throughput variant, hidden width 129..256 (the graded SIREN 5x256 shape), gauss encoder, sin layers
#define INR_NB 8
#define INR_NW 4
#include "inr_mlp_bf16_impl.h"
#include "inr_aux.h"

namespace inr {

hipError_t launch_mlp_nb8_bf16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid,
                               hipStream_t st) {
  if (nd.input != IN_GAUSS || nd.hact != ACT_SIN || a.save == nullptr) return hipErrorInvalidValue;
  if (mode == MODE_FWD) return launch_mlp_bf16<INR_NB, INR_NW, MODE_FWD>(nd, ld, a, grid, st);
  if (mode == MODE_FUSED) return launch_mlp_bf16<INR_NB, INR_NW, MODE_FUSED>(nd, ld, a, grid, st);
  return launch_mlp_bf16<INR_NB, INR_NW, MODE_BWD>(nd, ld, a, grid, st);
}

}  // namespace inr
