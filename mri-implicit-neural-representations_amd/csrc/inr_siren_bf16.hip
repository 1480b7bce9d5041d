// SIREN on the bf16 matrix pipe, "weight panels in LDS" design (inr_siren_bf16_impl.h): fused step and the two halves
// of a split step.  One translation unit per mode (INR_BF16_MODE), six depths each.
#include "inr_siren_bf16_impl.h"
#include "inr_aux.h"

namespace inr {

#if INR_BF16_MODE == 0
hipError_t launch_siren_bf16_fwd(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  return launch_siren_bf16_mode<MODE_FWD>(nd, ld, a, grid, st);
}
#elif INR_BF16_MODE == 1
hipError_t launch_siren_bf16_bwd(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  return launch_siren_bf16_mode<MODE_BWD>(nd, ld, a, grid, st);
}
#else
hipError_t launch_siren_bf16_fused(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  return launch_siren_bf16_mode<MODE_FUSED>(nd, ld, a, grid, st);
}

hipError_t launch_siren_bf16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st) {
  if (nd.input != IN_GAUSS || nd.hact != ACT_SIN || nd.NB != 8 || nd.w2_off < 0 || nd.D < 3 || nd.D > 8 || (nd.E % 32) != 0)
    return hipErrorInvalidValue;
  if (mode == MODE_FWD) return launch_siren_bf16_fwd(nd, ld, a, grid, st);
  if (mode == MODE_BWD) return launch_siren_bf16_bwd(nd, ld, a, grid, st);
  return launch_siren_bf16_fused(nd, ld, a, grid, st);
}
#endif

}  // namespace inr
