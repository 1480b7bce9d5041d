// fused SIREN training step of the bf16 throughput path, "weights in LDS" design (inr_siren_bf16_impl.h)
#include "inr_siren_bf16_impl.h"
#include "inr_aux.h"

namespace inr {

hipError_t launch_siren_bf16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  if (nd.input != IN_GAUSS || nd.hact != ACT_SIN || nd.NB != 8 || nd.w2_off < 0) return hipErrorInvalidValue;
  return launch_siren_bf16_fused(nd, ld, a, grid, st);
}

}  // namespace inr
