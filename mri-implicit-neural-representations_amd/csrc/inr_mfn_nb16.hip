// MFN, hidden widths 257..512 (BASELINE config 4: MultiscaleKFourier 8x512).  64-coordinate tiles; two waves per
// group of 32 coordinates split every GEMM by output rows (inr_mfn_wide_impl.h), so all four SIMDs of a CU run.
// 2 x (16+1) row blocks x 36 floats = 157 KB of LDS.
#define INR_DW_ATTR __noinline__
#include "inr_mfn_wide_impl.h"
#include "inr_aux.h"

namespace inr {

template <bool GABOR>
static hipError_t dispatch(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st) {
  switch (mode) {
    case MODE_FWD: return launch_mfn_wide<MODE_FWD, GABOR>(nd, ld, a, grid, st);
    case MODE_BWD: return launch_mfn_wide<MODE_BWD, GABOR>(nd, ld, a, grid, st);
    default: return launch_mfn_wide<MODE_FUSED, GABOR>(nd, ld, a, grid, st);
  }
}

hipError_t launch_mfn_nb16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st) {
  if ((nd.input != IN_GAUSS && nd.input != IN_X) || nd.NB != 16 || nd.NW != 2 || a.save == nullptr) return hipErrorInvalidValue;
  return nd.gabor ? dispatch<true>(nd, ld, a, mode, grid, st) : dispatch<false>(nd, ld, a, mode, grid, st);
}

}  // namespace inr
