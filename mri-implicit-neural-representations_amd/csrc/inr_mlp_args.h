// inr_mlp_args.h -- launch arguments of the fused MLP kernel (shared by the API and kernel TUs)
#pragma once
#include "inr_device.h"

namespace inr {

#define MODE_FWD 0
#define MODE_BWD 1
#define MODE_FUSED 2

struct MlpArgs {
  const float* params;
  const float* packed;
  const float* x;      // [B,K0] (IN_X) or coords [B,3] (IN_GAUSS)
  const float* encB;   // [E,3]
  const float* gt;     // [B,out_f]   (fused)
  const uint8_t* mask; // [B] or null (fused)
  const float* dout;   // [B,out_f]   (bwd); MFN: [n_heads][B,out_f]
  const float* dist;   // [B] distance to the k-space centre (multiscale consistency / bounded linears)
  float* out;          // [B,out_f]   (fwd; fused writes it when non-null); MFN: [n_heads][B,out_f]
  float* save;         // stash
  float* slabs;        // [grid][slab_floats]
  long long B;
  int n_tiles;
  int save_by_block;   // 1: stash slot = blockIdx (fused); 0: slot = tile
  int dw_gemm;         // 1: leave dZ_l of the hidden-width layers in the stash for inr_dw_gemm.hip, skip their dW passes
  long long* dbg;      // diagnostic builds (-DINR_STAMPS) only: per-wave phase time stamps
  long long dbg_cap;   // entries behind dbg (a stamp past it is dropped)
  int ll_lds;          // inr_mlp_kernel: the last layer's live A fragments (rows 0..3) sit in LDS (set by launch_mlp)
  int dz_lds;    // set by the launcher: dZ_last has its own LDS image (fused step; inr_mlp_impl.h)
  float* dz_state;     // bf16 plans: the plan's gradient-scale state (inr_w2.h), W2_STATE_FLOATS floats on the device
  int tile0;           // first tile of this launch (tiles [tile0, n_tiles)); accumulate: the workgroups' slabs and loss
  int accumulate;      // words already hold an earlier launch's sums of the same step -- add to them
  // row-split fused step (inr_mlp_rs_impl.h): tile t = round * grid + workgroup owns rs_hi column blocks of 16
  // coordinates if t < rs_x, else rs_lo, blocks in tile order; n_tiles = 128-coordinate stash slots
  int rs_hi, rs_lo, rs_x, rs_rounds;
};


}  // namespace inr
