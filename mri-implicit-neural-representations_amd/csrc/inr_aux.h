// inr_aux.h -- host-side launchers shared between the translation units of libinr_mi355x.so
#pragma once
#include "inr_device.h"
#include "inr_mlp_args.h"

namespace inr {

struct AdamArgs {
  int do_update;      // 0: pack only
  int all_real;       // set by the launcher: every layer is LT_REAL (fast scatter path)
  int has_dead;       // set by the launcher (adam_dead_ranges): some layer has live == 0 ...
  int n_dead;         // ... and its flat entries are these merged ranges [dead_lo, dead_hi); -1: more than fit, walk the layers
  int dead_lo[8], dead_hi[8];
  float step_size;    // lr / (1 - beta1^t), computed in double on the host like torch does
  float bc2_sqrt;     // sqrt(1 - beta2^t)
  float omb1;         // float(1 - beta1): the lerp weight torch passes to exp_avg.lerp_
  float beta2, omb2;  // float(beta2), float(1 - beta2)
  float eps, weight_decay, l1, l2;
  // graph-replayable form (inr_adam_step_dev): the step is read from device memory and indexes a table of
  // (step_size, bc2_sqrt) pairs, so that no kernel argument changes from one step to the next
  const float* sched;   // nullptr: use step_size / bc2_sqrt above
  const int* step_dev;  // steps taken so far
  int n_sched;
};


// entries of the layers in `mask` (bit l = L[l]) and the flat gradient entries [lo, hi) are summed over the n2 slabs
// that FOLLOW the n_blocks slabs of the fused kernel (partial sums of the batch-level weight-gradient GEMM);
// n2 == 0: one slab set
// ... and, of those, the flat entries [lo3, hi3) over n3 slabs of the same set instead of n2 (the bf16 GEMM gives its
// first-layer units, whose operand is arithmetic rather than a load, shorter chunks -- more of them; n3 == 0: none)
struct SlabSplit {
  int lo, hi, n2;
  unsigned mask;
  int lo3 = 0, hi3 = 0, n3 = 0;
};
hipError_t launch_reduce_slabs(const NetDesc& nd, const float* slabs, int n_blocks, float* grads, float* loss_out,
                               const float* params, const float* packed, hipStream_t st,
                               SlabSplit split = SlabSplit{0, 0, 0, 0});
hipError_t launch_step_advance(int* step_dev, hipStream_t st);
hipError_t launch_reduce_slabs_adam(const NetDesc& nd, const float* slabs, int n_blocks, float* grads, float* loss_out,
                                    float* params, float* m1, float* m2, float* packed, const AdamArgs& aa,
                                    hipStream_t st, SlabSplit split);
hipError_t launch_reg_grad(const NetDesc& nd, const float* params, float* grads, int lo, int hi, float l1, float l2,
                           const float* l2_dir, hipStream_t st);
hipError_t launch_adam_shard(const NetDesc& nd, float* params, const float* grads_shard, float* m1, float* m2, int lo,
                             int hi, const AdamArgs& aa, hipStream_t st);
hipError_t launch_adam_pack(const NetDesc& nd, float* params, const float* grads, float* m1, float* m2,
                            float* packed, const AdamArgs& aa, hipStream_t st);
hipError_t launch_encode_logf(const float* coords, const float* bands, long long B, int nb, float* out,
                              hipStream_t st);
hipError_t launch_encode_gauss(const float* coords, const float* encB, long long B, int E, float* out,
                               hipStream_t st);
hipError_t launch_loss_grad(const LossDesc& ld, const float* out, const float* gt, const float* kcoords,
                            const uint8_t* mask, long long B, float* loss_out, float* dout, hipStream_t st);
hipError_t launch_loss_grad_multi(const LossDesc& ld, const float* outs, const float* gt, const float* dist,
                                  const uint8_t* mask, int NH, long long B, float* loss_out, float* douts,
                                  hipStream_t st);
// CenterLoss random-pair term (losses.py:175-199): adds w * sum_p r_p^2 to loss_out[0] and its gradient to dout
hipError_t launch_center_pairs(const float* out, const float* gt, const long long* ia, const long long* ib, long long n,
                               long long B, float w, float* loss_out, float* dout, hipStream_t st);
hipError_t launch_loss_tv_grad(const LossDesc& ld, const float* out, const float* gt, const uint8_t* mask, long long R,
                               long long R_own, long long W, float cw, float ch, float* loss_out, float* dout,
                               hipStream_t st);
hipError_t launch_tv_grad(const float* out, long long R, long long R_own, long long W, float cw, float ch,
                          float* loss_out, float* dout, hipStream_t st);

// per-NB dispatchers (one translation unit each): mode 0 fwd, 1 bwd, 2 fused
hipError_t launch_mlp_nb1(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_mlp_nb2(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_mlp_nb4(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_siren_bf16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_siren_bf16_fwd(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_siren_bf16_bwd(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_siren_bf16_fused(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_mlp_nb16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
// row-split fused step, tiles of N column blocks of 16 coordinates (inr_mlp_rs_n*.hip)
hipError_t launch_mlp_rs_n1(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_mlp_rs_n2(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_mlp_rs_n3(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_mlp_rs_n4(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_mlp_rs_n5(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_mlp_rs_n6(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_mlp_rs_n7(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st);
hipError_t launch_mlp_nb8(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_wire_nb2(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_wire_nb4(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_wire_nb8(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_wire_nb12(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_wire2d_nb2(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_wire2d_nb4(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_wire2d_nb8(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_wire2d_nb16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_mfn_nb1(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_mfn_nb4(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_mfn_nb8(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);
hipError_t launch_mfn_nb16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int mode, int grid, hipStream_t st);

}  // namespace inr
