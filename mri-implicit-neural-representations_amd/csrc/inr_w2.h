// inr_w2.h -- layouts of the bf16 path's "weights in LDS" panel stream and of its 8-bit stash (inr_siren_bf16_impl.h,
// inr_dw_gemm_bf16.hip), shared by the kernels, the packing kernel and the host.
//
// PANEL = 16 MFMA A fragments of 64 lanes x 8 bf16 = 16 KB: what one LDS-DMA ring slot holds.  The stream lists the
// panels in the order a tile of coordinates consumes them:
//   S0  layer 0 forward, contraction outermost (its 2E encoder features are generated per K-step): chunk ch = K-steps
//       4ch .. 4ch+3 x 8 row blocks = two panels, fragment (s, mo) of panel 2ch + (s >> 1) at (s & 1) * 8 + mo;
//   S1  hidden layers l = 1 .. D-2 forward, OUTPUT ROW BLOCK outermost: panel (l, m) = the 16 K-steps of row block m --
//       a block's accumulator is final after one panel, its epilogue runs while the next panel multiplies;
//   S2  last layer forward (out_features <= 4 rows: one row block): one panel, 16 K-steps;
//   S3  last layer transposed (dH_{D-2} = W_last^T dZ_last: one K-step, 8 row blocks): one panel, 8 fragments used;
//   S4  hidden layers l = D-2 .. 1 transposed (dH_{l-1} = W_l^T dZ_l), output row block (= input feature block of
//       layer l) outermost: panel (l, m).
// Constant factors ride in the images: forward images of layers 0 .. D-2 hold bf16(W * w0 / 2 pi) and their bias table
// b * w0 / 2 pi, so that an accumulator is the sine's argument in REVOLUTIONS (what v_sin_f32 takes and what the stash
// keeps); transposed images of layer l hold bf16(W_l^T * w0_{l-1}), the factor of d sin(w0 z) / dz.
#pragma once

#define W2_PANEL_BYTES 16384
#define W2_PANEL_FLOATS 4096

#if defined(__HIPCC__)
#define INR_HD __host__ __device__
#else
#define INR_HD
#endif

INR_HD inline int w2_np0(int E) { return E / 16; }                  // panels of layer 0 (2E features / 32 per panel)
INR_HD inline int w2_p_fwd(int l, int E) { return w2_np0(E) + 8 * (l - 1); }          // first panel of hidden layer l
INR_HD inline int w2_p_last(int D, int E) { return w2_np0(E) + 8 * (D - 2); }         // S2
INR_HD inline int w2_n_fwd(int D, int E) { return w2_p_last(D, E) + 1; }              // panels of a forward pass
INR_HD inline int w2_p_lastT(int D, int E) { return w2_n_fwd(D, E); }                 // S3
INR_HD inline int w2_p_T(int l, int D, int E) { return w2_p_lastT(D, E) + 1 + 8 * (D - 2 - l); }  // S4, layer l in [1, D-2]
INR_HD inline int w2_np(int D, int E) { return w2_n_fwd(D, E) + 1 + 8 * (D - 2); }    // panels per tile

// contraction index k of a 256-wide layer -> (K-step t, lane half h, element j): the order in which an MFMA
// accumulator's registers become the next MFMA's B operand (k = 32 m + 16 s + 8 (j >> 2) + 4 h + (j & 3), t = 2 m + s)
INR_HD inline void w2_kperm_inv(int k, int& t, int& h, int& j) {
  const int m = k >> 5, rem = k & 31, s = rem >> 4, rem2 = rem & 15, rem3 = rem2 & 7;
  t = 2 * m + s;
  h = rem3 >> 2;
  j = ((rem2 >> 3) << 2) | (rem3 & 3);
}
// bf16 element index of (panel p, fragment f, lane, element j)
INR_HD inline long long w2_index(int p, int f, int lane, int j) { return (((long long)p * 16 + f) * 64 + lane) * 8 + j; }

// ---- stash of one tile of TL = 128 coordinates (dword offsets) --------------------------------------------------
// Per hidden layer l (0 .. D-2) two 8-bit tensors of [256 rows][128 coordinates] in ROW-QUAD layout, one 16 KB block per
// half tile: rows 4q .. 4q+3 of coordinate c share the dword at (c >> 6) * 4096 + q * 64 + (c & 63) (a lane of the fused
// kernel holds rows 8g + 4 half + (0..3) of its coordinate in four consecutive accumulator registers: one dword store per
// quad, 128 contiguous bytes per half-wave; the GEMM kernel's stage of 64 coordinates is one contiguous block -- it reads
// 8 consecutive coordinates of a quad per thread and transposes bytes while staging):
//   P_l  phase bytes  round(256 * frac(w0 z_l / 2 pi)):  sin / cos of the layer's pre-activation to 2 pi / 256
//   G_l  dZ_l as bf8 (e5m2: the high byte of the fp16 of the same value), times the step's power-of-two gradient scale
// then dZ_last as fp16 row pairs (2 dwords per coordinate) and act'(z_last) as fp32 (4 dwords per coordinate; written by
// the forward half of a split step for its backward half).
#define W2_TL 128
#define W2_HALF 64            // coordinates per half tile = per GEMM stage
#define W2_TENSOR_DWORDS 8192
INR_HD inline int w2_stash_P(int l) { return l * W2_TENSOR_DWORDS; }
INR_HD inline int w2_stash_G(int l, int D) { return (D - 1 + l) * W2_TENSOR_DWORDS; }
INR_HD inline int w2_stash_dzl(int D) { return 2 * (D - 1) * W2_TENSOR_DWORDS; }
INR_HD inline int w2_stash_dy(int D) { return w2_stash_dzl(D) + 2 * W2_TL; }
INR_HD inline int w2_stash_dwords(int D) { return w2_stash_dy(D) + 4 * W2_TL; }

// ---- gradient-scale state (8 floats on the device, owned by the plan): [0..3] fused steps, [4..7] split steps ----
//   [0] S      power-of-two scale the NEXT fused / backward kernel applies to the loss gradient
//   [1] amax   (uint bits of) max |dZ * mult| the last kernel saw, accumulated with atomicMax
//   [2] mult   what the last kernel multiplied the loss gradient by (S / inv_count; S in split steps): the GEMM divides by it
//   [3] S_used the S inside [2]
//   [8 + 2 kind], [9 + 2 kind]  (kind 0 fused, 1 split) steps so far whose largest |dZ * mult| was past bf8's largest finite
//        value (some gradients were CLIPPED) / under W2_DZ_LOW (the bulk of the gradient fell under bf8's subnormals): the
//        scale lags the gradient by one step, and a batch whose gradient jumps by more than the headroom either way is
//        rounded coarsely -- these counters say whether that ever happened in a fit (inr_plan_grad_scale_state)
#define W2_STATE_FLOATS 16
// The next step's scale puts this step's largest |dZ * mult| into [2^(W2_DZ_TARGET_EXP - 1), 2^W2_DZ_TARGET_EXP): the ONE
// place the window is defined.  Headroom above it: log2(57344) - 5 = 10.8 binary orders before bf8 saturates.
#define W2_DZ_TARGET_EXP 5
#define W2_BF8_MAX 57344.0f
#define W2_DZ_LOW 0.015625f  // 2^-6: values 2^10 below such a maximum are under bf8's smallest subnormal 2^-16
