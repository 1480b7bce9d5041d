// inr_device.h -- shared device-side definitions for the gfx950 INR engine.
// Written for CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x2_f32, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "inr_launch.h"

#define INR_MAX_LAYERS 32
#define INR_MAX_HEADS 4
#define INR_MAX_WAVES 4   // waves per workgroup (tile = 32 coordinates per wave)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Activation codes (must match enum inr_act in include/inr_abi.h)
#define ACT_ID 0
#define ACT_SIN 1
#define ACT_TANH 2
#define ACT_RELU 3
#define ACT_SIGMOID 4
#define ACT_GABOR 5   // WIRE complex Gabor wavelet on interleaved (Re, Im) rows; hidden layers only
#define ACT_CTANH 7   // last layer only: Re(tanh(a + jb)) of a complex output (WIRE2D last_tanh, wire2d.py:106-107,113-117)
#define ACT_GABOR2D 6 // WIRE2D (wire2d.py:49-60): the same with a second Linear `scale_orth` per layer feeding the
                      // Gaussian window; L[orth0 + l] describes that Linear of layer l

// how a layer's virtual real matrix [M x K] maps to the flat parameters
#define LT_REAL 0          // weight [M,K] f32, bias [M]                       (SIREN / FFN)
#define LT_WIRE_FIRST 1    // weight [M/2,K] f32 -> rows 2i, rows 2i+1 are zero (networks.py:185-188)
#define LT_WIRE_HIDDEN 2   // weight [M/2,K/2] complex64 -> [[Wr,-Wi],[Wi,Wr]] interleaved
#define LT_WIRE_LAST 3     // weight [M,K/2] complex64, output = real part: row o = [Wr, -Wi] interleaved
#define LT_GABOR_MU 4      // GaborLayer centres: "weight" = mu [M,K], "bias" = gamma [M] (mfn.py:106-111); packed
                           // bias image = [gamma | ||mu_j||^2], slab = [S1 (M x K) | s0 (NB*32) | T (NB*32)]

#define IN_X 0
#define IN_GAUSS 1

#define LOSS_L2_HALF 0
#define LOSS_L1_HALF 1
#define LOSS_TANH 2
#define LOSS_LOGSPACE 3
#define LOSS_HDR 4
#define LOSS_MSLE_HALF 5
#define LOSS_CENTER 6

#include "inr_stamp_rt.h"

struct LayerDesc {
  int K, M;          // in / out features of the (virtual) real matrix the kernel multiplies
  int Kpad8;         // K rounded up to a multiple of 8 (4 k-steps of 2 per A-fragment float4)
  int Kblk;          // ceil(K / 32): 32-wide column blocks of dW
  int Mblk;          // ceil(M / 32): 32-row blocks
  int Mpad8;         // M rounded up to a multiple of 8 (k extent of the transposed product)
  int ltype;         // LT_*
  int w_off, b_off;  // offsets (floats) into flat params / grads
  int wn, bn;        // floats of the weight / bias tensors in flat params
  int gw_off, gb_off;// offsets into a gradient slab: dW [M x K] and db [M] of the virtual matrix
  int pf_off;        // offset into packed: forward image  A[i=out][k=in]
  int pb_off;        // offset into packed: transposed image A[i=in][k=out] (unused for layer 0)
  int pbias_off;     // offset into packed: bias image, Mblk*32 entries, zero padded
  float omega, s0;   // activation constants of this layer's OUTPUT (SIREN w0 / WIRE omega_0, scale_0)
  int live;          // 0: dead layer of MultiscaleKFourier (never gets a gradient; Adam skips it)
  int korder;        // k order of the forward image: 0 natural (k = 2s+half), 1 gauss split (half ? E+s : s)
  int rf_off, rb_off;// row-split plans (inr_mlp_rs_impl.h): offsets into packed of the 16x16x4 fragment images, forward
                     // A[i=out][k=in] and transposed A[i=in][k=out]; -1: none
};

struct NetDesc {
  int D;             // number of Linear layers the network chains (MFN: all descriptors)
  int ND;            // number of LayerDesc entries in L[] (== D except WIRE2D: D + D - 1)
  int orth0;         // WIRE2D: L[orth0 + l] = scale_orth of layer l (0 <= l < D - 1)
  int rs;            // 1: fused steps run the row-split kernel (inr_mlp_rs_impl.h): L[l].rf_off / rb_off are valid
  int bf16;          // 1: bf16 throughput path: the packed images are the panel stream of inr_siren_bf16_impl.h
  int w2_off;        // bf16 plans: offset (floats) into packed of the "weight panels in LDS" stream (inr_w2.h); -1: none
  int w2_bias_off;   // ... and of its fp32 bias table [D][256]
  int NB;            // hidden width / 32
  int hact;          // hidden activation (ACT_SIN / ACT_RELU)
  int last_act;
  int input;         // IN_X / IN_GAUSS
  int E;             // gauss encoder size (in_features == 2E)
  int out_f;
  float w0;
  int P;             // total params (floats in flat params)
  int NW;            // waves per workgroup: tile = 32*NW coordinates
  int slab_floats;   // floats per gradient slab (virtual dW/db of every layer + loss word, padded to 64)
  int slab_loss_off; // position of the block's loss partial inside its slab
  int save_floats_per_tile;
  // multiplicative filter networks (models/mfn.py): L[] = filters 0..n | linears 0..n-1 | heads
  int mfn_n;                      // network_depth n (n+1 filters, n linears)
  int mfn_stages;                 // stages actually evaluated: 1 + last stage that feeds a head
  int n_heads;                    // 1 (FourierNet) or 4 (multiscale)
  int head_stage[INR_MAX_HEADS];  // stage whose h feeds head k
  int head_layer[INR_MAX_HEADS];  // index into L[] of head k
  int gabor;                      // GaborNet / KGaborNet: L[mu0 + i] describes (mu_i, gamma_i) of filter i
  int mu0;
  int bounded;                    // MultiscaleBoundedFourier: linears see h only where lo <= dist <= hi
  float bound_lo[INR_MAX_LAYERS / 2], bound_hi[INR_MAX_LAYERS / 2];  // per linear
  LayerDesc L[INR_MAX_LAYERS];
};

struct LossDesc {
  int kind;
  float eps, sigma, factor, inv_count, hdr_A;
  float scale;        // multiplies the pointwise loss (0.5 for the multiscale loop's 0.5*loss_fn, else 1)
  // ConsistencyLoss between consecutive heads (metrics/losses.py:315-324), multiscale only
  float cons_w;       // 0.1 (train_kspace_multiscale.py:179); 0 disables
  int cons_chan;      // channels compared: 2 (dist [B]) or 1 (per-coil dist [B,1]: channel 0 only, A.4 #4)
  float cons_lo[INR_MAX_HEADS], cons_hi[INR_MAX_HEADS];
  float cons_inv[INR_MAX_HEADS];  // 1 / (number of compared elements of pair i, over all ranks); 0: empty
};

// Row of a 32x32 MFMA accumulator held in register r by lane-half h (guide section 3:
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)).
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Padded [feature][33] LDS image of one wave's 32 coordinates: conflict-free both for
// "lane = coord" accesses (forward / dX B operand, epilogue stores) and for "lane = feature"
// reads (dW A operand), and every address is  lane-part + compile-time immediate.
// Row stride (floats) of the image.  36 where the workgroup's images fit 160 KB (16-byte aligned rows:
// the dW A operand is then ONE ds_read_b128 per 4 k-steps, conflict-free because 36*i mod 64 walks all
// 16 four-bank slots); 33 for the 12-block WIRE shape (3 x 384 x 36 x 4 B would be 166 KB).
#ifndef INR_LDS_LD
#define INR_LDS_LD 36
#endif
__device__ __forceinline__ int swz(int feat, int col) { return feat * INR_LDS_LD + col; }

// sincos for the fp32 parity path, branch-free (the OCML sincosf carries a Payne-Hanek slow path whose branches
// would split every MFMA k-step into basic blocks).
//
// Default: the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) behind a two-constant Cody-Waite
// reduction by 2 pi: k = rint(x / 2pi), r = x - k 2pi in two FMAs (2pi = hi - lo with hi = float(2pi)), r / 2pi in
// [-0.5, 0.5].  Measured on MI355X against double precision (tools/probes/sin_probe.hip): max abs error 3.8e-7,
// mean 5.5e-8 for |x| <= 300 rad -- 7 VALU issue slots + 2 transcendentals, against 26 VALU for the polynomial
// form below, and on the fp32 MFMA path every VALU instruction is time added to the matrix pipe's.
// -DINR_SINCOS_POLY: three-constant Cody-Waite by pi/2 + cephes minimax polynomials on [-pi/4, pi/4] (max abs
// error 9e-8, mean 1.6e-8): kept for A/B runs (make poly).
#ifndef INR_SINCOS_POLY
// Reduction in REVOLUTIONS (what v_sin / v_cos take), two-constant 1/(2 pi): k = rint(x * c_hi); x * c_hi - k is exact
// inside the FMA up to its one rounding at magnitude <= 1/2 (3e-8), then + x * c_lo (c_hi + c_lo = 1/(2 pi) to 1e-16).
// Four VALU instructions; round 2's first form reduced in radians and multiplied afterwards (five).
__device__ __forceinline__ void sincos_cw(float x, float& sn, float& cs) {
  constexpr float c_hi = 0.15915494309189535f;
  constexpr float c_lo = (float)(0.15915494309189533576888 - (double)c_hi);
  const float k = rintf(x * c_hi);
  const float rev = fmaf(x, c_lo, fmaf(x, c_hi, -k));
  sn = __builtin_amdgcn_sinf(rev);
  cs = __builtin_amdgcn_cosf(rev);
}
#else
__device__ __forceinline__ void sincos_cw(float x, float& sn, float& cs) {
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(k, -1.5707963705062866f, x);      // pi/2 hi
  r = fmaf(k, 4.3711388286737929e-8f, r);          // -(pi/2 mid)
  r = fmaf(k, 1.7151245100058e-15f, r);            // -(pi/2 lo)
  const float s = r * r;
  float ps = fmaf(s, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, s, -1.6666654611e-1f);
  const float sr = fmaf(ps * s, r, r);
  float pc = fmaf(s, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, s, 4.166664568298827e-2f);
  const float cr = fmaf(pc * s, s, fmaf(-0.5f, s, 1.0f));
  const int q = (int)k;
  const float a = (q & 1) ? cr : sr;
  const float b = (q & 1) ? sr : cr;
  sn = (q & 2) ? -a : a;
  cs = ((q + 1) & 2) ? -b : b;
}
#endif

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

namespace inr {
// ---------------------------------------------------------------------------------------------
// pointwise losses (metrics/losses.py).  y,t: the row's outputs / targets.
// Returns the row's loss contribution; g[] = d(loss)/d(y).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float loss_row(const LossDesc& ld, int out_f, const float* y, const float* t, float* g) {
  const float inv = ld.inv_count;
  float loss = 0.f;
  if (ld.kind == LOSS_L2_HALF) {  // 0.5 * mean((y - t)^2) over rows*out_f (train.py:82,182)
    const float s = inv / (float)out_f;
    for (int o = 0; o < out_f; ++o) {
      const float e = y[o] - t[o];
      loss += 0.5f * e * e * s;
      g[o] = e * s;
    }
  } else if (ld.kind == LOSS_L1_HALF) {  // 0.5 * mean(|y - t|)
    const float s = 0.5f * inv / (float)out_f;
    for (int o = 0; o < out_f; ++o) {
      const float e = y[o] - t[o];
      loss += fabsf(e) * s;
      g[o] = (e > 0.f ? s : (e < 0.f ? -s : 0.f));
    }
  } else if (ld.kind == LOSS_TANH) {  // TanhL2Loss (losses.py:130-139)
    const float s = inv / (float)out_f;
    for (int o = 0; o < out_f; ++o) {
      const float ty = tanhf(y[o]);
      const float d = ty - tanhf(t[o]);
      loss += d * d * s;
      g[o] = 2.f * d * (1.f - ty * ty) * s;
    }
  } else if (ld.kind == LOSS_LOGSPACE) {  // LogSpaceLoss (losses.py:214-223)
    const float er = y[0] - t[0], ei = y[1] - t[1];
    const float den = sqrtf(y[0] * y[0] + y[1] * y[1]) + ld.eps;
    const float q = inv / (den * den);
    loss = (er * er + ei * ei) * q;
    g[0] = 2.f * er * q;
    g[1] = 2.f * ei * q;
  } else if (ld.kind == LOSS_MSLE_HALF) {  // 0.5 * MSLELoss (losses.py:18-27; train.py:84,182): MSE of log(. + 1 + 1e-9)
    const float s = inv / (float)out_f;
    for (int o = 0; o < out_f; ++o) {
      const float ay = (y[o] + 1.f) + 1e-9f;
      const float e = logf(ay) - logf((t[o] + 1.f) + 1e-9f);
      loss += 0.5f * e * e * s;
      g[o] = e / ay * s;
    }
  } else if (ld.kind == LOSS_CENTER) {
    // CenterLoss, pointwise part (losses.py:157-173,201): 0.1 error_loss.mean() + 0.9 (abs_loss.mean() + reg.mean()) with
    // error_loss == abs_loss == (|y - t| / (|y|_detached + eps))^2 and reg the same [B,B] broadcast as HDRLoss_FF's
    // (separable: factor * A * |y|^2 / den^2).  The random-pair term is inr_center_pairs_grad.
    const float er = y[0] - t[0], ei = y[1] - t[1];
    const float ya2 = y[0] * y[0] + y[1] * y[1];
    const float den = sqrtf(ya2) + ld.eps;
    const float q = inv / (den * den);
    const float rq = 0.9f * ld.factor * ld.hdr_A * q;
    loss = (er * er + ei * ei) * q + rq * ya2;
    g[0] = 2.f * er * q + 2.f * rq * y[0];
    g[1] = 2.f * ei * q + 2.f * rq * y[1];
  } else {  // LOSS_HDR: HDRLoss_FF, separable form (losses.py:236-264)
    const float er = y[0] - t[0], ei = y[1] - t[1];
    const float ya2 = y[0] * y[0] + y[1] * y[1];
    const float den = sqrtf(ya2) + ld.eps;
    const float ea2 = er * er + ei * ei;
    const float lg = logf(sqrtf(ea2) / den);
    const float rq = ld.factor * ld.hdr_A / (den * den);
    loss = (lg * lg + rq * ya2) * inv;
    const float c1 = 2.f * lg / ea2 * inv, c2 = 2.f * rq * inv;
    g[0] = c1 * er + c2 * y[0];
    g[1] = c1 * ei + c2 * y[1];
  }
  return loss;
}

// ---------------------------------------------------------------------------------------------
// multiscale loss of one coordinate (train_kspace_multiscale.py:164-195): sum over heads of
// scale * loss_fn(o_k, gt) on SAMPLED rows (:176-182: out[mask], gt[mask]) + cons_w * ConsistencyLoss
// (losses.py:315-324) on every row.  y[k][o] in, g[k][o] out.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float mfn_loss_row(const LossDesc& ld, int n_heads, int out_f, const float (&y)[INR_MAX_HEADS][4],
                                              const float* t, float dist, float (&g)[INR_MAX_HEADS][4], bool sampled) {
  float loss = 0.f;
#pragma unroll
  for (int k = 0; k < INR_MAX_HEADS; ++k) {
    if (k < n_heads && sampled) {
      float gk[4] = {0.f, 0.f, 0.f, 0.f};
      loss += ld.scale * loss_row(ld, out_f, y[k], t, gk);
#pragma unroll
      for (int o = 0; o < 4; ++o) g[k][o] = ld.scale * gk[o];
    }
  }
  if (ld.cons_w != 0.f) {
#pragma unroll
    for (int i = 0; i + 1 < INR_MAX_HEADS; ++i) {
      if (i + 1 < n_heads && ld.cons_inv[i] != 0.f && (dist < ld.cons_lo[i] || dist > ld.cons_hi[i])) {
        for (int o = 0; o < ld.cons_chan; ++o) {
          const float e = y[i + 1][o] - y[i][o];  // first tensor is detached: gradient to head i+1 only
          loss += ld.cons_w * e * e * ld.cons_inv[i];
          g[i + 1][o] += ld.cons_w * 2.f * e * ld.cons_inv[i];
        }
      }
    }
  }
  return loss;
}


}  // namespace inr
