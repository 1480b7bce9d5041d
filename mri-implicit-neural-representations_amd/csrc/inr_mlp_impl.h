// inr_mlp_impl.h -- the fused coordinate-MLP kernel (fp32-exact path) for gfx950.
//
// One workgroup = 4 waves = one tile of 128 coordinates; each wave owns 32 coordinates and
// carries them through EVERY layer on its own (no inter-wave traffic in the forward pass):
//
//   * activations live TRANSPOSED, X^T [features x coords], coordinates on the MFMA lanes.
//     Y^T = W . X^T makes the PyTorch-layout weight matrix the A operand and the activations the
//     B operand of v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains -> bit-level agreement with a
//     k-ordered CPU fmaf chain, 157 TFLOP/s peak);
//   * between layers the pre-activations (bias added) sit in a per-wave XOR-swizzled LDS image
//     [feature][32 coords]; the CONSUMER applies sin(w0 z) lazily while it streams the image as
//     its B operand, so the transcendental VALU work overlaps the MFMA pipe, and stashes h and
//     w0*cos(w0 z) for the backward pass;
//   * weights are streamed as pre-packed A fragments (one float4 = 4 k-steps per lane), software
//     prefetched one group ahead;
//   * backward: dX^T = W^T . dZ^T reuses the same structure (dZ = dH * act' formed lazily and
//     written back into the LDS image), then dW = dZ^T . H contracts over the tile's 128
//     coordinates with dZ read "feature on lane" from the same LDS image (the swizzle makes both
//     access directions conflict-free) and H read from the stash;
//   * each persistent workgroup accumulates its dW in a private slab (plain stores, fixed order),
//     a second kernel sums the slabs in block order: deterministic, no float atomics.
//
// Reference semantics: SirenLayer.forward / SIREN (models/networks.py:91-96,121-124), FFN
// (:48-69), Positional_Encoder 'gauss' (:30-33), autograd adjoint per SURVEY.md Appendix A.2.
#pragma once
#include "inr_device.h"
#include "inr_mlp_args.h"

namespace inr {

// ---------------------------------------------------------------------------------------------
// activations: value and derivative w.r.t. the pre-activation z
// ---------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ void act_fwd(float z, float w0, float& h, float& d) {
  if (ACT == ACT_SIN) {
    float t = w0 * z;  // torch.sin(self.w0 * x): one fp32 rounding of the product (networks.py:96)
    float s, c;
    sincos_cw(t, s, c);
    h = s;
    d = w0 * c;
  } else if (ACT == ACT_RELU) {
    h = z > 0.f ? z : 0.f;
    d = z > 0.f ? 1.f : 0.f;
  } else if (ACT == ACT_TANH) {
    h = tanhf(z);
    d = 1.f - h * h;
  } else if (ACT == ACT_SIGMOID) {
    h = 1.f / (1.f + expf(-z));
    d = h * (1.f - h);
  } else {
    h = z;
    d = 1.f;
  }
}

__device__ __forceinline__ void act_fwd_rt(int act, float z, float w0, float& h, float& d) {
  switch (act) {
    case ACT_SIN: act_fwd<ACT_SIN>(z, w0, h, d); break;
    case ACT_TANH: act_fwd<ACT_TANH>(z, w0, h, d); break;
    case ACT_RELU: act_fwd<ACT_RELU>(z, w0, h, d); break;
    case ACT_SIGMOID: act_fwd<ACT_SIGMOID>(z, w0, h, d); break;
    default: h = z; d = 1.f; break;
  }
}

// ---------------------------------------------------------------------------------------------
// A-fragment streaming: image index ((s4 * NBM + m) * 64 + lane) in float4 units
// ---------------------------------------------------------------------------------------------
template <int NBM>
__device__ __forceinline__ void load_afrag(f32x4 (&a)[NBM], const f32x4* __restrict__ p) {
#pragma unroll
  for (int m = 0; m < NBM; ++m) a[m] = p[m * 64];
}

// acc (+bias) -> this wave's LDS image rows [0, NBM*32).  Rows of register group g = r>>2 are
// 32m + 8g + 4*half + (0..3): one float4 of bias per group.  The layer's M must equal NBM*32.
template <int NBM, bool BIAS>
__device__ __forceinline__ void acc_to_lds(const f32x16 (&acc)[NBM], float* R, const float* __restrict__ bias,
                                           int lane) {
  const int half = lane >> 5, col = lane & 31;
  float* Rl = R + (4 * half) * INR_LDS_LD + col;
  const float* bl = bias + 4 * half;
#pragma unroll
  for (int m = 0; m < NBM; ++m) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (BIAS) b4 = *reinterpret_cast<const f32x4*>(bl + 32 * m + 8 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) Rl[(32 * m + 8 * g + j) * INR_LDS_LD] = acc[m][4 * g + j] + b4[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward, layer 0, fused gauss encoder (networks.py:30-33):
//   k-step s in [0,E): lane half 0 feeds sin(p_s) (feature s), half 1 feeds cos(p_s) (feature E+s)
// ---------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void fwd_layer0_gauss(f32x16 (&acc)[NB], const float* __restrict__ wp,
                                                 const float* encB_lds, int E, float xs0, float xs1, float xs2,
                                                 int lane) {
  const int half = lane >> 5;
  const f32x4* p = reinterpret_cast<const f32x4*>(wp) + lane;
  const int n4 = E >> 2;
  f32x4 a_cur[NB], a_nxt[NB];
  load_afrag<NB>(a_cur, p);
#pragma unroll 1
  for (int s4 = 0; s4 < n4; ++s4) {
    const int nx = (s4 + 1 < n4) ? (s4 + 1) : s4;
    load_afrag<NB>(a_nxt, p + (size_t)nx * NB * 64);
    __builtin_amdgcn_sched_barrier(0);  // keep the next group's loads at the top of the iteration
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = 4 * s4 + e;
      // encoder rows come from LDS (wave-uniform broadcast reads): a vector global load here
      // would drag a vmcnt(0) behind it and drain the A-fragment prefetch every k-step
      const float b0 = encB_lds[3 * s + 0], b1 = encB_lds[3 * s + 1], b2 = encB_lds[3 * s + 2];
      // (2*pi*x) @ B^T, K = 3 (networks.py:31)
      const float ph = fmaf(xs2, b2, fmaf(xs1, b1, xs0 * b0));
      float sn, cs;
      sincos_cw(ph, sn, cs);
      const float b = half ? cs : sn;
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma32(a_cur[m][e], b, acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NB; ++m) a_cur[m] = a_nxt[m];
  }
}

// forward, layer 0, input matrix x [B,K0] in memory (what model.forward receives, train.py:169)
template <int NB>
__device__ __forceinline__ void fwd_layer0_x(f32x16 (&acc)[NB], const float* __restrict__ wp,
                                             const float* __restrict__ xrow, bool valid, int K0, int Kpad8,
                                             int lane) {
  const int half = lane >> 5;
  const f32x4* p = reinterpret_cast<const f32x4*>(wp) + lane;
  const int n4 = Kpad8 >> 3;
  f32x4 a_cur[NB], a_nxt[NB];
  load_afrag<NB>(a_cur, p);
#pragma unroll 1
  for (int s4 = 0; s4 < n4; ++s4) {
    const int nx = (s4 + 1 < n4) ? (s4 + 1) : s4;
    load_afrag<NB>(a_nxt, p + (size_t)nx * NB * 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 2 * (4 * s4 + e) + half;
      float b = 0.f;
      if (valid && k < K0) b = xrow[k];
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma32(a_cur[m][e], b, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < NB; ++m) a_cur[m] = a_nxt[m];
  }
}

// ---------------------------------------------------------------------------------------------
// forward, layer l >= 1: B operand = act(z_{l-1}) formed lazily from the LDS image.
//   sv_h / sv_d: stash rows [feature][128 coords] for the backward pass (may be null).
// ---------------------------------------------------------------------------------------------
template <int NB, int NBOUT, int HACT, bool SAVE>
__device__ __forceinline__ void fwd_layer(f32x16 (&acc)[NBOUT], const float* R, const float* __restrict__ wp,
                                          float w0, float* __restrict__ sv_h, float* __restrict__ sv_d, int wcol,
                                          int lane) {
  const int half = lane >> 5, col = lane & 31;
  const f32x4* p = reinterpret_cast<const f32x4*>(wp) + lane;
  constexpr int n4 = NB * 4;  // K = 32*NB features -> 16*NB k-steps -> 4*NB groups of 4
  const float* Rl = R + half * INR_LDS_LD + col;  // row k = 2*(4*s4+e) + half
  f32x4 a_cur[NBOUT], a_nxt[NBOUT];
  float z_cur[4], z_nxt[4];
  load_afrag<NBOUT>(a_cur, p);
#pragma unroll
  for (int e = 0; e < 4; ++e) z_cur[e] = Rl[(2 * e) * INR_LDS_LD];
#pragma unroll 1
  for (int s4 = 0; s4 < n4; ++s4) {
    const int nx = (s4 + 1 < n4) ? (s4 + 1) : s4;
    load_afrag<NBOUT>(a_nxt, p + (size_t)nx * NBOUT * 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) z_nxt[e] = Rl[(8 * nx + 2 * e) * INR_LDS_LD];
    __builtin_amdgcn_sched_barrier(0);  // next group's operands are in flight behind this group's MFMAs
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 2 * (4 * s4 + e) + half;
      float h, d;
      act_fwd<HACT>(z_cur[e], w0, h, d);
      if (SAVE) {
        sv_h[k * INR_TILE + wcol] = h;
        sv_d[k * INR_TILE + wcol] = d;
      }
#pragma unroll
      for (int m = 0; m < NBOUT; ++m) acc[m] = mfma32(a_cur[m][e], h, acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NBOUT; ++m) a_cur[m] = a_nxt[m];
#pragma unroll
    for (int e = 0; e < 4; ++e) z_cur[e] = z_nxt[e];
  }
}

// ---------------------------------------------------------------------------------------------
// backward: dH_{l-1}^T = W_l^T . dZ_l^T.  R holds dH_l (sv_d != null: multiplied in place by
// act'(z_l) to give dZ_l, which dW then reads) or dZ_l itself (sv_d == null, last layer).
//   k extent = Mpad8 of layer l.
// ---------------------------------------------------------------------------------------------
template <int NB, bool HASD>
__device__ __forceinline__ void bwd_dx(f32x16 (&acc)[NB], float* R, const float* __restrict__ wpT, int Mpad8,
                                       const float* __restrict__ sv_d, int wcol, int lane) {
  const int half = lane >> 5, col = lane & 31;
  const f32x4* p = reinterpret_cast<const f32x4*>(wpT) + lane;
  const int n4 = Mpad8 >> 3;
  float* Rl = R + half * INR_LDS_LD + col;
  const float* dl = HASD ? sv_d + half * INR_TILE + wcol : nullptr;
  f32x4 a_cur[NB], a_nxt[NB];
  float g_cur[4], g_nxt[4], d_cur[4], d_nxt[4];
  load_afrag<NB>(a_cur, p);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    g_cur[e] = Rl[(2 * e) * INR_LDS_LD];
    d_cur[e] = HASD ? dl[(2 * e) * INR_TILE] : 1.f;
  }
#pragma unroll 1
  for (int s4 = 0; s4 < n4; ++s4) {
    const int nx = (s4 + 1 < n4) ? (s4 + 1) : s4;
    load_afrag<NB>(a_nxt, p + (size_t)nx * NB * 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g_nxt[e] = Rl[(8 * nx + 2 * e) * INR_LDS_LD];
      d_nxt[e] = HASD ? dl[(8 * nx + 2 * e) * INR_TILE] : 1.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g = g_cur[e];
      if (HASD) {
        g *= d_cur[e];
        Rl[(8 * s4 + 2 * e) * INR_LDS_LD] = g;  // dZ_l, read again by dW (rows of group nx are untouched)
      }
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma32(a_cur[m][e], g, acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NB; ++m) a_cur[m] = a_nxt[m];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g_cur[e] = g_nxt[e];
      d_cur[e] = d_nxt[e];
    }
  }
}

// first layer: only dZ_0 = dH_0 * act'(z_0), in place (there is no dX for the input)
__device__ __forceinline__ void bwd_dz_inplace(float* R, int nfeat, const float* __restrict__ sv_d, int wcol,
                                               int lane) {
  const int half = lane >> 5, col = lane & 31;
  for (int k = half; k < nfeat; k += 2) {
    R[swz(k, col)] *= sv_d[k * INR_TILE + wcol];
  }
}

// ---------------------------------------------------------------------------------------------
// dW pass: MT row blocks x one 32-column block n, contraction over the tile's 128 coordinates.
//   A[k=coord][i=out feature] from the four waves' LDS images (lane = feature),
//   B[k=coord][j=in feature]  from the source functor (lane = feature),
//   k order: group q of 8 coordinates -> half h takes coords 8q+4h+(0..3) as 4 k-steps.
// ---------------------------------------------------------------------------------------------
struct BSrcStash {  // h_{l-1} stash [feature][128]
  const float* __restrict__ h;
  struct Raw { f32x4 v; };
  __device__ __forceinline__ void begin(int, int) {}
  __device__ __forceinline__ Raw fetch(int n, int q, int lane) const {
    const int j = 32 * n + (lane & 31);
    return Raw{*reinterpret_cast<const f32x4*>(h + j * INR_TILE + 8 * q + 4 * (lane >> 5))};
  }
  __device__ __forceinline__ f32x4 finish(const Raw& r) const { return r.v; }
};

struct BSrcX {  // x [B,K0] row-major
  const float* __restrict__ x;
  long long row0, B;
  int K0;
  struct Raw { f32x4 v; };
  __device__ __forceinline__ void begin(int, int) {}
  __device__ __forceinline__ Raw fetch(int n, int q, int lane) const {
    const int j = 32 * n + (lane & 31);
    Raw r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long row = row0 + 8 * q + 4 * (lane >> 5) + e;
      const bool ok = j < K0 && row < B;
      const float v = x[ok ? row * K0 + j : 0];
      r.v[e] = ok ? v : 0.f;
    }
    return r;
  }
  __device__ __forceinline__ f32x4 finish(const Raw& r) const { return r.v; }
};

struct BSrcGauss {  // recompute the Fourier features of the tile's coordinates (never stored)
  const float* __restrict__ coords;
  const float* __restrict__ encB;
  long long row0, B;
  int E;
  float b0, b1, b2;
  bool is_cos;
  struct Raw { float x[4][3]; };
  __device__ __forceinline__ void begin(int n, int lane) {
    const int f = 32 * n + (lane & 31);
    is_cos = f >= E;
    const int s = is_cos ? f - E : f;
    b0 = b1 = b2 = 0.f;
    if (s < E) {  // columns >= 2E are padding of the last 32-wide block (masked on store)
      b0 = encB[3 * s + 0];
      b1 = encB[3 * s + 1];
      b2 = encB[3 * s + 2];
    }
  }
  __device__ __forceinline__ Raw fetch(int, int q, int lane) const {
    Raw r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long row = row0 + 8 * q + 4 * (lane >> 5) + e;
      const bool ok = row < B;
      const float* c = coords + (ok ? 3 * row : 0);
      const float x0 = c[0], x1 = c[1], x2 = c[2];
      r.x[e][0] = ok ? x0 : 0.f;
      r.x[e][1] = ok ? x1 : 0.f;
      r.x[e][2] = ok ? x2 : 0.f;
    }
    return r;
  }
  __device__ __forceinline__ f32x4 finish(const Raw& r) const {
    f32x4 v;
    const float two_pi = 6.283185307179586f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ph = fmaf(two_pi * r.x[e][2], b2, fmaf(two_pi * r.x[e][1], b1, (two_pi * r.x[e][0]) * b0));
      float sn, cs;
      sincos_cw(ph, sn, cs);
      v[e] = is_cos ? cs : sn;
    }
    return v;
  }
};

template <int MT, bool FULLM, class BSrc>
__device__ __forceinline__ void dw_pass(const float* Rall, int region_stride, BSrc& bsrc, int n, float* slab_w,
                                        float* slab_b, int M, int K, bool first, bool do_bias, int lane) {
  const int half = lane >> 5, li = lane & 31;
  f32x16 acc[MT];
  float bsum[MT];
  const int jcol = 32 * n + li;
  const bool colok = jcol < K;
  const int lane_off = 4 * half * K + jcol;  // rows of register r: 32m + (r&3) + 8(r>>2) + 4*half
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    bsum[m] = 0.f;
    acc[m] = zero16();
    if (!first) {  // continue this block's running sum (second and later tiles of a persistent block)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowu = 32 * m + (r & 3) + 8 * (r >> 2);  // uniform part of the row
        const bool ok = colok && (FULLM || rowu + 4 * half < M);
        const float* rowp = slab_w + (size_t)(FULLM ? rowu : 0) * K;
        const float v = rowp[ok ? (FULLM ? lane_off : rowu * K + lane_off) : 0];
        acc[m][r] = ok ? v : 0.f;
      }
    }
  }
  bsrc.begin(n, lane);
  const float* Rl = Rall + li * INR_LDS_LD + 4 * half;
  // software pipeline: operands of coordinate group q+1 are fetched while group q is multiplied
  f32x4 bv_cur = bsrc.finish(bsrc.fetch(n, 0, lane)), bv_nxt;
  float a_cur[4][MT], a_nxt[4][MT];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[e][m] = Rl[32 * m * INR_LDS_LD + e];
#pragma unroll 1
  for (int q = 0; q < INR_TILE / 8; ++q) {
    const int qn = (q + 1 < INR_TILE / 8) ? q + 1 : q;
    const typename BSrc::Raw raw = bsrc.fetch(n, qn, lane);
    // coordinate 8q + 4*half + e lives in wave image q>>2, column 8(q&3) + 4*half + e
    const float* Rq = Rl + (qn >> 2) * region_stride + 8 * (qn & 3);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < MT; ++m) a_nxt[e][m] = Rq[32 * m * INR_LDS_LD + e];
    __builtin_amdgcn_sched_barrier(0);
    bv_nxt = bsrc.finish(raw);  // ALU part (sincos for the encoder source) interleaves with the MFMAs
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        bsum[m] += a_cur[e][m];
        acc[m] = mfma32(a_cur[e][m], bv_cur[e], acc[m]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    bv_cur = bv_nxt;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < MT; ++m) a_cur[e][m] = a_nxt[e][m];
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (FULLM) {
      if (colok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* rowp = slab_w + (size_t)(32 * m + (r & 3) + 8 * (r >> 2)) * K;
          rowp[lane_off] = acc[m][r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowu = 32 * m + (r & 3) + 8 * (r >> 2);
        if (colok && rowu + 4 * half < M) slab_w[(size_t)rowu * K + lane_off] = acc[m][r];
      }
    }
    if (do_bias) {
      const float tot = bsum[m] + __shfl_xor(bsum[m], 32);
      const int row = 32 * m + li;
      if (half == 0 && (FULLM || row < M)) slab_b[row] = first ? tot : slab_b[row] + tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// the kernel.  MODE 0: forward (save optional); 1: backward from dout + save; 2: fused
// forward + pointwise loss + backward.
// ---------------------------------------------------------------------------------------------
template <int NB, int INMODE, int HACT, int MODE>
__global__ __launch_bounds__(256) void inr_mlp_kernel(const NetDesc nd, const LossDesc ld, const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int wcol = w * 32 + col;
  constexpr int RS = NB * 32 * INR_LDS_LD;  // floats per wave image
  float* R = lds + w * RS;
  float* encB_lds = lds + INR_WAVES * RS;  // [E][3] encoder matrix (gauss mode)
  if (INMODE == IN_GAUSS) {
    for (int i = tid; i < 3 * nd.E; i += 256) encB_lds[i] = a.encB[i];
    __syncthreads();
  }
  const int D = nd.D;
  const float w0 = nd.w0;
  constexpr int HSZ = NB * 32 * INR_TILE;  // floats per stashed tensor
  float* slab = (MODE != MODE_FWD) ? a.slabs + (size_t)blockIdx.x * nd.slab_floats : nullptr;
  float loss_acc = 0.f;
  bool first = true;
  const LayerDesc& LL = nd.L[D - 1];

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long row0 = (long long)tile * INR_TILE;
    const long long crow = row0 + wcol;
    const bool valid = crow < a.B;
    const bool saving = (MODE != MODE_FWD) || (a.save != nullptr);
    float* sv = a.save;
    if (saving) sv += (size_t)(a.save_by_block ? blockIdx.x : tile) * nd.save_floats_per_tile;
    float* sv_last = sv + (size_t)2 * (D - 1) * HSZ;  // [4][128]: act'(z_last) of output rows 0..3

    // ================================ forward =================================
    if (MODE != MODE_BWD) {
      {
        f32x16 acc[NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[m] = zero16();
        const LayerDesc& L0 = nd.L[0];
        if (INMODE == IN_GAUSS) {
          float x0 = 0.f, x1 = 0.f, x2 = 0.f;
          if (valid) {
            x0 = a.x[3 * crow + 0];
            x1 = a.x[3 * crow + 1];
            x2 = a.x[3 * crow + 2];
          }
          const float two_pi = 6.283185307179586f;
          fwd_layer0_gauss<NB>(acc, a.packed + L0.pf_off, encB_lds, nd.E, two_pi * x0, two_pi * x1, two_pi * x2, lane);
        } else {
          fwd_layer0_x<NB>(acc, a.packed + L0.pf_off, a.x + (size_t)(valid ? crow : 0) * L0.K, valid, L0.K, L0.Kpad8,
                           lane);
        }
        acc_to_lds<NB, true>(acc, R, a.params + L0.b_off, lane);
      }
      for (int l = 1; l < D - 1; ++l) {
        const LayerDesc& Ll = nd.L[l];
        f32x16 acc[NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[m] = zero16();
        float* sh = sv + (size_t)(2 * (l - 1)) * HSZ;
        if (saving)
          fwd_layer<NB, NB, HACT, true>(acc, R, a.packed + Ll.pf_off, w0, sh, sh + HSZ, wcol, lane);
        else
          fwd_layer<NB, NB, HACT, false>(acc, R, a.packed + Ll.pf_off, w0, nullptr, nullptr, wcol, lane);
        acc_to_lds<NB, true>(acc, R, a.params + Ll.b_off, lane);
      }
      // last layer: out_f <= 4 rows -> registers 0..3 of the lane-half-0 lanes of one row block
      f32x16 accL[1];
      accL[0] = zero16();
      {
        float* sh = sv + (size_t)(2 * (D - 2)) * HSZ;
        if (saving)
          fwd_layer<NB, 1, HACT, true>(accL, R, a.packed + LL.pf_off, w0, sh, sh + HSZ, wcol, lane);
        else
          fwd_layer<NB, 1, HACT, false>(accL, R, a.packed + LL.pf_off, w0, nullptr, nullptr, wcol, lane);
      }
      float y[4], dy[4], g[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float z = accL[0][o];
        if (o < nd.out_f) z += a.params[LL.b_off + o];
        act_fwd_rt(nd.last_act, z, w0, y[o], dy[o]);
        g[o] = 0.f;
        if (half == 0 && valid && o < nd.out_f && a.out != nullptr) a.out[crow * nd.out_f + o] = y[o];
      }
      if (MODE == MODE_FWD) {
        if (saving && half == 0) {
#pragma unroll
          for (int o = 0; o < 4; ++o) sv_last[o * INR_TILE + wcol] = dy[o];
        }
      } else {
        // fused: pointwise loss of this row (both outputs of a row sit in one half-0 lane)
        if (half == 0 && valid && (a.mask == nullptr || a.mask[crow] != 0)) {
          float t[4] = {0.f, 0.f, 0.f, 0.f};
          for (int o = 0; o < nd.out_f; ++o) t[o] = a.gt[crow * nd.out_f + o];
          loss_acc += loss_row(ld, nd.out_f, y, t, g);
        }
        // dZ_last = dY * act'(z_last) -> image rows 0..3 (half 0); rows 4..31 are zero
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.f;
          if (r < 4 && half == 0 && r < nd.out_f) v = g[r & 3] * dy[r & 3];
          R[swz(acc_row(r, half), col)] = v;
        }
      }
    }

    // ================================ backward ================================
    if (MODE != MODE_FWD) {
      if (MODE == MODE_BWD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.f;
          if (r < 4 && half == 0 && r < nd.out_f && valid)
            v = a.dout[crow * nd.out_f + r] * sv_last[r * INR_TILE + wcol];
          R[swz(acc_row(r, half), col)] = v;
        }
      }
      __syncthreads();  // every wave's dZ_last is in LDS
      // ---- last layer: dW, db from (dZ_last, h_{D-2}); dH_{D-2} = W_last^T dZ_last
      {
        BSrcStash bs{sv + (size_t)(2 * (D - 2)) * HSZ};
        for (int n = w; n < LL.Kblk; n += INR_WAVES)
          dw_pass<1, false, BSrcStash>(lds, RS, bs, n, slab + LL.w_off, slab + LL.b_off, LL.M, LL.K, first, n == 0,
                                       lane);
      }
      f32x16 gacc[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) gacc[m] = zero16();
      bwd_dx<NB, false>(gacc, R, a.packed + LL.pb_off, LL.Mpad8, nullptr, wcol, lane);
      __syncthreads();  // all dW reads of the images are done
      acc_to_lds<NB, false>(gacc, R, nullptr, lane);  // R <- dH_{D-2}

      for (int l = D - 2; l >= 1; --l) {
        const LayerDesc& Ll = nd.L[l];
#pragma unroll
        for (int m = 0; m < NB; ++m) gacc[m] = zero16();
        // dZ_l = dH_l * act'(z_l) (in place), dH_{l-1} = W_l^T dZ_l
        bwd_dx<NB, true>(gacc, R, a.packed + Ll.pb_off, Ll.Mpad8, sv + (size_t)(2 * l + 1) * HSZ, wcol, lane);
        __syncthreads();
        {
          BSrcStash bs{sv + (size_t)(2 * (l - 1)) * HSZ};
          for (int n = w; n < Ll.Kblk; n += INR_WAVES)
            dw_pass<NB, true, BSrcStash>(lds, RS, bs, n, slab + Ll.w_off, slab + Ll.b_off, Ll.M, Ll.K, first, n == 0,
                                         lane);
        }
        __syncthreads();
        acc_to_lds<NB, false>(gacc, R, nullptr, lane);
      }
      // ---- first layer: dZ_0 in place, then dW_0 against the (recomputed) input features
      {
        const LayerDesc& L0 = nd.L[0];
        bwd_dz_inplace(R, NB * 32, sv + (size_t)1 * HSZ, wcol, lane);
        __syncthreads();
        if (INMODE == IN_GAUSS) {
          BSrcGauss bs{a.x, a.encB, row0, a.B, nd.E, 0.f, 0.f, 0.f, false};
          for (int n = w; n < L0.Kblk; n += INR_WAVES)
            dw_pass<NB, true, BSrcGauss>(lds, RS, bs, n, slab + L0.w_off, slab + L0.b_off, L0.M, L0.K, first, n == 0,
                                         lane);
        } else {
          BSrcX bs{a.x, row0, a.B, L0.K};
          for (int n = w; n < L0.Kblk; n += INR_WAVES)
            dw_pass<NB, true, BSrcX>(lds, RS, bs, n, slab + L0.w_off, slab + L0.b_off, L0.M, L0.K, first, n == 0, lane);
        }
        __syncthreads();  // images are overwritten by the next tile's forward / dZ_last
      }
      first = false;
    }
  }

  if (MODE == MODE_FUSED) {
    // block loss partial -> slab word P (fixed order: wave shuffle tree, then waves 0..3 in order)
    float v = loss_acc;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    if (tid == 0) slab[nd.P] = ((lds[0] + lds[1]) + lds[2]) + lds[3];
  }
}

// hipFuncSetAttribute + launch
template <int NB, int INMODE, int HACT, int MODE>
inline hipError_t launch_mlp(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  const size_t lds_bytes = ((size_t)INR_WAVES * NB * 32 * INR_LDS_LD + 3 * (size_t)nd.E) * sizeof(float);
  auto k = inr_mlp_kernel<NB, INMODE, HACT, MODE>;
  static thread_local bool attr_set = false;
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, st, nd, ld, a);
  return hipGetLastError();
}

}  // namespace inr
