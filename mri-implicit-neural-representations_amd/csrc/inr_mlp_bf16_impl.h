// inr_mlp_bf16_impl.h -- bf16-MFMA throughput variant of the fused SIREN kernel (gauss encoder, sin layers).
//
// Same machine mapping as inr_mlp_impl.h (one wave = 32 coordinates through every layer, activations
// transposed, per-wave fp32 LDS image of pre-activations, fp32 stash, private gradient slabs, fp32 master
// weights / Adam) with the three GEMM loops on v_mfma_f32_32x32x16_bf16 (fp32 accumulate):
//   * one MFMA consumes 16 k values; lane-half h supplies 8 of them.  k order is free as long as A and B
//     agree, so a K=16 step is simply TWO consecutive 4-k-step groups of the fp32 kernel: slot t of half h is
//     feature 16*s8 + 2t + h (natural order) / k-step 8*s8 + t of the sin (h=0) or cos (h=1) half (gauss
//     order) / coordinate 16p + 8*(t>>2) + 4h + (t&3) (dW).  The packed bf16 weight images use the same map
//     (adam_pack_kernel), everything else keeps the fp32 kernel's layouts.
//   * operands are converted on the fly (v_cvt_pk_bf16_f32); sin/cos use the hardware v_sin/v_cos (argument in
//     revolutions): with the matrix pipe 16x faster the layer loop is VALU-bound and the 25-instruction exact
//     sincos of the fp32 path would cap the gain at ~2x.
//   * the stash holds only the pre-activations z_l (copied out of the LDS image in one burst per layer): no
//     store sits inside a GEMM loop -- on gfx9 stores and loads share the in-order vmcnt queue, so a stash store per
//     k-step made every weight-fragment wait also a wait for the previous step's stores.  The backward half
//     recomputes sin / cos (two hardware transcendentals) where the fp32 kernel reads them back, and the encoder
//     features of dW_0 are regenerated from the tile's coordinates.
// This path is NOT held to the 1e-5 parity bar (bf16 operands): tests compare it with the fp32 path at bf16
// tolerances and bench.py reports PSNR next to the fp32 number.
#pragma once
#include "inr_mlp_impl.h"

namespace inr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 to_bf16x8(const float (&v)[8]) {
  f32x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = v[j];
  return __builtin_convertvector(t, bf16x8);
}

__device__ __forceinline__ bf16x8 to_bf16x8(const f32x4& a, const f32x4& b) {
  f32x8 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[j] = a[j];
    t[4 + j] = b[j];
  }
  return __builtin_convertvector(t, bf16x8);
}

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// hardware sin / cos of 2*pi*r (v_sin_f32 / v_cos_f32 take revolutions; fract keeps the argument in range)
__device__ __forceinline__ void sincos_rev(float r, float& s, float& c) {
  const float f = __builtin_amdgcn_fractf(r);
  s = __builtin_amdgcn_sinf(f);
  c = __builtin_amdgcn_cosf(f);
}

// The z stash is fp16 (10 mantissa bits: the recomputed phase w0*z is off by <= 30|z| 2^-11 ~ 0.015|z| rad, the
// size of the bf16 operand rounding already present; bf16 would triple that) -- half the bytes of the stream that,
// with the slabs, paces this kernel.
typedef _Float16 zst_t;
typedef _Float16 zst4 __attribute__((ext_vector_type(4)));

template <int NB, int TL>
__device__ __forceinline__ void image_to_zstash(const float* R, zst_t* __restrict__ G, int wcol, int lane) {
  const int half = lane >> 5, col = lane & 31;
#pragma unroll 8
  for (int r = half; r < NB * 32; r += 2) G[r * TL + wcol] = (zst_t)R[swz(r, col)];
}

__device__ __forceinline__ float sin_rev(float r) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r)); }
__device__ __forceinline__ float cos_rev(float r) { return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(r)); }

template <int NBM>
__device__ __forceinline__ void load_afrag16(bf16x8 (&a)[NBM], const bf16x8* __restrict__ p) {
#pragma unroll
  for (int m = 0; m < NBM; ++m) a[m] = p[m * 64];
}

// ---------------------------------------------------------------------------------------------
// forward, layer 0: gauss features generated on the fly, 8 k-steps (16 features) per MFMA
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gauss_feature_rev(const float* encB_lds, int s, float x0, float x1, float x2,
                                                   float quarter) {
  const float b0 = encB_lds[3 * s + 0], b1 = encB_lds[3 * s + 1], b2 = encB_lds[3 * s + 2];
  const float rev = fmaf(x2, b2, fmaf(x1, b1, fmaf(x0, b0, quarter)));  // x @ B^T (+ 1/4 turn: cos)
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev));
}

template <int NB, int TL>
__device__ __forceinline__ void fwd_layer0_gauss_bf16(f32x16 (&acc)[NB], const float* __restrict__ wp,
                                                      const float* encB_lds, int E, float x0, float x1, float x2,
                                                      int lane) {
  const int half = lane >> 5;
  const float quarter = half ? 0.25f : 0.f;
  const bf16x8* p = reinterpret_cast<const bf16x8*>(wp) + lane;
  const int n8 = E >> 3;
  bf16x8 A0[NB], A1[NB];
  float F0[8], F1[8];
  load_afrag16<NB>(A0, p);
#pragma unroll
  for (int t = 0; t < 8; ++t) F0[t] = gauss_feature_rev(encB_lds, t, x0, x1, x2, quarter);
#pragma unroll 1
  for (int s8 = 0; s8 < n8; s8 += 2) {
    const int s1 = (s8 + 1 < n8) ? s8 + 1 : s8, s2 = (s8 + 2 < n8) ? s8 + 2 : s8;
    load_afrag16<NB>(A1, p + (size_t)s1 * NB * 64);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 8; ++t) F1[t] = gauss_feature_rev(encB_lds, 8 * s1 + t, x0, x1, x2, quarter);
    {
      const bf16x8 b = to_bf16x8(F0);
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma16(A0[m], b, acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s8 + 1 < n8) {
      load_afrag16<NB>(A0, p + (size_t)s2 * NB * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 8; ++t) F0[t] = gauss_feature_rev(encB_lds, 8 * s2 + t, x0, x1, x2, quarter);
      const bf16x8 b = to_bf16x8(F1);
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma16(A1[m], b, acc[m]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward, layer l >= 1: B operand = sin(w0 z) of the LDS image rows 16*s8 + 2t + half, formed one K=16
// step ahead of the MFMAs; stash h and w0*cos (fp32) for the backward half.
// ---------------------------------------------------------------------------------------------
template <int NB, int NBOUT, int TL>
__device__ __forceinline__ void fwd_layer_bf16(f32x16 (&acc)[NBOUT], const float* R, const float* __restrict__ wp,
                                               float w0, int lane) {
  const int half = lane >> 5, col = lane & 31;
  const bf16x8* p = reinterpret_cast<const bf16x8*>(wp) + lane;
  constexpr int n8 = NB * 2;  // K = 32*NB features, 16 per step
  const float krev = w0 * 0.15915494309189535f;  // w0 / (2 pi)
  const float* Rl = R + half * INR_LDS_LD + col;
  bf16x8 A0[NBOUT], A1[NBOUT];
  float Z[8], H0[8], H1[8];
  load_afrag16<NBOUT>(A0, p);
#pragma unroll
  for (int t = 0; t < 8; ++t) Z[t] = Rl[(2 * t) * INR_LDS_LD];
#pragma unroll
  for (int t = 0; t < 8; ++t) H0[t] = sin_rev(Z[t] * krev);
#pragma unroll
  for (int t = 0; t < 8; ++t) Z[t] = Rl[(16 + 2 * t) * INR_LDS_LD];  // step 1 (n8 >= 2)
#pragma unroll 1
  for (int s8 = 0; s8 < n8; s8 += 2) {
    const int s1 = s8 + 1, s2 = (s8 + 2 < n8) ? s8 + 2 : s8, s3 = (s8 + 3 < n8) ? s8 + 3 : s8;
    // ---- step s8 multiplies while step s1's activations are formed and step s2's rows are fetched
    load_afrag16<NBOUT>(A1, p + (size_t)s1 * NBOUT * 64);
    float Zn[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) Zn[t] = Rl[(16 * s2 + 2 * t) * INR_LDS_LD];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 8; ++t) H1[t] = sin_rev(Z[t] * krev);
    {
      const bf16x8 b = to_bf16x8(H0);
#pragma unroll
      for (int m = 0; m < NBOUT; ++m) acc[m] = mfma16(A0[m], b, acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 8; ++t) Z[t] = Zn[t];
    // ---- step s1
    load_afrag16<NBOUT>(A0, p + (size_t)s2 * NBOUT * 64);
#pragma unroll
    for (int t = 0; t < 8; ++t) Zn[t] = Rl[(16 * s3 + 2 * t) * INR_LDS_LD];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 8; ++t) H0[t] = sin_rev(Z[t] * krev);
    {
      const bf16x8 b = to_bf16x8(H1);
#pragma unroll
      for (int m = 0; m < NBOUT; ++m) acc[m] = mfma16(A1[m], b, acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 8; ++t) Z[t] = Zn[t];
  }
}

// ---------------------------------------------------------------------------------------------
// backward: dH_{l-1}^T = W_l^T . dZ_l^T; HASD: dZ_l = dH_l * act'(z_l) formed on the way and written back to
// the image (the dW pass reads it).  k extent: n8 steps of 16 image rows.
// ---------------------------------------------------------------------------------------------
// HASD: sv_z = stashed z_l; act'(z) = w0 cos(w0 z) is recomputed here
template <int NB, int TL, bool HASD>
__device__ __forceinline__ void bwd_dx_bf16(f32x16 (&acc)[NB], float* R, const float* __restrict__ wpT, int n8,
                                            const zst_t* __restrict__ sv_z, float w0, int wcol, int lane) {
  const int half = lane >> 5, col = lane & 31;
  const bf16x8* p = reinterpret_cast<const bf16x8*>(wpT) + lane;
  float* Rl = R + half * INR_LDS_LD + col;
  const zst_t* dl = HASD ? sv_z + half * TL + wcol : nullptr;
  const float krev = w0 * 0.15915494309189535f;
  bf16x8 A0[NB], A1[NB];
  float G0[8], D0[8], G1[8], D1[8];
  load_afrag16<NB>(A0, p);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    G0[t] = Rl[(2 * t) * INR_LDS_LD];
    D0[t] = HASD ? (float)dl[(2 * t) * TL] : 1.f;
  }
#pragma unroll 1
  for (int s8 = 0; s8 < n8; s8 += 2) {
    const int s1 = (s8 + 1 < n8) ? s8 + 1 : s8, s2 = (s8 + 2 < n8) ? s8 + 2 : s8;
    load_afrag16<NB>(A1, p + (size_t)s1 * NB * 64);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      G1[t] = Rl[(16 * s1 + 2 * t) * INR_LDS_LD];
      D1[t] = HASD ? (float)dl[(16 * s1 + 2 * t) * TL] : 1.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      float g[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        g[t] = HASD ? G0[t] * (w0 * cos_rev(D0[t] * krev)) : G0[t];
        if (HASD) Rl[(16 * s8 + 2 * t) * INR_LDS_LD] = g[t];
      }
      const bf16x8 b = to_bf16x8(g);
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma16(A0[m], b, acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s8 + 1 < n8) {
      load_afrag16<NB>(A0, p + (size_t)s2 * NB * 64);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        G0[t] = Rl[(16 * s2 + 2 * t) * INR_LDS_LD];
        D0[t] = HASD ? (float)dl[(16 * s2 + 2 * t) * TL] : 1.f;
      }
      __builtin_amdgcn_sched_barrier(0);
      float g[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        g[t] = HASD ? G1[t] * (w0 * cos_rev(D1[t] * krev)) : G1[t];
        if (HASD) Rl[(16 * s1 + 2 * t) * INR_LDS_LD] = g[t];
      }
      const bf16x8 b = to_bf16x8(g);
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma16(A1[m], b, acc[m]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// dZ_0 = dH_0 * w0 cos(w0 z_0) -> image (the first layer has no dX); z_0 from the stash, all loads in flight
template <int NB, int TL>
__device__ __forceinline__ void acc_times_cos_to_lds(const f32x16 (&acc)[NB], float* R, const zst_t* __restrict__ sv_z,
                                                     float w0, int wcol, int lane) {
  const int half = lane >> 5, col = lane & 31;
  float* Rl = R + (4 * half) * INR_LDS_LD + col;
  const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(sv_z, NB * 32 * TL * 2);  // SGPR descriptor + one lane offset
  const int voff = ((4 * half) * TL + wcol) * 2;
  const float krev = w0 * 0.15915494309189535f;
#pragma unroll
  for (int m = 0; m < NB; ++m) {
    float z[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int soff = (32 * m + (r & 3) + 8 * (r >> 2)) * TL * 2;
      z[r] = (float)__builtin_bit_cast(zst_t, __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Rl[(32 * m + (r & 3) + 8 * (r >> 2)) * INR_LDS_LD] = acc[m][r] * (w0 * cos_rev(z[r] * krev));
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dW B operands.  h_{l-1} = sin(w0 z_{l-1}) from the z stash ("feature on lane", 4 coordinates per fetch) ...
template <int TL>
struct BSrcStashSin {
  const zst_t* __restrict__ z;
  float krev;
  struct Raw {
    zst4 v;
  };
  __device__ __forceinline__ Raw fetch(int n, int q, int lane) const {
    const int j = 32 * n + (lane & 31);
    return Raw{*reinterpret_cast<const zst4*>(z + j * TL + 8 * q + 4 * (lane >> 5))};
  }
  __device__ __forceinline__ f32x4 finish(const Raw& r) const {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = sin_rev((float)r.v[e] * krev);
    return o;
  }
};

// ... and the gauss encoder features of dW_0, regenerated from the tile's coordinates (xs [TL][3] and the encoder
// matrix are in LDS): feature j < E is sin(2 pi x.B_j), feature E + j its cosine.
struct BSrcGauss {
  const float* xs;        // LDS, [TL][3]
  const float* encB_lds;  // LDS, [E][3]
  int E;
  struct Raw {
    f32x4 v;
  };
  __device__ __forceinline__ Raw fetch(int n, int q, int lane) const {
    const int j = 32 * n + (lane & 31);
    const int s = j < E ? j : j - E;
    const float quarter = j < E ? 0.f : 0.25f;
    const float b0 = encB_lds[3 * s + 0], b1 = encB_lds[3 * s + 1], b2 = encB_lds[3 * s + 2];
    const float* x = xs + 3 * (8 * q + 4 * (lane >> 5));
    Raw r;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      r.v[e] = sin_rev(fmaf(x[3 * e + 2], b2, fmaf(x[3 * e + 1], b1, fmaf(x[3 * e], b0, quarter))));
    return r;
  }
  __device__ __forceinline__ f32x4 finish(const Raw& r) const { return r.v; }
};

// ---------------------------------------------------------------------------------------------
// dW pass: MT row blocks x one 32-column block n, contraction over the tile's TL coordinates, 16 per MFMA.
// A = dZ "feature on lane" (two ds_read_b128 per block), B = the stashed h / encoder features.
// ---------------------------------------------------------------------------------------------
// The gradient slab of a bf16 plan holds bf16 partial sums (same element offsets as the fp32 layout, so the first
// half of the slab's bytes): 258 MB of slab writes per launch were the largest single stream of a kernel that,
// with the matrix pipe 16x faster, is bound by what it writes to HBM.  Each entry is the fp32 sum over a tile's
// coordinates rounded once (2^-9 relative, far below the bf16 operand noise already in it); the block
// reduction adds the slabs in fp32.
template <int MT, int TL, bool FULLM, bool BIAS, class BSrc>
__device__ __forceinline__ void dw_pass_bf16_impl(const float* Rall, int region_stride, BSrc& bsrc, int n,
                                                  __bf16* slab_w, __bf16* slab_b, int M, int K, bool first, int lane) {
  const int half = lane >> 5, li = lane & 31;
  f32x16 acc[MT];
  float bsum[MT];
  const int jcol = 32 * n + li;
  const bool colok = jcol < K;
  const int lane_off = 4 * half * K + jcol;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    bsum[m] = 0.f;
    acc[m] = zero16();
    if (!first) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowu = 32 * m + (r & 3) + 8 * (r >> 2);
        const bool ok = colok && (FULLM || rowu + 4 * half < M);
        const __bf16* rowp = slab_w + (size_t)(FULLM ? rowu : 0) * K;
        const float v = (float)rowp[ok ? (FULLM ? lane_off : rowu * K + lane_off) : 0];
        acc[m][r] = ok ? v : 0.f;
      }
    }
  }
  const float* Rl = Rall + li * INR_LDS_LD + 4 * half;
  // pair p = coordinate groups q = 2p, 2p+1 (8 coordinates each); group q lives in wave image q>>2, columns 8(q&3)..
  f32x4 a0[MT], a1[MT], b0, b1;
#pragma unroll 1
  for (int pq = 0; pq < TL / 16; ++pq) {
    const int q0 = 2 * pq, q1 = 2 * pq + 1;
    const typename BSrc::Raw r0 = bsrc.fetch(n, q0, lane), r1 = bsrc.fetch(n, q1, lane);
    load_dw_a<MT>(a0, Rl + (q0 >> 2) * region_stride + 8 * (q0 & 3));
    load_dw_a<MT>(a1, Rl + (q1 >> 2) * region_stride + 8 * (q1 & 3));
    b0 = bsrc.finish(r0);
    b1 = bsrc.finish(r1);
    const bf16x8 b = to_bf16x8(b0, b1);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (BIAS) bsum[m] += ((a0[m][0] + a0[m][1]) + (a0[m][2] + a0[m][3])) + ((a1[m][0] + a1[m][1]) + (a1[m][2] + a1[m][3]));
      acc[m] = mfma16(to_bf16x8(a0[m], a1[m]), b, acc[m]);
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (FULLM) {
      if (colok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          __bf16* rowp = slab_w + (size_t)(32 * m + (r & 3) + 8 * (r >> 2)) * K;
          rowp[lane_off] = (__bf16)acc[m][r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowu = 32 * m + (r & 3) + 8 * (r >> 2);
        if (colok && rowu + 4 * half < M) slab_w[(size_t)rowu * K + lane_off] = (__bf16)acc[m][r];
      }
    }
    if (BIAS) {
      const float tot = bsum[m] + __shfl_xor(bsum[m], 32);
      const int row = 32 * m + li;
      if (half == 0 && (FULLM || row < M)) slab_b[row] = (__bf16)(first ? tot : (float)slab_b[row] + tot);
    }
  }
}

template <int MT, int TL, bool FULLM, class BSrc>
__device__ __forceinline__ void dw_pass_bf16(const float* Rall, int region_stride, BSrc& bsrc, int n, __bf16* slab_w,
                                             __bf16* slab_b, int M, int K, bool first, bool do_bias, int lane) {
  if (do_bias)
    dw_pass_bf16_impl<MT, TL, FULLM, true, BSrc>(Rall, region_stride, bsrc, n, slab_w, slab_b, M, K, first, lane);
  else
    dw_pass_bf16_impl<MT, TL, FULLM, false, BSrc>(Rall, region_stride, bsrc, n, slab_w, slab_b, M, K, first, lane);
}

// ---------------------------------------------------------------------------------------------
// the kernel: MODE_FWD (evaluation / first half of an unfused step), MODE_BWD (second half, from dout and the
// forward's stash) or MODE_FUSED (forward + pointwise loss + backward)
// ---------------------------------------------------------------------------------------------
template <int NB, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void inr_mlp_bf16_kernel(const NetDesc nd, const LossDesc ld, const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TL = NW * 32;
  constexpr int NS = 1;  // stashed tensors per hidden layer: z_l
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int wcol = w * 32 + col;
  constexpr int RS = NB * 32 * INR_LDS_LD;
  float* R = lds + w * RS;
  float* encB_lds = lds + NW * RS;
  float* xs_lds = encB_lds + 3 * nd.E;  // [TL][3] coordinates of the current tile (dW_0 regenerates the features)
  for (int i = tid; i < 3 * nd.E; i += NW * 64) encB_lds[i] = a.encB[i];
  __syncthreads();
  const int D = nd.D;
  constexpr int HSZ = NB * 32 * TL;
  float* slab_f = (MODE != MODE_FWD) ? a.slabs + (size_t)blockIdx.x * nd.slab_floats : nullptr;
  __bf16* slab = reinterpret_cast<__bf16*>(slab_f);  // bf16 entries at the fp32 layout's element offsets
  float loss_acc = 0.f;
  bool first = true;
  const LayerDesc& LL = nd.L[D - 1];

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long row0 = (long long)tile * TL;
    const long long crow = row0 + wcol;
    const bool valid = crow < a.B;
    float* sv = a.save + (size_t)(a.save_by_block ? blockIdx.x : tile) * nd.save_floats_per_tile;
    float* sv_last = sv + (size_t)NS * (D - 1) * HSZ;  // (the fp16 z tensors use half of their slots)
    zst_t* svz = reinterpret_cast<zst_t*>(sv);           // z_l at svz + l * HSZ
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (valid) {
      x0 = a.x[3 * crow + 0];
      x1 = a.x[3 * crow + 1];
      x2 = a.x[3 * crow + 2];
    }
    if (MODE != MODE_FWD && half == 0) {  // read by every wave's dW_0 passes, after several barriers
      xs_lds[3 * wcol + 0] = x0;
      xs_lds[3 * wcol + 1] = x1;
      xs_lds[3 * wcol + 2] = x2;
    }

    // ================================ forward =================================
    INR_STAMP(0);
    if (MODE != MODE_BWD) {
    {
      f32x16 acc[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = zero16();
      fwd_layer0_gauss_bf16<NB, TL>(acc, a.packed + nd.L[0].pf_off, encB_lds, nd.E, x0, x1, x2, lane);
      acc_to_lds<NB, true>(acc, R, a.packed + nd.L[0].pbias_off, lane);
      image_to_zstash<NB, TL>(R, svz, wcol, lane);  // z_0
    }
    INR_STAMP(1);
    for (int l = 1; l < D - 1; ++l) {
      const LayerDesc& Ll = nd.L[l];
      f32x16 acc[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = zero16();
      fwd_layer_bf16<NB, NB, TL>(acc, R, a.packed + Ll.pf_off, nd.L[l - 1].omega, lane);
      acc_to_lds<NB, true>(acc, R, a.packed + Ll.pbias_off, lane);
      image_to_zstash<NB, TL>(R, svz + (size_t)l * HSZ, wcol, lane);  // z_l
      INR_STAMP(1 + l);
    }
    f32x16 accL[1];
    accL[0] = zero16();
    fwd_layer_bf16<NB, 1, TL>(accL, R, a.packed + LL.pf_off, nd.L[D - 2].omega, lane);
    float y[4], dy[4], g[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float z = accL[0][o];
      if (o < nd.out_f) z += a.packed[LL.pbias_off + o];
      act_fwd_rt(nd.last_act, z, nd.w0, y[o], dy[o]);
      g[o] = 0.f;
      if (half == 0 && valid && o < nd.out_f && a.out != nullptr) a.out[crow * nd.out_f + o] = y[o];
    }
    if (MODE == MODE_FWD) {
      if (half == 0) {
#pragma unroll
        for (int o = 0; o < 4; ++o) sv_last[o * TL + wcol] = dy[o];  // act'(z_last) for a later MODE_BWD launch
      }
      continue;
    }

    if (half == 0 && valid && (a.mask == nullptr || a.mask[crow] != 0)) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      for (int o = 0; o < nd.out_f; ++o) t[o] = a.gt[crow * nd.out_f + o];
      loss_acc += loss_row(ld, nd.out_f, y, t, g);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = 0.f;
      if (r < 4 && half == 0 && r < nd.out_f) v = g[r & 3] * dy[r & 3];
      R[swz(acc_row(r, half), col)] = v;
    }
    } else {  // MODE_BWD: dZ_last = dout * act'(z_last) from the forward launch's stash
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = 0.f;
        if (r < 4 && half == 0 && r < nd.out_f && valid) v = a.dout[crow * nd.out_f + r] * sv_last[r * TL + wcol];
        R[swz(acc_row(r, half), col)] = v;
      }
    }

    // ================================ backward ================================
    INR_STAMP(10);
    __syncthreads();
    INR_STAMP(11);
    {
      BSrcStashSin<TL> bs{svz + (size_t)(D - 2) * HSZ, nd.L[D - 2].omega * 0.15915494309189535f};
      for (int n = w; n < LL.Kblk; n += NW)
        dw_pass_bf16<1, TL, false, BSrcStashSin<TL>>(lds, RS, bs, n, slab + LL.gw_off, slab + LL.gb_off, LL.M, LL.K, first,
                                                  n == 0, lane);
    }
    INR_STAMP(12);
    f32x16 gacc[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) gacc[m] = zero16();
    bwd_dx_bf16<NB, TL, false>(gacc, R, a.packed + LL.pb_off, 1, nullptr, 0.f, wcol, lane);  // rows 0..15 (4 used)
    __syncthreads();
    if (D == 2)
      acc_times_cos_to_lds<NB, TL>(gacc, R, svz, nd.L[0].omega, wcol, lane);
    else
      acc_to_lds<NB, false>(gacc, R, nullptr, lane);
    INR_STAMP(13);
    for (int l = D - 2; l >= 1; --l) {
      const LayerDesc& Ll = nd.L[l];
#pragma unroll
      for (int m = 0; m < NB; ++m) gacc[m] = zero16();
      bwd_dx_bf16<NB, TL, true>(gacc, R, a.packed + Ll.pb_off, NB * 2, svz + (size_t)l * HSZ, Ll.omega, wcol, lane);
      INR_STAMP(14 + 4 * l);
      __syncthreads();
      INR_STAMP(15 + 4 * l);
      {
        BSrcStashSin<TL> bs{svz + (size_t)(l - 1) * HSZ, nd.L[l - 1].omega * 0.15915494309189535f};
        for (int n = w; n < Ll.Kblk; n += NW)
          dw_pass_bf16<NB, TL, true, BSrcStashSin<TL>>(lds, RS, bs, n, slab + Ll.gw_off, slab + Ll.gb_off, Ll.M, Ll.K, first,
                                                    n == 0, lane);
      }
      INR_STAMP(16 + 4 * l);
      __syncthreads();
      if (l == 1)
        acc_times_cos_to_lds<NB, TL>(gacc, R, svz, nd.L[0].omega, wcol, lane);
      else
        acc_to_lds<NB, false>(gacc, R, nullptr, lane);
      INR_STAMP(17 + 4 * l);
    }
    {
      const LayerDesc& L0 = nd.L[0];
      __syncthreads();
      INR_STAMP(40);
      BSrcGauss bs{xs_lds, encB_lds, nd.E};
      for (int n = w; n < L0.Kblk; n += NW)
        dw_pass_bf16<NB, TL, true, BSrcGauss>(lds, RS, bs, n, slab + L0.gw_off, slab + L0.gb_off, L0.M, L0.K, first,
                                                  n == 0, lane);
      INR_STAMP(41);
      __syncthreads();
      INR_STAMP(42);
    }
    first = false;
  }

  if (MODE == MODE_FUSED) {
    float v = loss_acc;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int i = 0; i < NW; ++i) t += lds[i];
      slab_f[nd.slab_loss_off] = t;  // fp32 word behind the (half-length) bf16 region
    }
  }
}

template <int NB, int NW, int MODE>
inline hipError_t launch_mlp_bf16(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  const size_t lds_bytes = ((size_t)NW * NB * 32 * INR_LDS_LD + 3 * (size_t)nd.E + 3 * NW * 32) * sizeof(float);
  auto k = inr_mlp_bf16_kernel<NB, NW, MODE>;
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  {
    hipError_t e = allow_full_lds<inr_mlp_bf16_kernel<NB, NW, MODE>>();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds_bytes, st, nd, ld, a);
  return hipGetLastError();
}

}  // namespace inr
