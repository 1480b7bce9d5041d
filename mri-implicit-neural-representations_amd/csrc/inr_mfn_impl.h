// inr_mfn_impl.h -- fused kernel for the multiplicative filter networks of models/mfn.py
// (FourierNet :61-94, MultiscaleKFourier :206-267, MultiscaleBoundedFourier :288-355), fp32-exact.
//
//   f_i = sin(F_i x + c_i)                      x = gauss Fourier features of the coordinate (K0 = 2E)
//   h_0 = f_0,  l_i = L_{i-1} h_{i-1} + d_{i-1},  h_i = f_i * l_i,   head k: o_k = W_k h_{s_k} + b_k
//
// Same machine mapping as inr_mlp_impl.h (one wave = 32 coordinates through every stage, activations
// transposed, exact-fp32 MFMA, per-wave LDS image, private gradient slabs) with two GEMMs per stage:
//   V = L h      B operand streamed from the wave's LDS image (h of the previous stage),
//   U = F x      B operand streamed from the stash of encoder features written once in stage 0
// executed one after the other on ONE accumulator set (l is parked in the LDS image while U runs), so
// the 512-wide BASELINE shape fits the register file.  Stash per stage: f_i, l_i*cos(u_i), h_i.
// Backward: g_l = g_h*f, g_u = g_h*l*cos(u); the image keeps g_h and the dW passes
// multiply the A operand by the stashed factor on the fly, so ONE image serves dW_L, dW_F and dX.
// MultiscaleKFourier's dead stage / dead heads (A.4 #3) are neither evaluated nor updated.
#pragma once
#include "inr_mlp_impl.h"

namespace inr {

// ---------------------------------------------------------------------------------------------
// U += F . X^T with X^T rows read from the encoder-feature stash [2E rows][TL] (gauss k order:
// k-step s -> half 0: row s, half 1: row E+s), prefetched one group of 4 k-steps ahead.
// ---------------------------------------------------------------------------------------------
// Both operands come through buffer descriptors (wave-uniform base, per-lane byte offset formed once, group offset in an
// SGPR): with per-lane 64-bit pointers the loop carried two VGPR pairs that the 512-row build spilled -- a scratch
// reload with s_waitcnt vmcnt(0) at the head of every iteration, draining the prefetch it had just issued.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int NB, int TL>
__device__ __forceinline__ void stash_group(f32x16 (&acc)[NB], const f32x4 (&a_use)[NB], f32x4 (&a_load)[NB],
                                            __amdgpu_buffer_rsrc_t rw, int voff_a, int soff_a, const float (&b_use)[4],
                                            float (&b_load)[4], __amdgpu_buffer_rsrc_t rx, int voff_b, int soff_b) {
#pragma unroll
  for (int m = 0; m < NB; ++m)
    a_load[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff_a, soff_a + m * 1024, 0));
#pragma unroll
  for (int e = 0; e < 4; ++e)
    b_load[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff_b, soff_b + e * TL * 4, 0));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int m = 0; m < NB; ++m) acc[m] = mfma32(a_use[m][e], b_use[e], acc[m]);
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int NB, int TL, int NBT = NB>
__device__ __forceinline__ void gemm_enc_stash(f32x16 (&acc)[NB], const float* __restrict__ wp,
                                               const float* __restrict__ sv_enc, int E, int wcol, int lane) {
  const int half = lane >> 5;
  const int n4 = E >> 2;  // even
  const __amdgpu_buffer_rsrc_t rw = uniform_rsrc(wp, n4 * NBT * 1024);  // groups of NBT fragments of 1 KB
  const __amdgpu_buffer_rsrc_t rx = uniform_rsrc(sv_enc, 2 * E * TL * 4);
  const int voff_a = lane * 16;
  const int voff_b = ((half ? E : 0) * TL + wcol) * 4;
  f32x4 A0[NB], A1[NB];
  float B0[4], B1[4];
#pragma unroll
  for (int m = 0; m < NB; ++m)
    A0[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff_a, m * 1024, 0));
#pragma unroll
  for (int e = 0; e < 4; ++e) B0[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff_b, e * TL * 4, 0));
#pragma unroll 1
  for (int s4 = 0; s4 < n4; s4 += 2) {
    const int n2 = (s4 + 2 < n4) ? (s4 + 2) : s4;
    stash_group<NB, TL>(acc, A0, A1, rw, voff_a, (s4 + 1) * NBT * 1024, B0, B1, rx, voff_b, 4 * (s4 + 1) * TL * 4);
    stash_group<NB, TL>(acc, A1, A0, rw, voff_a, n2 * NBT * 1024, B1, B0, rx, voff_b, 4 * n2 * TL * 4);
  }
}

// INR_INPUT_X (the reference's call contract, mfn.py:34-43: forward receives the ENCODED x [B,in]): this
// coordinate's row of x goes into the feature image of the stash, from where every stage reads it exactly as it
// reads the fused encoder's output.  Lane (coordinate, half) writes rows [half E', half E' + E') -- the rows its
// own filter GEMMs read back -- and zeros past in_features (those k-steps carry zero weights).
template <int TL>
__device__ __forceinline__ void x_rows_to_stash(const float* __restrict__ x, long long crow, bool valid, int K, int E,
                                                float* __restrict__ sv_enc, int wcol, int lane) {
  const int k0 = (lane >> 5) ? E : 0;
  const float* xr = x + (size_t)(valid ? crow : 0) * K;
  float* dst = sv_enc + (size_t)k0 * TL + wcol;
  if ((K & 3) == 0) {  // rows of x are 16-byte aligned: four features per load
#pragma unroll 4
    for (int j = 0; j < E; j += 4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (valid && k0 + j < K) v = *reinterpret_cast<const f32x4*>(xr + k0 + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[(j + e) * TL] = v[e];
    }
  } else {
    for (int j = 0; j < E; ++j) dst[j * TL] = (valid && k0 + j < K) ? xr[k0 + j] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// stage epilogue for row blocks [m0, m0+MT): u = accU + c (bias image), f = sin u, fc = cos u;
//   GABOR (GaborLayer.forward, mfn.py:116-131): D = |x|^2 + |mu_j|^2 - 2 q (q = accQ = mu_j . x),
//          env = exp(-0.5 D gamma_j);  f *= env, fc *= env        (gm = [gamma | |mu|^2] image)
//   FIRST: h = f                         image <- h;  stash f, fc, h
//   else : l = image (parked), h = f*l   image <- h;  stash f, l*fc, h
// All accesses use this lane's own accumulator positions (row = 32m + (r&3) + 8(r>>2) + 4*half).
// ---------------------------------------------------------------------------------------------
template <int NB, int MT, int TL, bool FIRST, bool GABOR>
__device__ __forceinline__ void mfn_epilogue(const f32x16 (&accU)[MT], const f32x16 (&accQ)[MT], int m0, float* R,
                                             const float* __restrict__ cbias, const float* __restrict__ gm, float x2,
                                             float* __restrict__ sv, int wcol, int lane) {
  const int half = lane >> 5, col = lane & 31;
  constexpr int hsz = NB * 32 * TL;
  float* Rl = R + (32 * m0 + 4 * half) * INR_LDS_LD + col;
  // stash stores through a buffer descriptor (tensor and row offsets scalar): three per element, no 64-bit vector adds
  const __amdgpu_buffer_rsrc_t rsv = uniform_rsrc(sv, 3 * hsz * 4);
  const int voff = ((32 * m0 + 4 * half) * TL + wcol) * 4;
  const float* bl = cbias + 32 * m0 + 4 * half;
  const float* gl = GABOR ? gm + 32 * m0 + 4 * half : nullptr;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    // the row block's constants first (fetched where they are used, every float4 load got a vmcnt(0) behind it: a
    // serialized L2 round trip that also drained the stash stores in flight -- 32 of them per stage)
    f32x4 c4s[4], ga4s[4], m24s[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      c4s[g] = *reinterpret_cast<const f32x4*>(bl + 32 * m + 8 * g);
      ga4s[g] = m24s[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (GABOR) {
        ga4s[g] = *reinterpret_cast<const f32x4*>(gl + 32 * m + 8 * g);
        m24s[g] = *reinterpret_cast<const f32x4*>(gl + NB * 32 + 32 * m + 8 * g);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 c4 = c4s[g], ga4 = ga4s[g], m24 = m24s[g];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 32 * m + 8 * g + j;  // + 32*m0 + 4*half folded into Rl / svl
        float sn, cs;
        sincos_cw(accU[m][4 * g + j] + c4[j], sn, cs);
        if (GABOR) {
          const float D = (x2 + m24[j]) - 2.f * accQ[m][4 * g + j];
          const float env = expf((-0.5f * D) * ga4[j]);
          sn *= env;
          cs *= env;
        }
        float h = sn, lc = cs;
        if (!FIRST) {
          const float l = Rl[row * INR_LDS_LD];
          h = sn * l;
          lc = l * cs;
        }
        Rl[row * INR_LDS_LD] = h;
        stash_store(rsv, voff, row * TL * 4, sn);
        stash_store(rsv, voff, (hsz + row * TL) * 4, lc);
        stash_store(rsv, voff, (2 * hsz + row * TL) * 4, h);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dW pass whose A operand is (image value) * (stashed factor): g_l = g_h*f or g_u = g_h*(l cos u).
// Factor rows are read "feature on lane", 4 coordinates per float4, like the B operand.
// ---------------------------------------------------------------------------------------------
// h_{i-1} stash rows with BoundedLinear's row mask (mfn.py:281-286): coordinates whose distance to the
// k-space centre lies outside [lo, hi] contribute nothing to dW of the inner Linear (its bias still does).
template <int TL>
struct BSrcStashKeep {
  const float* __restrict__ h;
  const float* __restrict__ dist;  // [B]
  long long row0, B;
  float lo, hi;
  struct Raw {
    f32x4 v, d;
  };
  __device__ __forceinline__ Raw fetch(int n, int q, int lane) const {
    const int j = 32 * n + (lane & 31);
    Raw r;
    r.v = *reinterpret_cast<const f32x4*>(h + j * TL + 8 * q + 4 * (lane >> 5));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long row = row0 + 8 * q + 4 * (lane >> 5) + e;
      r.d[e] = dist[row < B ? row : 0];
    }
    return r;
  }
  __device__ __forceinline__ f32x4 finish(const Raw& r) const {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (r.d[e] < lo || r.d[e] > hi) ? 0.f : r.v[e];
    return o;
  }
};

template <int MT, int TL>
__device__ __forceinline__ void load_fac(f32x4 (&f)[MT], const float* __restrict__ fac, int m0, int q, int lane) {
  const int li = lane & 31;
#pragma unroll
  for (int m = 0; m < MT; ++m)
    f[m] = *reinterpret_cast<const f32x4*>(fac + (32 * (m0 + m) + li) * TL + 8 * q + 4 * (lane >> 5));
}

// T2 (Gabor centres): a second row sum weighted by |x_c|^2 of the coordinate (x2 [TL], see gabor notes below)
template <int MT, int TL, bool BIAS, bool T2, class BSrc>
__device__ __forceinline__ void dwf_group(f32x16 (&acc)[MT], float (&bsum)[MT], float (&bsum2)[MT],
                                          const f32x4 (&a_use)[MT], const f32x4 (&f_use)[MT], f32x4 (&a_load)[MT],
                                          f32x4 (&f_load)[MT], const f32x4& b_use, f32x4& b_load, const f32x4& w_use,
                                          f32x4& w_load, BSrc& bsrc, const float* fac, const float* x2, int m0, int n,
                                          int q_next, const float* Rq_next, int lane) {
  const typename BSrc::Raw raw = bsrc.fetch(n, q_next, lane);
  load_fac<MT, TL>(f_load, fac, m0, q_next, lane);
  load_dw_a<MT>(a_load, Rq_next);
  if (T2) w_load = *reinterpret_cast<const f32x4*>(x2 + 8 * q_next + 4 * (lane >> 5));
  __builtin_amdgcn_sched_barrier(0);
  b_load = bsrc.finish(raw);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float a = a_use[m][e] * f_use[m][e];
      if (BIAS) bsum[m] += a;
      if (BIAS && T2) bsum2[m] = fmaf(a, w_use[e], bsum2[m]);
      acc[m] = mfma32(a, b_use[e], acc[m]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// rows [32*m0, 32*(m0+MT)) x column block n of dW = (G*fac)^T . B over the tile's TL coordinates
template <int MT, int TL, bool BIAS, bool T2, class BSrc>
__device__ __noinline__ void dwf_pass_impl(const float* Rall, int region_stride, const float* fac, BSrc& bsrc,
                                           int m0, int n, float* slab_w, float* slab_b, float* slab_b2,
                                           const float* x2, int M, int K, bool first, int lane) {
  const int half = lane >> 5, li = lane & 31;
  f32x16 acc[MT];
  float bsum[MT], bsum2[MT];
  const int jcol = 32 * n + li;
  const bool colok = jcol < K;
  const int lane_off = 4 * half * K + jcol;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    bsum[m] = bsum2[m] = 0.f;
    acc[m] = zero16();
    if (!first) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowu = 32 * (m0 + m) + (r & 3) + 8 * (r >> 2);
        const bool ok = colok && rowu + 4 * half < M;
        const float v = slab_w[ok ? (size_t)rowu * K + lane_off : 0];
        acc[m][r] = ok ? v : 0.f;
      }
    }
  }
  const float* Rl = Rall + (32 * m0 + li) * INR_LDS_LD + 4 * half;
  f32x4 B0 = bsrc.finish(bsrc.fetch(n, 0, lane)), B1;
  f32x4 A0[MT], A1[MT], F0[MT], F1[MT];
  f32x4 W0 = {0.f, 0.f, 0.f, 0.f}, W1 = W0;
  load_dw_a<MT>(A0, Rl);
  load_fac<MT, TL>(F0, fac, m0, 0, lane);
  if (T2) W0 = *reinterpret_cast<const f32x4*>(x2 + 4 * half);
#pragma unroll 1
  for (int q = 0; q < TL / 8; q += 2) {
    const int q2 = (q + 2 < TL / 8) ? q + 2 : q;
    dwf_group<MT, TL, BIAS, T2, BSrc>(acc, bsum, bsum2, A0, F0, A1, F1, B0, B1, W0, W1, bsrc, fac, x2, m0, n, q + 1,
                                      Rl + ((q + 1) >> 2) * region_stride + 8 * ((q + 1) & 3), lane);
    dwf_group<MT, TL, BIAS, T2, BSrc>(acc, bsum, bsum2, A1, F1, A0, F0, B1, B0, W1, W0, bsrc, fac, x2, m0, n, q2,
                                      Rl + (q2 >> 2) * region_stride + 8 * (q2 & 3), lane);
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rowu = 32 * (m0 + m) + (r & 3) + 8 * (r >> 2);
      if (colok && rowu + 4 * half < M) slab_w[(size_t)rowu * K + lane_off] = acc[m][r];
    }
    if (BIAS) {
      const float tot = bsum[m] + __shfl_xor(bsum[m], 32);
      const int row = 32 * (m0 + m) + li;
      if (half == 0 && row < M) slab_b[row] = first ? tot : slab_b[row] + tot;
      if (T2) {
        const float tot2 = bsum2[m] + __shfl_xor(bsum2[m], 32);
        if (half == 0 && row < M) slab_b2[row] = first ? tot2 : slab_b2[row] + tot2;
      }
    }
  }
}

// T2: the layer is a GaborLayer centre matrix (LT_GABOR_MU): S1 = (G*fac)^T x into the dW region, its row sums
// s0 into the bias region and the |x|^2-weighted row sums T right behind them (NB*32 further)
template <int NB, int TL, bool T2, class BSrc>
__device__ __forceinline__ void dwf_layer(const float* lds, int RS, const float* fac, BSrc& bsrc, const LayerDesc& L,
                                          float* slab, const float* x2, bool first, int w, int nw, int lane) {
  constexpr int MT = NB > 8 ? 8 : NB;  // 16-block rows go in two halves (accumulators + operands <= 512 regs)
  float* sb = slab + L.gb_off;
  for (int n = w; n < L.Kblk; n += nw) {
#pragma unroll
    for (int m0 = 0; m0 < NB; m0 += MT) {
      if (n == 0)
        dwf_pass_impl<MT, TL, true, T2, BSrc>(lds, RS, fac, bsrc, m0, n, slab + L.gw_off, sb, sb + NB * 32, x2, L.M, L.K,
                                              first, lane);
      else
        dwf_pass_impl<MT, TL, false, false, BSrc>(lds, RS, fac, bsrc, m0, n, slab + L.gw_off, sb, sb, x2, L.M, L.K,
                                                  first, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wide stages in register-sized chunks: the 16 row blocks of the 512-wide shape are accumulated 8 at a
// time (the B operand is streamed twice), so a GEMM phase holds 256 accumulator + 64 A-fragment
// registers instead of 256 + 128 and stays inside the 512-register file without scratch.
// ---------------------------------------------------------------------------------------------
template <int MT, int NB>
__device__ __forceinline__ f32x16 (&chunk(f32x16 (&acc)[NB], int m0))[MT] {
  return *reinterpret_cast<f32x16(*)[MT]>(&acc[m0]);
}

// acc += A[image at wpT] . (image R)  for all NB row blocks;  HASD: R <- R * sv_d in place on the way
template <int NB, int TL, bool HASD>
__device__ __forceinline__ void gemm_image(f32x16 (&acc)[NB], float* R, const float* __restrict__ wpT, int Kpad8,
                                           const float* __restrict__ sv_d, int wcol, int lane) {
  constexpr int MT = NB > 8 ? 8 : NB;
  bwd_dx<MT, TL, false, HASD, NB>(chunk<MT, NB>(acc, 0), R, wpT, Kpad8, sv_d, wcol, lane);
#pragma unroll
  for (int m0 = MT; m0 < NB; m0 += MT)  // R already holds the products: plain second pass
    bwd_dx<MT, TL, false, false, NB>(chunk<MT, NB>(acc, m0), R, wpT + (size_t)m0 * 256, Kpad8, nullptr, wcol, lane);
}

template <int NB, int TL>
__device__ __forceinline__ void gemm_enc(f32x16 (&acc)[NB], const float* __restrict__ wp,
                                         const float* __restrict__ sv_enc, int E, int wcol, int lane) {
  constexpr int MT = NB > 8 ? 8 : NB;
#pragma unroll
  for (int m0 = 0; m0 < NB; m0 += MT)
    gemm_enc_stash<MT, TL, NB>(chunk<MT, NB>(acc, m0), wp + (size_t)m0 * 256, sv_enc, E, wcol, lane);
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// Head outputs y[k][o] and their gradients g[k][o] of a coordinate live in rows 8 + 4k + o of the wave's
// head-gradient image (own column), not in registers: the head code then exists once, indexed by the
// (wave-uniform) head number.  Rows >= 8 of that image are never consumed: the head dW pass keeps rows
// < out_features only, and dX = W_head^T g contracts over rows 0..7.
template <int NB, int NW, int MODE, bool GABOR>
__global__ __launch_bounds__(NW * 64) void inr_mfn_kernel(const NetDesc nd, const LossDesc ld, const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TL = NW * 32;
  constexpr int RS = NB * 32 * INR_LDS_LD;   // floats per wave image
  constexpr int HS = 32 * INR_LDS_LD;        // floats per wave head-gradient image
  constexpr int HSZ = NB * 32 * TL;          // floats per stashed tensor
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int wcol = w * 32 + col;
  float* R = lds + w * RS;
  float* HGall = lds + NW * RS;
  float* HG = HGall + w * HS;
  float* encB_lds = HGall + NW * HS;
  const bool xin = nd.input == IN_X;  // x [B,in] in memory instead of the fused gauss encoder
  if (!xin) {
    for (int i = tid; i < 3 * nd.E; i += NW * 64) encB_lds[i] = a.encB[i];
    __syncthreads();
  }
  const int n = nd.mfn_n, S = nd.mfn_stages, NH = nd.n_heads;
  const LayerDesc* Fl = nd.L;           // filters 0..n
  const LayerDesc* Ll = nd.L + n + 1;   // linears 0..n-1
  const LayerDesc* Ml = nd.L + nd.mu0;  // GABOR: (mu_i, gamma_i) of filter i
  constexpr int MT = NB > 8 ? 8 : NB;   // row blocks per register-sized chunk
  float* slab = (MODE != MODE_FWD) ? a.slabs + (size_t)blockIdx.x * nd.slab_floats : nullptr;
  float loss_acc = 0.f;
  bool first = a.accumulate == 0;  // accumulate: a follow-up launch of the same step (inr_api.hip, split launches)

  for (int tile = a.tile0 + blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long row0 = (long long)tile * TL;
    const long long crow = row0 + wcol;
    const bool valid = crow < a.B;
    const bool saving = (MODE != MODE_FWD) || (a.save != nullptr);
    float* sv = a.save;
    if (saving) sv += (size_t)(a.save_by_block ? blockIdx.x : tile) * nd.save_floats_per_tile;
    // stash: stage i -> [f_i | l_i cos u_i | h_i] ; then encoder features; without a save buffer the
    // encoder features still need a home for stages >= 1: the caller always passes one for MFN
    float* sv_enc = sv + (size_t)3 * S * HSZ;
    float* sv_x2 = sv_enc + (size_t)Fl[0].Kblk * 32 * TL;  // GABOR: |x_c|^2 of the tile's coordinates [TL]
    float x2 = 0.f;
    float keep = 1.f;  // bounded linears: per-coordinate 0/1 (recomputed per stage)
    const float dist = (a.dist != nullptr && valid) ? a.dist[crow] : 0.f;
    // ================================ forward =================================
    if (MODE != MODE_BWD) {
      {
        float x0 = 0.f, x1 = 0.f, x2c = 0.f;
        if (valid && !xin) {
          x0 = a.x[3 * crow + 0];
          x1 = a.x[3 * crow + 1];
          x2c = a.x[3 * crow + 2];
        }
        if (xin) x_rows_to_stash<TL>(a.x, crow, valid, Fl[0].K, nd.E, sv_enc, wcol, lane);
        const float two_pi = 6.283185307179586f;
#pragma unroll
        for (int m0 = 0; m0 < NB; m0 += MT) {
          f32x16 accU[MT], accQ[GABOR ? MT : 1];
#pragma unroll
          for (int m = 0; m < MT; ++m) accU[m] = zero16();
          if (m0 == 0 && !xin)  // generates the encoder features and leaves them in the stash for everything after
            fwd_layer0_gauss<MT, TL, true, NB>(accU, a.packed + Fl[0].pf_off, encB_lds, nd.E, two_pi * x0, two_pi * x1,
                                               two_pi * x2c, sv_enc, wcol, lane);
          else
            gemm_enc_stash<MT, TL, NB>(accU, a.packed + Fl[0].pf_off + (size_t)m0 * 256, sv_enc, nd.E, wcol, lane);
          if (GABOR) {
            if (m0 == 0) {  // |x_c|^2 = sum of squared features (mfn.py:125), this lane's half then both
              const float* svl = sv_enc + (half ? nd.E : 0) * TL + wcol;
              float acc2 = 0.f;
              for (int e = 0; e < nd.E; ++e) {
                const float v = svl[e * TL];
                acc2 = fmaf(v, v, acc2);
              }
              x2 = acc2 + __shfl_xor(acc2, 32);
              if (half == 0) sv_x2[wcol] = x2;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) accQ[m] = zero16();
            gemm_enc_stash<MT, TL, NB>(reinterpret_cast<f32x16(&)[MT]>(accQ), a.packed + Ml[0].pf_off + (size_t)m0 * 256,
                                       sv_enc, nd.E, wcol, lane);
            mfn_epilogue<NB, MT, TL, true, true>(accU, reinterpret_cast<f32x16(&)[MT]>(accQ), m0, R,
                                                 a.packed + Fl[0].pbias_off, a.packed + Ml[0].pbias_off, x2, sv, wcol,
                                                 lane);
          } else {
            mfn_epilogue<NB, MT, TL, true, false>(accU, accU, m0, R, a.packed + Fl[0].pbias_off, nullptr, 0.f, sv, wcol,
                                                  lane);
          }
        }
      }
#pragma unroll 1
      for (int i = 1; i < S; ++i) {
        f32x16 acc[NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[m] = zero16();
        if (nd.bounded) {  // BoundedLinear: rows outside [lo,hi] are zeroed before the Linear (mfn.py:281-286)
          keep = (dist < nd.bound_lo[i - 1] || dist > nd.bound_hi[i - 1]) ? 0.f : 1.f;
          if (keep == 0.f) {
            for (int r = half; r < NB * 32; r += 2) R[swz(r, col)] = 0.f;  // own column only
          }
        }
        gemm_image<NB, TL, false>(acc, R, a.packed + Ll[i - 1].pf_off, NB * 32, nullptr, wcol, lane);  // V = L h
        acc_to_lds<NB, true>(acc, R, a.packed + Ll[i - 1].pbias_off, lane);                               // park l
#pragma unroll
        for (int m0 = 0; m0 < NB; m0 += MT) {  // U = F x (and Q = mu x), epilogue, one register-sized chunk at a time
          f32x16 accU[MT], accQ[GABOR ? MT : 1];
#pragma unroll
          for (int m = 0; m < MT; ++m) accU[m] = zero16();
          gemm_enc_stash<MT, TL, NB>(accU, a.packed + Fl[i].pf_off + (size_t)m0 * 256, sv_enc, nd.E, wcol, lane);
          if (GABOR) {
            f32x16 (&q)[MT] = reinterpret_cast<f32x16(&)[MT]>(accQ);
#pragma unroll
            for (int m = 0; m < MT; ++m) q[m] = zero16();
            gemm_enc_stash<MT, TL, NB>(q, a.packed + Ml[i].pf_off + (size_t)m0 * 256, sv_enc, nd.E, wcol, lane);
            mfn_epilogue<NB, MT, TL, false, true>(accU, q, m0, R, a.packed + Fl[i].pbias_off,
                                                  a.packed + Ml[i].pbias_off, x2, sv + (size_t)3 * i * HSZ, wcol, lane);
          } else {
            mfn_epilogue<NB, MT, TL, false, false>(accU, accU, m0, R, a.packed + Fl[i].pbias_off, nullptr, 0.f,
                                                   sv + (size_t)3 * i * HSZ, wcol, lane);
          }
        }
        // head fed by this stage, if any (output_layers are distinct stages)
        int kh = -1;
        for (int k = 0; k < NH; ++k)
          if (nd.head_stage[k] == i) kh = k;
        if (kh >= 0) {
          const LayerDesc& Hd = nd.L[nd.head_layer[kh]];
          f32x16 accH[1];
          accH[0] = zero16();
          bwd_dx<1, TL, false, false>(accH, R, a.packed + Hd.pf_off, NB * 32, nullptr, wcol, lane);
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            float z = accH[0][o];  // half 0: rows 0..3 = output o
            if (o < nd.out_f) z += a.packed[Hd.pbias_off + o];
            if (half == 0) {
              HG[swz(8 + 4 * kh + o, col)] = z;
              if (valid && o < nd.out_f && a.out != nullptr) a.out[((size_t)kh * a.B + crow) * nd.out_f + o] = z;
            }
          }
        }
      }
      if (MODE == MODE_FUSED && half == 0) {
        float y[INR_MAX_HEADS][4], g[INR_MAX_HEADS][4];
#pragma unroll
        for (int k = 0; k < INR_MAX_HEADS; ++k)
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            y[k][o] = k < NH ? HG[swz(8 + 4 * k + o, col)] : 0.f;
            g[k][o] = 0.f;
          }
        if (valid) {
          float t[4] = {0.f, 0.f, 0.f, 0.f};
          for (int o = 0; o < nd.out_f; ++o) t[o] = a.gt[crow * nd.out_f + o];
          loss_acc += mfn_loss_row(ld, NH, nd.out_f, y, t, dist, g, a.mask == nullptr || a.mask[crow] != 0);
        }
#pragma unroll
        for (int k = 0; k < INR_MAX_HEADS; ++k)
#pragma unroll
          for (int o = 0; o < 4; ++o)
            if (k < NH) HG[swz(8 + 4 * k + o, col)] = g[k][o];
      }
    }

    // ================================ backward ================================
    if (MODE != MODE_FWD) {
      f32x16 gacc[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) gacc[m] = zero16();
      if (GABOR && MODE == MODE_BWD) x2 = 0.f;  // the passes read sv_x2 themselves
#pragma unroll 1
      for (int i = S - 1; i >= 1; --i) {
        float* svi = sv + (size_t)3 * i * HSZ;
        // ---- heads fed by stage i: dW_head, and their contribution W_k^T g_k to g_h_i
        // ---- head fed by stage i (if any): its contribution W_k^T g_k to g_h_i now, its dW below, once the
        //      accumulators are parked (a call with 16 live accumulator blocks would spill all of them)
        int kh = -1;
        for (int k = 0; k < NH; ++k)
          if (nd.head_stage[k] == i) kh = k;
        if (kh >= 0) {
          const LayerDesc& Hd = nd.L[nd.head_layer[kh]];
          float gv[4];
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            if (MODE == MODE_BWD)
              gv[o] = (half == 0 && valid && o < nd.out_f) ? a.dout[((size_t)kh * a.B + crow) * nd.out_f + o] : 0.f;
            else
              gv[o] = (half == 0 && o < nd.out_f) ? HG[swz(8 + 4 * kh + o, col)] : 0.f;
          }
          __syncthreads();  // previous readers of the head-gradient images (the last head's dW) are done
#pragma unroll
          for (int o = 0; o < 4; ++o) HG[swz(o + 4 * half, col)] = gv[o];  // rows 0..3 = g, rows 4..7 = 0
          gemm_image<NB, TL, false>(gacc, HG, a.packed + Hd.pb_off, Hd.Mpad8, nullptr, wcol, lane);  // own image only
        }
        __syncthreads();  // readers of the main images (previous stage's dW / dX) are done
        acc_to_lds<NB, false>(gacc, R, nullptr, lane);  // image <- g_h_i
        __syncthreads();
        if (kh >= 0) {
          const LayerDesc& Hd = nd.L[nd.head_layer[kh]];
          if (Hd.M <= 4 && (TL == 64 || TL == 128)) {  // head rows on the vector ALUs (inr_mlp_impl.h)
            dw_rows4_valu<(TL == 128 ? 128 : 64), NW>(HGall, HS, svi + (size_t)2 * HSZ, NB * 32, Hd.M, Hd.K,
                                                      slab + Hd.gw_off, slab + Hd.gb_off, first, w, lane);
          } else {
            BSrcStash<TL> bs{svi + (size_t)2 * HSZ};  // h_i
            for (int nn = w; nn < Hd.Kblk; nn += NW)
              dw_pass<1, TL, false, BSrcStash<TL>>(HGall, HS, bs, nn, slab + Hd.gw_off, slab + Hd.gb_off, Hd.M, Hd.K,
                                                   first, nn == 0, lane);
          }
        }
        // ---- dL_{i-1} = (g_h*f_i)^T h_{i-1} (+ db), dF_i = (g_h*l_i cos u_i)^T x (+ dc)
        {
          if (nd.bounded) {
            BSrcStashKeep<TL> bh{sv + (size_t)(3 * (i - 1) + 2) * HSZ, a.dist, row0, a.B, nd.bound_lo[i - 1],
                                 nd.bound_hi[i - 1]};
            dwf_layer<NB, TL, false, BSrcStashKeep<TL>>(lds, RS, svi, bh, Ll[i - 1], slab, nullptr, first, w, NW, lane);
          } else {
            BSrcStash<TL> bh{sv + (size_t)(3 * (i - 1) + 2) * HSZ};
            dwf_layer<NB, TL, false, BSrcStash<TL>>(lds, RS, svi, bh, Ll[i - 1], slab, nullptr, first, w, NW, lane);
          }
          BSrcStash<TL> bx{sv_enc};
          dwf_layer<NB, TL, false, BSrcStash<TL>>(lds, RS, svi + HSZ, bx, Fl[i], slab, nullptr, first, w, NW, lane);
          // GABOR: S1_i = (g_h * h_i)^T x, s0_i, T_i  ->  d mu_i, d gamma_i in the finishing kernel
          if (GABOR) dwf_layer<NB, TL, true, BSrcStash<TL>>(lds, RS, svi + 2 * HSZ, bx, Ml[i], slab, sv_x2, first, w, NW, lane);
        }
        __syncthreads();
        // ---- g_h_{i-1} = L_{i-1}^T (g_h_i * f_i)   (in place on the image, which is dead afterwards)
#pragma unroll
        for (int m = 0; m < NB; ++m) gacc[m] = zero16();
        gemm_image<NB, TL, true>(gacc, R, a.packed + Ll[i - 1].pb_off, NB * 32, svi, wcol, lane);
        if (nd.bounded) {
          keep = (dist < nd.bound_lo[i - 1] || dist > nd.bound_hi[i - 1]) ? 0.f : 1.f;
#pragma unroll
          for (int m = 0; m < NB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) gacc[m][r] *= keep;
        }
      }
      // ---- stage 0: dF_0 = (g_h_0 * cos u_0)^T x
      __syncthreads();
      acc_to_lds<NB, false>(gacc, R, nullptr, lane);
      __syncthreads();
      {
        BSrcStash<TL> bx{sv_enc};
        dwf_layer<NB, TL, false, BSrcStash<TL>>(lds, RS, sv + HSZ, bx, Fl[0], slab, nullptr, first, w, NW, lane);
        if (GABOR) dwf_layer<NB, TL, true, BSrcStash<TL>>(lds, RS, sv + 2 * HSZ, bx, Ml[0], slab, sv_x2, first, w, NW, lane);
      }
      __syncthreads();
      first = false;
    }
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
      slab[nd.slab_loss_off] = a.accumulate ? slab[nd.slab_loss_off] + t : t;
    }
  }
}

template <int NB, int NW, int MODE, bool GABOR>
inline hipError_t launch_mfn(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  const size_t lds_bytes = ((size_t)NW * (NB + 1) * 32 * INR_LDS_LD + (nd.input == IN_GAUSS ? 3 * (size_t)nd.E : 0)) * sizeof(float);
  auto k = inr_mfn_kernel<NB, NW, MODE, GABOR>;
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  {
    hipError_t e = allow_full_lds<inr_mfn_kernel<NB, NW, MODE, GABOR>>();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds_bytes, st, nd, ld, a);
  return hipGetLastError();
}

}  // namespace inr
