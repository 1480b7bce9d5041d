// inr_mlp_wide_impl.h -- SIREN / FFN with 257..512 hidden features (16 row blocks; the reference's shipped
// config/remote/config_siren_kspace.yaml is 8 x 512) and WIRE with 129..192 complex features (12 blocks of
// interleaved rows; network_width 256 -> 181) with TWO waves per group of 32 coordinates.
//
// Same idea as inr_mfn_wide_impl.h: a 512 x 32 fp32 image is 73.7 KB, a workgroup holds two, and with one wave per
// image only two SIMDs of the CU work.  Waves (g, hh) = (wave >> 1, wave & 1) share the image of coordinate group
// g and split every GEMM by output rows (row blocks [8 hh, 8 hh + 8)).  Both waves of a pair form the lazy
// activation of all 512 input rows (VALU work doubles, ~10% of the MFMA time at this width), wave hh = 0 writes
// the stash; dZ = dH * act' becomes a separate owner-rows pass; the last layer (one row block) runs on hh = 0.
// Arithmetic, stash and slab layouts are those of inr_mlp_impl.h.
#pragma once
#include "inr_mlp_impl.h"

namespace inr {

// owner-rows pass dZ = dH * act'(z) on image rows [r0, r0 + RH) of this wave's coordinate column.
// PAIR (WIRE): rows (2i, 2i+1) hold (p, q) = dL/d(y_r, y_i); dZ[row] = p * dA[row] + q * dB[row]
// JAC (WIRE, eager activation): the stash holds y (dA argument) and the pre-activations z = a + j b (dB argument) of
// every complex feature, and the four Jacobian entries of the Gabor wavelet are formed here from them -- the same
// expressions, in the same order, that round 1 evaluated in the forward epilogue and stashed (six stores per feature
// instead of four: the epilogues run at the CU's share of the HBM write rate).
// JAC == 2 (WIRE2D, second Linear of a layer): dB argument = the stashed orth rows (u, v); d y / d o = -2 s0^2 o y.
template <int TL, int RH, bool PAIR, int JAC = 0>
__device__ __forceinline__ void rows_times(float* R, const float* __restrict__ dA, const float* __restrict__ dB, int r0,
                                           int wcol, int lane, float omega = 0.f, float s0 = 0.f) {
  static_assert(JAC == 0 || PAIR, "the Jacobian forms are the complex ones");
  const float s2 = s0 * s0;
  const int half = lane >> 5, col = lane & 31;
  // The Jacobian entries come from the stash (HBM / L2): with four rows in flight per lane the pass ran at one memory
  // latency per four rows (phase stamps: 35 k cycles per layer at 192 rows per wave).  Batches of BT row steps are
  // requested through buffer descriptors (row offset in an SGPR) before the first is used; nothing else is live here.
  constexpr int STEP = PAIR ? 4 : 2;           // rows a lane advances per step (both halves together)
  constexpr int NSTEP = RH / STEP;
  constexpr int BMAX = PAIR ? 16 : 32;         // <= 64 loads in flight per lane
  constexpr int BT = NSTEP % BMAX == 0 ? BMAX : (NSTEP % (BMAX * 3 / 4) == 0 ? BMAX * 3 / 4 : (NSTEP % (BMAX / 2) == 0 ? BMAX / 2 : 4));
  static_assert(NSTEP % BT == 0, "row count per wave must be a multiple of 16");
  const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(dA + (size_t)r0 * TL, RH * TL * 4);
  const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc((PAIR ? dB : dA) + (size_t)r0 * TL, RH * TL * 4);
  const int voff = ((PAIR ? 2 * half : half) * TL + wcol) * 4;
#pragma unroll 1
  for (int b = 0; b < NSTEP; b += BT) {
    float a0[BT], a1[BT], b0[BT], b1[BT];
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      const int so = (b + i) * STEP * TL * 4;
      a0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, voff, so, 0));
      if (PAIR) {
        b0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, voff, so, 0));
        a1[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, voff, so + TL * 4, 0));
        b1[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, voff, so + TL * 4, 0));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      const int r = r0 + (b + i) * STEP + (PAIR ? 2 * half : half);
      if (JAC == 1) {
        const float p = R[swz(r, col)], q = R[swz(r + 1, col)];
        const float yr = a0[i], yi = a1[i], za = b0[i], zb = b1[i];
        const float ka = -2.f * s2 * za, kb = -omega - 2.f * s2 * zb;
        R[swz(r, col)] = fmaf(p, fmaf(ka, yr, -omega * yi), q * fmaf(ka, yi, omega * yr));
        R[swz(r + 1, col)] = fmaf(p, kb * yr, q * (kb * yi));
      } else if (JAC == 2) {
        const float p = R[swz(r, col)], q = R[swz(r + 1, col)];
        const float yr = a0[i], yi = a1[i];
        const float ku = -2.f * s2 * b0[i], kv = -2.f * s2 * b1[i];
        R[swz(r, col)] = fmaf(p, ku * yr, q * (ku * yi));
        R[swz(r + 1, col)] = fmaf(p, kv * yr, q * (kv * yi));
      } else if (PAIR) {
        const float p = R[swz(r, col)], q = R[swz(r + 1, col)];
        R[swz(r, col)] = fmaf(p, a0[i], q * b0[i]);
        R[swz(r + 1, col)] = fmaf(p, a1[i], q * b1[i]);
      } else {
        R[swz(r, col)] *= a0[i];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// First layer of the eager WIRE path: dZ_0 = J_0 dH_0 formed while the dX accumulators are stored to the image (own rows),
// J_0 from the stashed (y_0, z_0) as in rows_times<.., JAC>.  Registers (2p, 2p+1) of a lane are the (Re, Im) rows of a pair.
template <int NBM, int TL, bool ORTH = false>
__device__ __forceinline__ void acc_times_jac_to_lds(const f32x16 (&acc)[NBM], float* R, const float* __restrict__ sv_y,
                                                     const float* __restrict__ sv_z, float omega, float s0, int wcol,
                                                     int lane) {
  const int half = lane >> 5, col = lane & 31;
  float* Rl = R + (4 * half) * INR_LDS_LD + col;
  const __amdgpu_buffer_rsrc_t rsY = uniform_rsrc(sv_y, NBM * 32 * TL * 4);
  const __amdgpu_buffer_rsrc_t rsZ = uniform_rsrc(sv_z, NBM * 32 * TL * 4);
  const int voff = ((4 * half) * TL + wcol) * 4;
  const float s2 = s0 * s0;
  constexpr int MB = NBM % 2 == 0 ? 2 : 1;  // 64 loads in flight
#pragma unroll
  for (int m0 = 0; m0 < NBM; m0 += MB) {
    float y[MB][16], z[MB][16];
#pragma unroll
    for (int mm = 0; mm < MB; ++mm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int soff = (32 * (m0 + mm) + (r & 3) + 8 * (r >> 2)) * TL * 4;
        y[mm][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsY, voff, soff, 0));
        z[mm][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsZ, voff, soff, 0));
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mm = 0; mm < MB; ++mm)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int m = m0 + mm;
        const float p = acc[m][r], q = acc[m][r + 1];
        const float yr = y[mm][r], yi = y[mm][r + 1], za = z[mm][r], zb = z[mm][r + 1];
        const int row = 32 * m + (r & 3) + 8 * (r >> 2);
        if (ORTH) {  // sv_z = the orth rows (u, v): d y / d o = -2 s0^2 o y
          const float ku = -2.f * s2 * za, kv = -2.f * s2 * zb;
          Rl[row * INR_LDS_LD] = fmaf(p, ku * yr, q * (ku * yi));
          Rl[(row + 1) * INR_LDS_LD] = fmaf(p, kv * yr, q * (kv * yi));
        } else {
          const float ka = -2.f * s2 * za, kb = -omega - 2.f * s2 * zb;
          Rl[row * INR_LDS_LD] = fmaf(p, fmaf(ka, yr, -omega * yi), q * fmaf(ka, yi, omega * yr));
          Rl[(row + 1) * INR_LDS_LD] = fmaf(p, kb * yr, q * (kb * yi));
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// WIRE, eager activation: the owner of row blocks [m0, m0+MT) holds both rows of every complex feature in registers
// (2p, 2p+1), so the Gabor wavelet (one sincos + one exp per feature, networks.py:199-204) and the four Jacobian
// entries of the pair are formed ONCE -- the lazy form evaluates them in both lane halves of both waves of the pair
// (4x the transcendentals).  Writes y into the image rows (the next GEMM's plain B operand) and y, dA, dB to the
// stash rows (Rown / sv_* already point at this wave's rows): slot 0 <- y, slot 1 <- z = (a, b); slot 1 is where backward
// later leaves dZ for the batch GEMM, slot 2 of these plans stays unused.
template <int MT, int TL>
__device__ __forceinline__ void wire_epilogue(const f32x16 (&acc)[MT], const float* __restrict__ bias, float* Rown,
                                              float* __restrict__ sv_h, float* __restrict__ sv_z, bool save, float omega,
                                              float s0, int wcol, int lane) {
  const int half = lane >> 5, col = lane & 31;
  float* Rl = Rown + (4 * half) * INR_LDS_LD + col;
  const int voff = ((4 * half) * TL + wcol) * 4;  // per-lane byte offset; the row offset goes in an SGPR (stash_store)
  const __amdgpu_buffer_rsrc_t rh = uniform_rsrc(save ? sv_h : (const float*)bias, MT * 32 * TL * 4);
  const __amdgpu_buffer_rsrc_t rz = uniform_rsrc(save ? sv_z : (const float*)bias, MT * 32 * TL * 4);
  const float* bl = bias + 4 * half;
  const float s2 = s0 * s0;
  // (the bias float4s of all row blocks first: fetched where they are used, every one of the 4 MT loads got a
  // vmcnt(0) behind it -- a serialized L2 round trip that also drained the stash stores in flight)
  f32x4 bias4[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[m][g] = *reinterpret_cast<const f32x4*>(bl + 32 * m + 8 * g);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b4 = bias4[m][g];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = 32 * m + 8 * g + 2 * p;  // Re row (+ 4*half folded into Rl / so); Im row = row + 1
        const float za = acc[m][4 * g + 2 * p] + b4[2 * p], zb = acc[m][4 * g + 2 * p + 1] + b4[2 * p + 1];
        float sn, cs;
        sincos_cw(omega * za, sn, cs);
        const float E = expf(-omega * zb - s2 * (za * za + zb * zb));
        const float yr = E * cs, yi = E * sn;
        Rl[row * INR_LDS_LD] = yr;
        Rl[(row + 1) * INR_LDS_LD] = yi;
        if (save) {  // y for the next GEMM's dW and z for the backward pass, which forms the Jacobian from both
          stash_store(rh, voff, row * TL * 4, yr);
          stash_store(rh, voff, (row + 1) * TL * 4, yi);
          stash_store(rz, voff, row * TL * 4, za);
          stash_store(rz, voff, (row + 1) * TL * 4, zb);
        }
      }
    }
  }
}

// WIRE2D, eager activation (wire2d.py:49-60): the owner of a row block holds the rows (a, b) of the layer's Linear in `acc`
// and the rows (u, v) of its second Linear (scale_orth) in `acc2` -- same registers, same lane -- so
//   y = exp(j omega lin) * exp(-s0^2 (|lin|^2 + |orth|^2))
// is formed once per complex feature (the lazy form evaluated it in both lane halves of both waves of the pair, inside
// the GEMM loop, and needed a second GEMM that read h back from the stash).  Writes y into the image rows and y, z, orth to
// the stash (slots 0, 1, 5 of the layer); backward rebuilds both Jacobians from those (rows_times<.., 1 / 2>).
template <int MT, int TL>
__device__ __forceinline__ void wire2d_epilogue(const f32x16 (&acc)[MT], const f32x16 (&acc2)[MT],
                                                const float* __restrict__ bias, const float* __restrict__ bias2,
                                                float* Rown, float* __restrict__ sv_h, float* __restrict__ sv_z,
                                                float* __restrict__ sv_o, bool save, float omega, float s0, int wcol,
                                                int lane) {
  const int half = lane >> 5, col = lane & 31;
  float* Rl = Rown + (4 * half) * INR_LDS_LD + col;
  const int voff = ((4 * half) * TL + wcol) * 4;
  const __amdgpu_buffer_rsrc_t rh = uniform_rsrc(save ? sv_h : (const float*)bias, MT * 32 * TL * 4);
  const __amdgpu_buffer_rsrc_t rz = uniform_rsrc(save ? sv_z : (const float*)bias, MT * 32 * TL * 4);
  const __amdgpu_buffer_rsrc_t ro = uniform_rsrc(save ? sv_o : (const float*)bias, MT * 32 * TL * 4);
  const float* bl = bias + 4 * half;
  const float* bl2 = bias2 + 4 * half;
  const float s2 = s0 * s0;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x4 bias4[4], bias24[4];  // (both bias sets of the row block first, see wire_epilogue)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bias4[g] = *reinterpret_cast<const f32x4*>(bl + 32 * m + 8 * g);
      bias24[g] = *reinterpret_cast<const f32x4*>(bl2 + 32 * m + 8 * g);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b4 = bias4[g];
      const f32x4 c4 = bias24[g];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = 32 * m + 8 * g + 2 * p;  // Re row (+ 4*half folded into Rl / voff); Im row = row + 1
        const float za = acc[m][4 * g + 2 * p] + b4[2 * p], zb = acc[m][4 * g + 2 * p + 1] + b4[2 * p + 1];
        const float u = acc2[m][4 * g + 2 * p] + c4[2 * p], v = acc2[m][4 * g + 2 * p + 1] + c4[2 * p + 1];
        float sn, cs;
        sincos_cw(omega * za, sn, cs);
        const float Ef = expf(-omega * zb);                                    // |exp(1j*omega*lin)|
        const float G = expf(-s2 * ((za * za + zb * zb) + (u * u + v * v)));   // gauss_term (association of act_gabor2d)
        const float yr = (Ef * cs) * G, yi = (Ef * sn) * G;
        Rl[row * INR_LDS_LD] = yr;
        Rl[(row + 1) * INR_LDS_LD] = yi;
        if (save) {
          stash_store(rh, voff, row * TL * 4, yr);
          stash_store(rh, voff, (row + 1) * TL * 4, yi);
          stash_store(rz, voff, row * TL * 4, za);
          stash_store(rz, voff, (row + 1) * TL * 4, zb);
          stash_store(ro, voff, row * TL * 4, u);
          stash_store(ro, voff, (row + 1) * TL * 4, v);
        }
      }
    }
  }
}

// own rows of the image <-> global scratch [rows][TL] (WIRE2D: a layer's output gradient is needed twice)
template <int TL, int RH, bool TO_GLOBAL>
__device__ __forceinline__ void rows_copy(float* R, float* __restrict__ G, int r0, int wcol, int lane) {
  // 16 bytes per lane: eight lanes cover the 32 coordinates of a row, a wave eight rows per instruction pair
  const int seg = lane & 7, rr = lane >> 3;
  float* g = G + (wcol & ~31) + 4 * seg;
#pragma unroll 8
  for (int r = r0 + rr; r < r0 + RH; r += 8) {
    if (TO_GLOBAL)
      *reinterpret_cast<f32x4*>(g + r * TL) = *reinterpret_cast<const f32x4*>(R + swz(r, 4 * seg));
    else
      *reinterpret_cast<f32x4*>(R + swz(r, 4 * seg)) = *reinterpret_cast<const f32x4*>(g + r * TL);
  }
}

// dW / db of a first layer on raw coordinates (K = 3: WIRE, WIRE2D) on the VECTOR ALUs.  As MFMA passes it is M/32
// row blocks x one column block of which 3 columns are real, on two of the four waves: 35 k cycles of a 910 k WIRE tile
// for 0.1 % of its FLOPs.  Here a thread owns a row of dZ_0 (read from the groups' images, a float4 per 4
// coordinates), the tile's 64 x 3 coordinates sit in LDS (`xs`, broadcast reads) and the three sums + the row sum stay
// in registers: 4 FMAs per coordinate.  Coordinates past the batch were loaded clamped; their dZ columns are zero.
template <int TL, int NG>
__device__ __forceinline__ void dw_first3_valu(const float* lds_img, int region_stride, const float* xs, int M,
                                               float* slab_w_generic, float* slab_b_generic, bool first, int tid) {
  typedef __attribute__((address_space(1))) float gfloat;
  typedef const __attribute__((address_space(3))) f32x4 lf4;
  gfloat* slab_w = (gfloat*)slab_w_generic;
  gfloat* slab_b = (gfloat*)slab_b_generic;
  for (int r = tid; r < M; r += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, bs = 0.f;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, ob = 0.f;
    if (!first) {
      o0 = slab_w[r * 3 + 0];
      o1 = slab_w[r * 3 + 1];
      o2 = slab_w[r * 3 + 2];
      ob = slab_b[r];
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const f32x4 d = *(lf4*)(lds_img + gi * region_stride + r * INR_LDS_LD + 4 * c4);
        // coordinates 32 gi + 4 c4 + (0..3): twelve consecutive floats of xs
        const f32x4 x0 = *(lf4*)(xs + (gi * 32 + 4 * c4) * 3 + 0);
        const f32x4 x1 = *(lf4*)(xs + (gi * 32 + 4 * c4) * 3 + 4);
        const f32x4 x2 = *(lf4*)(xs + (gi * 32 + 4 * c4) * 3 + 8);
        a0 = fmaf(d[0], x0[0], a0), a1 = fmaf(d[0], x0[1], a1), a2 = fmaf(d[0], x0[2], a2);
        a0 = fmaf(d[1], x0[3], a0), a1 = fmaf(d[1], x1[0], a1), a2 = fmaf(d[1], x1[1], a2);
        a0 = fmaf(d[2], x1[2], a0), a1 = fmaf(d[2], x1[3], a1), a2 = fmaf(d[2], x2[0], a2);
        a0 = fmaf(d[3], x2[1], a0), a1 = fmaf(d[3], x2[2], a1), a2 = fmaf(d[3], x2[3], a2);
        bs += (d[0] + d[1]) + (d[2] + d[3]);
      }
    }
    slab_w[r * 3 + 0] = o0 + a0;
    slab_w[r * 3 + 1] = o1 + a1;
    slab_w[r * 3 + 2] = o2 + a2;
    slab_b[r] = ob + bs;
  }
}

template <int NB, int INMODE, int HACT, int MODE>
__global__ __launch_bounds__(256) void inr_mlp_wide_kernel(const NetDesc nd, const LossDesc ld, const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool G2D = HACT == ACT_GABOR2D;  // WIRE2D: second Linear (scale_orth) per layer, L[orth0 + l]
  constexpr bool PAIR = HACT == ACT_GABOR || G2D;
  constexpr bool EAGER = PAIR;  // WIRE / WIRE2D: the image holds activations y, formed once by the row owners
  constexpr int MT = NB / 2, NG = 2, NW = 4, NS = G2D ? 7 : (PAIR ? 3 : 2);
  constexpr int RH = MT * 32;  // image rows per wave of a pair
  constexpr int TL = NG * 32;
  constexpr int RS = NB * 32 * INR_LDS_LD;
  constexpr int HSZ = NB * 32 * TL;
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int g = w >> 1, hh = w & 1, m0 = MT * hh;
  const int half = lane >> 5, col = lane & 31;
  const int wcol = g * 32 + col;
  float* R = lds + g * RS;
  float* Rown = R + m0 * 32 * INR_LDS_LD;
  float* encB_lds = lds + NG * RS;
  if (INMODE == IN_GAUSS) {
    for (int i = tid; i < 3 * nd.E; i += NW * 64) encB_lds[i] = a.encB[i];
    __syncthreads();
  }
  const int D = nd.D;
  // EAGER kernels, last layer of <= 4 rows: its fragments (rows 0..3; float4 (s4*2 + half)*4 + row) and the exchange
  // buffer of the wave pairs' partial sums sit behind the encoder matrix (see the last layer below)
  float* llw = encB_lds + (INMODE == IN_GAUSS ? ((3 * nd.E + 3) & ~3) : 0);
  float* xw = llw + NB * 4 * 8 * 4;  // [NG][4][32]
  const bool ll_valu = EAGER && MODE != MODE_BWD && nd.L[D - 1].M <= 4;
  if (ll_valu) {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.packed + nd.L[D - 1].pf_off);
    for (int c = tid; c < NB * 4 * 8; c += NW * 64)
      reinterpret_cast<f32x4*>(llw)[c] = src[(c >> 3) * 64 + ((c >> 2) & 1) * 32 + (c & 3)];
    __syncthreads();
  }
  float* slab = (MODE != MODE_FWD) ? a.slabs + (size_t)blockIdx.x * nd.slab_floats : nullptr;
  float loss_acc = 0.f;
  bool first = a.accumulate == 0;  // accumulate: a follow-up launch of the same step (inr_api.hip, split launches)
  const LayerDesc& LL = nd.L[D - 1];
  const size_t aoff = (size_t)m0 * 256;

  for (int tile = a.tile0 + blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long row0 = (long long)tile * TL;
    const long long crow = row0 + wcol;
    const bool valid = crow < a.B;
    const bool saving = G2D || (MODE != MODE_FWD) || (a.save != nullptr);  // WIRE2D: the API insists on a buffer
    float* sv = a.save;
    if (saving) sv += (size_t)(a.save_by_block ? blockIdx.x : tile) * nd.save_floats_per_tile;
    float* sv_last = sv + (size_t)NS * (D - 1) * HSZ;
    float* sv_enc = sv_last + 4 * TL;
    float* sv_g = sv_last + 4 * TL;  // WIRE2D (never gauss): copy of a layer's output gradient [NB*32][TL]
    const bool stash = saving && hh == 0;  // one wave of the pair writes the (shared) lazy-activation stash
    int si = 0;  // diagnostic builds: phase stamps 0, 1, 2, ... in program order (tools/stamps.py wire)
    (void)si;
    // what the loss section needs from memory, requested now (fetched where it is used, the last-layer biases, the
    // sampling mask and the target row were three serialized round trips between the last layer and the loss)
    float gt_pre[4] = {0.f, 0.f, 0.f, 0.f}, lb_pre[4] = {0.f, 0.f, 0.f, 0.f};
    bool sampled_pre = false;
    constexpr bool PRE = !G2D;  // (the WIRE2D build has no registers to carry nine values across the tile: 147 -> 160 scratch
                                // instructions, 14 of them inside short loops instead of 4)
    if (PRE && MODE != MODE_BWD && hh == 0) {
      const int nrows_b = nd.last_act == ACT_CTANH ? 2 * nd.out_f : nd.out_f;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < nrows_b) lb_pre[o] = a.packed[LL.pbias_off + o];
      if (MODE == MODE_FUSED && half == 0 && valid) {
        sampled_pre = a.mask == nullptr || a.mask[crow] != 0;
#pragma unroll
        for (int o = 0; o < 4; ++o)
          if (o < nd.out_f) gt_pre[o] = a.gt[crow * nd.out_f + o];
      }
    }
    INR_STAMP(si); ++si;

    // ================================ forward =================================
    if (MODE != MODE_BWD) {
      {
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = zero16();
        const LayerDesc& L0 = nd.L[0];
        if (INMODE == IN_GAUSS) {
          float x0 = 0.f, x1 = 0.f, x2 = 0.f;
          if (valid) {
            x0 = a.x[3 * crow + 0];
            x1 = a.x[3 * crow + 1];
            x2 = a.x[3 * crow + 2];
          }
          const float two_pi = 6.283185307179586f;
          if (stash)
            fwd_layer0_gauss<MT, TL, true, NB>(acc, a.packed + L0.pf_off + aoff, encB_lds, nd.E, two_pi * x0, two_pi * x1,
                                               two_pi * x2, sv_enc, wcol, lane);
          else
            fwd_layer0_gauss<MT, TL, false, NB>(acc, a.packed + L0.pf_off + aoff, encB_lds, nd.E, two_pi * x0,
                                                two_pi * x1, two_pi * x2, nullptr, wcol, lane);
        } else {
          fwd_layer0_x<MT, NB>(acc, a.packed + L0.pf_off + aoff, a.x + (size_t)(valid ? crow : 0) * L0.K, valid, L0.K,
                               L0.Kpad8, lane);
        }
        INR_STAMP(si); ++si;
        if (G2D) {  // orth_0 = V_0 x + c_0 in a second set of accumulators, then the wavelet of both
          const LayerDesc& O0 = nd.L[nd.orth0];
          f32x16 acc2[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) acc2[m] = zero16();
          fwd_layer0_x<MT, NB>(acc2, a.packed + O0.pf_off + aoff, a.x + (size_t)(valid ? crow : 0) * O0.K, valid, O0.K,
                               O0.Kpad8, lane);
          wire2d_epilogue<MT, TL>(acc, acc2, a.packed + L0.pbias_off + m0 * 32, a.packed + O0.pbias_off + m0 * 32, Rown,
                                  sv + (size_t)m0 * 32 * TL, sv + (size_t)1 * HSZ + (size_t)m0 * 32 * TL,
                                  sv + (size_t)5 * HSZ + (size_t)m0 * 32 * TL, saving, L0.omega, L0.s0, wcol, lane);
        } else if (EAGER) {
          wire_epilogue<MT, TL>(acc, a.packed + L0.pbias_off + m0 * 32, Rown, sv + (size_t)m0 * 32 * TL,
                                sv + (size_t)1 * HSZ + (size_t)m0 * 32 * TL, saving, L0.omega, L0.s0, wcol, lane);
        } else {
          acc_to_lds<MT, true>(acc, Rown, a.packed + L0.pbias_off + m0 * 32, lane);
        }
      }
      INR_STAMP(si); ++si;
      __syncthreads();  // z_0 (and orth_0) complete
      for (int l = 1; l < D - 1; ++l) {
        const LayerDesc& Ll = nd.L[l];
        const ActParams ap{nd.L[l - 1].omega, nd.L[l - 1].s0};
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = zero16();
        float* sh = sv + (size_t)(NS * (l - 1)) * HSZ;
        f32x16 acc2[G2D ? MT : 1];
        if (EAGER)  // plain GEMM on the activations in the image
          bwd_dx<MT, TL, false, false, NB>(acc, R, a.packed + Ll.pf_off + aoff, NB * 32, nullptr, wcol, lane);
        else if (saving && !G2D && hh == 0)  // both waves of the pair form every activation: each stashes half the groups
          fwd_layer<NB, MT, TL, HACT, 3, NB>(acc, R, a.packed + Ll.pf_off + aoff, ap, sh, wcol, lane);
        else if (saving && !G2D)
          fwd_layer<NB, MT, TL, HACT, 4, NB>(acc, R, a.packed + Ll.pf_off + aoff, ap, sh, wcol, lane);
        else if (stash)
          fwd_layer<NB, MT, TL, HACT, true, NB>(acc, R, a.packed + Ll.pf_off + aoff, ap, sh, wcol, lane);
        else
          fwd_layer<NB, MT, TL, HACT, false, NB>(acc, R, a.packed + Ll.pf_off + aoff, ap, nullptr, wcol, lane);
        if constexpr (G2D) {  // orth_l = V_l y_{l-1} + c_l: a second plain GEMM on the same image, second accumulators
#pragma unroll
          for (int m = 0; m < MT; ++m) acc2[m] = zero16();
          bwd_dx<MT, TL, false, false, NB>(acc2, R, a.packed + nd.L[nd.orth0 + l].pf_off + aoff, NB * 32, nullptr, wcol, lane);
        }
        INR_STAMP(si); ++si;
        __syncthreads();  // both waves of the pair have read the image (z_{l-1}, or y_{l-1} in the eager kernels)
        if constexpr (G2D) {
          const LayerDesc& Ol = nd.L[nd.orth0 + l];
          float* so = sv + (size_t)(NS * l) * HSZ + (size_t)m0 * 32 * TL;
          wire2d_epilogue<MT, TL>(acc, acc2, a.packed + Ll.pbias_off + m0 * 32, a.packed + Ol.pbias_off + m0 * 32, Rown, so,
                                  so + HSZ, so + 5 * (size_t)HSZ, saving, Ll.omega, Ll.s0, wcol, lane);
        } else if (EAGER) {
          float* so = sv + (size_t)(NS * l) * HSZ + (size_t)m0 * 32 * TL;
          wire_epilogue<MT, TL>(acc, a.packed + Ll.pbias_off + m0 * 32, Rown, so, so + HSZ, saving, Ll.omega, Ll.s0, wcol,
                                lane);
        } else {
          acc_to_lds<MT, true>(acc, Rown, a.packed + Ll.pbias_off + m0 * 32, lane);
        }
        INR_STAMP(si); ++si;
        __syncthreads();  // the layer's image rows are complete
      }
      float g4[4] = {0.f, 0.f, 0.f, 0.f}, dy[4] = {0.f, 0.f, 0.f, 0.f};
      int nrows_last = nd.out_f;
      // The last layer (<= 4 rows) over the activated image on the VECTOR ALUs, both waves of a pair: as one 32-row MFMA
      // block it kept ONE wave of each pair busy for 32 k cycles (192 k-steps, 4 useful rows of 32) while its partner
      // waited.  A lane sums W[o][k] y[k][col] over the rows k = 8 s4 + 2e + half of its wave's half of the groups
      // (weights: a broadcast float4 per row and group from LDS); lane halves, then the two waves (through `xw`) are
      // added -- fixed order.
      float o4[4] = {0.f, 0.f, 0.f, 0.f};
      if (ll_valu) {
        constexpr int per = NB * 4 / 2;
        typedef const __attribute__((address_space(3))) f32x4 lf4;
        typedef const __attribute__((address_space(3))) float lfl;
        lfl* Ry = (lfl*)(R + col + half * INR_LDS_LD);
        lf4* wq = (lf4*)llw + half * 4;
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int s4 = hh * per; s4 < (hh + 1) * per; ++s4) {
          float yv[4];
          f32x4 wv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) yv[e] = Ry[(8 * s4 + 2 * e) * INR_LDS_LD];
#pragma unroll
          for (int o = 0; o < 4; ++o) wv[o] = wq[s4 * 8 + o];
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int o = 0; o < 4; ++o) ps[o] = fmaf(wv[o][e], yv[e], ps[o]);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) ps[o] += __shfl_xor(ps[o], 32);
        if (hh == 1 && half == 0) {
#pragma unroll
          for (int o = 0; o < 4; ++o) xw[(g * 4 + o) * 32 + col] = ps[o];
        }
        __syncthreads();
        if (hh == 0) {
#pragma unroll
          for (int o = 0; o < 4; ++o) o4[o] = ps[o] + xw[(g * 4 + o) * 32 + col];
        }
      }
      if (hh == 0) {  // last layer: one row block
        f32x16 accL[1];
        accL[0] = zero16();
        const ActParams ap{nd.L[D - 2].omega, nd.L[D - 2].s0};
        float* sh = sv + (size_t)(NS * (D - 2)) * HSZ;
        if (ll_valu) {
#pragma unroll
          for (int o = 0; o < 4; ++o) accL[0][o] = o4[o];
        } else if (EAGER)
          bwd_dx<1, TL, false, false>(accL, R, a.packed + LL.pf_off, NB * 32, nullptr, wcol, lane);
        else if (saving)
          fwd_layer<NB, 1, TL, HACT, true>(accL, R, a.packed + LL.pf_off, ap, sh, wcol, lane);
        else
          fwd_layer<NB, 1, TL, HACT, false>(accL, R, a.packed + LL.pf_off, ap, nullptr, wcol, lane);
        float zl[4], y[4];
        const bool ctanh = nd.last_act == ACT_CTANH;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          zl[o] = accL[0][o];
          if (o < (ctanh ? 2 * nd.out_f : nd.out_f)) zl[o] += PRE ? lb_pre[o] : a.packed[LL.pbias_off + o];
        }
        nrows_last = last_layer_act(nd.last_act, nd.out_f, nd.w0, zl, y, dy);
#pragma unroll
        for (int o = 0; o < 4; ++o)
          if (half == 0 && valid && o < nd.out_f && a.out != nullptr) a.out[crow * nd.out_f + o] = y[o];
        if (MODE == MODE_FWD) {
          if (saving && half == 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o) sv_last[o * TL + wcol] = dy[o];
          }
        } else if (PRE ? sampled_pre : (half == 0 && valid && (a.mask == nullptr || a.mask[crow] != 0))) {
          if (!PRE) {
            for (int o = 0; o < nd.out_f; ++o) gt_pre[o] = a.gt[crow * nd.out_f + o];
          }
          loss_acc += loss_row(ld, nd.out_f, y, gt_pre, g4);
        }
      }
      INR_STAMP(si); ++si;
      __syncthreads();  // the last layer has read z_{D-2}: rows may be overwritten (dZ_last / the next tile's z_0)
      if (MODE == MODE_FUSED) {
        if (hh == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = 0.f;
            if (r < 4 && half == 0 && r < nrows_last)
              v = g4[nd.last_act == ACT_CTANH ? (r & 3) >> 1 : (r & 3)] * dy[r & 3];
            R[swz(acc_row(r, half), col)] = v;
          }
        }
      }
    }

    // ================================ backward ================================
    if (MODE != MODE_FWD) {
      if (MODE == MODE_BWD && hh == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.f;
          const bool ct = nd.last_act == ACT_CTANH;
          if (r < 4 && half == 0 && r < (ct ? 2 * nd.out_f : nd.out_f) && valid)
            v = a.dout[crow * nd.out_f + (ct ? (r & 3) >> 1 : (r & 3))] * sv_last[(r & 3) * TL + wcol];
          R[swz(acc_row(r, half), col)] = v;
        }
      }
      INR_STAMP(si); ++si;
      __syncthreads();  // every group's dZ_last is in LDS
      {
        if (LL.M <= 4) {  // <= 4 rows: on the vector ALUs (inr_mlp_impl.h), not as a padded 32-row MFMA block
          dw_rows4_valu<TL, NW>(lds, RS, sv + (size_t)(NS * (D - 2)) * HSZ, NB * 32, LL.M, LL.K, slab + LL.gw_off,
                                slab + LL.gb_off, first, w, lane);
        } else {
          BSrcStash<TL> bs{sv + (size_t)(NS * (D - 2)) * HSZ};
          for (int n = w; n < LL.Kblk; n += NW)
            dw_pass<1, TL, false, BSrcStash<TL>>(lds, RS, bs, n, slab + LL.gw_off, slab + LL.gb_off, LL.M, LL.K, first,
                                                 n == 0, lane);
        }
      }
      INR_STAMP(si); ++si;
      f32x16 gacc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) gacc[m] = zero16();
      bwd_dx<MT, TL, false, false, NB>(gacc, R, a.packed + LL.pb_off + aoff, LL.Mpad8, nullptr, wcol, lane);
      INR_STAMP(si); ++si;
      __syncthreads();  // all reads of dZ_last are done
      if (D == 2 && (EAGER || G2D))
        acc_times_jac_to_lds<MT, TL>(gacc, Rown, sv + (size_t)m0 * 32 * TL, sv + (size_t)1 * HSZ + (size_t)m0 * 32 * TL,
                                     nd.L[0].omega, nd.L[0].s0, wcol, lane);  // dZ_0 (own rows)
      else if (D == 2)
        acc_times_d_to_lds<MT, TL, PAIR>(gacc, Rown, sv + (size_t)1 * HSZ + (size_t)m0 * 32 * TL,
                                         sv + (size_t)2 * HSZ + (size_t)m0 * 32 * TL, wcol, lane);  // dZ_0 (own rows)
      else
        acc_to_lds<MT, false>(gacc, Rown, nullptr, lane);  // dH_{D-2} (own rows)
      for (int l = D - 2; l >= 1; --l) {
        const LayerDesc& Ll = nd.L[l];
        INR_STAMP(si); ++si;
        __syncthreads();  // dH_l complete
        if (G2D) rows_copy<TL, RH, true>(R, sv_g, RH * hh, wcol, lane);  // needed again for the orth Linear
        if (EAGER || G2D)  // own rows: dZ_l = J_l dH_l with J_l from the stashed (y_l, z_l)
          rows_times<TL, RH, true, 1>(R, sv + (size_t)(NS * l) * HSZ, sv + (size_t)(NS * l + 1) * HSZ, RH * hh, wcol, lane,
                                         Ll.omega, Ll.s0);
        else  // own rows: dZ_l = dH_l * act'
          rows_times<TL, RH, PAIR>(R, sv + (size_t)(NS * l + 1) * HSZ, sv + (size_t)(NS * l + 2) * HSZ, RH * hh, wcol, lane);
        INR_STAMP(si); ++si;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m) gacc[m] = zero16();
        bwd_dx<MT, TL, false, false, NB>(gacc, R, a.packed + Ll.pb_off + aoff, Ll.Mpad8, nullptr, wcol, lane);
        INR_STAMP(si); ++si;
        // dW_l is left to the batch GEMM (inr_dw_gemm.hip): own rows of dZ_l over the act' slot they were formed from
        rows_copy<TL, RH, true>(R, sv + (size_t)(NS * l + 1) * HSZ, RH * hh, wcol, lane);
        INR_STAMP(si); ++si;
        __syncthreads();  // dZ_l has been read by every wave's dX
        if (G2D) {  // second Linear of the layer: dZ_orth = J_orth dH_l, dH_{l-1} += V_l^T dZ_orth, dV_l
          const LayerDesc& Ol = nd.L[nd.orth0 + l];
          rows_copy<TL, RH, false>(R, sv_g, RH * hh, wcol, lane);
          rows_times<TL, RH, true, 2>(R, sv + (size_t)(NS * l) * HSZ, sv + (size_t)(NS * l + 5) * HSZ, RH * hh, wcol, lane,
                                      Ll.omega, Ll.s0);  // dZ_orth = J_orth dH_l from the stashed (y_l, orth_l)
          __syncthreads();
          bwd_dx<MT, TL, false, false, NB>(gacc, R, a.packed + Ol.pb_off + aoff, Ol.Mpad8, nullptr, wcol, lane);
          rows_copy<TL, RH, true>(R, sv + (size_t)(NS * l + 3) * HSZ, RH * hh, wcol, lane);  // dZ_orth (own rows)
          __syncthreads();
        }
        if (l == 1 && (EAGER || G2D))
          acc_times_jac_to_lds<MT, TL>(gacc, Rown, sv + (size_t)m0 * 32 * TL, sv + (size_t)1 * HSZ + (size_t)m0 * 32 * TL,
                                       nd.L[0].omega, nd.L[0].s0, wcol, lane);
        else if (l == 1)
          acc_times_d_to_lds<MT, TL, PAIR>(gacc, Rown, sv + (size_t)1 * HSZ + (size_t)m0 * 32 * TL,
                                           sv + (size_t)2 * HSZ + (size_t)m0 * 32 * TL, wcol, lane);
        else
          acc_to_lds<MT, false>(gacc, Rown, nullptr, lane);
      }
      {
        const LayerDesc& L0 = nd.L[0];
        INR_STAMP(si); ++si;
        __syncthreads();  // dZ_0 complete
        if (INMODE == IN_GAUSS) {
          rows_copy<TL, RH, true>(R, sv + (size_t)1 * HSZ, RH * hh, wcol, lane);  // dZ_0 (own rows), for the GEMM
        } else if (EAGER && !G2D && L0.K == 3) {
          // (xs = the pair exchange buffer of the last layer, free here: 64 coordinates x 3, clamped to the batch.  Not in the
          // WIRE2D build: with dH_0 still live in 128 accumulator registers the unrolled pass spills -- 134 -> 528 scratch
          // instructions -- and the step got 3 % slower; its dW_0 / dV_0 stay MFMA passes, compiled as real functions)
          if (tid < TL * 3) {
            const long long xr = row0 + tid / 3 < a.B ? row0 + tid / 3 : a.B - 1;
            xw[tid] = a.x[xr * 3 + tid % 3];
          }
          __syncthreads();
          dw_first3_valu<TL, NG>(lds, RS, xw, L0.M, slab + L0.gw_off, slab + L0.gb_off, first, tid);
        } else {
          BSrcX bs{a.x, row0, a.B, L0.K};
          for (int it = w; it < 2 * L0.Kblk; it += NW) {
            const int n = it >> 1, c = it & 1;
            dw_pass<MT, TL, true, BSrcX>(lds + c * MT * 32 * INR_LDS_LD, RS, bs, n,
                                         slab + L0.gw_off + (size_t)c * RH * L0.K, slab + L0.gb_off + c * RH, L0.M,
                                         L0.K, first, n == 0, lane);
          }
          if (G2D) {  // dV_0 from dZ_0,orth = J_orth,0 dH_0 (the accumulators still hold dH_0)
            const LayerDesc& O0 = nd.L[nd.orth0];
            __syncthreads();
            acc_times_jac_to_lds<MT, TL, true>(gacc, Rown, sv + (size_t)m0 * 32 * TL, sv + (size_t)5 * HSZ + (size_t)m0 * 32 * TL,
                                               nd.L[0].omega, nd.L[0].s0, wcol, lane);  // J_orth,0 from (y_0, orth_0)
            __syncthreads();
            for (int it = w; it < 2 * O0.Kblk; it += NW) {
              const int n = it >> 1, c = it & 1;
              dw_pass<MT, TL, true, BSrcX>(lds + c * MT * 32 * INR_LDS_LD, RS, bs, n,
                                           slab + O0.gw_off + (size_t)c * RH * O0.K, slab + O0.gb_off + c * RH, O0.M,
                                           O0.K, first, n == 0, lane);
            }
          }
        }
        __syncthreads();
      }
      INR_STAMP(si); ++si;
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

template <int NB, int INMODE, int HACT, int MODE>
inline hipError_t launch_mlp_wide(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  constexpr bool eager = HACT == ACT_GABOR || HACT == ACT_GABOR2D;  // + last-layer fragments and the pair exchange buffer
  const size_t lds_bytes = ((size_t)2 * NB * 32 * INR_LDS_LD + (INMODE == IN_GAUSS ? ((3 * (size_t)nd.E + 3) & ~(size_t)3) : 0) +
                            (eager ? (size_t)NB * 4 * 8 * 4 + 2 * 4 * 32 : 0)) * sizeof(float);
  auto k = inr_mlp_wide_kernel<NB, INMODE, HACT, MODE>;
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  // backward has no dW passes for the hidden-width layers: the caller runs the batch GEMM on the stash
  if (MODE != MODE_FWD && !a.dw_gemm && (nd.D > 2 || INMODE == IN_GAUSS)) return hipErrorInvalidValue;
  {
    hipError_t e = allow_full_lds<inr_mlp_wide_kernel<NB, INMODE, HACT, MODE>>();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, st, nd, ld, a);
  return hipGetLastError();
}

}  // namespace inr
