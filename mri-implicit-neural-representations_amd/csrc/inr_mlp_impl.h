// inr_mlp_impl.h -- the fused coordinate-MLP kernel (fp32-exact path) for gfx950.
//
// One workgroup = NW waves = one tile of TL = 32*NW coordinates; each wave owns 32 coordinates and
// carries them through EVERY layer on its own (no inter-wave traffic in the forward pass):
//
//   * activations live TRANSPOSED, X^T [features x coords], coordinates on the MFMA lanes.
//     Y^T = W . X^T makes the PyTorch-layout weight matrix the A operand and the activations the
//     B operand of v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains -> bit-level agreement with a
//     k-ordered CPU fmaf chain, 157 TFLOP/s peak);
//   * between layers the pre-activations (bias added) sit in a per-wave padded LDS image
//     [feature][33]; the CONSUMER applies the activation lazily, one 4-k-step group ahead of the
//     MFMAs that use it, and stashes h and act' for the backward pass;
//   * weights are streamed as pre-packed A fragments (one float4 = 4 k-steps per lane), software
//     prefetched one group ahead into ping-pong register sets;
//   * backward: dX^T = W^T . dZ^T reuses the same structure (dZ = dH * act' formed lazily and
//     written back into the LDS image), then dW = dZ^T . H contracts over the tile's coordinates
//     with dZ read "feature on lane" from the same LDS image (the padding makes both access
//     directions conflict-free) and H read from the stash;
//   * each persistent workgroup accumulates its dW in a private slab (plain stores, fixed order),
//     a second kernel sums the slabs in block order: deterministic, no float atomics.
//
// Families (HACT): SIN = SIREN (models/networks.py:74-124), RELU = FFN (:48-69), GABOR = WIRE
// (:160-260; complex layers run as interleaved real rows 2i = Re, 2i+1 = Im of twice the width).
// Input stage (INMODE): fused gauss Positional_Encoder (:30-33) or a matrix x [B,K0].
// Adjoint per SURVEY.md Appendix A.2 / A.3.
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

// Complex Gabor wavelet of WIRE (networks.py:199-204): lin = a + jb, y = exp(j*omega*lin - |s0*lin|^2)
//   = E (cos(omega a) + j sin(omega a)),  E = exp(-omega b - s0^2 (a^2 + b^2)).
// Pre-activation rows are interleaved (row 2i = a_i, row 2i+1 = b_i); a lane of half h owns row
// 2i+h, emits y_r (h = 0) or y_i (h = 1) and the two Jacobian entries of ITS row:
//   dZ[row] = p * dA + q * dB   with (p, q) = dL/d(y_r, y_i) of the pair.
__device__ __forceinline__ void act_gabor(float za, float zb, float omega, float s0, int half, float& h,
                                          float& dA, float& dB) {
  float sn, cs;
  sincos_cw(omega * za, sn, cs);
  const float s2 = s0 * s0;
  const float E = expf(-omega * zb - s2 * (za * za + zb * zb));
  const float yr = E * cs, yi = E * sn;
  h = half ? yi : yr;
  const float ka = -2.f * s2 * za, kb = -omega - 2.f * s2 * zb;
  dA = half ? kb * yr : fmaf(ka, yr, -omega * yi);  // d(y_r)/d(row)
  dB = half ? kb * yi : fmaf(ka, yi, omega * yr);   // d(y_i)/d(row)
}

// WIRE2D (wire2d.py:49-60): y = exp(j omega lin) * exp(-s0^2 (|lin|^2 + |orth|^2)), orth = u + jv the second
// Linear of the layer.  o = this lane's own orth row (u for half 0, v for half 1), q = u^2 + v^2 of the pair.
// Besides the Jacobian entries of the lane's lin row (as act_gabor) it emits those of its orth row:
//   dZ_orth[row] = p * dA2 + q * dB2,   d y / d o = -2 s0^2 o y.
__device__ __forceinline__ void act_gabor2d(float za, float zb, float o, float q, float omega, float s0, int half,
                                            float& h, float& dA, float& dB, float& dA2, float& dB2) {
  float sn, cs;
  sincos_cw(omega * za, sn, cs);
  const float s2 = s0 * s0;
  const float Ef = expf(-omega * zb);                          // |exp(1j*omega*lin)|
  const float G = expf(-s2 * ((za * za + zb * zb) + q));       // gauss_term
  const float yr = (Ef * cs) * G, yi = (Ef * sn) * G;
  h = half ? yi : yr;
  const float ka = -2.f * s2 * za, kb = -omega - 2.f * s2 * zb;
  dA = half ? kb * yr : fmaf(ka, yr, -omega * yi);
  dB = half ? kb * yi : fmaf(ka, yi, omega * yr);
  const float ko = -2.f * s2 * o;
  dA2 = ko * yr;
  dB2 = ko * yi;
}

__device__ __forceinline__ void act_fwd_rt(int act, float z, float w0, float& h, float& d);

// WIRE2D last_tanh: torch.nn.Tanh() on the complex output, then .real.  tanh(a + jb) = (sinh 2a + j sin 2b) /
// (cosh 2a + cos 2b) = t_r + j t_i;  y = t_r,  dy/da = Re(1 - tanh^2) = 1 - t_r^2 + t_i^2,  dy/db = 2 t_r t_i.
__device__ __forceinline__ void act_ctanh(float a, float b, float& y, float& dya, float& dyb) {
  const float e = expf(2.f * a), ei = 1.f / e;
  float s2, c2;
  sincos_cw(2.f * b, s2, c2);
  const float inv = 1.f / (0.5f * (e + ei) + c2);
  const float tr = 0.5f * (e - ei) * inv, ti = s2 * inv;
  y = tr;
  dya = 1.f - tr * tr + ti * ti;
  dyb = 2.f * tr * ti;
}

// last-layer epilogue shared by the fused kernels: z[r] = the lane's first four accumulator rows (+ bias).  Real
// outputs: y[o] = act(z[o]), one image row per output.  ACT_CTANH: rows (2o, 2o+1) = (Re, Im) of output o.
// drow[r] = d y / d z[r];  returns the number of image rows that carry a gradient.
__device__ __forceinline__ int last_layer_act(int last_act, int out_f, float w0, const float (&z)[4], float (&y)[4],
                                              float (&drow)[4]) {
  if (last_act == ACT_CTANH) {
#pragma unroll
    for (int o = 0; o < 2; ++o) act_ctanh(z[2 * o], z[2 * o + 1], y[o], drow[2 * o], drow[2 * o + 1]);
    y[2] = y[3] = 0.f;
    return 2 * out_f;
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) act_fwd_rt(last_act, z[o], w0, y[o], drow[o]);
  return out_f;
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

// A-fragment stream through a buffer descriptor: wave-uniform base, the lane's byte offset, the group offset in an SGPR --
// `p + n` (n in float4s, as with the plain pointer it replaces) only moves the scalar offset, so the loops carry no
// per-lane 64-bit address and spend no vector ALU on it.
struct AFragPtr {
  __amdgpu_buffer_rsrc_t rs;
  int voff, soff;
  __device__ __forceinline__ AFragPtr operator+(size_t n) const { return AFragPtr{rs, voff, soff + (int)n * 16}; }
};

// Buffer descriptor from a wave-uniform pointer (readfirstlane makes the uniformity provable, so
// hipcc emits plain buffer_load ... offen with an SGPR descriptor instead of waterfall loops or --
// worse -- 128 hoisted 64-bit VGPR addresses that it then spills around the tile loop).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, int bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, bytes, 0x00020000);
}

// One dword per lane into the stash through an SGPR buffer descriptor: the address is  descriptor base + per-lane
// byte offset `voff` (formed once) + wave-uniform byte offset `soff` (SALU), so a store inside a GEMM loop costs no
// vector-ALU address arithmetic -- on the fp32 MFMA path every VALU instruction is time added to the matrix pipe's.
__device__ __forceinline__ void stash_store(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}

// acc (+bias) -> this wave's LDS image rows [0, NBM*32).  Rows of register group g = r>>2 are
// 32m + 8g + 4*half + (0..3): one float4 of bias per group.  `bias` is the zero-padded bias image
// (NBM*32 entries) kept inside the packed buffer.
template <int NBM, bool BIAS>
__device__ __forceinline__ void acc_to_lds(const f32x16 (&acc)[NBM], float* R, const float* __restrict__ bias,
                                           int lane) {
  const int half = lane >> 5, col = lane & 31;
  float* Rl = R + (4 * half) * INR_LDS_LD + col;
  const float* bl = bias + 4 * half;
  // The bias float4s of MB row blocks are requested TOGETHER, then used: written as "load, add, store" per group the
  // compiler kept that order and put s_waitcnt vmcnt(0) behind every one of the 4 NBM loads -- 32 serialized L2 round
  // trips per layer (each also draining the stash stores in flight), a third of the forward layers' overhead.
  constexpr int MB = NBM % 4 == 0 ? 4 : (NBM % 2 == 0 ? 2 : 1);
#pragma unroll
  for (int m0 = 0; m0 < NBM; m0 += MB) {
    f32x4 b4[MB][4];
#pragma unroll
    for (int mm = 0; mm < MB; ++mm)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        b4[mm][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (BIAS) b4[mm][g] = *reinterpret_cast<const f32x4*>(bl + 32 * (m0 + mm) + 8 * g);
      }
    if (BIAS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mm = 0; mm < MB; ++mm)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          Rl[(32 * (m0 + mm) + 8 * g + j) * INR_LDS_LD] = acc[m0 + mm][4 * g + j] + b4[mm][g][j];
  }
}

__device__ __forceinline__ AFragPtr afrag_ptr(const float* wp, int lane) {  // wp: wave-uniform start of a packed image
  return AFragPtr{uniform_rsrc(wp, 0x7ffffff0), lane * 16, 0};
}
template <int NBM>
__device__ __forceinline__ void load_afrag(f32x4 (&a)[NBM], const AFragPtr& p) {
#pragma unroll
  for (int m = 0; m < NBM; ++m)
    a[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(p.rs, p.voff, p.soff + m * 1024, 0));
}

// First layer: dZ_0 = dH_0 * act'(z_0) is formed while the dX accumulators are stored to the image
// (there is no dX for the input): 16*NBM independent coalesced stash loads, all in flight together.
// PAIR (WIRE): rows (2i, 2i+1) = registers (2p, 2p+1) of one lane:  dZ[row] = p*dA[row] + q*dB[row].
template <int NBM, int TL, bool PAIR>
__device__ __forceinline__ void acc_times_d_to_lds(const f32x16 (&acc)[NBM], float* R,
                                                   const float* __restrict__ sv_dA,
                                                   const float* __restrict__ sv_dB, int wcol, int lane) {
  const int half = lane >> 5, col = lane & 31;
  float* Rl = R + (4 * half) * INR_LDS_LD + col;
  const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(sv_dA, NBM * 32 * TL * 4);
  const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc(PAIR ? sv_dB : sv_dA, NBM * 32 * TL * 4);
  const int voff = ((4 * half) * TL + wcol) * 4;  // one per-lane byte offset for all loads
  if (!PAIR && NBM % 4 == 0) {
    // 64 loads (four row blocks) in flight before the first use: a block at a time exposed one HBM latency per block
    // (the phase stamps showed 20 k cycles here against 4 k for the plain image copies); all 16 * NBM at once is past
    // what the register allocator places without spilling
#pragma unroll
    for (int m0 = 0; m0 < NBM; m0 += 4) {
      float dq[4][16];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dq[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                   rsA, voff, (32 * (m0 + m) + (r & 3) + 8 * (r >> 2)) * TL * 4, 0));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Rl[(32 * (m0 + m) + (r & 3) + 8 * (r >> 2)) * INR_LDS_LD] = acc[m0 + m][r] * dq[m][r];
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  // (PAIR, or a block count that is not a multiple of four) two row blocks = 64 loads at a time when possible
  constexpr int MB = (PAIR && NBM % 2 == 0) ? 2 : 1;
#pragma unroll
  for (int m0 = 0; m0 < NBM; m0 += MB) {
    float dA[MB][16], dB[MB][16];
#pragma unroll
    for (int mm = 0; mm < MB; ++mm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int soff = (32 * (m0 + mm) + (r & 3) + 8 * (r >> 2)) * TL * 4;
        dA[mm][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, voff, soff, 0));
        dB[mm][r] = PAIR ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, voff, soff, 0)) : 0.f;
      }
    __builtin_amdgcn_sched_barrier(0);  // all loads of the blocks in flight before the first use
#pragma unroll
    for (int mm = 0; mm < MB; ++mm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mm;
        float v;
        if (PAIR) {
          const float pp = acc[m][r & ~1], qq = acc[m][r | 1];
          v = fmaf(pp, dA[mm][r], qq * dB[mm][r]);
        } else {
          v = acc[m][r] * dA[mm][r];
        }
        Rl[(32 * m + (r & 3) + 8 * (r >> 2)) * INR_LDS_LD] = v;
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Ask the scheduler for  N x { 1 MFMA, VALU_PER VALU }  in this order, so the VALU work of the next
// group is spread between this group's MFMAs instead of sitting in a block in front of them.
template <int N, int VALU_PER>
__device__ __forceinline__ void interleave_mfma_valu() {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER, 0);  // VALU
  }
}

// ---------------------------------------------------------------------------------------------
// forward, layer 0, fused gauss encoder (networks.py:30-33):
//   k-step s in [0,E): lane half 0 feeds sin(p_s) (feature s), half 1 feeds cos(p_s) (feature E+s)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gauss_feature(const float* encB_lds, int s, float xs0, float xs1, float xs2,
                                               int half) {
  // encoder rows come from LDS (wave-uniform broadcast reads): a vector global load here would
  // drag a vmcnt(0) behind it and drain the A-fragment prefetch every k-step
  const float b0 = encB_lds[3 * s + 0], b1 = encB_lds[3 * s + 1], b2 = encB_lds[3 * s + 2];
  const float ph = fmaf(xs2, b2, fmaf(xs1, b1, xs0 * b0));  // (2*pi*x) @ B^T, K = 3 (networks.py:31)
#ifndef INR_SINCOS_POLY
  // one transcendental: the reduced argument in revolutions (sincos_cw's reduction), + 1/4 turn for the cosine half
  const float k = rintf(ph * 0.15915494309189535f);
  float r = fmaf(k, -6.2831854820251465f, ph);
  r = fmaf(k, 1.7484555e-7f, r);
  return __builtin_amdgcn_sinf(fmaf(r, 0.15915494309189535f, half ? 0.25f : 0.f));
#else
  float sn, cs;
  sincos_cw(ph, sn, cs);
  return half ? cs : sn;
#endif
}

template <int NB, int TL, bool SAVE>
__device__ __forceinline__ void gauss_group(f32x16 (&acc)[NB], const f32x4 (&a_use)[NB], f32x4 (&a_load)[NB],
                                            const AFragPtr& p_next, const float* encB_lds, int s4_next, float xs0,
                                            float xs1, float xs2, int half, const float (&b_use)[4],
                                            float (&b_load)[4], __amdgpu_buffer_rsrc_t rs, int voff, int s4) {
  load_afrag<NB>(a_load, p_next);
  __builtin_amdgcn_sched_barrier(0);  // the next group's A fragments fly behind this group's MFMAs
#pragma unroll
  for (int e = 0; e < 4; ++e) b_load[e] = gauss_feature(encB_lds, 4 * s4_next + e, xs0, xs1, xs2, half);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (SAVE) stash_store(rs, voff, (4 * s4 + e) * TL * 4, b_use[e]);  // feature row (half ? E : 0) + 4*s4 + e, reused by dW_0
#pragma unroll
    for (int m = 0; m < NB; ++m) acc[m] = mfma32(a_use[m][e], b_use[e], acc[m]);
  }
  interleave_mfma_valu<4 * NB, 5>();
  __builtin_amdgcn_sched_barrier(0);
}

// NBT: row blocks of the whole packed image (stride between k-groups); NB of them, starting at the block
// `wp` already points to, are accumulated (NBT > NB: wide layers done in register-sized chunks)
template <int NB, int TL, bool SAVE, int NBT = NB>
__device__ __forceinline__ void fwd_layer0_gauss(f32x16 (&acc)[NB], const float* __restrict__ wp,
                                                 const float* encB_lds, int E, float xs0, float xs1, float xs2,
                                                 float* __restrict__ sv_enc, int wcol, int lane) {
  const int half = lane >> 5;
  const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(SAVE ? sv_enc : wp, 2 * E * TL * 4);
  const int voff = ((half ? E : 0) * TL + wcol) * 4;
  const AFragPtr p = afrag_ptr(wp, lane);
  const int n4 = E >> 2;  // even (E % 8 == 0)
  f32x4 A0[NB], A1[NB];
  float F0[4], F1[4];
  load_afrag<NB>(A0, p);
#pragma unroll
  for (int e = 0; e < 4; ++e) F0[e] = gauss_feature(encB_lds, e, xs0, xs1, xs2, half);
#pragma unroll 1
  for (int s4 = 0; s4 < n4; s4 += 2) {
    const int n2 = (s4 + 2 < n4) ? (s4 + 2) : s4;
    gauss_group<NB, TL, SAVE>(acc, A0, A1, p + (size_t)(s4 + 1) * NBT * 64, encB_lds, s4 + 1, xs0, xs1, xs2, half, F0,
                              F1, rs, voff, s4);
    gauss_group<NB, TL, SAVE>(acc, A1, A0, p + (size_t)n2 * NBT * 64, encB_lds, n2, xs0, xs1, xs2, half, F1, F0,
                              rs, voff, s4 + 1);
  }
}

// forward, layer 0, input matrix x [B,K0] in memory (what model.forward receives, train.py:169)
template <int NB, int NBT = NB>
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
    load_afrag<NB>(a_nxt, p + (size_t)nx * NBT * 64);
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
// forward, layer l >= 1: B operand = act(z_{l-1}) formed lazily from the LDS image, one group of
// four k-steps ahead of the MFMAs that consume it (four independent chains).
//   stash (SAVE): rows [feature][TL coords] of h, d (HACT != GABOR) or h, dA, dB (GABOR).
// ---------------------------------------------------------------------------------------------
struct ActParams {
  float w0;  // SIREN omega (30) / WIRE omega_0 of the producing layer
  float s0;  // WIRE scale_0 of the producing layer
};

// ZSTASH (WIRE2D on the two-waves-per-group kernel): only y is formed; `d` carries the lane's own pre-activation row
// (a for the Re row, b for the Im row) to the stash in the slot the Jacobian entry used to take -- every entry of the
// wavelet's Jacobian is y times a polynomial in (a, b, u, v), which the backward row passes evaluate from the stashed
// y, z and orth rows (three stashed tensors per layer instead of seven; the lazy loop loses 12 of its 20 stores per group).
template <int HACT, bool ZSTASH = false>
__device__ __forceinline__ void lazy_act(const float (&z)[4], const float (&zp)[4], const float (&zo)[4],
                                         const float (&zq)[4], const ActParams& ap, int half, float (&h)[4],
                                         float (&d)[4], float (&d2)[4], float (&d3)[4], float (&d4)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    d3[e] = d4[e] = 0.f;
    if (HACT == ACT_GABOR2D && ZSTASH) {
      float sn, cs;
      sincos_cw(ap.w0 * z[e], sn, cs);
      const float s2 = ap.s0 * ap.s0;
      const float Ef = expf(-ap.w0 * zp[e]);
      const float G = expf(-s2 * ((z[e] * z[e] + zp[e] * zp[e]) + zq[e]));
      h[e] = half ? (Ef * sn) * G : (Ef * cs) * G;  // (same association as act_gabor2d)
      d[e] = half ? zp[e] : z[e];
      d2[e] = 0.f;
    } else if (HACT == ACT_GABOR2D) {
      act_gabor2d(z[e], zp[e], zo[e], zq[e], ap.w0, ap.s0, half, h[e], d[e], d2[e], d3[e], d4[e]);
    } else if (HACT == ACT_GABOR) {
      // z = Re row value, zp = Im row value of the pair (both halves read both rows)
      act_gabor(z[e], zp[e], ap.w0, ap.s0, half, h[e], d[e], d2[e]);
    } else {
      act_fwd<HACT>(z[e], ap.w0, h[e], d[e]);
      d2[e] = 0.f;
    }
  }
}

// image rows of group s4 for a lane: PAIR: the pair rows 8*s4 + 2e (Re) and 8*s4 + 2e + 1 (Im);
// otherwise the lane's own row 8*s4 + 2e + half
template <bool PAIR>
__device__ __forceinline__ void load_z(float (&z)[4], float (&zp)[4], const float* Rcol, int s4, int half) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (PAIR) {
      z[e] = Rcol[(8 * s4 + 2 * e) * INR_LDS_LD];
      zp[e] = Rcol[(8 * s4 + 2 * e + 1) * INR_LDS_LD];
    } else {
      z[e] = Rcol[(8 * s4 + 2 * e + half) * INR_LDS_LD];
      zp[e] = 0.f;
    }
  }
}

// WIRE2D: own orth row value and the pair's u^2+v^2 of group s4, from the stash of the producing layer
template <int TL, bool ON>
__device__ __forceinline__ void load_oq(float (&zo)[4], float (&zq)[4], const float* __restrict__ svo, int hsz, int s4) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    zo[e] = ON ? svo[(8 * s4 + 2 * e) * TL] : 0.f;
    zq[e] = ON ? svo[hsz + (8 * s4 + 2 * e) * TL] : 0.f;
  }
}

// SAVE: 0 nothing, 1 h and the activation's derivative terms, 2 the derivative terms only (the fused step's last hidden
// layer: its h is only ever read by the last layer's dW, which re-activates z from the LDS images -- dw_rows4_valu).
// fwd_layer also takes 3 / 4: everything, but only for the even / odd groups of 8 rows -- the two waves of a pair
// (inr_mlp_wide_impl.h) both form the whole lazy activation and share the stash: each writes half of it
template <int NBOUT, int TL, int HACT, int SAVE, bool ALDS = false, bool ZSTASH = false, class AP = AFragPtr>
__device__ __forceinline__ void fwd_group(f32x16 (&acc)[NBOUT], const f32x4 (&a_use)[NBOUT],
                                          f32x4 (&a_load)[NBOUT], const AP& p_next, float (&z_buf)[4],
                                          float (&zp_buf)[4], float (&zo_buf)[4], float (&zq_buf)[4],
                                          const float* Rcol, int s4_next2, int s4,
                                          const ActParams& ap, int half, float* __restrict__ svl, int hsz,
                                          __amdgpu_buffer_rsrc_t rs, int voff,
                                          const float (&h_use)[4], const float (&d_use)[4],
                                          const float (&d2_use)[4], const float (&d3_use)[4],
                                          const float (&d4_use)[4], float (&h_load)[4], float (&d_load)[4],
                                          float (&d2_load)[4], float (&d3_load)[4], float (&d4_load)[4]) {
  // z_buf holds the pre-activations of group s4+1 (fetched one group ago); they become h_load/d_load
  // during this group's MFMAs, and z_buf is refilled with group s4+2.
  constexpr bool G2D = HACT == ACT_GABOR2D;
  if constexpr (ALDS)
    a_load[0] = *p_next;  // one-block last layer: its live rows' fragments sit in LDS (see fwd_layer)
  else
    load_afrag<NBOUT>(a_load, p_next);
  float z_next[4], zp_next[4], zo_next[4], zq_next[4];
  load_z<HACT == ACT_GABOR || G2D>(z_next, zp_next, Rcol, s4_next2, half);
  load_oq<TL, G2D>(zo_next, zq_next, svl + 5 * hsz, hsz, s4_next2);
  __builtin_amdgcn_sched_barrier(0);  // operands of the following groups are in flight behind the MFMAs
  lazy_act<HACT, ZSTASH>(z_buf, zp_buf, zo_buf, zq_buf, ap, half, h_load, d_load, d2_load, d3_load, d4_load);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (SAVE) {
      const int so = (8 * s4 + 2 * e) * TL * 4;  // bytes; the tensors of a layer are hsz floats apart
      if (SAVE == 1) stash_store(rs, voff, so, h_use[e]);
      stash_store(rs, voff, so + hsz * 4, d_use[e]);
      if ((HACT == ACT_GABOR || G2D) && !ZSTASH) stash_store(rs, voff, so + 2 * hsz * 4, d2_use[e]);
      if (G2D && !ZSTASH) {
        stash_store(rs, voff, so + 3 * hsz * 4, d3_use[e]);
        stash_store(rs, voff, so + 4 * hsz * 4, d4_use[e]);
      }
    }
#pragma unroll
    for (int m = 0; m < NBOUT; ++m) acc[m] = mfma32(a_use[m][e], h_use[e], acc[m]);
  }
  if (NBOUT >= 4) interleave_mfma_valu<4 * NBOUT, (40 + NBOUT - 1) / NBOUT>();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    z_buf[e] = z_next[e];
    zp_buf[e] = zp_next[e];
    zo_buf[e] = zo_next[e];
    zq_buf[e] = zq_next[e];
  }
}

// ALDS (NBOUT == 1, the last layer): `wp` is the LDS copy of the fragments of output rows 0..3 -- float4
// (s4 * 2 + half) * 4 + row, then one float4 of zeros that the lanes of rows 4..31 read.  With the fragments in L2 the
// loop ran at one L2 + store-acknowledge latency per group: four MFMAs do not cover a load that waits, in the in-order
// vmcnt, behind the eight stash stores issued before it.
template <int NB, int NBOUT, int TL, int HACT, int SAVE, int NBT = NBOUT, bool ALDS = false, bool ZSTASH = false>
__device__ __forceinline__ void fwd_layer(f32x16 (&acc)[NBOUT], const float* R, const float* wp,
                                          const ActParams& ap, float* __restrict__ sv, int wcol, int lane) {
  static_assert(!ALDS || NBOUT == 1, "LDS fragments: one-block layers only");
  const int half = lane >> 5, col = lane & 31;
  constexpr int n4tot = NB * 4;
  // fragment stream: LDS pointer (ALDS: per-lane stride, rows >= 4 read the zero slot) or buffer-descriptor "pointer"
  auto make_p = [&]() {
    if constexpr (ALDS)
      return reinterpret_cast<const f32x4*>(wp) + (col < 4 ? half * 4 + col : n4tot * 8);
    else
      return afrag_ptr(wp, lane);
  };
  const auto p = make_p();
  const int gstride = ALDS ? (col < 4 ? 8 : 0) : NBT * 64;  // float4s between consecutive groups
  constexpr int n4 = NB * 4;  // K = 32*NB features -> 16*NB k-steps -> 4*NB groups of 4 (even)
  constexpr int hsz = NB * 32 * TL;
  constexpr bool G2D = HACT == ACT_GABOR2D;  // always called with SAVE: its orth inputs live in the stash
  constexpr bool PAIR = HACT == ACT_GABOR || G2D;
  const float* Rcol = R + col;
  float* svl = (SAVE || G2D) ? sv + half * TL + wcol : nullptr;  // stash row k = 8*s4 + 2e + half
  const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(SAVE ? (const void*)sv : (const void*)wp, 5 * hsz * 4);
  const int voff = (half * TL + wcol) * 4;
  f32x4 A0[NBOUT], A1[NBOUT];
  float Z[4], ZP[4], ZO[4], ZQ[4], H0[4], D0[4], E0[4], F0[4], G0[4], H1[4], D1[4], E1[4], F1[4], G1[4];
  if constexpr (ALDS)
    A0[0] = *p;
  else
    load_afrag<NBOUT>(A0, p);
  load_z<PAIR>(Z, ZP, Rcol, 0, half);
  load_oq<TL, G2D>(ZO, ZQ, svl + 5 * hsz, hsz, 0);
  lazy_act<HACT, ZSTASH>(Z, ZP, ZO, ZQ, ap, half, H0, D0, E0, F0, G0);
  load_z<PAIR>(Z, ZP, Rcol, 1, half);  // group 1 (n4 >= 4)
  load_oq<TL, G2D>(ZO, ZQ, svl + 5 * hsz, hsz, 1);
#pragma unroll 1
  for (int s4 = 0; s4 < n4; s4 += 2) {
    const int n2 = (s4 + 2 < n4) ? (s4 + 2) : s4;
    const int n3 = (s4 + 3 < n4) ? (s4 + 3) : s4;
    constexpr int SA = SAVE <= 2 ? SAVE : (SAVE == 3 ? 1 : 0);  // group s4 (even)
    constexpr int SB = SAVE <= 2 ? SAVE : (SAVE == 4 ? 1 : 0);  // group s4 + 1 (odd)
    fwd_group<NBOUT, TL, HACT, SA, ALDS, ZSTASH, decltype(p)>(acc, A0, A1, p + (size_t)(s4 + 1) * gstride, Z, ZP, ZO, ZQ, Rcol, n2, s4, ap,
                                           half, svl, hsz, rs, voff, H0, D0, E0, F0, G0, H1, D1, E1, F1, G1);
    fwd_group<NBOUT, TL, HACT, SB, ALDS, ZSTASH, decltype(p)>(acc, A1, A0, p + (size_t)n2 * gstride, Z, ZP, ZO, ZQ, Rcol, n3, s4 + 1, ap,
                                           half, svl, hsz, rs, voff, H1, D1, E1, F1, G1, H0, D0, E0, F0, G0);
  }
}

// ---------------------------------------------------------------------------------------------
// The last layer (<= 4 output rows) of a real-activation network on the VECTOR ALUs.  As an MFMA layer it is one
// 32-row block of which 2 rows are real: 128 MFMAs (8.2 k cycles on a matrix pipe that shares the FP32 ALUs) for
// 16 k useful multiply-adds per wave.  Here a lane keeps its coordinate's partial sums over the features of its half
// (k = 8 s4 + 2e + half, the rows the lazy activation gives it anyway): M FMAs per activated element, weights from
// the LDS fragment copy `ll` (float4 (s4*2 + half)*4 + row = W[row][8 s4 + 2e + half], e = 0..3: a broadcast read),
// the two halves added at the end.  out4[o], o < M, is complete in every lane.  SAVE as in fwd_group; HBACK: h is
// written back over z in the image (the fused step's dW of this layer reads it there, dw_rows4_valu).
// ---------------------------------------------------------------------------------------------
template <int NB, int TL, int HACT, int SAVE, bool HBACK, bool WIDE>
__device__ __forceinline__ void fwd_last_valu_impl(float (&out4)[4], float* R, const float* ll, const ActParams& ap,
                                                   float* __restrict__ sv, int wcol, int lane) {
  const int half = lane >> 5, col = lane & 31;
  constexpr int n4 = NB * 4;
  constexpr int hsz = NB * 32 * TL;
  typedef __attribute__((address_space(3))) float lfloat;
  typedef const __attribute__((address_space(3))) f32x4 lf4;
  lfloat* Rz = (lfloat*)(R + col + half * INR_LDS_LD);  // row 8 s4 + 2e + half, this lane's column
  lf4* wq = (lf4*)ll + half * 4;
  const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(SAVE ? (const void*)sv : (const void*)ll, 2 * hsz * 4);
  const int voff = (half * TL + wcol) * 4;
  constexpr bool wide = WIDE;  // rows 2, 3 exist (a branch per group otherwise)
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int s4 = 0; s4 < n4; ++s4) {
    float z[4];
    f32x4 wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = Rz[(8 * s4 + 2 * e) * INR_LDS_LD];
    wv[0] = wq[s4 * 8 + 0];
    wv[1] = wq[s4 * 8 + 1];
    if (wide) {
      wv[2] = wq[s4 * 8 + 2];
      wv[3] = wq[s4 * 8 + 3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float h, d;
      act_fwd<HACT>(z[e], ap.w0, h, d);
      if (SAVE) {
        const int so = (8 * s4 + 2 * e) * TL * 4;
        if (SAVE == 1) stash_store(rs, voff, so, h);
        stash_store(rs, voff, so + hsz * 4, d);
      }
      if (HBACK) Rz[(8 * s4 + 2 * e) * INR_LDS_LD] = h;
      acc[0] = fmaf(wv[0][e], h, acc[0]);
      acc[1] = fmaf(wv[1][e], h, acc[1]);
      if (wide) {
        acc[2] = fmaf(wv[2][e], h, acc[2]);
        acc[3] = fmaf(wv[3][e], h, acc[3]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) out4[o] = (o < 2 || WIDE) ? acc[o] + __shfl_xor(acc[o], 32) : 0.f;
}

template <int NB, int TL, int HACT, int SAVE, bool HBACK>
__device__ __forceinline__ void fwd_last_valu(float (&out4)[4], float* R, const float* ll, int M, const ActParams& ap,
                                              float* __restrict__ sv, int wcol, int lane) {
  if (M > 2)
    fwd_last_valu_impl<NB, TL, HACT, SAVE, HBACK, true>(out4, R, ll, ap, sv, wcol, lane);
  else
    fwd_last_valu_impl<NB, TL, HACT, SAVE, HBACK, false>(out4, R, ll, ap, sv, wcol, lane);
}

// ---------------------------------------------------------------------------------------------
// WIRE2D helpers.  acc += A . h^T with h read back from the stash (natural k order: k-step s -> rows 2s, 2s+1),
// the second Linear of a layer; its epilogue leaves the orth pre-activations and u^2+v^2 of each pair in the
// stash slots the next layer's lazy activation reads.
// ---------------------------------------------------------------------------------------------
// NBK: row blocks of h (the k extent); NBT: row blocks of the whole packed image (stride between k-groups)
template <int NB, int TL, int NBK = NB, int NBT = NB>
__device__ __forceinline__ void gemm_stash_nat(f32x16 (&acc)[NB], const float* __restrict__ wp,
                                               const float* __restrict__ sv_h, int wcol, int lane) {
  const int half = lane >> 5;
  const float* svl = sv_h + half * TL + wcol;
  const f32x4* p = reinterpret_cast<const f32x4*>(wp) + lane;
  constexpr int n4 = NBK * 4;
  f32x4 A0[NB], A1[NB];
  float B0[4], B1[4];
  load_afrag<NB>(A0, p);
#pragma unroll
  for (int e = 0; e < 4; ++e) B0[e] = svl[(2 * e) * TL];
#pragma unroll 1
  for (int s4 = 0; s4 < n4; s4 += 2) {
    const int n2 = (s4 + 2 < n4) ? (s4 + 2) : s4;
    load_afrag<NB>(A1, p + (size_t)(s4 + 1) * NBT * 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) B1[e] = svl[(8 * (s4 + 1) + 2 * e) * TL];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma32(A0[m][e], B0[e], acc[m]);
    __builtin_amdgcn_sched_barrier(0);
    load_afrag<NB>(A0, p + (size_t)n2 * NBT * 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) B0[e] = svl[(8 * n2 + 2 * e) * TL];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma32(A1[m][e], B1[e], acc[m]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// registers (2p, 2p+1) of a lane are the (u, v) rows of one complex feature
template <int NB, int TL, int NBT = NB>
__device__ __forceinline__ void orth_epilogue(const f32x16 (&acc)[NB], const float* __restrict__ bias,
                                              float* __restrict__ svO, int wcol, int lane) {
  const int half = lane >> 5;
  constexpr int hsz = NBT * 32 * TL;
  float* so = svO + (4 * half) * TL + wcol;
  const float* bl = bias + 4 * half;
#pragma unroll
  for (int m = 0; m < NB; ++m) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + 32 * m + 8 * g);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[m][4 * g + j] + b4[j];
      const float q01 = v[0] * v[0] + v[1] * v[1], q23 = v[2] * v[2] + v[3] * v[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        so[(32 * m + 8 * g + j) * TL] = v[j];
        so[hsz + (32 * m + 8 * g + j) * TL] = j < 2 ? q01 : q23;
      }
    }
  }
}

// wave image <-> global scratch [NB*32 rows][TL] (own column): the layer's output gradient is needed twice
template <int NB, int TL, bool TO_GLOBAL>
__device__ __forceinline__ void image_copy(float* R, float* __restrict__ G, int wcol, int lane) {
  // 16 bytes per lane: eight lanes cover the 32 coordinates of a row, a wave eight rows per instruction pair
  const int seg = lane & 7, rr = lane >> 3;
  float* g = G + (wcol & ~31) + 4 * seg;
#pragma unroll 8
  for (int r = rr; r < NB * 32; r += 8) {
    if (TO_GLOBAL)
      *reinterpret_cast<f32x4*>(g + r * TL) = *reinterpret_cast<const f32x4*>(R + swz(r, 4 * seg));
    else
      *reinterpret_cast<f32x4*>(R + swz(r, 4 * seg)) = *reinterpret_cast<const f32x4*>(g + r * TL);
  }
}

// ---------------------------------------------------------------------------------------------
// backward: dH_{l-1}^T = W_l^T . dZ_l^T.  R holds dH_l (HASD: turned in place into dZ_l with the
// stashed act', which dW then reads) or dZ_l itself (last layer).  k extent = Mpad8 of layer l.
// ---------------------------------------------------------------------------------------------
template <int NB, int TL, bool PAIR, bool HASD>
__device__ __forceinline__ void dx_group(f32x16 (&acc)[NB], const f32x4 (&a_use)[NB], f32x4 (&a_load)[NB],
                                         const AFragPtr& p_next, const float (&g_use)[4], const float (&gp_use)[4],
                                         float (&g_load)[4], float (&gp_load)[4], const float (&d_use)[4],
                                         const float (&d2_use)[4], float (&d_load)[4], float (&d2_load)[4],
                                         float* Rcol, int s4, int s4_next, __amdgpu_buffer_rsrc_t rd, int voff_d, int hsz,
                                         int half, bool prefetch) {
  if (prefetch) {
    load_afrag<NB>(a_load, p_next);
    load_z<HASD && PAIR>(g_load, gp_load, Rcol, s4_next, half);
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // act' entries of the lane's row through the descriptor: row offset in an SGPR
      const int so = (8 * s4_next + 2 * e) * TL * 4;
      d_load[e] = HASD ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, voff_d, so, 0)) : 1.f;
      d2_load[e] = (HASD && PAIR) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, voff_d, so + hsz * 4, 0)) : 0.f;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float g = g_use[e];
    if (HASD) {
      if (PAIR)
        g = fmaf(g_use[e], d_use[e], gp_use[e] * d2_use[e]);  // p*dA + q*dB (g_use = Re row, gp_use = Im row)
      else
        g *= d_use[e];
      // dZ_l, read again by dW.  Rows of the prefetched group are untouched; for PAIR both halves
      // have already read this pair (the loads were issued one group ago, LDS ops stay in order).
      Rcol[(8 * s4 + 2 * e + half) * INR_LDS_LD] = g;
    }
#pragma unroll
    for (int m = 0; m < NB; ++m) acc[m] = mfma32(a_use[m][e], g, acc[m]);
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int NB, int TL, bool PAIR, bool HASD, int NBT = NB>
__device__ __forceinline__ void bwd_dx(f32x16 (&acc)[NB], float* R, const float* __restrict__ wpT, int Mpad8,
                                       const float* __restrict__ sv_d, int wcol, int lane) {
  const int half = lane >> 5, col = lane & 31;
  const AFragPtr p = afrag_ptr(wpT, lane);
  const int n4 = Mpad8 >> 3;
  constexpr int hsz = NB * 32 * TL;
  float* Rcol = R + col;
  // own row's act' entries (the two tensors of a PAIR layer are hsz floats apart)
  const __amdgpu_buffer_rsrc_t rd = uniform_rsrc(HASD ? (const void*)sv_d : (const void*)wpT, 2 * hsz * 4);
  const int voff_d = (half * TL + wcol) * 4;
  f32x4 A0[NB], A1[NB];
  float G0[4], P0[4], G1[4], P1[4], D0[4], E0[4], D1[4], E1[4];
  load_afrag<NB>(A0, p);
  load_z<HASD && PAIR>(G0, P0, Rcol, 0, half);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    D0[e] = HASD ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, voff_d, (2 * e) * TL * 4, 0)) : 1.f;
    E0[e] = (HASD && PAIR) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, voff_d, ((2 * e) * TL + hsz) * 4, 0)) : 0.f;
  }
  if (n4 == 1) {  // last layer: out_features <= 8 -> a single group
    dx_group<NB, TL, PAIR, HASD>(acc, A0, A1, p, G0, P0, G1, P1, D0, E0, D1, E1, Rcol, 0, 0, rd, voff_d, hsz, half, false);
    return;
  }
#pragma unroll 1
  for (int s4 = 0; s4 < n4; s4 += 2) {  // n4 even for hidden layers
    const int n2 = (s4 + 2 < n4) ? (s4 + 2) : s4;
    dx_group<NB, TL, PAIR, HASD>(acc, A0, A1, p + (size_t)(s4 + 1) * NBT * 64, G0, P0, G1, P1, D0, E0, D1, E1, Rcol, s4,
                                 s4 + 1, rd, voff_d, hsz, half, true);
    dx_group<NB, TL, PAIR, HASD>(acc, A1, A0, p + (size_t)n2 * NBT * 64, G1, P1, G0, P0, D1, E1, D0, E0, Rcol, s4 + 1,
                                 n2, rd, voff_d, hsz, half, true);
  }
}

// ---------------------------------------------------------------------------------------------
// dW pass: MT row blocks x one 32-column block n, contraction over the tile's TL coordinates.
//   A[k=coord][i=out feature] from the waves' LDS images (lane = feature),
//   B[k=coord][j=in feature]  from the source functor (lane = feature),
//   k order: group q of 8 coordinates -> half h takes coords 8q+4h+(0..3) as 4 k-steps.
// ---------------------------------------------------------------------------------------------
template <int TL>
struct BSrcStash {  // h_{l-1} stash [feature][TL]
  const float* __restrict__ h;
  struct Raw {
    f32x4 v;
  };
  __device__ __forceinline__ Raw fetch(int n, int q, int lane) const {
    const int j = 32 * n + (lane & 31);
    typedef const __attribute__((address_space(1))) f32x4* gptr;  // (global, not generic: see BSrcX::fetch)
    return Raw{*(gptr)(h + j * TL + 8 * q + 4 * (lane >> 5))};
  }
  __device__ __forceinline__ f32x4 finish(const Raw& r) const { return r.v; }
};

struct BSrcX {  // x [B,K0] row-major
  const float* __restrict__ x;
  long long row0, B;
  int K0;
  struct Raw {
    f32x4 v;
    unsigned ok;  // bit e: element e is inside the matrix
  };
  // (the selects happen in finish(), a group later: "load, select" per element put a vmcnt(0) behind every load)
  __device__ __forceinline__ Raw fetch(int n, int q, int lane) const {
    const int j = 32 * n + (lane & 31);
    Raw r;
    r.ok = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long row = row0 + 8 * q + 4 * (lane >> 5) + e;
      const bool ok = j < K0 && row < B;
      // (global address space spelled out: through the struct reference of the non-inlined pass the pointer is generic,
      // and a flat load counts on both vmcnt and lgkmcnt -- every wait behind one is a full drain)
      r.v[e] = ((const __attribute__((address_space(1))) float*)x)[ok ? row * K0 + j : 0];
      r.ok |= ok ? (1u << e) : 0u;
    }
    return r;
  }
  __device__ __forceinline__ f32x4 finish(const Raw& r) const {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (r.ok >> e) & 1u ? r.v[e] : 0.f;
    return o;
  }
};

// the 4 k-steps of one row block: A[coord 8q+4h+e][feature 32m+li], e = 0..3 (contiguous in the image row)
template <int MT>
__device__ __forceinline__ void load_dw_a(f32x4 (&a)[MT], const float* Rq) {
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if ((INR_LDS_LD & 3) == 0) {
      // (LDS address space spelled out: inside the non-inlined passes Rq is a generic pointer and this would be a flat load)
      a[m] = *(const __attribute__((address_space(3))) f32x4*)(Rq + 32 * m * INR_LDS_LD);  // ds_read_b128
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[m][e] = Rq[32 * m * INR_LDS_LD + e];
    }
  }
}

template <int MT, bool BIAS, class BSrc>
__device__ __forceinline__ void dw_group(f32x16 (&acc)[MT], float (&bsum)[MT], const f32x4 (&a_use)[MT],
                                         f32x4 (&a_load)[MT], const f32x4& b_use, f32x4& b_load, BSrc& bsrc,
                                         int n, int q_next, const float* Rq_next, int lane) {
  const typename BSrc::Raw raw = bsrc.fetch(n, q_next, lane);
  load_dw_a<MT>(a_load, Rq_next);
  __builtin_amdgcn_sched_barrier(0);
  b_load = bsrc.finish(raw);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (BIAS) bsum[m] += a_use[m][e];
      acc[m] = mfma32(a_use[m][e], b_use[e], acc[m]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// INR_DW_ATTR: a translation unit whose kernels are register-bound (WIRE, MFN) compiles the pass as a real
// function: it is self-contained (operands from LDS / the stash, results to the slab), so a call costs
// nothing measurable and keeps the pass's ~200 registers and its hoisted addresses out of the
// allocation of the surrounding kernel.
#ifndef INR_DW_ATTR
#define INR_DW_ATTR __forceinline__
#endif
template <int MT, int TL, bool FULLM, bool BIAS, class BSrc>
__device__ INR_DW_ATTR void dw_pass_impl(const float* Rall, int region_stride, BSrc& bsrc, int n, float* slab_w_generic,
                                             float* slab_b_generic, int M, int K, bool first, int lane) {
  // (compiled as a real function in the WIRE / filter-network units, the pass sees its pointer arguments as GENERIC:
  // flat loads and stores, which count on vmcnt and lgkmcnt both and make every wait a full drain.  The slabs are global.)
  typedef __attribute__((address_space(1))) float gfloat;
  gfloat* slab_w = (gfloat*)slab_w_generic;
  gfloat* slab_b = (gfloat*)slab_b_generic;
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
      float v[16];  // (all 16 loads first: "load, select" per element became a vmcnt(0) behind every load)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowu = 32 * m + (r & 3) + 8 * (r >> 2);  // uniform part of the row
        const bool ok = colok && (FULLM || rowu + 4 * half < M);
        const gfloat* rowp = slab_w + (size_t)(FULLM ? rowu : 0) * K;
        v[r] = rowp[ok ? (FULLM ? lane_off : rowu * K + lane_off) : 0];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowu = 32 * m + (r & 3) + 8 * (r >> 2);
        const bool ok = colok && (FULLM || rowu + 4 * half < M);
        acc[m][r] = ok ? v[r] : 0.f;
      }
    }
  }
  const float* Rl = Rall + li * INR_LDS_LD + 4 * half;
  // software pipeline over groups q of 8 coordinates: operands of group q+1 are fetched while
  // group q is multiplied; coordinate 8q + 4*half + e lives in wave image q>>2, column 8(q&3)+4*half+e
  if (MT == 1 && TL == 64) {
    // one row block on the 64-coordinate tiles (last layer / heads of the two-waves-per-group kernels): 4 MFMAs per group
    // cannot cover a stash load fetched one group ahead -- all eight B operands are requested first
    typename BSrc::Raw raws[TL / 8];
#pragma unroll
    for (int q = 0; q < TL / 8; ++q) raws[q] = bsrc.fetch(n, q, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < TL / 8; ++q) {
      f32x4 A[MT];
      load_dw_a<MT>(A, Rl + (q >> 2) * region_stride + 8 * (q & 3));
      const f32x4 B = bsrc.finish(raws[q]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (BIAS) bsum[0] += A[0][e];
        acc[0] = mfma32(A[0][e], B[e], acc[0]);
      }
    }
  } else {
    f32x4 B0 = bsrc.finish(bsrc.fetch(n, 0, lane)), B1;
    f32x4 A0[MT], A1[MT];
    load_dw_a<MT>(A0, Rl);
#pragma unroll 1
    for (int q = 0; q < TL / 8; q += 2) {
      const int q2 = (q + 2 < TL / 8) ? q + 2 : q;
      dw_group<MT, BIAS, BSrc>(acc, bsum, A0, A1, B0, B1, bsrc, n, q + 1,
                               Rl + ((q + 1) >> 2) * region_stride + 8 * ((q + 1) & 3), lane);
      dw_group<MT, BIAS, BSrc>(acc, bsum, A1, A0, B1, B0, bsrc, n, q2, Rl + (q2 >> 2) * region_stride + 8 * (q2 & 3),
                               lane);
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (FULLM) {
      if (colok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          gfloat* rowp = slab_w + (size_t)(32 * m + (r & 3) + 8 * (r >> 2)) * K;
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
    if (BIAS) {
      const float tot = bsum[m] + __shfl_xor(bsum[m], 32);
      const int row = 32 * m + li;
      if (half == 0 && (FULLM || row < M)) slab_b[row] = first ? tot : slab_b[row] + tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dW / db of a layer with at most 4 output rows (the last layer: 2 rows, 4 for the complex heads) on the VECTOR ALUs.
// As an MFMA pass this is one 32-row block of which 2-4 rows are real -- 64 MFMAs per column block whose operands
// arrive one stash load per four of them: 26 k cycles of a 746 k tile for 0.1 % of its FLOPs, all latency.  Here a
// group of TL/4 lanes reads one row of h_{l-1} (TL coordinates, contiguous in the stash: 512 B, a float4 per lane),
// multiplies by the lane's slice of the dZ rows (loaded once from the waves' LDS images) and the group's partial sums
// are added across lanes with DPP adds; 16 row loads are in flight per wave.  Wave w takes rows (it NW + w) RPI + sub.
// The sum of a row lands in all lanes of the LAST 16-lane row of its group; iteration `it` is kept in lane (it & 15)
// of that row, so the slab is touched with M loads / stores per 16 iterations instead of one per row.
// Fixed summation order: deterministic.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, t);
}

// sums over the TL/4 lanes of a group (32: a half wave, 16: one DPP row) of N independent values, step by step across
// all of them (a DPP operand needs two wait states behind the instruction that wrote it: N chains in lockstep fill
// them); complete in the lanes of each group's last 16-lane row
template <int LPR, int N>
__device__ __forceinline__ void group_sum_n(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add<0xB1, 0xf>(v[i]);  // quad_perm [1,0,3,2]
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add<0x4E, 0xf>(v[i]);  // quad_perm [2,3,0,1]
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add<0x141, 0xf>(v[i]);  // row_half_mirror
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add<0x140, 0xf>(v[i]);  // row_mirror
  if (LPR == 32) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x142, 0xa>(v[i]);  // row_bcast:15 into rows 1 and 3
  }
}

// shift register along each 16-lane row: lanes 1..15 take their left neighbour's `keep`, lane 0 takes `v`
__device__ __forceinline__ float row_push(float keep, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, keep),
                                                               0x111 /* row_shr:1 */, 0xf, 0xf, false));
}

// ZACT < 0: h rows from the stash `h`, the dZ rows are rows 0..3 of the waves' images (lds_img, region_stride).
// ZACT >= 0 (fused step): the rows are in the waves' LDS images -- h_{l-1} itself, written back over z by fwd_last_valu
// (ZACT = ACT_ID), or z_{l-1} to be activated again on the way -- the dZ rows come from their own small images
// `dz_img` (dz_stride apart), nothing is read from memory:
// with all workgroups in lockstep the stash version read the whole h_{D-2} of the batch in one burst, 2.8 TB/s for
// 21 k cycles with nothing to overlap.
// MO: output rows computed (2, or 4 for the complex heads).  The sum of a row is pushed into a per-row shift register
// (row_push): after a batch of 16 iterations lane l of a group's last 16-lane row holds iteration 15 - l, and the slab
// is touched with MO loads / stores per batch.
template <int TL, int NWAVES, int ZACT, int MO>
__device__ __forceinline__ void dw_rows4_valu_impl(const float* lds_img, int region_stride, const float* __restrict__ h,
                                                   int h_rows, int M, int K, float* slab_w, float* slab_b, bool first,
                                                   int w, int lane, const float* dz_img, int dz_stride, float w0) {
  constexpr int LPR = TL / 4;    // lanes per row of h
  constexpr int RPI = 64 / LPR;  // rows per wave-wide load
  constexpr int BATCH = 16;
  static_assert(LPR == 32 || LPR == 16, "TL is 128 or 64");
  typedef const __attribute__((address_space(3))) f32x4 lf4;
  const int sub = lane / LPR, li = lane % LPR;
  // this lane's 4 coordinates of the dZ rows: coordinate 4 li lives in wave image li >> 3, column 4 (li & 7)
  f32x4 dz[MO];
  {
    const float* q = ZACT >= 0 ? dz_img + (li >> 3) * dz_stride + 4 * (li & 7)
                               : lds_img + (li >> 3) * region_stride + 4 * (li & 7);
#pragma unroll
    for (int o = 0; o < MO; ++o) dz[o] = *(lf4*)(q + o * INR_LDS_LD);
  }
  const float* zq = lds_img + (li >> 3) * region_stride + 4 * (li & 7);  // this lane's columns of the images
  const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(ZACT >= 0 ? (const void*)slab_w : (const void*)h, h_rows * TL * 4);
  const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(slab_w, M * K * 4);
  const int voff = (sub * TL + 4 * li) * 4;
  const int rows_per_it = NWAVES * RPI;
  const int n_it = (K + rows_per_it - 1) / rows_per_it;
  const bool res_lane = LPR == 32 ? (lane & 16) != 0 : true;  // lanes that end up holding complete sums
  const int bl = 15 - (lane & 15);                            // the iteration of a batch this lane keeps
  // (rows past h_rows: the descriptor's bound returns zeros; rows in [K, h_rows) are padding and never stored)
#pragma unroll 1
  for (int it0 = 0; it0 < n_it; it0 += BATCH) {
    f32x4 hv[BATCH];
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const int row0 = ((it0 + b) * NWAVES + w) * RPI;
      if (ZACT >= 0) {
        const int jr = row0 + sub < h_rows ? row0 + sub : 0;
        hv[b] = *(lf4*)(zq + jr * INR_LDS_LD);
      } else {
        hv[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, row0 * TL * 4, 0));
      }
    }
    const int j = ((it0 + bl) * NWAVES + w) * RPI + sub;  // the row whose sums this lane keeps
    const bool jok = res_lane && j < K && it0 + bl < n_it;
    int so[MO];  // byte offset of slab entry (o, j); out of the descriptor's range for lanes without one
    float old[MO], keep[MO];
#pragma unroll
    for (int o = 0; o < MO; ++o) {
      so[o] = (jok && o < M) ? (o * K + j) * 4 : 0x7ffffff0;
      keep[o] = 0.f;
      old[o] = first ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsw, so[o], 0, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < BATCH; b += 2) {
      float p[2 * MO];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x4 hh = hv[b + u];
        if (ZACT > 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a, dd;
            act_fwd<(ZACT > 0 ? ZACT : 0)>(hh[e], w0, a, dd);
            hh[e] = a;
          }
        }
#pragma unroll
        for (int o = 0; o < MO; ++o) {
          float t = hh[0] * dz[o][0];
          t = fmaf(hh[1], dz[o][1], t);
          t = fmaf(hh[2], dz[o][2], t);
          p[u * MO + o] = fmaf(hh[3], dz[o][3], t);
        }
      }
      group_sum_n<LPR, 2 * MO>(p);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int o = 0; o < MO; ++o) keep[o] = row_push(keep[o], p[u * MO + o]);
    }
#pragma unroll
    for (int o = 0; o < MO; ++o)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, old[o] + keep[o]), rsw, so[o], 0, 0);
  }
  if (w == 0) {  // db: the row sums of dZ over the tile's coordinates (lanes of group 0 hold all TL of them)
    float sdz[MO];
#pragma unroll
    for (int o = 0; o < MO; ++o) sdz[o] = (dz[o][0] + dz[o][1]) + (dz[o][2] + dz[o][3]);
    group_sum_n<LPR, MO>(sdz);
    typedef __attribute__((address_space(1))) float gfloat;
    gfloat* sb = (gfloat*)slab_b;
#pragma unroll
    for (int o = 0; o < MO; ++o)
      if (lane == LPR - 1 && o < M) sb[o] = first ? sdz[o] : sb[o] + sdz[o];
  }
}

template <int TL, int NWAVES, int ZACT = -1>
__device__ __forceinline__ void dw_rows4_valu(const float* lds_img, int region_stride, const float* __restrict__ h,
                                              int h_rows, int M, int K, float* slab_w, float* slab_b, bool first, int w,
                                              int lane, const float* dz_img = nullptr, int dz_stride = 0,
                                              float w0 = 0.f) {
  if (M > 2)
    dw_rows4_valu_impl<TL, NWAVES, ZACT, 4>(lds_img, region_stride, h, h_rows, M, K, slab_w, slab_b, first, w, lane,
                                            dz_img, dz_stride, w0);
  else
    dw_rows4_valu_impl<TL, NWAVES, ZACT, 2>(lds_img, region_stride, h, h_rows, M, K, slab_w, slab_b, first, w, lane,
                                            dz_img, dz_stride, w0);
}

// the column-block-0 pass also produces db (the row sums of dZ), the others skip that VALU work
template <int MT, int TL, bool FULLM, class BSrc>
__device__ __forceinline__ void dw_pass(const float* Rall, int region_stride, BSrc& bsrc, int n, float* slab_w,
                                        float* slab_b, int M, int K, bool first, bool do_bias, int lane) {
  if (do_bias)
    dw_pass_impl<MT, TL, FULLM, true, BSrc>(Rall, region_stride, bsrc, n, slab_w, slab_b, M, K, first, lane);
  else
    dw_pass_impl<MT, TL, FULLM, false, BSrc>(Rall, region_stride, bsrc, n, slab_w, slab_b, M, K, first, lane);
}

// ---------------------------------------------------------------------------------------------
// the kernel.  MODE 0: forward (save optional); 1: backward from dout + save; 2: fused
// forward + pointwise loss + backward.
// ---------------------------------------------------------------------------------------------
// Diagnostic phase stamps (never compiled into the shipped library): s_memtime per wave at phase
// boundaries, written to a buffer nothing else reads; s_memrealtime beside the first and the latest one, so that
// (stamp_last - stamp_0) / (slot 63 - slot 62) x 100 MHz is the clock the wave ran at (MI355X_MICROARCH.md, DVFS (6)).
#ifdef INR_STAMPS
#define INR_STAMP(i)                                                                               \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    const long long stamp_at_ = ((long long)blockIdx.x * NW + w) * 64 + (i);                       \
    if (a.dbg != nullptr && lane == 0 && (i) < 62 && stamp_at_ - (i) + 63 < a.dbg_cap) {           \
      a.dbg[stamp_at_] = (long long)__builtin_amdgcn_s_memtime();                                  \
      /* slots 62 / 63: the 100 MHz counter at the wave's first and latest stamp (in-kernel clock) */ \
      const long long rt_ = (long long)__builtin_amdgcn_s_memrealtime();                           \
      if ((i) == 0) a.dbg[stamp_at_ + 62] = rt_;                                                   \
      a.dbg[stamp_at_ - (i) + 63] = rt_;                                                           \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#else
#define INR_STAMP(i) \
  do {               \
  } while (0)
#endif

template <int NB, int NW, int INMODE, int HACT, int MODE>
__global__ __launch_bounds__(NW * 64) void inr_mlp_kernel(const NetDesc nd, const LossDesc ld, const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TL = NW * 32;                      // coordinates per tile
  constexpr bool G2D = HACT == ACT_GABOR2D;        // WIRE2D: a second Linear (scale_orth) per layer
  constexpr bool PAIR = HACT == ACT_GABOR || G2D;  // complex layers as interleaved (Re, Im) rows
  // stashed tensors per hidden layer: h, act' (SIREN/FFN) | h, dA, dB (WIRE) | h, dA, dB, dA2, dB2, orth, u^2+v^2 (2D)
  constexpr int NS = G2D ? 7 : (PAIR ? 3 : 2);
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int wcol = w * 32 + col;
  constexpr int RS = NB * 32 * INR_LDS_LD;  // floats per wave image
  float* R = lds + w * RS;
  float* encB_lds = lds + NW * RS;  // [E][3] encoder matrix (gauss mode)
  // a.ll_lds: the last layer's fragments of output rows 0..3, [NB*4 groups][2 halves][4 rows] float4 + a float4 of zeros
  float* ll_lds = encB_lds + (INMODE == IN_GAUSS ? ((3 * nd.E + 3) & ~3) : 0);
  // a.dz_lds (fused step, real activations): dZ_last gets its own 8-row image per wave behind the fragments, so that the
  // waves' images still hold h_{D-2} (fwd_last_valu writes it back over z) when the last layer's dW is formed
  float* dz_all = ll_lds + (NB * 4 * 8 + 1) * 4;  // [NW][8][INR_LDS_LD]
  constexpr bool ZLDS_OK = MODE == MODE_FUSED && !PAIR && (TL == 128 || TL == 64);
  const bool zlds = ZLDS_OK && a.dz_lds != 0;
  float* dzw = dz_all + w * 8 * INR_LDS_LD;
  const int D = nd.D;
  if (a.ll_lds && MODE != MODE_BWD) {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.packed + nd.L[D - 1].pf_off);
    for (int c = tid; c <= NB * 4 * 8; c += NW * 64) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < NB * 4 * 8) v = src[(c >> 3) * 64 + ((c >> 2) & 1) * 32 + (c & 3)];
      reinterpret_cast<f32x4*>(ll_lds)[c] = v;
    }
  }
  if (INMODE == IN_GAUSS) {
    for (int i = tid; i < 3 * nd.E; i += NW * 64) encB_lds[i] = a.encB[i];
  }
  if (INMODE == IN_GAUSS || (a.ll_lds && MODE != MODE_BWD)) __syncthreads();
  constexpr int HSZ = NB * 32 * TL;  // floats per stashed tensor
  float* slab = (MODE != MODE_FWD) ? a.slabs + (size_t)blockIdx.x * nd.slab_floats : nullptr;
  float loss_acc = 0.f;
  bool first = a.accumulate == 0;  // accumulate: a follow-up launch of the same step (inr_api.hip, split launches)
#ifdef INR_DWG_STATIC  // 256-row builds: hidden-width dW ALWAYS comes from the batch-level GEMM (launch_mlp checks)
  constexpr bool dwg = !G2D;
#else
  const bool dwg = !G2D && a.dw_gemm != 0;  // hidden-width dW by the batch-level GEMM (inr_dw_gemm.hip)
#endif
  const LayerDesc& LL = nd.L[D - 1];
  // hidden rows == NB*32 except for WIRE's 181 complex features (362 rows padded to 384); the plan
  // only pairs NB == 12 with that width (inr_api.hip)
  constexpr bool HFULL = true;  // hidden-layer slabs span all NB*32 rows (inr_plan_create)

  for (int tile = a.tile0 + blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long row0 = (long long)tile * TL;
    const long long crow = row0 + wcol;
    const bool valid = crow < a.B;
    const bool saving = G2D || (MODE != MODE_FWD) || (a.save != nullptr);  // WIRE2D: the API insists on a buffer
    float* sv = a.save;
    if (saving) sv += (size_t)(a.save_by_block ? blockIdx.x : tile) * nd.save_floats_per_tile;
    float* sv_last = sv + (size_t)NS * (D - 1) * HSZ;  // [4][TL]: act'(z_last) of output rows 0..3
    float* sv_enc = sv_last + 4 * TL;                   // [Kblk0*32][TL] encoder features (gauss mode)
    float* sv_g = sv_last + 4 * TL;                     // WIRE2D (never gauss): copy of a layer's output gradient

    // what the loss section needs from memory, requested now: fetched where it is used, the sampling mask, the target row
    // and the four last-layer biases were up to seven serialized round trips between the last GEMM and the loss
    float gt_pre[4] = {0.f, 0.f, 0.f, 0.f}, lb_pre[4] = {0.f, 0.f, 0.f, 0.f};
    bool sampled_pre = false;
    if (MODE != MODE_BWD) {
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
    // ================================ forward =================================
    INR_STAMP(0);
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
          if (saving)
            fwd_layer0_gauss<NB, TL, true>(acc, a.packed + L0.pf_off, encB_lds, nd.E, two_pi * x0, two_pi * x1,
                                           two_pi * x2, sv_enc, wcol, lane);
          else
            fwd_layer0_gauss<NB, TL, false>(acc, a.packed + L0.pf_off, encB_lds, nd.E, two_pi * x0, two_pi * x1,
                                            two_pi * x2, nullptr, wcol, lane);
        } else {
          fwd_layer0_x<NB>(acc, a.packed + L0.pf_off, a.x + (size_t)(valid ? crow : 0) * L0.K, valid, L0.K, L0.Kpad8,
                           lane);
        }
        acc_to_lds<NB, true>(acc, R, a.packed + L0.pbias_off, lane);
        if (G2D) {  // orth_0 = V_0 x + c_0 -> stash slots 5, 6 of layer 0
          const LayerDesc& O0 = nd.L[nd.orth0];
#pragma unroll
          for (int m = 0; m < NB; ++m) acc[m] = zero16();
          fwd_layer0_x<NB>(acc, a.packed + O0.pf_off, a.x + (size_t)(valid ? crow : 0) * O0.K, valid, O0.K, O0.Kpad8,
                           lane);
          orth_epilogue<NB, TL>(acc, a.packed + O0.pbias_off, sv + (size_t)5 * HSZ, wcol, lane);
        }
      }
      INR_STAMP(1);
      for (int l = 1; l < D - 1; ++l) {
        const LayerDesc& Ll = nd.L[l];
        const ActParams ap{nd.L[l - 1].omega, nd.L[l - 1].s0};
        f32x16 acc[NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[m] = zero16();
        float* sh = sv + (size_t)(NS * (l - 1)) * HSZ;
        if (saving)
          fwd_layer<NB, NB, TL, HACT, true>(acc, R, a.packed + Ll.pf_off, ap, sh, wcol, lane);
        else
          fwd_layer<NB, NB, TL, HACT, false>(acc, R, a.packed + Ll.pf_off, ap, nullptr, wcol, lane);
        acc_to_lds<NB, true>(acc, R, a.packed + Ll.pbias_off, lane);
        if (G2D) {  // orth_l = V_l h_{l-1} + c_l with h_{l-1} back from the stash -> slots 5, 6 of layer l
          const LayerDesc& Ol = nd.L[nd.orth0 + l];
#pragma unroll
          for (int m = 0; m < NB; ++m) acc[m] = zero16();
          gemm_stash_nat<NB, TL>(acc, a.packed + Ol.pf_off, sh, wcol, lane);
          orth_epilogue<NB, TL>(acc, a.packed + Ol.pbias_off, sv + (size_t)(NS * l + 5) * HSZ, wcol, lane);
        }
        INR_STAMP(1 + l);
      }
      // last layer: out_f <= 4 rows -> registers 0..3 of the lane-half-0 lanes of one row block
      f32x16 accL[1];
      accL[0] = zero16();
      {
        const ActParams ap{nd.L[D - 2].omega, nd.L[D - 2].s0};
        float* sh = sv + (size_t)(NS * (D - 2)) * HSZ;
        if (!PAIR && a.ll_lds) {  // (ll_lds: <= 4 output rows, their fragments in LDS)
          float o4[4];
          constexpr int HA = PAIR ? ACT_ID : HACT;
          if (ZLDS_OK && zlds)
            fwd_last_valu<NB, TL, HA, 2, true>(o4, R, ll_lds, LL.M, ap, sh, wcol, lane);
          else if (saving)
            fwd_last_valu<NB, TL, HA, 1, false>(o4, R, ll_lds, LL.M, ap, sh, wcol, lane);
          else
            fwd_last_valu<NB, TL, HA, 0, false>(o4, R, ll_lds, LL.M, ap, nullptr, wcol, lane);
#pragma unroll
          for (int o = 0; o < 4; ++o) accL[0][o] = o4[o];
        } else if (a.ll_lds) {
          if (saving)
            fwd_layer<NB, 1, TL, HACT, true, 1, true>(accL, R, ll_lds, ap, sh, wcol, lane);
          else
            fwd_layer<NB, 1, TL, HACT, false, 1, true>(accL, R, ll_lds, ap, nullptr, wcol, lane);
        } else if (saving) {
          fwd_layer<NB, 1, TL, HACT, true>(accL, R, a.packed + LL.pf_off, ap, sh, wcol, lane);
        } else {
          fwd_layer<NB, 1, TL, HACT, false>(accL, R, a.packed + LL.pf_off, ap, nullptr, wcol, lane);
        }
      }
      float zl[4], y[4], dy[4], g[4];
      const bool ctanh = nd.last_act == ACT_CTANH;
      const int nrows_in = ctanh ? 2 * nd.out_f : nd.out_f;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        zl[o] = accL[0][o];
        if (o < nrows_in) zl[o] += lb_pre[o];
        g[o] = 0.f;
      }
      const int nrows = last_layer_act(nd.last_act, nd.out_f, nd.w0, zl, y, dy);  // dy[r]: per image row
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (half == 0 && valid && o < nd.out_f && a.out != nullptr) a.out[crow * nd.out_f + o] = y[o];
      if (MODE == MODE_FWD) {
        if (saving && half == 0) {
#pragma unroll
          for (int o = 0; o < 4; ++o) sv_last[o * TL + wcol] = dy[o];
        }
      } else {
        // fused: pointwise loss of this row (both outputs of a row sit in one half-0 lane)
        if (sampled_pre) loss_acc += loss_row(ld, nd.out_f, y, gt_pre, g);
        // dZ_last = dY * act'(z_last) -> image rows 0..3 (half 0); rows 4..31 are zero
        if (ZLDS_OK && zlds) {  // ... of the wave's dZ image: 8 rows, all that the dX GEMM of a <= 8-row layer reads
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = 0.f;
            if (half == 0 && r < nrows) v = g[ctanh ? r >> 1 : r] * dy[r];
            dzw[swz(r + 4 * half, col)] = v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = 0.f;
            if (r < 4 && half == 0 && r < nrows) v = g[ctanh ? (r & 3) >> 1 : (r & 3)] * dy[r & 3];
            R[swz(acc_row(r, half), col)] = v;
          }
        }
      }
    }

    // ================================ backward ================================
    INR_STAMP(10);
    if (MODE != MODE_FWD) {
      if (MODE == MODE_BWD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.f;
          const bool ct = nd.last_act == ACT_CTANH;
          if (r < 4 && half == 0 && r < (ct ? 2 * nd.out_f : nd.out_f) && valid)
            v = a.dout[crow * nd.out_f + (ct ? (r & 3) >> 1 : (r & 3))] * sv_last[(r & 3) * TL + wcol];
          R[swz(acc_row(r, half), col)] = v;
        }
      }
      __syncthreads();  // every wave's dZ_last is in LDS
      INR_STAMP(11);
      // ---- last layer: dW, db from (dZ_last, h_{D-2}); dH_{D-2} = W_last^T dZ_last
      {
        if (ZLDS_OK && zlds) {
          dw_rows4_valu<TL == 128 ? 128 : 64, NW, ACT_ID>(lds, RS, nullptr, NB * 32, LL.M, LL.K, slab + LL.gw_off,
                                                          slab + LL.gb_off, first, w, lane, dz_all, 8 * INR_LDS_LD, 0.f);
        } else if (LL.M <= 4 && (TL == 128 || TL == 64)) {
          dw_rows4_valu<TL == 128 ? 128 : 64, NW>(lds, RS, sv + (size_t)(NS * (D - 2)) * HSZ, NB * 32, LL.M, LL.K,
                                                  slab + LL.gw_off, slab + LL.gb_off, first, w, lane);
        } else {
          BSrcStash<TL> bs{sv + (size_t)(NS * (D - 2)) * HSZ};
          for (int n = w; n < LL.Kblk; n += NW)
            dw_pass<1, TL, false, BSrcStash<TL>>(lds, RS, bs, n, slab + LL.gw_off, slab + LL.gb_off, LL.M, LL.K, first,
                                                 n == 0, lane);
        }
      }
      INR_STAMP(12);
      f32x16 gacc[NB];
#pragma unroll
      for (int m = 0; m < NB; ++m) gacc[m] = zero16();
      bwd_dx<NB, TL, PAIR, false>(gacc, (ZLDS_OK && zlds) ? dzw : R, a.packed + LL.pb_off, LL.Mpad8, nullptr, wcol, lane);
      __syncthreads();  // all dW reads of the images are done
      if (D == 2)
        acc_times_d_to_lds<NB, TL, PAIR>(gacc, R, sv + (size_t)1 * HSZ, sv + (size_t)2 * HSZ, wcol, lane);  // dZ_0
      else
        acc_to_lds<NB, false>(gacc, R, nullptr, lane);  // R <- dH_{D-2}
      INR_STAMP(13);

      for (int l = D - 2; l >= 1; --l) {
        const LayerDesc& Ll = nd.L[l];
#pragma unroll
        for (int m = 0; m < NB; ++m) gacc[m] = zero16();
        if (G2D) image_copy<NB, TL, true>(R, sv_g, wcol, lane);  // dH_l is needed again for the orth Linear
        // dZ_l = dH_l * act'(z_l) (in place), dH_{l-1} = W_l^T dZ_l
        bwd_dx<NB, TL, PAIR, true>(gacc, R, a.packed + Ll.pb_off, Ll.Mpad8, sv + (size_t)(NS * l + 1) * HSZ, wcol,
                                   lane);
        INR_STAMP(14 + 4 * l);
        if (dwg) {  // dZ_l (this wave's columns) over the act' slot it was formed from: operand of the batch GEMM
          image_copy<NB, TL, true>(R, sv + (size_t)(NS * l + 1) * HSZ, wcol, lane);
        } else {
          __syncthreads();
          INR_STAMP(15 + 4 * l);
          {
            BSrcStash<TL> bs{sv + (size_t)(NS * (l - 1)) * HSZ};
            for (int n = w; n < Ll.Kblk; n += NW)
              dw_pass<NB, TL, HFULL, BSrcStash<TL>>(lds, RS, bs, n, slab + Ll.gw_off, slab + Ll.gb_off, Ll.M, Ll.K,
                                                    first, n == 0, lane);
          }
          INR_STAMP(16 + 4 * l);
          __syncthreads();
        }
        if (G2D) {  // second Linear of the layer: dZ_orth = J_orth dH_l, dH_{l-1} += V_l^T dZ_orth, dV_l
          const LayerDesc& Ol = nd.L[nd.orth0 + l];
          image_copy<NB, TL, false>(R, sv_g, wcol, lane);
          bwd_dx<NB, TL, PAIR, true>(gacc, R, a.packed + Ol.pb_off, Ol.Mpad8, sv + (size_t)(NS * l + 3) * HSZ, wcol,
                                     lane);
          __syncthreads();
          {
            BSrcStash<TL> bs{sv + (size_t)(NS * (l - 1)) * HSZ};
            for (int n = w; n < Ol.Kblk; n += NW)
              dw_pass<NB, TL, HFULL, BSrcStash<TL>>(lds, RS, bs, n, slab + Ol.gw_off, slab + Ol.gb_off, Ol.M, Ol.K,
                                                    first, n == 0, lane);
          }
          __syncthreads();
        }
        if (l == 1)
          acc_times_d_to_lds<NB, TL, PAIR>(gacc, R, sv + (size_t)1 * HSZ, sv + (size_t)2 * HSZ, wcol, lane);  // dZ_0
        else
          acc_to_lds<NB, false>(gacc, R, nullptr, lane);  // R <- dH_{l-1}
        INR_STAMP(17 + 4 * l);
      }
      // ---- first layer: dW_0 against the stashed encoder features / the input matrix
      {
        const LayerDesc& L0 = nd.L[0];
        if (INMODE == IN_GAUSS && dwg) {
          image_copy<NB, TL, true>(R, sv + (size_t)1 * HSZ, wcol, lane);  // dZ_0 -> stash
        } else if (INMODE == IN_GAUSS) {
          __syncthreads();  // every wave's dZ_0 is in LDS
          INR_STAMP(40);
          BSrcStash<TL> bs{sv_enc};
          for (int n = w; n < L0.Kblk; n += NW)
            dw_pass<NB, TL, HFULL, BSrcStash<TL>>(lds, RS, bs, n, slab + L0.gw_off, slab + L0.gb_off, L0.M, L0.K, first,
                                                 n == 0, lane);
        } else {
          __syncthreads();  // every wave's dZ_0 is in LDS
          BSrcX bs{a.x, row0, a.B, L0.K};
          for (int n = w; n < L0.Kblk; n += NW)
            dw_pass<NB, TL, HFULL, BSrcX>(lds, RS, bs, n, slab + L0.gw_off, slab + L0.gb_off, L0.M, L0.K, first,
                                          n == 0, lane);
        }
        INR_STAMP(41);
        __syncthreads();  // images are overwritten by the next tile's forward / dZ_last
        if (G2D) {  // dV_0 from dZ_0,orth = J_orth,0 dH_0 (the accumulators still hold dH_0)
          const LayerDesc& O0 = nd.L[nd.orth0];
          acc_times_d_to_lds<NB, TL, PAIR>(gacc, R, sv + (size_t)3 * HSZ, sv + (size_t)4 * HSZ, wcol, lane);
          __syncthreads();
          BSrcX bs{a.x, row0, a.B, O0.K};
          for (int n = w; n < O0.Kblk; n += NW)
            dw_pass<NB, TL, HFULL, BSrcX>(lds, RS, bs, n, slab + O0.gw_off, slab + O0.gb_off, O0.M, O0.K, first, n == 0,
                                          lane);
          __syncthreads();
        }
        INR_STAMP(42);
      }
      first = false;
    }
  }

  if (MODE == MODE_FUSED) {
    // block loss partial -> slab loss word (fixed order: wave shuffle tree, then waves in order)
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

// hipFuncSetAttribute + launch
template <int NB, int NW, int INMODE, int HACT, int MODE>
inline hipError_t launch_mlp(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a_in, int grid, hipStream_t st) {
  MlpArgs a = a_in;
  size_t lds_bytes = ((size_t)NW * NB * 32 * INR_LDS_LD + 3 * (size_t)nd.E) * sizeof(float);
  auto k = inr_mlp_kernel<NB, NW, INMODE, HACT, MODE>;
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  // the last layer's fragments in LDS when they fit (4 KB + 16 B behind the 16-byte aligned encoder matrix) and the
  // layer really has <= 4 output rows
  const size_t with_ll = ((size_t)NW * NB * 32 * INR_LDS_LD + (INMODE == IN_GAUSS ? ((3 * (size_t)nd.E + 3) & ~(size_t)3) : 0) +
                          (size_t)(NB * 4 * 8 + 1) * 4) * sizeof(float);
  a.ll_lds = (MODE != MODE_BWD && nd.L[nd.D - 1].M <= 4 && with_ll <= 160 * 1024) ? 1 : 0;
  if (a.ll_lds) lds_bytes = with_ll;
  // ... and, in the fused step, an 8-row dZ_last image per wave behind them (see the kernel)
  const size_t with_dz = with_ll + (size_t)NW * 8 * INR_LDS_LD * sizeof(float);
  constexpr bool real_act = HACT != ACT_GABOR && HACT != ACT_GABOR2D;
  a.dz_lds = (a.ll_lds && MODE == MODE_FUSED && real_act && nd.L[nd.D - 1].Mpad8 <= 8 && with_dz <= 160 * 1024) ? 1 : 0;
  if (a.dz_lds) lds_bytes = with_dz;
#ifdef INR_DWG_STATIC  // the kernel has no in-kernel dW passes for layers the GEMM can take: the caller must run it
  if (MODE != MODE_FWD && HACT != ACT_GABOR2D && !a.dw_gemm && (nd.D > 2 || INMODE == IN_GAUSS))
    return hipErrorInvalidValue;
#endif
  {
    hipError_t e = allow_full_lds<inr_mlp_kernel<NB, NW, INMODE, HACT, MODE>>();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds_bytes, st, nd, ld, a);
  return hipGetLastError();
}

}  // namespace inr
