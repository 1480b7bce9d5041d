// inr_mlp_rs_impl.h -- the fused SIREN / FFN training step with the OUTPUT ROWS split over the four waves of a
// workgroup ("row split"), for hidden widths 129..256 behind the fused gauss encoder (fp32-exact path, gfx950).
//
// Why: inr_mlp_kernel gives every wave 32 coordinates and all 256 rows.  The unit of work is then 32 coordinates per
// SIMD, and 25 000 rows are 782 such units on 1 024 SIMDs: the busiest SIMD carries 32 coordinates where 24.4 would do
// (DESIGN 4.1, 7.1).  Here a workgroup (one per CU) owns a tile of up to 8 column blocks of 16 coordinates; wave w owns
// output rows [64 w, 64 w + 64) of every layer for ALL coordinates of the tile, so the unit is 16 coordinates per CU:
// 25 000 rows -> 7 column blocks on the busiest CU (112 coordinates) instead of 128.
//
//   * MFMA: v_mfma_f32_16x16x4_f32 (same FLOP rate as 32x32x2, exact fp32 FMA chain in k order).  A = weights
//     [16 rows x 4 k] from a packed image in L2 (one 16-byte load per lane and k-step: the four row blocks of the wave;
//     every weight is read ONCE per CU and tile instead of four times), B = activations [4 k x 16 coordinates] from the
//     workgroup's shared LDS image, C = [4 row blocks][NCB column blocks] x 4 registers.
//   * LDS image [256 features][16 jj][8 c] (pitch 132 floats): a lane reads its B operands of all column blocks with
//     two conflict-free ds_read_b128 per k-step, and writes its activated outputs with ds_write_b128.
//   * activation is EAGER (row owners activate their 64 x 16 NCB outputs once; the lazy form of inr_mlp_kernel would
//     repeat it in all four waves): between two GEMMs the waves meet at two barriers.
//   * stash traffic rides the GEMM loops, never a burst: h_{l-1} / dZ_l / encoder features are stored from the B
//     registers of the GEMM that consumes them (each wave every fourth k-step), act' is kept in registers by its owner
//     and stored / reloaded one 8-byte access per k-step.  (A burst of stores sits in front of the next GEMM's weight
//     loads in the in-order vmcnt queue.)
//   * the <= 4-row last layer, the loss and its adjoint never touch the matrix pipe: partial sums over each lane's 16
//     rows, summed across lanes and waves in a fixed order.
//   * register budget: gfx950 code addresses 256 arch VGPRs + 256 AGPRs.  The accumulators (16 NCB) live in AGPRs;
//     act' (16 NCB) is the only large VGPR array -- activations go to the LDS image row by row, never through a
//     second register array.
//   * the stash has exactly the layout of inr_mlp_kernel ([tensor][feature][128 coordinates] per 128-coordinate slot,
//     coordinate g in slot g >> 7), so the batch-level weight-gradient GEMM (inr_dw_gemm.hip) is unchanged.
//
// Arithmetic: models/networks.py:23-35 (encoder), :91-96 (SIREN layer), :48-69 (FFN); loop train.py:158-192.
#pragma once
#include "inr_mlp_impl.h"

namespace inr {

constexpr int RS_PITCH = 132;                     // floats per feature row of the LDS image: 16 jj x 8 c + 4
constexpr int RS_IMG_FLOATS = 256 * RS_PITCH;     // 135 168 B
constexpr int RS_CP = 32;                         // encoder phases per chunk: 64 feature rows (sines | cosines)
constexpr int RS_CHUNK_FLOATS = 2 * RS_CP * RS_PITCH;
constexpr int RS_CHUNK_STEPS = 2 * RS_CP / 4;     // k-steps of 4 per chunk
constexpr int RS_PF = 8;                          // weight fragments are requested this many k-steps ahead
constexpr int RS_LR = 8;                          // act' loads of the backward GEMMs in flight (8-byte accesses)
constexpr int RS_OOB = 0x7ffffff0;                // per-lane offset of a column block that does not exist: loads 0, stores dropped
constexpr int RS_HSZ = 256 * 128;                 // floats per stashed tensor of a slot

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lfloat;
typedef __attribute__((address_space(3))) f32x4 lf32x4;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Local coordinate (0 .. 16 NCB - 1) of column block c, column jj.  Column blocks come in PAIRS whose coordinates
// interleave (pair p: coordinates 32 p + 2 jj + {0, 1}), so that the two values a lane holds for a pair are neighbours in
// the stash and travel as one 8-byte access; an odd NCB ends with a single block of 16 consecutive coordinates.
template <int NCB>
__device__ __forceinline__ int rs_tl(int c, int jj) {
  return ((NCB & 1) && c == NCB - 1) ? 16 * (NCB - 1) + jj : 32 * (c >> 1) + 2 * jj + (c & 1);
}
// does column block c exist in a tile of n blocks?  (n == NCB, or n even: inr_api.hip rs_schedule)
template <int NCB>
__device__ __forceinline__ bool rs_active(int c, int n) {
  return ((NCB & 1) && c == NCB - 1) ? n == NCB : (c | 1) < n;
}

template <int NCB>
struct RsAddr {
  static constexpr int NP = (NCB + 1) / 2;
  __amdgpu_buffer_rsrc_t rs;  // the (at most two) stash slots the tile's coordinates live in
  int cB[NP];                 // byte offset of the lane's coordinate pair p, + row part of the B layout (row 4 s + kq)
  int cC[NP];                 // ... + row part of the C layout (row 16 rb + 4 kq + reg)
};

__device__ __forceinline__ void rs_store2(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float x, float y) {
  u32x2 v = {__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y)};
  __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff, soff, 0);
}
__device__ __forceinline__ void rs_store1(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float x) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rs, voff, soff, 0);
}
__device__ __forceinline__ f32x2 rs_load2(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}
__device__ __forceinline__ float rs_load1(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}

// act' (16 NCB values per lane) lives in AGPRs for its whole life, explicitly: with the accumulators (16 NCB AGPRs) that is
// the whole accumulator file at NCB = 8 and leaves the 256 arch VGPRs to everything else.  Left to the register allocator
// the array sat in VGPRs (the budget then holds by a handful of registers and any edit tips it into scratch, with an
// s_waitcnt vmcnt(0) per k-step) or in AGPRs used directly as vector-memory data -- which stalls the matrix pipe (42 cycles
// per MFMA instead of 35: profiles/r04_rs_agpr_stores.txt).  So: v_accvgpr_write where a value is formed,
// v_accvgpr_read into a VGPR right before it is stored or multiplied; loads land in VGPRs and move over later.
// (NCB = 8 would be all 256 AGPRs: the compiler's own AGPR copies then push act' into scratch.  Keeping four of the 16 rows
// in VGPRs instead -- in_agpr = false -- was tried and lost them to scratch in the backward GEMM all the same: tiles stop
// at 7 column blocks, inr_api.hip kRsMaxNcb.)
__device__ __forceinline__ float rs_dk_put(bool in_agpr, float v) {  // (in_agpr folds once the row loops are unrolled)
  if (!in_agpr) return v;
  float a;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
  return a;
}
__device__ __forceinline__ float rs_dk_get(bool in_agpr, float a) {
  if (!in_agpr) return a;
  float v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}
// is act' of row (rb, reg) of an NCB-wide tile kept in AGPRs?
template <int NCB>
constexpr bool rs_dk_agpr(int rb, int reg) { return true; }

// column block c of a lane's B fragment: b0 = column blocks 0..3, b1 = 4..7
__device__ __forceinline__ float rs_bval(const f32x4& b0, const f32x4& b1, int c) { return c < 4 ? b0[c] : b1[c - 4]; }

// rows 4 s + kq (B layout) of the tile -> stash tensor at byte offset soff (row 4 s included): pair stores
template <int NCB>
__device__ __forceinline__ void rs_store_brows(const RsAddr<NCB>& ad, int soff, const f32x4& b0, const f32x4& b1) {
  constexpr int NP = RsAddr<NCB>::NP;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if ((NCB & 1) && p == NP - 1)
      rs_store1(ad.rs, ad.cB[p], soff, rs_bval(b0, b1, 2 * p));
    else
      rs_store2(ad.rs, ad.cB[p], soff, rs_bval(b0, b1, 2 * p), rs_bval(b0, b1, 2 * p + 1));
  }
}

// values of the lane's row (16 rb + reg) (C layout), column blocks 0..3 | 4..7 -> stash tensor row at byte offset soff
template <int NCB>
__device__ __forceinline__ void rs_store_crow(const RsAddr<NCB>& ad, int soff, const f32x4 (&v)[2]) {
  constexpr int NP = RsAddr<NCB>::NP;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if ((NCB & 1) && p == NP - 1)
      rs_store1(ad.rs, ad.cC[p], soff, v[(2 * p) >> 2][(2 * p) & 3]);
    else
      rs_store2(ad.rs, ad.cC[p], soff, v[(2 * p) >> 2][(2 * p) & 3], v[(2 * p + 1) >> 2][(2 * p + 1) & 3]);
  }
}

#define RS_KIND_FEAT 1  // layer 0: B rows = encoder features -> sv_enc
#define RS_KIND_FWD 2   // hidden forward: B rows = h_{l-1} -> its stash slot; dk (act' of layer l-1) -> its slot
#define RS_KIND_BWD 3   // backward: B rows = dZ_l -> the act' slot of layer l; dk <- act' of layer l-1

// ---------------------------------------------------------------------------------------------
// One GEMM: acc[rb][c] += A[rows of this wave][k] . B[k][coordinates], NSTEPS k-steps of 4, fully unrolled.
//   Ar: ring of RS_PF weight fragments, holding k-steps 0 .. RS_PF-1 on entry and the first RS_PF k-steps of the NEXT
//       GEMM (image at byte offset asoff_next) on exit -- the weight stream does not restart between GEMMs.
//   Bl: the lane's base into the LDS image: region + kq PITCH + 8 jj.
//   soffB: where the B rows go (KIND FEAT: sines; soffB2: cosines), row 0 of the region; soffD: act' tensor, row 64 w.
//   INIT: 0: accumulate into acc; 1: the accumulators start at zero (the first k-step's MFMAs take a constant C: nothing is
//   zeroed or kept live); 2: they start at cinit[rb] (the layer's bias of the lane's rows: z = b + W h in one chain).
// ---------------------------------------------------------------------------------------------
template <int NCB, int NSTEPS, int KIND, int INIT>
__device__ __forceinline__ void rs_gemm(f32x4 (&acc)[4][NCB], f32x4 (&Ar)[RS_PF], const __amdgpu_buffer_rsrc_t ars,
                                        const int avoff, const int asoff, const int asoff_next, const lfloat* Bl,
                                        const RsAddr<NCB>& ad, const int w, const int soffB, const int soffB2,
                                        f32x2 (&dk)[4][4][(NCB + 1) / 2], const int soffD, const f32x4 (&cinit)[4]) {
  constexpr int NP = (NCB + 1) / 2;
  constexpr int NQ = (NCB + 3) / 4;
  static_assert(NSTEPS % RS_PF == 0, "the fragment ring must be in phase at the start of every GEMM");
  f32x2 lr[RS_LR];  // (KIND BWD) act' pairs on their way from memory to the AGPRs
  f32x4 b0, b1 = {0.f, 0.f, 0.f, 0.f};
  b0 = *(const lf32x4*)(Bl);
  if (NQ > 1) b1 = *(const lf32x4*)(Bl + 4);
  const lfloat* Blw = Bl + w * 4 * RS_PITCH;  // rows of k-step s + w
  f32x4 st0, st1 = {0.f, 0.f, 0.f, 0.f};
  st0 = *(const lf32x4*)(Blw);
  if (NQ > 1) st1 = *(const lf32x4*)(Blw + 4);
#pragma unroll
  for (int s = 0; s < NSTEPS; ++s) {
    const f32x4 afr = Ar[s % RS_PF];
    {  // the fragment RS_PF k-steps ahead (of this GEMM, then of the next one)
      const int so = s + RS_PF < NSTEPS ? asoff + (s + RS_PF) * 4096 : asoff_next + (s + RS_PF - NSTEPS) * 4096;
      Ar[s % RS_PF] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff, so, 0));
    }
    f32x4 n0 = b0, n1 = b1;
    if (s + 1 < NSTEPS) {
      n0 = *(const lf32x4*)(Bl + (s + 1) * 4 * RS_PITCH);
      if (NQ > 1) n1 = *(const lf32x4*)(Bl + (s + 1) * 4 * RS_PITCH + 4);
    }
    // stash traffic of this k-step.  B rows: at every fourth k-step each wave stores the rows of k-step s + w, which it reads
    // from the image for that purpose one k-step earlier (two more LDS reads per four k-steps) -- storing from the operand
    // registers needed a branch on the wave number in every k-step, and a GEMM that is one basic block can have its loads
    // and stores placed between the MFMAs.
    if ((s & 3) == 0) {
      const int rowb = KIND == RS_KIND_FEAT ? (s < NSTEPS / 2 ? soffB + 4 * s * 512 : soffB2 + 4 * (s - NSTEPS / 2) * 512)
                                            : soffB + 4 * s * 512;
      rs_store_brows<NCB>(ad, rowb + w * 2048, st0, st1);
    }
    if ((s & 3) == 3 && s + 1 < NSTEPS) {  // rows of k-step (s + 1) + w, stored at the next k-step
      st0 = *(const lf32x4*)(Blw + (s + 1) * 4 * RS_PITCH);
      if (NQ > 1) st1 = *(const lf32x4*)(Blw + (s + 1) * 4 * RS_PITCH + 4);
    }
    if (KIND == RS_KIND_FWD || KIND == RS_KIND_BWD) {
      // accesses j of 16 NP spread over the k-steps; j -> (rb, reg, p)
      constexpr int TOT = 16 * NP;
      const int j0 = s * TOT / NSTEPS, j1 = (s + 1) * TOT / NSTEPS;
#pragma unroll
      for (int j = j0; j < j1; ++j) {
        const int rr = j / NP, p = j % NP, rb = rr >> 2, reg = rr & 3;
        const int so = soffD + (16 * rb + reg) * 512;
        const bool single = (NCB & 1) && p == NP - 1;
        if (KIND == RS_KIND_FWD) {
          if (single)
            rs_store1(ad.rs, ad.cC[p], so, rs_dk_get(rs_dk_agpr<NCB>(rb, reg), dk[rb][reg][p][0]));
          else
            rs_store2(ad.rs, ad.cC[p], so, rs_dk_get(rs_dk_agpr<NCB>(rb, reg), dk[rb][reg][p][0]),
                      rs_dk_get(rs_dk_agpr<NCB>(rb, reg), dk[rb][reg][p][1]));
        } else {
          // the load of RS_LR accesses ago has arrived (it is RS_LR / NP .. k-steps old): to its AGPRs
          if (j >= RS_LR) {
            const int jo = j - RS_LR, ro = jo / NP, po = jo % NP;
            dk[ro >> 2][ro & 3][po][0] = rs_dk_put(rs_dk_agpr<NCB>(ro >> 2, ro & 3), lr[jo % RS_LR][0]);
            if (!((NCB & 1) && po == NP - 1))
              dk[ro >> 2][ro & 3][po][1] = rs_dk_put(rs_dk_agpr<NCB>(ro >> 2, ro & 3), lr[jo % RS_LR][1]);
          }
          if (single) {
            lr[j % RS_LR][0] = rs_load1(ad.rs, ad.cC[p], so);
            lr[j % RS_LR][1] = 0.f;
          } else {
            lr[j % RS_LR] = rs_load2(ad.rs, ad.cC[p], so);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const float bv = rs_bval(b0, b1, c);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        acc[rb][c] = mfma16(afr[rb], bv,
                            (INIT == 1 && s == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : ((INIT == 2 && s == 0) ? cinit[rb] : acc[rb][c]));
    }
    // one memory / LDS / vector instruction behind each of the first MFMAs: they issue while the matrix pipe works
    // (bunched in front of the MFMAs they cost the pipe ~100 idle cycles per k-step)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x1b6, 1, 0);  // one of: VALU, SALU, vector memory, LDS
    }
    __builtin_amdgcn_sched_barrier(0);
    b0 = n0;
    b1 = n1;
  }
  if (KIND == RS_KIND_BWD) {  // the last RS_LR loads
    constexpr int TOT = 16 * NP;
#pragma unroll
    for (int jo = TOT - RS_LR; jo < TOT; ++jo) {
      const int ro = jo / NP, po = jo % NP;
      dk[ro >> 2][ro & 3][po][0] = rs_dk_put(rs_dk_agpr<NCB>(ro >> 2, ro & 3), lr[jo % RS_LR][0]);
      if (!((NCB & 1) && po == NP - 1))
        dk[ro >> 2][ro & 3][po][1] = rs_dk_put(rs_dk_agpr<NCB>(ro >> 2, ro & 3), lr[jo % RS_LR][1]);
    }
  }
}

// 64 encoder feature rows (32 phases: sines | cosines) of chunk ch for the tile's coordinates -> LDS chunk buffer.
// Thread tid: coordinate column jj = tid & 15 (all NCB blocks), phases (tid >> 4) and (tid >> 4) + 16.
template <int NCB>
__device__ __forceinline__ void rs_gen(lfloat* buf, const lfloat* encB_lds, int ch, const float (&xs)[NCB][3], int tid) {
  const int jj = tid & 15, p0 = tid >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = p0 + 16 * i, s = ch * RS_CP + p;
    const float e0 = encB_lds[3 * s + 0], e1 = encB_lds[3 * s + 1], e2 = encB_lds[3 * s + 2];
    f32x4 sn[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, cs[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const float ph = fmaf(xs[c][2], e2, fmaf(xs[c][1], e1, xs[c][0] * e0));  // (2 pi x) @ B^T (networks.py:31)
      float a, b;
      sincos_cw(ph, a, b);
      sn[c >> 2][c & 3] = a;
      cs[c >> 2][c & 3] = b;
    }
    lfloat* q = buf + p * RS_PITCH + jj * 8;
    *(lf32x4*)(q) = sn[0];
    *(lf32x4*)(q + RS_CP * RS_PITCH) = cs[0];
    if (NCB > 4) {
      *(lf32x4*)(q + 4) = sn[1];
      *(lf32x4*)(q + RS_CP * RS_PITCH + 4) = cs[1];
    }
  }
}

// What the last hidden layer parks in LDS across the loss section instead of (h, act'): the sine's reduced argument in
// revolutions (SIREN: h = sin, act' = w0 cos are one transcendental each from it) or z itself (FFN).
template <int HACT>
__device__ __forceinline__ float rs_act_park(float z, float w0, float& h) {
  if (HACT == ACT_SIN) {
    constexpr float c_hi = 0.15915494309189535f;
    constexpr float c_lo = (float)(0.15915494309189533576888 - (double)c_hi);
    const float t = w0 * z;  // (as act_fwd / sincos_cw: the same roundings)
    const float k = rintf(t * c_hi);
    const float rev = fmaf(t, c_lo, fmaf(t, c_hi, -k));
    h = __builtin_amdgcn_sinf(rev);
    return rev;
  }
  float d;
  act_fwd<HACT>(z, w0, h, d);
  return z;
}
template <int HACT>
__device__ __forceinline__ void rs_act_unpark(float p, float w0, float& h, float& d) {
  if (HACT == ACT_SIN) {
    h = __builtin_amdgcn_sinf(p);
    d = w0 * __builtin_amdgcn_cosf(p);
  } else {
    act_fwd<HACT>(p, w0, h, d);
  }
}

// ---------------------------------------------------------------------------------------------
// Epilogue of the LAST hidden layer + the last layer (MO = 2 or 4 output rows computed) + the pointwise loss + their
// adjoint, all on the vector ALUs.  In: acc = z_{D-2} of the lane's rows (C layout); the image is free.
// The parked form of z_{D-2} (rs_act_park) goes to the lane's own rows of the image -- only this lane reads them back,
// after the barriers, and forms h_{D-2} and act' again: no register array lives across the section.
// Out: dZ_{D-2} in the lane's rows of the image; dW_last / db_last added to the workgroup's slab; returns this thread's
// loss contribution.  Two workgroup barriers inside; the caller syncs before the image is read.
// ---------------------------------------------------------------------------------------------
template <int NCB, int HACT>
struct RsLast {
  lfloat *wl_lds, *dzl, *red, *part, *Cl;
  float* slab;
  bool lvalid, sampled, first;
  long long lrow;
  int w, kq, jj, tid, lane;

  template <int MO>
  __device__ __forceinline__ float run(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, const f32x4 (&acc)[4][NCB],
                                       const float (&gt_pre)[4], const float (&lb_pre)[4], float w0) {
    constexpr int NQ = (NCB + 3) / 4;
    constexpr int NW = 4;
    (void)NW;  // (INR_STAMP)
    const LayerDesc& LL = nd.L[nd.D - 1];
    float loss = 0.f;
    // h = act(z), act'; partial outputs over the lane's 16 rows: W_last[o][64 w + 16 rb + 4 kq + (0..3)] . h
    f32x4 yp[MO][2];
#pragma unroll
    for (int o = 0; o < MO; ++o) yp[o][0] = yp[o][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      f32x4 wv[MO];
#pragma unroll
      for (int o = 0; o < MO; ++o) wv[o] = *(const lf32x4*)(wl_lds + o * 256 + 64 * w + 16 * rb + 4 * kq);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        f32x4 hq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, pq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          float h;
          pq[c >> 2][c & 3] = rs_act_park<HACT>(acc[rb][c][reg], w0, h);
          hq[c >> 2][c & 3] = h;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          *(lf32x4*)(Cl + (16 * rb + reg) * RS_PITCH + 4 * q) = pq[q];
#pragma unroll
          for (int o = 0; o < MO; ++o) yp[o][q] += wv[o][reg] * hq[q];
        }
        __builtin_amdgcn_sched_barrier(0);  // one row at a time: hoisting the rows' reads over each other ran into scratch
      }
    }
    INR_STAMP(50);
    // the four lane quarters (kq) hold different rows: add them (fixed tree), quarter 0 writes part[w][o][jj][c]
#pragma unroll
    for (int o = 0; o < MO; ++o)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // (v_permlane16_swap / v_permlane32_swap would do this without LDS round trips; as clang builtins they returned
          // wrong sums inside this kernel though not alone -- tools/probes/permlane_swap_probe.hip -- and are not used)
          float v = yp[o][q][e];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          yp[o][q][e] = v;
        }
    if (kq == 0) {
#pragma unroll
      for (int o = 0; o < MO; ++o) {
        lfloat* pp = part + (w * MO + o) * 128 + jj * 8;
        *(lf32x4*)pp = yp[o][0];
        *(lf32x4*)(pp + 4) = yp[o][1];
      }
    }
    INR_STAMP(51);
    __syncthreads();
    INR_STAMP(52);
    if (tid < 128) {  // one coordinate per thread: the four waves' partials in wave order, activation, loss, dZ_last
      float zl[4], y[4], dy[4], g[4] = {0.f, 0.f, 0.f, 0.f}, dz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float s = 0.f;
        if (o < MO)
          s = ((part[o * 128 + tid] + part[(MO + o) * 128 + tid]) + part[(2 * MO + o) * 128 + tid]) +
              part[(3 * MO + o) * 128 + tid];
        zl[o] = o < nd.out_f ? s + lb_pre[o] : 0.f;
      }
      const int nrows = nd.out_f;  // (real last activations only: INR_ACT_CTANH is WIRE2D's)
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        y[o] = dy[o] = 0.f;
        if (o < nd.out_f) act_fwd_rt(nd.last_act, zl[o], nd.w0, y[o], dy[o]);
      }
      if (lvalid) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
          if (o < nd.out_f && a.out != nullptr) a.out[lrow * nd.out_f + o] = y[o];
        if (sampled) loss = loss_row(ld, nd.out_f, y, gt_pre, g);
#pragma unroll
        for (int o = 0; o < 4; ++o)
          if (o < nrows) dz[o] = g[o] * dy[o];
      }
#pragma unroll
      for (int o = 0; o < MO; ++o) dzl[o * 128 + tid] = dz[o];
    }
    INR_STAMP(53);
    __syncthreads();
    INR_STAMP(54);
    // dZ_last of this lane's coordinates
    f32x4 dzv[MO][2];
#pragma unroll
    for (int o = 0; o < MO; ++o)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        dzv[o][q] = q < NQ ? *(const lf32x4*)(dzl + o * 128 + jj * 8 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    // db_last[o] = sum of dZ_last[o] over the tile: this lane's NCB blocks, then the 16 jj (wave 0, quarter 0 stores).
    // All slab entries of this section are stored together at its end: a store followed by anything that waits on the
    // vector-memory counter (a spilled register coming back, say) costs the store's whole round trip.
    const __amdgpu_buffer_rsrc_t rsb = uniform_rsrc(slab + LL.gb_off, LL.M * 4);
    float db_mine, dw_mine[4];
    {
      float sb[MO];
#pragma unroll
      for (int o = 0; o < MO; ++o) {
        sb[o] = 0.f;
#pragma unroll
        for (int c = 0; c < NCB; ++c) sb[o] += dzv[o][c >> 2][c & 3];
      }
      group_sum_n<16, MO>(sb);
      float mine = sb[0];
#pragma unroll
      for (int o = 1; o < MO; ++o) mine = jj == o ? sb[o] : mine;
      db_mine = mine;
    }
    // dW_last: one slab entry per lane and row block -- lane jj of a 16-lane row keeps entry (o, reg) = (jj >> 2, jj & 3)
    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(slab + LL.gw_off, LL.M * LL.K * 4);
    {
      // the lane's 16 rows, one at a time, the next row's parked values requested while this one is worked on
      f32x4 nx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int q = 0; q < NQ; ++q) nx[q] = *(const lf32x4*)(Cl + 4 * q);
      f32x4 wv[MO];
      float pw[MO * 4];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int rb = rr >> 2, reg = rr & 3;
        lfloat* row = Cl + (16 * rb + reg) * RS_PITCH;
        if (reg == 0) {
#pragma unroll
          for (int o = 0; o < MO; ++o) wv[o] = *(const lf32x4*)(wl_lds + o * 256 + 64 * w + 16 * rb + 4 * kq);
        }
        f32x4 hq[2] = {nx[0], nx[1]}, dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 dd[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (rr + 1 < 16) {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            nx[q] = *(const lf32x4*)(Cl + (16 * ((rr + 1) >> 2) + ((rr + 1) & 3)) * RS_PITCH + 4 * q);
        }
#pragma unroll
        for (int c = 0; c < NCB; ++c) {  // h_{D-2} and act' again, from the parked form
          float h, d;
          rs_act_unpark<HACT>(hq[c >> 2][c & 3], w0, h, d);
          hq[c >> 2][c & 3] = h;
          dd[c >> 2][c & 3] = d;
        }
        // dW_last[o][row] = sum over the tile's coordinates of dZ_last[o] h[row]: the lane's NCB blocks here, the 16 jj below
#pragma unroll
        for (int o = 0; o < MO; ++o) {
          float sp = 0.f;
#pragma unroll
          for (int c = 0; c < NCB; ++c) sp = fmaf(dzv[o][c >> 2][c & 3], hq[c >> 2][c & 3], sp);
          pw[o * 4 + reg] = sp;
        }
        // dZ_{D-2} = (W_last^T dZ_last) * act'(z_{D-2}) -> the lane's own row of the image
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          float sd = 0.f;
#pragma unroll
          for (int o = 0; o < MO; ++o) sd = fmaf(wv[o][reg], dzv[o][c >> 2][c & 3], sd);
          dq[c >> 2][c & 3] = sd * dd[c >> 2][c & 3];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) *(lf32x4*)(row + 4 * q) = dq[q];
        if (reg == 3) {
          group_sum_n<16, MO * 4>(pw);  // (every lane of a 16-lane row now holds the row's sums)
          float mine = pw[0];
#pragma unroll
          for (int i = 1; i < MO * 4; ++i) mine = jj == i ? pw[i] : mine;
          dw_mine[rb] = mine;
        }
        __builtin_amdgcn_sched_barrier(0);  // one row at a time: hoisting the rows' reads over each other ran into scratch
      }
    }
    {
      const int so_ = jj >> 2;
      int off[5];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const int rowi = 64 * w + 16 * rb + 4 * kq + (jj & 3);
        off[rb] = (jj < MO * 4 && so_ < LL.M && rowi < LL.K) ? (so_ * LL.K + rowi) * 4 : RS_OOB;
      }
      off[4] = (w == 0 && kq == 0 && jj < MO && jj < LL.M) ? jj * 4 : RS_OOB;
      if (!first) {  // a later tile of this workgroup: add to the slab (all five loads in flight together)
        float old[5];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) old[rb] = rs_load1(rsw, off[rb], 0);
        old[4] = rs_load1(rsb, off[4], 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) dw_mine[rb] += old[rb];
        db_mine += old[4];
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) rs_store1(rsw, off[rb], 0, dw_mine[rb]);
      rs_store1(rsb, off[4], 0, db_mine);
    }
    INR_STAMP(55);
    return loss;
  }
};

// ---------------------------------------------------------------------------------------------
// the kernel: fused forward + pointwise loss + backward of a tile, hidden-width weight gradients left to the batch GEMM
// ---------------------------------------------------------------------------------------------
template <int NCB, int HACT>
__global__ __launch_bounds__(256) void inr_mlp_rs_kernel(const NetDesc nd, const LossDesc ld, const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NW = 4;
  (void)NW;  // (INR_STAMP)
  constexpr int NP = (NCB + 1) / 2;
  constexpr int NQ = (NCB + 3) / 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  INR_RT_STAMP(a.dbg, a.dbg_cap, NW, w, lane, 44);  // (diagnostic builds: kernel entry / exit of each wave, slots 44 / 45)
  const int kq = lane >> 4, jj = lane & 15;
  const int D = nd.D, E = nd.E;
  lfloat* img = (lfloat*)lds;
  lfloat* encB_lds = img + RS_IMG_FLOATS;
  lfloat* wl_lds = encB_lds + ((3 * E + 3) & ~3);  // W_last [4][256], zero padded
  lfloat* dzl = wl_lds + 1024;                     // dZ_last [4][128]
  lfloat* red = dzl + 512;                         // [8] sums of dZ_last of waves 0, 1 | [4] loss partials of the waves
  lfloat* part = red + 32;                         // the waves' partial outputs of the last layer [4][MO][128]
  const LayerDesc& LL = nd.L[D - 1];
  for (int i = tid; i < 3 * E; i += 256) encB_lds[i] = a.encB[i];
  for (int i = tid; i < 1024; i += 256) {
    const int o = i >> 8, k = i & 255;
    wl_lds[i] = (o < LL.M && k < LL.K) ? a.params[LL.w_off + o * LL.K + k] : 0.f;
  }
  float* slab = a.slabs + (size_t)blockIdx.x * nd.slab_floats;
  float loss_acc = 0.f;
  if (((int)blockIdx.x < a.rs_x ? a.rs_hi : a.rs_lo) == 0 && a.accumulate == 0) {
    // a workgroup without column blocks (tiles past rs_x of a schedule whose `lo` is 0) still owns a slab the reduction
    // sums: its last-layer entries are zero
    for (int i = tid; i < LL.M * LL.K; i += 256) slab[LL.gw_off + i] = 0.f;
    if (tid < LL.M) slab[LL.gb_off + tid] = 0.f;
  }
  const long long spt = nd.save_floats_per_tile;
  const __amdgpu_buffer_rsrc_t ars = uniform_rsrc(a.packed, 0x7ffffff0);
  const int avoff = lane * 16 + w * 1024;
  const lfloat* Bl = img + kq * RS_PITCH + jj * 8;            // B layout: row 4 s + kq
  lfloat* Cl = img + (64 * w + 4 * kq) * RS_PITCH + jj * 8;  // C layout: row 64 w + 16 rb + 4 kq + reg
  const int enc_off = (2 * (D - 1) * RS_HSZ + 4 * 128) * 4;   // bytes from a slot's start to sv_enc
  const int nch = E / RS_CP;
  f32x4 Ar[RS_PF];
  {  // the weight stream starts: first k-steps of layer 0
    const int so = nd.L[0].rf_off * 4;
#pragma unroll
    for (int i = 0; i < RS_PF; ++i)
      Ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff, so + i * 4096, 0));
  }

  for (int r = 0; r < a.rs_rounds; ++r) {
    const int t = r * gridDim.x + blockIdx.x;
    const int n = t < a.rs_x ? a.rs_hi : a.rs_lo;
    const int blk0 = t < a.rs_x ? t * a.rs_hi : a.rs_x * a.rs_hi + (t - a.rs_x) * a.rs_lo;
    if (n == 0) continue;  // (workgroup-uniform)
    const bool first = r == 0 && a.accumulate == 0;
    const long long g0 = (long long)blk0 * 16;
    const int slot0 = blk0 >> 3;
    RsAddr<NCB> ad;
    {
      const int nslots = a.n_tiles - slot0 < 2 ? a.n_tiles - slot0 : 2;
      ad.rs = uniform_rsrc(a.save + (size_t)slot0 * spt, (int)(nslots * spt * 4));
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int gl = (blk0 & 7) * 16 + rs_tl<NCB>(2 * p, jj);
        const int off = (int)(((gl >> 7) * spt + (gl & 127)) * 4);
        const bool act = rs_active<NCB>(2 * p, n);
        ad.cB[p] = act ? off + kq * 512 : RS_OOB;
        ad.cC[p] = act ? off + kq * 2048 : RS_OOB;
      }
    }
    // this thread's coordinates as the encoder sees them (column jj of every block), times 2 pi
    float xs[NCB][3];
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const long long g = g0 + rs_tl<NCB>(c, jj);
      const bool ok = rs_active<NCB>(c, n) && g < a.B;
      const float two_pi = 6.283185307179586f;
#pragma unroll
      for (int i = 0; i < 3; ++i) xs[c][i] = ok ? two_pi * a.x[3 * g + i] : 0.f;
    }
    // what the loss lanes (threads 0..127: coordinate (c', jj') = (tid & 7, tid >> 3)) need from memory, requested now
    float gt_pre[4] = {0.f, 0.f, 0.f, 0.f}, lb_pre[4] = {0.f, 0.f, 0.f, 0.f};
    bool sampled_pre = false, lvalid = false;
    long long lrow = 0;
    if (tid < 128) {
      const int cq = tid & 7, jq = tid >> 3;
      if (cq < NCB) {
        lrow = g0 + rs_tl<NCB>(cq, jq);
        lvalid = rs_active<NCB>(cq, n) && lrow < a.B;
      }
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < nd.out_f) lb_pre[o] = a.packed[LL.pbias_off + o];
      if (lvalid) {
        sampled_pre = a.mask == nullptr || a.mask[lrow] != 0;
#pragma unroll
        for (int o = 0; o < 4; ++o)
          if (o < nd.out_f) gt_pre[o] = a.gt[lrow * nd.out_f + o];
      }
    }
    INR_STAMP(0);
    __syncthreads();  // the previous tile's last GEMM has read the image; encB / W_last are staged
    f32x4 acc[4][NCB];
    f32x2 dk[4][4][NP];
    f32x4 bnext[4];  // the bias of the lane's rows (C layout) of the layer whose GEMM comes next: its accumulators start there
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      bnext[rb] = *reinterpret_cast<const f32x4*>(a.packed + nd.L[0].pbias_off + 64 * w + 4 * kq + 16 * rb);
#pragma unroll
      for (int c = 0; c < NCB; ++c) acc[rb][c] = bnext[rb];
    }
    // ================================ layer 0: encoder features by chunks ================================
    rs_gen<NCB>(img, encB_lds, 0, xs, tid);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      if (ch == 1) INR_STAMP(56);
      if (ch + 1 < nch) rs_gen<NCB>(img + ((ch + 1) & 1) * RS_CHUNK_FLOATS, encB_lds, ch + 1, xs, tid);
      if (ch == 1) INR_STAMP(57);
      const int ao = (nd.L[0].rf_off + ch * RS_CHUNK_STEPS * 1024) * 4;
      const int an = ch + 1 < nch ? ao + RS_CHUNK_STEPS * 4096 : (D > 2 ? nd.L[1].rf_off : nd.L[0].rf_off) * 4;
      const int sb = enc_off + ch * RS_CP * 512;
      rs_gemm<NCB, RS_CHUNK_STEPS, RS_KIND_FEAT, 0>(acc, Ar, ars, avoff, ao, an, Bl + (ch & 1) * RS_CHUNK_FLOATS, ad, w, sb,
                                                    sb + E * 512, dk, 0, bnext);
      if (ch == 1) INR_STAMP(58);
      __syncthreads();  // chunk buffer (ch & 1) may be refilled; after the last chunk: the image is free
      if (ch == 1) INR_STAMP(59);
    }
    INR_STAMP(1);
    // ================================ forward: epilogue of layer l-1, GEMM of layer l ================================
    // (the loop is rotated -- epilogue first -- so that the last hidden layer's section below sits BEHIND it: inside the
    // loop the compiler must assume another iteration follows and keeps act' alive across everything in the body)
    for (int l = 1; l <= D - 2; ++l) {
      {
        // (requested now, used by this iteration's first MFMAs)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          bnext[rb] = *reinterpret_cast<const f32x4*>(a.packed + nd.L[l].pbias_off + 64 * w + 4 * kq + 16 * rb);
        // z_{l-1} = acc (bias included), h = act(z) -> the owner's rows of the image, act'(z) -> dk (stored during the GEMM)
        const float w0 = nd.L[l - 1].omega;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            f32x4 hq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
              float h, d;
              act_fwd<HACT>(acc[rb][c][reg], w0, h, d);
              hq[c >> 2][c & 3] = h;
              dk[rb][reg][c >> 1][c & 1] = rs_dk_put(rs_dk_agpr<NCB>(rb, reg), d);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) *(lf32x4*)(Cl + (16 * rb + reg) * RS_PITCH + 4 * q) = hq[q];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (l == 1) INR_STAMP(60);
      __syncthreads();  // the image holds h_{l-1}
      INR_STAMP(20 + l - 1);
      const LayerDesc& Ll = nd.L[l];
      const int an = l < D - 2 ? nd.L[l + 1].rf_off * 4 : nd.L[D - 2].rb_off * 4;
      rs_gemm<NCB, 64, RS_KIND_FWD, 2>(acc, Ar, ars, avoff, Ll.rf_off * 4, an, Bl, ad, w, (2 * (l - 1)) * RS_HSZ * 4, 0, dk,
                                       ((2 * (l - 1) + 1) * RS_HSZ + 64 * w * 128) * 4, bnext);
      INR_STAMP(1 + l);
      __syncthreads();  // every wave has read h_{l-1}: the owners may overwrite their rows
      if (l == 1) INR_STAMP(61);
    }
    // ======================== last hidden layer, last layer, loss, adjoint of the last layer ========================
    INR_STAMP(10);
    {
      RsLast<NCB, HACT> ls{wl_lds, dzl, red, part, Cl, slab, lvalid, sampled_pre, first, lrow, w, kq, jj, tid, lane};
      if (LL.M > 2)
        loss_acc += ls.template run<4>(nd, ld, a, acc, gt_pre, lb_pre, nd.L[D - 2].omega);
      else
        loss_acc += ls.template run<2>(nd, ld, a, acc, gt_pre, lb_pre, nd.L[D - 2].omega);
    }
    __syncthreads();  // the image holds dZ_{D-2}
    INR_STAMP(20 + D - 2);
    INR_STAMP(11);
    // ================================ backward ================================
    // the image holds dZ_l; l = D-2 .. 1: dH_{l-1} = W_l^T dZ_l, dZ_{l-1} = dH_{l-1} * act'(z_{l-1})
    for (int l = D - 2; l >= 1; --l) {
      const LayerDesc& Ll = nd.L[l];
      const int an = (l > 1 ? nd.L[l - 1].rb_off : nd.L[0].rf_off) * 4;  // (last GEMM of the tile: the next tile's layer 0)
      rs_gemm<NCB, 64, RS_KIND_BWD, 1>(acc, Ar, ars, avoff, Ll.rb_off * 4, an, Bl, ad, w, (2 * l + 1) * RS_HSZ * 4, 0, dk,
                                       ((2 * (l - 1) + 1) * RS_HSZ + 64 * w * 128) * 4, bnext);
      INR_STAMP(12 + l);
      if (l > 1) {
        __syncthreads();  // every wave has read dZ_l
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < NCB; ++c) dq[c >> 2][c & 3] = acc[rb][c][reg] * rs_dk_get(rs_dk_agpr<NCB>(rb, reg), dk[rb][reg][c >> 1][c & 1]);
#pragma unroll
            for (int q = 0; q < NQ; ++q) *(lf32x4*)(Cl + (16 * rb + reg) * RS_PITCH + 4 * q) = dq[q];
          }
        __syncthreads();
        INR_STAMP(30 + l);
      }
    }
    // dZ_0 of this lane's rows -> the act' slot of layer 0 (operand of the batch GEMM; there is no dX of the input):
    // from the accumulators, or (D == 2: layer 0 is the last hidden layer) from the lane's rows of the image
    {
      const int so0 = (1 * RS_HSZ + 64 * w * 128) * 4;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
          if (D > 2) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) dq[c >> 2][c & 3] = acc[rb][c][reg] * rs_dk_get(rs_dk_agpr<NCB>(rb, reg), dk[rb][reg][c >> 1][c & 1]);
          } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) dq[q] = *(const lf32x4*)(Cl + (16 * rb + reg) * RS_PITCH + 4 * q);
          }
          rs_store_crow<NCB>(ad, so0 + (16 * rb + reg) * 512, dq);
        }
    }
    INR_STAMP(40);
  }

  // workgroup loss partial -> slab loss word (fixed order: wave shuffle tree, then waves in order)
  {
    float v = loss_acc;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (tid == 0) {
      const float tsum = ((red[0] + red[1]) + red[2]) + red[3];
      slab[nd.slab_loss_off] = a.accumulate ? slab[nd.slab_loss_off] + tsum : tsum;
    }
  }
  INR_RT_STAMP(a.dbg, a.dbg_cap, NW, w, lane, 45);
}

inline size_t rs_lds_bytes(const NetDesc& nd) {
  return ((size_t)RS_IMG_FLOATS + ((3 * (size_t)nd.E + 3) & ~(size_t)3) + 1024 + 512 + 32 + 2048) * sizeof(float);
}

template <int NCB, int HACT>
inline hipError_t launch_mlp_rs(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  const size_t lds_bytes = rs_lds_bytes(nd);
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = allow_full_lds<inr_mlp_rs_kernel<NCB, HACT>>();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((inr_mlp_rs_kernel<NCB, HACT>), dim3(grid), dim3(256), lds_bytes, st, nd, ld, a);
  return hipGetLastError();
}

}  // namespace inr
