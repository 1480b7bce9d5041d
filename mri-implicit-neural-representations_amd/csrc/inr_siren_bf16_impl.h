// inr_siren_bf16_impl.h -- SIREN on the bf16 matrix pipe, third design: "weight panels in LDS, activations in registers,
// one row block at a time, two staggered wave groups".  Forward + pointwise loss + backward-to-inputs for tiles of 256
// coordinates (MODE_FUSED), or the two halves of a split step (MODE_FWD: outputs + stash; MODE_BWD: from d(loss)/d(out));
// the weight gradients are left to inr_dw_gemm_bf16.hip, which reads the 8-bit operands this kernel stashes.
//
// What the second design (round 2) measured, and what changed:
//   * its stash -- z_l as fp16, dZ_l as bf16, 4 KB per coordinate, ~690 MB of HBM traffic per step at 65 536 rows --
//     was the bound (backward epilogues at the CU's share of HBM bandwidth).  Now 8 bits per element (inr_w2.h): the
//     PHASE of the sine, round(256 frac(w0 z / 2 pi)) -- all that sin and cos need, good to 2 pi / 256 --, and dZ as bf8
//     (e5m2) under a power-of-two scale that follows the gradient's magnitude from step to step.  2 KB per coordinate,
//     one dword store per four rows;
//   * every GEMM phase ended in an epilogue that all eight waves ran at the same time (vector work and memory traffic
//     with the matrix pipe idle; 18.5 k cycles per layer for 8.2 k of MFMAs).  Now the hidden layers run OUTPUT ROW
//     BLOCK outermost: a panel = the 16 K-steps of one 32-row block, so a block's accumulator (16 registers instead of
//     128) is final after 16 MFMAs and its epilogue -- bias, sine, phase byte, stash, conversion into the next layer's B
//     operand -- runs while the next block multiplies.  Waves 0-3 (role A) do [MFMAs of block m][epilogue of block m],
//     waves 4-7 (role B, the SIMD partners) [epilogue of block m-1][MFMAs of block m] between the same barriers: at
//     any time one wave of a SIMD is on the matrix pipe and the other on the vector ALUs;
//   * weights still travel HBM/L2 -> LDS once per workgroup by LDS-DMA, now as 16 KB panels through an 8-slot ring with
//     seven panels in flight ahead of the one being multiplied.
// Activations never touch LDS: the fp32 accumulator of layer l (features in registers, coordinates on lanes), after
// bias + v_sin_f32 + v_cvt_pk_bf16_f32, IS the B operand of layer l+1 (element j of lane-half h of K-step (m, s) is
// feature 32 m + 16 s + 8 (j >> 2) + 4 h + (j & 3); the fragments are packed in the same k order, inr_w2.h).
#pragma once
#include "inr_mlp_impl.h"
#include "inr_w2.h"
#include <type_traits>

namespace inr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) -- wait counts and register indices that depend on
// the row block are template arguments
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int PN_SLOTS = 7;   // ring slots of W2_PANEL_BYTES: six panels in flight ahead of the one being multiplied
constexpr int PN_WAVES = 8;   // waves per workgroup; each issues 16 / 8 = 2 of a panel's 1 KB LDS-DMA pieces
constexpr int PN_PD = 3;      // backward epilogues whose phase loads are in flight ahead of the one being computed
constexpr int PN_PHASE_BYTES = PN_WAVES * 4 * 1024;  // their landing zone in LDS: per wave four sets of 4 dwords per lane

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  u32x4 u = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, u);
}

// Phase byte of t revolutions into byte N of `word`: t + 1.5 * 2^15 has its unit in the last place at 2^-8, so the low
// 8 bits of the sum's encoding are round-to-nearest-even(256 t) mod 256 (two's complement for t < 0), and an SDWA
// destination select writes exactly those bits (tools/probes/fmt8_probe.hip checks both on the device).
template <int N>
__device__ __forceinline__ void phase_byte(unsigned& word, float t, float magic) {
  static_assert(N >= 0 && N < 4, "byte");
  if (N == 0)
    asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(word) : "v"(t), "v"(magic));
  else if (N == 1)
    asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(word) : "v"(t), "v"(magic));
  else if (N == 2)
    asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(word) : "v"(t), "v"(magic));
  else
    asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(word) : "v"(t), "v"(magic));
}

// ---- the panel ring ---------------------------------------------------------------------------------------------
// Stream position p lives in slot p mod 7.  An INTERVAL consumes c panels (two per layer-0 chunk, one everywhere else):
//   pn_begin<N>():  s_waitcnt vmcnt(N) -- this wave's pieces of the interval's panels have landed --, s_barrier --
//   everybody's have, and everybody is done with the panels of the previous interval --, then requests up to position
//   p + 6 into the slots just freed.
// N: the vector-memory counter retires IN ORDER, so "at most N operations outstanding" covers a panel when at least N
// operations were issued after its pieces.  Behind the pieces of the interval's last panel lie the requests of the
// panels after it: two pieces each, 7 - c_prev - c panels (c_prev: what the previous interval consumed) -- N0 =
// 2 (7 - c_prev - c).  On top, where the preceding intervals are known to have issued S stash operations each (the
// steady state of the hidden layers: 4 stores forward, 4 phase fetches + 4 stores backward, instructions that are
// issued unconditionally), N = N0 + S h for the h <= 6 such intervals directly in front.  A smaller N only waits longer.
struct PnRing {
  const char* gbase;  // panel 0 of the image
  char* ring;
  int first, len;     // the stream cycles through images [first, first + len)
  int p, slot;        // next position to consume, its slot
  int req, req_img, req_slot;  // next position to request, its image (relative to first), its slot
  int w, lane;
};

__device__ __forceinline__ int pn_next(int slot) { return slot + 1 == PN_SLOTS ? 0 : slot + 1; }
__device__ __forceinline__ void pn_issue(const PnRing& r, int img, int slot) {
  const char* src = r.gbase + (size_t)img * W2_PANEL_BYTES + (size_t)(2 * r.w) * 1024 + r.lane * 16;
  char* dst = r.ring + slot * W2_PANEL_BYTES + (2 * r.w) * 1024;
#pragma unroll
  for (int n = 0; n < 2; ++n)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + n * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + n * 1024), 16, 0, 0);
}
__device__ __forceinline__ void pn_request(PnRing& r) {
  while (r.req < r.p + PN_SLOTS) {
    pn_issue(r, r.first + r.req_img, r.req_slot);
    ++r.req;
    r.req_slot = pn_next(r.req_slot);
    if (++r.req_img == r.len) r.req_img = 0;
  }
}
template <int N>
__device__ __forceinline__ void pn_begin(PnRing& r) {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_s_barrier();
  pn_request(r);
}
// the interval is over: its c panels are consumed
__device__ __forceinline__ void pn_advance(PnRing& r, int c) {
  r.p += c;
  for (int i = 0; i < c; ++i) r.slot = pn_next(r.slot);
}
__device__ __forceinline__ bf16x8 pn_frag(const PnRing& r, int slot, int f) {
  return *reinterpret_cast<const bf16x8*>(r.ring + slot * W2_PANEL_BYTES + (f * 64 + r.lane) * 16);
}
// one row block: 16 K-steps of the panel in `slot` against the 16 B operands
__device__ __forceinline__ f32x16 pn_mma_block(const PnRing& r, int slot, const bf16x8 (&b)[16]) {
  f32x16 acc = zero16();
#pragma unroll
  for (int t = 0; t < 16; ++t) acc = mfma_bf16(pn_frag(r, slot, t), b[t], acc);
  return acc;
}

// ---- phase fetches ------------------------------------------------------------------------------------------------
// The phase bytes of a backward epilogue are fetched PN_PD epilogues ahead and waited for with a counted vmcnt: hipcc's own
// wait insertion answers a load that is pending together with stores (and LDS-DMA) with vmcnt(0), which would drain the
// six weight panels in flight at every row block.  They travel by LDS-DMA into a landing zone of the wave (four sets of
// 4 x 64 dwords) and are read back with plain LDS loads behind the wait.  Not into registers: a load whose wait the
// compiler does not know about stays in flight across whatever the register allocator puts in between -- round 2 found a
// spill there, this round's first build a loop-carried copy (at the back edge of the layer loop the loads of the next
// layer's first row blocks were "moved" to where the next iteration expected them before they had landed), and
// amdgpu_num_vgpr does not keep the allocator out of a register range on this compiler.  An LDS destination has no
// allocator.

// ---- one tile group (four waves, 128 coordinates = one stash tile) ----------------------------------------------
// RB: role B (waves 4-7).  ACTIVE = false: the group has no tile in this launch shape (small batches run one stash
// tile per workgroup): it takes part in every barrier and issues its share of the DMA pieces, nothing else.
template <int MODE, bool RB, bool ACTIVE>
struct SirenTile {
  static constexpr int TL = W2_TL;
  const NetDesc& nd;
  const LossDesc& ld;
  const MlpArgs& a;
  PnRing& r;
  const float* bias_lds;
  const float* encB_lds;
  unsigned* phz;  // this wave's landing zone for phase fetches: [4 sets][4][64 lanes] dwords
  int lane, half, col, wcol, w;
  float mult;        // what the loss gradient is multiplied by (inr_w2.h: gradient-scale state)
  float amax = 0.f;  // max |dZ * mult| this wave has stashed
  float loss_acc = 0.f;

  // per tile
  unsigned* sv;
  int ts_bytes, voff;
  long long crow;
  bool valid, tile_ok;
  __amdgpu_buffer_rsrc_t rs_tile;
  bf16x8 hIn[16], hOut[16];

  __device__ __forceinline__ SirenTile(const NetDesc& nd_, const LossDesc& ld_, const MlpArgs& a_, PnRing& r_,
                                       const float* bias, const float* encB, unsigned* phz_, int lane_, int w_, float mult_)
      : nd(nd_), ld(ld_), a(a_), r(r_), bias_lds(bias), encB_lds(encB), phz(phz_), lane(lane_), half(lane_ >> 5), col(lane_ & 31),
        wcol((w_ & 3) * 32 + (lane_ & 31)), w(w_), mult(mult_) {}

  // -------- forward epilogue of row block mm of hidden layer l: t = acc + bias (revolutions) -> phase bytes to the
  // stash, h = sin(2 pi t) -> B operands 2 mm, 2 mm + 1 of the next layer
  __device__ __forceinline__ void epi_fwd(int l, int mm, const f32x16& acc, bf16x8 (&out)[16], int slot) {
    const float* bl = bias_lds + l * 256 + 32 * mm + 4 * half;
    const float magic = 49152.0f;
    const int so0 = (w2_stash_P(l) + 8 * mm * TL) * 4;
    float hv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + 8 * g);
      unsigned pk = 0;
      const float t0 = acc[4 * g + 0] + b4[0], t1 = acc[4 * g + 1] + b4[1], t2 = acc[4 * g + 2] + b4[2],
                  t3 = acc[4 * g + 3] + b4[3];
      hv[4 * g + 0] = __builtin_amdgcn_sinf(t0);
      hv[4 * g + 1] = __builtin_amdgcn_sinf(t1);
      hv[4 * g + 2] = __builtin_amdgcn_sinf(t2);
      hv[4 * g + 3] = __builtin_amdgcn_sinf(t3);
      phase_byte<0>(pk, t0, magic);
      phase_byte<1>(pk, t1, magic);
      phase_byte<2>(pk, t2, magic);
      phase_byte<3>(pk, t3, magic);
      __builtin_amdgcn_raw_buffer_store_b32(pk, rs_tile, voff, so0 + 2 * g * TL * 4, 0);  // quad 8 mm + 2 g (+ half in voff)
    }
    out[slot] = pack8(hv);
    out[slot + 1] = pack8(hv + 8);
  }

  // -------- backward epilogues: sequence e = 0, 1, ...: row block e & 7 of dZ_lz, lz = D-2 - (e >> 3)
  // fetch the four phase dwords of row block mm of layer lz (quads 8 mm + 2 g + half, g = 0..3) into set SET (= row block
  // & 3) of the landing zone.  lz < 0: past the tile's last epilogue -- the fetches are issued all the same, from layer 0,
  // so that the counted waits see a uniform sequence (tiles past the batch point at tile 0: the reads are always valid)
  template <int SET>
  __device__ __forceinline__ void bwd_loads(int lz, int mm) {
    const unsigned* src = sv + w2_stash_P(lz < 0 ? 0 : lz) + (8 * mm + half) * TL + wcol;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * g * TL),
                                       (__attribute__((address_space(3))) void*)(phz + (SET * 4 + g) * 64), 4, 0, 0);
  }
  // dZ = acc * cos(2 pi phase) (the transposed image carries w0): bf8 to the stash, bf16 into the next B operands.
  // WAIT: vector-memory operations issued since the set's fetches
  template <int SET, int WAIT>
  __device__ __forceinline__ void epi_bwd(int lz, int mm, const f32x16& acc, bf16x8 (&out)[16], int slot) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
    const int so0 = (w2_stash_G(lz, nd.D) + 8 * mm * TL) * 4;
    float dz[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned pw = phz[(SET * 4 + g) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float c = __builtin_amdgcn_cosf((float)((pw >> (8 * j)) & 255u) * 0.00390625f);
        dz[4 * g + j] = acc[4 * g + j] * c;
      }
      amax = fmaxf(amax, fmaxf(fabsf(dz[4 * g]), fabsf(dz[4 * g + 1])));
      amax = fmaxf(amax, fmaxf(fabsf(dz[4 * g + 2]), fabsf(dz[4 * g + 3])));
      int pk = __builtin_amdgcn_cvt_pk_bf8_f32(dz[4 * g], dz[4 * g + 1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_bf8_f32(dz[4 * g + 2], dz[4 * g + 3], pk, true);
      __builtin_amdgcn_raw_buffer_store_b32((unsigned)pk, rs_tile, voff, so0 + 2 * g * TL * 4, 0);
    }
    out[slot] = pack8(dz);
    out[slot + 1] = pack8(dz + 8);
  }

  __device__ __forceinline__ void copy_out_to_in() {
#pragma unroll
    for (int t = 0; t < 16; ++t) hIn[t] = hOut[t];
  }

  // -------- the tile ------------------------------------------------------------------------------------------------
  // stile: this group's stash tile (may be past the batch: then every stash access is a no-op and the lanes compute on
  // zeros).  Returns with every epilogue of the tile done.
  __device__ __forceinline__ void run(int stile) {
    const int D = nd.D, E = nd.E;
    const int nq0 = E / 32;  // layer-0 chunks of 64 encoder features (two panels each)
    constexpr bool FWD = MODE != MODE_BWD, BWD = MODE != MODE_FWD;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f, gtv[4] = {0.f, 0.f, 0.f, 0.f};
    bool sampled = false;
    if (ACTIVE) {
      tile_ok = stile < a.n_tiles;
      crow = (long long)stile * TL + wcol;
      valid = tile_ok && crow < a.B;
      sv = reinterpret_cast<unsigned*>(a.save) + (size_t)(tile_ok ? stile : 0) * nd.save_floats_per_tile;
      ts_bytes = (tile_ok && a.save != nullptr) ? w2_stash_dwords(D) * 4 : 0;  // 0: every stash access is a no-op
      voff = (half * TL + wcol) * 4;  // the lane's byte offset inside a quad pair: quad parity = lane half, own coordinate
      rs_tile = uniform_rsrc(sv, ts_bytes);
      if (FWD) {  // coordinates, sampling mask and target row in one batch of loads (row 0 where the lane has none)
        const long long cr = valid ? crow : 0;
        x0 = a.x[3 * cr + 0];
        x1 = a.x[3 * cr + 1];
        x2 = a.x[3 * cr + 2];
        unsigned char mk = 1;
        if (MODE == MODE_FUSED) {
          if (a.mask != nullptr) mk = a.mask[cr];
#pragma unroll
          for (int o = 0; o < 4; ++o)
            if (o < nd.out_f) gtv[o] = a.gt[cr * nd.out_f + o];
        }
        if (!valid) x0 = x1 = x2 = 0.f;
        sampled = valid && half == 0 && mk != 0;
      }
    }
    const float quarter = half ? 0.25f : 0.f;
    int si = 0;  // diagnostic builds: phase stamps 0, 1, 2, ... in program order (tools/stamps.py)
    constexpr int NW = PN_WAVES;
    (void)si;
    (void)NW;
    INR_STAMP(si); ++si;

    float dzl[4] = {0.f, 0.f, 0.f, 0.f};
    if (FWD) {
      // ================================ layer 0 ================================
      // 2E encoder features per coordinate, generated per K-step on the vector ALUs (half 0: sines, half 1: cosines of
      // features 8t .. 8t+7), contraction outermost: eight accumulator blocks.  Role A forms the features of chunk
      // ch + 1 behind the MFMAs of chunk ch, role B those of chunk ch in front of them.
      f32x16 acc8[8];
      bf16x8 bq[4];
      auto gen = [&](int ch) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int t = 4 * ch + s;
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float* bj = encB_lds + 3 * (8 * t + j);
            f[j] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(fmaf(x2, bj[2], fmaf(x1, bj[1], fmaf(x0, bj[0], quarter)))));
          }
          bq[s] = pack8(f);
        }
      };
      auto mma_chunk = [&]() {  // panels r.p (K-steps 0, 1 of the chunk) and r.p + 1 (K-steps 2, 3)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bf16x8 A[8];
#pragma unroll
          for (int m = 0; m < 8; ++m) A[m] = pn_frag(r, (s >> 1) ? pn_next(r.slot) : r.slot, (s & 1) * 8 + m);
#pragma unroll
          for (int m = 0; m < 8; ++m) acc8[m] = mfma_bf16(A[m], bq[s], acc8[m]);
        }
      };
      if (ACTIVE) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc8[m] = zero16();
        if (!RB) gen(0);
      }
      for (int ch = 0; ch < nq0; ++ch) {
        if (ch == 0)
          pn_begin<8>(r);  // c = 2 behind c_prev <= 1
        else
          pn_begin<6>(r);  // c = 2 behind c_prev = 2
        if (ACTIVE) {
          if (RB) gen(ch);
          mma_chunk();
          if (!RB && ch + 1 < nq0) gen(ch + 1);
        }
        pn_advance(r, 2);
      }
      INR_STAMP(si); ++si;
      // epilogue of layer 0, all eight row blocks (both roles: the 128 accumulator registers are free before the hidden
      // layers, whose loop keeps two sets of B operands)
      if (ACTIVE) {
#pragma unroll
        for (int m = 0; m < 8; ++m) epi_fwd(0, m, acc8[m], hIn, 2 * m);
      }
      INR_STAMP(si); ++si;

      // ================================ hidden layers 1 .. D-2 ================================
      f32x16 acc = zero16();
      for (int l = 1; l < D - 1; ++l) {
        const bool first = l == 1;
        static_for<0, 8>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          // requests: 10 operations behind the panel's pieces (8 behind a layer-0 chunk); + 4 stores per hidden interval (at
          // most the six in front), counted from the layer's second interval in the first layer (role B's first has no epilogue)
          constexpr int N0 = m == 0 ? 8 : 10, NF = N0 + 4 * (m > 0 ? (m - 1 > 6 ? 6 : m - 1) : 0), NS = 34;
          if (!ACTIVE) {
            if (first) pn_begin<N0>(r); else pn_begin<10>(r);
          } else if (first) {
            pn_begin<NF>(r);
          } else {
            pn_begin<NS>(r);
          }
          if (ACTIVE) {
            if (RB) {
              if (m == 0) {
                if (!first) {
                  epi_fwd(l - 1, 7, acc, hOut, 14);
                  copy_out_to_in();
                }
              } else {
                epi_fwd(l, m - 1, acc, hOut, 2 * (m > 0 ? m - 1 : 0));
              }
              acc = pn_mma_block(r, r.slot, hIn);
            } else {
              acc = pn_mma_block(r, r.slot, hIn);
              epi_fwd(l, m, acc, hOut, 2 * m);
              if (m == 7) copy_out_to_in();
            }
          }
          pn_advance(r, 1);
        });
        INR_STAMP(si); ++si;
      }

      // ================================ last layer: one row block (rows 0 .. out_f-1 live) ================================
      if (ACTIVE) pn_begin<34>(r); else pn_begin<10>(r);
      if (ACTIVE) {
        if (RB) {
          epi_fwd(D - 2, 7, acc, hOut, 14);
          copy_out_to_in();
        }
        const f32x16 accL = pn_mma_block(r, r.slot, hIn);
        float y[4], dy[4], g[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          float z = accL[o];  // half 0: rows 0..3
          if (o < nd.out_f) z += bias_lds[(D - 1) * 256 + o];
          act_fwd_rt(nd.last_act, z, nd.w0, y[o], dy[o]);
          g[o] = 0.f;
          if (half == 0 && valid && o < nd.out_f && a.out != nullptr) a.out[crow * nd.out_f + o] = y[o];
        }
        if (MODE == MODE_FUSED) {
          if (sampled) loss_acc += loss_row(ld, nd.out_f, y, gtv, g);
#pragma unroll
          for (int o = 0; o < 4; ++o) dzl[o] = (half == 0 && o < nd.out_f) ? g[o] * dy[o] * mult : 0.f;
        } else if (half == 0 && ts_bytes != 0) {  // split step: act'(z_last) for the backward half
          f32x4 d4 = {dy[0], dy[1], dy[2], dy[3]};
          *reinterpret_cast<f32x4*>(sv + w2_stash_dy(D) + 4 * wcol) = d4;
        }
      }
      pn_advance(r, 1);
      INR_STAMP(si); ++si;
    }

    if (BWD) {
      if (MODE == MODE_BWD && ACTIVE) {  // d(loss)/d(out) from the caller, act'(z_last) from the forward half
        if (valid && half == 0) {
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(sv + w2_stash_dy(D) + 4 * wcol);
#pragma unroll
          for (int o = 0; o < 4; ++o)
            if (o < nd.out_f) dzl[o] = a.dout[crow * nd.out_f + o] * d4[o] * mult;
        }
      }
      // ================================ dH_{D-2} = W_last^T dZ_last: one K-step, eight row blocks ================================
      pn_begin<10>(r);
      f32x16 acc = zero16();
      if (ACTIVE) {
        // dZ_last rows (0,1), (2,3) of this coordinate as fp16 pairs: two dwords behind the 8-bit tensors
        if (half == 0 && ts_bytes != 0) {
          unsigned* dzL = sv + w2_stash_dzl(D);
          dzL[wcol] = pack_f16(dzl[0], dzl[1]);
          dzL[TL + wcol] = pack_f16(dzl[2], dzl[3]);
        }
        float v[8] = {dzl[0], dzl[1], dzl[2], dzl[3], 0.f, 0.f, 0.f, 0.f};
        const bf16x8 b0 = pack8(v);  // k = output row: element j of half 0 is row j for j < 4
        static_for<0, PN_PD>([&](auto ec) { bwd_loads<decltype(ec)::value & 3>(D - 2, decltype(ec)::value); });
        static_for<0, 8>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          bwd_loads<(m + PN_PD) & 3>(D - 2 - ((m + PN_PD) >> 3), (m + PN_PD) & 7);
          const f32x16 a1 = mfma_bf16(pn_frag(r, r.slot, m), b0, zero16());
          // behind these loads: 4 (PD + e) operations for the first PD epilogues, then 8 PD
          epi_bwd<m & 3, (m < PN_PD ? 4 * (PN_PD + m) : 8 * PN_PD)>(D - 2, m, a1, hOut, 2 * m);
        });
        copy_out_to_in();
      }
      pn_advance(r, 1);
      INR_STAMP(si); ++si;

      // ================================ dH_{l-1} = W_l^T dZ_l, l = D-2 .. 1 ================================
      for (int l = D - 2; l >= 1; --l) {
        const bool first = l == D - 2;
        static_for<0, 8>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          // + 8 stash operations per interval (at most the six in front), counted from the second interval of the first layer
          // (role B's first has none)
          constexpr int NF = 10 + 8 * (m > 0 ? (m - 1 > 6 ? 6 : m - 1) : 0), NS = 58;
          if (!ACTIVE) {
            pn_begin<10>(r);
          } else if (first) {
            pn_begin<NF>(r);
          } else {
            pn_begin<NS>(r);
          }
          if (ACTIVE) {
            // epilogue e = 8 (D-1-l) + m produces row block m of dZ_{l-1}; its loads were issued PD epilogues ago
            if (RB) {
              if (m == 0) {
                if (!first) {  // row block 7 of dZ_l, then the layer's B operands are complete
                  bwd_loads<(7 + PN_PD) & 3>(l - ((7 + PN_PD) >> 3), (7 + PN_PD) & 7);
                  epi_bwd<3, 8 * PN_PD>(l, 7, acc, hOut, 14);
                  copy_out_to_in();
                }
              } else {
                constexpr int mp = m > 0 ? m - 1 : 0;
                bwd_loads<(mp + PN_PD) & 3>(l - 1 - ((mp + PN_PD) >> 3), (mp + PN_PD) & 7);
                epi_bwd<mp & 3, 8 * PN_PD>(l - 1, mp, acc, hOut, 2 * mp);
              }
              acc = pn_mma_block(r, r.slot, hIn);
            } else {
              acc = pn_mma_block(r, r.slot, hIn);
              bwd_loads<(m + PN_PD) & 3>(l - 1 - ((m + PN_PD) >> 3), (m + PN_PD) & 7);
              epi_bwd<m & 3, 8 * PN_PD>(l - 1, m, acc, hOut, 2 * m);
              if (m == 7 && l > 1) copy_out_to_in();
            }
          }
          pn_advance(r, 1);
        });
        INR_STAMP(si); ++si;
      }
      if (ACTIVE && RB) {  // role B's last epilogue: row block 7 of dZ_0
        bwd_loads<(7 + PN_PD) & 3>(-1, (7 + PN_PD) & 7);
        epi_bwd<3, 8 * PN_PD>(0, 7, acc, hOut, 14);
      }
      INR_STAMP(si); ++si;
    }
  }
};

// ---- the kernel ---------------------------------------------------------------------------------------------------
// Workgroup = eight waves = two tile groups; group g = waves 4g .. 4g+3 works on stash tile tpw * (workgroup tile) + g;
// wave w owns coordinates [32 (w & 3), + 32) of it: lane (col, half).
template <int MODE>
__global__ __launch_bounds__(512, 2) void inr_siren_bf16_kernel(const NetDesc nd, const LossDesc ld, const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  constexpr int NW = PN_WAVES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA destinations go through M0
  unsigned* phz = reinterpret_cast<unsigned*>(lds_raw + PN_SLOTS * W2_PANEL_BYTES) + w * 1024;  // [8 waves][4 sets][4][64]
  float* bias_lds = reinterpret_cast<float*>(lds_raw + PN_SLOTS * W2_PANEL_BYTES + PN_PHASE_BYTES);  // [D][256]
  float* encB_lds = bias_lds + nd.D * 256;                                          // [E][3]
  float* red_lds = encB_lds + 3 * nd.E;                                             // [8]
  const int D = nd.D, E = nd.E;
  for (int i = tid; i < D * 256; i += 64 * NW) bias_lds[i] = a.packed[nd.w2_bias_off + i];
  for (int i = tid; i < 3 * E; i += 64 * NW) encB_lds[i] = a.encB != nullptr ? a.encB[i] : 0.f;
  PnRing r;
  r.gbase = reinterpret_cast<const char*>(a.packed + nd.w2_off);
  r.ring = lds_raw;
  r.first = MODE == MODE_BWD ? w2_n_fwd(D, E) : 0;
  r.len = MODE == MODE_FUSED ? w2_np(D, E) : (MODE == MODE_FWD ? w2_n_fwd(D, E) : w2_np(D, E) - w2_n_fwd(D, E));
  r.p = 0, r.slot = 0, r.req = 0, r.req_img = 0, r.req_slot = 0;
  r.w = w, r.lane = lane;
  pn_request(r);    // prime the ring: positions 0 .. 6
  __syncthreads();  // tables in LDS (the DMAs are waited for by the first interval)

  // gradient-scale state (inr_w2.h): fused steps and split steps keep their own
  float* st = a.dz_state != nullptr ? a.dz_state + (MODE == MODE_BWD ? 4 : 0) : nullptr;
  float mult = 1.f;
  if (MODE != MODE_FWD && st != nullptr) {
    const float S = st[0];
    mult = MODE == MODE_FUSED ? S / ld.inv_count : S;
  }
  // two stash tiles per workgroup -- unless the batch is too small to fill the chip that way (25 000 rows = 196 tiles:
  // 98 workgroups on 256 CUs): then one, and waves 4..7 only help with the weight stream
  const int tpw = a.n_tiles > 256 ? 2 : 1;
  const int n_wtiles = (a.n_tiles + tpw - 1) / tpw;
  float loss_acc = 0.f, amax = 0.f;
  if (w < 4) {
    SirenTile<MODE, false, true> t(nd, ld, a, r, bias_lds, encB_lds, phz, lane, w, mult);
    for (int wtile = blockIdx.x; wtile < n_wtiles; wtile += gridDim.x) t.run(tpw * wtile);
    loss_acc = t.loss_acc, amax = t.amax;
  } else if (tpw == 2) {
    SirenTile<MODE, true, true> t(nd, ld, a, r, bias_lds, encB_lds, phz, lane, w, mult);
    for (int wtile = blockIdx.x; wtile < n_wtiles; wtile += gridDim.x) t.run(tpw * wtile + 1);
    loss_acc = t.loss_acc, amax = t.amax;
  } else {
    SirenTile<MODE, true, false> t(nd, ld, a, r, bias_lds, encB_lds, phz, lane, w, mult);
    for (int wtile = blockIdx.x; wtile < n_wtiles; wtile += gridDim.x) t.run(0);
  }
  // every DMA this wave issued has landed before the workgroup (and its LDS) goes away
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE != MODE_FWD) {
    // the step's largest scaled |dZ|: non-negative floats order like their bit patterns
    float m = amax;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0 && st != nullptr && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(st) + 1, __builtin_bit_cast(unsigned, m));
    if (blockIdx.x == 0 && tid == 0 && st != nullptr) {
      st[2] = mult;
      st[3] = st[0];
    }
  }
  if (MODE == MODE_FUSED) {
    // block loss partial -> slab loss word (fixed order: wave shuffle tree, then waves in order)
    float v = loss_acc;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if (lane == 0) red_lds[w] = v;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int i = 0; i < NW; ++i) t += red_lds[i];
      a.slabs[(size_t)blockIdx.x * nd.slab_floats + nd.slab_loss_off] = t;
    }
  }
}

template <int MODE>
inline hipError_t launch_siren_bf16_mode(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  const size_t lds_bytes = (size_t)PN_SLOTS * W2_PANEL_BYTES + PN_PHASE_BYTES +
                           ((size_t)nd.D * 256 + 3 * (size_t)nd.E + PN_WAVES) * sizeof(float);
  if (lds_bytes > 160 * 1024 || a.save_by_block) return hipErrorInvalidValue;
  if (MODE != MODE_FWD && (a.save == nullptr || a.dz_state == nullptr)) return hipErrorInvalidValue;
  if (MODE == MODE_FUSED && a.slabs == nullptr) return hipErrorInvalidValue;
  hipError_t e = allow_full_lds<inr_siren_bf16_kernel<MODE>>();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(inr_siren_bf16_kernel<MODE>, dim3(grid), dim3(64 * PN_WAVES), lds_bytes, st, nd, ld, a);
  return hipGetLastError();
}

}  // namespace inr
