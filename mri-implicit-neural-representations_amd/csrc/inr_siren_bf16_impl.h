// inr_siren_bf16_impl.h -- SIREN on the bf16 matrix pipe, third design: "weight panels in LDS, activations in registers,
// one row block at a time, one interleaved MFMA + epilogue stream per wave".  Forward + pointwise loss + backward-to-inputs
// for tiles of 256 coordinates (MODE_FUSED), or the two halves of a split step (MODE_FWD: outputs + stash; MODE_BWD: from
// d(loss)/d(out)); the weight gradients are left to inr_dw_gemm_bf16.hip, which reads the 8-bit operands this kernel stashes.
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
//     operand -- is cut into 16 slices that ride behind the next block's MFMAs, one value a slot, in EVERY wave (see
//     SirenTile; the first form of this design had the two waves of a SIMD take turns instead, which did not overlap);
//   * weights still travel HBM/L2 -> LDS once per workgroup by LDS-DMA, now as 16 KB panels through an 8-slot ring that
//     holds the interval being multiplied (four row blocks) and the next one.
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

constexpr int PN_SLOTS = 8;   // ring slots of W2_PANEL_BYTES
constexpr int PN_WAVES = 8;   // waves per workgroup; each moves 16 / 8 = 2 of a panel's 1 KB pieces
#ifndef PN_FD_N
#define PN_FD_N 6
#endif
constexpr int PN_FD = PN_FD_N;  // A fragments read ahead of the MFMA that takes them
#ifndef PN_PD_N
#define PN_PD_N 3
#endif
constexpr int PN_PD = PN_PD_N;      // backward epilogues whose phase loads are in flight ahead of the one being computed

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

// "This value exists HERE": an empty volatile statement that takes the operand's four registers.  LLVM sinks arithmetic
// whose result is only consumed much later (the sines of an epilogue feed the NEXT layer's GEMM) down to that consumer --
// the first build of this kernel computed the sines of seven row blocks in one burst behind the layer's last panel, with
// the pre-activations of all seven alive until then.  Volatile statements keep their order with the barriers.
__device__ __forceinline__ void pin(bf16x8& v) {
  u32x4 u = __builtin_bit_cast(u32x4, v);
  asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]));
  v = __builtin_bit_cast(bf16x8, u);
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
// Stream position p lives in slot p & 7.  An INTERVAL consumes c <= 4 panels between two barriers: two layer-0 chunks (one
// at an odd tail), FOUR row blocks of a hidden layer, the last layer, its transpose.  The ring holds the interval being
// multiplied and the next one: right behind barrier k every wave requests its two 1 KB pieces of each panel of interval
// k + 1 by LDS-DMA (global_load_lds_dwordx4: no registers) into the slots interval k - 1 was the last to read; a whole
// interval later, pn_begin<N>() = s_waitcnt vmcnt(N) -- this wave's pieces have landed -- + s_barrier -- everybody's have,
// and everybody is done with interval k.  N: the vector-memory counter retires IN ORDER, so "at most N outstanding" covers
// the requests when at least N operations were issued behind them; every interval of an active wave ends with the four
// stash stores of an epilogue (N = 4: those may stay in flight), layer 0 and the idle waves of a one-tile workgroup issue
// nothing else (N = 0).  A smaller N only waits longer.
// (Round 3a had two row blocks per barrier and seven panels in flight; removing the barriers from that build took 10 us off
// 112.  Panels through registers -- load an interval, ds_write the next -- measured the same as LDS-DMA within 1 % and cost
// 16 registers: profiles/r03_fused_bf16_knockouts.txt.)
struct PnRing {
  const char* gbase;  // panel 0 of the image
  char* ring;
  int first, len;     // the stream cycles through images [first, first + len)
  int p;              // next position to consume
  int req, req_img;   // next position to request, its image (relative to first)
  int w, lane;
  __amdgpu_buffer_rsrc_t rs;  // the whole panel image
  int voff;                   // this lane's byte offset inside a panel: its wave's two pieces, 16 bytes a lane
};
constexpr int pn_min(int a, int b) { return a < b ? a : b; }
constexpr int pn_max(int a, int b) { return a > b ? a : b; }

__device__ __forceinline__ void pn_issue(const PnRing& r, int img, int slot) {
  // (through a buffer descriptor: scalar image offset + one 32-bit lane offset formed once -- as global_load_lds the
  // compiler formed a 64-bit vector address per image up front, and spilled them)
  char* dst = r.ring + slot * W2_PANEL_BYTES + (2 * r.w) * 1024;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r.rs, (__attribute__((address_space(3))) void*)dst, 16, r.voff, img * W2_PANEL_BYTES, 0, 0);
  // (the second piece's +1 KB rides in the scalar offset: an immediate offset would move the LDS address as well)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r.rs, (__attribute__((address_space(3))) void*)(dst + 1024), 16, r.voff,
                                           img * W2_PANEL_BYTES + 1024, 0, 0);
}
// the next c panels of the stream
__device__ __forceinline__ void pn_fetch(PnRing& r, int c) {
  for (int i = 0; i < c; ++i) {
    pn_issue(r, r.first + r.req_img, r.req & (PN_SLOTS - 1));
    ++r.req;
    if (++r.req_img == r.len) r.req_img = 0;
  }
}
template <int N>
__device__ __forceinline__ void pn_begin(PnRing& r) {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_s_barrier();
}
// fragment f of the panel at stream position p: ONE address register per panel (opaque to the compiler, which otherwise
// forms the sixteen lane addresses up front, a register each, and adds the slot to every one of them) + f KB as an immediate
__device__ __forceinline__ unsigned pn_base(const PnRing& r, int p) {
  unsigned b = (unsigned)((p & (PN_SLOTS - 1)) * W2_PANEL_BYTES + r.lane * 16);
  asm volatile("" : "+v"(b));
  return b;
}
__device__ __forceinline__ bf16x8 pn_frag(const PnRing& r, unsigned base, int f) {
  return *reinterpret_cast<const bf16x8*>(r.ring + base + f * 1024);
}
// (row blocks: SirenTile::block -- 16 K-steps of one panel, each MFMA followed by a slice of the row block before's epilogue)

// ---- vector loads the compiler does not see ------------------------------------------------------------------------
// The phase bytes of a backward epilogue are fetched PN_PD epilogues ahead by inline assembly and waited for with a
// counted vmcnt: hipcc's own wait insertion answers a load that is pending together with stores (and LDS-DMA) with
// vmcnt(0), which would drain the seven weight panels in flight at every row block (and an LDS-DMA landing zone instead
// of registers costs an LDS-DMA issue, 100-185 cycles, per dword fetched: measured, +11 k cycles per layer).  The price:
// between such a load and its wait the destination registers belong to the memory system, and the register allocator
// does not know.  Three ways it has gone wrong, each closed structurally and checked on every build by
// tools/check_inflight_regs.py (tests/test_host.py):
//   * a load whose result nobody uses (the uniform tail of the sequence) gets a destination the allocator hands to the
//     next instruction -- the tail issues stores that a zero-sized descriptor drops instead;
//   * a load in flight across a loop's back edge is "moved" to where the next iteration expects it before it has landed
//     -- the layer loops are unrolled at compile time (the kernel is instantiated per depth): the only back edge left is
//     the tile loop's, and no load is in flight there;
//   * an SGPR operand reloaded from a spill (v_readlane) right in front of the load, which gfx9 only tolerates five wait
//     states later and the compiler only pads for its own memory instructions -- the four loads of a row block are ONE
//     statement with the pad in front and immediates for the later offsets.
__device__ __forceinline__ u32x4 w2_rsrc_words(const void* p, int bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  r[2] = (unsigned)bytes;
  r[3] = 0x00020000u;
  return r;
}
// the four phase dwords of a row block (quads 8 mm + 2 g + half, g = 0..3: 512 B apart)
__device__ __forceinline__ void w2_load4_asm(unsigned (&z)[4], const u32x4& rsrc, int voff, int soff) {
  asm volatile(
      "s_nop 4\n\t"
      "buffer_load_dword %0, %4, %5, %6 offen\n\t"
      "buffer_load_dword %1, %4, %5, %6 offen offset:512\n\t"
      "buffer_load_dword %2, %4, %5, %6 offen offset:1024\n\t"
      "buffer_load_dword %3, %4, %5, %6 offen offset:1536"
      : "=&v"(z[0]), "=&v"(z[1]), "=&v"(z[2]), "=&v"(z[3])
      : "v"(voff), "s"(rsrc), "s"(soff)
      : "memory");
}
// at most N vector-memory operations outstanding; the four registers are operands so that no use moves above the wait
template <int N>
__device__ __forceinline__ void w2_wait4(unsigned (&z)[4]) {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]) : "n"(N) : "memory");
}

// ---- one tile group (four waves, 128 coordinates = one stash tile) ----------------------------------------------
// NH: hidden layers behind the first (D = NH + 2 Linear layers).  ACTIVE = false: the group has no tile in this launch
// shape (small batches run one stash tile per workgroup): it takes part in every barrier and issues its share of the DMA
// pieces, nothing else.
//
// ONE instruction stream, the same for all eight waves: a row block is 16 slots of  [MFMA t][slice t of the epilogue of
// the row block BEFORE]  -- one value (bias, sine, phase byte; or cosine, product, bf8) per slot, ~20-26 cycles of vector
// issue beside the MFMA's 32 on the pipe --, fenced so that the compiler keeps the interleaving; two accumulators
// alternate.  The first row block of a layer reads ALL of the layer below, including the two B operands its last epilogue
// produces: there the slices run at double rate (two values a slot, done behind MFMA 7) and the block's K-steps 14 and 15
// are the ones that wait for them.  (Rounds 2-3a staggered the two waves of a SIMD instead -- one multiplies while the
// other runs an epilogue.  Measured on the dW GEMM, which had the same plan: the partner made no progress beside the
// multiplying wave, the kernel took the SUM of its matrix and vector times -- profiles/r03_dw_gemm_knockouts.txt.)
template <int MODE, int NH, bool ACTIVE>
struct SirenTile {
  static constexpr int TL = W2_TL, D = NH + 2;
  static constexpr int NE = 8 * (NH + 1);  // backward epilogues per tile: row block e & 7 of dZ_lz, lz = D-2 - (e >> 3)
  const NetDesc& nd;
  const LossDesc& ld;
  const MlpArgs& a;
  PnRing& r;
  const float* bias_lds;
  const float* encB_lds;
  int lane, half, col, wcol, w;
  int bias_voff;     // byte offset of the lane's part of the bias table (+ 4 half rows) from the ring's base
  float mult;        // what the loss gradient is multiplied by (inr_w2.h: gradient-scale state)
  float amax = 0.f;  // max |dZ * mult| this wave has stashed
  float loss_acc = 0.f;

  // per tile
  unsigned* sv;
  int ts_bytes, voff;
  long long crow;
  bool valid, tile_ok;
  __amdgpu_buffer_rsrc_t rs_tile;
  u32x4 rz_tile;
  u32x4 hb[2][16];   // B operands (bf16 pairs): layer l reads hb[(l - 1) & 1] and writes hb[l & 1]; backward likewise
  f32x16 acc2[2];    // row block i accumulates in acc2[i & 1] while the epilogue slices read the other
  unsigned pz[PN_PD + 1][4];  // phase dwords of backward epilogues e, e+1, .. (set e % (PD + 1))
  // state of an epilogue between its slices
  f32x4 eb4, eb4n;   // bias of the value group in work / the next one
  float eprev = 0.f;
  unsigned epk = 0;

  __device__ __forceinline__ SirenTile(const NetDesc& nd_, const LossDesc& ld_, const MlpArgs& a_, PnRing& r_,
                                       const float* bias, const float* encB, int lane_, int w_, float mult_)
      : nd(nd_), ld(ld_), a(a_), r(r_), bias_lds(bias), encB_lds(encB), lane(lane_), half(lane_ >> 5), col(lane_ & 31),
        wcol((w_ & 3) * 32 + (lane_ & 31)), w(w_), mult(mult_) {
    bias_voff = PN_SLOTS * W2_PANEL_BYTES + 16 * half;
    asm volatile("" : "+v"(bias_voff));
  }

  // -------- forward epilogue of row block MM of hidden layer L, value V of 16 (accumulator register V: row
  // 32 MM + 8 (V >> 2) + 4 half + (V & 3)): t = acc + bias (revolutions) -> phase byte (a dword store per four values),
  // h = sin(2 pi t) -> bf16, dword (V >> 1) & 3 of B operand 2 MM + (V >> 3) of the next layer
  template <int L, int MM, int V>
  __device__ __forceinline__ void fwd_value(const f32x16& acc) {
    constexpr int g = V >> 2, j = V & 3;
    // (one opaque base register + immediates: left to itself the compiler forms the 4 x 8 x D table addresses up front,
    // a register each -- 36 of them spilled)
    const float* bl = reinterpret_cast<const float*>(r.ring + bias_voff) + L * 256 + 32 * MM;
    if constexpr (V == 0) {
      eb4 = *reinterpret_cast<const f32x4*>(bl);
      eb4n = *reinterpret_cast<const f32x4*>(bl + 8);
    } else if constexpr (j == 0) {
      eb4 = eb4n;
      if constexpr (g < 3) eb4n = *reinterpret_cast<const f32x4*>(bl + 8 * (g + 1));
    }
    if constexpr (j == 0) epk = 0;
    const float t = acc[V] + eb4[j];
    const float h = __builtin_amdgcn_sinf(t);
    phase_byte<j>(epk, t, 49152.0f);
    // ("this value exists HERE": the sine feeds the NEXT layer's GEMM, and left alone is computed down there)
    if constexpr (j & 1) {
      unsigned d = pack_bf16(eprev, h);
      asm volatile("" : "+v"(d));
      hb[L & 1][2 * MM + (V >> 3)][(V >> 1) & 3] = d;
    } else {
      eprev = h;
      asm volatile("" : "+v"(eprev));
    }
    if constexpr (j == 3) {
      constexpr int so0 = (L * W2_TENSOR_DWORDS + 8 * MM * W2_HALF) * 4;  // w2_stash_P(L), quad 8 MM
      __builtin_amdgcn_raw_buffer_store_b32(epk, rs_tile, voff, so0 + 2 * g * W2_HALF * 4, 0);  // quad 8 mm + 2 g (+ half in voff)
    }
  }
  // slice T of 16 (RATE 1) or of 8 (RATE 2: two values, nothing behind slot 7)
  template <int L, int MM, int RATE, int T>
  __device__ __forceinline__ void fwd_slice(const f32x16& acc) {
    if constexpr (RATE == 1) {
      fwd_value<L, MM, T>(acc);
    } else if constexpr (T < 8) {
      fwd_value<L, MM, 2 * T>(acc);
      fwd_value<L, MM, 2 * T + 1>(acc);
    }
  }
  template <int L, int MM>
  __device__ __forceinline__ void epi_fwd(const f32x16& acc) {  // a whole epilogue at once (layer 0)
    static_for<0, 16>([&](auto vc) { fwd_value<L, MM, decltype(vc)::value>(acc); });
  }

  // -------- backward epilogues, sequence e = 0 .. NE-1.  Each wave runs  [loads of e + PD][epilogue e]  in this order.
  // the four phase loads of epilogue E; E >= NE: four stores that a zero-sized descriptor drops (the counted waits see a
  // uniform sequence; a load nobody reads would have a destination the allocator reuses while it is in flight)
  template <int E>
  __device__ __forceinline__ void bwd_loads() {
    static_assert(2 * W2_HALF * 4 == 512, "w2_load4_asm's immediates");
    if constexpr (E >= NE) {
      // (assembly: four identical stores through the builtin are one store after dead-store elimination -- and three
      // operations fewer than the waits count)
      const u32x4 none = w2_rsrc_words(sv, 0);
      asm volatile(
          "s_nop 4\n\t"
          "buffer_store_dword %0, %0, %1, 0 offen\n\tbuffer_store_dword %0, %0, %1, 0 offen\n\t"
          "buffer_store_dword %0, %0, %1, 0 offen\n\tbuffer_store_dword %0, %0, %1, 0 offen"
          :
          : "v"(voff), "s"(none)
          : "memory");
    } else {
      constexpr int lz = D - 2 - (E >> 3), mm = E & 7;
      w2_load4_asm(pz[E % (PN_PD + 1)], rz_tile, voff, (lz * W2_TENSOR_DWORDS + 8 * mm * W2_HALF) * 4);
    }
  }
  // value V of backward epilogue E: dZ = acc * cos(2 pi phase) (the transposed image carries w0): bf8 to the stash (a dword
  // store per four values), bf16 into the next B operands.  Value 0 first issues the loads of epilogue E + PD and waits for
  // its own: behind those lie 4 (PD + e) operations for the first PD epilogues of a tile, 8 PD from then on.
  template <int E, int V>
  __device__ __forceinline__ void bwd_value(const f32x16& acc) {
    constexpr int lz = D - 2 - (E >> 3), mm = E & 7, set = E % (PN_PD + 1), g = V >> 2, j = V & 3;
    if constexpr (V == 0) {
      bwd_loads<E + PN_PD>();
      w2_wait4<(E < PN_PD ? 4 * (PN_PD + E) : 8 * PN_PD)>(pz[set]);
    }
    const unsigned pw = pz[set][g];
    // cos(2 pi p / 256) with ONE instruction in front of v_cos_f32 (which takes revolutions): the float 128 + p / 256 =
    // 0x43000000 | p << 8, its bytes picked by v_perm_b32 -- {0x43 from the constant, 0, byte j of the phase dword, 0}.
    // (Convert + multiply were two; at one value per MFMA slot every vector instruction counts: the kernel issues ~8 per
    // MFMA where ~5 hide, profiles/r03_bf16_pmc_by_grid.csv.)
    const unsigned cb = __builtin_amdgcn_perm(0x43000000u, pw, 0x070C000Cu | ((unsigned)j << 8));
    const float c = __builtin_amdgcn_cosf(__builtin_bit_cast(float, cb));
    const float dz = acc[V] * c;
    if constexpr (j & 1) {
      // amax = max(amax, |eprev|, |dz|) as the one instruction it is (the compiler made three of it)
      asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(eprev), "v"(dz));
      epk = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(eprev, dz, j == 1 ? 0 : (int)epk, j == 3);
      unsigned d = pack_bf16(eprev, dz);
      asm volatile("" : "+v"(d), "+v"(epk));
      hb[(E >> 3) & 1][2 * mm + (V >> 3)][(V >> 1) & 3] = d;
    } else {
      eprev = dz;
      asm volatile("" : "+v"(eprev));
    }
    if constexpr (j == 3) {
      constexpr int so0 = ((D - 1 + lz) * W2_TENSOR_DWORDS + 8 * mm * W2_HALF) * 4;  // w2_stash_G(lz, D), quad 8 mm
      __builtin_amdgcn_raw_buffer_store_b32(epk, rs_tile, voff, so0 + 2 * g * W2_HALF * 4, 0);
    }
    if constexpr (V == 15) asm volatile("" : "+v"(amax));  // (sunk to the kernel's end the maximum kept every dZ alive)
  }
  template <int E, int RATE, int T>
  __device__ __forceinline__ void bwd_slice(const f32x16& acc) {
    if constexpr (RATE == 1) {
      bwd_value<E, T>(acc);
    } else if constexpr (T < 8) {
      bwd_value<E, 2 * T>(acc);
      bwd_value<E, 2 * T + 1>(acc);
    }
  }
  template <int E>
  __device__ __forceinline__ void epi_bwd(const f32x16& acc) {  // a whole epilogue at once
    static_for<0, 16>([&](auto vc) { bwd_value<E, decltype(vc)::value>(acc); });
  }

  // -------- one row block: 16 K-steps of panel p against the 16 B operands b, MFMA t followed by slice(t).  The sixteen
  // MFMAs are one dependent chain (a single accumulator).  Fragment reads: PN_FD up front, then one per slot.
  template <class Slice>
  __device__ __forceinline__ void block(f32x16& acc, int p, const u32x4 (&b)[16], Slice&& slice) {
    bf16x8 A[16];
    const unsigned pb = pn_base(r, p);
    static_for<0, PN_FD>([&](auto tc) { A[decltype(tc)::value] = pn_frag(r, pb, decltype(tc)::value); });
    acc = zero16();
    static_for<0, 16>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if constexpr (t + PN_FD < 16) A[t + PN_FD] = pn_frag(r, pb, t + PN_FD);
      acc = mfma_bf16(A[t], __builtin_bit_cast(bf16x8, b[t]), acc);
      slice(tc);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  // -------- the two row blocks of an interval (panels p, p + 1; accumulators acc2[0], acc2[1]) as ONE stream of 32 slots:
  // the second block's first fragments are read behind the first block's last MFMAs (a block that starts with its own
  // reads waits out an LDS latency with the matrix pipe idle -- under eight waves' reads, some hundreds of cycles);
  // `between` (the panel staging) runs behind slot 15.
  template <class SliceA, class Between, class SliceB>
  __device__ __forceinline__ void block_pair(int p, const u32x4 (&b)[16], SliceA&& slice_a, Between&& between, SliceB&& slice_b) {
    bf16x8 A[32];
    const unsigned pb0 = pn_base(r, p), pb1 = pn_base(r, p + 1);
    static_for<0, PN_FD>([&](auto tc) { A[decltype(tc)::value] = pn_frag(r, pb0, decltype(tc)::value); });
    static_for<0, 32>([&](auto qc) {
      constexpr int q = decltype(qc)::value, nq = q + PN_FD;
      if constexpr (nq < 32) A[nq] = pn_frag(r, nq < 16 ? pb0 : pb1, nq & 15);
      // (acc2[1] still feeds the first block's slices: it is cleared by its own first MFMA)
      acc2[q >> 4] = mfma_bf16(A[q], __builtin_bit_cast(bf16x8, b[q & 15]), (q & 15) == 0 ? zero16() : acc2[q >> 4]);
      if constexpr (q < 16)
        slice_a(std::integral_constant<int, q>{});
      else
        slice_b(std::integral_constant<int, q - 16>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (q == 15) between();
    });
  }

  // -------- the tile ------------------------------------------------------------------------------------------------
  // stile: this group's stash tile (may be past the batch: then every stash access is a no-op and the lanes compute on
  // zeros).  Returns with every epilogue of the tile done.
  __device__ __forceinline__ void run(int stile) {
    const int E = nd.E;
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
      // the lane's byte offset inside an 8-bit tensor (inr_w2.h): its half tile, quad parity = lane half, own coordinate
      voff = ((wcol >> 6) * (W2_TENSOR_DWORDS / 2) + half * W2_HALF + (wcol & (W2_HALF - 1))) * 4;
      rs_tile = uniform_rsrc(sv, ts_bytes);
      if (BWD) rz_tile = w2_rsrc_words(sv, ts_bytes);
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
    int si = 0;  // diagnostic builds: phase stamps 0, 1, 2, ... in program order (tools/stamps.py)
    constexpr int NW = PN_WAVES;
    (void)si;
    (void)NW;
    INR_STAMP(si); ++si;

    float dzl[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FWD) {
      // ================================ layer 0 ================================
      // 2E encoder features per coordinate, generated per K-step on the vector ALUs (half 0: sines, half 1: cosines of
      // features 8t .. 8t+7), contraction outermost: eight accumulator blocks.  A chunk = 4 K-steps x 8 row blocks = 32
      // slots of  [MFMA][one feature of the NEXT chunk]; the fragments of a K-step are read during the K-step before.
      {
        f32x16 acc8[8];
        u32x4 bq[2][4];  // B operands of the chunk being multiplied / being generated
        bf16x8 A[2][8];
        float fprev = 0.f, fph = 0.f;
        // Element j of K-step s of chunk ch into dst.  A lane half holds the sine AND the cosine of four frequencies per
        // K-step -- elements (2i, 2i+1) = (sin, cos) of frequency 8 s + 4 half + i of the chunk (the weight panels are packed
        // in that k order, inr_aux.hip put_w2) -- so ONE phase chain serves two elements (both halves used to run the same
        // eight chains, the cosine half from a start of 1/4 turn).  The frequency's row of the encoder matrix was read from
        // LDS two slots earlier; the next one's (of chunk chn: the same chunk, or the next behind its last element) here.
        f32x4 nbr = {0.f, 0.f, 0.f, 0.f};
        auto enc_row = [&](int ch, int q, f32x4& dst) {  // q = 8 s + j < 32, j even
          dst = *reinterpret_cast<const f32x4*>(encB_lds + 4 * (32 * ch + (q & ~7) + 4 * half + ((q & 7) >> 1)));
        };
        auto gen_value = [&](int ch, int chn, auto SC, auto JC, u32x4 (&dst)[4]) {
          constexpr int s = decltype(SC)::value, j = decltype(JC)::value, q = 8 * s + j;
          if constexpr ((j & 1) == 0) {
            const f32x4 b = nbr;
            fph = __builtin_amdgcn_fractf(fmaf(x2, b[2], fmaf(x1, b[1], x0 * b[0])));
            fprev = __builtin_amdgcn_sinf(fph);
            if constexpr (q + 2 < 32)
              enc_row(ch, q + 2, nbr);
            else
              enc_row(chn, q + 2 - 32, nbr);
            asm volatile("" : "+v"(fprev));
          } else {
            unsigned d = pack_bf16(fprev, __builtin_amdgcn_cosf(fph));
            asm volatile("" : "+v"(d));
            dst[s][j >> 1] = d;
          }
        };
        if (ACTIVE) {
#pragma unroll
          for (int m = 0; m < 8; ++m) acc8[m] = zero16();
          enc_row(0, 0, nbr);
          static_for<0, 4>([&](auto sc) { static_for<0, 8>([&](auto jc) { gen_value(0, nq0 > 1 ? 1 : 0, sc, jc, bq[0]); }); });
        }
        for (int ch = 0; ch < nq0; ch += 2) {  // an interval = two chunks (one at an odd tail); the B operand sets alternate
          const int nch = nq0 - ch >= 2 ? 2 : 1, rem = nq0 - ch - nch;
          pn_begin<0>(r);
          pn_fetch(r, rem == 1 ? 2 : 4);  // the next interval: two chunks, the odd last one, or the first four hidden row blocks
          static_for<0, 2>([&](auto hc) {
            constexpr int hh = decltype(hc)::value;
            const int chh = ch + hh;
            if (hh < nch && ACTIVE) {
              const int nxt = chh + 1 < nq0 ? chh + 1 : chh;  // (past the end: the last chunk again, never multiplied)
              const int nxt2 = nxt + 1 < nq0 ? nxt + 1 : nxt;
              const unsigned pb[2] = {pn_base(r, r.p + 2 * hh), pn_base(r, r.p + 2 * hh + 1)};  // K-steps 0, 1 / 2, 3 of the chunk
              static_for<0, 8>([&](auto mc) { A[0][decltype(mc)::value] = pn_frag(r, pb[0], decltype(mc)::value); });
              static_for<0, 4>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                static_for<0, 8>([&](auto mc) {
                  constexpr int m = decltype(mc)::value;
                  if constexpr (s < 3) A[(s + 1) & 1][m] = pn_frag(r, pb[(s + 1) >> 1], ((s + 1) & 1) * 8 + m);
                  acc8[m] = mfma_bf16(A[s & 1][m], __builtin_bit_cast(bf16x8, bq[hh][s]), acc8[m]);
                  gen_value(nxt, nxt2, sc, mc, bq[hh ^ 1]);
                  __builtin_amdgcn_sched_barrier(0);
                });
              });
            }
          });
          r.p += 2 * nch;
        }
        INR_STAMP(si); ++si;
        // epilogue of layer 0, all eight row blocks
        if (ACTIVE) static_for<0, 8>([&](auto mc) { epi_fwd<0, decltype(mc)::value>(acc8[decltype(mc)::value]); });
        INR_STAMP(si); ++si;
      }

      // ================================ hidden layers 1 .. D-2, row block by row block ================================
      // Four row blocks (four panels) per barrier.  Row block i = 8 (l - 1) + m carries the epilogue of row block i - 1.
      static_for<0, 2 * NH>([&](auto qc) {
        constexpr int Q = decltype(qc)::value, i0 = 4 * Q, l = 1 + (i0 >> 3), m0 = i0 & 7;
        pn_begin<(ACTIVE ? 4 : 0)>(r);
        pn_fetch(r, Q + 1 < 2 * NH ? 4 : 1);
        if (ACTIVE) {
          block_pair(
              r.p, hb[(l - 1) & 1],
              [&](auto tc) {
                // (m0 == 0: the epilogue of the layer below's last row block, whose B operands this block's last K-steps read)
                if constexpr (i0 > 0) fwd_slice<1 + ((i0 - 1) >> 3), (i0 - 1) & 7, (m0 == 0 ? 2 : 1), decltype(tc)::value>(acc2[1]);
              },
              [&]() {}, [&](auto tc) { fwd_slice<l, m0, 1, decltype(tc)::value>(acc2[0]); });
          block_pair(
              r.p + 2, hb[(l - 1) & 1], [&](auto tc) { fwd_slice<l, m0 + 1, 1, decltype(tc)::value>(acc2[1]); }, [&]() {},
              [&](auto tc) { fwd_slice<l, m0 + 2, 1, decltype(tc)::value>(acc2[0]); });
        }
        r.p += 4;
        if constexpr (m0 == 4) {
          INR_STAMP(si); ++si;
        }
      });

      // ================================ last layer: one row block (rows 0 .. out_f-1 live) ================================
      pn_begin<(ACTIVE ? 4 : 0)>(r);
      pn_fetch(r, MODE == MODE_FUSED ? 1 : (nq0 >= 2 ? 4 : 2));  // the transpose, or (MODE_FWD) the next tile's first chunks
      if (ACTIVE) {
        block(acc2[0], r.p, hb[NH & 1], [&](auto tc) { fwd_slice<NH, 7, 2, decltype(tc)::value>(acc2[1]); });
        const f32x16& accL = acc2[0];
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
      r.p += 1;
      INR_STAMP(si); ++si;
    }

    if constexpr (BWD) {
      if (MODE == MODE_BWD && ACTIVE) {  // d(loss)/d(out) from the caller, act'(z_last) from the forward half
        if (valid && half == 0) {
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(sv + w2_stash_dy(D) + 4 * wcol);
#pragma unroll
          for (int o = 0; o < 4; ++o)
            if (o < nd.out_f) dzl[o] = a.dout[crow * nd.out_f + o] * d4[o] * mult;
        }
      }
      // ================================ dH_{D-2} = W_last^T dZ_last: one K-step, eight row blocks ================================
      pn_begin<(ACTIVE ? 4 : 0)>(r);  // (behind the last layer's interval, or the tile before's last epilogue)
      pn_fetch(r, 4);
      if (ACTIVE) {
        // dZ_last rows (0,1), (2,3) of this coordinate as fp16 pairs: two dwords behind the 8-bit tensors
        if (half == 0 && ts_bytes != 0) {
          unsigned* dzL = sv + w2_stash_dzl(D);
          dzL[wcol] = pack_f16(dzl[0], dzl[1]);
          dzL[TL + wcol] = pack_f16(dzl[2], dzl[3]);
        }
        float v[8] = {dzl[0], dzl[1], dzl[2], dzl[3], 0.f, 0.f, 0.f, 0.f};
        const bf16x8 b0 = pack8(v);  // k = output row: element j of half 0 is row j for j < 4
        static_for<0, PN_PD>([&](auto ec) { bwd_loads<decltype(ec)::value>(); });
        // the MFMA of row block m + 1 is on the pipe while the epilogue of row block m runs
        const unsigned pb = pn_base(r, r.p);
        acc2[0] = mfma_bf16(pn_frag(r, pb, 0), b0, zero16());
        static_for<0, 8>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          if constexpr (m < 7) acc2[(m + 1) & 1] = mfma_bf16(pn_frag(r, pb, m + 1), b0, zero16());
          epi_bwd<m>(acc2[m & 1]);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      r.p += 1;
      INR_STAMP(si); ++si;

      // ================================ dH_{l-1} = W_l^T dZ_l, l = D-2 .. 1, row block by row block ================================
      // Row block j = 8 li + m carries backward epilogue 8 + j - 1 (row block j - 1 of this sequence; the first one has
      // none: the eight epilogues of dZ_{D-2} ran above).
      static_for<0, 2 * NH>([&](auto rc) {
        constexpr int R = decltype(rc)::value, j0 = 4 * R, li = j0 >> 3, m0 = j0 & 7;
        pn_begin<(ACTIVE ? 4 : 0)>(r);
        // the next interval: four more row blocks; behind the last one the next tile's first chunks (or, MODE_BWD, its transpose)
        pn_fetch(r, R + 1 < 2 * NH ? 4 : (MODE == MODE_BWD ? 1 : (E >= 64 ? 4 : 2)));
        if (ACTIVE) {
          block_pair(
              r.p, hb[li & 1],
              [&](auto tc) {
                if constexpr (j0 > 0) bwd_slice<8 + j0 - 1, (m0 == 0 ? 2 : 1), decltype(tc)::value>(acc2[1]);
              },
              [&]() {}, [&](auto tc) { bwd_slice<8 + j0, 1, decltype(tc)::value>(acc2[0]); });
          block_pair(
              r.p + 2, hb[li & 1], [&](auto tc) { bwd_slice<8 + j0 + 1, 1, decltype(tc)::value>(acc2[1]); }, [&]() {},
              [&](auto tc) { bwd_slice<8 + j0 + 2, 1, decltype(tc)::value>(acc2[0]); });
        }
        r.p += 4;
        if constexpr (m0 == 4) {
          INR_STAMP(si); ++si;
        }
      });
      if (ACTIVE) epi_bwd<NE - 1>(acc2[1]);  // the last epilogue: row block 7 of dZ_0
      INR_STAMP(si); ++si;
    }
  }
};

// ---- the kernel ---------------------------------------------------------------------------------------------------
// Workgroup = eight waves = two tile groups; group g = waves 4g .. 4g+3 works on stash tile tpw * (workgroup tile) + g;
// wave w owns coordinates [32 (w & 3), + 32) of it: lane (col, half).
template <int MODE, int NH>
__global__ __launch_bounds__(512, 2) void inr_siren_bf16_kernel(const NetDesc nd, const LossDesc ld, const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  constexpr int NW = PN_WAVES, D = NH + 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA destinations go through M0
  // MODE.FP16_OVFL: conversions to fp16 and bf8 saturate at the largest finite value instead of overflowing to infinity
  // (tools/probes/bf8_clamp_probe.hip) -- a dZ beyond the gradient scale's headroom is clipped, not turned into NaNs
#ifdef INR_STAMPS  // slot 60 / 61: the 100 MHz counter at kernel entry / exit of each wave (tools/stamps.py: prologue, drain)
  if (a.dbg != nullptr && lane == 0 && ((long long)blockIdx.x * NW + w) * 64 + 63 < a.dbg_cap)
    a.dbg[((long long)blockIdx.x * NW + w) * 64 + 60] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  if (MODE != MODE_FWD) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  float* bias_lds = reinterpret_cast<float*>(lds_raw + PN_SLOTS * W2_PANEL_BYTES);  // [D][256]
  float* encB_lds = bias_lds + D * 256;                                             // [E][4]
  float* red_lds = encB_lds + 4 * nd.E;                                             // [2][8]: loss partials, |dZ| maxima
  const int E = nd.E;
  for (int i = tid; i < D * 256; i += 64 * NW) bias_lds[i] = a.packed[nd.w2_bias_off + i];
  for (int i = tid; i < 4 * E; i += 64 * NW)  // rows padded to 16 bytes: one broadcast read per feature
    encB_lds[i] = (a.encB != nullptr && (i & 3) != 3) ? a.encB[3 * (i >> 2) + (i & 3)] : 0.f;
  PnRing r;
  r.gbase = reinterpret_cast<const char*>(a.packed + nd.w2_off);
  r.ring = lds_raw;
  r.first = MODE == MODE_BWD ? w2_n_fwd(D, E) : 0;
  r.len = MODE == MODE_FUSED ? w2_np(D, E) : (MODE == MODE_FWD ? w2_n_fwd(D, E) : w2_np(D, E) - w2_n_fwd(D, E));
  r.p = 0, r.req = 0, r.req_img = 0;
  r.w = w, r.lane = lane;
  r.rs = uniform_rsrc(r.gbase, 0x7ffffff0);
  r.voff = (2 * w) * 1024 + lane * 16;
  asm volatile("" : "+v"(r.voff));
  pn_fetch(r, MODE == MODE_BWD ? 1 : (E >= 64 ? 4 : 2));  // the first interval's panels
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // tables and panels in LDS

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
  if (w < 4 || tpw == 2) {
    SirenTile<MODE, NH, true> t(nd, ld, a, r, bias_lds, encB_lds, lane, w, mult);
    for (int wtile = blockIdx.x; wtile < n_wtiles; wtile += gridDim.x) t.run(tpw * wtile + (w >> 2));
    loss_acc = t.loss_acc, amax = t.amax;
  } else {
    SirenTile<MODE, NH, false> t(nd, ld, a, r, bias_lds, encB_lds, lane, w, mult);
    for (int wtile = blockIdx.x; wtile < n_wtiles; wtile += gridDim.x) t.run(0);
  }
  // every DMA this wave issued has landed before the workgroup (and its LDS) goes away
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE != MODE_FWD) {
    // the step's largest scaled |dZ|: non-negative floats order like their bit patterns
    // ONE atomic per workgroup: 2 048 waves updating the same word took 12 us of a 102 us launch (the updates of one
    // address are served one after the other at the memory side; profiles/r04_stamps_bf16_65536.txt)
    float m = amax;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) red_lds[NW + w] = m;
    __syncthreads();
    if (tid == 0 && st != nullptr) {
      float mw = 0.f;
      for (int i = 0; i < NW; ++i) mw = fmaxf(mw, red_lds[NW + i]);
      if (mw > 0.f) atomicMax(reinterpret_cast<unsigned*>(st) + 1, __builtin_bit_cast(unsigned, mw));
    }
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
#ifdef INR_STAMPS
  if (a.dbg != nullptr && lane == 0 && ((long long)blockIdx.x * NW + w) * 64 + 63 < a.dbg_cap)
    a.dbg[((long long)blockIdx.x * NW + w) * 64 + 61] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
}

template <int MODE, int NH>
inline hipError_t launch_siren_bf16_nh(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  const size_t lds_bytes = (size_t)PN_SLOTS * W2_PANEL_BYTES + ((size_t)nd.D * 256 + 4 * (size_t)nd.E + 2 * PN_WAVES) * sizeof(float);
  if (lds_bytes > 160 * 1024 || a.save_by_block) return hipErrorInvalidValue;
  if (MODE != MODE_FWD && (a.save == nullptr || a.dz_state == nullptr)) return hipErrorInvalidValue;
  if (MODE == MODE_FUSED && a.slabs == nullptr) return hipErrorInvalidValue;
  hipError_t e = allow_full_lds<inr_siren_bf16_kernel<MODE, NH>>();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((inr_siren_bf16_kernel<MODE, NH>), dim3(grid), dim3(64 * PN_WAVES), lds_bytes, st, nd, ld, a);
  return hipGetLastError();
}

// one instantiation per depth (3 .. 8 Linear layers): the layer loops are unrolled at compile time (see "vector loads the
// compiler does not see")
template <int MODE>
inline hipError_t launch_siren_bf16_mode(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
#ifdef INR_ONLY_NH  // experiment builds (tools/build_exp.sh): one depth
  return nd.D == INR_ONLY_NH + 2 ? launch_siren_bf16_nh<MODE, INR_ONLY_NH>(nd, ld, a, grid, st) : hipErrorInvalidValue;
#else
  switch (nd.D) {
    case 3: return launch_siren_bf16_nh<MODE, 1>(nd, ld, a, grid, st);
    case 4: return launch_siren_bf16_nh<MODE, 2>(nd, ld, a, grid, st);
    case 5: return launch_siren_bf16_nh<MODE, 3>(nd, ld, a, grid, st);
    case 6: return launch_siren_bf16_nh<MODE, 4>(nd, ld, a, grid, st);
    case 7: return launch_siren_bf16_nh<MODE, 5>(nd, ld, a, grid, st);
    case 8: return launch_siren_bf16_nh<MODE, 6>(nd, ld, a, grid, st);
    default: return hipErrorInvalidValue;
  }
#endif
}

}  // namespace inr
