// inr_dw_gemm_bf16.hip -- batch-level weight gradients of the bf16 throughput path:  dW_l = dZ_l^T h_{l-1},
// db_l = sum_c dZ_l  over all coordinates of a chunk of tiles, on v_mfma_f32_32x32x16_f16 (fp32 accumulate), split-K over
// chunks of tiles; inr_siren_bf16_impl.h leaves the operands behind.
//
// Why it exists: with the matrix pipe 16x faster than on the fp32 path a fused kernel that forms dW itself is paced by
// the per-tile gradient slabs it writes.  Here ~40 workgroups per layer each keep a 256 x 256 block of dW in registers
// across ~13 tiles: the slab stream is (chunks x 1.3 MB), written once in fp32.
//
// Operands (8-bit stash of inr_siren_bf16_kernel, inr_w2.h; per tile of TL coordinates, ROW-QUAD layout: rows 4q .. 4q+3
// of coordinate c share the dword at q * TL + c):  A = dZ_l as bf8 under the step's gradient scale -- a bf8 byte is the
// high byte of the fp16 of the same value, so "conversion" is the byte transposition the staging does anyway, and the
// product runs on the fp16 MFMA;  B = h_{l-1} = sin(2 pi phase / 256) from the stashed phase bytes (the fused kernel never
// stores h), or, for the first layer, the gauss encoder features regenerated from the tile's coordinates exactly as the
// forward pass formed them (sine of x.B_j revolutions, + 1/4 turn for the cosine half).  A thread stages one (row quad,
// 8 coordinates) item per operand and stage: two 16-byte loads, v_perm_b32 into four 16-byte LDS pieces (one per row);
// the row sums db come from the same registers (v_dot2c_f32_f16 against ones), reduced over the 8 threads of a quad at the
// end.  Sums are divided by the factor the fused kernel multiplied the loss gradient with (state word 2) on the way out;
// the first thread of the launch then derives the next step's power-of-two scale from the largest |dZ| the step saw.
//
// Workgroup = one 256 x 256 block of one layer's dW over one chunk of tiles; EIGHT waves 4 x 2 (two per SIMD), 2 x 4
// MFMA blocks each (128 accumulator registers): the staging of the next stage runs in one wave of a SIMD while the other
// multiplies.  K-steps of 64 coordinates staged through two LDS stages, rows pitched 144 B so that the 16-byte fragment
// reads of 16 consecutive rows hit 16 distinct 4-bank slots.
#include <hip/hip_runtime.h>

#include "inr_dw_gemm_bf16.h"
#include "inr_stamp_rt.h"
#include "inr_launch.h"
#include "inr_w2.h"

namespace inr {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GB_KS = 64;                 // coordinates per stage
constexpr int GB_PITCH = 72;              // LDS row pitch in 2-byte elements: 64 coordinates + 16 bytes of padding
constexpr int GB_TILE = 256 * GB_PITCH;   // one operand tile (elements)
constexpr int GB_STAGE = 2 * GB_TILE;     // A tile + B tile
constexpr int GB_NT = 512;                // threads: eight waves, two per SIMD (staging of one hides under MFMAs of the other)

#ifdef GB_STAMPS  // timing experiment: per-wave cycle sums of the parts of a stage (tools/gemm_stamps.py)
__device__ long long gb_stamps[512 * 8 * 8];
#define GB_T(k)                                                \
  do {                                                         \
    const long long now_ = (long long)__builtin_readcyclecounter(); \
    tsum[k] += now_ - tlast;                                   \
    tlast = now_;                                              \
  } while (0)
#else
#define GB_T(k) do {} while (0)
#endif

__device__ __forceinline__ unsigned pk_f16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// 8 dwords (coordinates c .. c+7, each = rows (2p, 2p+1) as fp16) -> the two rows as 8 fp16 each  (dZ_last)
__device__ __forceinline__ void split_rows(const u32x4& a, const u32x4& b, u32x4& even, u32x4& odd) {
  even[0] = __builtin_amdgcn_perm(a[1], a[0], 0x05040100u);
  even[1] = __builtin_amdgcn_perm(a[3], a[2], 0x05040100u);
  even[2] = __builtin_amdgcn_perm(b[1], b[0], 0x05040100u);
  even[3] = __builtin_amdgcn_perm(b[3], b[2], 0x05040100u);
  odd[0] = __builtin_amdgcn_perm(a[1], a[0], 0x07060302u);
  odd[1] = __builtin_amdgcn_perm(a[3], a[2], 0x07060302u);
  odd[2] = __builtin_amdgcn_perm(b[1], b[0], 0x07060302u);
  odd[3] = __builtin_amdgcn_perm(b[3], b[2], 0x07060302u);
}

// sum of 8 fp16 (4 dwords) into acc: v_dot2c_f32_f16 against a pair of ones (fp32 accumulate, one instruction per
// dword).  Inline assembly: through the builtin hipcc 7.2 selected the FIRST dword for all four dot products of an
// unrolled loop (round 2, bf16 form) -- the row sums came out as 4 x the first coordinate pair.
__device__ __forceinline__ float sum8(const u32x4& v, float acc, unsigned ones = 0x3c003c00u) {
#pragma unroll
  for (int i = 0; i < 4; ++i) asm("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(v[i]), "v"(ones));
  return acc;
}

__device__ __forceinline__ float dot2c(float acc, unsigned v, unsigned ones) {
  asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(v), "v"(ones));
  return acc;
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int I>
struct gb_ic {
  static constexpr int value = I;
};
template <int N, int I = 0, class F>
__device__ __forceinline__ void gb_for(F&& f) {
  if constexpr (I < N) {
    f(gb_ic<I>{});
    gb_for<N, I + 1>(f);
  }
}

// phase bytes R of two dwords (coordinates 2p, 2p+1 of row 4q + R) -> the fp16 pair of their sines.  In [4, 8) an fp16 ulp
// is 1/256: the bits 0x4400 | p ARE the value 4 + p / 256, and v_sin_f16 (argument in revolutions) of that is
// sin(2 pi p / 256), correctly rounded for all 256 phases (tools/probes/sinf16_probe.hip) -- no conversion, no multiply.
template <int R>
__device__ __forceinline__ unsigned sin_pair(unsigned lo, unsigned hi) {
  constexpr unsigned sel = (0x0cu << 24) | ((4u + R) << 16) | (0x0cu << 8) | (unsigned)R;
  const unsigned x = __builtin_amdgcn_perm(hi, lo, sel) | 0x44004400u;
  unsigned o;
  asm volatile("v_sin_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(o) : "v"(x));
  asm volatile("v_sin_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(o) : "v"(x));
  return o;
}
// encoder phase x . B_j (revolutions, reduced to [0, 1)) of one coordinate: the forward pass's fma chain
__device__ __forceinline__ float gauss_phase(const float* x, float b0, float b1, float b2) {
  return __builtin_amdgcn_fractf(fmaf(x[2], b2, fmaf(x[1], b1, fmaf(x[0], b0, 0.f))));
}

// ENC: B = encoder features (first layer); BIAS: this column block also produces db; LASTROWS: dZ has only its first
// four rows (the out_features <= 4 rows of the last layer, fp16 row pairs): the rest of the A tile stays zero.
//
// The stage loop is ONE instruction stream per wave, the same for all eight: stage s multiplies out of LDS stage s & 1
// while the operands of stage s+1 -- fetched two stages earlier into one of two register sets -- are converted and stored
// into the other LDS stage; one barrier per stage.  The stream is cut into 32 slots of one MFMA + its share of everything
// else (a fragment read for the next K = 16 group, a slice of the staging arithmetic: ~14 cycles of vector issue against
// the MFMA's 32 on the pipe), fenced so that the compiler keeps the interleaving.  Why not "one wave of a SIMD stages
// while the other multiplies" (rounds 2-3a): measured, the staging wave made no progress beside the multiplying one --
// the launch took the SUM of its matrix, vector and LDS-store times (profiles/r03_dw_gemm_knockouts.txt).
// Rotation: the last K = 16 group of a stage is multiplied after the barrier, from fragments read before it, under the
// first fragment reads of the next stage -- no exposed LDS latency at the head of a stage.
// (Inlined: as real functions the variants would take their arguments through memory -- flat loads, whose waits are
// vmcnt(0) and drain the prefetch -- and buffer descriptors from memory cost waterfall loops.)
template <int TL, bool ENC, bool BIAS, bool LASTROWS>
__device__ __forceinline__ void dwgb_body(const DwGemmBf16Args& a, const DwGemmBf16Unit& it, int kc, int tiles_per_chunk,
                                          char* lds_raw) {
  _Float16* lds = reinterpret_cast<_Float16*>(lds_raw);
  float* xs_lds = reinterpret_cast<float*>(lds_raw + (size_t)2 * GB_STAGE * 2);  // [2 stages][64 coords][3]
  float* encB_lds = xs_lds + 2 * GB_KS * 3;                                      // [E][3]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int wm = w >> 1, wn = w & 1;  // wave tile: rows [64 wm, +64) x columns [128 wn, +128) = 2 x 4 MFMA blocks
  constexpr int KS_PER_TILE = TL / GB_KS;
  constexpr int NI = LASTROWS ? 1 : 2;   // row blocks of the wave tile that hold anything
  constexpr int NA = LASTROWS ? 4 : 2;   // 16-byte loads of an A item
  const int t0 = kc * tiles_per_chunk;
  int n_mine = a.n_tiles - t0;
  if (n_mine > tiles_per_chunk) n_mine = tiles_per_chunk;
  const int n_steps = (n_mine > 0 ? n_mine : 0) * KS_PER_TILE;  // even: two stages per tile

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (ENC)
    for (int i = t; i < 3 * a.E; i += GB_NT) encB_lds[i] = a.encB[i];
  if (LASTROWS) {  // rows the loader never writes must read as zeros (both stages; B tiles are always written whole)
    u32x4 z4 = {0u, 0u, 0u, 0u};
    for (int i = t; i < 2 * 256 * 9; i += GB_NT) {
      const int st = i / (256 * 9), rem = i - st * 256 * 9, row = rem / 9, piece = rem - row * 9;
      *reinterpret_cast<u32x4*>(lds + (size_t)st * GB_STAGE + row * GB_PITCH + 8 * piece) = z4;
    }
  }
  // loader: the thread's item = row quad t >> 3 (rows 4q .. 4q+3), coordinates 8 (t & 7) .. + 7 of the stage
  const int seg = t & 7, quad = t >> 3;
  const unsigned* sv = reinterpret_cast<const unsigned*>(a.save);
  const size_t tile_dwords = (size_t)a.save_floats_per_tile;
  u32x4 ra[2][NA], rb[2][2];  // two register sets of fetched operands
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // (through a buffer descriptor on the tile's wave-uniform base: the thread's part of an address is the byte offset of its
  // (row quad, segment), formed once; tensor and K-step offsets are scalar -- no 64-bit vector adds in the loop)
  static_assert(GB_KS == W2_HALF, "a stage is one half-tile block of the 8-bit tensors");
  const int voffA = (quad * W2_HALF + 8 * seg) * 4;
  const int voffB = ((it.n0 / 4 + quad) * W2_HALF + 8 * seg) * 4;
  const int voffL = 8 * seg * 4;  // LASTROWS: row pair p at p * TL dwords (fp16 pairs, whole-tile rows)
  auto fetch = [&](int s, u32x4 (&A)[NA], u32x4 (&B)[2]) {
    if (s >= n_steps) s = n_steps - 1;  // past the end: the last stage again (loaded, never multiplied)
    const int tile = t0 + s / KS_PER_TILE, c0 = (s % KS_PER_TILE) * GB_KS;
    const unsigned long long ba = reinterpret_cast<unsigned long long>(sv + (size_t)tile * tile_dwords);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ba), hi = __builtin_amdgcn_readfirstlane((unsigned)(ba >> 32));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7ffffff0, 0x00020000);
    // the stage's half-tile block of the 8-bit tensors; dZ_last (fp16 row pairs): its 64 coordinates of the tile's rows
    const int blk = (s % KS_PER_TILE) * (W2_TENSOR_DWORDS / 2);
    const int soA = (it.dz_off + (LASTROWS ? c0 : blk)) * 4, soB = (it.z_off + blk) * 4;
    if constexpr (LASTROWS) {
      if (quad == 0) {
        A[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffL, soA, 0));
        A[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffL, soA + 16, 0));
        A[2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffL, soA + TL * 4, 0));
        A[3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffL, soA + TL * 4 + 16, 0));
      }
    } else {
      A[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffA, soA, 0));
      A[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffA, soA + 16, 0));
    }
    if (!ENC) {
      B[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffB, soB, 0));
      B[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffB, soB + 16, 0));
    }
  };
  // the coordinates of stage s (first-layer units: the encoder features are regenerated from them): 64 coordinates =
  // 192 floats, one per thread t < 192, loaded a stage before they are put into LDS; rows past B read as coordinate 0
  auto xs_get = [&](int s) -> float {
    float v = 0.f;
    if (ENC && t < GB_KS * 3) {
      if (s >= n_steps) s = n_steps - 1;
      const int tile = t0 + s / KS_PER_TILE, c0 = (s % KS_PER_TILE) * GB_KS;
      const long long r = (long long)tile * TL + c0 + t / 3;
      if (r < a.B) v = a.coords[3 * r + (t % 3)];
    }
    return v;
  };
  auto xs_put = [&](int s, float v) {
    if (ENC && t < GB_KS * 3) xs_lds[(s & 1) * GB_KS * 3 + t] = v;
  };

  // ---- the staging of one stage, in 32 slices (slice k runs in slot k of the stage before) ---------------------------
  // even k: pair (k/2) % 4 of B row k / 8 (coordinates 2p, 2p+1), the row stored after its fourth pair;
  // odd k : piece (k/2) % 4 of A row k / 8: two byte permutes, two more + the store, then the row sum in two halves.
  // Sums of a stage past the end (loaded, never multiplied) are kept out with a zero "ones" vector: no branch in a stage.
  u32x4 rowA = {0u, 0u, 0u, 0u}, rowB = {0u, 0u, 0u, 0u}, rowC = {0u, 0u, 0u, 0u};
  float eb0 = 0.f, eb1 = 0.f, eb2 = 0.f, eph0 = 0.f, eph1 = 0.f, xq[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto slice = [&](auto K, int s, unsigned ones, const u32x4 (&A)[NA], const u32x4 (&B)[2]) {
    constexpr int k = decltype(K)::value, R = k / 8, pc = (k / 2) % 4;
    _Float16* sa = lds + (size_t)(s & 1) * GB_STAGE + (4 * quad) * GB_PITCH + 8 * seg;
    _Float16* sb = sa + GB_TILE;
    if constexpr (k % 2 == 0) {
      if constexpr (ENC) {
        // first-layer units: the thread's four rows are the sine (rows 0, 1) and cosine (rows 2, 3) features of encoder
        // frequencies n0 + 2 quad + (0, 1) -- one phase chain serves a sine and a cosine (v_cos_f32 of the sine's phase;
        // the forward pass starts the cosine's chain at 1/4 turn instead: the same value to an ulp of the phase).
        // Even slot e = k / 2: frequency f = e / 8, coordinate pair (e / 2) % 4; first the two phases, next slot the four
        // features; a frequency's two rows are stored after its fourth pair.
        constexpr int e = k / 2, f = e / 8, cp = (e / 2) % 4;
        const float* xs = xs_lds + (s & 1) * GB_KS * 3 + 24 * seg;
        // (the six coordinates and the frequency's row of the encoder matrix are read from LDS one slot before the chain
        // that takes them -- at the head of a stage there is no such slot: they were written across the barrier)
        auto get = [&](auto CP, auto F) {
          constexpr int ncp = decltype(CP)::value, nf = decltype(F)::value;
#pragma unroll
          for (int i = 0; i < 6; ++i) xq[i] = xs[6 * ncp + i];
          if constexpr (ncp == 0) {
            int j = it.n0 + 2 * quad + nf;
            j = j < a.E ? j : a.E - 1;  // frequencies past E (enc_size not a multiple of 128) are computed and never stored
            eb0 = encB_lds[3 * j + 0], eb1 = encB_lds[3 * j + 1], eb2 = encB_lds[3 * j + 2];
          }
        };
        if constexpr (e % 2 == 0) {
          if constexpr (e == 0) get(gb_ic<0>{}, gb_ic<0>{});
          eph0 = gauss_phase(xq, eb0, eb1, eb2), eph1 = gauss_phase(xq + 3, eb0, eb1, eb2);
        } else {
          rowB[cp] = pk_f16(__builtin_amdgcn_sinf(eph0), __builtin_amdgcn_sinf(eph1));
          rowC[cp] = pk_f16(__builtin_amdgcn_cosf(eph0), __builtin_amdgcn_cosf(eph1));
          if constexpr (cp == 3) {
            *reinterpret_cast<u32x4*>(sb + f * GB_PITCH) = rowB;
            *reinterpret_cast<u32x4*>(sb + (2 + f) * GB_PITCH) = rowC;
          }
          if constexpr (e < 15) get(gb_ic<((e + 1) / 2) % 4>{}, gb_ic<(e + 1) / 8>{});
        }
      } else {
        rowB[pc] = sin_pair<R>(B[pc >> 1][2 * (pc & 1)], B[pc >> 1][2 * (pc & 1) + 1]);
        if constexpr (pc == 3) *reinterpret_cast<u32x4*>(sb + R * GB_PITCH) = rowB;
      }
    } else if constexpr (LASTROWS) {
      if constexpr (k == 1) {  // the four rows of dZ_last: row quad 0's threads only
        if (quad == 0) {
          u32x4 row[4];
          split_rows(A[0], A[1], row[0], row[1]);
          split_rows(A[2], A[3], row[2], row[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            *reinterpret_cast<u32x4*>(sa + r * GB_PITCH) = row[r];
            bsum[r] = sum8(row[r], bsum[r], ones);
          }
        }
      }
    } else {
      constexpr unsigned sel = ((4u + R) << 24) | (0x0cu << 16) | ((unsigned)R << 8) | 0x0cu;  // bf8 byte -> fp16 high byte
      if constexpr (pc == 0) {
        rowA[0] = __builtin_amdgcn_perm(A[0][1], A[0][0], sel);
        rowA[1] = __builtin_amdgcn_perm(A[0][3], A[0][2], sel);
      } else if constexpr (pc == 1) {
        rowA[2] = __builtin_amdgcn_perm(A[1][1], A[1][0], sel);
        rowA[3] = __builtin_amdgcn_perm(A[1][3], A[1][2], sel);
        *reinterpret_cast<u32x4*>(sa + R * GB_PITCH) = rowA;
      } else if constexpr (BIAS) {
        constexpr int h = 2 * (pc - 2);
        bsum[R] = dot2c(dot2c(bsum[R], rowA[h], ones), rowA[h + 1], ones);
      }
    }
  };

  const _Float16* As = lds + (wm * 64 + li) * GB_PITCH + 8 * half;
  const _Float16* Bs = lds + GB_TILE + (wn * 128 + li) * GB_PITCH + 8 * half;
  f16x8 fa[2][NI], fb[2][4];  // MFMA fragments of two K = 16 groups
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) fa[b][i][e] = (_Float16)0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) fb[b][j][e] = (_Float16)0.f;
  }
  // one MFMA of K = 16 group g (fragment buffer (g + 1) & 1); m = its place among the group's 8 slots
  auto mma = [&](auto G, auto Mi) {
    constexpr int g = decltype(G)::value, m = decltype(Mi)::value, cb = (g + 1) & 1;
    if constexpr (NI == 2) {
      constexpr int i = m / 4, j = m % 4;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cb][i], fb[cb][j], acc[i][j], 0, 0, 0);
    } else if constexpr (m % 2 == 0) {
      acc[0][m / 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cb][0], fb[cb][m / 2], acc[0][m / 2], 0, 0, 0);
    }
  };
  // Stage s: slot k = 8 g + m.  Group 0 multiplies the last K = 16 group of stage s-1 (fragment buffer 1: zeros before
  // stage 0), groups 1 .. 3 the groups 0 .. 2 of stage s; group g's first slots read the fragments of (s, group g) into
  // buffer g & 1.  Slice k of the staging of stage s+1 (register set NA / NB) rides along; then the set is refilled.
  auto stage = [&](int s, u32x4 (&SA)[NA], u32x4 (&SB)[2]) {
    const _Float16* Ab = As + (size_t)(s & 1) * GB_STAGE;
    const _Float16* Bb = Bs + (size_t)(s & 1) * GB_STAGE;
    unsigned ones = s + 1 < n_steps ? 0x3c003c00u : 0u;
    asm volatile("" : "+v"(ones));  // (opaque: a select folded into the sums' asm operands becomes a branch)
    gb_for<32>([&](auto K) {
      constexpr int k = decltype(K)::value, g = k / 8, m = k % 8, rbuf = g & 1;
      if constexpr (m < NI)
        fa[rbuf][m] = *reinterpret_cast<const f16x8*>(Ab + m * 32 * GB_PITCH + 16 * g);
      else if constexpr (m < NI + 4)
        fb[rbuf][m - NI] = *reinterpret_cast<const f16x8*>(Bb + (m - NI) * 32 * GB_PITCH + 16 * g);
      mma(gb_ic<g>{}, gb_ic<m>{});
      slice(K, s + 1, ones, SA, SB);
      __builtin_amdgcn_sched_barrier(0);
    });
    fetch(s + 3, SA, SB);
  };

  float xs_next = 0.f;
  if (n_steps > 0) {
    fetch(0, ra[0], rb[0]);
    xs_put(0, xs_get(0));
    xs_next = xs_get(1);
    fetch(1, ra[1], rb[1]);
    if (ENC || LASTROWS) __syncthreads();  // xs of stage 0, the encoder matrix, the zeroed A tiles
    gb_for<32>([&](auto K) { slice(K, 0, 0x3c003c00u, ra[0], rb[0]); });
    fetch(2, ra[0], rb[0]);
    xs_put(1, xs_next);
    xs_next = xs_get(2);
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < n_steps; s += 2) {
    stage(s, ra[1], rb[1]);
    xs_put(s + 2, xs_next);  // xs buffer s & 1: last read while stage s was staged, one barrier ago
    xs_next = xs_get(s + 3);
    __syncthreads();
    stage(s + 1, ra[0], rb[0]);
    xs_put(s + 3, xs_next);
    xs_next = xs_get(s + 4);
    __syncthreads();
  }
  gb_for<8>([&](auto Mi) { mma(gb_ic<0>{}, Mi); });  // the last stage's last K = 16 group
  // ---- chunk slab: dW rows follow the MFMA C layout (register r of lane (li, half): row (r&3)+8(r>>2)+4 half, col li);
  // the gradient scale comes off here
  const float unscale = a.dz_state != nullptr ? 1.0f / a.dz_state[2] : 1.0f;
  float* slab = a.slabs + (size_t)kc * a.slab_floats;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int rb0 = 32 * (2 * wm + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // column of dW behind B-tile row rr; first-layer units: rows (4 qd + 0, 1) = sine, (4 qd + 2, 3) = cosine features
      // of frequencies n0 + 2 qd + (0, 1)
      const int rr = 32 * (4 * wn + j) + li;
      const int freq = it.n0 + 2 * (rr >> 2) + (rr & 1);
      const int col = ENC ? (freq < a.E ? ((rr & 2) ? a.E : 0) + freq : it.K) : it.n0 + rr;
      if (col < it.K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rb0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < it.M) slab[it.gw_off + (size_t)row * it.K + col] = acc[i][j][r] * unscale;
        }
      }
    }
  }
  if (BIAS) {  // the 8 threads of a row quad (consecutive lanes) each hold the sums of their 8-coordinate segments
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = bsum[r];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      const int row = 4 * quad + r;
      if (seg == 0 && row < it.M) slab[it.gb_off + row] = v * unscale;
    }
  }
}

template <int TL>
__global__ __launch_bounds__(GB_NT, 2) void dw_gemm_bf16_kernel(const DwGemmBf16Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  INR_RT_STAMP(a.dbg, a.dbg_cap, GB_NT / 64, threadIdx.x >> 6, threadIdx.x & 63, 44);
  int b = blockIdx.x, kc, u, tpc;
  const int enc_wgs = a.n_enc_units * a.n_chunks_enc;
  if (b < enc_wgs) {
    kc = b / a.n_enc_units, u = b - kc * a.n_enc_units, tpc = a.tiles_per_chunk_enc;
  } else {
    b -= enc_wgs;
    const int others = a.n_units - a.n_enc_units;
    kc = b / others, u = a.n_enc_units + b - kc * others, tpc = a.tiles_per_chunk;
  }
  const DwGemmBf16Unit& it = a.unit[u];
  if (it.z_off < 0) {
    if (it.n0 == 0)
      dwgb_body<TL, true, true, false>(a, it, kc, tpc, lds_raw);
    else
      dwgb_body<TL, true, false, false>(a, it, kc, tpc, lds_raw);
  } else if (it.M <= 4) {
    dwgb_body<TL, false, true, true>(a, it, kc, tpc, lds_raw);  // last layer: out_features <= 4 rows of dZ
  } else {
    dwgb_body<TL, false, true, false>(a, it, kc, tpc, lds_raw);  // hidden layers: one column tile (K <= 256), always with db
  }
  // the next step's gradient scale (inr_w2.h): nothing in this launch reads words 0 and 1
  if (a.dz_state != nullptr && blockIdx.x == 0 && threadIdx.x == 0) dz_state_roll(a.dz_state, a.dz_count);
  INR_RT_STAMP(a.dbg, a.dbg_cap, GB_NT / 64, threadIdx.x >> 6, threadIdx.x & 63, 45);
}

__global__ void dz_roll_kernel(float* st, float* cnt) { dz_state_roll(st, cnt); }

hipError_t launch_dz_roll(float* st, float* cnt, hipStream_t stream) {
  hipLaunchKernelGGL(dz_roll_kernel, dim3(1), dim3(1), 0, stream, st, cnt);
  return hipGetLastError();
}

#ifdef GB_STAMPS
extern "C" int inr_debug_gemm_stamps(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gb_stamps), sizeof(long long) * 512 * 8 * 8);
}
#endif

#ifdef INR_STAMPS
extern long long* g_stamp_buf;  // inr_api.hip
extern long long g_stamp_cap;
#endif

hipError_t launch_dw_gemm_bf16(const DwGemmBf16Args& a_in, hipStream_t st) {
  DwGemmBf16Args a = a_in;
#ifdef INR_STAMPS
  a.dbg = g_stamp_buf, a.dbg_cap = g_stamp_cap;
#endif
  if (a.n_units <= 0 || a.n_units > INR_DWGB_MAX_UNITS || a.n_chunks <= 0 || a.tiles_per_chunk <= 0 || a.TL != 128 ||
      a.E > 1024 || a.n_enc_units < 0 || a.n_enc_units >= a.n_units ||
      (a.n_enc_units > 0 && (a.n_chunks_enc <= 0 || a.tiles_per_chunk_enc <= 0)))
    return hipErrorInvalidValue;
  for (int u = 0; u < a.n_units; ++u)  // the first-layer units come first
    if ((a.unit[u].z_off < 0) != (u < a.n_enc_units)) return hipErrorInvalidValue;
  const size_t lds_bytes = (size_t)2 * GB_STAGE * 2 + (size_t)(2 * GB_KS * 3 + 3 * a.E) * sizeof(float);
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = allow_full_lds<dw_gemm_bf16_kernel<128>>();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(dw_gemm_bf16_kernel<128>, dim3((unsigned)(a.n_enc_units * a.n_chunks_enc + (a.n_units - a.n_enc_units) * a.n_chunks)), dim3(GB_NT), lds_bytes, st, a);
  return hipGetLastError();
}

}  // namespace inr
