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

// 8 dwords (coordinates c .. c+7, byte R of each = row 4q + R as bf8) -> row R as 8 fp16: byte R of coordinates (2i, 2i+1)
// into the HIGH bytes of the two halves of dword i, zeros below (selector 0x0c)
template <int R>
__device__ __forceinline__ u32x4 bf8_row(const u32x4& a, const u32x4& b) {
  constexpr unsigned sel = ((4u + R) << 24) | (0x0cu << 16) | ((unsigned)R << 8) | 0x0cu;
  u32x4 o;
  o[0] = __builtin_amdgcn_perm(a[1], a[0], sel);
  o[1] = __builtin_amdgcn_perm(a[3], a[2], sel);
  o[2] = __builtin_amdgcn_perm(b[1], b[0], sel);
  o[3] = __builtin_amdgcn_perm(b[3], b[2], sel);
  return o;
}

// sum of 8 fp16 (4 dwords) into acc: v_dot2c_f32_f16 against a pair of ones (fp32 accumulate, one instruction per
// dword).  Inline assembly: through the builtin hipcc 7.2 selected the FIRST dword for all four dot products of an
// unrolled loop (round 2, bf16 form) -- the row sums came out as 4 x the first coordinate pair.
__device__ __forceinline__ float sum8(const u32x4& v, float acc, unsigned ones = 0x3c003c00u) {
#pragma unroll
  for (int i = 0; i < 4; ++i) asm("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(v[i]), "v"(ones));
  return acc;
}

// 8 dwords (byte R of each = phase of row 4q + R at coordinates c .. c+7) -> sin(2 pi phase / 256) of that row as 8 fp16
template <int R>
__device__ __forceinline__ u32x4 sin_row(const u32x4& a, const u32x4& b) {
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned d = i < 4 ? a[i] : b[i - 4];
#ifdef GB_EXP_NOSIN
    s[i] = (float)((d >> (8 * R)) & 255u) * 0.00390625f;
#else
    s[i] = __builtin_amdgcn_sinf((float)((d >> (8 * R)) & 255u) * 0.00390625f);
#endif
  }
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pk_f16(s[2 * i], s[2 * i + 1]);
  return o;
}

// feature `row` (< E: sine, >= E: cosine of the same phase) of 8 consecutive coordinates whose (x0,x1,x2) sit in xs
__device__ __forceinline__ u32x4 gauss_features(const float* xs, const float* encB, int E, int row) {
  int s = row < E ? row : row - E;
  s = s < E ? s : E - 1;  // rows past 2E (column blocks wider than the layer) are computed and never stored
  const float quarter = row < E ? 0.f : 0.25f;
  const float b0 = encB[3 * s + 0], b1 = encB[3 * s + 1], b2 = encB[3 * s + 2];
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    f[j] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(fmaf(xs[3 * j + 2], b2, fmaf(xs[3 * j + 1], b1, fmaf(xs[3 * j], b0, quarter)))));
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pk_f16(f[2 * i], f[2 * i + 1]);
  return o;
}

// ENC: B = encoder features (first layer); BIAS: this column block also produces db; LASTROWS: dZ has only its first
// four rows (the out_features <= 4 rows of the last layer, fp16 row pairs): the rest of the A tile stays zero
// NSETS: register sets of fetched operands -- a stage's operands are requested NSETS stages before they are staged.  The
// hidden-layer units afford a third set (128 accumulator registers + 3 x 16); the first-layer units, which also hold
// encoder arithmetic, keep two.  (Inlined: as real functions the variants would take their arguments through memory --
// flat loads, whose waits are vmcnt(0) and drain the prefetch -- and buffer descriptors from memory cost waterfall loops.)
template <int TL, bool ENC, bool BIAS, bool LASTROWS, int NSETS>
__device__ __forceinline__ void dwgb_body(const DwGemmBf16Args& a, const DwGemmBf16Unit& it, int kc, char* lds_raw) {
  _Float16* lds = reinterpret_cast<_Float16*>(lds_raw);
  float* xs_lds = reinterpret_cast<float*>(lds_raw + (size_t)2 * GB_STAGE * 2);  // [2 stages][64 coords][3]
  float* encB_lds = xs_lds + 2 * GB_KS * 3;                                      // [E][3]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int wm = w >> 1, wn = w & 1;  // wave tile: rows [64 wm, +64) x columns [128 wn, +128) = 2 x 4 MFMA blocks
  constexpr int KS_PER_TILE = TL / GB_KS;
  const int t0 = kc * a.tiles_per_chunk;
  int n_mine = a.n_tiles - t0;
  if (n_mine > a.tiles_per_chunk) n_mine = a.tiles_per_chunk;
  const int n_steps = (n_mine > 0 ? n_mine : 0) * KS_PER_TILE;

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
#ifdef GB_EXP_QUADSWAP
  // (16 consecutive lanes hold row quads Q and Q+2: their rows lie 8 x 144 B = 32 banks apart -- with Q and Q+1, 16 banks
  // apart, the two 128-byte row pieces of a 16-lane group of a ds_write_b128 share 16 banks)
  const int seg = t & 7, quad = ((t >> 3) & ~3) | (((t >> 3) & 1) << 1) | ((t >> 4) & 1);
#else
  const int seg = t & 7, quad = t >> 3;
#endif
  const unsigned* sv = reinterpret_cast<const unsigned*>(a.save);
  const size_t tile_dwords = (size_t)a.save_floats_per_tile;
  // NSETS register sets: while stage s multiplies, the set of stage s+1 -- fetched NSETS stages earlier -- is staged into
  // LDS and then refilled with stage s+1+NSETS (global latency is several microseconds under load, a stage 1.5-3: with two
  // sets the launch ran at the pace of its loads -- a loads -> LDS -> barrier skeleton took three quarters of its time)
  u32x4 ra[NSETS][LASTROWS ? 4 : 2], rb[NSETS][2];
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // (through a buffer descriptor on the tile's wave-uniform base: the thread's part of an address is the byte offset of its
  // (row quad, segment), formed once; tensor and K-step offsets are scalar -- no 64-bit vector adds in the loop)
  static_assert(GB_KS == W2_HALF, "a stage is one half-tile block of the 8-bit tensors");
  const int voffA = (quad * W2_HALF + 8 * seg) * 4;
  const int voffB = ((it.n0 / 4 + quad) * W2_HALF + 8 * seg) * 4;
  const int voffL = 8 * seg * 4;  // LASTROWS: row pair p at p * TL dwords (fp16 pairs, whole-tile rows)
  auto fetch = [&](int s, u32x4 (&A)[LASTROWS ? 4 : 2], u32x4 (&B)[2]) {
#ifdef GB_EXP_NOLOAD  // timing experiment: no global loads (the sets keep whatever they hold)
    if (s > 2) {
      asm volatile("" : "+v"(A[0][0]), "+v"(A[1][0]), "+v"(B[0][0]), "+v"(B[1][0]));
      return;
    }
#endif
    if (s >= n_steps) s = n_steps - 1;  // past the end: the last stage again (loaded, never multiplied)
    const int tile = t0 + s / KS_PER_TILE, c0 = (s % KS_PER_TILE) * GB_KS;
    const unsigned long long ba = reinterpret_cast<unsigned long long>(sv + (size_t)tile * tile_dwords);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ba), hi = __builtin_amdgcn_readfirstlane((unsigned)(ba >> 32));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7ffffff0, 0x00020000);
    // the stage's half-tile block of the 8-bit tensors; dZ_last (fp16 row pairs): its 64 coordinates of the tile's rows
    const int blk = (s % KS_PER_TILE) * (W2_TENSOR_DWORDS / 2);
    const int soA = (it.dz_off + (LASTROWS ? c0 : blk)) * 4, soB = (it.z_off + blk) * 4;
    if (LASTROWS) {
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
  // staging of the A item of stage s: rows as fp16 (and into the row sums, once: `count`)
  auto stage_a = [&](int s, bool count, const u32x4 (&A)[LASTROWS ? 4 : 2]) {
    _Float16* st = lds + (size_t)(s & 1) * GB_STAGE + (4 * quad) * GB_PITCH + 8 * seg;
    u32x4 row[4];
#ifdef GB_EXP_INTERLEAVE
    unsigned ones = count ? 0x3c003c00u : 0u;  // (branch-free row sums: one scheduling region per stage)
    asm volatile("" : "+v"(ones));
#endif
#ifdef GB_EXP_NOSTAGEVALU
    if (!LASTROWS) {
      row[0] = A[0], row[1] = A[1], row[2] = A[0], row[3] = A[1];
    } else
#endif
    if (LASTROWS) {
      if (quad != 0) return;
      split_rows(A[0], A[1], row[0], row[1]);
      split_rows(A[2], A[3], row[2], row[3]);
    } else {
      row[0] = bf8_row<0>(A[0], A[1]);
      row[1] = bf8_row<1>(A[0], A[1]);
      row[2] = bf8_row<2>(A[0], A[1]);
      row[3] = bf8_row<3>(A[0], A[1]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#ifdef GB_EXP_NOLDSWRITE
      asm volatile("" ::"v"(row[r][0]), "v"(row[r][1]), "v"(row[r][2]), "v"(row[r][3]));
#else
      *reinterpret_cast<u32x4*>(st + r * GB_PITCH) = row[r];
#endif
#ifdef GB_EXP_INTERLEAVE
      if (BIAS) bsum[r] = sum8(row[r], bsum[r], ones);
#else
      if (BIAS && count) bsum[r] = sum8(row[r], bsum[r]);
#endif
    }
  };
  // ... and of the B item: rows through the sine (or the encoder)
  auto stage_b = [&](int s, const u32x4 (&B)[2]) {
    _Float16* st = lds + (size_t)(s & 1) * GB_STAGE + GB_TILE + (4 * quad) * GB_PITCH + 8 * seg;
    u32x4 row[4];
#ifdef GB_EXP_NOSTAGEVALU
    if (true) {
      row[0] = B[0], row[1] = B[1], row[2] = B[0], row[3] = B[1];
      if (ENC) row[0] = row[1] = row[2] = row[3] = u32x4{1u, 2u, 3u, (unsigned)seg};
    } else
#endif
    if (ENC) {
      const float* xs = xs_lds + (s & 1) * GB_KS * 3 + 24 * seg;
#pragma unroll
      for (int r = 0; r < 4; ++r) row[r] = gauss_features(xs, encB_lds, a.E, it.n0 + 4 * quad + r);
    } else {
      row[0] = sin_row<0>(B[0], B[1]);
      row[1] = sin_row<1>(B[0], B[1]);
      row[2] = sin_row<2>(B[0], B[1]);
      row[3] = sin_row<3>(B[0], B[1]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#ifdef GB_EXP_NOLDSWRITE
      asm volatile("" ::"v"(row[r][0]), "v"(row[r][1]), "v"(row[r][2]), "v"(row[r][3]));
#else
      *reinterpret_cast<u32x4*>(st + r * GB_PITCH) = row[r];
#endif
    }
  };
  const _Float16* As = lds + (wm * 64 + li) * GB_PITCH + 8 * half;
  const _Float16* Bs = lds + GB_TILE + (wn * 128 + li) * GB_PITCH + 8 * half;
  // stage s multiplies out of LDS stage s & 1 while register set `N` (stage s+1) is staged into the other one and then
  // refilled with stage s+3.  The two waves of a SIMD (w, w + 4) run the same program between the same barriers; left in
  // lockstep both stage at the same time -- ~150 vector instructions, the matrix pipe idle -- and then both multiply.  So
  // they take the two halves in opposite order: waves 0-3 multiply first and stage behind it, waves 4-7 stage first.  (The
  // refill stays behind both for every wave: issued right behind the staging of the stage-first waves it measured 14 %
  // slower -- their loads then compete with the other waves' at the head of every stage.)
#ifdef GB_EXP_NOSTAGGER
  const bool stage_first = false;
#else
  const bool stage_first = w >= 4;
#endif
#if defined(GB_EXP_PRIO) && GB_EXP_PRIO == 1
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef GB_STAMPS
  long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter();
#endif
  auto compute = [&](int s, u32x4 (&NA)[LASTROWS ? 4 : 2], u32x4 (&NB)[2]) {
    const bool more = s + 1 < n_steps;
    const _Float16* Ab = As + (size_t)(s & 1) * GB_STAGE;
    const _Float16* Bb = Bs + (size_t)(s & 1) * GB_STAGE;
#ifdef GB_EXP_INTERLEAVE
    // one scheduling region: the stage's 32 MFMAs with the next stage's staging spread between them
    stage_a(s + 1, more, NA);
    stage_b(s + 1, NB);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f16x8 A[2], B[4];
#pragma unroll
      for (int i = 0; i < (LASTROWS ? 1 : 2); ++i) A[i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * GB_PITCH + 16 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) B[j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * GB_PITCH + 16 * q);
#pragma unroll
      for (int i = 0; i < (LASTROWS ? 1 : 2); ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);  // 6 VALU
      if (i < 24) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 LDS read
      if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 LDS write
    }
#else
    GB_T(0);
    if (stage_first) {
#if defined(GB_EXP_PRIO) && GB_EXP_PRIO == 2
      __builtin_amdgcn_s_setprio(1);
#endif
      stage_a(s + 1, more, NA);
      stage_b(s + 1, NB);
#if defined(GB_EXP_PRIO) && GB_EXP_PRIO == 2
      __builtin_amdgcn_s_setprio(0);
#endif
      __builtin_amdgcn_sched_barrier(0);
      GB_T(1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // four K = 16 sub-steps of the stage's 64 coordinates
      f16x8 A[2], B[4];
#pragma unroll
      for (int i = 0; i < (LASTROWS ? 1 : 2); ++i) A[i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * GB_PITCH + 16 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) B[j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * GB_PITCH + 16 * q);
#pragma unroll
      for (int i = 0; i < (LASTROWS ? 1 : 2); ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#ifdef GB_EXP_NOMFMA
          acc[i][j][q] += (float)A[i][0] + (float)B[j][0];
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[j], acc[i][j], 0, 0, 0);
#endif
    }
    GB_T(2);
    if (!stage_first) {
      __builtin_amdgcn_sched_barrier(0);
#if defined(GB_EXP_PRIO) && GB_EXP_PRIO == 2
      __builtin_amdgcn_s_setprio(1);
#endif
      stage_a(s + 1, more, NA);
      stage_b(s + 1, NB);
#if defined(GB_EXP_PRIO) && GB_EXP_PRIO == 2
      __builtin_amdgcn_s_setprio(0);
#endif
      GB_T(1);
    }
#endif
    fetch(s + 1 + NSETS, NA, NB);
    GB_T(3);
  };

  float xs_next = 0.f;
  if (n_steps > 0) {  // n_steps is even (two stages per tile)
    fetch(0, ra[0], rb[0]);
    xs_put(0, xs_get(0));
    xs_next = xs_get(1);
#pragma unroll
    for (int k = 1; k < NSETS; ++k) fetch(k, ra[k], rb[k]);
    if (ENC || LASTROWS) __syncthreads();  // xs of stage 0, the encoder matrix, the zeroed A tiles
    stage_a(0, true, ra[0]);
    stage_b(0, rb[0]);
    fetch(NSETS, ra[0], rb[0]);
    xs_put(1, xs_next);
    xs_next = xs_get(2);
  }
  __syncthreads();
  // stage k multiplies while stage k+1 goes from set (k+1) % NSETS into LDS; compute() then refills that set
  auto step = [&](int k, u32x4 (&NA)[LASTROWS ? 4 : 2], u32x4 (&NB)[2]) {
    compute(k, NA, NB);
    xs_put(k + 2, xs_next);  // xs buffer k & 1: last read while stage k was staged, one barrier ago
    xs_next = xs_get(k + 3);
    __syncthreads();
    GB_T(4);
  };
  if (NSETS == 2) {
#pragma unroll 1
    for (int s = 0; s < n_steps; s += 2) {
      step(s, ra[1], rb[1]);
      step(s + 1, ra[0], rb[0]);
    }
  } else {
#pragma unroll 1
    for (int s = 0; s < n_steps; s += 3) {
      step(s, ra[1 % NSETS], rb[1 % NSETS]);
      if (s + 1 >= n_steps) break;
      step(s + 1, ra[2 % NSETS], rb[2 % NSETS]);
      if (s + 2 >= n_steps) break;
      step(s + 2, ra[0], rb[0]);
    }
  }
#ifdef GB_STAMPS
  GB_T(0);
#endif
  // ---- chunk slab: dW rows follow the MFMA C layout (register r of lane (li, half): row (r&3)+8(r>>2)+4 half, col li);
  // the gradient scale comes off here
  const float unscale = a.dz_state != nullptr ? 1.0f / a.dz_state[2] : 1.0f;
  float* slab = a.slabs + (size_t)kc * a.slab_floats;
#pragma unroll
  for (int i = 0; i < (LASTROWS ? 1 : 2); ++i) {
    const int rb0 = 32 * (2 * wm + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = it.n0 + 32 * (4 * wn + j) + li;
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
#ifdef GB_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  GB_T(5);
  if (lane == 0 && blockIdx.x < 512) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    tsum[6] = id, tsum[7] = n_steps;
    for (int k = 0; k < 8; ++k) gb_stamps[(blockIdx.x * 8 + w) * 8 + k] = tsum[k];
  }
#endif
}

template <int TL>
__global__ __launch_bounds__(GB_NT, 2) void dw_gemm_bf16_kernel(const DwGemmBf16Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int kc = blockIdx.x / a.n_units;
  const DwGemmBf16Unit& it = a.unit[blockIdx.x - kc * a.n_units];
#ifdef GB_EXP_NOENC  // timing experiment: the first-layer units leave at once
  if (it.z_off < 0) return;
#endif
  if (it.z_off < 0) {
    if (it.n0 == 0)
      dwgb_body<TL, true, true, false, 2>(a, it, kc, lds_raw);
    else
      dwgb_body<TL, true, false, false, 2>(a, it, kc, lds_raw);
  } else if (it.M <= 4) {
    dwgb_body<TL, false, true, true, 3>(a, it, kc, lds_raw);  // last layer: out_features <= 4 rows of dZ
  } else {
    dwgb_body<TL, false, true, false, 3>(a, it, kc, lds_raw);  // hidden layers: one column tile (K <= 256), always with db
  }
  // the next step's gradient scale (inr_w2.h): nothing in this launch reads words 0 and 1
  if (a.dz_state != nullptr && blockIdx.x == 0 && threadIdx.x == 0) dz_state_roll(a.dz_state);
}

__global__ void dz_roll_kernel(float* st) { dz_state_roll(st); }

hipError_t launch_dz_roll(float* st, hipStream_t stream) {
  hipLaunchKernelGGL(dz_roll_kernel, dim3(1), dim3(1), 0, stream, st);
  return hipGetLastError();
}

#ifdef GB_STAMPS
extern "C" int inr_debug_gemm_stamps(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gb_stamps), sizeof(long long) * 512 * 8 * 8);
}
#endif

hipError_t launch_dw_gemm_bf16(const DwGemmBf16Args& a, hipStream_t st) {
  if (a.n_units <= 0 || a.n_units > INR_DWGB_MAX_UNITS || a.n_chunks <= 0 || a.tiles_per_chunk <= 0 || a.TL != 128 ||
      a.E > 1024)
    return hipErrorInvalidValue;
  const size_t lds_bytes = (size_t)2 * GB_STAGE * 2 + (size_t)(2 * GB_KS * 3 + 3 * a.E) * sizeof(float);
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = allow_full_lds<dw_gemm_bf16_kernel<128>>();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(dw_gemm_bf16_kernel<128>, dim3((unsigned)(a.n_chunks * a.n_units)), dim3(GB_NT), lds_bytes, st, a);
  return hipGetLastError();
}

}  // namespace inr
