// inr_dw_gemm_bf16.hip -- batch-level weight gradients of the bf16 throughput path:  dW_l = dZ_l^T h_{l-1}  over all
// coordinates of a chunk of tiles, on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), split-K over chunks of tiles.
//
// Why it exists: with the matrix pipe 16x faster than on the fp32 path the fused bf16 kernel was paced by what it
// WROTE -- one private 0.66 MB gradient slab per 128-coordinate tile (196 of them at 25 000 rows: 129 MB per launch,
// in 2-byte stores).  Here the fused kernel only leaves its operands behind (z_l as fp16, dZ_l as bf16, 4 KB per
// coordinate) and ~50 workgroups-per-layer each keep a 256 x 256 block of dW in registers across ~10 tiles: the slab
// stream shrinks from (tiles x 0.66 MB) to (chunks x 1.3 MB), written once in 16-byte fp32 stores.
//
// Operands (stash of inr_mlp_bf16_kernel, per tile of TL = 128 coordinates, element (row r, coordinate c) at
// [r * TL + c]):  A = dZ_l  bf16;  B = h_{l-1} = sin(w0 z_{l-1}) RECOMPUTED here from the stashed fp16 z_{l-1} with the
// hardware sine (the fused kernel never stores h), or, for the first layer, the gauss encoder features regenerated from
// the tile's coordinates exactly as the forward pass formed them (sin of x.B_j revolutions, + 1/4 turn for the cosine
// half).  db_l = row sums of dZ_l come out of the same matrix pipe: one extra MFMA per A fragment against a constant
// all-ones B fragment (every column of that accumulator is the row sum).
//
// Workgroup = one 256 x 256 block of one layer's dW over one chunk of tiles; four waves 2 x 2, 4 x 4 MFMA blocks each
// (256 accumulator registers); K-steps of 64 coordinates (whole 128-byte lines of both operands) staged through two
// LDS stages, rows pitched 144 B so that the 16-byte fragment reads of 16 consecutive rows hit 16 distinct 4-bank
// slots; the sine of stage s+1 is computed in the shadow of the MFMAs of stage s (4.5 VALU per MFMA).
#include <hip/hip_runtime.h>

#include "inr_dw_gemm_bf16.h"
#include "inr_launch.h"

namespace inr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GB_KS = 64;                 // coordinates per stage
constexpr int GB_PITCH = 72;              // LDS row pitch in 2-byte elements: 64 coordinates + 16 bytes of padding
constexpr int GB_TILE = 256 * GB_PITCH;   // one operand tile (elements)
constexpr int GB_STAGE = 2 * GB_TILE;     // A tile + B tile
constexpr int GB_NF = 8;                  // 16-byte pieces per thread, operand and stage (256 rows x 8 pieces / 256 threads)

__device__ __forceinline__ bf16x8 sin_of_z(const f16x8 z, float krev) {
  f32x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf((float)z[j] * krev));
  return __builtin_convertvector(o, bf16x8);
}

// feature `row` (< E: sine, >= E: cosine of the same phase) of 8 consecutive coordinates whose (x0,x1,x2) sit in xs
__device__ __forceinline__ bf16x8 gauss_features(const float* xs, const float* encB, int E, int row) {
  const int s = row < E ? row : row - E;
  const float quarter = row < E ? 0.f : 0.25f;
  const float b0 = encB[3 * s + 0], b1 = encB[3 * s + 1], b2 = encB[3 * s + 2];
  f32x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float rev = fmaf(xs[3 * j + 2], b2, fmaf(xs[3 * j + 1], b1, fmaf(xs[3 * j], b0, quarter)));
    o[j] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev));
  }
  return __builtin_convertvector(o, bf16x8);
}

template <int TL, bool ENC, bool BIAS>
__device__ __forceinline__ void dwgb_body(const DwGemmBf16Args& a, const DwGemmBf16Unit& it, int kc, char* lds_raw) {
  __bf16* lds = reinterpret_cast<__bf16*>(lds_raw);
  float* xs_lds = reinterpret_cast<float*>(lds_raw + (size_t)2 * GB_STAGE * 2);  // [2 stages][64 coords][3]
  float* encB_lds = xs_lds + 2 * GB_KS * 3;                                      // [E][3]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int wm = w >> 1, wn = w & 1;
  constexpr int KS_PER_TILE = TL / GB_KS;
  const int t0 = kc * a.tiles_per_chunk;
  int n_mine = a.n_tiles - t0;
  if (n_mine > a.tiles_per_chunk) n_mine = a.tiles_per_chunk;
  const int n_steps = (n_mine > 0 ? n_mine : 0) * KS_PER_TILE;

  f32x16 acc[4][4], accb[BIAS ? 4 : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (BIAS)
#pragma unroll
      for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
  }
  if (ENC) {
    for (int i = t; i < 3 * a.E; i += 256) encB_lds[i] = a.encB[i];
  }
  // loader: piece p = t & 7 (8 coordinates = 16 bytes), rows (t >> 3) + 32 k
  const int seg = t & 7, row0 = t >> 3;
  const __bf16* sv = reinterpret_cast<const __bf16*>(a.save);
  const size_t tile_elems = (size_t)a.save_floats_per_tile * 2;
  bf16x8 ra[GB_NF];
  f16x8 rb[GB_NF];

  auto fetch = [&](int s) {
    const int tile = t0 + s / KS_PER_TILE, c0 = (s % KS_PER_TILE) * GB_KS;
    const __bf16* base = sv + (size_t)tile * tile_elems;
#pragma unroll
    for (int k = 0; k < GB_NF; ++k) {
      const int row = row0 + 32 * k;
      ra[k] = *reinterpret_cast<const bf16x8*>(base + it.dz_off + (size_t)row * TL + c0 + 8 * seg);
      if (!ENC)
        rb[k] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(base) + it.z_off +
                                                (size_t)(it.n0 + row) * TL + c0 + 8 * seg);
    }
    if (ENC && t < GB_KS * 3) {  // the stage's 64 coordinates, 192 floats: rows past B read as coordinate 0
      const long long r = (long long)tile * TL + c0 + t / 3;
      xs_lds[(s & 1) * GB_KS * 3 + t] = r < a.B ? a.coords[3 * r + (t % 3)] : 0.f;
    }
  };
  // second half of a stage's staging: A pieces as they are, B pieces through the sine; `part` = which two of the 8
  auto stash_part = [&](int s, int part) {
    __bf16* st = lds + (size_t)(s & 1) * GB_STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int k = 2 * part + kk, row = row0 + 32 * k;
      *reinterpret_cast<bf16x8*>(st + row * GB_PITCH + 8 * seg) = ra[k];
      bf16x8 hb;
      if (ENC)
        hb = gauss_features(xs_lds + (s & 1) * GB_KS * 3 + 24 * seg, encB_lds, a.E, it.n0 + row);
      else
        hb = sin_of_z(rb[k], it.krev);
      *reinterpret_cast<bf16x8*>(st + GB_TILE + row * GB_PITCH + 8 * seg) = hb;
    }
  };

  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

  if (n_steps > 0) {
    fetch(0);
    if (ENC) __syncthreads();  // xs of stage 0 (and the encoder matrix) are in LDS
#pragma unroll
    for (int part = 0; part < 4; ++part) stash_part(0, part);
  }
  __syncthreads();
  const __bf16* As = lds + (wm * 128 + li) * GB_PITCH + 8 * half;
  const __bf16* Bs = lds + GB_TILE + (wn * 128 + li) * GB_PITCH + 8 * half;
#pragma unroll 1
  for (int s = 0; s < n_steps; ++s) {
    // stage s+1 is fetched and staged while stage s multiplies.  The last iteration re-stages stage s itself
    // (identical bytes over identical bytes) instead of branching: one basic block per iteration, so that the
    // scheduler may lay the sines between the MFMAs.
    const int sn = s + 1 < n_steps ? s + 1 : s;
    fetch(sn);
    const __bf16* Ab = As + (size_t)(s & 1) * GB_STAGE;
    const __bf16* Bb = Bs + (size_t)(s & 1) * GB_STAGE;
    if (ENC) __syncthreads();  // xs of stage sn is complete before any stash_part(sn, .) reads it
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // four K = 16 sub-steps of the stage's 64 coordinates
      bf16x8 A[4], B[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) A[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * GB_PITCH + 16 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) B[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * GB_PITCH + 16 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
        if (BIAS) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], ones, accb[i], 0, 0, 0);
      }
      stash_part(sn, q);  // a quarter of the next stage's sines in the shadow of this sub-step's MFMAs
#pragma unroll
      for (int n = 0; n < (BIAS ? 20 : 16); ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, BIAS ? 4 : 5, 0);      // the VALU work it hides
      }
    }
    __syncthreads();
  }
  // ---- chunk slab: dW rows follow the MFMA C layout (register r of lane (li, half): row (r&3)+8(r>>2)+4 half, col li)
  float* slab = a.slabs + (size_t)kc * a.slab_floats;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rb0 = 32 * (4 * wm + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = it.n0 + 32 * (4 * wn + j) + li;
      if (col < it.K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rb0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          slab[it.gw_off + (size_t)row * it.K + col] = acc[i][j][r];
        }
      }
    }
    if (BIAS && wn == 0 && li == 0) {  // every column of accb is the row sum: column 0 stores it
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[it.gb_off + rb0 + (r & 3) + 8 * (r >> 2) + 4 * half] = accb[i][r];
    }
  }
}

template <int TL>
__global__ __launch_bounds__(256) void dw_gemm_bf16_kernel(const DwGemmBf16Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int kc = blockIdx.x / a.n_units;
  const DwGemmBf16Unit& it = a.unit[blockIdx.x - kc * a.n_units];
  if (it.z_off < 0) {
    if (it.n0 == 0)
      dwgb_body<TL, true, true>(a, it, kc, lds_raw);
    else
      dwgb_body<TL, true, false>(a, it, kc, lds_raw);
  } else {
    dwgb_body<TL, false, true>(a, it, kc, lds_raw);  // hidden layers: one column tile (K <= 256), always with db
  }
}

hipError_t launch_dw_gemm_bf16(const DwGemmBf16Args& a, hipStream_t st) {
  if (a.n_units <= 0 || a.n_units > INR_DWGB_MAX_UNITS || a.n_chunks <= 0 || a.tiles_per_chunk <= 0 || a.TL != 128 ||
      a.E > 1024)
    return hipErrorInvalidValue;
  const size_t lds_bytes = (size_t)2 * GB_STAGE * 2 + (size_t)(2 * GB_KS * 3 + 3 * a.E) * sizeof(float);
  if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = allow_full_lds<dw_gemm_bf16_kernel<128>>();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(dw_gemm_bf16_kernel<128>, dim3((unsigned)(a.n_chunks * a.n_units)), dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}

}  // namespace inr
