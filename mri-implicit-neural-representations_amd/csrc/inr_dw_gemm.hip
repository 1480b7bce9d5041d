// inr_dw_gemm.hip -- batch-level weight gradients: dW_l = dZ_l^T h_{l-1} over ALL coordinates of a step, as a
// split-K fp32 MFMA GEMM that reads both operands from the per-tile stash the fused kernel leaves behind.
//
// Why not inside the fused kernel (as inr_mlp_impl.h can, and did): there a workgroup owns a slab of every dW and adds
// its tile's contribution to it.  For one tile per wave the slab is written once (196 x 1.57 MB at the graded
// shape, then read again by the reduction), the dW passes occupy the 196 busy CUs for a third of the kernel, and
// three barriers per layer fence them.  Here every CU works, a workgroup keeps a 256 x 256 block of dW in
// registers across a whole chunk of tiles (K = 512 coordinates and more), and a quarter as many slabs exist.
//
// Operand layout (written by the fused kernels): tile t, tensor at offset `off` (floats) inside the tile's stash,
// row r (a feature), coordinate c of the tile:  save[t * save_floats_per_tile + off + r * TL + c].
// A workgroup stages K-steps of 32 coordinates (512 rows x 128 B, whole cache lines) through two LDS buffers; each of
// its four waves accumulates a 128 x 128 block: 4 x 4 MFMA blocks (v_mfma_f32_32x32x2_f32), 8 ds_read_b128 per 64 MFMAs.
// dW rows follow the C layout of the MFMA: register r of lane (li, half) is row (r&3) + 8(r>>2) + 4 half, column li.
#include <hip/hip_runtime.h>

#include "inr_dw_gemm.h"
#include "inr_stamp_rt.h"
#include "inr_launch.h"

namespace inr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DWG_LD = 36;  // LDS row pitch (floats): 16-byte reads of 8 consecutive rows hit 32 distinct banks
// WB: 32-row blocks per wave tile side (4: 128 x 128 per wave, 256 x 256 per workgroup; 3: 96 / 192 for the 384-row
// WIRE shape).  A stage holds the G tile [64 WB][32] and the H tile [64 WB][32].
// WBM x WB: 32-row blocks per wave tile (rows x columns of dW); the workgroup tile is (64 WBM) x (64 WB).  WBM < WB
// (128 x 256) is for SHORT chunks: at 25 000 rows a 256 x 256 workgroup sees only K = 512 coordinates, and the 245
// workgroups' 256 KB of results -- all stored at the same moment, with nothing left to overlap -- were a sixth of the
// launch; half-height tiles over twice the K halve the slab stream (and what reduce_slabs reads) for one more pass over h.
template <int WBM, int WB>
constexpr int dwg_stage() { return 64 * (WBM + WB) * DWG_LD; }

template <int WBM, int WB, bool BIAS>
__device__ __forceinline__ void dwg_mma(f32x16 (&acc)[WBM][WB], float (&bsum)[WBM], const f32x4 (&A)[WBM],
                                        const f32x4 (&B)[WB]) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int i = 0; i < WBM; ++i) {
      if (BIAS) bsum[i] += A[i][e];
#pragma unroll
      for (int j = 0; j < WB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i][e], B[j][e], acc[i][j], 0, 0, 0);
    }
}

// one K-step (32 coordinates) of a workgroup's operands: 128 WB rows x 128 B, thread t fetches 16-byte segment t & 7
// of rows (t >> 3) + 32 k -- whole cache lines per 8 lanes
// (through a buffer descriptor on the tile's wave-uniform base + the K-step: the per-thread part of the address is the
// 32-bit byte offset roff[k], formed once -- no 64-bit vector adds per load inside the MFMA stream)
template <int NF>
__device__ __forceinline__ void dwg_fetch(f32x4 (&v)[NF], const float* __restrict__ sv, const int (&roff)[NF],
                                          int kstep) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(sv + 32 * kstep);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
  for (int k = 0; k < NF; ++k) v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, roff[k], 0, 0));
}

template <int NF>
__device__ __forceinline__ void dwg_stash(float* buf, const f32x4 (&v)[NF], int t) {
  float* p = buf + (t >> 3) * DWG_LD + (t & 7) * 4;
#pragma unroll
  for (int k = 0; k < NF; ++k) *reinterpret_cast<f32x4*>(p + k * 32 * DWG_LD) = v[k];
}

// Workgroup = a (64 WBM) x (64 WB) block of one item's dW over one chunk of tiles; waves 2 x 2, (32 WBM) x (32 WB) each.
template <int TL, int WBM, int WB, bool BIAS>
__device__ __forceinline__ void dwg_body(const DwGemmArgs& a, const DwGemmItem& it, int kc, int mb0, int nb0,
                                         float* lds) {
  constexpr int NF = 2 * (WBM + WB);  // 16-byte fetches per thread and stage
  constexpr int TR = 64 * WBM;        // rows of the G (dZ) tile; the H tile has 64 WB
  constexpr int DWG_STAGE = dwg_stage<WBM, WB>();
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int wm = w >> 1, wn = w & 1;
  f32x16 acc[WBM][WB];
  float bsum[WBM];
#pragma unroll
  for (int i = 0; i < WBM; ++i) {
    bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < WB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // global row of each of this thread's fetches (rows past a tensor's extent: row 0, never stored)
  int roff[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    const int r = (t >> 3) + 32 * k;  // [0, TR): G rows, [TR, 2 TR): H rows
    const bool isg = r < TR;
    const int blk = isg ? mb0 + (r >> 5) : nb0 + ((r - TR) >> 5);
    const int row = (blk < (isg ? it.Mblk : it.Kblk)) ? blk * 32 + (r & 31) : 0;
    roff[k] = ((isg ? it.g_off : it.h_off) + row * TL + (t & 7) * 4) * 4;  // bytes
  }
  constexpr int KS = TL / 32;  // K-steps per tile
  // chunk kc = tiles [tile0 + kc tpc, tile0 + (kc + 1) tpc), below n_tiles
  const int t0 = a.tile0 + kc * a.tiles_per_chunk;
  int n_mine = a.n_tiles - t0;
  if (n_mine > a.tiles_per_chunk) n_mine = a.tiles_per_chunk;
  const int n_steps = (n_mine > 0 ? n_mine : 0) * KS;
  const float* As = lds + (wm * 32 * WBM + li) * DWG_LD + 4 * half;
  const float* Bs = lds + (TR + wn * 32 * WB + li) * DWG_LD + 4 * half;
  f32x4 v[NF];
  if (n_steps > 0) {  // (an empty chunk still writes its zeros)
    dwg_fetch(v, a.save + (size_t)t0 * a.save_floats_per_tile, roff, 0);
    dwg_stash(lds, v, t);
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < n_steps; ++s) {
    const int sn = s + 1;
    // (unconditional, so that the stage is one basic block: the last stage fetches itself again and writes the buffer
    // nobody reads any more)
    const int sf = sn < n_steps ? sn : s;
    dwg_fetch(v, a.save + (size_t)(t0 + sf / KS) * a.save_floats_per_tile, roff, sf % KS);
    const float* Ab = As + (s & 1) * DWG_STAGE;
    const float* Bb = Bs + (s & 1) * DWG_STAGE;
    // The stage as ONE scheduled stream (round 4): fragments of quarter q + 1 are read while quarter q multiplies, the next
    // stage's global loads go out behind the first MFMAs and its LDS writes behind those of the last quarter (the other
    // stage buffer has been free since the barrier that ended the previous stage) -- every memory / LDS instruction in the
    // matrix pipe's shadow.  Left to the compiler's order (loads, then per quarter: reads, MFMAs; writes; barrier) a
    // stage took 4.5 us for 3.6 us of MFMAs.
    f32x4 A[2][WBM], B[2][WB];
#pragma unroll
    for (int i = 0; i < WBM; ++i) A[0][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * DWG_LD);
#pragma unroll
    for (int j = 0; j < WB; ++j) B[0][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * DWG_LD);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < 3) {
#pragma unroll
        for (int i = 0; i < WBM; ++i) A[(q + 1) & 1][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * DWG_LD + 8 * (q + 1));
#pragma unroll
        for (int j = 0; j < WB; ++j) B[(q + 1) & 1][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * DWG_LD + 8 * (q + 1));
      } else {
        dwg_stash(lds + (sn & 1) * DWG_STAGE, v, t);
      }
      dwg_mma<WBM, WB, BIAS>(acc, bsum, A[q & 1], B[q & 1]);
      if (q == 0) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one global load of the next stage
        }
      }
      if (q < 3) {
#pragma unroll
        for (int i = 0; i < WBM + WB; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one fragment read of the next quarter
        }
      } else {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // one LDS write of the next stage
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  float* slab = a.slabs + (size_t)kc * a.slab_floats;
  const int mb = mb0 + WBM * wm, nb = nb0 + WB * wn;
#pragma unroll
  for (int i = 0; i < WBM; ++i) {
    if (mb + i >= it.Mblk) continue;
#pragma unroll
    for (int j = 0; j < WB; ++j) {
      const int colj = 32 * (nb + j) + li;
      if (colj < it.K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * (mb + i) + (r & 3) + 8 * (r >> 2) + 4 * half;
          slab[it.gw_off + (size_t)row * it.K + colj] = acc[i][j][r];
        }
      }
    }
    if (BIAS && wn == 0) {
      const float tot = bsum[i] + __shfl_xor(bsum[i], 32);
      if (half == 0) slab[it.gb_off + 32 * (mb + i) + li] = tot;
    }
  }
}

template <int TL, int WBM, int WB>
__global__ __launch_bounds__(256) void dw_gemm_kernel(const DwGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  INR_RT_STAMP(a.dbg, a.dbg_cap, 4, threadIdx.x >> 6, threadIdx.x & 63, 44);
  const int kc = blockIdx.x / a.blocks_per_chunk;
  const int unit = blockIdx.x - kc * a.blocks_per_chunk;
  int k = 0;
  while (k + 1 < a.n_items && unit >= a.it[k + 1].unit0) ++k;
  const DwGemmItem& it = a.it[k];
  const int u = unit - it.unit0;
  const int mi = u / it.nt, ni = u % it.nt;
  if (ni == 0)
    dwg_body<TL, WBM, WB, true>(a, it, kc, 2 * WBM * mi, 0, lds);
  else
    dwg_body<TL, WBM, WB, false>(a, it, kc, 2 * WBM * mi, 2 * WB * ni, lds);
  INR_RT_STAMP(a.dbg, a.dbg_cap, 4, threadIdx.x >> 6, threadIdx.x & 63, 45);
}

template <int TL, int WBM, int WB>
static hipError_t launch_tl(const DwGemmArgs& a, dim3 grid, hipStream_t st) {
  constexpr size_t lds_bytes = (size_t)2 * dwg_stage<WBM, WB>() * sizeof(float);  // two stages (4 x 4: 147 KB)
  auto k = dw_gemm_kernel<TL, WBM, WB>;
  {
    hipError_t e = allow_full_lds<dw_gemm_kernel<TL, WBM, WB>>();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}

#ifdef INR_STAMPS
extern long long* g_stamp_buf;  // inr_api.hip
extern long long g_stamp_cap;
#endif

hipError_t launch_dw_gemm(DwGemmArgs& a, hipStream_t st) {
  if (a.n_items <= 0) return hipSuccess;
#ifdef INR_STAMPS
  a.dbg = g_stamp_buf, a.dbg_cap = g_stamp_cap;
#endif
  if (a.n_items > INR_DWG_MAX_ITEMS || a.n_chunks <= 0 || a.tiles_per_chunk <= 0) return hipErrorInvalidValue;
  a.blocks_per_chunk = dw_gemm_units(a);
  a.units = a.blocks_per_chunk;
  const dim3 grid((unsigned)(a.n_chunks * a.blocks_per_chunk));
  const int wbm = a.WBM > 0 ? a.WBM : a.WB;
  if (a.TL == 128 && a.WB == 4 && wbm == 4) return launch_tl<128, 4, 4>(a, grid, st);
  if (a.TL == 128 && a.WB == 4 && wbm == 2) return launch_tl<128, 2, 4>(a, grid, st);
  if (a.TL == 64 && a.WB == 4 && wbm == 4) return launch_tl<64, 4, 4>(a, grid, st);
  if (a.TL == 64 && a.WB == 3 && wbm == 3) return launch_tl<64, 3, 3>(a, grid, st);
  return hipErrorInvalidValue;
}

// number of (64 WB)^2 workgroup tiles of all items; also fills mt / nt / unit0
int dw_gemm_units(DwGemmArgs& a) {
  int units = 0;
  for (int k = 0; k < a.n_items; ++k) {
    DwGemmItem& it = a.it[k];
    const int wbm = a.WBM > 0 ? a.WBM : a.WB;
    it.mt = (it.Mblk + 2 * wbm - 1) / (2 * wbm);
    it.nt = (it.Kblk + 2 * a.WB - 1) / (2 * a.WB);
    it.unit0 = units;
    units += it.mt * it.nt;
  }
  return units;
}

}  // namespace inr
