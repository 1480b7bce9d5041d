// inr_dw_gemm.h -- arguments of the batch-level weight-gradient GEMM (inr_dw_gemm.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace inr {

#define INR_DWG_MAX_ITEMS 32

struct DwGemmItem {
  int g_off, h_off;    // floats from the start of a tile's stash to dZ_l [Mblk*32][TL] and h_{l-1} [Kblk*32][TL]
  int gw_off, gb_off;  // slab offsets of dW [M x K] and db [M]
  int Mblk, Kblk, K;   // 32-row blocks of dZ, 32-row blocks of h, columns of dW actually stored
  int mt, nt, unit0;   // set by dw_gemm_units: workgroup tiles (64 WB rows / columns), first workgroup of this item
};

struct DwGemmArgs {
  long long* dbg = nullptr;  // diagnostic builds (-DINR_STAMPS) only: entry / exit stamps (set by the launcher)
  long long dbg_cap = 0;
  const float* save;   // per-tile stash, n_tiles slots
  float* slabs;        // n_chunks slabs of slab_floats floats (layout of the fused kernels' slabs)
  long long save_floats_per_tile;
  int slab_floats;
  int n_tiles, n_chunks, tiles_per_chunk;  // chunk kc = tiles [tile0 + kc tpc, ...) below n_tiles
  int tile0;
  int TL;              // coordinates per tile: 64 or 128
  int WB;              // 32-row blocks per wave-tile side: 4 (workgroup tile 256 x 256) or 3 (192 x 192)
  int WBM;             // 0: square tiles; 2 (with WB = 4, TL = 128): 128-row x 256-column workgroup tiles for short chunks
  int n_items;
  int units, blocks_per_chunk;  // set by the launcher
  DwGemmItem it[INR_DWG_MAX_ITEMS];
};

// launches n_chunks * (workgroup tiles of all items) workgroups
hipError_t launch_dw_gemm(DwGemmArgs& a, hipStream_t st);
// number of workgroup tiles of all items; fills mt / nt / unit0
int dw_gemm_units(DwGemmArgs& a);

}  // namespace inr
