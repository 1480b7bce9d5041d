// inr_dw_gemm_bf16.h -- arguments of the bf16 batch-level weight-gradient GEMM (inr_dw_gemm_bf16.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace inr {

#define INR_DWGB_MAX_UNITS 48

// one 256-row x 256-column block of one layer's dW.  Offsets into a tile's stash are in DWORDS (= row pairs x
// coordinates: rows (2p, 2p+1) of coordinate c share the dword at p * TL + c).
struct DwGemmBf16Unit {
  int dz_off;          // dwords from the start of a tile's stash to dZ_l (bf16 pairs)
  int z_off;           // ... to z_{l-1} (fp16 pairs); < 0: first layer, h = gauss encoder features of the coordinates
  float krev;          // w0 / (2 pi) of the layer that produced z_{l-1}
  int gw_off, gb_off;  // slab offsets (floats) of dW [M x K] and db [M]
  int M;               // rows of dW stored (256 for the hidden layers' padded slabs, out_features for the last layer)
  int K;               // columns of dW
  int n0;              // first column of this block
};

struct DwGemmBf16Args {
  const void* save;    // per-tile stash, n_tiles slots
  float* slabs;        // n_chunks slabs of slab_floats floats
  const float* coords; // [B,3] (first-layer units)
  const float* encB;   // [E,3]
  long long B;
  long long save_floats_per_tile;
  int slab_floats;
  int n_tiles, n_chunks, tiles_per_chunk;
  int TL, E, n_units;
  DwGemmBf16Unit unit[INR_DWGB_MAX_UNITS];
};

hipError_t launch_dw_gemm_bf16(const DwGemmBf16Args& a, hipStream_t st);

}  // namespace inr
