// inr_dw_gemm_bf16.h -- arguments of the bf16 path's batch-level weight-gradient GEMM (inr_dw_gemm_bf16.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "inr_w2.h"

namespace inr {

#define INR_DWGB_MAX_UNITS 48

// one 256-row x 256-column block of one layer's dW.  Offsets into a tile's stash are in DWORDS (inr_w2.h: 8-bit tensors
// in row-quad layout, rows 4q .. 4q+3 of coordinate c in the dword at q * TL + c; dZ_last as fp16 row pairs).
struct DwGemmBf16Unit {
  int dz_off;          // dwords from the start of a tile's stash to dZ_l (bf8 quads; last layer: fp16 pairs)
  int z_off;           // ... to the phase bytes of z_{l-1}; < 0: first layer, h = gauss encoder features of the coordinates
  int gw_off, gb_off;  // slab offsets (floats) of dW [M x K] and db [M]
  int M;               // rows of dW stored (256 for the hidden layers' padded slabs, out_features for the last layer)
  int K;               // columns of dW
  int n0;              // first column of this block; first-layer units: first of its 128 encoder frequencies (the block
                       // holds their sine AND cosine columns, n0 + j and E + n0 + j)
};

struct DwGemmBf16Args {
  long long* dbg = nullptr;  // diagnostic builds (-DINR_STAMPS) only: entry / exit stamps (set by the launcher)
  long long dbg_cap = 0;
  const void* save;    // per-tile stash, n_tiles slots
  float* slabs;        // n_chunks slabs of slab_floats floats
  const float* coords; // [B,3] (first-layer units)
  const float* encB;   // [E,3]
  float* dz_state;     // the 4 gradient-scale words of this kind of step (inr_w2.h); nullptr: sums leave as they are
  float* dz_count;     // its two counters of clipped / flushed steps (nullptr: not counted)
  long long B;
  long long save_floats_per_tile;
  int slab_floats;
  int n_tiles;
  // split-K over chunks of tiles, two classes of units: the first n_enc_units (first layer: their B operand is encoder
  // arithmetic, ~1.3 x the time of a hidden unit per tile) take n_chunks_enc chunks of tiles_per_chunk_enc tiles, the others
  // n_chunks of tiles_per_chunk; workgroups = n_enc_units * n_chunks_enc + (n_units - n_enc_units) * n_chunks.  Chunk kc of a
  // unit writes the unit's entries of slab kc.
  int n_enc_units, n_chunks_enc, tiles_per_chunk_enc;
  int n_chunks, tiles_per_chunk;
  int TL, E, n_units;
  DwGemmBf16Unit unit[INR_DWGB_MAX_UNITS];
};

hipError_t launch_dw_gemm_bf16(const DwGemmBf16Args& a, hipStream_t st);
// the roll alone (after a calibration pass of the fused kernel: inr_api.hip)
hipError_t launch_dz_roll(float* st, float* cnt, hipStream_t stream);

#if defined(__HIPCC__)
// Next step's gradient scale from what this step saw: st[1] = bits of max |dZ * mult|, st[3] = the power of two S inside
// mult.  amax / S is the step's largest |dZ| in units of the loss gradient's own normalisation (fused steps: of
// d(loss)/d(out) * count); the next S puts it into [2^4, 2^5) (inr_w2.h W2_DZ_TARGET_EXP): 2^10.8 of headroom below bf8's largest finite value 57 344
// -- sequential batches of a k-space differ by two orders of magnitude in their largest gradient (the centre of a coil
// against its periphery); beyond the headroom the conversion saturates (the kernel runs with MODE.FP16_OVFL set: a
// clipped step, not an infinity) --, 2^18 above bf8's smallest normal value, 2^20 above its smallest subnormal.  A step
// whose gradient is exactly zero keeps the scale.  `cnt` (two words of the plan's state, inr_w2.h): steps whose gradients
// were clipped / mostly flushed are counted.
__device__ __forceinline__ void dz_state_roll(float* st, float* cnt) {
  const float amax = __builtin_bit_cast(float, reinterpret_cast<unsigned*>(st)[1]);
  if (amax > 0.f && amax < 3.0e38f) {
    int ex;
    (void)frexpf(amax / st[3], &ex);  // amax / S = f * 2^ex, f in [0.5, 1)
    ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
    st[0] = ldexpf(1.0f, W2_DZ_TARGET_EXP - ex);
    if (cnt != nullptr) {
      if (amax > W2_BF8_MAX) cnt[0] += 1.f;
      if (amax < W2_DZ_LOW) cnt[1] += 1.f;
    }
  }
  reinterpret_cast<unsigned*>(st)[1] = 0u;
}
#endif

}  // namespace inr
