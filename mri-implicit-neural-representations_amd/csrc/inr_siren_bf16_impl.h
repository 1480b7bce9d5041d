// inr_siren_bf16_impl.h -- fused SIREN training step on the bf16 matrix pipe, second design ("weights in LDS,
// activations in registers"): forward + pointwise loss + backward-to-inputs for tiles of 256 coordinates; the weight
// gradients are left to inr_dw_gemm_bf16.hip, which reads the operands this kernel stashes.
//
// Why a second design.  The first bf16 kernel (inr_mlp_bf16_impl.h, still used for the unfused forward / backward
// entry points) kept the fp32 kernel's mapping: every wave streams every weight fragment from L2 for its own 32
// coordinates.  At fp32 MFMA rates that stream is hidden; at bf16 rates (8 MFMAs = 256 cycles per 8 KB of fragments
// per wave) it is the bound -- phase stamps show its GEMM loops at 5-6x their MFMA time -- and half of the kernel
// went into per-tile weight-gradient slabs.  Here:
//   * the layer's weights travel HBM/L2 -> LDS ONCE per workgroup and K-chunk (LDS-DMA, global_load_lds_dwordx4, no
//     VGPRs, no VALU), as pre-packed MFMA A fragments: a chunk = 64 input features x 256 output rows = 32 fragments
//     of 1 KB = 32 KB; a 4-slot ring (128 KB) keeps three chunks in flight ahead of the one being multiplied; the eight
//     waves of the workgroup read the same fragments (conflict-free ds_read_b128: lane l reads bytes [16 l, 16 l + 16));
//   * activations never touch LDS: the fp32 accumulator of layer l (features in registers, coordinates on lanes),
//     after bias + sin + v_cvt_pk_bf16_f32, IS the B operand of layer l+1 (cdna_hip_programming.md section 3, "An
//     accumulator tile as the next MFMA's operand"): element j of lane-half h of K-step (m, s) is feature
//     32 m + 16 s + 8 (j >> 2) + 4 h + (j & 3), and the weight fragments are packed with the same k order
//     (adam_pack_kernel, put_w2);
//   * sin / cos of the hidden layers: v_sin_f32 / v_cos_f32 on w0 z / 2 pi directly -- the instruction reduces its
//     argument itself for |revolutions| <= 256, i.e. |z| <= 53 at w0 = 30, an order of magnitude beyond what a SIREN's
//     pre-activations reach (the encoder phases, which scale with the configured embedding scale, keep their v_fract);
//   * backward mirrors it with the transposed images: dZ_l = dH_l * w0 cos(w0 z_l) is formed in registers from the
//     accumulator of the previous dX GEMM and the stashed z_l, and is the B operand of dH_{l-1} = W_l^T dZ_l;
//   * the stash holds z_l (fp16) and dZ_l (bf16), 4 KB per coordinate, in "row-pair" layout: element (row 2p + e,
//     coordinate c) of a tile at dword p * TL + c, half e -- registers (4g, 4g+1) of an accumulator are rows
//     (2p, 2p+1) of the lane's coordinate, so one v_cvt_pk + one dword store per pair, 128 contiguous bytes per
//     half-wave; the GEMM kernel de-interleaves while staging.
// Chunk stream: the packed image (NetDesc::w2_off) holds the chunks in the order a tile consumes them -- forward
// layers 0 .. D-1, then the transposed images of layers D-1 .. 1 -- so chunk q of the stream is base + 32 KB * q.
#pragma once
#include "inr_mlp_impl.h"
#include "inr_w2.h"

namespace inr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W2_SLOTS = 4;  // ring slots of W2_CHUNK_BYTES

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
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 u = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, u);
}

// ---- the weight-chunk ring ------------------------------------------------------------------------------------
// A workgroup is EIGHT waves = two per SIMD: while one wave of a SIMD issues LDS-DMA pieces (~60-100 cycles of issue
// each), runs an epilogue (VALU + stash traffic) or sits in a wait, its partner keeps the matrix pipe busy.  (With one
// wave per SIMD the phase stamps showed the GEMM phases at 3x their MFMA time: 8 DMA issues + 32 fragment reads + a
// barrier per 32 MFMAs, all in one in-order stream.)  Hence <= 256 registers per wave.
constexpr int W2_WAVES = 8;
constexpr int W2_FPW = 32 / W2_WAVES;  // fragments (1 KB LDS-DMA pieces) per wave and chunk

// issue: this wave's share of chunk image `img` into ring slot `slot`
__device__ __forceinline__ void w2_issue(const char* __restrict__ gbase, int img, int slot, char* ring, int w, int lane) {
  const char* src = gbase + (size_t)img * W2_CHUNK_BYTES + (size_t)(W2_FPW * w) * 1024 + lane * 16;
  char* dst = ring + slot * W2_CHUNK_BYTES + (W2_FPW * w) * 1024;
#pragma unroll
  for (int n = 0; n < W2_FPW; ++n)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + n * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + n * 1024), 16, 0, 0);
}

// Ring protocol.  The GEMMs come in PHASES (one layer's chunks) with an epilogue between them, and an epilogue is 64
// stash stores per wave -- 128 KB per workgroup, a whole GEMM phase's worth of the CU's share of HBM write bandwidth.
// The vector-memory counter retires IN ORDER, so a DMA issued behind those stores cannot be waited for without
// draining them.  Hence every DMA a phase needs is issued BEFORE the epilogue in front of it:
//   * chunk stream position p lives in slot p & 3; four chunks are in flight or in use at any time;
//   * w2_chunk<WAIT>(): wait for this wave's pieces of chunk p, barrier (all pieces landed; everyone is done with
//     chunk p - 1); then -- except for the first chunk of a phase, whose predecessor's slot was refilled by
//     w2_phase_end -- w2_request(p): chunk p + 3 into the slot chunk p - 1 just left;
//   * w2_phase_end(p_next): barrier (everyone is done with the phase's last chunk p_next - 1), request chunk
//     p_next + 3 into its slot: the one request that would otherwise sit behind the epilogue's stores.
// WAIT is the s_waitcnt vmcnt operand: "at most WAIT operations outstanding" covers chunk p when at least WAIT
// vector-memory operations were issued after its DMAs.  Chunk p was requested when chunk p - 4 was done, so for the
// first four chunks after a 64-store epilogue that is >= 64 (the field's maximum, 63, then retires only the oldest
// few of those stores, a GEMM phase old); otherwise only the requests of chunks p + 1, p + 2 are guaranteed: 8.
// Stream position p holds chunk image p mod NQ (a tile's chunk sequence repeats); past the end of the launch the extra
// requests are harmless and keep the number of operations in flight what the waits assume.
// The request of chunk p + 3 is issued by the caller a K-step into chunk p's MFMAs (w2_chunk_mma<true>), not here: four
// LDS-DMA instructions take a few hundred cycles to issue, and right behind the barrier both waves of a SIMD would
// spend them with the matrix pipe empty.
template <int WAIT>
__device__ __forceinline__ void w2_chunk() {
  static_assert(W2_FPW == 4 && W2_SLOTS == 4 && WAIT >= 0 && WAIT <= 63, "the counts assume 4 DMA pieces per wave and chunk, 4 slots");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ void w2_request(const char* __restrict__ gbase, int p, int NQ, char* ring, int w, int lane) {
  w2_issue(gbase, (p + 3) % NQ, (p + 3) & (W2_SLOTS - 1), ring, w, lane);
}
__device__ __forceinline__ void w2_phase_end(const char* __restrict__ gbase, int p_next, int NQ, char* ring, int w, int lane) {
  __builtin_amdgcn_s_barrier();
  w2_issue(gbase, (p_next + 3) % NQ, (p_next + 3) & (W2_SLOTS - 1), ring, w, lane);
}
// a whole phase of n chunks for a wave that only feeds the ring (small batches: the half of the workgroup without a tile)
__device__ __forceinline__ void w2_phase_loader_only(const char* __restrict__ gbase, int& p, int n, bool drain, int NQ, char* ring,
                                                     int w, int lane) {
  for (int c = 0; c < n; ++c) {
    if (c == 0) {
      if (drain)
        w2_chunk<0>();
      else
        w2_chunk<8>();
    } else {
      w2_chunk<8>();
      w2_request(gbase, p, NQ, ring, w, lane);
    }
    ++p;
  }
  w2_phase_end(gbase, p, NQ, ring, w, lane);
}

__device__ __forceinline__ bf16x8 w2_frag(const char* ring, int q, int s_l, int mo, int lane) {
  return *reinterpret_cast<const bf16x8*>(ring + (q & (W2_SLOTS - 1)) * W2_CHUNK_BYTES + ((s_l * 8 + mo) * 64 + lane) * 16);
}

// ---- vector loads the compiler does not see (see the backward epilogue) ------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 w2_rsrc_words(const void* p, int bytes) {  // the descriptor uniform_rsrc() builds, as SGPR words
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  r[2] = (unsigned)bytes;
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ unsigned w2_load_asm(const u32x4& rsrc, int voff, int soff) {
  unsigned v;
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return v;
}
// The first load of a batch: five wait states in front.  gfx9 needs them between a VALU instruction that writes an SGPR
// (v_readlane: how the compiler reloads a spilled SGPR) and a vector-memory instruction that reads it; the compiler pads
// its own memory instructions but does not look inside inline assembly, and one build reloaded this load's offset SGPR
// right in front of it -- the load then read a stale offset and the gradients differed from launch to launch
// (tools/check_inflight_regs.py checks every built kernel for the pattern; tests/test_host.py runs it).
__device__ __forceinline__ unsigned w2_load_asm_first(const u32x4& rsrc, int voff, int soff) {
  unsigned v;
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return v;
}
// at most N vector-memory operations outstanding; the eight registers are operands so that no use moves above the wait
template <int N>
__device__ __forceinline__ void w2_wait_rows(unsigned (&z)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7])
               : "n"(N)
               : "memory");
}

// one chunk = four K = 16 steps against B fragments b[0..3], 8 row blocks (the partner wave hides the LDS latency)
// REQUEST: chunk q is not the first of its phase -- the slot chunk q - 1 left takes chunk q + 3, behind the first K-step
template <bool REQUEST>
__device__ __forceinline__ void w2_chunk_mma(f32x16 (&acc)[8], char* ring, int q, const bf16x8 (&b)[4], int lane,
                                             const char* __restrict__ gbase, int NQ, int w) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8 A[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) A[m] = w2_frag(ring, q, s, m, lane);
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = mfma_bf16(A[m], b[s], acc[m]);
    if (REQUEST && s == 0) w2_request(gbase, q, NQ, ring, w, lane);
  }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------
// Workgroup tile = 256 coordinates = two stash tiles of TL = 128 (what the GEMM kernel and the host count in); wave w
// owns coordinates [32 (w & 3), + 32) of stash tile 2 * (workgroup tile) + (w >> 2): lane (col, half).  MODE_FUSED only.
__global__ __launch_bounds__(512, 2) void inr_siren_bf16_kernel(const NetDesc nd, const LossDesc ld, const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  constexpr int TL = 128, NB = 8, NW = W2_WAVES;
  constexpr int HSZ2 = NB * 32 * TL;  // 2-byte elements per stashed tensor
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA destinations go through M0
  const int half = lane >> 5, col = lane & 31;
  const int wcol = (w & 3) * 32 + col;
  char* ring = lds_raw;
  float* bias_lds = reinterpret_cast<float*>(lds_raw + W2_SLOTS * W2_CHUNK_BYTES);  // [D][256]
  float* encB_lds = bias_lds + nd.D * 256;                                          // [E][3]
  float* red_lds = encB_lds + 3 * nd.E;                                             // [8]
  const int D = nd.D, E = nd.E;
  const char* gbase = reinterpret_cast<const char*>(a.packed + nd.w2_off);
  for (int i = tid; i < D * 256; i += 64 * NW) bias_lds[i] = a.packed[nd.w2_bias_off + i];
  for (int i = tid; i < 3 * E; i += 64 * NW) encB_lds[i] = a.encB[i];
  const int nq0 = w2_nq0(E), NQ = w2_nq(D, E);
  // prime the ring: stream positions 0 .. 3 (the first chunk of a phase requests nothing)
  w2_issue(gbase, 0, 0, ring, w, lane);
  w2_issue(gbase, 1 % NQ, 1, ring, w, lane);
  w2_issue(gbase, 2 % NQ, 2, ring, w, lane);
  w2_issue(gbase, 3 % NQ, 3, ring, w, lane);
  __syncthreads();  // tables in LDS (the DMAs are waited for by the first acquire)
  float loss_acc = 0.f;
  int qs = 0;  // stream position (chunks consumed so far by this workgroup); chunk image = qs % NQ
  const float w0 = nd.w0, krev = nd.w0 * 0.15915494309189535f;
  // two stash tiles per workgroup tile -- unless the batch is too small to fill the chip that way (25 000 rows = 196
  // tiles: 98 workgroups on 256 CUs): then one, and waves 4..7 only help with the weight stream
  const int tpw = a.n_tiles > 256 ? 2 : 1;
  const int n_wtiles = (a.n_tiles + tpw - 1) / tpw;

  for (int wtile = blockIdx.x; wtile < n_wtiles; wtile += gridDim.x) {
    const int stile = tpw * wtile + (w >> 2);     // this wave's stash tile
    const bool tile_ok = (w >> 2) < tpw && stile < a.n_tiles;  // (odd tile counts: the last workgroup tile is half empty)
    if ((w >> 2) >= tpw) {  // small batches: this half of the workgroup has no tile -- it only feeds the ring
      // the working half's phases: layer 0, hidden layers, last layer, dX of the last layer, dX of the hidden layers
      w2_phase_loader_only(gbase, qs, nq0, true, NQ, ring, w, lane);
      for (int l = 1; l < D - 1; ++l) w2_phase_loader_only(gbase, qs, 4, false, NQ, ring, w, lane);
      w2_phase_loader_only(gbase, qs, 1, false, NQ, ring, w, lane);
      w2_phase_loader_only(gbase, qs, 1, false, NQ, ring, w, lane);
      for (int l = D - 2; l >= 1; --l) w2_phase_loader_only(gbase, qs, 4, false, NQ, ring, w, lane);
      continue;
    }
    const long long crow = (long long)stile * TL + wcol;
    const bool valid = tile_ok && crow < a.B;
    unsigned* sv = reinterpret_cast<unsigned*>(a.save + (size_t)(tile_ok ? stile : 0) * nd.save_floats_per_tile);
    const int ts_bytes = tile_ok ? HSZ2 * 2 : 0;  // extent of a stashed tensor: 0 makes every buffer access a no-op
    // per-lane byte offset inside a stashed tensor: pair (2 half) of block 0 group 0, own coordinate
    const int voff = (2 * half * TL + wcol) * 4;
    // coordinates, sampling mask and target row in ONE batch of loads (from row 0 where the lane has no coordinate): nested
    // under `if (valid) ... if (mask[crow])` they were three serialized round trips at the start of every tile
    float x0, x1, x2, gtv[4] = {0.f, 0.f, 0.f, 0.f};
    bool sampled;
    {
      const long long cr = valid ? crow : 0;
      x0 = a.x[3 * cr + 0];
      x1 = a.x[3 * cr + 1];
      x2 = a.x[3 * cr + 2];
      const unsigned char mk = a.mask != nullptr ? a.mask[cr] : (unsigned char)1;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < nd.out_f) gtv[o] = a.gt[cr * nd.out_f + o];
      if (!valid) x0 = x1 = x2 = 0.f;
      sampled = valid && half == 0 && mk != 0;
    }
    const float quarter = half ? 0.25f : 0.f;
    bf16x8 hB[16];  // B operands of the next GEMM: K-step t = 2 m + s  <-  registers 8s .. 8s+7 of accumulator block m
    f32x16 acc[NB];
    int si = 0;  // diagnostic builds: phase stamps 0, 1, 2, ... in program order (tools/stamps.py bf16)

    // epilogue of a forward hidden layer l: z = acc + bias -> stash (fp16 pairs), h = sin(w0 z) -> hB
    auto fwd_epilogue = [&](int l) {
      const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(sv + (size_t)l * (HSZ2 / 2), ts_bytes);
      const float* bl = bias_lds + l * 256 + 4 * half;
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        float hv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + 32 * m + 8 * g);
          float z[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            z[j] = acc[m][4 * g + j] + b4[j];
            hv[4 * g + j] = __builtin_amdgcn_sinf(z[j] * krev);
          }
          const int so = (16 * m + 4 * g) * TL * 4;  // pair 16 m + 4 g (+ 2 half in voff), then the next pair
          __builtin_amdgcn_raw_buffer_store_b32(pack_f16(z[0], z[1]), rs, voff, so, 0);
          __builtin_amdgcn_raw_buffer_store_b32(pack_f16(z[2], z[3]), rs, voff, so + TL * 4, 0);
        }
        float lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          lo[j] = hv[j];
          hi[j] = hv[8 + j];
        }
        hB[2 * m] = pack8(lo);
        hB[2 * m + 1] = pack8(hi);
      }
    };

    // ================================ forward =================================
    INR_STAMP(si); ++si;
    // ---- layer 0: 2E encoder features generated per K-step (half 0: sines, half 1: cosines of features 8t .. 8t+7)
#pragma unroll
    for (int m = 0; m < NB; ++m) acc[m] = zero16();
    for (int ch = 0; ch < nq0; ++ch) {
      if (ch == 0)
        w2_chunk<0>();  // tile start: drain (the previous tile's last epilogue, the x loads)
      else
        w2_chunk<8>();
      bf16x8 b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int t = 4 * ch + s;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float* bj = encB_lds + 3 * (8 * t + j);
          f[j] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(fmaf(x2, bj[2], fmaf(x1, bj[1], fmaf(x0, bj[0], quarter)))));
        }
        b[s] = pack8(f);
      }
      if (ch == 0)
        w2_chunk_mma<false>(acc, ring, qs, b, lane, gbase, NQ, w);
      else
        w2_chunk_mma<true>(acc, ring, qs, b, lane, gbase, NQ, w);
      ++qs;
    }
    w2_phase_end(gbase, qs, NQ, ring, w, lane);
    INR_STAMP(si); ++si;
    fwd_epilogue(0);
    INR_STAMP(si); ++si;
    // ---- hidden layers 1 .. D-2
    for (int l = 1; l < D - 1; ++l) {
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = zero16();
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        // the four chunks behind fwd_epilogue(l - 1)'s 64 stores
        w2_chunk<63>();
        const bf16x8 b[4] = {hB[4 * ch], hB[4 * ch + 1], hB[4 * ch + 2], hB[4 * ch + 3]};
        if (ch == 0)
          w2_chunk_mma<false>(acc, ring, qs, b, lane, gbase, NQ, w);
        else
          w2_chunk_mma<true>(acc, ring, qs, b, lane, gbase, NQ, w);
        ++qs;
      }
      w2_phase_end(gbase, qs, NQ, ring, w, lane);
      INR_STAMP(si); ++si;
      fwd_epilogue(l);
      INR_STAMP(si); ++si;
    }
    // ---- last layer: one row block (rows 0 .. out_f-1 live), its 16 K-steps in ONE chunk
    f32x16 accL = zero16();
    w2_chunk<63>();  // first chunk behind fwd_epilogue(D - 2)'s 64 stores
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const bf16x8 A = *reinterpret_cast<const bf16x8*>(ring + (qs & (W2_SLOTS - 1)) * W2_CHUNK_BYTES + (t * 64 + lane) * 16);
      accL = mfma_bf16(A, hB[t], accL);
    }
    ++qs;
    w2_phase_end(gbase, qs, NQ, ring, w, lane);
    INR_STAMP(si); ++si;
    float y[4], dy[4], g[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float z = accL[o];  // half 0: rows 0..3
      if (o < nd.out_f) z += bias_lds[(D - 1) * 256 + o];
      act_fwd_rt(nd.last_act, z, w0, y[o], dy[o]);
      g[o] = 0.f;
      if (half == 0 && valid && o < nd.out_f && a.out != nullptr) a.out[crow * nd.out_f + o] = y[o];
    }
    if (sampled) loss_acc += loss_row(ld, nd.out_f, y, gtv, g);
    float dzl[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) dzl[o] = (half == 0 && o < nd.out_f) ? g[o] * dy[o] : 0.f;
    // dZ_last rows (0,1), (2,3) of this coordinate: two dwords behind the hidden tensors (read by the GEMM kernel)
    if (half == 0 && tile_ok) {
      unsigned* dzL = sv + (size_t)2 * (D - 1) * (HSZ2 / 2);
      dzL[wcol] = pack_bf16(dzl[0], dzl[1]);
      dzL[TL + wcol] = pack_bf16(dzl[2], dzl[3]);
    }

    // ================================ backward ================================
    INR_STAMP(si); ++si;
    // dH_{D-2} = W_last^T dZ_last: one K-step (k = output row: element j of half 0 is row j for j < 4)
    {
      float v[8] = {dzl[0], dzl[1], dzl[2], dzl[3], 0.f, 0.f, 0.f, 0.f};
      const bf16x8 b0 = pack8(v);
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = zero16();
      w2_chunk<63>();  // second chunk behind fwd_epilogue(D - 2)'s 64 stores
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma_bf16(w2_frag(ring, qs, 0, m, lane), b0, acc[m]);  // K-steps 1..3: zero weights
      ++qs;
      w2_phase_end(gbase, qs, NQ, ring, w, lane);
    }
    INR_STAMP(si); ++si;
    for (int l = D - 2; l >= 0; --l) {
      // dZ_l = dH_l * w0 cos(w0 z_l): z_l back from the stash (row pairs), dZ_l to the stash and into hB
      const __amdgpu_buffer_rsrc_t rg = uniform_rsrc(sv + (size_t)(D - 1 + l) * (HSZ2 / 2), ts_bytes);
      // the layer's 64 row pairs of this coordinate, all in flight before the first use (the B operands of the GEMM
      // that just ended are dead, the next ones not yet formed: the registers are there)
      // The loads are INLINE ASSEMBLY with hand-placed waits (w2_wait_rows): gfx9 has one counter for vector loads and
      // stores, hipcc's wait insertion treats loads and stores pending together as able to return out of order and
      // answers every such wait with vmcnt(0) -- here that drained all 64 loads AND the weight DMAs in flight in front
      // of the first row block (the backward epilogues took 14-18 k cycles against 5 k forward; without the dZ stores,
      // i.e. with only loads pending, the same code ran at forward speed).  The counter retires in order: behind the
      // last load of row block m lie the 8 (7 - m) later loads and the 8 m stores of the blocks already done = 56.
      unsigned zall[NB][8];
      const u32x4 rz4 = w2_rsrc_words(sv + (size_t)l * (HSZ2 / 2), ts_bytes);
#pragma unroll
      for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int so = (16 * m + 4 * gq) * TL * 4;
          zall[m][2 * gq] = (m == 0 && gq == 0) ? w2_load_asm_first(rz4, voff, so) : w2_load_asm(rz4, voff, so);
          zall[m][2 * gq + 1] = w2_load_asm(rz4, voff, so + TL * 4);
        }
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        w2_wait_rows<56>(zall[m]);
        const unsigned (&zz)[8] = zall[m];
        float dz[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f16x2 zp = __builtin_bit_cast(f16x2, zz[2 * gq + e]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const float c = __builtin_amdgcn_cosf((float)zp[k] * krev);
              dz[4 * gq + 2 * e + k] = acc[m][4 * gq + 2 * e + k] * (w0 * c);
            }
          }
          const int so = (16 * m + 4 * gq) * TL * 4;
          __builtin_amdgcn_raw_buffer_store_b32(pack_bf16(dz[4 * gq], dz[4 * gq + 1]), rg, voff, so, 0);
          __builtin_amdgcn_raw_buffer_store_b32(pack_bf16(dz[4 * gq + 2], dz[4 * gq + 3]), rg, voff, so + TL * 4, 0);
        }
        float lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          lo[j] = dz[j];
          hi[j] = dz[8 + j];
        }
        hB[2 * m] = pack8(lo);  // (hB is free: the forward pass is over)
        hB[2 * m + 1] = pack8(hi);
        __builtin_amdgcn_sched_barrier(0);  // row block by row block, so that each waits for its own eight loads only
      }
      INR_STAMP(si); ++si;
      if (l == 0) break;
      // dH_{l-1} = W_l^T dZ_l
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = zero16();
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        // the four chunks behind the backward epilogue's 64 stores
        w2_chunk<63>();
        const bf16x8 b[4] = {hB[4 * ch], hB[4 * ch + 1], hB[4 * ch + 2], hB[4 * ch + 3]};
        if (ch == 0)
          w2_chunk_mma<false>(acc, ring, qs, b, lane, gbase, NQ, w);
        else
          w2_chunk_mma<true>(acc, ring, qs, b, lane, gbase, NQ, w);
        ++qs;
      }
      w2_phase_end(gbase, qs, NQ, ring, w, lane);
      INR_STAMP(si); ++si;
    }
  }
  // every DMA this wave issued has landed before the workgroup (and its LDS) goes away
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

inline hipError_t launch_siren_bf16_fused(const NetDesc& nd, const LossDesc& ld, const MlpArgs& a, int grid, hipStream_t st) {
  const size_t lds_bytes = (size_t)W2_SLOTS * W2_CHUNK_BYTES + ((size_t)nd.D * 256 + 3 * (size_t)nd.E + W2_WAVES) * sizeof(float);
  if (lds_bytes > 160 * 1024 || a.save == nullptr || a.slabs == nullptr || a.save_by_block) return hipErrorInvalidValue;
  hipError_t e = allow_full_lds<inr_siren_bf16_kernel>();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(inr_siren_bf16_kernel, dim3(grid), dim3(64 * W2_WAVES), lds_bytes, st, nd, ld, a);
  return hipGetLastError();
}

}  // namespace inr
