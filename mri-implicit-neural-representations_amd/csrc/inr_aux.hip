// inr_aux.hip -- HBM-bound helper kernels of the INR engine (gfx950): slab reduction,
// Adam + weight re-packing, gauss encoder, pointwise losses.  All are coalesced streaming
// kernels; none reshapes its work into a GEMM.
#include <algorithm>
#include <cstdlib>
#include "inr_device.h"
#include "inr_aux.h"
#include "inr_w2.h"

namespace inr {

// ---------------------------------------------------------------------------------------------
// Flat-parameter element -> where it lives in the virtual real matrices the fused kernel works on.
// A flat float is a real weight (LT_REAL, LT_WIRE_FIRST), the Re or Im part of a complex weight
// (LT_WIRE_HIDDEN: W = Wr + jWi acts as [[Wr,-Wi],[Wi,Wr]] on interleaved (Re,Im) rows;
// LT_WIRE_LAST: only the real output row [Wr,-Wi] exists), or a bias entry.
// ---------------------------------------------------------------------------------------------
struct VirtualPos {
  int n;          // number of virtual entries (0, 1 or 2)
  int row[2], col[2];
  float sign[2];
  int bias_row;   // >= 0: this flat element is a bias entry of that virtual row (n == 0)
};

__device__ __forceinline__ bool locate(const NetDesc& nd, int i, int& l, VirtualPos& vp) {
  vp.n = 0;
  vp.bias_row = -1;
  for (l = 0; l < nd.ND; ++l) {
    const LayerDesc& L = nd.L[l];
    const int off = i - L.w_off;
    if (off >= 0 && off < L.wn) {
      if (L.ltype == LT_REAL || L.ltype == LT_GABOR_MU) {
        vp.n = 1; vp.row[0] = off / L.K; vp.col[0] = off - vp.row[0] * L.K; vp.sign[0] = 1.f;
      } else if (L.ltype == LT_WIRE_FIRST) {
        const int r = off / L.K;
        vp.n = 1; vp.row[0] = 2 * r; vp.col[0] = off - r * L.K; vp.sign[0] = 1.f;
      } else {
        const int kc = L.K >> 1;  // complex in-features
        const int r = off / (2 * kc), rem = off - r * 2 * kc, j = rem >> 1, c = rem & 1;
        if (L.ltype == LT_WIRE_HIDDEN) {
          vp.n = 2;
          if (c == 0) {  // Wr -> (2r,2j) and (2r+1,2j+1)
            vp.row[0] = 2 * r; vp.col[0] = 2 * j; vp.sign[0] = 1.f;
            vp.row[1] = 2 * r + 1; vp.col[1] = 2 * j + 1; vp.sign[1] = 1.f;
          } else {       // Wi -> +(2r+1,2j) and -(2r,2j+1)
            vp.row[0] = 2 * r + 1; vp.col[0] = 2 * j; vp.sign[0] = 1.f;
            vp.row[1] = 2 * r; vp.col[1] = 2 * j + 1; vp.sign[1] = -1.f;
          }
        } else {  // LT_WIRE_LAST: out = Re(W h + beta): row r = [Wr, -Wi]
          vp.n = 1; vp.row[0] = r; vp.col[0] = 2 * j + c; vp.sign[0] = c ? -1.f : 1.f;
        }
      }
      return true;
    }
    const int ob = i - L.b_off;
    if (ob >= 0 && ob < L.bn) {
      if (L.ltype == LT_REAL || L.ltype == LT_GABOR_MU) vp.bias_row = ob;
      else if (L.ltype == LT_WIRE_FIRST) vp.bias_row = 2 * ob;
      else if (L.ltype == LT_WIRE_HIDDEN) vp.bias_row = ob;            // (i, c) -> row 2i + c
      else vp.bias_row = (ob & 1) ? -2 : (ob >> 1);                    // imaginary output bias: no effect
      return true;
    }
  }
  return false;
}

// Sum of the per-slab loss words by one whole workgroup (the extra last one of the reduction grids): lane t adds
// slabs t, t + 256, ..., then a fixed LDS tree -- reproducible, and 196 dependent cache misses shorter than the
// single-thread loop it replaces (which was the critical path of the whole reduction kernel).
__device__ __forceinline__ void loss_words_sum(const float* __restrict__ slabs, int n, size_t stride, int off,
                                               float* __restrict__ loss_out) {
  __shared__ float red[256];
  const int t = threadIdx.x;
  float s = 0.f;
  for (int b = t; b < n; b += 256) s += slabs[(size_t)b * stride + off];
  red[t] = s;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (t < w) red[t] += red[t + w];
    __syncthreads();
  }
  if (t == 0) loss_out[0] = red[0];
}

// ---------------------------------------------------------------------------------------------
// grads[i] = sum over blocks (in block order: deterministic) of the slab entries that flat
// element i owns; the loss word is summed by thread 0.
// ---------------------------------------------------------------------------------------------
// sum of n slab entries `stride` floats apart, 16 independent loads in flight, added in slab order
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, size_t stride, int n) {
  float s = 0.f;
  int b = 0;
  for (; b + 16 <= n; b += 16) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = p[(size_t)(b + k) * stride];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
  }
  // the tail the same way (clamped addresses, selects after the loads): a chunk-slab set of 23 is a batch of 16 and then 7
  // -- as a scalar loop those were 7 serialized L2 / HBM round trips.  A short tail takes a short batch (wave-uniform n): the
  // 4 chunk slabs of the 8 x 512 filter network were 16 loads per entry, 12 of them of the same word.
  if (n - b > 4) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = p[(size_t)(b + k < n ? b + k : n - 1) * stride];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += b + k < n ? v[k] : 0.f;
  } else if (b < n) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = p[(size_t)(b + k < n ? b + k : n - 1) * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k) s += b + k < n ? v[k] : 0.f;
  }
  return s;
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const NetDesc nd, const float* __restrict__ slabs_all,
                                                           int n_blocks_all, float* __restrict__ grads,
                                                           float* __restrict__ loss_out, SlabSplit split) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const size_t sf = (size_t)nd.slab_floats;
  if (i < nd.P) {
    int l;
    VirtualPos vp;
    float s = 0.f;
    if (locate(nd, i, l, vp) && nd.L[l].live != 0) {  // dead layers: slab region never written -> 0
      const LayerDesc& L = nd.L[l];
      const bool second = split.n2 > 0 && ((((split.mask >> l) & 1u) != 0) || (i >= split.lo && i < split.hi));
      const float* __restrict__ slabs = second ? slabs_all + (size_t)n_blocks_all * sf : slabs_all;
      const int n_blocks = second ? split.n2 : n_blocks_all;
      if (vp.n >= 1) {
        const size_t o0 = (size_t)L.gw_off + (size_t)vp.row[0] * L.K + vp.col[0];
        s = vp.sign[0] * slab_sum(slabs + o0, sf, n_blocks);
        if (vp.n == 2) {
          const size_t o1 = (size_t)L.gw_off + (size_t)vp.row[1] * L.K + vp.col[1];
          s += vp.sign[1] * slab_sum(slabs + o1, sf, n_blocks);
        }
      } else if (vp.bias_row >= 0) {
        s = slab_sum(slabs + (size_t)L.gb_off + vp.bias_row, sf, n_blocks);
      }
    }
    grads[i] = s;
  }
  if (loss_out != nullptr && blockIdx.x == gridDim.x - 1) loss_words_sum(slabs_all, n_blocks_all, sf, nd.slab_loss_off, loss_out);
}

// Fast path when a slab has the flat-parameter layout (every layer LT_REAL).  A workgroup owns 256
// consecutive gradient entries (64 lanes x 16 B); its four waves each sum a contiguous quarter of
// the slabs (8 independent 16-byte loads in flight per lane) and the quarters are combined through
// LDS in wave order -- a fixed summation tree, so results are bitwise reproducible.
__global__ __launch_bounds__(256) void reduce_slabs_real_kernel(const float* __restrict__ slabs_all, int n_blocks_all,
                                                                int slab_floats, int P, int loss_off,
                                                                float* __restrict__ grads,
                                                                float* __restrict__ loss_out, SlabSplit split) {
  __shared__ f32x4 part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i4 = (blockIdx.x * 64 + lane) * 4;
  // split.lo / split.hi are multiples of 4 (checked by the launcher): a lane's four entries share a slab set
  const bool second = split.n2 > 0 && i4 >= split.lo && i4 < split.hi;
  const float* __restrict__ slabs = second ? slabs_all + (size_t)n_blocks_all * slab_floats : slabs_all;
  const int n_blocks = second ? (split.n3 > 0 && i4 >= split.lo3 && i4 < split.hi3 ? split.n3 : split.n2) : n_blocks_all;
  const int per = (n_blocks + 3) / 4;
  const int b0 = w * per, b1 = (b0 + per < n_blocks) ? b0 + per : n_blocks;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 < P) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(slabs + (size_t)(b + u) * slab_floats + i4);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < b1; ++b) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)b * slab_floats + i4);
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && i4 < P) {
    const f32x4 t = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    for (int j = 0; j < 4 && i4 + j < P; ++j) grads[i4 + j] = t[j];
  }
  if (loss_out != nullptr && blockIdx.x == gridDim.x - 1) loss_words_sum(slabs_all, n_blocks_all, (size_t)slab_floats, loss_off, loss_out);
}

// GaborLayer centres (mfn.py:116-131).  The fused kernel leaves, per filter and row j,
//   S1[j,k] = sum_c a_cj x_ck,  s0_j = sum_c a_cj,  T_j = sum_c a_cj |x_c|^2      (a = g_h * h, so that
// d h / d q = h * gamma with q = mu_j . x, and D = |x|^2 + |mu_j|^2 - 2 q); after the block reduction
// grads[mu] holds S1 and grads[gamma] holds s0.  This kernel (one wave per row) finishes
//   d gamma_j = -0.5 (T_j + |mu_j|^2 s0_j - 2 sum_k mu_jk S1_jk),   d mu_jk = gamma_j (S1_jk - s0_j mu_jk).
__global__ __launch_bounds__(64) void gabor_finish_kernel(const NetDesc nd, const float* __restrict__ slabs,
                                                          int n_blocks, const float* __restrict__ params,
                                                          const float* __restrict__ packed, float* __restrict__ grads) {
  const LayerDesc& L = nd.L[nd.mu0 + blockIdx.y];
  const int j = blockIdx.x, lane = threadIdx.x;
  if (j >= L.M) return;
  const int NBW = nd.NB * 32;
  float t = 0.f, dot = 0.f;
  for (int b = lane; b < n_blocks; b += 64) t += slabs[(size_t)b * nd.slab_floats + L.gb_off + NBW + j];
  const float* mu = params + L.w_off + (size_t)j * L.K;
  float* g = grads + L.w_off + (size_t)j * L.K;
  for (int k = lane; k < L.K; k += 64) dot = fmaf(mu[k], g[k], dot);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    t += __shfl_xor(t, off);
    dot += __shfl_xor(dot, off);
  }
  const float s0 = grads[L.b_off + j], gamma = params[L.b_off + j], m2 = packed[L.pbias_off + NBW + j];
  for (int k = lane; k < L.K; k += 64) g[k] = gamma * (g[k] - s0 * mu[k]);
  if (lane == 0) grads[L.b_off + j] = -0.5f * ((t + m2 * s0) - 2.f * dot);
}

// |mu_j|^2 into the second half of the centre layer's bias image (after every (re)pack of the parameters)
__global__ __launch_bounds__(64) void gabor_m2_kernel(const NetDesc nd, const float* __restrict__ params,
                                                      float* __restrict__ packed) {
  const LayerDesc& L = nd.L[nd.mu0 + blockIdx.y];
  const int j = blockIdx.x, lane = threadIdx.x;
  if (j >= L.M) return;
  const float* mu = params + L.w_off + (size_t)j * L.K;
  float s = 0.f;
  for (int k = lane; k < L.K; k += 64) s = fmaf(mu[k], mu[k], s);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) packed[L.pbias_off + nd.NB * 32 + j] = s;
}

hipError_t launch_reduce_slabs(const NetDesc& nd, const float* slabs, int n_blocks, float* grads, float* loss_out,
                               const float* params, const float* packed, hipStream_t st, SlabSplit split) {
  bool all_real = true;  // ... and slab layout == flat layout (not the case for MFN: L[] order != flat order)
  for (int l = 0; l < nd.ND; ++l) all_real = all_real && nd.L[l].ltype == LT_REAL && nd.L[l].gw_off == nd.L[l].w_off;
  if (split.n3 > 0 && !(all_real && ((split.lo | split.hi | split.lo3 | split.hi3) & 3) == 0 && split.mask == 0))
    return hipErrorInvalidValue;  // a third slab count is the flat-layout reduction's only
  if (all_real && ((split.lo | split.hi) & 3) == 0 && split.mask == 0) {  // slab offsets == flat offsets; slab_floats % 64 == 0 keeps every slab 16-byte aligned
    const int grid = (nd.P + 255) / 256;
    hipLaunchKernelGGL(reduce_slabs_real_kernel, dim3(grid + 1), dim3(256), 0, st, slabs, n_blocks, nd.slab_floats, nd.P,
                       nd.slab_loss_off, grads, loss_out, split);
  } else {
    const int grid = (nd.P + 255) / 256;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid + 1), dim3(256), 0, st, nd, slabs, n_blocks, grads, loss_out, split);
  }
  if (nd.gabor)
    hipLaunchKernelGGL(gabor_finish_kernel, dim3(nd.L[nd.mu0].M, nd.mfn_n + 1), dim3(64), 0, st, nd, slabs, n_blocks,
                       params, packed, grads);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor algorithm, amsgrad=False; train.py:76,190) fused with the
// re-pack of every weight into the two MFMA A-fragment images and of every bias into its padded
// bias image.  do_update == 0: pack only.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_fwd(const NetDesc& nd, const LayerDesc& L, int l, float* packed, int row, int k,
                                        float v) {
  int h, s;
  if (L.korder == 1) {
    h = k >= nd.E;
    s = h ? k - nd.E : k;
  } else {
    h = k & 1;
    s = k >> 1;
  }
  packed[L.pf_off + ((size_t)((s >> 2) * L.Mblk + (row >> 5)) * 64 + h * 32 + (row & 31)) * 4 + (s & 3)] = v;
}

__device__ __forceinline__ void put_tr(const LayerDesc& L, float* packed, int row, int k, float v) {
  const int h = row & 1, s = row >> 1;  // transposed image A'[i=k][k'=row]
  packed[L.pb_off + ((size_t)((s >> 2) * L.Kblk + (k >> 5)) * 64 + h * 32 + (k & 31)) * 4 + (s & 3)] = v;
}

// Fragment images of the row-split fused step (inr_mlp_rs_impl.h): v_mfma_f32_16x16x4_f32 A operands, one float4 per lane
// and k-step = the four 16-row blocks of a wave's 64 rows: W[row][k] -> float4 ((k >> 2) * 4 + (row >> 6)) * 64 +
// (k & 3) * 16 + (row & 15), component (row >> 4) & 3.  Layer 0 keeps its k in CHUNK order: 32 encoder phases at a time,
// their sines then their cosines (the kernel generates the features chunk by chunk).
__device__ __forceinline__ void put_rs(const NetDesc& nd, const LayerDesc& L, int l, float* packed, int row, int k, float v) {
  if (l >= nd.D - 1) return;  // the last layer's rows are read from the flat parameters
  int kk = k;
  if (l == 0) {
    const int trig = k >= nd.E, sp = trig ? k - nd.E : k;
    kk = (sp >> 5) * 64 + trig * 32 + (sp & 31);
  }
  packed[L.rf_off + ((size_t)(((kk >> 2) * 4 + (row >> 6)) * 64 + (kk & 3) * 16 + (row & 15))) * 4 + ((row >> 4) & 3)] = v;
  if (l >= 1)  // transposed image A'[i = k][k' = row]
    packed[L.rb_off + ((size_t)(((row >> 2) * 4 + (k >> 6)) * 64 + (row & 3) * 16 + (k & 15))) * 4 + ((k >> 4) & 3)] = v;
}

// "Weight panels in LDS" images of the bf16 path (inr_siren_bf16_impl.h; layout and the constant factors that ride in the
// images: inr_w2.h): weight W_l[row][k] goes into the forward panels of layer l -- times w0 / 2 pi for the sine layers, so
// that their accumulators are phases in revolutions -- and, for l >= 1, into the transposed panels (out index = k,
// contraction index = row) times the w0 of the layer below; biases into the fp32 table, scaled like their weights.
// Plain SIREN plans only (every layer LT_REAL, hidden rows = 256 padded).
__device__ __forceinline__ float w2_krev(const NetDesc& nd, int l) {
  return l < nd.D - 1 ? nd.L[l].omega * 0.15915494309189535f : 1.0f;
}
__device__ __forceinline__ void put_w2(const NetDesc& nd, int l, float* packed, int row, int k, float v) {
  __bf16* img = reinterpret_cast<__bf16*>(packed + nd.w2_off);
  const int D = nd.D, E = nd.E;
  int t, h, j;
  if (l == 0) {  // gauss features: K-step t holds frequencies 8t .. 8t+7 -- lane half h the four 8t + 4h .. + 3, as (sine,
    // cosine) pairs in elements (2i, 2i+1): one phase chain per pair in the kernel; two K-steps per panel
    const int trig = k >= E;
    const int kk = trig ? k - E : k;
    t = kk >> 3;
    h = (kk & 7) >> 2;
    j = 2 * (kk & 3) + trig;
    img[w2_index(t >> 1, (t & 1) * 8 + (row >> 5), h * 32 + (row & 31), j)] = (__bf16)(v * w2_krev(nd, 0));
  } else {
    w2_kperm_inv(k, t, h, j);
    if (l == D - 1)  // last layer (rows < 32): one panel, its 16 K-steps
      img[w2_index(w2_p_last(D, E), t, h * 32 + row, j)] = (__bf16)v;
    else
      img[w2_index(w2_p_fwd(l, E) + (row >> 5), t, h * 32 + (row & 31), j)] = (__bf16)(v * w2_krev(nd, l));
    w2_kperm_inv(row, t, h, j);
    const float vt = v * nd.L[l - 1].omega;
    if (l == D - 1)  // one K-step (rows < 4: t = 0), eight row blocks
      img[w2_index(w2_p_lastT(D, E), k >> 5, h * 32 + (k & 31), j)] = (__bf16)vt;
    else
      img[w2_index(w2_p_T(l, D, E) + (k >> 5), t, h * 32 + (k & 31), j)] = (__bf16)vt;
  }
}

// one flat entry: the Adam update from gradient `g` (do_update); returns the entry's (new) value
__device__ __forceinline__ float adam_update_entry(const NetDesc& nd, int i, float g, float* __restrict__ params,
                                                   float* __restrict__ m1, float* __restrict__ m2, const AdamArgs& aa) {
  float p = params[i];
  bool live = true;
  if (aa.has_dead) {
    if (aa.n_dead >= 0) {  // (a handful of ranges in kernel arguments: scalar compares -- the walk over 26 layer descriptors
      for (int r = 0; r < aa.n_dead; ++r)  //  below was most of the update's time on the 8 x 512 filter network)
        if (i >= aa.dead_lo[r] && i < aa.dead_hi[r]) live = false;
    } else {
      for (int l = 0; l < nd.ND; ++l) {
        const LayerDesc& L = nd.L[l];
        if ((i >= L.w_off && i < L.w_off + L.wn) || (i >= L.b_off && i < L.b_off + L.bn)) live = L.live != 0;
      }
    }
  }
  if (aa.do_update && live) {
    if (aa.weight_decay != 0.f) g = fmaf(aa.weight_decay, p, g);
    if (aa.l1 != 0.f) g += aa.l1 * (p > 0.f ? 1.f : (p < 0.f ? -1.f : 0.f));  // d/dp lambda*sum|p|
    if (aa.l2 != 0.f) g = fmaf(2.f * aa.l2, p, g);                             // d/dp lambda*|sum p^2|
    float step_size = aa.step_size, bc2_sqrt = aa.bc2_sqrt;
    if (aa.sched != nullptr) {  // wave-uniform: scalar loads
      const int t = min(*aa.step_dev, aa.n_sched - 1);
      step_size = aa.sched[2 * t];
      bc2_sqrt = aa.sched[2 * t + 1];
    }
    float m = m1[i], v = m2[i];
    m = m + (g - m) * aa.omb1;                // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(g * g, aa.omb2, v * aa.beta2);   // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + aa.eps;
    p = p - step_size * (m / denom);          // param.addcdiv_(exp_avg, denom, value=-step_size)
    m1[i] = m;
    m2[i] = v;
    params[i] = p;
  }
  return p;
}

// one flat entry: Adam update (adam_update_entry), then its image entries
__device__ __forceinline__ void adam_pack_entry(const NetDesc& nd, int i, float g, float* __restrict__ params,
                                                float* __restrict__ m1, float* __restrict__ m2,
                                                float* __restrict__ packed, const AdamArgs& aa) {
  const float p = adam_update_entry(nd, i, g, params, m1, m2, aa);
  if (aa.all_real) {  // SIREN / FFN: one weight entry or one bias entry, no sign, no pair
    for (int l = 0; l < nd.ND; ++l) {
      const LayerDesc& L = nd.L[l];
      const int off = i - L.w_off;
      if (off >= 0 && off < L.wn) {
        const int row = off / L.K, k = off - row * L.K;
        if (nd.bf16) {  // bf16 plans keep the panel stream only
          put_w2(nd, l, packed, row, k, p);
          return;
        }
        put_fwd(nd, L, l, packed, row, k, p);
        if (L.pb_off >= 0) put_tr(L, packed, row, k, p);
        if (nd.rs) put_rs(nd, L, l, packed, row, k, p);
        return;
      }
      const int ob = i - L.b_off;
      if (ob >= 0 && ob < L.bn) {
        if (nd.bf16)
          packed[nd.w2_bias_off + l * 256 + ob] = p * w2_krev(nd, l);
        else
          packed[L.pbias_off + ob] = p;
        return;
      }
    }
    return;
  }
  int l;
  VirtualPos vp;
  if (!locate(nd, i, l, vp)) return;
  const LayerDesc& L = nd.L[l];
  for (int t = 0; t < vp.n; ++t) {
    put_fwd(nd, L, l, packed, vp.row[t], vp.col[t], vp.sign[t] * p);
    if (L.pb_off >= 0) put_tr(L, packed, vp.row[t], vp.col[t], vp.sign[t] * p);
  }
  if (vp.bias_row >= 0) packed[L.pbias_off + vp.bias_row] = p;
}

__global__ __launch_bounds__(256) void adam_pack_kernel(const NetDesc nd, float* __restrict__ params,
                                                        const float* __restrict__ grads, float* __restrict__ m1,
                                                        float* __restrict__ m2, float* __restrict__ packed,
                                                        AdamArgs aa) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nd.P) return;
  adam_pack_entry(nd, i, aa.do_update ? grads[i] : 0.f, params, m1, m2, packed, aa);
}

// reduce_slabs_real_kernel + adam_pack_kernel in one launch (single-rank steps: nothing sits between the reduction and
// the update).  Same summation tree, same update arithmetic: bit-identical to the two launches; the gradient is still
// written (callers read it).  The four waves of a workgroup each take one of a lane's four entries for the update.
__global__ __launch_bounds__(256) void reduce_adam_real_kernel(const NetDesc nd, const float* __restrict__ slabs_all,
                                                               int n_blocks_all, float* __restrict__ grads,
                                                               float* __restrict__ loss_out, SlabSplit split,
                                                               float* __restrict__ params, float* __restrict__ m1,
                                                               float* __restrict__ m2, float* __restrict__ packed,
                                                               AdamArgs aa) {
  __shared__ f32x4 part[4][64];
  const int slab_floats = nd.slab_floats, P = nd.P;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i4 = (blockIdx.x * 64 + lane) * 4;
  const bool second = split.n2 > 0 && i4 >= split.lo && i4 < split.hi;
  const float* __restrict__ slabs = second ? slabs_all + (size_t)n_blocks_all * slab_floats : slabs_all;
  const int n_blocks = second ? (split.n3 > 0 && i4 >= split.lo3 && i4 < split.hi3 ? split.n3 : split.n2) : n_blocks_all;
  const int per = (n_blocks + 3) / 4;
  const int b0 = w * per, b1 = (b0 + per < n_blocks) ? b0 + per : n_blocks;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 < P) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(slabs + (size_t)(b + u) * slab_floats + i4);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < b1; ++b) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)b * slab_floats + i4);
  }
  part[w][lane] = s;
  __syncthreads();
  if (i4 < P) {
    const f32x4 t = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    if (w == 0)
      for (int j = 0; j < 4 && i4 + j < P; ++j) grads[i4 + j] = t[j];
    if (i4 + w < P) adam_pack_entry(nd, i4 + w, t[w], params, m1, m2, packed, aa);
  }
  if (loss_out != nullptr && blockIdx.x == gridDim.x - 1)
    loss_words_sum(slabs_all, n_blocks_all, (size_t)slab_floats, nd.slab_loss_off, loss_out);
}

// Sharded data-parallel update: the entries [lo, hi) of the flat vector, gradient = this rank's chunk of the
// reduce-scatter (grads_shard[i - lo]); no image is written -- every rank re-packs after the all-gather of the parameters.
__global__ __launch_bounds__(256) void adam_shard_kernel(const NetDesc nd, float* __restrict__ params,
                                                         const float* __restrict__ grads_shard, float* __restrict__ m1,
                                                         float* __restrict__ m2, int lo, int hi, AdamArgs aa) {
  const int i = lo + blockIdx.x * 256 + threadIdx.x;
  if (i >= hi) return;
  adam_update_entry(nd, i, grads_shard[i - lo], params, m1, m2, aa);
}

// The images of LT_REAL layers written in IMAGE order (one float4 slot per thread, blockIdx.y = layer): coalesced
// stores, the weights gathered through the L2 -- where adam_pack_entry scatters 4-byte stores (16 segments per wave).
// Slots: forward image Kpad8 * Mblk * 8 | transposed image Mpad8 * Kblk * 8 (if any) | bias image Mblk * 8; padding slots
// are written as zeros (what they hold anyway).  The inverse of put_fwd / put_tr above.
__global__ __launch_bounds__(256) void pack_images_real_kernel(const NetDesc nd, const float* __restrict__ params,
                                                               float* __restrict__ packed) {
  const LayerDesc& L = nd.L[blockIdx.y];
  const int nf = L.Kpad8 * L.Mblk * 8, nt = L.pb_off >= 0 ? L.Mpad8 * L.Kblk * 8 : 0, nb = L.Mblk * 8;
  int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nf + nt + nb) return;
  const float* __restrict__ W = params + L.w_off;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (q < nf) {
    const int lane = q & 63, g = (q >> 6) / L.Mblk, mb = (q >> 6) - g * L.Mblk;
    const int h = lane >> 5, row = mb * 32 + (lane & 31);
    if (row < L.M) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int sidx = 4 * g + c;
        const int k = L.korder == 1 ? (sidx < nd.E ? h * nd.E + sidx : L.K) : 2 * sidx + h;
        if (k < L.K) v[c] = W[(size_t)row * L.K + k];
      }
    }
    *reinterpret_cast<f32x4*>(packed + L.pf_off + (size_t)q * 4) = v;
    return;
  }
  q -= nf;
  if (q < nt) {
    const int lane = q & 63, g = (q >> 6) / L.Kblk, kb = (q >> 6) - g * L.Kblk;
    const int h = lane >> 5, k = kb * 32 + (lane & 31);
    if (k < L.K) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int row = 2 * (4 * g + c) + h;
        if (row < L.M) v[c] = W[(size_t)row * L.K + k];
      }
    }
    *reinterpret_cast<f32x4*>(packed + L.pb_off + (size_t)q * 4) = v;
    return;
  }
  q -= nt;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (4 * q + c < L.bn) v[c] = params[L.b_off + 4 * q + c];
  *reinterpret_cast<f32x4*>(packed + L.pbias_off + (size_t)q * 4) = v;
}

// Gradient of the penalties of models/regularization.py:21-36 added to the flat gradient entries [lo, hi) (grads[i - lo]),
// for plans with complex64 tensors too (WIRE / WIRE2D hidden and last layers, stored as interleaved (re, im) pairs):
//   L1  lambda * sum |p|:  real entries sign(p); a complex z = a + ib contributes |z|, i.e. (a, b) / |z| (0 at z = 0)
//   L2  lambda * |S|, S = sum p^2 over every Parameter -- a COMPLEX sum when the model has complex tensors (z^2 = a^2 - b^2
//       + 2iab):  with u = conj(S) / |S| = (ur, ui):  real entry 2 p ur;  d/da = 2 (ur a - ui b),  d/db = -2 (ur b + ui a).
// u is read from device memory (the caller forms S, which also holds the squares of the frozen omega_0 / scale_0
// Parameters, networks.py:191-192); l2_dir == nullptr means u = (1, 0): all-real models.  Reads params, writes grads only:
// a pair's two entries never race.
__global__ __launch_bounds__(256) void reg_grad_kernel(const NetDesc nd, const float* __restrict__ params,
                                                       float* __restrict__ grads, int lo, int hi, float l1, float l2,
                                                       const float* __restrict__ l2_dir) {
  const int i = lo + blockIdx.x * 256 + threadIdx.x;
  if (i >= hi) return;
  int role = 0;  // 0: real entry, 1 / 2: real / imaginary part of a complex entry
  for (int l = 0; l < nd.ND; ++l) {
    const LayerDesc& L = nd.L[l];
    if (L.ltype != LT_WIRE_HIDDEN && L.ltype != LT_WIRE_LAST) continue;
    const int off = i - L.w_off, ob = i - L.b_off;
    if (off >= 0 && off < L.wn) role = 1 + (off & 1);
    if (ob >= 0 && ob < L.bn) role = 1 + (ob & 1);
  }
  float ur = 1.f, ui = 0.f;
  if (l2_dir != nullptr) {
    ur = l2_dir[0];
    ui = l2_dir[1];
  }
  const float p = params[i];
  float add = 0.f;
  if (role == 0) {
    if (l1 != 0.f) add += l1 * (p > 0.f ? 1.f : (p < 0.f ? -1.f : 0.f));
    if (l2 != 0.f) add += 2.f * l2 * p * ur;
  } else {
    const float q = params[role == 1 ? i + 1 : i - 1];
    const float a = role == 1 ? p : q, b = role == 1 ? q : p;
    if (l1 != 0.f) {
      const float r = hypotf(a, b);
      if (r > 0.f) add += l1 * (p / r);
    }
    if (l2 != 0.f) add += role == 1 ? 2.f * l2 * (ur * a - ui * b) : -2.f * l2 * (ur * b + ui * a);
  }
  grads[i - lo] += add;
}

__global__ void step_advance_kernel(int* step_dev) { *step_dev += 1; }

hipError_t launch_step_advance(int* step_dev, hipStream_t st) {
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, st, step_dev);
  return hipGetLastError();
}

// has_dead / the dead layers' flat entries as merged ranges (MultiscaleKFourier's unused heads and last stage: SURVEY A.4 #3)
static void adam_dead_ranges(const NetDesc& nd, AdamArgs& aa) {
  aa.has_dead = 0;
  aa.n_dead = 0;
  int lo[2 * INR_MAX_LAYERS], hi[2 * INR_MAX_LAYERS], n = 0;
  for (int l = 0; l < nd.ND; ++l) {
    const LayerDesc& L = nd.L[l];
    if (L.live != 0) continue;
    aa.has_dead = 1;
    if (L.wn > 0) lo[n] = L.w_off, hi[n] = L.w_off + L.wn, ++n;
    if (L.bn > 0) lo[n] = L.b_off, hi[n] = L.b_off + L.bn, ++n;
  }
  for (int i = 1; i < n; ++i)  // by offset (insertion sort: a few dozen entries at most)
    for (int j = i; j > 0 && lo[j] < lo[j - 1]; --j) std::swap(lo[j], lo[j - 1]), std::swap(hi[j], hi[j - 1]);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (m > 0 && lo[i] <= hi[m - 1]) {
      hi[m - 1] = std::max(hi[m - 1], hi[i]);
    } else {
      lo[m] = lo[i], hi[m] = hi[i], ++m;
    }
  }
  if (m > 8) {
    aa.n_dead = -1;
    return;
  }
  aa.n_dead = m;
  for (int i = 0; i < m; ++i) aa.dead_lo[i] = lo[i], aa.dead_hi[i] = hi[i];
}

hipError_t launch_adam_pack(const NetDesc& nd, float* params, const float* grads, float* m1, float* m2,
                            float* packed, const AdamArgs& aa_in, hipStream_t st) {
  AdamArgs aa = aa_in;
  aa.all_real = 1;
  for (int l = 0; l < nd.ND; ++l) aa.all_real = aa.all_real && (nd.L[l].ltype == LT_REAL || nd.L[l].ltype == LT_GABOR_MU);
  adam_dead_ranges(nd, aa);
  // large plain-layer networks (BASELINE config 4: 4.47 M entries): the update elementwise, then the images in image order
  // (INR_PACK_BY_IMAGE=0 / 1 forces the choice where both exist: the tests hold one against the other)
  const char* force = getenv("INR_PACK_BY_IMAGE");
  const bool big = force != nullptr ? force[0] == '1' : nd.P >= (1 << 18);
  bool by_image = aa.all_real && !nd.bf16 && !nd.rs && !nd.gabor && big;
  int slots = 0;
  for (int l = 0; l < nd.ND && by_image; ++l) {
    const LayerDesc& L = nd.L[l];
    by_image = L.ltype == LT_REAL && L.pf_off >= 0 && (L.pf_off & 3) == 0 && (L.pb_off < 0 || (L.pb_off & 3) == 0) &&
               (L.pbias_off & 3) == 0;
    slots = std::max(slots, L.Kpad8 * L.Mblk * 8 + (L.pb_off >= 0 ? L.Mpad8 * L.Kblk * 8 : 0) + L.Mblk * 8);
  }
  if (by_image) {
    if (aa.do_update)
      hipLaunchKernelGGL(adam_shard_kernel, dim3((nd.P + 255) / 256), dim3(256), 0, st, nd, params, grads, m1, m2, 0, nd.P,
                         aa);
    hipLaunchKernelGGL(pack_images_real_kernel, dim3((slots + 255) / 256, nd.ND), dim3(256), 0, st, nd, params, packed);
    return hipGetLastError();
  }
  const int grid = (nd.P + 255) / 256;
  hipLaunchKernelGGL(adam_pack_kernel, dim3(grid), dim3(256), 0, st, nd, params, grads, m1, m2, packed, aa);
  if (nd.gabor)
    hipLaunchKernelGGL(gabor_m2_kernel, dim3(nd.L[nd.mu0].M, nd.mfn_n + 1), dim3(64), 0, st, nd, params, packed);
  return hipGetLastError();
}

hipError_t launch_reg_grad(const NetDesc& nd, const float* params, float* grads, int lo, int hi, float l1, float l2,
                           const float* l2_dir, hipStream_t st) {
  if (hi <= lo || (l1 == 0.f && l2 == 0.f)) return hipSuccess;
  hipLaunchKernelGGL(reg_grad_kernel, dim3((hi - lo + 255) / 256), dim3(256), 0, st, nd, params, grads, lo, hi, l1, l2,
                     l2_dir);
  return hipGetLastError();
}

hipError_t launch_adam_shard(const NetDesc& nd, float* params, const float* grads_shard, float* m1, float* m2, int lo,
                             int hi, const AdamArgs& aa_in, hipStream_t st) {
  if (hi <= lo) return hipSuccess;
  AdamArgs aa = aa_in;
  aa.all_real = 0;
  adam_dead_ranges(nd, aa);
  hipLaunchKernelGGL(adam_shard_kernel, dim3((hi - lo + 255) / 256), dim3(256), 0, st, nd, params, grads_shard, m1, m2, lo,
                     hi, aa);
  return hipGetLastError();
}

// reduction (+ the Adam update when `adam` is given: one launch on the flat-layout fast path, two otherwise)
hipError_t launch_reduce_slabs_adam(const NetDesc& nd, const float* slabs, int n_blocks, float* grads, float* loss_out,
                                    float* params, float* m1, float* m2, float* packed, const AdamArgs& aa_in,
                                    hipStream_t st, SlabSplit split) {
  bool flat = ((split.lo | split.hi | split.lo3 | split.hi3) & 3) == 0 && split.mask == 0 && !nd.gabor;
  for (int l = 0; l < nd.ND; ++l) flat = flat && nd.L[l].ltype == LT_REAL && nd.L[l].gw_off == nd.L[l].w_off;
  if (!flat) {
    hipError_t e = launch_reduce_slabs(nd, slabs, n_blocks, grads, loss_out, params, packed, st, split);
    if (e != hipSuccess) return e;
    return launch_adam_pack(nd, params, grads, m1, m2, packed, aa_in, st);
  }
  AdamArgs aa = aa_in;
  aa.all_real = 1;
  adam_dead_ranges(nd, aa);
  const int grid = (nd.P + 255) / 256;
  hipLaunchKernelGGL(reduce_adam_real_kernel, dim3(grid + 1), dim3(256), 0, st, nd, slabs, n_blocks, grads, loss_out, split,
                     params, m1, m2, packed, aa);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// Positional_Encoder.embedding 'gauss' (networks.py:30-33): out[r] = [sin(p) | cos(p)],
// p = (2*pi*x_r) @ B^T.  One thread per (row, frequency); writes are coalesced along s.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void encode_gauss_kernel(const float* __restrict__ coords,
                                                           const float* __restrict__ encB, long long B, int E,
                                                           float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * E) return;
  const long long r = idx / E;
  const int s = (int)(idx - r * E);
  const float two_pi = 6.283185307179586f;
  const float x0 = two_pi * coords[3 * r + 0], x1 = two_pi * coords[3 * r + 1], x2 = two_pi * coords[3 * r + 2];
  const float ph = fmaf(x2, encB[3 * s + 2], fmaf(x1, encB[3 * s + 1], x0 * encB[3 * s + 0]));
  float sn, cs;
  sincosf(ph, &sn, &cs);
  out[r * 2 * E + s] = sn;
  out[r * 2 * E + E + s] = cs;
}

// Positional_Encoder.embedding 'LogF' (networks.py:16,24-29): per axis a = 0..2,
// [sin(2 pi x_a b) | cos(2 pi x_a b)] over the nb log-spaced bands b, concatenated: out [B, 6 nb].
__global__ __launch_bounds__(256) void encode_logf_kernel(const float* __restrict__ coords,
                                                          const float* __restrict__ bands, long long B, int nb,
                                                          float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * 3 * nb) return;
  const long long r = idx / (3 * nb);
  const int t = (int)(idx - r * 3 * nb), a = t / nb, s = t - a * nb;
  const float ph = (6.283185307179586f * coords[3 * r + a]) * bands[s];  // (2 pi x_a) @ B^T with K = 1
  float sn, cs;
  sincosf(ph, &sn, &cs);
  float* o = out + r * 6 * nb + a * 2 * nb;
  o[s] = sn;
  o[nb + s] = cs;
}

hipError_t launch_encode_logf(const float* coords, const float* bands, long long B, int nb, float* out,
                              hipStream_t st) {
  const long long n = B * 3 * nb;
  hipLaunchKernelGGL(encode_logf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, coords, bands, B, nb, out);
  return hipGetLastError();
}

hipError_t launch_encode_gauss(const float* coords, const float* encB, long long B, int E, float* out,
                               hipStream_t st) {
  const long long n = B * E;
  const int grid = (int)((n + 255) / 256);
  hipLaunchKernelGGL(encode_gauss_kernel, dim3(grid), dim3(256), 0, st, coords, encB, B, E, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Pointwise losses + their gradient w.r.t. the network output (tier 1; metrics/losses.py and
// train.py:172-182).  64 blocks write ordered partial sums to loss_out[1..64]; a single thread
// then folds them into loss_out[0] (deterministic).
// ---------------------------------------------------------------------------------------------
#define LOSS_BLOCKS 64

__global__ __launch_bounds__(256) void loss_grad_kernel(const LossDesc ld, const float* __restrict__ out,
                                                        const float* __restrict__ gt,
                                                        const uint8_t* __restrict__ mask, long long B,
                                                        float* __restrict__ loss_out, float* __restrict__ dout) {
  __shared__ float red[256];
  // contiguous row range per block, rows visited in order by a fixed thread -> fixed sum order
  const long long per = (B + LOSS_BLOCKS - 1) / LOSS_BLOCKS;
  const long long lo = (long long)blockIdx.x * per;
  const long long hi = lo + per < B ? lo + per : B;
  float acc = 0.f;
  for (long long r = lo + threadIdx.x; r < hi; r += 256) {
    float g[2] = {0.f, 0.f};
    if (mask == nullptr || mask[r] != 0) {
      const float y[2] = {out[2 * r], out[2 * r + 1]};
      const float t[2] = {gt[2 * r], gt[2 * r + 1]};
      acc += loss_row(ld, 2, y, t, g);
    }
    dout[2 * r] = g[0];
    dout[2 * r + 1] = g[1];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[1 + blockIdx.x] = red[0];
}

__global__ void loss_fold_kernel(float* loss_out) {
  float s = 0.f;
  for (int b = 0; b < LOSS_BLOCKS; ++b) s += loss_out[1 + b];
  loss_out[0] = s;
}

hipError_t launch_loss_grad(const LossDesc& ld_in, const float* out, const float* gt, const float* kcoords,
                            const uint8_t* mask, long long B, float* loss_out, float* dout, hipStream_t st) {
  LossDesc ld = ld_in;
  hipLaunchKernelGGL(loss_grad_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, ld, out, gt, mask, B, loss_out, dout);
  hipLaunchKernelGGL(loss_fold_kernel, dim3(1), dim3(1), 0, st, loss_out);
  (void)kcoords;
  return hipGetLastError();
}

// Multi-head version (tier 1 of the multiscale loop, train_kspace_multiscale.py:176-195): outs / douts [NH][B][2],
// pointwise terms on sampled rows, ConsistencyLoss on every row (mfn_loss_row).
__global__ __launch_bounds__(256) void loss_grad_multi_kernel(const LossDesc ld, const float* __restrict__ outs,
                                                              const float* __restrict__ gt,
                                                              const float* __restrict__ dist,
                                                              const uint8_t* __restrict__ mask, int NH, long long B,
                                                              float* __restrict__ loss_out, float* __restrict__ douts) {
  __shared__ float red[256];
  const long long per = (B + LOSS_BLOCKS - 1) / LOSS_BLOCKS;
  const long long lo = (long long)blockIdx.x * per;
  const long long hi = lo + per < B ? lo + per : B;
  float acc = 0.f;
  for (long long r = lo + threadIdx.x; r < hi; r += 256) {
    float y[INR_MAX_HEADS][4], g[INR_MAX_HEADS][4];
#pragma unroll
    for (int k = 0; k < INR_MAX_HEADS; ++k)
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        y[k][o] = (k < NH && o < 2) ? outs[((size_t)k * B + r) * 2 + o] : 0.f;
        g[k][o] = 0.f;
      }
    const float t[4] = {gt[2 * r], gt[2 * r + 1], 0.f, 0.f};
    acc += mfn_loss_row(ld, NH, 2, y, t, dist != nullptr ? dist[r] : 0.f, g, mask == nullptr || mask[r] != 0);
#pragma unroll
    for (int k = 0; k < INR_MAX_HEADS; ++k)
      if (k < NH) {
        douts[((size_t)k * B + r) * 2] = g[k][0];
        douts[((size_t)k * B + r) * 2 + 1] = g[k][1];
      }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[1 + blockIdx.x] = red[0];
}

hipError_t launch_loss_grad_multi(const LossDesc& ld_in, const float* outs, const float* gt, const float* dist,
                                  const uint8_t* mask, int NH, long long B, float* loss_out, float* douts,
                                  hipStream_t st) {
  LossDesc ld = ld_in;
  hipLaunchKernelGGL(loss_grad_multi_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, ld, outs, gt, dist, mask, NH, B,
                     loss_out, douts);
  hipLaunchKernelGGL(loss_fold_kernel, dim3(1), dim3(1), 0, st, loss_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Total-variation regulariser on one coil's predicted k-space image (metrics/losses.py tv_loss;
// train.py:172-175):  w * ( mean|img[:, :-1] - img[:, 1:]| + mean|img[:-1] - img[1:]| ) over
// img [H][W][2].  `out` holds rows [y0, y0 + R) of the image; the first R_own rows are this
// caller's (a trailing halo row lets the vertical pair (y0+R_own-1, y0+R_own) be evaluated by the
// rank that owns its upper row).  Gather form: every element sums the sign terms of the <= 4 pairs
// it belongs to, so there are no atomics and the result is deterministic.  Adds into dout and
// loss_out[0].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void tv_grad_kernel(const float* __restrict__ out, long long R, long long R_own,
                                                      long long W, float cw, float ch,
                                                      float* __restrict__ loss_out, float* __restrict__ dout) {
  __shared__ float red[256];
  const long long n = R * W * 2;
  const long long per = (n + LOSS_BLOCKS - 1) / LOSS_BLOCKS;
  const long long lo = (long long)blockIdx.x * per;
  const long long hi = lo + per < n ? lo + per : n;
  const long long rs = W * 2;  // row stride
  float acc = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const long long r = i / rs;
    const long long x = (i - r * rs) >> 1;
    const float v = out[i];
    float g = 0.f;
    if (r < R_own) {
      if (x + 1 < W) {
        const float d = v - out[i + 2];
        acc += fabsf(d) * cw;
        g += cw * sgn(d);
      }
      if (x > 0) g -= cw * sgn(out[i - 2] - v);
      if (r + 1 < R) {
        const float d = v - out[i + rs];
        acc += fabsf(d) * ch;
        g += ch * sgn(d);
      }
    }
    if (r > 0 && r - 1 < R_own) g -= ch * sgn(out[i - rs] - v);
    dout[i] += g;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[1 + blockIdx.x] = red[0];
}

__global__ void loss_fold_add_kernel(float* loss_out) {
  float s = 0.f;
  for (int b = 0; b < LOSS_BLOCKS; ++b) s += loss_out[1 + b];
  loss_out[0] += s;
}

hipError_t launch_tv_grad(const float* out, long long R, long long R_own, long long W, float cw, float ch,
                          float* loss_out, float* dout, hipStream_t st) {
  hipLaunchKernelGGL(tv_grad_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, out, R, R_own, W, cw, ch, loss_out, dout);
  hipLaunchKernelGGL(loss_fold_add_kernel, dim3(1), dim3(1), 0, st, loss_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// inr_loss_grad + inr_tv_grad of one coil's rows in ONE pass (the per-coil step of train.py:163-189 with use_tv): a thread
// per coordinate, 256 workgroups over the R x W grid (the two kernels above are 64 workgroups each for 6 MB of traffic:
// 18 + 40 us of a 0.58 ms step, plus two single-thread folds and the copy that zeroed the halo rows' mask).  Pointwise loss
// on the sampled rows the caller OWNS (r < R_own: a data-parallel rank's halo row below its slab stays with its owner),
// TV as tv_grad_kernel.  dout is WRITTEN (not added to).  Partial sums: loss_out[1 + block], fixed order within a thread
// (its coordinates in order), fixed tree within a block; loss_tv_fold_kernel sums the 256 partials by a fixed tree.
// ---------------------------------------------------------------------------------------------
#define LOSS_TV_BLOCKS 256

__global__ __launch_bounds__(256) void loss_tv_grad_kernel(const LossDesc ld, const float* __restrict__ out,
                                                           const float* __restrict__ gt, const uint8_t* __restrict__ mask,
                                                           long long R, long long R_own, long long W, float cw, float ch,
                                                           float* __restrict__ loss_out, float* __restrict__ dout) {
  __shared__ float red[256];
  const long long n = R * W;
  const long long per = (n + LOSS_TV_BLOCKS - 1) / LOSS_TV_BLOCKS;
  const long long lo = (long long)blockIdx.x * per;
  const long long hi = lo + per < n ? lo + per : n;
  const float2* __restrict__ o2 = reinterpret_cast<const float2*>(out);
  float acc = 0.f;
  for (long long c = lo + threadIdx.x; c < hi; c += 256) {
    const long long r = c / W, x = c - r * W;
    const float2 v = o2[c];
    float g[2] = {0.f, 0.f};
    if (r < R_own && (mask == nullptr || mask[c] != 0)) {
      const float y[2] = {v.x, v.y};
      const float2 tt = reinterpret_cast<const float2*>(gt)[c];
      const float t[2] = {tt.x, tt.y};
      acc += loss_row(ld, 2, y, t, g);
    }
    if (r < R_own) {
      if (x + 1 < W) {
        const float2 e = o2[c + 1];
        const float d0 = v.x - e.x, d1 = v.y - e.y;
        acc += (fabsf(d0) + fabsf(d1)) * cw;
        g[0] += cw * sgn(d0);
        g[1] += cw * sgn(d1);
      }
      if (x > 0) {
        const float2 e = o2[c - 1];
        g[0] -= cw * sgn(e.x - v.x);
        g[1] -= cw * sgn(e.y - v.y);
      }
      if (r + 1 < R) {
        const float2 e = o2[c + W];
        const float d0 = v.x - e.x, d1 = v.y - e.y;
        acc += (fabsf(d0) + fabsf(d1)) * ch;
        g[0] += ch * sgn(d0);
        g[1] += ch * sgn(d1);
      }
    }
    if (r > 0 && r - 1 < R_own) {
      const float2 e = o2[c - W];
      g[0] -= ch * sgn(e.x - v.x);
      g[1] -= ch * sgn(e.y - v.y);
    }
    reinterpret_cast<float2*>(dout)[c] = float2{g[0], g[1]};
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[1 + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void loss_tv_fold_kernel(float* loss_out) {
  __shared__ float red[256];
  red[threadIdx.x] = loss_out[1 + threadIdx.x];
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = red[0];
}

hipError_t launch_loss_tv_grad(const LossDesc& ld_in, const float* out, const float* gt, const uint8_t* mask, long long R,
                               long long R_own, long long W, float cw, float ch, float* loss_out, float* dout,
                               hipStream_t st) {
  LossDesc ld = ld_in;
  hipLaunchKernelGGL(loss_tv_grad_kernel, dim3(LOSS_TV_BLOCKS), dim3(256), 0, st, ld, out, gt, mask, R, R_own, W, cw, ch,
                     loss_out, dout);
  hipLaunchKernelGGL(loss_tv_fold_kernel, dim3(1), dim3(256), 0, st, loss_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// CenterLoss, random-pair term of one band (metrics/losses.py:175-199): for pairs p < n of rows (a_p, b_p) -- drawn by the
// caller with torch.randperm exactly as the reference draws them -- r_p = (|t_a| - |t_b|) - (|y_a| - |y_b|), the band adds
// w * sum_p r_p^2 (w = 0.1 / n) to the loss and  -2 w r_p y_a / |y_a|  to dout[a_p],  +2 w r_p y_b / |y_b|  to dout[b_p].
// The a rows of a band are distinct, so are the b rows, and no row is in both (they come from disjoint radial masks): plain
// read-modify-write, no atomics; bands are separate launches on one stream.  Rows outside [0, B) are ignored.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void center_pairs_kernel(const float* __restrict__ out, const float* __restrict__ gt,
                                                           const long long* __restrict__ ia, const long long* __restrict__ ib,
                                                           long long n, long long B, float w, float* __restrict__ loss_out,
                                                           float* __restrict__ dout) {
  __shared__ float red[256];
  const long long per = (n + LOSS_BLOCKS - 1) / LOSS_BLOCKS;
  const long long lo = (long long)blockIdx.x * per;
  const long long hi = lo + per < n ? lo + per : n;
  float acc = 0.f;
  for (long long p = lo + threadIdx.x; p < hi; p += 256) {
    const long long a = ia[p], b = ib[p];
    if (a < 0 || a >= B || b < 0 || b >= B) continue;
    const float yar = out[2 * a], yai = out[2 * a + 1], ybr = out[2 * b], ybi = out[2 * b + 1];
    const float na = sqrtf(yar * yar + yai * yai), nb = sqrtf(ybr * ybr + ybi * ybi);
    const float ta = sqrtf(gt[2 * a] * gt[2 * a] + gt[2 * a + 1] * gt[2 * a + 1]);
    const float tb = sqrtf(gt[2 * b] * gt[2 * b] + gt[2 * b + 1] * gt[2 * b + 1]);
    const float r = (ta - tb) - (na - nb);
    acc += w * r * r;
    const float ca = na > 0.f ? -2.f * w * r / na : 0.f;  // d|y|/dy = y / |y| (0 at the origin, as torch.abs)
    const float cb = nb > 0.f ? 2.f * w * r / nb : 0.f;
    dout[2 * a] += ca * yar;
    dout[2 * a + 1] += ca * yai;
    dout[2 * b] += cb * ybr;
    dout[2 * b + 1] += cb * ybi;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[1 + blockIdx.x] = red[0];
}

hipError_t launch_center_pairs(const float* out, const float* gt, const long long* ia, const long long* ib, long long n,
                               long long B, float w, float* loss_out, float* dout, hipStream_t st) {
  hipLaunchKernelGGL(center_pairs_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, out, gt, ia, ib, n, B, w, loss_out, dout);
  hipLaunchKernelGGL(loss_fold_add_kernel, dim3(1), dim3(1), 0, st, loss_out);
  return hipGetLastError();
}

}  // namespace inr
