// inr_w2.h -- layout of the "weights in LDS" images of the bf16 path (inr_siren_bf16_impl.h), shared by the kernel,
// the packing kernel and the host.  A chunk = 64 contraction indices x 256 output rows as 4 (K-steps) x 8 (row
// blocks) MFMA A fragments of 64 lanes x 8 bf16 = 32 KB.  The image lists the chunks in the order a tile consumes
// them: forward layers 0 .. D-1 (the 2-row last layer as one 16-fragment chunk), then the transposed images of
// layers D-1 .. 1.
#pragma once

#define W2_CHUNK_BYTES 32768
#define W2_CHUNK_FLOATS 8192

#if defined(__HIPCC__) || defined(__CUDACC__)
#define INR_HD __host__ __device__
#else
#define INR_HD
#endif

INR_HD inline int w2_nq0(int E) { return 2 * E / 64; }                       // chunks of layer 0 (2E encoder features)
// first forward chunk of layer l: hidden layers take 4 chunks each; the LAST layer (out_features <= 4 rows: one row
// block) takes ONE chunk holding its 16 K-steps x 1 block = 16 fragments (w2_index_last)
INR_HD inline int w2_qf(int l, int D, int E) { return l == 0 ? 0 : w2_nq0(E) + 4 * (l - 1); }
INR_HD inline int w2_qt(int l, int D, int E) {                                // first transposed chunk of layer l >= 1
  const int base = w2_nq0(E) + 4 * (D - 2) + 1;
  return l == D - 1 ? base : base + 1 + 4 * (D - 2 - l);
}
INR_HD inline int w2_nq(int D, int E) { return w2_nq0(E) + 4 * (D - 2) + 1 + 1 + 4 * (D - 2); }  // chunks per tile

// contraction index k of a 256-wide layer -> (K-step t, lane half h, element j): the k order in which an MFMA
// accumulator's registers become the next MFMA's B operand (k = 32 m + 16 s + 8 (j >> 2) + 4 h + (j & 3), t = 2 m + s)
INR_HD inline void w2_kperm_inv(int k, int& t, int& h, int& j) {
  const int m = k >> 5, rem = k & 31, s = rem >> 4, rem2 = rem & 15, rem3 = rem2 & 7;
  t = 2 * m + s;
  h = rem3 >> 2;
  j = ((rem2 >> 3) << 2) | (rem3 & 3);
}
// bf16 element index inside the image of the last layer's forward chunk q: (K-step t of 16, lane, element j)
INR_HD inline long long w2_index_last(int q, int t, int lane, int j) {
  return (((long long)q * 32 + t) * 64 + lane) * 8 + j;
}
// bf16 element index inside the image of (chunk q, K-step s_l, row block mo, lane, element j)
INR_HD inline long long w2_index(int q, int s_l, int mo, int lane, int j) {
  return ((((long long)q * 4 + s_l) * 8 + mo) * 64 + lane) * 8 + j;
}
