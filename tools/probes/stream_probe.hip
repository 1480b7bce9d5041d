// stream_probe.hip -- the fused bf16 kernel's hidden-layer stream in isolation: eight waves (two per SIMD), an interval =
// 32 slots of  [ds_read_b128 of an A fragment 6 slots ahead][v_mfma_f32_32x32x16_bf16][NV vector instructions, one of them
// v_sin_f32], a barrier per interval; the B operands sit in registers (16 fragments).  Which ingredient costs what?
//   mode 0  MFMAs only (one accumulator chain per 16 slots), operands in registers
//   mode 1  + fragment reads from LDS           mode 2  + 5 vector instructions per slot
//   mode 3  as 2 with 8 vector instructions     mode 4  as 2, MFMAs on 4 rotating accumulators (no dependent chain)
//   mode 5  as 2 without the fragment reads     mode 6  as 2 with a buffer_store_dword every 4th slot
//   mode 7  as 2, one wave per SIMD (256 threads)
//   mode 8  as 2, the five vector instructions of a slot independent of each other (each continues a chain that the slot
//           before left: software-pipelined slices)        mode 9  as 8 with 8 vector instructions
// Prints ns and cycles (s_memtime) per interval: 2 x 32 x 32 = 2048 cycles is the matrix pipe's time at two waves a SIMD.
// hipcc --offload-arch=gfx950 -O2 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* cyc, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 255);
  __syncthreads();
  u32x4 b[16];
  for (int t = 0; t < 16; ++t)
    for (int e = 0; e < 4; ++e) b[t][e] = 0x3c003c00u + (unsigned)(lane * 7 + t * 13 + e);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[6];
  for (int i = 0; i < 6; ++i) v[i] = 0.01f * lane + i;
  constexpr bool READS = MODE != 0 && MODE != 5;
  constexpr int NV = MODE == 0 || MODE == 1 ? 0 : ((MODE == 3 || MODE == 9) ? 8 : 5);
  constexpr bool INDEP = MODE == 8 || MODE == 9;
  float u[8];
  for (int i = 0; i < 8; ++i) u[i] = 0.02f * lane + i;
  const char* base = lds + lane * 16;
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    bf16x8 A[32];
#pragma unroll
    for (int q = 0; q < 6; ++q) A[q] = READS ? *reinterpret_cast<const bf16x8*>(base + q * 1024) : __builtin_bit_cast(bf16x8, b[q & 15]);
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      if (q + 6 < 32) A[q + 6] = READS ? *reinterpret_cast<const bf16x8*>(base + (q + 6) * 1024) : __builtin_bit_cast(bf16x8, b[(q + 6) & 15]);
      const int ai = MODE == 4 ? (q & 3) : (q >> 4);
      acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q], __builtin_bit_cast(bf16x8, b[q & 15]), acc[ai], 0, 0, 0);
      if (INDEP) {
        // five chains, one instruction of each per slot: no instruction reads what another of the same slot wrote
        asm volatile("v_add_f32 %0, %0, %5\n v_sin_f32 %1, %1\n v_fma_f32 %2, %2, %5, %2\n v_add_f32 %3, %3, %5\n v_fma_f32 %4, %4, %5, %4"
                     : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]) : "v"(v[5]));
        if (NV >= 8) asm volatile("v_add_f32 %0, %0, %3\n v_fma_f32 %1, %1, %3, %1\n v_add_f32 %2, %2, %3" : "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(v[5]));
      } else if (NV >= 5) {
        asm volatile("v_add_f32 %0, %0, %1\n v_sin_f32 %2, %0\n v_fma_f32 %1, %2, %1, %0\n v_add_f32 %3, %3, %2\n v_fma_f32 %4, %2, %3, %4"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]));
      }
      if (!INDEP && NV >= 8) asm volatile("v_add_f32 %0, %0, %1\n v_fma_f32 %1, %0, %1, %2\n v_add_f32 %2, %2, %0" : "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
      if (MODE == 6 && (q & 3) == 3) sink[(size_t)blockIdx.x * 512 + threadIdx.x] = v[0];
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 6; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  (void)w;
}

template <int MODE>
static void run(float* out, long long* cyc, float* sink, const char* what, int threads = 512) {
  const int iters = 1000, blocks = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  k<MODE><<<blocks, threads, 32 * 1024>>>(out, cyc, 100, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, threads, 32 * 1024>>>(out, cyc, iters, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long c[256];
  (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += (double)c[i];
  m /= 256.0 * iters;
  printf("mode %d %-52s %8.1f ns  %8.0f cycles per interval  (%.2f GHz)\n", MODE, what, ms * 1e6 / iters, m, m / (ms * 1e6 / iters));
}

int main() {
  float *out, *sink;
  long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4), (void)hipMalloc(&sink, 256 * 512 * 4), (void)hipMalloc(&cyc, 256 * 8);
  run<0>(out, cyc, sink, "MFMA chain only");
  run<1>(out, cyc, sink, "+ fragment reads");
  run<2>(out, cyc, sink, "+ 5 vector instructions a slot");
  run<3>(out, cyc, sink, "+ 8 vector instructions a slot");
  run<4>(out, cyc, sink, "5 vector, 4 rotating accumulators");
  run<5>(out, cyc, sink, "5 vector, no fragment reads");
  run<6>(out, cyc, sink, "5 vector + a store every 4th slot");
  run<8>(out, cyc, sink, "5 INDEPENDENT vector instructions a slot");
  run<9>(out, cyc, sink, "8 INDEPENDENT vector instructions a slot");
  run<8>(out, cyc, sink, "5 independent, ONE wave per SIMD", 256);
  run<2>(out, cyc, sink, "5 vector, ONE wave per SIMD", 256);
  run<0>(out, cyc, sink, "MFMA chain only, ONE wave per SIMD", 256);
  return 0;
}
