// mfma_shape_probe.hip -- would the bf16 weight-gradient GEMM gain from v_mfma_f32_16x16x32_f16 instead of
// v_mfma_f32_32x32x16_f16?  (round-3 VERDICT item 3c; MI355X_MICROARCH.md DVFS (7): in bare, clock-limited loops on random
// data the 16x16x32 shape held 1.12-1.15 x the FLOP/s.)  The GEMM's stage as each wave sees it -- a 64 x 128 output tile, K = 64
// coordinates, operands read from LDS with 24 ds_read_b128, 512-thread workgroups (two waves per SIMD), a barrier per stage --
// in both shapes, same FLOPs, same LDS bytes, same accumulator registers (128), on RANDOM fp16 operands:
//   shape 0: 8 blocks of 32 x 32, four K = 16 steps  -> 32 MFMAs a stage
//   shape 1: 32 blocks of 16 x 16, two K = 32 steps  -> 64 MFMAs a stage
// and with `fill` plain vector instructions + `trans` transcendental ones per 32 x 32 MFMA's worth of work interleaved (the
// GEMM carries ~6 vector instructions per MFMA, one of them a v_sin_f16).  Prints TFLOP/s by wall clock (hipEvents) and the
// in-kernel clock (s_memtime / s_memrealtime) after 2 s of launches.
//   hipcc --offload-arch=gfx950 -O2 mfma_shape_probe.hip -o mfma_shape_probe && ./mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PITCH = 72;                 // halves per LDS row (64 + 8: conflict-free ds_read_b128)
constexpr int TILE = 256 * PITCH;         // one operand tile: 256 rows x 64 k
constexpr int STAGES = 2000;

template <int FILL, int TRANS>
__device__ __forceinline__ void filler(float (&v)[4]) {
#pragma unroll
  for (int i = 0; i < TRANS; ++i) asm volatile("v_sin_f32 %0, %0" : "+v"(v[i & 3]));
#pragma unroll
  for (int i = 0; i < FILL; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 3]) : "v"(v[(i + 1) & 3]));
}

template <int SHAPE, int FILL, int TRANS>
__global__ __launch_bounds__(512) void probe(const _Float16* __restrict__ src, float* __restrict__ out, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  for (int i = t; i < 2 * TILE / 8; i += 512)  // random operands, once
    reinterpret_cast<f16x8*>(lds)[i] = reinterpret_cast<const f16x8*>(src)[(blockIdx.x * 131 + i) % (2 * TILE / 8)];
  __syncthreads();
  float v[4] = {0.1f * lane, 0.2f, 0.3f, 0.4f};
  long long c0 = 0, r0 = 0;
  if (lane == 0) c0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
  if constexpr (SHAPE == 0) {
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const _Float16* As = lds + (wm * 64 + (lane & 31)) * PITCH + 8 * (lane >> 5);
    const _Float16* Bs = lds + TILE + (wn * 128 + (lane & 31)) * PITCH + 8 * (lane >> 5);
    for (int s = 0; s < STAGES; ++s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f16x8 A[2], B[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) A[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * PITCH + 16 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) B[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * PITCH + 16 * q);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[j], acc[i][j], 0, 0, 0);
            filler<FILL, TRANS>(v);
          }
      }
      __syncthreads();
    }
    float sum = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 512 + t] = sum;
  } else {
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // 16 x 16 x 32: lane (row = lane & 15, k group = lane >> 4) holds 8 consecutive k
    const _Float16* As = lds + (wm * 64 + (lane & 15)) * PITCH + 8 * (lane >> 4);
    const _Float16* Bs = lds + TILE + (wn * 128 + (lane & 15)) * PITCH + 8 * (lane >> 4);
    for (int s = 0; s < STAGES; ++s) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f16x8 A[4], B[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) A[i] = *reinterpret_cast<const f16x8*>(As + i * 16 * PITCH + 32 * q);
#pragma unroll
        for (int j = 0; j < 8; ++j) B[j] = *reinterpret_cast<const f16x8*>(Bs + j * 16 * PITCH + 32 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i], B[j], acc[i][j], 0, 0, 0);
            if ((j & 1) == 1) filler<FILL, TRANS>(v);  // the same vector work per FLOP as shape 0
          }
      }
      __syncthreads();
    }
    float sum = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 512 + t] = sum;
  }
  if (lane == 0) {
    const long long c1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
    stamps[(blockIdx.x * 8 + w) * 2 + 0] = c1 - c0;
    stamps[(blockIdx.x * 8 + w) * 2 + 1] = r1 - r0;
  }
}

template <int SHAPE, int FILL, int TRANS>
static void run(const char* name, const _Float16* src, float* out, long long* stamps) {
  const size_t lds_bytes = (size_t)2 * TILE * sizeof(_Float16);
  auto k = probe<SHAPE, FILL, TRANS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f, warm = 0.f;
  while (warm < 2000.f) {  // >= 2 s of back-to-back launches first
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), lds_bytes, 0, src, out, stamps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    warm += ms;
  }
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), lds_bytes, 0, src, out, stamps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> st(256 * 8 * 2);
  hipMemcpy(st.data(), stamps, st.size() * sizeof(long long), hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  double cyc = 0;
  for (int i = 0; i < 256 * 8; ++i) ghz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1), cyc += (double)st[2 * i];
  std::sort(ghz.begin(), ghz.end());
  const double flop = 2.0 * 64 * 128 * 64 * 8 * 256 * (double)STAGES;  // per launch
  printf("%-34s %7.1f us/launch  %7.1f TFLOP/s by wall   in-kernel clock %.3f GHz (median)  %8.0f cycles/stage/wave\n", name,
         ms / 20 * 1e3, flop / (ms / 20 * 1e-3) / 1e12, ghz[ghz.size() / 2], cyc / (256 * 8) / STAGES);
}

int main() {
  _Float16* src;
  float* out;
  long long* stamps;
  const size_t n = 2 * TILE;
  std::vector<_Float16> h(n);
  srand(1);
  for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  hipMalloc(&src, n * sizeof(_Float16));
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&stamps, 256 * 8 * 2 * sizeof(long long));
  hipMemcpy(src, h.data(), n * sizeof(_Float16), hipMemcpyHostToDevice);
  printf("per wave and stage: 64 x 128 x 64 tile, 24 ds_read_b128, one barrier; 8 waves a workgroup, 256 workgroups, random fp16\n");
  run<0, 0, 0>("32x32x16, MFMAs + reads only", src, out, stamps);
  run<1, 0, 0>("16x16x32, MFMAs + reads only", src, out, stamps);
  run<0, 5, 1>("32x32x16, + 5 plain + 1 trans / MFMA", src, out, stamps);
  run<1, 5, 1>("16x16x32, same vector work per FLOP", src, out, stamps);
  run<0, 2, 0>("32x32x16, + 2 plain / MFMA", src, out, stamps);
  run<1, 2, 0>("16x16x32, same vector work per FLOP", src, out, stamps);
  return 0;
}
