// overlap_probe.hip -- do the matrix pipe and the vector ALU of one SIMD overlap across the two waves of a 512-thread
// workgroup the way the dW GEMM's staggered stages assume?  Per iteration a wave issues 32 v_mfma_f32_32x32x16_f16 and a
// block of 108 plain + 32 transcendental vector instructions (the staging of one GEMM stage), then a barrier.
//   mode 0  MFMA block only            mode 1  vector block only
//   mode 2  every wave: MFMA block, then vector block (lockstep)
//   mode 3  waves 0-3 MFMA first, waves 4-7 vector first        mode 4  odd waves vector first
//   mode 5  waves 2,3,6,7 vector first                           mode 6  each wave interleaves 1 MFMA : 4-5 vector ops
//   mode 7  as 3, MFMA operands read from LDS (24 ds_read_b128 per block, rows pitched 144 B)
//   mode 8  as 7, the vector block ends its four quarters with two ds_write_b128 each     mode 9  as 8, lockstep
//   mode 10 as 8 without the transcendental ops' results feeding the stores (stores of constants)
// Prints ns per iteration and the SIMD each wave of workgroup 0 ran on (HW_ID bits 5:4).
// hipcc --offload-arch=gfx950 -O2 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_block(f32x16 (&acc)[8], const f16x8& a, const f16x8& b) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
}
__device__ __forceinline__ void valu_block(float (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_sin_f32 %0, %0" : "+v"(v[i]));
      asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_add_f32 %0, %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 7]));
    }
    asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %0\n v_add_f32 %0, %0, %1" : "+v"(v[0]), "+v"(v[1]));
  }
}

__device__ __forceinline__ void mfma_block_lds(f32x16 (&acc)[8], const _Float16* As, const _Float16* Bs) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f16x8 A[2], B[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) A[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * 72 + 16 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) B[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * 72 + 16 * q);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[j], acc[i * 4 + j], 0, 0, 0);
  }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool REAL>
__device__ __forceinline__ void valu_block_w(float (&v)[8], _Float16* st) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_sin_f32 %0, %0" : "+v"(v[i]));
      asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_add_f32 %0, %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 7]));
    }
    asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %0\n v_add_f32 %0, %0, %1" : "+v"(v[0]), "+v"(v[1]));
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    if (!REAL) a = f32x4{1.f, 2.f, 3.f, 4.f}, b = f32x4{5.f, 6.f, 7.f, 8.f};
    *reinterpret_cast<f32x4*>(st + r * 72) = a;
    *reinterpret_cast<f32x4*>(st + 256 * 72 + r * 72) = b;
  }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned* hw, int iters) {
  const int w = threadIdx.x >> 6;
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int lane = threadIdx.x & 63, li = lane & 31, hf = lane >> 5;
  const _Float16* As = lds + ((w >> 1) * 64 + li) * 72 + 8 * hf;
  const _Float16* Bs = lds + 256 * 72 + ((w & 1) * 128 + li) * 72 + 8 * hf;
  _Float16* st = lds + (4 * (threadIdx.x >> 3)) * 72 + 8 * (threadIdx.x & 7);
  if (MODE >= 7) {
    for (int i = threadIdx.x; i < 4 * 256 * 72; i += 512) lds[i] = (_Float16)(i * 1e-4f);
    __syncthreads();
  }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(threadIdx.x * 0.001f + j), b[j] = (_Float16)(j * 0.01f);
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const bool first = (MODE == 3 || MODE == 7 || MODE == 8 || MODE == 10) ? w >= 4 : MODE == 4 ? (w & 1) : MODE == 5 ? ((w >> 1) & 1) : false;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    hw[w] = id;
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      mfma_block(acc, a, b);
    } else if (MODE == 1) {
      valu_block(v);
    } else if (MODE == 6) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
          asm volatile("v_sin_f32 %0, %0" : "+v"(v[i]));
          asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %0\n v_add_f32 %0, %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 7]));
        }
        asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %0\n v_add_f32 %0, %0, %1" : "+v"(v[0]), "+v"(v[1]));
      }
    } else if (MODE >= 7) {
      const int cur = (it & 1) * 2 * 256 * 72, nxt = ((it + 1) & 1) * 2 * 256 * 72;
      if (first) {
        if (MODE == 7) valu_block(v); else valu_block_w<MODE != 10>(v, st + nxt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block_lds(acc, As + cur, Bs + cur);
      } else {
        mfma_block_lds(acc, As + cur, Bs + cur);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 7) valu_block(v); else valu_block_w<MODE != 10>(v, st + nxt);
      }
    } else {
      if (first) {
        valu_block(v);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(acc, a, b);
      } else {
        mfma_block(acc, a, b);
        __builtin_amdgcn_sched_barrier(0);
        valu_block(v);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][5] + v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
static void run(float* out, unsigned* hw, const char* what) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const size_t lds_bytes = MODE >= 7 ? 4 * 256 * 72 * 2 : 0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<MODE><<<blocks, 512, lds_bytes>>>(out, hw, 200);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 512, lds_bytes>>>(out, hw, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned h[8];
  hipMemcpy(h, hw, sizeof(h), hipMemcpyDeviceToHost);
  printf("mode %d %-44s %8.1f ns/iter   SIMD of waves 0-7:", MODE, what, ms * 1e6 / iters);
  for (int i = 0; i < 8; ++i) printf(" %u", (h[i] >> 4) & 3);
  printf("\n");
}

int main() {
  float* out;
  unsigned* hw;
  hipMalloc(&out, 256 * 512 * 4), hipMalloc(&hw, 64);
  run<0>(out, hw, "32 MFMA");
  run<1>(out, hw, "108 VALU + 32 sin");
  run<2>(out, hw, "lockstep MFMA, VALU");
  run<3>(out, hw, "waves 4-7 VALU first");
  run<4>(out, hw, "odd waves VALU first");
  run<5>(out, hw, "waves 2,3,6,7 VALU first");
  run<6>(out, hw, "own-wave interleave");
  run<7>(out, hw, "stagger, MFMA operands from LDS");
  run<8>(out, hw, "stagger, LDS operands + 8 ds_write_b128");
  run<9>(out, hw, "lockstep, LDS operands + 8 ds_write_b128");
  run<10>(out, hw, "stagger, LDS operands + 8 constant stores");
  run<0>(out, hw, "32 MFMA (again)");
  return 0;
}
