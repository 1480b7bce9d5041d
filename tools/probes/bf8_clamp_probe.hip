// bf8_clamp_probe.hip -- does MODE.FP16_OVFL (bit 23) make v_cvt_pk_bf8_f32 saturate (largest finite value instead of inf)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float* x, unsigned* o, float* back, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  unsigned p = 0;
  float a = x[i], b = -x[i];
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\tv_cvt_pk_bf8_f32 %0, %1, %2" : "+v"(p) : "v"(a), "v"(b));
  o[i] = p;
  back[i] = __builtin_amdgcn_cvt_f32_bf8((int)p, 0);
}
int main() {
  const float x[] = {1.3f, 57344.f, 60000.f, 61440.f, 65536.f, 1e9f, INFINITY, NAN, 1e-6f};
  const int n = sizeof(x) / 4;
  float *dx, *db; unsigned* d;
  hipMalloc(&dx, n * 4); hipMalloc(&db, n * 4); hipMalloc(&d, n * 4);
  hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, d, db, n);
  unsigned o[16]; float b[16];
  hipMemcpy(o, d, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b, db, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("x=%-10g bytes %04x back %g\n", x[i], o[i] & 0xffff, b[i]);
  return 0;
}
