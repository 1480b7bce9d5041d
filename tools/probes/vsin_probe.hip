// vsin_probe.hip -- accuracy of the bare v_sin_f32 / v_cos_f32 (argument in revolutions) on [0, 1), [-4, 4] and on the
// multiples of 1/256, against double precision; and of v_cvt_pk f32 -> f16.  hipcc --offload-arch=gfx950 -O2; ./vsin_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* s, float* c, float* sf, unsigned* h, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = __builtin_amdgcn_sinf(x[i]);
  c[i] = __builtin_amdgcn_cosf(x[i]);
  sf[i] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x[i]));
  f32x2 v = {x[i], -x[i]};
  h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
int main() {
  std::vector<float> x;
  for (int i = 0; i < 200000; ++i) x.push_back(i / 200000.0f);
  const int n1 = (int)x.size();
  for (int i = 0; i < 200000; ++i) x.push_back(-40.0f + 80.0f * i / 200000.0f);
  const int n2 = (int)x.size();
  for (int i = 0; i < 256; ++i) x.push_back(i / 256.0f);
  const int n = (int)x.size();
  float *dx, *ds, *dc, *dsf; unsigned* dh;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dsf, n * 4); hipMalloc(&dh, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(dx, ds, dc, dsf, dh, n);
  std::vector<float> s(n), c(n), sf(n); std::vector<unsigned> h(n);
  hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(sf.data(), dsf, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost);
  auto rep = [&](const char* name, int a, int b) {
    double ms = 0, mc = 0, mf = 0, rs = 0;
    for (int i = a; i < b; ++i) {
      const double t = 2.0 * M_PI * (double)x[i];
      ms = fmax(ms, fabs(s[i] - sin(t))); mc = fmax(mc, fabs(c[i] - cos(t))); mf = fmax(mf, fabs(sf[i] - sin(t)));
      rs += (s[i] - sin(t)) * (s[i] - sin(t));
    }
    printf("%-22s max |sin err| %.3g  rms %.3g  max |cos err| %.3g  max |sin(fract) err| %.3g\n", name, ms, sqrt(rs / (b - a)), mc, mf);
  };
  rep("[0,1)", 0, n1); rep("[-40,40)", n1, n2); rep("multiples of 1/256", n2, n);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const _Float16 want = (_Float16)x[i];  // host: round to nearest even
    const unsigned short w = __builtin_bit_cast(unsigned short, want);
    if ((h[i] & 0xffff) != w) { if (bad < 5) printf("cvt f16: x=%.9g got %04x want %04x\n", x[i], h[i] & 0xffff, w); ++bad; }
  }
  printf("f32 -> f16 pack: %d of %d differ from round-to-nearest-even\n", bad, n);
  return 0;
}
