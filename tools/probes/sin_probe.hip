// Diagnostic (not part of the library): accuracy of the hardware v_sin_f32 / v_cos_f32 behind a two-constant
// Cody-Waite reduction, against double precision, next to the polynomial sincos_cw the fp32 kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 -o sin_probe tools/probes/sin_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../mri-implicit-neural-representations_amd/csrc/inr_device.h"

__device__ __forceinline__ void sincos_hw(float x, float& s, float& c) {
  const float k = rintf(x * 0.15915494309189535f);           // revolutions
  float r = fmaf(k, -6.2831854820251465f, x);                 // 2 pi hi (float(2 pi))
  r = fmaf(k, 1.7484555e-7f, r);                              // -(2 pi lo): float(2pi) - 2pi = 1.7484555e-7
  const float rr = r * 0.15915494309189535f;                  // |rr| <= 0.5
  s = __builtin_amdgcn_sinf(rr);
  c = __builtin_amdgcn_cosf(rr);
}

__global__ void probe(const float* x, int n, float* o) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s, c, s2, c2;
  sincos_hw(x[i], s, c);
  sincos_cw(x[i], s2, c2);
  o[4 * i] = s; o[4 * i + 1] = c; o[4 * i + 2] = s2; o[4 * i + 3] = c2;
}

int main() {
  for (double range : {0.5, 3.2, 30.0, 300.0}) {
    const int n = 1 << 22;
    std::vector<float> hx(n), ho(4 * (size_t)n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
      st ^= st << 13; st ^= st >> 7; st ^= st << 17;
      hx[i] = (float)(((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0) * range);
    }
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, (size_t)n * 16);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<n / 256, 256>>>(dx, n, dout);
    hipMemcpy(ho.data(), dout, (size_t)n * 16, hipMemcpyDeviceToHost);
    double mx[4] = {0, 0, 0, 0}, sm[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
      const double rs = sin((double)hx[i]), rc = cos((double)hx[i]);
      const double e[4] = {fabs(ho[4 * i] - rs), fabs(ho[4 * i + 1] - rc), fabs(ho[4 * i + 2] - rs), fabs(ho[4 * i + 3] - rc)};
      for (int k = 0; k < 4; ++k) { mx[k] = fmax(mx[k], e[k]); sm[k] += e[k]; }
    }
    printf("range +-%g: hw sin max %.3e mean %.3e | hw cos max %.3e mean %.3e | poly sin max %.3e mean %.3e | poly cos max %.3e mean %.3e\n",
           range, mx[0], sm[0] / n, mx[1], sm[1] / n, mx[2], sm[2] / n, mx[3], sm[3] / n);
    hipFree(dx); hipFree(dout);
  }
  return 0;
}
