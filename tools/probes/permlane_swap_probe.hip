// Probe: what v_permlane16_swap / v_permlane32_swap (gfx950) return through the clang builtins when both operands are the
// same value: prints result[0] and result[1] per lane for v = lane id.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/permlane_swap_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned u = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  out[threadIdx.x] = a[0];
  out[64 + threadIdx.x] = a[1];
  out[128 + threadIdx.x] = b[0];
  out[192 + threadIdx.x] = b[1];
}
int main() {
  unsigned* d;
  unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]"};
  for (int r = 0; r < 4; ++r) {
    printf("%s:", names[r]);
    for (int i = 0; i < 64; ++i) printf(" %u", h[64 * r + i]);
    printf("\n");
  }
  return 0;
}
