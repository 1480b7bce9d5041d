// fmt8_probe.hip -- semantics the 8-bit stash of the bf16 path relies on, checked on the device:
//  (1) v_add_f32_sdwa dst_sel:BYTE_n writes the LOW 8 bits of the fp32 sum (phase byte = round(t*256) mod 256 by the
//      magic-number add t + 1.5*2^15);  (2) v_cvt_pk_bf8_f32: rounding (nearest even), overflow, subnormals;
//  (3) a bf8 byte in the high byte of a half word IS the fp16 of the same value.
// build: hipcc --offload-arch=gfx950 -O2 fmt8_probe.hip -o fmt8_probe ; run: ./fmt8_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_phase(const float* t, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned packed = 0xdeadbeefu;
  const float magic = 49152.0f;
  float a = t[i];
  asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
               : "+v"(packed) : "v"(a), "v"(magic));
  out[i] = packed;
}
__global__ void k_bf8(const float* x, unsigned* out, float* back, _Float16* h, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int p = __builtin_amdgcn_cvt_pk_bf8_f32(x[i], 0.f, 0, false);
  out[i] = (unsigned)p;
  back[i] = __builtin_amdgcn_cvt_f32_bf8(p, 0);
  const unsigned short hb = (unsigned short)((p & 0xff) << 8);
  h[i] = __builtin_bit_cast(_Float16, hb);
}

int main() {
  int bad = 0;
  {  // phase bytes
    std::vector<float> t;
    for (int i = -70000; i <= 70000; i += 7) t.push_back(i / 256.0f * 0.37f + 0.001f * (i % 13));
    t.push_back(255.998f); t.push_back(-255.998f); t.push_back(0.f); t.push_back(0.5f / 256); t.push_back(1.5f / 256);
    const int n = (int)t.size();
    float* dt; unsigned* dout;
    hipMalloc(&dt, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dt, t.data(), n * 4, hipMemcpyHostToDevice);
    k_phase<<<(n + 255) / 256, 256>>>(dt, dout, n);
    std::vector<unsigned> o(n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
      const long r = lrintf(t[i] * 256.0f);  // nearest even, as the fp32 add
      const unsigned want = (unsigned)(r & 255);
      const unsigned got = (o[i] >> 8) & 255;
      const bool keep = (o[i] & 0xffff00ffu) == (0xdeadbeefu & 0xffff00ffu);
      if (got != want || !keep) { if (bad < 10) printf("phase: t=%g want %u got %u word %08x\n", t[i], want, got, o[i]); ++bad; }
    }
    printf("phase bytes: %d values, %d bad\n", n, bad);
  }
  {  // bf8
    std::vector<float> x = {0.f, 1.f, 1.125f, 1.25f, 1.375f, 1.5f, 1.625f, 1.75f, 1.875f, 2.f, -1.3f, 3.1f, 57344.f, 60000.f,
                            65536.f, 1e6f, 1e30f, INFINITY, -INFINITY, NAN, 6.1035e-5f, 3.05e-5f, 1.5259e-5f, 7.6e-6f,
                            1e-6f, -7.7e-6f, 40000.f, 49152.f, 53248.f, 61439.f, 61440.f, 61441.f};
    const int n = (int)x.size();
    float *dx, *db; unsigned* dout; _Float16* dh;
    hipMalloc(&dx, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dh, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k_bf8<<<1, 64>>>(dx, dout, db, dh, n);
    std::vector<unsigned> o(n); std::vector<float> b(n); std::vector<_Float16> h(n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h.data(), dh, n * 2, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
      printf("bf8: x=%-12g byte=%02x back=%-12g as_fp16=%-12g %s\n", x[i], o[i] & 255, b[i], (float)h[i],
             (b[i] == (float)h[i] || (std::isnan(b[i]) && std::isnan((float)h[i]))) ? "" : "MISMATCH");
  }
  return bad != 0;
}
