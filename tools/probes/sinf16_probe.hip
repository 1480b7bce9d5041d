// sinf16_probe.hip -- v_sin_f16 / v_cos_f16 on the 256 phases p / 256 handed over as the fp16 value 4 + p / 256 (bits
// 0x4400 | p: in [4, 8) an fp16 ulp is 1/256, so a phase BYTE becomes the sine's argument in revolutions with one v_perm_b32
// and no conversion), against fp16(sin(2 pi p / 256)) rounded to nearest.   hipcc --offload-arch=gfx950 -O2; ./sinf16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(unsigned* s, unsigned* c) {
  const unsigned p = threadIdx.x;
  const unsigned x = 0x44004400u | p | (((p + 1) & 255u) << 16);  // (p, p+1)
  unsigned so = 0, co = 0;
  asm volatile("v_sin_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(so) : "v"(x));
  asm volatile("v_sin_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(so) : "v"(x));
  asm volatile("v_cos_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(co) : "v"(x));
  asm volatile("v_cos_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(co) : "v"(x));
  s[p] = so, c[p] = co;
}
static float h2f(unsigned short h) { _Float16 v = __builtin_bit_cast(_Float16, h); return (float)v; }
int main() {
  unsigned *ds, *dc, s[256], c[256];
  hipMalloc(&ds, 1024), hipMalloc(&dc, 1024);
  k<<<1, 256>>>(ds, dc);
  hipMemcpy(s, ds, 1024, hipMemcpyDeviceToHost), hipMemcpy(c, dc, 1024, hipMemcpyDeviceToHost);
  int bad_s = 0, bad_c = 0, bad_hi = 0;
  double es = 0, ec = 0;
  for (int p = 0; p < 256; ++p) {
    const double ts = sin(2 * M_PI * p / 256.0), tc = cos(2 * M_PI * p / 256.0);
    const unsigned short ws = __builtin_bit_cast(unsigned short, (_Float16)ts), wc = __builtin_bit_cast(unsigned short, (_Float16)tc);
    const unsigned short gs = s[p] & 0xffff, gc = c[p] & 0xffff;
    es = fmax(es, fabs(h2f(gs) - ts)), ec = fmax(ec, fabs(h2f(gc) - tc));
    if (gs != ws) { if (bad_s < 6) printf("sin p=%d got %04x (%g) want %04x (%g)\n", p, gs, h2f(gs), ws, h2f(ws)); ++bad_s; }
    if (gc != wc) { if (bad_c < 6) printf("cos p=%d got %04x (%g) want %04x (%g)\n", p, gc, h2f(gc), wc, h2f(wc)); ++bad_c; }
    if ((s[p] >> 16) != (s[(p + 1) & 255] & 0xffff) || (c[p] >> 16) != (c[(p + 1) & 255] & 0xffff)) ++bad_hi;
  }
  printf("v_sin_f16: %d of 256 differ from fp16 round-to-nearest, max |err| %.3g;  v_cos_f16: %d, max |err| %.3g;  high-half mismatches %d\n",
         bad_s, es, bad_c, ec, bad_hi);
  return 0;
}
