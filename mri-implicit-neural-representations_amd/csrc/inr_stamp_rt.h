// inr_stamp_rt.h -- entry / exit time stamps of diagnostic builds
#pragma once
// Diagnostic builds (-DINR_STAMPS) only: the 100 MHz counter of wave `w` of this workgroup into slot `slot` of its 64-entry
// record in a buffer nothing else reads (tools/stamps*.py: entry / exit of a kernel's waves, prologue and drain times); slots
// below 60 also get the shader-clock counter two slots further on, so that entry -> exit gives the clock the wave ran at.
#ifdef INR_STAMPS
#define INR_RT_STAMP(dbg, cap, nw, w, lane, slot)                                                        \
  do {                                                                                                   \
    const long long rt_at_ = ((long long)blockIdx.x * (nw) + (w)) * 64;                                  \
    if ((dbg) != nullptr && (lane) == 0 && rt_at_ + 63 < (cap)) {                                        \
      (dbg)[rt_at_ + (slot)] = (long long)__builtin_amdgcn_s_memrealtime();                              \
      if ((slot) < 60) (dbg)[rt_at_ + (slot) + 2] = (long long)__builtin_amdgcn_s_memtime(); /* shader clock */ \
    }                                                                                                    \
  } while (0)
#else
#define INR_RT_STAMP(dbg, cap, nw, w, lane, slot) \
  do {                                            \
  } while (0)
#endif

