// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libconceptattn.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/conceptattn.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CA_WAVE 64

typedef const __attribute__((address_space(1))) void *ca_gptr;
typedef __attribute__((address_space(3))) void *ca_lptr;

// Asynchronous 16-byte-per-lane global -> LDS copy (global_load_lds_dwordx4).  The LDS
// destination is wave-uniform base + lane*16; the global source is per lane.
__device__ __forceinline__ void ca_glds16(const void *gsrc, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((ca_gptr)gsrc, (ca_lptr)lds_wave_base, 16, 0, 0);
}

// The same copy issued from inline asm: hipcc then neither counts it in its own s_waitcnt bookkeeping
// nor orders later LDS reads behind it (with the builtin it drains vmcnt(0) before the next
// ds_read_b64_tr_b16, i.e. right after the DMA was issued).  The caller owns the wait: a counted
// `s_waitcnt vmcnt(N)` and a barrier before any wave reads the bytes.  lds_wave_base must be
// wave-uniform; M0 is saved and restored inside the statement (it is compiler-reserved).
__device__ __forceinline__ void ca_glds16_asm(const void *gsrc, void *lds_wave_base) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(ca_lptr)lds_wave_base);
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(dst)
      : "memory");
}

// Same, in the "scalar base + 32-bit lane offset" addressing form: no 64-bit vector add per copy when the lane
// offsets are loop constants and only the (wave-uniform) base moves.
__device__ __forceinline__ void ca_glds16_asm_s(const void *sbase_uniform, uint32_t lane_byte_off, void *lds_wave_base) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(ca_lptr)lds_wave_base);
  const uint64_t b = (uint64_t)(uintptr_t)sbase_uniform;
  const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const uint64_t base = ((uint64_t)bhi << 32) | blo;
  uint32_t keep;
  // s_nop 4: the base pair comes straight from v_readfirstlane / scalar arithmetic, and an SGPR written by VALU or
  // SALU needs 5 wait states before a VMEM instruction may use it as its address (hipcc pads nothing inside an asm
  // statement; with s_nop 0 here the copy intermittently read from a stale base: a memory fault that came and went
  // with the launch timing).  It also covers the one state between the M0 write and the LDS-DMA.
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 4\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_byte_off), "s"(base), "s"(dst)
      : "memory");
}

__device__ __forceinline__ float ca_bf2f(bf16 x) { return (float)x; }

// pack two floats into one dword of 2 x bf16 (RNE; hipcc emits v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t ca_pack2(float lo, float hi) {
  bf16x2 v = {(bf16)lo, (bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// pack two floats into one dword of 2 x IEEE half (RNE; hipcc emits v_cvt_pk_f16_f32 on gfx950)
__device__ __forceinline__ uint32_t ca_pack2_f16(float lo, float hi) {
  typedef _Float16 ca_h2 __attribute__((ext_vector_type(2)));
  ca_h2 v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float ca_gelu_tanh(float x) {
  // nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).  With
  // 0.5 (1 + tanh(u)) = sigmoid(2u) this is x / (1 + exp(-2u)); exp(-2u) = exp2(x (k0 + k1 x^2)),
  // k0 = -2 sqrt(2/pi) log2(e), k1 = 0.044715 k0: 3 mul/fma + v_exp_f32 + add + v_rcp_f32 + mul
  // (an IEEE divide alone is ~10 instructions).  x -> -inf: exp2 -> inf, rcp -> 0; x -> +inf: exp2 -> 0.
  const float p = __builtin_fmaf(x * x, -0.10294323f, -2.3022082f);
  const float e = __builtin_amdgcn_exp2f(x * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ float ca_silu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950409f * x));
}

// bit of the calling thread's current device, for "done once per device" flags (hipFuncSetAttribute is per device)
inline unsigned long long ca_device_bit() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return 1ull << (dev & 63);
}

// A/B switches read from the environment exist only in diagnostic builds (-DCA_AB_SWITCHES:
// `python -m conceptattention_amd.csrc.build --ab` -> tools/ab/switches/libca.so, used through CA_LIB_PATH by the tools
// under tools/); the product library reads no environment variable and always takes the default.
#ifdef CA_AB_SWITCHES
#include <stdlib.h>
inline int ca_ab_env(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}
#else
inline int ca_ab_env(const char *, int dflt) { return dflt; }
#endif

// host-side plumbing (ca_api.hip): error text; CU count of the current device (cached; -1 if the query fails)
void ca_set_error(const char *fmt, ...);
int ca_cu_count();
