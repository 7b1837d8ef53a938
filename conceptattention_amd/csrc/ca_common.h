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

__device__ __forceinline__ float ca_bf2f(bf16 x) { return (float)x; }

// pack two floats into one dword of 2 x bf16 (RNE; hipcc emits v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t ca_pack2(float lo, float hi) {
  bf16x2 v = {(bf16)lo, (bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float ca_gelu_tanh(float x) {
  // nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float e = __expf(2.0f * u);
  const float t = 1.0f - 2.0f / (e + 1.0f);  // tanh(u); e=inf -> 1, e=0 -> -1
  return 0.5f * x * (1.0f + t);
}

__device__ __forceinline__ float ca_silu(float x) { return x / (1.0f + __expf(-x)); }

// host-side error plumbing (ca_api.cpp)
void ca_set_error(const char *fmt, ...);
