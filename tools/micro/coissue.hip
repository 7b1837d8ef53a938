// Microbenchmark: do an MFMA-only wave and a VALU-only wave on the SAME SIMD overlap?
// 512-thread workgroups (2 waves per SIMD), 256 workgroups.  mode 0: waves 0-3 MFMA, 4-7 idle;
// mode 1: 0-3 idle, 4-7 VALU; mode 2: 0-3 MFMA + 4-7 VALU; mode 3: all 8 waves MFMA; mode 4: all VALU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(int mode, int iters, float *out, unsigned long long *cyc) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0 || mode == 2 || mode == 3) && (wave < 4 || mode == 3);
  const bool do_valu = (mode == 1 || mode == 2 || mode == 4) && (wave >= 4 || mode == 4);
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x ^ j)); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float acc_out = 0.f;
  if (do_mfma) {
    if (SHAPE == 32) {
      f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
      acc_out = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
      f32x4 c[8] = {};
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
      for (int j = 0; j < 8; ++j) acc_out += c[j][0];
    }
  }
  if (do_valu) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; ++i) {  // per iteration: 4 exp + 12 fma  (~ the softmax mix)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        x0 = __builtin_amdgcn_exp2f(x0 * 0.5f - 1.0f); x1 = __builtin_amdgcn_exp2f(x1 * 0.5f - 1.0f);
        x0 = fmaf(x0, 0.9f, 0.1f); x1 = fmaf(x1, 0.9f, 0.1f); x2 = fmaf(x2, 0.9f, x0); x3 = fmaf(x3, 0.9f, x1);
        x2 = fmaf(x2, 0.5f, 0.2f); x3 = fmaf(x3, 0.5f, 0.2f);
      }
    }
    acc_out += x0 + x1 + x2 + x3;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = acc_out;
  if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

int main() {
  float *out; unsigned long long *cyc, h[8];
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  const int iters = 20000;
  for (int shape : {32, 16}) {
    for (int mode = 0; mode < 5; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
        else hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      const double mf = (shape == 32 ? 4.0 : 8.0) * iters;
      printf("shape %d mode %d: wave0 %.1f cyc/MFMA (%llu cyc), wave4 %.2f cyc/valu-iter (%llu cyc)\n", shape, mode,
             h[0] / mf, h[0], (double)h[4] / iters, h[4]);
    }
  }
  return 0;
}
