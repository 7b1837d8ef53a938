// Microbenchmark: v_mfma_f32_32x32x16_bf16 issue rate vs accumulator-chain length and operand data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void k(const __bf16 *data, int iters, int active_waves, float *out, unsigned long long *cyc) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *(const bf16x8 *)(data + ((threadIdx.x * 8 + i * 4096) & 32767));
    b[i] = *(const bf16x8 *)(data + ((threadIdx.x * 8 + i * 4096 + 16384) & 32767));
  }
  f32x16 c[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < active_waves) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u >> 1) & 3], c[u % NACC], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int j = 0; j < NACC; ++j) s += c[j][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

int main() {
  __bf16 *data; float *out; unsigned long long *cyc, h[8];
  hipMalloc(&data, 65536); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  unsigned short hd[32768];
  const int iters = 5000;
  for (int rnd = 0; rnd < 2; ++rnd) {
    for (int i = 0; i < 32768; ++i) {
      float v = rnd ? ((rand() / (float)RAND_MAX) * 2 - 1) : 0.001f;
      unsigned u; memcpy(&u, &v, 4); hd[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(data, hd, 65536, hipMemcpyHostToDevice);
    for (int nacc : {1, 2, 4}) for (int aw : {4, 8}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, data, iters, aw, out, cyc);
        if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, data, iters, aw, out, cyc);
        if (nacc == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, data, iters, aw, out, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      const double flop = 256.0 * aw * 8.0 * iters * 32768.0;
      printf("%s data, %d accumulators, %d MFMA waves/WG: %.1f ticks per MFMA per wave, kernel %.3f ms = %.0f TFLOP/s, ticks/us %.0f\n",
             rnd ? "random" : "const", nacc, aw, h[0] / (8.0 * iters), ms, flop / ms / 1e9, h[0] / (ms * 1e3));
    }
  }
  return 0;
}
