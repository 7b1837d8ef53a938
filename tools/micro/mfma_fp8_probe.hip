// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands) on gfx950: (1) which (row, k) does byte j of
// lane l's A / B fragment hold, checked with exact small-integer data against a host product for a few
// candidate maps; (2) what an E8M0 scale byte does; (3) issue rate next to v_mfma_f32_16x16x32_bf16.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_fp8_probe mfma_fp8_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void one_mfma(const v8i *a, const v8i *b, float *c, int scale_a, int scale_b) {
  const int l = threadIdx.x;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, scale_a, 0, scale_b);
  for (int r = 0; r < 4; ++r) c[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];  // row = 4*(l>>4)+r, col = l&15
}

template <int FP8>
__global__ void rate(float *out, int iters) {
  v4f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838 ^ (threadIdx.x << 3); }
  bf16x8 ha = __builtin_bit_cast(bf16x8, v4i{a[0], a[1], a[2], a[3]});
  bf16x8 hb = __builtin_bit_cast(bf16x8, v4i{b[0], b[1], b[2], b[3]});
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (FP8) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

// e4m3fn encoding of small integers -8..8 (exact)
static uint8_t e4m3(int v) {
  if (v == 0) return 0;
  uint8_t s = v < 0 ? 0x80 : 0;
  int a = abs(v), e = 0;
  while ((1 << (e + 1)) <= a) ++e;                 // a in [2^e, 2^(e+1))
  int mant = ((a << 3) >> e) & 7;                  // 3 mantissa bits (exact for a <= 15 when low bits are 0)
  return s | (uint8_t)((e + 7) << 3) | (uint8_t)mant;
}

int main() {
  const int M = 16, N = 16, K = 128;
  std::vector<int> A(M * K), B(N * K);
  srand(1);
  for (auto &x : A) x = rand() % 5 - 2;
  for (auto &x : B) x = rand() % 7 - 3;
  std::vector<float> ref(M * N);
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      int s = 0;
      for (int k = 0; k < K; ++k) s += A[m * K + k] * B[n * K + k];
      ref[m * N + n] = (float)s;
    }
  v8i *da, *db;
  float *dc;
  hipMalloc(&da, 64 * 32);
  hipMalloc(&db, 64 * 32);
  hipMalloc(&dc, 256 * 4);
  const char *names[] = {"k = 32*(l>>4) + j", "k = 16*(l>>4) + (j&15) + 64*(j>>4)", "k = 8*(l>>4) + (j&7) + 32*(j>>3)",
                         "k = 4*j + (l>>4)"};
  for (int h = 0; h < 4; ++h) {
    uint8_t fa[64][32], fb[64][32];
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j) {
        int g = l >> 4, k;
        if (h == 0) k = 32 * g + j;
        else if (h == 1) k = 16 * g + (j & 15) + 64 * (j >> 4);
        else if (h == 2) k = 8 * g + (j & 7) + 32 * (j >> 3);
        else k = 4 * j + g;
        fa[l][j] = e4m3(A[(l & 15) * K + k]);
        fb[l][j] = e4m3(B[(l & 15) * K + k]);
      }
    hipMemcpy(da, fa, sizeof(fa), hipMemcpyHostToDevice);
    hipMemcpy(db, fb, sizeof(fb), hipMemcpyHostToDevice);
    // operand order: the builtin's first operand indexes the C ROW?  test both readings
    hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, 0x7f7f7f7f, 0x7f7f7f7f);
    float c[256];
    hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    int bad_ab = 0, bad_ba = 0;
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        bad_ab += c[m * 16 + n] != ref[m * 16 + n];
        bad_ba += c[n * 16 + m] != ref[m * 16 + n];
      }
    printf("map %d (%s): mismatches C[m][n]=A.B^T: %d, transposed: %d   c[0][1]=%g ref=%g\n", h, names[h], bad_ab, bad_ba,
           c[1], ref[1]);
    if (bad_ab == 0 || bad_ba == 0) {
      // scale semantics: scale_a = 128 in byte 0 should double the result if E8M0 with bias 127
      hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, 0x7f7f7f80, 0x7f7f7f7f);
      hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
      printf("  scale_a byte0 = 0x80: c[0][1] = %g (x%g)\n", c[1], c[1] / ref[1]);
      hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, 0x7f7f7f7f, 0x7f7f7f7e);
      hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
      printf("  scale_b byte0 = 0x7e: c[0][1] = %g (x%g)\n", c[1], c[1] / ref[1]);
    }
  }
  float *dout;
  hipMalloc(&dout, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int fp8 = 0; fp8 < 2; ++fp8) {
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (fp8) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, dout, iters);
      else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, dout, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 16 * 16 * (fp8 ? 128 : 32) * 8.0 * iters * 256 * 4;
    printf("%s: %.3f ms, %.0f TFLOP/s (1 wave per SIMD, 8 independent accumulators)\n",
           fp8 ? "mfma_scale_f32_16x16x128_f8f6f4 (e4m3)" : "mfma_f32_16x16x32_bf16", ms, flop / ms / 1e9);
  }
  return 0;
}
