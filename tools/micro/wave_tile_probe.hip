// Upper-bound probe for the GEMM inner loop on gfx950: MFMA throughput when every operand fragment is re-read from
// LDS each K tile (no global traffic, no barriers), for two ways of covering a 256x256 block tile:
//   8 waves x (128 x 64) per wave  : 24 ds_read_b128 per 64 MFMAs  (the structure of ca_gemm_pp_kernel)
//   4 waves x (128 x 128) per wave : 32 ds_read_b128 per 128 MFMAs (one wave per SIMD, accumulators in AGPRs)
// Build: hipcc --offload-arch=gfx950 -O3 -o wave_tile_probe wave_tile_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int MI, int NJ>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(float *out, int iters, unsigned seed) {
  extern __shared__ __attribute__((aligned(128))) char smem[];  // 2 buffers x (A 32 KB + W 32 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // fill LDS with pseudo-random bf16 bits (finite, small exponents)
  for (int i = tid; i < 2 * 65536 / 4; i += WAVES * 64) {
    unsigned x = (i * 2654435761u) ^ seed;
    x = (x & 0x807f807fu) | 0x3c003c00u;
    ((unsigned *)smem)[i] = x;
  }
  __syncthreads();
  const int lane_off = (lane & 15) * 128 + ((((lane >> 4) ^ ((lane & 15) >> 1)) & 7) << 4);
  const int wm = WAVES == 8 ? wave >> 2 : wave >> 1;
  const int wn = WAVES == 8 ? wave & 3 : wave & 1;
  const int a_off = ((wm * (MI * 16)) % 256) * 128 + lane_off;
  const int w_off = 32768 + ((wn * (NJ * 16)) % 256) * 128 + lane_off;
  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const char *buf = smem + (it & 1) * 65536;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[MI], w[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *(const bf16x8 *)(buf + (((a_off + i * 16 * 128) & 32767) ^ (ks * 64)));
#pragma unroll
      for (int j = 0; j < NJ; ++j) w[j] = *(const bf16x8 *)(buf + 32768 + (((w_off - 32768 + j * 16 * 128) & 32767) ^ (ks * 64)));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[i], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 1234.5f) out[0] = s;
}

template <int WAVES, int MI, int NJ>
void run(const char *name) {
  float *out;
  hipMalloc(&out, 4);
  auto k = probe<WAVES, MI, NJ>;
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), 131072, 0, out, iters, 12345u + rep);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double flop = 2.0 * 16 * 16 * 32 * (double)MI * NJ * 2 * iters * WAVES * 256;
  printf("%-34s %8.3f ms  %7.0f TFLOP/s   (%d ds_read_b128 per %d MFMA per wave per K tile)\n", name, ms, flop / ms / 1e9,
         2 * (MI + NJ), 2 * MI * NJ);
  hipFree(out);
}

int main() {
  run<8, 8, 4>("8 waves x 128x64 (current)");
  run<4, 8, 8>("4 waves x 128x128");
  run<8, 4, 4>("8 waves x 64x64 (no reuse gain)");
  run<4, 8, 4>("4 waves x 128x64 (half the CU's work)");
  return 0;
}
