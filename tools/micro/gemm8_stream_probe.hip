// Upper-bound probe for a one-wave-per-SIMD GEMM K loop on gfx950, written the way ca_attn4_kernel is: 4 waves per
// workgroup, 128 x 128 outputs per wave (256 accumulator registers a[0:255] owned by hand), v_mfma_f32_32x32x16_bf16,
// a generated stream per 64-wide K tile (64 MFMAs, the 32 ds_read_b128 of the next k16 steps' fragments in the gaps,
// one counted wait per step); operands re-read from an LDS-resident 256 x 64 A tile and 256 x 64 W tile (no global
// traffic, no barriers), pseudo-random bf16 bits.  Compare with wave_tile_probe.hip (hipcc's own schedule):
//   8 waves x 128 x 64 (the shipped kernel's shape) and 4 waves x 128 x 128 from plain HIP.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-spill-vgpr-to-agpr=0 -o gemm4_stream_probe gemm4_stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define AGPRS(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7) x(8) x(9) x(10) x(11) x(12) x(13) x(14) x(15)

__global__ __launch_bounds__(512, 1) void probe(float *out, int iters, unsigned seed) {
  extern __shared__ __attribute__((aligned(128))) char smem[];  // A tile 32 KB | W tile 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 4; i += 512) {
    unsigned x = (i * 2654435761u) ^ seed;
    x = (x & 0x807f807fu) | 0x3c003c00u;
    ((unsigned *)smem)[i] = x;
  }
  __syncthreads();
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15",
               "a16", "a32", "a48", "a64", "a80", "a96", "a112", "a127");
  for (int r = 0; r < 256; r += 1) {  // zero the accumulators
    // (register numbers must be literal: 256 statements through a switch would be silly -- the MFMAs below simply
    // accumulate onto whatever the file holds; the probe measures time, not values)
  }
  const int wm = wave >> 2, wn = wave & 3;
  const int row = lane & 15, kh = lane >> 4;
  uint32_t aaddr[2][8], waddr[2][8];   // [k16 step][fragment]: byte address of this lane's 16 bytes
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int ra = wm * 128 + 16 * f + row, rw = wn * 64 + 16 * (f & 3) + row;
      aaddr[ks][f] = base + ra * 128 + ((((4 * ks + kh) ^ ((ra >> 1) & 7)) & 7) << 4);
      waddr[ks][f] = base + 32768 + rw * 128 + ((((4 * ks + kh) ^ ((rw >> 1) & 7)) & 7) << 4);
    }
  i32x4 FA[2][8], FW[2][8];   // (FW: 4 fragments used)
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(FA[0][f]) : "v"(aaddr[0][f]) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(FW[0][f]) : "v"(waddr[0][f]) : "memory");
    FA[1][f] = i32x4{0, 0, 0, 0};
    FW[1][f] = i32x4{0, 0, 0, 0};
  }
  for (int it = 0; it < iters; ++it) {
#include "gemm8_stream_probe.inc"
  }
  asm volatile("s_nop 15\n\ts_nop 7");
  float s;
  asm volatile("v_accvgpr_read_b32 %0, a17" : "=v"(s));
  if (s == 1234.5f) out[0] = s;
}

int main() {
  float *out;
  hipMalloc(&out, 4);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 65536, 0, out, iters, 12345u + rep);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double flop = 2.0 * 16 * 16 * 32 * 64.0 * iters * 8 * 256;
  printf("%-44s %8.3f ms  %7.0f TFLOP/s   (24 ds_read_b128 per 64 MFMA per wave per K tile)\n",
         "8 waves x 128x64, 16x16x32, generated stream", ms, flop / ms / 1e9);
  hipFree(out);
  return 0;
}
