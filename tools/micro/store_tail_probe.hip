// Microbenchmark: what bounds the store burst of a GEMM epilogue?  One 512-thread workgroup per CU (128 KB of LDS
// reserved, as the ping-pong GEMM), every workgroup writes 256x256 bf16 tiles (128 KB, 16 store instructions of
// 16 bytes per lane and wave) to a [rows][ld] bf16 matrix, tile after tile, nothing else.  Lane -> address maps:
//   0  MFMA layout: lane = row (lane&15), 16-byte piece (lane>>4) of the wave's 64-byte row segment
//   1  store layout: lane = row (lane>>2), piece (lane&3): 4 adjacent lanes = 64 contiguous bytes
//   2  whole rows: lane = row (lane>>5), piece (lane&31): 32 adjacent lanes = one 512-byte tile row
//   3  as 2 with non-temporal stores
//   4  as 0 but 32 stores of 8 bytes per lane (same bytes, twice the instructions)
//   5  as 2, one wave per SIMD only (waves 4-7 idle, waves 0-3 store twice as much)
// Prints microseconds per tile and bytes per clock per CU for grids of 256 / 128 / 32 workgroups (a chip-wide
// limit scales with the grid, a per-CU limit does not).  Build: hipcc --offload-arch=gfx950 -O3 -o store_tail_probe store_tail_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(char *out, int ld_bytes, int tiles_n, int ntiles, int reps, unsigned long long *cyc) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  u32x4 v = {threadIdx.x * 2654435761u, blockIdx.x * 40503u, 0x3f803f80u, threadIdx.x ^ 0x5bd1e995u};
  if (smem[threadIdx.x] == 77) v[2] ^= 1;  // keeps the LDS allocation
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r)
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int tm_ = t / tiles_n, tn_ = t % tiles_n;
      char *tile = out + (size_t)tm_ * 256 * ld_bytes + (size_t)tn_ * 512;
      if (MODE == 0 || MODE == 4) {
        // wave (wm, wn): rows {lo, hi half} x wm*64 + 16*(mi&3) + lane&15, columns {lo, hi} x wn*64 bytes + 16*(lane>>4)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int mi = 0; mi < 8; ++mi) {
            const int row = (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15);
            char *p = tile + (size_t)row * ld_bytes + half * 256 + wn * 64 + 16 * (lane >> 4);
            if (MODE == 0) {
              *(u32x4 *)p = v;
            } else {
              *(u32x2 *)p = u32x2{v[0], v[1]};
              *(u32x2 *)(p + 8) = u32x2{v[2], v[3]};
            }
            v[0] += 1;
          }
      } else if (MODE == 1) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int mi = 0; mi < 8; ++mi) {
            const int row = (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane >> 2);
            *(u32x4 *)(tile + (size_t)row * ld_bytes + half * 256 + wn * 64 + 16 * (lane & 3)) = v;
            v[0] += 1;
          }
      } else if (MODE == 2 || MODE == 3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = wave * 32 + 2 * i + (lane >> 5);
          u32x4 *p = (u32x4 *)(tile + (size_t)row * ld_bytes + 16 * (lane & 31));
          if (MODE == 3) __builtin_nontemporal_store(v, p);
          else *p = v;
          v[0] += 1;
        }
      } else if (MODE == 5) {
        if (wave < 4) {
#pragma unroll
          for (int i = 0; i < 32; ++i) {
            const int row = wave * 64 + 2 * i + (lane >> 5);
            *(u32x4 *)(tile + (size_t)row * ld_bytes + 16 * (lane & 31)) = v;
            v[0] += 1;
          }
        }
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, char *out, int M, int N, unsigned long long *cyc) {
  const int ld = N * 2, tn = N / 256, ntiles = (M / 256) * tn;
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int grid : {256, 128, 32}) {
    const int reps = 4;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 131072, 0, out, ld, tn, ntiles, 1, cyc);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 131072, 0, out, ld, tn, ntiles, reps, cyc);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[256];
    hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double mc = 0;
    for (int i = 0; i < grid; ++i) mc += (double)h[i] / grid;
    const double tiles_per_wg = (double)ntiles * reps / grid;
    printf("mode %d %-28s grid %3d: %7.2f us per tile and workgroup, %6.0f cycles per tile (%5.1f B/clk/CU), chip %5.2f TB/s\n", MODE, name, grid,
           ms * 1e3 / tiles_per_wg, mc / tiles_per_wg, 131072.0 / (mc / tiles_per_wg), (double)ntiles * reps * 131072.0 / (ms * 1e-3) / 1e12);
    fflush(stdout);
  }
}

int main() {
  const int M = 21760, N = 9216;  // the qkv output of 5 work items: 401 MB, 3060 tiles
  char *out;
  unsigned long long *cyc;
  if (hipMalloc(&out, (size_t)M * N * 2) != hipSuccess || hipMalloc(&cyc, 256 * 8) != hipSuccess) return 1;
  hipMemset(out, 0, (size_t)M * N * 2);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("MFMA layout", out, M, N, cyc);
    run<1>("4 lanes = 64 B", out, M, N, cyc);
    run<2>("whole 512 B rows", out, M, N, cyc);
    run<3>("whole rows, non-temporal", out, M, N, cyc);
    run<4>("MFMA layout, 8 B stores", out, M, N, cyc);
    run<5>("whole rows, 4 waves", out, M, N, cyc);
  }
  return 0;
}
