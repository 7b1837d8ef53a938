// Prototype of a one-wave-per-SIMD bf16 GEMM for gfx950 in the style of ca_attn4_kernel (NOT the product kernel: a
// measurement of what that structure reaches with real global traffic): C[M,N] = A[M,K] W[N,K]^T + bias, bf16 in / out.
// 4 waves per workgroup, 256 x 256 tile, 128 x 128 per wave in hand-owned accumulator registers a[0:255],
// v_mfma_f32_16x16x32_bf16 issued transposed, K tiles of 64 staged by LDS-DMA (buffer_load ... lds, source-side XOR
// swizzle) into two 64 KB stages, ONE barrier per K tile, the whole K tile a generated stream (gen_gemm4_kernel.py).
// M, N multiples of 256, K a multiple of 64.
// Build: python gen_gemm4_kernel.py && hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-spill-vgpr-to-agpr=0 \
//        -o gemm4_kernel_probe gemm4_kernel_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16;
typedef __attribute__((address_space(3))) char *lptr;

#define GEMM4_READ_ACC
#define ST 2
#include "gemm4_kernel_probe.inc"
#undef ST

__global__ __launch_bounds__(256, 1) void gemm4(const bf16 *__restrict__ A, const bf16 *__restrict__ W,
                                                const bf16 *__restrict__ bias, bf16 *__restrict__ C, int M, int N, int K,
                                                int lda, int ldw, int ldc) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // A stage 0 | A stage 1 | W stage 0 | W stage 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (workgroup b runs on XCD b & 7): the workgroups of one XCD walk the tiles in groups of 8 row
  // tiles x all column tiles, column-major inside a group, so that the 32 concurrent tiles of an XCD share A and W panels
  const int nt_n = N / 256, nt_m = M / 256, total = nt_m * nt_n;
  const int bid = blockIdx.x, xcd = bid & 7, q8 = total >> 3, r8 = total & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int grp = lid / (8 * nt_n), first_m = grp * 8, gm = min(8, nt_m - first_m), in_grp = lid - grp * 8 * nt_n;
  const int m0 = (first_m + in_grp % gm) * 256, n0 = (in_grp / gm) * 256;
  const int nk = K / 64;
  asm volatile("" ::: GEMM4_AGPR_CLOBBERS);
  GEMM4_ZERO();

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr)smem;
  auto uni32 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
  auto make_desc = [&](const bf16 *p, uint32_t records) {
    const uint64_t b = (uint64_t)(uintptr_t)p;
    return i32x4{(int)uni32((uint32_t)b), (int)(uni32((uint32_t)(b >> 32)) & 0xffffu), (int)records, 0x00020000};
  };
  const i32x4 dsa = make_desc(A + (size_t)m0 * lda, 0xffffffffu), dsw = make_desc(W + (size_t)n0 * ldw, 0xffffffffu);
  const i32x4 dsnull = make_desc(A, 0u);
  const uint32_t LW = uni32(lds0 + wave * 8192);
  // lane offsets of this wave's 8 + 8 DMA pieces (piece p = tile rows 64 wave + 8 p .. + 7; a lane moves 16 bytes)
  uint32_t aoff[8], woff[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = 64 * wave + 8 * p + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    aoff[p] = (uint32_t)r * (uint32_t)lda * 2u + (uint32_t)c * 16u;
    woff[p] = (uint32_t)r * (uint32_t)ldw * 2u + (uint32_t)c * 16u;
  }
  // byte addresses of this lane's fragment pieces (stage 0): [k32 step][fragment]
  uint32_t aaddr[2][8], waddr[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int ra = wm * 128 + 16 * f + (lane & 15), rw = wn * 128 + 16 * f + (lane & 15);
      aaddr[s][f] = lds0 + ra * 128 + ((((4 * s + (lane >> 4)) ^ ((ra >> 1) & 7)) & 7) << 4);
      waddr[s][f] = lds0 + 65536 + rw * 128 + ((((4 * s + (lane >> 4)) ^ ((rw >> 1) & 7)) & 7) << 4);
    }
  auto stage_tile = [&](int tile, int stage) {   // prologue staging (the loop's own pieces ride in its stream)
    const i32x4 da = tile < nk ? dsa : dsnull, dw = tile < nk ? dsw : dsnull;
    const uint32_t so = (uint32_t)tile * 128u;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                   : : "v"(aoff[p]), "s"(da), "s"(so), "s"(LW + stage * 32768 + 1024 * p) : "memory");
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                   : : "v"(woff[p]), "s"(dw), "s"(so), "s"(LW + 65536 + stage * 32768 + 1024 * p) : "memory");
    }
  };
  stage_tile(0, 0);
  stage_tile(1, 1);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  i32x4 FA[2][8], FW[2][8];
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(FA[0][f]) : "v"(aaddr[0][f]) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(FW[0][f]) : "v"(waddr[0][f]) : "memory");
    FA[1][f] = FW[1][f] = i32x4{0, 0, 0, 0};
  }
  for (int t = 0; t < nk; t += 2) {
    {
      const bool more = t + 2 < nk;
      const i32x4 DSA = more ? dsa : dsnull, DSW = more ? dsw : dsnull;
      const uint32_t SO = (uint32_t)(t + 2) * 128u;
#define ST 0
#include "gemm4_kernel_probe.inc"
#undef ST
    }
    if (t + 1 < nk) {
      const bool more = t + 3 < nk;
      const i32x4 DSA = more ? dsa : dsnull, DSW = more ? dsw : dsnull;
      const uint32_t SO = (uint32_t)(t + 3) * 128u;
#define ST 1
#include "gemm4_kernel_probe.inc"
#undef ST
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");
  // ---- epilogue: block (i, j) = C rows 16 i .., columns 16 j ..; a lane: row (lane & 15), 4 columns at 4 (lane >> 4)
  const int mrow = m0 + wm * 128 + (lane & 15), ncol = n0 + wn * 128 + 4 * (lane >> 4);
#define GEMM4_STORE(i, j, B)                                                                                      \
  {                                                                                                               \
    float v[4];                                                                                                   \
    GEMM4_READ_BLOCK_##B(v);                                                                                      \
    const int n = ncol + 16 * (j);                                                                                \
    bf16 o[4];                                                                                                    \
    for (int r = 0; r < 4; ++r) o[r] = (bf16)(v[r] + (float)bias[n + r]);                                         \
    *(uint2 *)(C + (size_t)(mrow + 16 * (i)) * ldc + n) = *(const uint2 *)o;                                      \
  }
#include "gemm4_kernel_probe_rows.inc"
}

int main(int argc, char **argv) {
  struct Shape { int M, N, K; const char *name; };
  const Shape shapes[] = {{1024, 1024, 512, "small (checked everywhere)"}, {8192, 8192, 8192, "8k cube"},
                          {21760, 12288, 3072, "mlp0 x5"}, {21760, 3072, 12288, "mlp2 x5"}, {21760, 9216, 3072, "qkv x5"}};
  hipFuncSetAttribute((const void *)gemm4, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (const Shape &s : shapes) {
    const size_t na = (size_t)s.M * s.K, nw = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
    std::vector<uint16_t> ha(na), hw(nw), hb(s.N);
    uint32_t x = 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (uint16_t)(((x >> 9) & 0x807f) | 0x3c00 | ((x >> 3) & 0x0380 & 0x0100)); };
    for (auto &v : ha) v = rnd();
    for (auto &v : hw) v = rnd();
    for (auto &v : hb) v = rnd();
    bf16 *A, *W, *B, *C;
    hipMalloc(&A, na * 2), hipMalloc(&W, nw * 2), hipMalloc(&B, s.N * 2), hipMalloc(&C, nc * 2);
    hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), s.N * 2, hipMemcpyHostToDevice);
    hipMemset(C, 0, nc * 2);
    const int grid = (s.M / 256) * (s.N / 256);
    hipLaunchKernelGGL(gemm4, dim3(grid), dim3(256), 131072, 0, A, W, B, C, s.M, s.N, s.K, s.K, s.K, s.N);
    hipDeviceSynchronize();
    std::vector<uint16_t> hc(nc);
    hipMemcpy(hc.data(), C, nc * 2, hipMemcpyDeviceToHost);
    auto f = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float r; memcpy(&r, &u, 4); return (double)r; };
    double worst = 0;
    uint32_t y = 777u;
    const int checks = s.M <= 1024 ? s.M * s.N : 4096;
    for (int c = 0; c < checks; ++c) {
      int m, n;
      if (s.M <= 1024) m = c / s.N, n = c % s.N;
      else { y = y * 1664525u + 1013904223u; m = (y >> 8) % s.M; y = y * 1664525u + 1013904223u; n = (y >> 8) % s.N; }
      double acc = f(hb[n]);
      for (int k = 0; k < s.K; ++k) acc += f(ha[(size_t)m * s.K + k]) * f(hw[(size_t)n * s.K + k]);
      const double e = fabs(f(hc[(size_t)m * s.N + n]) - acc) / (fabs(acc) + 1.0);
      if (e > worst) worst = e;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(gemm4, dim3(grid), dim3(256), 131072, 0, A, W, B, C, s.M, s.N, s.K, s.K, s.K, s.N);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-28s M=%5d N=%5d K=%5d: %9.1f us  %7.1f TFLOP/s   worst relative error %.2e %s\n", s.name, s.M, s.N, s.K,
           ms * 200.0, 2.0 * s.M * s.N * s.K / (ms / 5 * 1e-3) / 1e12, worst, worst < 2e-2 ? "ok" : "WRONG");
    fflush(stdout);
    hipFree(A), hipFree(W), hipFree(B), hipFree(C);
  }
  return 0;
}
