// Probe: v_mfma_f32_32x32x16_bf16 with A and B in AGPRs, C in one VGPR tuple and D in ANOTHER VGPR tuple
// (the "accumulators start from -reference" form of ca_attn4): is D = A.B + C for every lane and register,
// for both operand-register choices the kernel makes?   hipcc --offload-arch=gfx950 -O2 mfma_cinit_probe.hip -o mfma_cinit_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16;
__global__ void probe(const bf16 *A, const bf16 *B, const float *cval, float *out) {
  const int lane = threadIdx.x;
  // A fragment: lane holds A[row = lane&31][k = 8*(lane>>5) + j]; B fragment: B[k = 8*(lane>>5)+j][col = lane&31]
  unsigned a[4], b[4];
  for (int j = 0; j < 4; ++j) {
    a[j] = *(const unsigned *)(A + (lane & 31) * 16 + 8 * (lane >> 5) + 2 * j);
    b[j] = *(const unsigned *)(B + (lane & 31) * 16 + 8 * (lane >> 5) + 2 * j);   // B stored [col][k]
  }
  asm volatile("" ::: "a192", "a193", "a194", "a195", "a128", "a129", "a130", "a131");
  asm volatile("v_accvgpr_write_b32 a192, %0\n\tv_accvgpr_write_b32 a193, %1\n\tv_accvgpr_write_b32 a194, %2\n\t"
               "v_accvgpr_write_b32 a195, %3\n\tv_accvgpr_write_b32 a128, %4\n\tv_accvgpr_write_b32 a129, %5\n\t"
               "v_accvgpr_write_b32 a130, %6\n\tv_accvgpr_write_b32 a131, %7\n\ts_nop 7"
               :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
  f32x16 C, D;
  const float c = cval[lane & 31];
  for (int r = 0; r < 16; ++r) C[r] = c;
  asm volatile("s_nop 7" : "+v"(C));
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[192:195], a[128:131], %1\n\ts_nop 15\n\ts_nop 7" : "=&v"(D) : "v"(C));
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = D[r];
  // and the accumulate form on top (C = D)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[192:195], a[128:131], %0\n\ts_nop 15\n\ts_nop 7" : "+v"(D));
  for (int r = 0; r < 16; ++r) out[1024 + lane * 16 + r] = D[r];
}
int main() {
  bf16 hA[32 * 16], hB[32 * 16];
  float hc[32], hout[2048];
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) {
    hA[i * 16 + k] = (bf16)(float)((i * 3 + k * 5) % 7 - 3);
    hB[i * 16 + k] = (bf16)(float)((i * 5 + k * 3) % 5 - 2);
  }
  for (int i = 0; i < 32; ++i) hc[i] = -(float)(10 + i);
  bf16 *dA, *dB; float *dc, *dout;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dc, sizeof hc); hipMalloc(&dout, sizeof hout);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dc, dout);
  hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) {
    const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float dot = 0; for (int k = 0; k < 16; ++k) dot += (float)hA[row * 16 + k] * (float)hB[col * 16 + k];
    if (std::fabs(hout[lane * 16 + r] - (dot + hc[col])) > 1e-3) ++bad1;
    if (std::fabs(hout[1024 + lane * 16 + r] - (2 * dot + hc[col])) > 1e-3) ++bad2;
  }
  printf("C-init form (D != C): %d wrong of 1024; accumulate on top: %d wrong; sample %g %g\n", bad1, bad2, hout[0], hout[1024]);
  return bad1 || bad2;
}
