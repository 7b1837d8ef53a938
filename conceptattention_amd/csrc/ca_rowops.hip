// HBM-bound row kernels of the ConceptAttention path for gfx950: LayerNorm+modulation,
// QK-RMSNorm+RoPE, small-batch GEMV (weight streaming), concept heat-map reduction, Euler axpy.
// All use 16-byte-per-lane coalesced accesses and fp32 arithmetic; none of them is shaped into
// an MFMA product (they are bandwidth-bound: see DESIGN.md for bytes per unit).
#include <atomic>
#include <type_traits>

#include "ca_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------------
// out = (1 + scale) * LayerNorm(x) + shift ; one wave per row, row kept in registers.
struct LnArgs {
  ca_mod_segment seg[CA_MAX_SEGMENTS];
  int32_t n_segs;
};
constexpr int LN_MAXCH = 8;  // H <= 8 * 512

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// 8 floats (already divided by the row scale) -> 8 OCP e4m3 bytes (v_cvt_pk_fp8_f32, RNE; inputs clamped to
// the largest finite e4m3 value so nothing overflows to NaN)
constexpr float E4M3_MAX = 448.0f;
__device__ __forceinline__ uint2 ca_pack_fp8x8(const float *y) {
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = fminf(fmaxf(y[j], -E4M3_MAX), E4M3_MAX);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return make_uint2((uint32_t)lo, (uint32_t)hi);
}

// FP8: the modulated row is quantised to e4m3 with one absmax scale per row (out8 row stride ldo BYTES,
// out_scale[row] = absmax / 448): the A operand of ca_gemm_fp8.
// LO (bf16 output only): a second plane out_lo = bf16(y - float(bf16(y))), the part of the modulated row its bf16
// rounding drops; out + out_lo carries ~16 mantissa bits (the q projection of the layers whose cross-attention-space
// vectors are captured is corrected with a product of this plane).
// y = (1 + scale) * ((v - mean) * rstd) + shift, with its one fused multiply-add spelled out (both LayerNorm kernels
// must round alike: a 5-item forward and a single-item one may run different ones)
__device__ __forceinline__ float ln_apply(float v, float mean, float rstd, float scale1, float shift) {
  return __builtin_fmaf(scale1, (v - mean) * rstd, shift);
}

template <bool FP8, typename XT = bf16, bool LO = false>
__global__ __launch_bounds__(256) void ca_ln_modulate_kernel(const XT *__restrict__ x, int ldx,
                                                             void *__restrict__ out_, int ldo, int M, int H,
                                                             float eps, float *__restrict__ out_scale,
                                                             const LnArgs A, bf16 *__restrict__ out_lo = nullptr,
                                                             int ldlo = 0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  int si = 0;
#pragma unroll
  for (int s = 0; s < CA_MAX_SEGMENTS - 1; ++s)
    if (s + 1 < A.n_segs && row >= A.seg[s].row_end) si = s + 1;
  const float *shift = A.seg[si].shift;
  const float *scale = A.seg[si].scale;

  const XT *xr = x + (size_t)row * ldx;
  float v[LN_MAXCH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXCH; ++c) {
    const int k = c * 512 + lane * 8;
    if (k < H) {
      if constexpr (std::is_same<XT, float>::value) {
        const f32x4 t0 = *(const f32x4 *)(xr + k), t1 = *(const f32x4 *)(xr + k + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[c][j] = t0[j];
          v[c][4 + j] = t1[j];
        }
      } else {
        const bf16x8 t = *(const bf16x8 *)(xr + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c][j] = (float)t[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[c][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXCH; ++c) {
    const int k = c * 512 + lane * 8;
    if (k < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[c][j] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXCH; ++c) {
    const int k = c * 512 + lane * 8;
    if (k < H) {
      const f32x4 sc0 = *(const f32x4 *)(scale + k), sc1 = *(const f32x4 *)(scale + k + 4);
      const f32x4 sh0 = *(const f32x4 *)(shift + k), sh1 = *(const f32x4 *)(shift + k + 4);
      float y[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = ln_apply(v[c][j], mean, rstd, 1.f + sc0[j], sh0[j]);
        y[4 + j] = ln_apply(v[c][4 + j], mean, rstd, 1.f + sc1[j], sh1[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[c][j] = y[j];
        if constexpr (FP8) amax = fmaxf(amax, fabsf(y[j]));
      }
    }
  }
  if constexpr (!FP8) {
    // stores after ALL modulation-vector loads: issued chunk by chunk between the stores, every load waited (vmcnt
    // retires in order and counts stores) for the acknowledgement of the chunk stored before it
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c) {
      const int k = c * 512 + lane * 8;
      if (k < H) {
        *(uint4 *)((bf16 *)out_ + (size_t)row * ldo + k) =
            make_uint4(ca_pack2(v[c][0], v[c][1]), ca_pack2(v[c][2], v[c][3]), ca_pack2(v[c][4], v[c][5]),
                       ca_pack2(v[c][6], v[c][7]));
        if constexpr (LO) {
          float r[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] = v[c][j] - (float)(bf16)v[c][j];
          *(uint4 *)(out_lo + (size_t)row * ldlo + k) =
              make_uint4(ca_pack2(r[0], r[1]), ca_pack2(r[2], r[3]), ca_pack2(r[4], r[5]), ca_pack2(r[6], r[7]));
        }
      }
    }
  }
  if constexpr (FP8) {
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / E4M3_MAX) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) out_scale[row] = sc;
    uint8_t *orow = (uint8_t *)out_ + (size_t)row * ldo;
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c) {
      const int k = c * 512 + lane * 8;
      if (k < H) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = v[c][j] * inv;
        *(uint2 *)(orow + k) = ca_pack_fp8x8(y);
      }
    }
  }
}

// The same LayerNorm + modulation for the model's own shape (fp32 residual stream in, bf16 out, H = NCH x 512), with a
// wave walking ROWS_PER_WAVE consecutive rows: the shift / scale vectors of the rows' segment (24 KB per row at
// H = 3072, twice the row itself) stay in registers across the rows instead of being pulled through L2 for every
// row.  Per row the arithmetic is ca_ln_modulate_kernel's,
// operation for operation (same summation order, same fused multiply-adds): the two kernels agree bit for bit.
constexpr int LN_ROWS_PER_WAVE = 8;
template <int NCH, bool LO>
__global__ __launch_bounds__(256, 2) void ca_ln_modulate_rows_kernel(const float *__restrict__ x, int ldx,
                                                                  bf16 *__restrict__ out, int ldo, int M, float eps,
                                                                  const LnArgs A, bf16 *__restrict__ out_lo, int ldlo) {
  constexpr int H = NCH * 512;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int row0 = wave * LN_ROWS_PER_WAVE;
  if (row0 >= M) return;
  const int nrows = min(LN_ROWS_PER_WAVE, M - row0);
  float sc1[NCH][8], sh[NCH][8];   // 1 + scale, shift of the current segment
  int cur_seg = -1;
  float v[1][NCH][8];
  auto load_row = [&](int buf, int row) {
    const float *xr = x + (size_t)row * ldx + lane * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const f32x4 t0 = *(const f32x4 *)(xr + c * 512), t1 = *(const f32x4 *)(xr + c * 512 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[buf][c][j] = t0[j], v[buf][c][4 + j] = t1[j];
    }
  };
  auto do_row = [&](int buf, int row) {
    int si = 0;
#pragma unroll
    for (int s = 0; s < CA_MAX_SEGMENTS - 1; ++s)
      if (s + 1 < A.n_segs && row >= A.seg[s].row_end) si = s + 1;
    if (si != cur_seg) {   // (wave-uniform; once per wave except where its rows cross a segment boundary)
      cur_seg = si;
      const float *scale = A.seg[si].scale + lane * 8, *shift = A.seg[si].shift + lane * 8;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const f32x4 a0 = *(const f32x4 *)(scale + c * 512), a1 = *(const f32x4 *)(scale + c * 512 + 4);
        const f32x4 b0 = *(const f32x4 *)(shift + c * 512), b1 = *(const f32x4 *)(shift + c * 512 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sc1[c][j] = 1.f + a0[j], sc1[c][4 + j] = 1.f + a1[j];
          sh[c][j] = b0[j], sh[c][4 + j] = b1[j];
        }
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[buf][c][j];
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[buf][c][j] - mean;
        sq += d * d;
      }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = ln_apply(v[buf][c][j], mean, rstd, sc1[c][j], sh[c][j]);
      const int k = c * 512 + lane * 8;
      *(uint4 *)(out + (size_t)row * ldo + k) =
          make_uint4(ca_pack2(y[0], y[1]), ca_pack2(y[2], y[3]), ca_pack2(y[4], y[5]), ca_pack2(y[6], y[7]));
      if constexpr (LO) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = y[j] - (float)(bf16)y[j];
        *(uint4 *)(out_lo + (size_t)row * ldlo + k) =
            make_uint4(ca_pack2(r[0], r[1]), ca_pack2(r[2], r[3]), ca_pack2(r[4], r[5]), ca_pack2(r[6], r[7]));
      }
    }
  };
  for (int i = 0; i < nrows; ++i) {   // (two waves per SIMD; requesting the next row ahead of the reduction -- a second row
    load_row(0, row0 + i);            // buffer, 254 registers -- measured no faster: 82.7 us either way, 4.85 TB/s)
    do_row(0, row0 + i);
  }
}

// ------------------------------------------------------------------------------------------
// Row-wise absmax quantisation bf16 -> e4m3 (weights once per model; activations that leave a GEMM or the
// attention kernel in bf16).  One wave per row, two passes over the row (the second one hits L2).
__global__ __launch_bounds__(256) void ca_quantize_rows_fp8_kernel(const bf16 *__restrict__ x, int ldx,
                                                                   uint8_t *__restrict__ out, int ldo,
                                                                   float *__restrict__ out_scale, int M, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const bf16 *xr = x + (size_t)row * ldx;
  float amax = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const bf16x8 t = *(const bf16x8 *)(xr + k);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)t[j]));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.0f / E4M3_MAX) : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) out_scale[row] = sc;
  uint8_t *orow = out + (size_t)row * ldo;
  for (int k = lane * 8; k < K; k += 512) {
    const bf16x8 t = *(const bf16x8 *)(xr + k);
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = (float)t[j] * inv;
    *(uint2 *)(orow + k) = ca_pack_fp8x8(y);
  }
}

// ------------------------------------------------------------------------------------------
// QK-RMSNorm + RoPE in place; 16 lanes per (row, q|k, head) unit, 8 elements (4 rotation pairs)/lane
struct NormArgs {
  ca_norm_segment seg[CA_MAX_SEGMENTS];
  int32_t n_segs;
};

__global__ __launch_bounds__(256) void ca_qknorm_rope_kernel(bf16 *__restrict__ qkv, int ld, int M, int NH,
                                                             const float *__restrict__ rope,
                                                             bf16 *__restrict__ q_prerope, int ldp,
                                                             const NormArgs A) {
  const int t16 = threadIdx.x & 15;
  const long unit = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const long nunits = (long)M * 2 * NH;
  const bool valid = unit < nunits;
  const long u = valid ? unit : nunits - 1;  // keep all lanes alive for the shuffles
  const int row = (int)(u / (2 * NH));
  const int rem = (int)(u % (2 * NH));
  const int which = rem / NH, head = rem % NH;
  int si = 0;
#pragma unroll
  for (int s = 0; s < CA_MAX_SEGMENTS - 1; ++s)
    if (s + 1 < A.n_segs && row >= A.seg[s].row_end) si = s + 1;
  const bf16 *scale = (const bf16 *)(which ? A.seg[si].k_scale : A.seg[si].q_scale);

  bf16 *p = qkv + (size_t)row * ld + (size_t)which * NH * 128 + head * 128 + t16 * 8;
  const bf16x8 xv = *(const bf16x8 *)p;
  const bf16x8 sv = *(const bf16x8 *)(scale + t16 * 8);
  float x[8];
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    x[j] = (float)xv[j];
    sq += x[j] * x[j];
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
  const float rrms = rsqrtf(sq * (1.0f / 128.0f) + 1e-6f);
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = x[j] * rrms * (float)sv[j];
  if (!valid) return;
  if (q_prerope && which == 0) {
    *(uint4 *)(q_prerope + (size_t)row * ldp + head * 128 + t16 * 8) =
        make_uint4(ca_pack2(x[0], x[1]), ca_pack2(x[2], x[3]), ca_pack2(x[4], x[5]), ca_pack2(x[6], x[7]));
  }
  // rope table row: [64 pairs][cos, sin]; this lane owns pairs 4*t16 .. 4*t16+3
  const float *rp = rope + (size_t)row * 128 + t16 * 8;
  const f32x4 r0 = *(const f32x4 *)rp, r1 = *(const f32x4 *)(rp + 4);
  const float cs[4] = {r0[0], r0[2], r1[0], r1[2]};
  const float sn[4] = {r0[1], r0[3], r1[1], r1[3]};
  float y[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    y[2 * i] = cs[i] * x[2 * i] - sn[i] * x[2 * i + 1];
    y[2 * i + 1] = sn[i] * x[2 * i] + cs[i] * x[2 * i + 1];
  }
  *(uint4 *)p = make_uint4(ca_pack2(y[0], y[1]), ca_pack2(y[2], y[3]), ca_pack2(y[4], y[5]), ca_pack2(y[6], y[7]));
}

// ------------------------------------------------------------------------------------------
// out[v,n] (+)= sum_k f(x[v,k]) W[n,k] + bias[n]; x (after f) staged in LDS as fp32.  A wave streams
// GEMV_ROWS weight rows at a time so each 8-element slice of the NV input vectors is read from LDS
// once per GEMV_ROWS rows (with one row per pass and NV = 8 the kernel was LDS-bound at 1.1 TB/s).
constexpr int GEMV_ROWS = 4;
template <int NV>
__global__ __launch_bounds__(256) void ca_gemv_kernel(const float *__restrict__ x, int ldx,
                                                      const bf16 *__restrict__ W, const bf16 *__restrict__ bias,
                                                      float *__restrict__ out, int ldo, int N, int K, int silu,
                                                      int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *xs = (float *)smem_raw;  // [NV][K]
  for (int i = threadIdx.x; i < NV * K; i += 256) {
    const int v = i / K, k = i - v * K;
    const float t = x[(size_t)v * ldx + k];
    xs[i] = silu ? ca_silu(t) : t;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int n0 = (blockIdx.x * 4 + wave) * GEMV_ROWS; n0 < N; n0 += gridDim.x * 4 * GEMV_ROWS) {
    float acc[GEMV_ROWS][NV];
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r)
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[r][v] = 0.f;
    const bf16 *wr[GEMV_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r) wr[r] = W + (size_t)min(n0 + r, N - 1) * K;
    for (int k = lane * 8; k < K; k += 512) {
      bf16x8 w8[GEMV_ROWS];
#pragma unroll
      for (int r = 0; r < GEMV_ROWS; ++r) w8[r] = *(const bf16x8 *)(wr[r] + k);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x4 a = *(const f32x4 *)(xs + v * K + k), b = *(const f32x4 *)(xs + v * K + k + 4);
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[r][v] = fmaf((float)w8[r][j], a[j], acc[r][v]);
            acc[r][v] = fmaf((float)w8[r][4 + j], b[j], acc[r][v]);
          }
      }
    }
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r)
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[r][v] = wave_sum(acc[r][v]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < GEMV_ROWS; ++r) {
        const int n = n0 + r;
        if (n < N) {
          const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            float *o = out + (size_t)v * ldo + n;
            const float res = acc[r][v] + bv;
            *o = accumulate ? (*o + res) : res;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Per (row, head): y = (x + d) * rsqrt(mean((x + d)^2) + 1e-6) * scale, in place on x (fp32).  x = the q third of a qkv
// projection before its RMS norm (what CA_EPI_QKV_NORM_ROPE stores to q_prerope with qpre_f32 = 2), d = the product
// of the LayerNorm output's low plane with the same weights: the sum is the projection of the UNROUNDED LayerNorm
// output to ~16 mantissa bits.  16 lanes per (row, head), 8 columns each.
// (round 4) With `rope` and `q_out` the normalised vector is also rotated (apply_rope, flux/math.py:25-30), multiplied by
// q_out_scale and stored as the ATTENTION's q (bf16, or IEEE half with q_f16) -- exactly what the qkv epilogue does with
// the projection of bf16(y), but from the projection of the unrounded y: the bf16 rounding of that GEMM operand, as it
// reaches q, is what bounds a single output-space heat map (tests/tools/diag_out_space.py: 7.7e-4 of 7.8e-4).
__global__ __launch_bounds__(256) void ca_qpre_finish_kernel(float *__restrict__ x, int ldx, const float *__restrict__ d,
                                                             int ldd, const bf16 *__restrict__ scale, int M, int heads,
                                                             const float *__restrict__ rope, void *__restrict__ q_out,
                                                             int ldq, float q_out_scale, int q_f16) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long unit = gid >> 4;           // (row, head)
  const int t16 = (int)(gid & 15);
  if (unit >= (long)M * heads) return;  // (whole 16-lane groups leave together: M * heads units of 16 lanes)
  const int row = (int)(unit / heads), head = (int)(unit % heads);
  float *xp = x + (size_t)row * ldx + head * 128 + t16 * 8;
  f32x4 a0 = *(const f32x4 *)xp, a1 = *(const f32x4 *)(xp + 4);
  if (d) {
    const float *dp = d + (size_t)row * ldd + head * 128 + t16 * 8;
    const f32x4 b0 = *(const f32x4 *)dp, b1 = *(const f32x4 *)(dp + 4);
    a0 += b0;
    a1 += b1;
  }
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) sq = fmaf(a0[j], a0[j], fmaf(a1[j], a1[j], sq));
  sq += __shfl_xor(sq, 1);
  sq += __shfl_xor(sq, 2);
  sq += __shfl_xor(sq, 4);
  sq += __shfl_xor(sq, 8);
  const float rrms = rsqrtf(fmaf(sq, 1.0f / 128.0f, 1e-6f));
  const bf16x8 s8 = *(const bf16x8 *)(scale + t16 * 8);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a0[j] = a0[j] * rrms * (float)s8[j];
    a1[j] = a1[j] * rrms * (float)s8[4 + j];
  }
  *(f32x4 *)xp = a0;
  *(f32x4 *)(xp + 4) = a1;
  if (q_out) {
    const float *rp = rope + (size_t)row * 128 + t16 * 8;   // [64 pairs][cos, sin]: pairs 4 t16 .. 4 t16 + 3
    const f32x4 r0 = *(const f32x4 *)rp, r1 = *(const f32x4 *)(rp + 4);
    const float y[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const float cs[4] = {r0[0], r0[2], r1[0], r1[2]}, sn[4] = {r0[1], r0[3], r1[1], r1[3]};
    const float qos = q_out_scale == 0.0f ? 1.0f : q_out_scale;
    float z[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // (the expressions of the qkv epilogue, ca_gemm.hip)
      z[2 * i] = __builtin_fmaf(cs[i], y[2 * i], -(sn[i] * y[2 * i + 1])) * qos;
      z[2 * i + 1] = __builtin_fmaf(sn[i], y[2 * i], cs[i] * y[2 * i + 1]) * qos;
    }
    uint4 o;
    if (q_f16) o = make_uint4(ca_pack2_f16(z[0], z[1]), ca_pack2_f16(z[2], z[3]), ca_pack2_f16(z[4], z[5]), ca_pack2_f16(z[6], z[7]));
    else o = make_uint4(ca_pack2(z[0], z[1]), ca_pack2(z[2], z[3]), ca_pack2(z[4], z[5]), ca_pack2(z[6], z[7]));
    *(uint4 *)((char *)q_out + ((size_t)row * ldq + head * 128 + t16 * 8) * 2) = o;
  }
}

// logits[c,p] = <img_vec[p,:], con_vec[c,:]> for CC concepts per pass; one wave per patch.
// The concept vectors sit in LDS as fp32 (they are either bf16 or already fp32 in HBM).
template <int CC, typename CT, typename IT = bf16>
__global__ __launch_bounds__(256) void ca_heatmap_logits_kernel(const IT *__restrict__ img, int ldi,
                                                                const CT *__restrict__ con, int ldc, int L,
                                                                int C, int c0, int dim,
                                                                float *__restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *cs = (float *)smem_raw;  // [CC][dim]
  for (int c = 0; c < CC; ++c) {  // (dim % 8 == 0, rows 16-byte aligned: four elements per load)
    const CT *cr = con + (size_t)min(c0 + c, C - 1) * ldc;
    for (int k = threadIdx.x * 4; k < dim; k += 1024) {
      f32x4 v;
      if constexpr (std::is_same<CT, float>::value) {
        v = *(const f32x4 *)(cr + k);
      } else {
        const bf16x4 t = *(const bf16x4 *)(cr + k);
        v = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
      }
      *(f32x4 *)(cs + c * dim + k) = v;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // two patches per wave at a time (their row loads in flight together; one read of the concept vectors from LDS
  // serves both), each patch with its own accumulation chain in k order
  for (int p = (blockIdx.x * 4 + wave) * 2; p < L; p += gridDim.x * 8) {
    const IT *ir0 = img + (size_t)p * ldi;
    const IT *ir1 = img + (size_t)min(p + 1, L - 1) * ldi;
    float acc0[CC], acc1[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) acc0[c] = acc1[c] = 0.f;
    for (int k = lane * 8; k < dim; k += 512) {
      float a0[8], a1[8];   // the same k order per patch for bf16 and fp32 image vectors
      if constexpr (std::is_same<IT, float>::value) {
        const f32x4 u0 = *(const f32x4 *)(ir0 + k), u1 = *(const f32x4 *)(ir0 + k + 4);
        const f32x4 w0 = *(const f32x4 *)(ir1 + k), w1 = *(const f32x4 *)(ir1 + k + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a0[j] = u0[j]; a0[4 + j] = u1[j]; a1[j] = w0[j]; a1[4 + j] = w1[j]; }
      } else {
        const bf16x8 u = *(const bf16x8 *)(ir0 + k), w = *(const bf16x8 *)(ir1 + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) { a0[j] = (float)u[j]; a1[j] = (float)w[j]; }
      }
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        const f32x4 b0 = *(const f32x4 *)(cs + c * dim + k), b1 = *(const f32x4 *)(cs + c * dim + k + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0[c] = fmaf(a0[j], b0[j], acc0[c]);
          acc0[c] = fmaf(a0[4 + j], b1[j], acc0[c]);
          acc1[c] = fmaf(a1[j], b0[j], acc1[c]);
          acc1[c] = fmaf(a1[4 + j], b1[j], acc1[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      acc0[c] = wave_sum(acc0[c]);
      acc1[c] = wave_sum(acc1[c]);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < CC; ++c)
        if (c0 + c < C) {
          logits[(size_t)(c0 + c) * L + p] = acc0[c];
          if (p + 1 < L) logits[(size_t)(c0 + c) * L + p + 1] = acc1[c];
        }
    }
  }
}

// The weighting across the C concepts of one patch, shared by the three-launch kernels below and the fused kernel, so
// that both forms evaluate the same expressions.
// softmax: v[c] = exp(z[c] - max z) for c < C, sum = their sum in c order (the caller multiplies by weight / sum).
template <int CMAX>
__device__ __forceinline__ void hm_softmax_terms(const float (&z)[CMAX], int C, float (&v)[CMAX], float &sum) {
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) mx = fmaxf(mx, z[c]);
  sum = 0.f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    v[c] = c < C ? __expf(z[c] - mx) : 0.f;
    if (c < C) sum += v[c];
  }
}

// sparsemax_c / entmax15_c (logits[:,p]): the two sparse alternatives to the softmax over concepts
// (concept_attention_pipeline.py:66-69 calls the third-party `entmax` package for them).  Both are the
// published sort-and-threshold algorithms: sparsemax (Martins & Astudillo 2016, Alg. 1): z sorted descending,
// k = max{j : 1 + j z_(j) > sum_{i<=j} z_(i)}, tau = (sum_{i<=k} z_(i) - 1) / k, p = max(z - tau, 0);
// 1.5-entmax (Peters, Niculae & Martins 2019, Alg. 2): x = (z - max z) / 2 sorted, for every prefix j
// mean M_j, ss_j = j (mean of squares - M_j^2), tau_j = M_j - sqrt(max((1 - ss_j) / j, 0)),
// k = #{j : tau_j <= x_(j)}, p = max(x - tau_k, 0)^2.  The C <= CMAX logits in registers; z_in[c >= C] is ignored.
template <int CMAX, bool ENTMAX15>
__device__ __forceinline__ void hm_sparse_terms(const float (&z_in)[CMAX], int C, float (&v)[CMAX]) {
  float z[CMAX], srt[CMAX];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    z[c] = c < C ? z_in[c] : -INFINITY;
    mx = fmaxf(mx, z[c]);
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    z[c] = (z[c] - mx) * (ENTMAX15 ? 0.5f : 1.0f);  // both maps are shift-invariant; -inf padding stays -inf
    srt[c] = z[c];
  }
  // odd-even transposition sort, descending (fully unrolled: the array never leaves the registers)
#pragma unroll
  for (int pass = 0; pass < CMAX; ++pass)
#pragma unroll
    for (int i = pass & 1; i + 1 < CMAX; i += 2) {
      const float a = srt[i], b = srt[i + 1];
      srt[i] = fmaxf(a, b);
      srt[i + 1] = fminf(a, b);
    }
  float tau = 0.f;
  float cs = 0.f, cs2 = 0.f;
#pragma unroll
  for (int j = 0; j < CMAX; ++j) {
    if (j < C) {
      cs += srt[j];
      cs2 += srt[j] * srt[j];
      const float rho = (float)(j + 1);
      float tj;
      bool in;
      if (ENTMAX15) {
        const float mean = cs / rho;
        const float ss = rho * (cs2 / rho - mean * mean);
        tj = mean - sqrtf(fmaxf((1.0f - ss) / rho, 0.0f));
        in = tj <= srt[j];
      } else {
        tj = (cs - 1.0f) / rho;
        in = 1.0f + rho * srt[j] > cs;
      }
      if (in) tau = tj;  // the support is a prefix of the sorted order, so the last j that passes is k
    }
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    const float d = fmaxf(z[c] - tau, 0.f);
    v[c] = c < C ? (ENTMAX15 ? d * d : d) : 0.f;
  }
}

// acc[c,p] += weight * softmax_c(logits[:,p])
__global__ __launch_bounds__(256) void ca_heatmap_softmax_kernel(const float *__restrict__ logits, int C, int L,
                                                                 float weight, float *__restrict__ acc) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= L) return;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[(size_t)c * L + p]);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += __expf(logits[(size_t)c * L + p] - mx);
  const float inv = weight / sum;
  for (int c = 0; c < C; ++c) acc[(size_t)c * L + p] += __expf(logits[(size_t)c * L + p] - mx) * inv;
}

// acc[c,p] += weight * sparsemax_c / entmax15_c (logits[:,p]); one thread per patch.
template <int CMAX, bool ENTMAX15>
__global__ __launch_bounds__(256) void ca_heatmap_sparse_kernel(const float *__restrict__ logits, int C, int L,
                                                                float weight, float *__restrict__ acc) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= L) return;
  float z[CMAX], v[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) z[c] = c < C ? logits[(size_t)c * L + p] : -INFINITY;
  hm_sparse_terms<CMAX, ENTMAX15>(z, C, v);
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) acc[(size_t)c * L + p] += weight * v[c];
}

// ------------------------------------------------------------------------------------------
// Fused heat maps of one layer (ca_heatmap_fused): blockIdx.y = problem (work item x space), a wave takes two patches
// at a time: all C logits of both in one pass over their image vectors (every 16-byte piece of the two rows requested
// before the first is used: 24 KB in flight per wave for fp32 vectors), the concept vectors fp32 in LDS (one LDS read
// serves both patches), then lanes 0 / 1 weight their patch's C logits and update the accumulators.  Per (patch,
// concept) the accumulation chain is ca_heatmap_logits_kernel's (lane l: k = 8 l + 512 i in i order, then the wave
// sum), and the weighting is the three-launch kernels' expressions: bit-identical results.
struct HeatmapLaunch {
  ca_heatmap_problem p[CA_HEATMAP_MAX_PROBLEMS];
  int32_t L, C, dim, norm;
};
constexpr int HM_PRE = 6;  // 512-element chunks of a row requested ahead (dim <= 3072 in one go)

// the weighting of one patch's C logits and the update of the problem's accumulators (shared by both bodies below)
template <int CC>
__device__ __forceinline__ void hm_weight_and_accumulate(const ca_heatmap_problem &P, const float (&z)[CC], int pp, int L,
                                                         int C, int norm) {
  float v[CC];
  if (P.logits) {
#pragma unroll
    for (int c = 0; c < CC; ++c)
      if (c < C) P.logits[(size_t)c * L + pp] = z[c];
  }
  if (norm == CA_NORM_SOFTMAX) {
    float sum;
    hm_softmax_terms<CC>(z, C, v, sum);
    if (P.acc) {
      const float inv = P.weight / sum;
#pragma unroll
      for (int c = 0; c < CC; ++c)
        if (c < C) P.acc[(size_t)c * L + pp] += v[c] * inv;
    }
    if (P.acc2) {
      const float inv = P.weight2 / sum;
#pragma unroll
      for (int c = 0; c < CC; ++c)
        if (c < C) P.acc2[(size_t)c * L + pp] += v[c] * inv;
    }
  } else {
    if (norm == CA_NORM_SPARSEMAX) hm_sparse_terms<CC, false>(z, C, v);
    else hm_sparse_terms<CC, true>(z, C, v);
    if (P.acc) {
#pragma unroll
      for (int c = 0; c < CC; ++c)
        if (c < C) P.acc[(size_t)c * L + pp] += P.weight * v[c];
    }
    if (P.acc2) {
#pragma unroll
      for (int c = 0; c < CC; ++c)
        if (c < C) P.acc2[(size_t)c * L + pp] += P.weight2 * v[c];
    }
  }
}

template <int CC, typename IT>
__device__ __forceinline__ void heatmap_fused_body(const ca_heatmap_problem &P, int L, int C, int dim, int norm,
                                                   const float *cs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const IT *img = (const IT *)P.img_vec;
  constexpr bool F32 = std::is_same<IT, float>::value;
  for (int p = (blockIdx.x * nw + wave) * 2; p < L; p += gridDim.x * nw * 2) {
    const IT *ir0 = img + (size_t)p * P.ldi;
    const IT *ir1 = img + (size_t)min(p + 1, L - 1) * P.ldi;
    float acc0[CC], acc1[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) acc0[c] = acc1[c] = 0.f;
    for (int kb = lane * 8; kb < dim; kb += 512 * HM_PRE) {
      f32x4 u0[HM_PRE][F32 ? 2 : 1], u1[HM_PRE][F32 ? 2 : 1];   // fp32: 8 floats; bf16: 8 x bf16 in one 16-byte piece
#pragma unroll
      for (int i = 0; i < HM_PRE; ++i) {
        const int k = kb + 512 * i;
        if (k < dim) {
          if constexpr (F32) {
            u0[i][0] = *(const f32x4 *)(ir0 + k), u0[i][1] = *(const f32x4 *)(ir0 + k + 4);
            u1[i][0] = *(const f32x4 *)(ir1 + k), u1[i][1] = *(const f32x4 *)(ir1 + k + 4);
          } else {
            u0[i][0] = *(const f32x4 *)(ir0 + k);
            u1[i][0] = *(const f32x4 *)(ir1 + k);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < HM_PRE; ++i) {
        const int k = kb + 512 * i;
        if (k < dim) {
          float a0[8], a1[8];
          if constexpr (F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0[j] = u0[i][0][j]; a0[4 + j] = u0[i][1][j]; a1[j] = u1[i][0][j]; a1[4 + j] = u1[i][1][j]; }
          } else {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, u0[i][0]), b1 = __builtin_bit_cast(bf16x8, u1[i][0]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { a0[j] = (float)b0[j]; a1[j] = (float)b1[j]; }
          }
#pragma unroll
          for (int c = 0; c < CC; ++c) {
            const f32x4 b0 = *(const f32x4 *)(cs + c * dim + k), b1 = *(const f32x4 *)(cs + c * dim + k + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // (ca_heatmap_logits_kernel's order)
              acc0[c] = fmaf(a0[j], b0[j], acc0[c]);
              acc0[c] = fmaf(a0[4 + j], b1[j], acc0[c]);
              acc1[c] = fmaf(a1[j], b0[j], acc1[c]);
              acc1[c] = fmaf(a1[4 + j], b1[j], acc1[c]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      acc0[c] = wave_sum(acc0[c]);
      acc1[c] = wave_sum(acc1[c]);
    }
    const int pp = p + lane;
    if (lane < 2 && pp < L) {
      float z[CC];
#pragma unroll
      for (int c = 0; c < CC; ++c) z[c] = lane ? acc1[c] : acc0[c];
      hm_weight_and_accumulate<CC>(P, z, pp, L, C, norm);
    }
  }
}

// img_f32 = 2: the logits of a patch are the sum over the heads (in head order) of the partial logits an attention launch
// left in ca_attn_problem.hm_part ([heads][L][8] fp32); one thread per patch, 32 bytes per head and thread, coalesced
// over the patches.  Nothing of the image vectors is read: the dot products were formed from the attention kernel's
// accumulators.
template <int CC>
__device__ __forceinline__ void heatmap_part_body(const ca_heatmap_problem &P, int L, int C, int norm) {
  const float *part = (const float *)P.img_vec;
  const int heads = P.ldi;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < L; p += gridDim.x * blockDim.x) {
    float z[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) z[c] = 0.f;
    for (int hd = 0; hd < heads; ++hd) {
      const float *q = part + ((size_t)hd * L + p) * 8;
      const f32x4 a = *(const f32x4 *)q;
#pragma unroll
      for (int c = 0; c < 4 && c < CC; ++c) z[c] = z[c] + a[c];
      if constexpr (CC > 4) {
        const f32x4 b = *(const f32x4 *)(q + 4);
#pragma unroll
        for (int c = 4; c < CC; ++c) z[c] = z[c] + b[c - 4];
      }
    }
    hm_weight_and_accumulate<CC>(P, z, p, L, C, norm);
  }
}

template <int CC>
__global__ __launch_bounds__(CC <= 4 ? 256 : 512) void ca_heatmap_fused_kernel(HeatmapLaunch A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *cs = (float *)smem_raw;  // [CC][dim]: rows >= C repeat row C - 1 (their logits are never used)
  const ca_heatmap_problem &P = A.p[blockIdx.y];
  const int dim = A.dim, C = A.C;
  if (P.img_f32 == 2) {   // (workgroup-uniform: a problem of partial logits reads no vectors at all)
    heatmap_part_body<CC>(P, A.L, C, A.norm);
    return;
  }
  for (int c = 0; c < CC; ++c) {
    const int cr = min(c, C - 1);
    for (int k = threadIdx.x * 4; k < dim; k += blockDim.x * 4) {
      f32x4 v;
      if (P.con_f32) {
        v = *(const f32x4 *)((const float *)P.con_vec + (size_t)cr * P.ldc + k);
      } else {
        const bf16x4 t = *(const bf16x4 *)((const bf16 *)P.con_vec + (size_t)cr * P.ldc + k);
        v = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
      }
      *(f32x4 *)(cs + c * dim + k) = v;
    }
  }
  __syncthreads();
  if (P.img_f32) heatmap_fused_body<CC, float>(P, A.L, C, dim, A.norm, cs);
  else heatmap_fused_body<CC, bf16>(P, A.L, C, dim, A.norm, cs);
}

__global__ __launch_bounds__(256) void ca_axpy_kernel(bf16 *__restrict__ x, const bf16 *__restrict__ y, float a,
                                                      long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const bf16x8 xv = *(const bf16x8 *)(x + i), yv = *(const bf16x8 *)(y + i);
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = fmaf(a, (float)yv[j], (float)xv[j]);
    *(uint4 *)(x + i) = make_uint4(ca_pack2(r[0], r[1]), ca_pack2(r[2], r[3]), ca_pack2(r[4], r[5]), ca_pack2(r[6], r[7]));
  } else {
    for (long j = i; j < n; ++j) x[j] = (bf16)fmaf(a, (float)y[j], (float)x[j]);
  }
}

// the same with the state in fp32: x (fp32) += a * y (bf16 or fp32)
template <typename TY>
__global__ __launch_bounds__(256) void ca_axpy_f32_kernel(float *__restrict__ x, const TY *__restrict__ y, float a, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    float yv[8];
    if constexpr (sizeof(TY) == 2) {
      const bf16x8 y8 = *(const bf16x8 *)(y + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) yv[j] = (float)y8[j];
    } else {
      const f32x4 y0 = *(const f32x4 *)(y + i), y1 = *(const f32x4 *)(y + i + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) yv[j] = y0[j], yv[4 + j] = y1[j];
    }
    f32x4 x0 = *(const f32x4 *)(x + i), x1 = *(const f32x4 *)(x + i + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) x0[j] = fmaf(a, yv[j], x0[j]), x1[j] = fmaf(a, yv[4 + j], x1[j]);
    *(f32x4 *)(x + i) = x0;
    *(f32x4 *)(x + i + 4) = x1;
  } else {
    for (long j = i; j < n; ++j) x[j] = fmaf(a, (float)y[j], x[j]);
  }
}

// sinusoidal timestep embedding: out[v, 0:half] = cos(tf*t[v]*f_i), out[v, half:] = sin(...)
__global__ __launch_bounds__(256) void ca_timestep_embedding_kernel(const float *__restrict__ t, int nt,
                                                                    float *__restrict__ out, int dim,
                                                                    float time_factor, float max_period) {
  const int half = dim / 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nt * half) return;
  const int v = i / half, k = i - v * half;
  const float freq = expf(-logf(max_period) * (float)k / (float)half);
  const float arg = time_factor * t[v] * freq;
  out[(size_t)v * dim + k] = cosf(arg);
  out[(size_t)v * dim + half + k] = sinf(arg);
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ca_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}

}  // namespace

namespace {
int ln_modulate_impl(const char *FN, const void *x, int32_t ldx, void *out, int32_t ldo, float *out_scale, int32_t M,
                     int32_t H, const ca_mod_segment *segs, int32_t n_segs, float eps, ca_stream_t stream,
                     bool x_f32 = false, void *out_lo = nullptr, int32_t ldlo = 0) {
  const bool fp8 = out_scale != nullptr;
  if (out_lo && (fp8 || !x_f32 || ldlo % 8 || ldlo < H || ((uintptr_t)out_lo & 15))) {
    ca_set_error("%s: the low plane needs an fp32 input, a bf16 output, ldlo %% 8 == 0 and ldlo >= H", FN);
    return CA_ERR_ARG;
  }
  if (!x || !out || !segs || M < 1 || H < 8 || H % 8 || H > LN_MAXCH * 512 || n_segs < 1 ||
      n_segs > CA_MAX_SEGMENTS || ldx % (x_f32 ? 4 : 8) || ldo % (fp8 ? 16 : 8) || ldx < H || ldo < H ||
      (((uintptr_t)x | (uintptr_t)out) & 15) || ((uintptr_t)out_scale & 3)) {
    ca_set_error("%s: bad arguments (M=%d H=%d n_segs=%d ldx=%d ldo=%d; need H%%8==0, H<=%d)", FN, M, H, n_segs, ldx,
                 ldo, LN_MAXCH * 512);
    return CA_ERR_ARG;
  }
  LnArgs A = {};
  A.n_segs = n_segs;
  int prev = 0;
  for (int i = 0; i < n_segs; ++i) {
    if (!segs[i].shift || !segs[i].scale || segs[i].row_end < prev ||
        (((uintptr_t)segs[i].shift | (uintptr_t)segs[i].scale) & 15)) {
      ca_set_error("%s: segment %d invalid (row_end must be non-decreasing, vectors 16-byte aligned)", FN, i);
      return CA_ERR_ARG;
    }
    prev = segs[i].row_end;
    A.seg[i] = segs[i];
  }
  if (prev < M) {
    ca_set_error("%s: segments cover %d rows, M=%d", FN, prev, M);
    return CA_ERR_ARG;
  }
  const dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  static const bool rows_kernel = ca_ab_env("CA_LN_ROWS", 1) != 0;
  if (rows_kernel && x_f32 && !fp8 && H == 3072) {   // the model's shape: a wave walks 8 rows (vectors kept in registers)
    const int waves = (M + LN_ROWS_PER_WAVE - 1) / LN_ROWS_PER_WAVE;
    const dim3 g2((waves + 3) / 4);
    if (out_lo)
      hipLaunchKernelGGL((ca_ln_modulate_rows_kernel<6, true>), g2, block, 0, st, (const float *)x, ldx, (bf16 *)out, ldo,
                         M, eps, A, (bf16 *)out_lo, ldlo);
    else
      hipLaunchKernelGGL((ca_ln_modulate_rows_kernel<6, false>), g2, block, 0, st, (const float *)x, ldx, (bf16 *)out, ldo,
                         M, eps, A, (bf16 *)nullptr, 0);
    return check_launch(FN);
  }
  if (out_lo)
    hipLaunchKernelGGL((ca_ln_modulate_kernel<false, float, true>), grid, block, 0, st, (const float *)x, ldx, out, ldo,
                       M, H, eps, (float *)nullptr, A, (bf16 *)out_lo, ldlo);
  else if (x_f32 && fp8)
    hipLaunchKernelGGL((ca_ln_modulate_kernel<true, float>), grid, block, 0, st, (const float *)x, ldx, out, ldo, M, H,
                       eps, out_scale, A);
  else if (x_f32)
    hipLaunchKernelGGL((ca_ln_modulate_kernel<false, float>), grid, block, 0, st, (const float *)x, ldx, out, ldo, M,
                       H, eps, (float *)nullptr, A);
  else if (fp8)
    hipLaunchKernelGGL((ca_ln_modulate_kernel<true, bf16>), grid, block, 0, st, (const bf16 *)x, ldx, out, ldo, M, H,
                       eps, out_scale, A);
  else
    hipLaunchKernelGGL((ca_ln_modulate_kernel<false, bf16>), grid, block, 0, st, (const bf16 *)x, ldx, out, ldo, M, H,
                       eps, (float *)nullptr, A);
  return check_launch(FN);
}
}  // namespace

extern "C" int ca_ln_modulate_bf16(const void *x, int32_t ldx, void *out, int32_t ldo, int32_t M, int32_t H,
                                   const ca_mod_segment *segs, int32_t n_segs, float eps, ca_stream_t stream) {
  return ln_modulate_impl("ca_ln_modulate_bf16", x, ldx, out, ldo, nullptr, M, H, segs, n_segs, eps, stream);
}

extern "C" int ca_ln_modulate_fp8(const void *x, int32_t ldx, void *out8, int32_t ldo, float *out_scale, int32_t M,
                                  int32_t H, const ca_mod_segment *segs, int32_t n_segs, float eps,
                                  ca_stream_t stream) {
  if (!out_scale) {
    ca_set_error("ca_ln_modulate_fp8: out_scale is NULL");
    return CA_ERR_ARG;
  }
  return ln_modulate_impl("ca_ln_modulate_fp8", x, ldx, out8, ldo, out_scale, M, H, segs, n_segs, eps, stream);
}

extern "C" int ca_ln_modulate_f32in(const float *x, int32_t ldx, void *out, int32_t ldo, int32_t M, int32_t H,
                                    const ca_mod_segment *segs, int32_t n_segs, float eps, ca_stream_t stream) {
  return ln_modulate_impl("ca_ln_modulate_f32in", x, ldx, out, ldo, nullptr, M, H, segs, n_segs, eps, stream, true);
}

extern "C" int ca_ln_modulate_f32in_split(const float *x, int32_t ldx, void *out, int32_t ldo, void *out_lo,
                                          int32_t ldlo, int32_t M, int32_t H, const ca_mod_segment *segs,
                                          int32_t n_segs, float eps, ca_stream_t stream) {
  if (!out_lo) {
    ca_set_error("ca_ln_modulate_f32in_split: out_lo is NULL");
    return CA_ERR_ARG;
  }
  return ln_modulate_impl("ca_ln_modulate_f32in_split", x, ldx, out, ldo, nullptr, M, H, segs, n_segs, eps, stream,
                          true, out_lo, ldlo);
}

extern "C" int ca_ln_modulate_f32in_fp8(const float *x, int32_t ldx, void *out8, int32_t ldo, float *out_scale,
                                        int32_t M, int32_t H, const ca_mod_segment *segs, int32_t n_segs, float eps,
                                        ca_stream_t stream) {
  if (!out_scale) {
    ca_set_error("ca_ln_modulate_f32in_fp8: out_scale is NULL");
    return CA_ERR_ARG;
  }
  return ln_modulate_impl("ca_ln_modulate_f32in_fp8", x, ldx, out8, ldo, out_scale, M, H, segs, n_segs, eps, stream,
                          true);
}

extern "C" int ca_quantize_rows_fp8(const void *x, int32_t ldx, void *out8, int32_t ldo, float *out_scale, int32_t M,
                                    int32_t K, ca_stream_t stream) {
  if (!x || !out8 || !out_scale || M < 1 || K < 8 || K % 8 || ldx % 8 || ldo % 8 || ldx < K || ldo < K ||
      (((uintptr_t)x) & 15) || (((uintptr_t)out8) & 7) || ((uintptr_t)out_scale & 3)) {
    ca_set_error("ca_quantize_rows_fp8: bad arguments (M=%d K=%d ldx=%d ldo=%d; need K%%8==0, ld%%8==0)", M, K, ldx, ldo);
    return CA_ERR_ARG;
  }
  hipLaunchKernelGGL(ca_quantize_rows_fp8_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     (const bf16 *)x, ldx, (uint8_t *)out8, ldo, out_scale, M, K);
  return check_launch("ca_quantize_rows_fp8");
}

extern "C" int ca_qknorm_rope_bf16(void *qkv, int32_t ld, int32_t M, int32_t num_heads,
                                   const ca_norm_segment *segs, int32_t n_segs, const float *rope_cos_sin,
                                   void *q_prerope, int32_t ldp, ca_stream_t stream) {
  if (!qkv || !segs || !rope_cos_sin || M < 1 || num_heads < 1 || n_segs < 1 || n_segs > CA_MAX_SEGMENTS ||
      ld % 8 || ld < 3 * num_heads * 128 || (((uintptr_t)qkv | (uintptr_t)rope_cos_sin | (uintptr_t)q_prerope) & 15) ||
      (q_prerope && (ldp % 8 || ldp < num_heads * 128))) {
    ca_set_error("ca_qknorm_rope_bf16: bad arguments (M=%d heads=%d ld=%d n_segs=%d)", M, num_heads, ld, n_segs);
    return CA_ERR_ARG;
  }
  NormArgs A = {};
  A.n_segs = n_segs;
  int prev = 0;
  for (int i = 0; i < n_segs; ++i) {
    if (!segs[i].q_scale || !segs[i].k_scale || segs[i].row_end < prev ||
        (((uintptr_t)segs[i].q_scale | (uintptr_t)segs[i].k_scale) & 15)) {
      ca_set_error("ca_qknorm_rope_bf16: segment %d invalid", i);
      return CA_ERR_ARG;
    }
    prev = segs[i].row_end;
    A.seg[i] = segs[i];
  }
  if (prev < M) {
    ca_set_error("ca_qknorm_rope_bf16: segments cover %d rows, M=%d", prev, M);
    return CA_ERR_ARG;
  }
  const long units = (long)M * 2 * num_heads;
  hipLaunchKernelGGL(ca_qknorm_rope_kernel, dim3((unsigned)((units + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                     (bf16 *)qkv, ld, M, num_heads, rope_cos_sin, (bf16 *)q_prerope, ldp, A);
  return check_launch("ca_qknorm_rope_bf16");
}

extern "C" int ca_gemv_bf16(const float *x, int32_t nv, int32_t ldx, const void *W, const void *bias, float *out,
                            int32_t ldo, int32_t N, int32_t K, int32_t silu_input, int32_t accumulate,
                            ca_stream_t stream) {
  if (!x || !W || !out || nv < 1 || nv > 8 || N < 1 || K < 8 || K % 8 || K > 4096 || ldx < K || ldo < N ||
      (((uintptr_t)W) & 15)) {
    ca_set_error("ca_gemv_bf16: bad arguments (nv=%d N=%d K=%d; need 1<=nv<=8, K%%8==0, K<=4096)", nv, N, K);
    return CA_ERR_ARG;
  }
  const int rows_per_block = 4 * GEMV_ROWS;
  const int grid = (N + rows_per_block - 1) / rows_per_block < 4096 ? (N + rows_per_block - 1) / rows_per_block : 4096;
  const size_t lds = (size_t)nv * K * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define CA_GEMV_LAUNCH(NV)                                                                                     \
  hipLaunchKernelGGL(ca_gemv_kernel<NV>, dim3(grid), dim3(256), lds, s, x, ldx, (const bf16 *)W, (const bf16 *)bias, \
                     out, ldo, N, K, silu_input, accumulate)
  if (lds > 64 * 1024) {  // 5..8 vectors of K > 2048: opt in to the large dynamic-LDS carve-out once
    static std::atomic<unsigned long long> attr_done{0};  // one bit per device: the attribute is per device
    const unsigned long long dev_bit = ca_device_bit();
    if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
      hipError_t e = hipSuccess;
      const void *fns[] = {(const void *)ca_gemv_kernel<5>, (const void *)ca_gemv_kernel<6>,
                           (const void *)ca_gemv_kernel<7>, (const void *)ca_gemv_kernel<8>};
      for (const void *f : fns)
        if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4096 * 4);
      if (e != hipSuccess) {
        ca_set_error("ca_gemv_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return CA_ERR_LAUNCH;
      }
      attr_done.fetch_or(dev_bit, std::memory_order_release);  // idempotent: a race only repeats the call
    }
  }
  switch (nv) {
    case 1: CA_GEMV_LAUNCH(1); break;
    case 2: CA_GEMV_LAUNCH(2); break;
    case 3: CA_GEMV_LAUNCH(3); break;
    case 4: CA_GEMV_LAUNCH(4); break;
    case 5: CA_GEMV_LAUNCH(5); break;
    case 6: CA_GEMV_LAUNCH(6); break;
    case 7: CA_GEMV_LAUNCH(7); break;
    default: CA_GEMV_LAUNCH(8); break;
  }
#undef CA_GEMV_LAUNCH
  return check_launch("ca_gemv_bf16");
}

extern "C" int ca_heatmap_logits_bf16(const void *img_vec, int32_t ldi, const void *con_vec, int32_t ldc,
                                      int32_t con_is_f32, int32_t L, int32_t C, int32_t dim, float *logits,
                                      ca_stream_t stream) {
  const bool con32 = con_is_f32 & 1, img32 = con_is_f32 & 2;   // bit 0: concept vectors fp32, bit 1: image vectors fp32
  if (!img_vec || !con_vec || !logits || L < 1 || C < 1 || dim < 8 || dim % 8 || dim > 4096 || ldi % 8 ||
      ldi < dim || ldc < dim || ldc % 4 || (((uintptr_t)img_vec | (uintptr_t)con_vec) & 15) || (con_is_f32 & ~3) ||
      (img32 && !con32)) {
    ca_set_error("ca_heatmap_logits_bf16: bad arguments (L=%d C=%d dim=%d ldi=%d ldc=%d)", L, C, dim, ldi, ldc);
    return CA_ERR_ARG;
  }
  // 8 patches per workgroup and pass (2 per wave): the 4 * dim floats of concept vectors a workgroup puts into LDS are
  // then read once per 8 patches, and at most 512 workgroups keep them to 2 per CU's worth of L2 reads
  const int grid = (L + 7) / 8 < 512 ? (L + 7) / 8 : 512;
  const size_t lds = (size_t)4 * dim * sizeof(float);
  for (int c0 = 0; c0 < C; c0 += 4) {
    if (img32)
      hipLaunchKernelGGL((ca_heatmap_logits_kernel<4, float, float>), dim3(grid), dim3(256), lds, (hipStream_t)stream,
                         (const float *)img_vec, ldi, (const float *)con_vec, ldc, L, C, c0, dim, logits);
    else if (con32)
      hipLaunchKernelGGL((ca_heatmap_logits_kernel<4, float>), dim3(grid), dim3(256), lds, (hipStream_t)stream,
                         (const bf16 *)img_vec, ldi, (const float *)con_vec, ldc, L, C, c0, dim, logits);
    else
      hipLaunchKernelGGL((ca_heatmap_logits_kernel<4, bf16>), dim3(grid), dim3(256), lds, (hipStream_t)stream,
                         (const bf16 *)img_vec, ldi, (const bf16 *)con_vec, ldc, L, C, c0, dim, logits);
    const int rc = check_launch("ca_heatmap_logits_bf16");
    if (rc) return rc;
  }
  return CA_OK;
}

extern "C" int ca_qpre_finish_rope_f32(float *x, int32_t ldx, const float *d, int32_t ldd, const void *norm_scale,
                                       const float *rope, void *q_out, int32_t ldq, float q_out_scale, int32_t q_f16,
                                       int32_t M, int32_t heads, ca_stream_t stream) {
  if (!x || !norm_scale || M < 1 || heads < 1 || ldx % 4 || ldx < heads * 128 || (d && (ldd % 4 || ldd < heads * 128)) ||
      (((uintptr_t)x | (uintptr_t)d | (uintptr_t)norm_scale) & 15)) {
    ca_set_error("ca_qpre_finish_f32: bad arguments (M=%d heads=%d ldx=%d ldd=%d)", M, heads, ldx, ldd);
    return CA_ERR_ARG;
  }
  if (q_out && (!rope || ldq % 8 || ldq < heads * 128 || (((uintptr_t)rope | (uintptr_t)q_out) & 15) ||
                !(q_out_scale >= 0.0f) || (q_f16 != 0 && q_f16 != 1))) {
    ca_set_error("ca_qpre_finish_rope_f32: q_out needs rope [M,64,2], ldq %% 8 == 0, ldq >= heads*128, 16-byte alignment");
    return CA_ERR_ARG;
  }
  const long threads = (long)M * heads * 16;
  hipLaunchKernelGGL(ca_qpre_finish_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, d, ldd, (const bf16 *)norm_scale, M, heads, rope, q_out, ldq, q_out_scale, q_f16);
  return check_launch("ca_qpre_finish_f32");
}

extern "C" int ca_qpre_finish_f32(float *x, int32_t ldx, const float *d, int32_t ldd, const void *norm_scale, int32_t M,
                                  int32_t heads, ca_stream_t stream) {
  return ca_qpre_finish_rope_f32(x, ldx, d, ldd, norm_scale, nullptr, nullptr, 0, 0.0f, 0, M, heads, stream);
}

extern "C" int ca_heatmap_softmax_accumulate(const float *logits, int32_t C, int32_t L, float weight, float *acc,
                                             ca_stream_t stream) {
  if (!logits || !acc || C < 1 || L < 1) {
    ca_set_error("ca_heatmap_softmax_accumulate: bad arguments (C=%d L=%d)", C, L);
    return CA_ERR_ARG;
  }
  hipLaunchKernelGGL(ca_heatmap_softmax_kernel, dim3((L + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, C, L,
                     weight, acc);
  return check_launch("ca_heatmap_softmax_accumulate");
}

extern "C" int ca_heatmap_norm_accumulate(const float *logits, int32_t C, int32_t L, int32_t norm, float weight,
                                          float *acc, ca_stream_t stream) {
  if (norm == CA_NORM_SOFTMAX) return ca_heatmap_softmax_accumulate(logits, C, L, weight, acc, stream);
  if (!logits || !acc || C < 1 || C > 16 || L < 1 || (norm != CA_NORM_SPARSEMAX && norm != CA_NORM_ENTMAX15)) {
    ca_set_error("ca_heatmap_norm_accumulate: bad arguments (C=%d L=%d norm=%d; sparse norms need C <= 16)", C, L, norm);
    return CA_ERR_ARG;
  }
  const dim3 grid((L + 255) / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (norm == CA_NORM_SPARSEMAX) {
    if (C <= 8) hipLaunchKernelGGL((ca_heatmap_sparse_kernel<8, false>), grid, block, 0, s, logits, C, L, weight, acc);
    else hipLaunchKernelGGL((ca_heatmap_sparse_kernel<16, false>), grid, block, 0, s, logits, C, L, weight, acc);
  } else {
    if (C <= 8) hipLaunchKernelGGL((ca_heatmap_sparse_kernel<8, true>), grid, block, 0, s, logits, C, L, weight, acc);
    else hipLaunchKernelGGL((ca_heatmap_sparse_kernel<16, true>), grid, block, 0, s, logits, C, L, weight, acc);
  }
  return check_launch("ca_heatmap_norm_accumulate");
}

extern "C" int ca_heatmap_fused(const ca_heatmap_problem *problems, int32_t n_problems, int32_t L, int32_t C,
                                int32_t dim, int32_t norm, ca_stream_t stream) {
  if (!problems || n_problems < 1 || n_problems > CA_HEATMAP_MAX_PROBLEMS || L < 1 || C < 1 || C > 8 || dim < 8 ||
      dim % 8 || dim > 4096 || (norm != CA_NORM_SOFTMAX && norm != CA_NORM_SPARSEMAX && norm != CA_NORM_ENTMAX15)) {
    ca_set_error("ca_heatmap_fused: bad arguments (n_problems=%d L=%d C=%d dim=%d norm=%d; need C <= 8, dim %% 8 == 0, "
                 "dim <= 4096)", n_problems, L, C, dim, norm);
    return CA_ERR_ARG;
  }
  HeatmapLaunch A = {};
  A.L = L, A.C = C, A.dim = dim, A.norm = norm;
  for (int i = 0; i < n_problems; ++i) {
    const ca_heatmap_problem &p = problems[i];
    const bool part = p.img_f32 == 2;   // per-head partial logits [ldi heads][L][8] instead of vectors
    if (!p.img_vec || (!p.acc && !p.acc2 && !p.logits) || ((uintptr_t)p.img_vec & 15) ||
        (((uintptr_t)p.acc | (uintptr_t)p.acc2 | (uintptr_t)p.logits) & 3) || p.img_f32 < 0 || p.img_f32 > 2 ||
        (part ? (p.ldi < 1 || p.ldi > 1024)
              : (!p.con_vec || p.ldi < dim || p.ldc < dim || p.ldi % (p.img_f32 ? 4 : 8) || p.ldc % 4 ||
                 (p.con_f32 & ~1) || ((uintptr_t)p.con_vec & 15)))) {
      ca_set_error("ca_heatmap_fused: problem %d invalid (ldi=%d ldc=%d img_f32=%d con_f32=%d; vectors 16-byte aligned, "
                   "at least one of acc / acc2 / logits)", i, p.ldi, p.ldc, p.img_f32, p.con_f32);
      return CA_ERR_ARG;
    }
    A.p[i] = p;
  }
  const int cc = C <= 4 ? 4 : 8, threads = cc == 4 ? 256 : 512;
  const size_t lds = (size_t)cc * dim * sizeof(float);
  if (lds > 64 * 1024) {   // C > 4 at the model's dim: above the default dynamic LDS limit
    static std::atomic<unsigned long long> attr_done{0};
    const unsigned long long dev_bit = ca_device_bit();
    if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
      const hipError_t e = hipFuncSetAttribute((const void *)ca_heatmap_fused_kernel<8>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4096 * (int)sizeof(float));
      if (e != hipSuccess) {
        ca_set_error("ca_heatmap_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return CA_ERR_LAUNCH;
      }
      attr_done.fetch_or(dev_bit, std::memory_order_release);
    }
  }
  // workgroups per problem: a wave takes 2 patches per pass; about 3 (C <= 4: 48 KB of LDS each) or 2 workgroups per CU
  // in all, so that the concept vectors are pulled into LDS a few hundred times per launch, not once per 8 patches
  const int n_cu = ca_cu_count() > 0 ? ca_cu_count() : 256;
  const int per_pass = threads / 64 * 2;
  int gx = ((cc == 4 ? 3 : 2) * n_cu + n_problems - 1) / n_problems;
  gx = gx < (L + per_pass - 1) / per_pass ? gx : (L + per_pass - 1) / per_pass;
  if (gx < 1) gx = 1;
  if (cc == 4)
    hipLaunchKernelGGL(ca_heatmap_fused_kernel<4>, dim3(gx, n_problems), dim3(threads), lds, (hipStream_t)stream, A);
  else
    hipLaunchKernelGGL(ca_heatmap_fused_kernel<8>, dim3(gx, n_problems), dim3(threads), lds, (hipStream_t)stream, A);
  return check_launch("ca_heatmap_fused");
}

namespace {
template <bool SILU>
__global__ __launch_bounds__(256) void ca_silu_split_kernel(const float *__restrict__ x, int ldx, bf16 *__restrict__ hi,
                                                            bf16 *__restrict__ lo, int ldo, int rows, int K) {
  const int per_row = K >> 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * per_row; i += gridDim.x * 256) {
    const int r = i / per_row, k = (i - r * per_row) << 2;
    const f32x4 v = *(const f32x4 *)(x + (size_t)r * ldx + k);
    bf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = SILU ? ca_silu(v[j]) : v[j];
      h[j] = (bf16)s;
      l[j] = (bf16)(s - (float)h[j]);
    }
    *(bf16x4 *)(hi + (size_t)r * ldo + k) = h;
    *(bf16x4 *)(lo + (size_t)r * ldo + k) = l;
  }
}
}  // namespace

extern "C" int ca_silu_split_bf16(const float *x, int32_t ldx, void *hi, void *lo, int32_t ldo, int32_t rows, int32_t K,
                                  ca_stream_t stream) {
  if (!x || !hi || !lo || rows < 1 || K < 4 || K % 4 || ldx < K || ldo < K || ldx % 4 || ldo % 4 ||
      (((uintptr_t)x & 15) | (((uintptr_t)hi | (uintptr_t)lo) & 7))) {
    ca_set_error("ca_silu_split_bf16: bad arguments (rows=%d K=%d ldx=%d ldo=%d)", rows, K, ldx, ldo);
    return CA_ERR_ARG;
  }
  const long n = (long)rows * (K / 4);
  const long blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
  hipLaunchKernelGGL(ca_silu_split_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     (bf16 *)hi, (bf16 *)lo, ldo, rows, K);
  return check_launch("ca_silu_split_bf16");
}

extern "C" int ca_split_bf16(const float *x, int32_t ldx, void *hi, void *lo, int32_t ldo, int32_t rows, int32_t K,
                             ca_stream_t stream) {
  if (!x || !hi || !lo || rows < 1 || K < 4 || K % 4 || ldx < K || ldo < K || ldx % 4 || ldo % 4 ||
      (((uintptr_t)x & 15) | (((uintptr_t)hi | (uintptr_t)lo) & 7))) {
    ca_set_error("ca_split_bf16: bad arguments (rows=%d K=%d ldx=%d ldo=%d)", rows, K, ldx, ldo);
    return CA_ERR_ARG;
  }
  const long n = (long)rows * (K / 4);
  const long blocks = (n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024;
  hipLaunchKernelGGL(ca_silu_split_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     (bf16 *)hi, (bf16 *)lo, ldo, rows, K);
  return check_launch("ca_split_bf16");
}

namespace {
__global__ __launch_bounds__(256) void ca_modulation_combine_kernel(const float *__restrict__ pair, int ldp,
                                                                    const bf16 *__restrict__ bias, float *__restrict__ out,
                                                                    int ldo, int nv, int N) {
  const long per_row = N >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv * per_row; i += (long)gridDim.x * 256) {
    const int v = (int)(i / per_row), n = (int)(i - v * per_row) << 2;
    const f32x4 h = *(const f32x4 *)(pair + (size_t)v * ldp + n), l = *(const f32x4 *)(pair + (size_t)(nv + v) * ldp + n);
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
      const bf16x4 b4 = *(const bf16x4 *)(bias + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = (float)b4[j];
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (h[j] + b[j]) + l[j];
    *(f32x4 *)(out + (size_t)v * ldo + n) = o;
  }
}
}  // namespace

extern "C" int ca_modulation_combine_f32(const float *pair, int32_t ldp, const void *bias, float *out, int32_t ldo,
                                         int32_t nv, int32_t N, ca_stream_t stream) {
  if (!pair || !out || nv < 1 || N < 4 || N % 4 || ldp < N || ldo < N || ldp % 4 || ldo % 4 ||
      ((((uintptr_t)pair | (uintptr_t)out) & 15) | ((uintptr_t)bias & 7))) {
    ca_set_error("ca_modulation_combine_f32: bad arguments (nv=%d N=%d ldp=%d ldo=%d)", nv, N, ldp, ldo);
    return CA_ERR_ARG;
  }
  const long n = (long)nv * (N / 4);
  const long blocks = (n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048;
  hipLaunchKernelGGL(ca_modulation_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pair, ldp,
                     (const bf16 *)bias, out, ldo, nv, N);
  return check_launch("ca_modulation_combine_f32");
}

extern "C" int ca_axpy_bf16(void *x, const void *y, float a, int64_t n, ca_stream_t stream) {
  if (!x || !y || n < 1 || (((uintptr_t)x | (uintptr_t)y) & 15)) {
    ca_set_error("ca_axpy_bf16: bad arguments (n=%lld)", (long long)n);
    return CA_ERR_ARG;
  }
  const long blocks = (n + 2047) / 2048;
  hipLaunchKernelGGL(ca_axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (bf16 *)x,
                     (const bf16 *)y, a, (long)n);
  return check_launch("ca_axpy_bf16");
}

extern "C" int ca_axpy_f32(float *x, const void *y, int32_t y_is_f32, float a, int64_t n, ca_stream_t stream) {
  if (!x || !y || n < 1 || (((uintptr_t)x | (uintptr_t)y) & 15)) {
    ca_set_error("ca_axpy_f32: bad arguments (n=%lld)", (long long)n);
    return CA_ERR_ARG;
  }
  const long blocks = (n + 2047) / 2048;
  if (y_is_f32)
    hipLaunchKernelGGL(ca_axpy_f32_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                       (const float *)y, a, (long)n);
  else
    hipLaunchKernelGGL(ca_axpy_f32_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                       (const bf16 *)y, a, (long)n);
  return check_launch("ca_axpy_f32");
}

extern "C" int ca_timestep_embedding_f32(const float *t, int32_t nt, float *out, int32_t dim, float time_factor,
                                         float max_period, ca_stream_t stream) {
  if (!t || !out || nt < 1 || dim < 2 || dim % 2) {
    ca_set_error("ca_timestep_embedding_f32: bad arguments (nt=%d dim=%d)", nt, dim);
    return CA_ERR_ARG;
  }
  const int n = nt * (dim / 2);
  hipLaunchKernelGGL(ca_timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, nt,
                     out, dim, time_factor, max_period);
  return check_launch("ca_timestep_embedding_f32");
}
