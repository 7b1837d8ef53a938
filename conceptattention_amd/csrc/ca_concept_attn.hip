// Attention for a handful of query rows (the C <= 8 concept rows of a work item: modified_double_stream_block.py:162-168,
// of which only the C concept query rows are ever used) as a bandwidth problem instead of an MFMA one.  In the MFMA
// kernel such a problem is one workgroup per head that sweeps all keys with 4 of 256 query rows valid (190 us for the 120
// units of a 5-item launch, each pulling its head's K / V alone); here the keys are split over workgroups (512 per
// workgroup: 9 x 24 x 5 = 1080 of them), every score is an fp32 dot product, every probability stays fp32, and a second
// tiny kernel merges the per-chunk softmax partials.  q carries softmax_scale * log2(e) (CA_ATTN_Q_PRESCALED), q / k are
// bf16 or IEEE half (qk_f16), v is bf16.  The key -> chunk split depends on the key count only, so a problem's bits do not
// depend on the launch it shares.
#include <type_traits>

#include "ca_common.h"

namespace {

constexpr int CC_MAX = 8;        // query rows per problem
constexpr int CHUNK = 512;       // keys per workgroup: 4 waves x 32 steps x 4 keys
constexpr int PART_STRIDE = 136; // floats per (chunk, row): O[128], m, l (+ pad to 16-byte multiples)

struct ConceptLaunch {
  ca_attn_problem p[CA_ATTN_MAX_PROBLEMS];
  float *ws;
  int32_t num_heads, qk_f16, max_chunks;
};

__device__ __forceinline__ void decode8(const uint4 raw, bool f16, float (&out)[8]) {
  if (f16) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8 v = __builtin_bit_cast(h8, raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = (float)v[j];
  } else {
    const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = (float)v[j];
  }
}

// sum over the 16 lanes of a DPP row (the lanes that share a key), result in every lane: two quad permutes, the half
// mirror and the mirror -- four v_add_f32 with a DPP operand instead of four ds_bpermute round trips through LDS
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
  return v;
}

// merge (m2, l2, o2) into (m, l, o): both are partial softmax states of the same row over disjoint key sets
__device__ __forceinline__ void merge_state(float &m, float &l, float (&o)[8], float m2, float l2, const float (&o2)[8]) {
  const float mn = fmaxf(m, m2);
  const float a = __builtin_amdgcn_exp2f(m - mn), b = __builtin_amdgcn_exp2f(m2 - mn);
  l = l * a + l2 * b;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = o[j] * a + o2[j] * b;
  m = mn;
}

// grid (chunks, heads, problems); 256 threads.  Lane = (key slot ks = lane >> 4, dim slot ds = lane & 15): 16 lanes share
// a key row (8 dims each), a wave takes 4 keys per step.
template <int CC>   // 4 or 8: query rows the register arrays are sized for (C <= 4: 130 instead of 208 VGPRs)
__global__ __launch_bounds__(256) void ca_concept_attn_partial_kernel(const ConceptLaunch A) {
  const ca_attn_problem &P = A.p[blockIdx.z];
  const int head = blockIdx.y, chunk = blockIdx.x;
  const int nk = P.n0 + P.n1, C = P.nq;
  if (chunk * CHUNK >= nk) return;   // (problems of different key counts share a grid)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ks = lane >> 4, ds = lane & 15;
  const bool f16 = A.qk_f16 != 0;
  float q[CC][8];
#pragma unroll
  for (int c = 0; c < CC; ++c) {
    if (c < C) decode8(*(const uint4 *)((const bf16 *)P.q + (size_t)c * P.ldq + head * 128 + ds * 8), f16, q[c]);
    else
#pragma unroll
      for (int j = 0; j < 8; ++j) q[c][j] = 0.f;
  }
  float m[CC], l[CC], o[CC][8];
#pragma unroll
  for (int c = 0; c < CC; ++c) {
    m[c] = -1e30f, l[c] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[c][j] = 0.f;
  }
  const int k_begin = chunk * CHUNK + wave * (CHUNK / 4);
  constexpr int AHEAD = 4;   // steps whose 16-byte K and V pieces are requested before the first is used (8 KB per wave)
  for (int step0 = 0; step0 < CHUNK / 16; step0 += AHEAD) {
    uint4 kr[AHEAD], vr[AHEAD];
    bool okv[AHEAD];
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      const int key = k_begin + (step0 + u) * 4 + ks;
      okv[u] = key < nk;
      const int kc = okv[u] ? key : nk - 1;
      const bool seg0 = kc < P.n0;
      const size_t off = (size_t)(seg0 ? kc : kc - P.n0) * P.ldkv + head * 128 + ds * 8;
      kr[u] = *(const uint4 *)((seg0 ? (const bf16 *)P.k0 : (const bf16 *)P.k1) + off);
      vr[u] = *(const uint4 *)((seg0 ? (const bf16 *)P.v0 : (const bf16 *)P.v1) + off);
    }
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      const bool ok = okv[u];
      float kf[8], vf[8];
      decode8(kr[u], f16, kf);
      decode8(vr[u], false, vf);
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        if (c < C) {   // (C is uniform per problem: no divergence)
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) s = __builtin_fmaf(q[c][j], kf[j], s);
          s = row16_sum(s);
          if (!ok) s = -1e30f;
          const float mn = fmaxf(m[c], s);
          const float a = __builtin_amdgcn_exp2f(m[c] - mn), pr = ok ? __builtin_amdgcn_exp2f(s - mn) : 0.f;
          l[c] = l[c] * a + pr;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[c][j] = __builtin_fmaf(pr, vf[j], o[c][j] * a);
          m[c] = mn;
        }
      }
    }
  }
  // the 4 key slots of the wave (lanes ds, ds + 16, ds + 32, ds + 48), then the 4 waves through LDS
  __shared__ float sh[4][CC][16][10];   // [wave][row][dim slot][8 dims, m, l]
#pragma unroll
  for (int c = 0; c < CC; ++c) {
    if (c < C) {
#pragma unroll
      for (int x = 16; x <= 32; x <<= 1) {
        const float m2 = __shfl_xor(m[c], x), l2 = __shfl_xor(l[c], x);
        float o2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o2[j] = __shfl_xor(o[c][j], x);
        merge_state(m[c], l[c], o[c], m2, l2, o2);
      }
      if (ks == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[wave][c][ds][j] = o[c][j];
        sh[wave][c][ds][8] = m[c], sh[wave][c][ds][9] = l[c];
      }
    }
  }
  __syncthreads();
  if (wave == 0 && ks == 0) {
    for (int c = 0; c < C; ++c) {
      float mm = sh[0][c][ds][8], ll = sh[0][c][ds][9], oo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) oo[j] = sh[0][c][ds][j];
      for (int w = 1; w < 4; ++w) {
        float o2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o2[j] = sh[w][c][ds][j];
        merge_state(mm, ll, oo, sh[w][c][ds][8], sh[w][c][ds][9], o2);
      }
      float *dst = A.ws + ((((size_t)blockIdx.z * A.num_heads + head) * A.max_chunks + chunk) * CC_MAX + c) * PART_STRIDE;
      *(f32x4 *)(dst + ds * 8) = f32x4{oo[0], oo[1], oo[2], oo[3]};
      *(f32x4 *)(dst + ds * 8 + 4) = f32x4{oo[4], oo[5], oo[6], oo[7]};
      if (ds == 0) dst[128] = mm, dst[129] = ll;
    }
  }
}

// grid (heads, problems); 128 threads = one per dim; the chunks of a row in chunk order
__global__ __launch_bounds__(128) void ca_concept_attn_merge_kernel(const ConceptLaunch A) {
  const ca_attn_problem &P = A.p[blockIdx.y];
  const int head = blockIdx.x, d = threadIdx.x, C = P.nq;
  const int chunks = (P.n0 + P.n1 + CHUNK - 1) / CHUNK;
  for (int c = 0; c < C; ++c) {
    const float *src = A.ws + ((((size_t)blockIdx.y * A.num_heads + head) * A.max_chunks) * CC_MAX + c) * PART_STRIDE;
    float M = -1e30f;
    for (int s = 0; s < chunks; ++s) M = fmaxf(M, src[(size_t)s * CC_MAX * PART_STRIDE + 128]);
    float Lsum = 0.f, O = 0.f;
    for (int s = 0; s < chunks; ++s) {
      const float *ps = src + (size_t)s * CC_MAX * PART_STRIDE;
      const float w = __builtin_amdgcn_exp2f(ps[128] - M);
      Lsum = __builtin_fmaf(ps[129], w, Lsum);
      O = __builtin_fmaf(ps[d], w, O);
    }
    const float r = O / Lsum;
    ((bf16 *)P.out)[(size_t)c * P.ldo + head * 128 + d] = (bf16)r;
    if (P.out_f32) P.out_f32[(size_t)c * P.ldo32 + head * 128 + d] = r;
  }
}

}  // namespace

extern "C" int ca_concept_attn_fwd(const ca_attn_problem *problems, int32_t n_problems, int32_t num_heads, int32_t qk_f16,
                                   float *workspace, int64_t workspace_floats, ca_stream_t stream) {
  if (!problems || n_problems < 1 || n_problems > CA_ATTN_MAX_PROBLEMS || num_heads < 1 || !workspace ||
      ((uintptr_t)workspace & 15) || (qk_f16 != 0 && qk_f16 != 1)) {
    ca_set_error("ca_concept_attn_fwd: n_problems=%d (max %d) num_heads=%d, workspace 16-byte aligned", n_problems,
                 CA_ATTN_MAX_PROBLEMS, num_heads);
    return CA_ERR_ARG;
  }
  ConceptLaunch A = {};
  A.ws = workspace, A.num_heads = num_heads, A.qk_f16 = qk_f16;
  int max_chunks = 1;
  for (int i = 0; i < n_problems; ++i) {
    const ca_attn_problem &p = problems[i];
    if (!p.q || !p.out || !p.k0 || !p.v0 || p.nq < 1 || p.nq > CC_MAX || p.n0 < 1 || p.n1 < 0 ||
        (p.n1 > 0 && (!p.k1 || !p.v1)) || (p.nq0 != 0 && p.nq0 != p.nq) || p.ldq % 8 || p.ldo % 8 || p.ldkv % 8 ||
        p.ldq < num_heads * 128 || p.ldo < num_heads * 128 || p.ldkv < num_heads * 128 ||
        (p.out_f32 && (p.ldo32 % 4 || p.ldo32 < num_heads * 128)) ||
        (((uintptr_t)p.q | (uintptr_t)p.out | (uintptr_t)p.k0 | (uintptr_t)p.v0 | (uintptr_t)p.k1 | (uintptr_t)p.v1 |
          (uintptr_t)p.out_f32) & 15)) {
      ca_set_error("ca_concept_attn_fwd[%d]: 1 <= nq <= %d query rows in one segment, strides >= num_heads*128 and "
                   "multiples of 8, 16-byte aligned pointers", i, CC_MAX);
      return CA_ERR_ARG;
    }
    A.p[i] = p;
    if (p.n1 == 0) A.p[i].k1 = p.k0, A.p[i].v1 = p.v0;
    const int ch = (p.n0 + p.n1 + CHUNK - 1) / CHUNK;
    if (ch > max_chunks) max_chunks = ch;
  }
  A.max_chunks = max_chunks;
  const int64_t need = (int64_t)n_problems * num_heads * max_chunks * CC_MAX * PART_STRIDE;
  if (workspace_floats < need) {
    ca_set_error("ca_concept_attn_fwd: workspace of %lld floats, %lld needed (problems x heads x ceil(keys / %d) x %d x %d)",
                 (long long)workspace_floats, (long long)need, CHUNK, CC_MAX, PART_STRIDE);
    return CA_ERR_ARG;
  }
  int cmax = 1;
  for (int i = 0; i < n_problems; ++i) cmax = problems[i].nq > cmax ? problems[i].nq : cmax;
  if (cmax <= 4)
    hipLaunchKernelGGL(ca_concept_attn_partial_kernel<4>, dim3(max_chunks, num_heads, n_problems), dim3(256), 0,
                       (hipStream_t)stream, A);
  else
    hipLaunchKernelGGL(ca_concept_attn_partial_kernel<8>, dim3(max_chunks, num_heads, n_problems), dim3(256), 0,
                       (hipStream_t)stream, A);
  hipLaunchKernelGGL(ca_concept_attn_merge_kernel, dim3(num_heads, n_problems), dim3(128), 0, (hipStream_t)stream, A);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ca_set_error("ca_concept_attn_fwd: launch failed: %s", hipGetErrorString(e));
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}
