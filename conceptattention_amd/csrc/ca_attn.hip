// Flash-attention forward for gfx950 (MI355X), head_dim 128, bf16 in/out, fp32 softmax state.
//
// One workgroup = 8 waves = 256 query rows of one head (32 rows per wave, Q fragments in
// registers); K/V tiles of 64 keys are register-staged (issue early, ds_write late) into a
// 2-deep LDS ring shared by the 8 waves.
//
//  * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (K fragment as the A operand): every lane then
//    owns ONE query row (column lane&31 of S^T), so the row max / row sum are in-lane reductions
//    plus one exchange with lane^32, and the online-softmax rescale of O^T is a per-lane scalar.
//  * The exponentiated S^T accumulator registers are converted pairwise to bf16 and used directly
//    as the B operand of O^T += V^T P^T (no LDS round trip for P); the k-order permutation this
//    implies is matched by reading V^T with ds_read_b64_tr_b16 from the row-major V tile.
//  * LDS images: K [64][256 B] with 16-B chunk ^= key&15 (conflict-free ds_read_b128 of 32 keys);
//    V [64][256 B] with chunk ^= ((key&3)<<2)|((key>>2)&3) (conflict-free transposed reads).
//  * A problem's key/value set is the concatenation of two row segments, so the concept rows
//    attend to [concept keys ; image keys] straight out of the projection buffers; a second
//    problem (the concept query rows) rides in the same launch on otherwise idle CUs.
//
// Replaces F.scaled_dot_product_attention of the reference (see include/conceptattn.h).
#include "ca_common.h"

namespace {

struct AttnLaunch {
  ca_attn_problem p[2];
  int32_t nqb[2];  // 256-row query blocks per head
  int32_t num_heads;
  int32_t blocks_p1;  // workgroups of problem 1 (they come first in the grid)
  float scale_log2;   // softmax scale * log2(e)
};

constexpr int KV_TILE = 64;
constexpr int TILE_BYTES = KV_TILE * 256;  // one K or V tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;
constexpr int ATTN_LDS = 2 * BUF_BYTES;

__device__ __forceinline__ bf16x8 pack8(const f32x16 &s, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)s[base + j];
  return r;
}

__global__ __launch_bounds__(512, 2) void ca_attn_kernel(const AttnLaunch L) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which (problem, head, query block)
  int bid = blockIdx.x;
  const int prob = bid < L.blocks_p1 ? 1 : 0;
  if (!prob) bid -= L.blocks_p1;
  const int nqb = L.nqb[prob];
  // heads are dealt to the 8 XCD groups (blockIdx % 8) so that a head's query blocks share an L2
  const int xg = bid & 7, idx = bid >> 3;
  const int head = xg + 8 * (idx / nqb);
  const int qb = idx % nqb;
  if (head >= L.num_heads) return;  // whole workgroup exits together
  const ca_attn_problem &P = L.p[prob];
  const int nq = P.nq, n0 = P.n0, nkeys = P.n0 + P.n1;
  const int ldkv = P.ldkv;

  const int h = lane >> 5;    // lane half
  const int ql = lane & 31;   // query row within the wave / operand row
  const int qrow0 = qb * 256 + wave * 32;
  const bool active = qrow0 < nq;  // wave-uniform
  const int qrow = min(qrow0 + ql, nq - 1);

  // ---- Q fragments (B operand: lane holds Q[q = lane&31][d = 16*ks + 8*h + j])
  bf16x8 qf[8];
  {
    const bf16 *qp = (const bf16 *)P.q + (size_t)qrow * P.ldq + head * 128 + h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8 *)(qp + ks * 16);
  }

  // ---- staging geometry: thread moves pieces (tid) and (tid+512): key row r, 16-B chunk ch
  const int st_r = tid >> 4, st_ch = tid & 15;
  const int st_koff = st_r * 256 + ((st_ch ^ (st_r & 15)) << 4);
  const int st_voff = st_r * 256 + ((st_ch ^ (((st_r & 3) << 2) | ((st_r >> 2) & 3))) << 4);
  const bf16 *k0p = (const bf16 *)P.k0 + head * 128 + st_ch * 8;
  const bf16 *v0p = (const bf16 *)P.v0 + head * 128 + st_ch * 8;
  const bf16 *k1p = (const bf16 *)P.k1 + head * 128 + st_ch * 8;
  const bf16 *v1p = (const bf16 *)P.v1 + head * 128 + st_ch * 8;

  uint4 kreg0, kreg1, vreg0, vreg1;  // named (not arrays): keeps them in VGPRs
#define CA_ATTN_LOAD_ROW(I, KR, VR)                                      \
  {                                                                      \
    const int kk = min(tile * KV_TILE + st_r + 32 * (I), nkeys - 1);     \
    const bool s0 = kk < n0;                                             \
    const size_t ro = (size_t)(s0 ? kk : kk - n0) * ldkv;                \
    KR = *(const uint4 *)((s0 ? k0p : k1p) + ro);                        \
    VR = *(const uint4 *)((s0 ? v0p : v1p) + ro);                        \
  }
  auto issue_loads = [&](int tile) {
    CA_ATTN_LOAD_ROW(0, kreg0, vreg0)
    CA_ATTN_LOAD_ROW(1, kreg1, vreg1)
  };
  auto write_lds = [&](int buf) {
    char *kb = smem + buf * BUF_BYTES;
    *(uint4 *)(kb + st_koff) = kreg0;
    *(uint4 *)(kb + TILE_BYTES + st_voff) = vreg0;
    *(uint4 *)(kb + st_koff + 8192) = kreg1;
    *(uint4 *)(kb + TILE_BYTES + st_voff + 8192) = vreg1;
  };

  // ---- fragment read offsets
  // K (A operand of S^T): row = 32*kb + ql, chunk = (2*ks + h) ^ (ql & 15)
  const int k_lane = ql * 256 + (((h ^ (ql & 15)) & 15) << 4);
  // V^T (A operand of O^T): transposed read, lane supplies row qq of a 4x16 block
  const int qq = (lane & 15) >> 2;
  const int c_lane = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
  int v_lane[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int x = (qq << 2) | ((2 * jj + h) & 3);
    v_lane[jj] = (4 * h + qq) * 256 + (((c_lane ^ x) & 15) << 4) + 8 * (lane & 1);
  }

  f32x16 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sl2 = L.scale_log2;

  const int nt = (nkeys + KV_TILE - 1) / KV_TILE;
  issue_loads(0);
  write_lds(0);
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) issue_loads(t + 1);
    if (active) {
      const char *kbuf = smem + cur * BUF_BYTES;
      const char *vbuf = kbuf + TILE_BYTES;
      // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]
      f32x16 s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const bf16x8 kf = *(const bf16x8 *)(kbuf + kb * 8192 + (k_lane ^ (ks << 5)));
          s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
        }
      }
      // ---- mask the ragged last tile: key = 64*t + 32*kb + (r&3) + 8*(r>>2) + 4*h
      if (t == nt - 1 && (nkeys & (KV_TILE - 1))) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KV_TILE + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (key >= nkeys) s[kb][r] = -INFINITY;
          }
      }
      // ---- online softmax (this lane: query row ql, 32 of the tile's 64 keys; lane^32 has the rest)
      float mx = s[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx * sl2);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float rs = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sl2, -m_new));
          s[kb][r] = p;
          rs += p;
        }
      l_run = l_run * alpha + rs;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      // ---- O^T[d][q] += sum_key V[key][d] P[q][key]; P^T fragments straight from the S^T registers
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
          const bf16x8 pf = pack8(s[kb], 8 * sk);
          const char *vrow = vbuf + (32 * kb + 16 * sk) * 256;
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4 *)(vrow + (v_lane[0] ^ (db << 6))));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4 *)(vrow + 8 * 256 + (v_lane[1] ^ (db << 6))));
            const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
          }
        }
      }
    }
    if (t + 1 < nt) write_lds(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l ;  d = 32*db + (r&3) + 8*(r>>2) + 4*h
  if (active) {
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qrow0 + ql < nq) {
      bf16 *op = (bf16 *)P.out + (size_t)(qrow0 + ql) * P.ldo + head * 128 + 4 * h;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint2 v = make_uint2(ca_pack2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv),
                                     ca_pack2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
          *(uint2 *)(op + 32 * db + 8 * g) = v;
        }
      if (P.out_f32) {
        float *fp = P.out_f32 + (size_t)(qrow0 + ql) * P.ldo32 + head * 128 + 4 * h;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(f32x4 *)(fp + 32 * db + 8 * g) =
                f32x4{o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv};
      }
    }
  }
}

}  // namespace

extern "C" int ca_attn_fwd_bf16(const ca_attn_problem *problems, int32_t n_problems, int32_t num_heads,
                                float scale, ca_stream_t stream) {
  if (!problems || n_problems < 1 || n_problems > 2 || num_heads < 1) {
    ca_set_error("ca_attn_fwd_bf16: n_problems=%d num_heads=%d", n_problems, num_heads);
    return CA_ERR_ARG;
  }
  AttnLaunch L = {};
  L.num_heads = num_heads;
  L.scale_log2 = scale * 1.4426950408889634f;
  const int hx = (num_heads + 7) / 8;  // heads per XCD group
  int total = 0;
  for (int i = 0; i < n_problems; ++i) {
    const ca_attn_problem &p = problems[i];
    if (!p.q || !p.out || !p.k0 || !p.v0 || p.nq < 1 || p.n0 < 1 || p.n1 < 0 || (p.n1 > 0 && (!p.k1 || !p.v1))) {
      ca_set_error("ca_attn_fwd_bf16[%d]: null pointer or empty shape (nq=%d n0=%d n1=%d)", i, p.nq, p.n0, p.n1);
      return CA_ERR_ARG;
    }
    if (p.ldq % 8 || p.ldo % 8 || p.ldkv % 8 || p.ldq < num_heads * 128 || p.ldo < num_heads * 128 ||
        p.ldkv < num_heads * 128) {
      ca_set_error("ca_attn_fwd_bf16[%d]: row strides must be >= num_heads*128 and multiples of 8", i);
      return CA_ERR_ARG;
    }
    if (p.out_f32 && (p.ldo32 % 4 || p.ldo32 < num_heads * 128)) {
      ca_set_error("ca_attn_fwd_bf16[%d]: ldo32 must be >= num_heads*128 and a multiple of 4", i);
      return CA_ERR_ARG;
    }
    if (((uintptr_t)p.q | (uintptr_t)p.out | (uintptr_t)p.k0 | (uintptr_t)p.v0 | (uintptr_t)p.k1 |
         (uintptr_t)p.v1 | (uintptr_t)p.out_f32) & 15) {
      ca_set_error("ca_attn_fwd_bf16[%d]: pointers must be 16-byte aligned", i);
      return CA_ERR_ARG;
    }
    L.p[i] = p;
    if (p.n1 == 0) {  // keep segment-1 pointers dereferenceable for the (never selected) arm
      L.p[i].k1 = p.k0;
      L.p[i].v1 = p.v0;
    }
    L.nqb[i] = (p.nq + 255) / 256;
    total += 8 * hx * L.nqb[i];
  }
  if (n_problems == 1) {
    L.p[1] = L.p[0];
    L.nqb[1] = 1;
    L.blocks_p1 = 0;
  } else {
    L.blocks_p1 = 8 * hx * L.nqb[1];
  }
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void *)ca_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       ATTN_LDS);
    if (e != hipSuccess) {
      ca_set_error("ca_attn_fwd_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return CA_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(ca_attn_kernel, dim3(total), dim3(512), ATTN_LDS, (hipStream_t)stream, L);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ca_set_error("ca_attn_fwd_bf16: launch failed: %s", hipGetErrorString(e));
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}
