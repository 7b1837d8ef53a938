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
#include <stdlib.h>

#include <type_traits>

#include <atomic>

#include "ca_common.h"
#include "ca_attn_common.h"
#ifndef CA_ATTN_KPF
#define CA_ATTN_KPF 4
#endif

namespace {
using namespace ca_attn_detail;

__device__ __forceinline__ bf16x8 pack8(const f32x16 &s, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)s[base + j];
  return r;
}

// NW = waves per workgroup: 8 (256 query rows, 1 workgroup per CU) or 4 (128 rows, 2 per CU).
// PRE = the q rows already carry softmax_scale * log2(e) (CA_ATTN_Q_PRESCALED): a tile's scores then leave the
// K Q^T chain as s - reference (the chain's first MFMA takes a register block holding -reference as its C operand),
// so a probability is ONE v_exp_f32 per score: no multiply, no subtract.
template <int NW, bool PRE = false>
__global__ __launch_bounds__(NW * 64, 2) void ca_attn_kernel(const AttnLaunch L) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which (problem, head, query block)
  int bid = blockIdx.x;
  int prob = 0;
  while (prob + 1 < L.n_problems && bid >= L.blk_end[prob]) ++prob;  // scalar: a handful of problems per launch
  if (prob) bid -= L.blk_end[prob - 1];
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
  const int qrow0 = qb * (NW * 32) + wave * 32;
  const bool active = qrow0 < nq;  // wave-uniform
  const int qrow = min(qrow0 + ql, nq - 1);

  // ---- Q fragments (B operand: lane holds Q[q = lane&31][d = 16*ks + 8*h + j])
  bf16x8 qf[8];
  {
    // query / output rows come in up to two row segments: rows [0, nq0) from q / out, the rest from q1 / out1
    const bf16 *qp = (qrow < P.nq0 ? (const bf16 *)P.q + (size_t)qrow * P.ldq
                                   : (const bf16 *)P.q1 + (size_t)(qrow - P.nq0) * P.ldq) + head * 128 + h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8 *)(qp + ks * 16);
  }

  // ---- staging by LDS-DMA (global_load_lds_dwordx4 from inline asm: no VGPR round trip, and hipcc does
  // not order the tile's ds_reads behind it).  A tile is 16 pieces of 1 KiB (4 key rows) per matrix; the
  // LDS image is lane-linear, so the XOR swizzles go on the per-lane SOURCE chunk.  The DMA of tile t+1
  // is issued at the top of iteration t (its buffer was last read before the previous barrier) and
  // retired with vmcnt(0) just before the barrier that ends iteration t.
  const int st_row = lane >> 4, st_cp = lane & 15;
  const bf16 *k0p = (const bf16 *)P.k0 + head * 128;
  const bf16 *v0p = (const bf16 *)P.v0 + head * 128;
  const bf16 *k1p = (const bf16 *)P.k1 + head * 128;
  const bf16 *v1p = (const bf16 *)P.v1 + head * 128;
  // Lane constants of the copies: piece q = wave*(16/NW)+j covers key rows 4q..4q+3 of the tile; byte offset of
  // this lane's 16 bytes inside a tile whose first row is at offset 0 (row r, swizzled chunk).
  uint32_t koff[16 / NW], voff[16 / NW];
#pragma unroll
  for (int j = 0; j < 16 / NW; ++j) {
    const int r = 4 * (wave * (16 / NW) + j) + st_row;
    koff[j] = ((uint32_t)r * (uint32_t)ldkv + ((st_cp ^ (r & 15)) << 3)) * 2u;
    voff[j] = ((uint32_t)r * (uint32_t)ldkv + ((st_cp ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 3)) * 2u;
  }
  auto stage_tile = [&](int tile, int buf) {
    char *kb = smem + buf * BUF_BYTES;
    const int lo = tile * KV_TILE;
    // fast path (wave-uniform): all 64 rows exist and lie in one segment -> scalar tile base + lane constant
    const bool in0 = lo + KV_TILE <= n0, in1 = lo >= n0 && lo + KV_TILE <= nkeys;
    if (in0 || in1) {
      const size_t ro = (size_t)(in0 ? lo : lo - n0) * ldkv;
      const bf16 *kt = (in0 ? k0p : k1p) + ro, *vt = (in0 ? v0p : v1p) + ro;
#pragma unroll
      for (int j = 0; j < 16 / NW; ++j) {
        const int q = wave * (16 / NW) + j;
        ca_glds16_asm_s(kt, koff[j], kb + q * 1024);
        ca_glds16_asm_s(vt, voff[j], kb + TILE_BYTES + q * 1024);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) {
      const int q = wave * (16 / NW) + j;
      const int r = 4 * q + st_row;
      const int kk = min(tile * KV_TILE + r, nkeys - 1);
      const bool s0 = kk < n0;
      const size_t ro = (size_t)(s0 ? kk : kk - n0) * ldkv;
      ca_glds16_asm((s0 ? k0p : k1p) + ro + ((st_cp ^ (r & 15)) << 3), kb + q * 1024);
      ca_glds16_asm((s0 ? v0p : v1p) + ro + ((st_cp ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 3),
                    kb + TILE_BYTES + q * 1024);
    }
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
  const float sl2 = PRE ? 1.0f : L.scale_log2;
  f32x16 negm;   // PRE: every register = -m_run of this lane's query row (rewritten only when the reference moves)
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;

  const int nt = (nkeys + KV_TILE - 1) / KV_TILE;
  stage_tile(0, 0);
  // vmcnt(0) through the builtin (0x0F70): it also tells hipcc's wait-count pass that the Q-fragment
  // loads have landed, so it adds no vmcnt waits for them inside the loop (they would drain the DMA)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  // One K/V tile for this wave's 32 query rows.  MASKED is the ragged last tile only: kept out of the main
  // loop's code (hipcc otherwise turns the uniform test into 32 v_cndmask per tile for every tile).
  // The LDS buffer index is a compile-time constant (tile t lives in buffer t & 1, the loop is unrolled by two),
  // so every fragment address is a loop-invariant lane register plus an instruction immediate.
  auto tile_body = [&](int t, auto cur_tag, auto masked_tag, auto first_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;  // tile 0: the reference is still unset
    constexpr int cur = decltype(cur_tag)::value;
    const char *kbuf = smem + cur * BUF_BYTES;
    const char *vbuf = kbuf + TILE_BYTES;
    // ---- S^T[key][q] = sum_d K[key][d] Q[q][d], then p = exp2(s * scale - m_run).
    // The running reference m_run of a row is NOT the running maximum: softmax is invariant to the reference,
    // and fp32 sums / bf16 P fragments keep their relative precision at any scale, so the reference only has to
    // keep 2^(s - m_run) inside the exponent range.  The first tile sets it to the tile's row maximum; after that
    // a tile is exponentiated against the stale reference WITHOUT computing its maximum (23 max + a shuffle +
    // compare per tile saved: the loop is vector-issue bound), and only if a row sum comes out above 2^30 (some
    // score more than ~30 octaves above the reference: rare after the first tile) the tile is redone the classical
    // way: S recomputed, reference raised to the new maximum, O^T and l rescaled.  Nothing of the tile has been
    // added to l or O^T at that point, so everything at the old reference is rescaled exactly once.
    f32x16 s[2];
    float rs = 0.f;
    // K Q^T of the tile; CINIT = the accumulators start from `negm` (PRE fast pass) instead of zero
    auto qk = [&](auto cinit_tag) {
      constexpr bool CINIT = decltype(cinit_tag)::value;
      asm volatile("" ::: "memory");  // the K fragments are re-read per pass (hoisted, they would pin 64 VGPRs)
      // K fragments PF MFMAs ahead of their use: an LDS read takes 2-4 MFMA slots to come back, and left to itself
      // hipcc sinks every read to its MFMA (one fragment register, lgkmcnt(0) before each MFMA) to save registers.
      // sched_barrier(0) pins the order; the wait counts are still the compiler's.
      constexpr int PF = CA_ATTN_KPF;
      bf16x8 kq[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) kq[i] = *(const bf16x8 *)(kbuf + (i >> 3) * 8192 + (k_lane ^ ((i & 7) << 5)));
      if constexpr (CINIT) {
        asm volatile("" : "+v"(negm));  // opaque: hipcc must keep the block in registers, not re-splat it per tile
        s[0] = negm;
        s[1] = negm;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[0][r] = s[1][r] = 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[i >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kq[i % PF], qf[i & 7], s[i >> 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + PF < 16) {
          kq[i % PF] = *(const bf16x8 *)(kbuf + ((i + PF) >> 3) * 8192 + (k_lane ^ (((i + PF) & 7) << 5)));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (MASKED) {  // key = 64*t + 32*kb + (r&3) + 8*(r>>2) + 4*h
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KV_TILE + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (key >= nkeys) s[kb][r] = -INFINITY;
          }
      }
    };
    bool done = false;
    if constexpr (PRE && !FIRST) {
      // fast pass: scores arrive as s - reference, a probability is one v_exp_f32
      qk(std::true_type{});
      rs = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[kb][r]);
          s[kb][r] = p;
          rs += p;
        }
      done = __builtin_amdgcn_ballot_w64(!(rs <= REDO_LIMIT)) == 0;
    }
    if (!done) {
      bool with_max = FIRST || PRE;   // (PRE: this block is the first tile or a redo, both with the maximum)
#pragma nounroll
      for (;;) {
        qk(std::false_type{});
        if (with_max) {  // wave-uniform
          // this lane: query row ql, 32 of the tile's 64 keys; lane^32 has the rest
          float mx = s[0][0];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
          mx = fmaxf(mx, __shfl_xor(mx, 32)) * sl2;
          const float m_new = fmaxf(m_run, mx);
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
          l_run *= alpha;
#pragma unroll
          for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
          if constexpr (PRE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -m_run;
          }
        }
        rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(PRE ? s[kb][r] - m_run : fmaf(s[kb][r], sl2, -m_run));
            s[kb][r] = p;
            rs += p;
          }
        // (a row whose keys are all masked so far has m_run = -1e30 and rs = 0: not > the limit; inf compares true)
        if (with_max || __builtin_amdgcn_ballot_w64(!(rs <= REDO_LIMIT)) == 0) break;
        with_max = true;
      }
    }
    l_run += rs;
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
  };

  const bool ragged = (nkeys & (KV_TILE - 1)) != 0;
  const int nt_full = ragged ? nt - 1 : nt;  // tiles the main loop handles (no key masking)
  auto iteration = [&](int t, auto cur_tag, auto first_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    if (t + 1 < nt) stage_tile(t + 1, cur ^ 1);
    if (active) tile_body(t, cur_tag, std::false_type{}, first_tag);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has landed (this wave's pieces)
    __syncthreads();
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  const std::false_type later{};
  int t = 0;
  if (nt_full > 0) {  // tile 0 sets the softmax reference of every row
    iteration(0, B0{}, std::true_type{});
    t = 1;
  }
  for (; t + 1 < nt_full; t += 2) {
    iteration(t, B1{}, later);
    iteration(t + 1, B0{}, later);
  }
  if (t < nt_full) iteration(t, B1{}, later);
  if (ragged && active) {
    if (nt == 1) tile_body(0, B0{}, std::true_type{}, std::true_type{});
    else if ((nt - 1) & 1) tile_body(nt - 1, B1{}, std::true_type{}, later);
    else tile_body(nt - 1, B0{}, std::true_type{}, later);
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l ;  d = 32*db + (r&3) + 8*(r>>2) + 4*h
  if (active) {
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    // bf16 output.  A lane holds columns 8g+4h .. 8g+4h+3 of its row (8 bytes) for g = 0..3 of every 32-column
    // block; one v_permlane32_swap per dword trades group g of the upper half-wave for group g+1 of the lower
    // one, after which every lane owns 16 contiguous bytes: 8 dwordx4 stores per lane instead of 16 dwordx2
    // (the store tail is issue-bound, not bandwidth-bound).  All lanes take part in the swaps; only the
    // store is predicated on the row being valid.
    {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const int orow = min(qrow0 + ql, nq - 1);
      bf16 *op = (orow < P.nq0 ? (bf16 *)P.out + (size_t)orow * P.ldo : (bf16 *)P.out1 + (size_t)(orow - P.nq0) * P.ldo) +
                 head * 128 + 8 * h;
      const bool row_ok = qrow0 + ql < nq;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          const uint32_t ax = ca_pack2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
          const uint32_t ay = ca_pack2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
          const uint32_t bx = ca_pack2(o[db][4 * g + 4] * inv, o[db][4 * g + 5] * inv);
          const uint32_t by = ca_pack2(o[db][4 * g + 6] * inv, o[db][4 * g + 7] * inv);
          const u32x2 sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
          const u32x2 sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
          if (row_ok) *(uint4 *)(op + 32 * db + 8 * g) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
    }
    if (qrow0 + ql < nq) {
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

static int ca_attn_fwd_impl(const ca_attn_problem *problems, int32_t n_problems, int32_t num_heads, float scale,
                            bool qk_f16, ca_stream_t stream);

extern "C" int ca_attn_fwd_bf16(const ca_attn_problem *problems, int32_t n_problems, int32_t num_heads,
                                float scale, ca_stream_t stream) {
  return ca_attn_fwd_impl(problems, n_problems, num_heads, scale, false, stream);
}

extern "C" int ca_attn_fwd_qk16(const ca_attn_problem *problems, int32_t n_problems, int32_t num_heads,
                                ca_stream_t stream) {
  return ca_attn_fwd_impl(problems, n_problems, num_heads, CA_ATTN_Q_PRESCALED, true, stream);
}

static int ca_attn_fwd_impl(const ca_attn_problem *problems, int32_t n_problems, int32_t num_heads, float scale,
                            bool qk_f16, ca_stream_t stream) {
  if (!problems || n_problems < 1 || n_problems > CA_ATTN_MAX_PROBLEMS || num_heads < 1 || !(scale >= 0.0f)) {
    ca_set_error("ca_attn_fwd_bf16: n_problems=%d (max %d) num_heads=%d scale=%g", n_problems, CA_ATTN_MAX_PROBLEMS,
                 num_heads, (double)scale);
    return CA_ERR_ARG;
  }
  const bool pre = scale == CA_ATTN_Q_PRESCALED;
  // 8-wave workgroups (256 query rows, one per CU), K/V tiles by LDS-DMA: ~250 us for 4352x4352x24 heads on MI355X.
  // (The 128-row, 4-wave form of rounds 1-4 -- slower, and its pre-scaled instantiation spilled -- is gone.)
  constexpr int nw = 8;
  // pre-scaled q (the model path): the one-wave-per-SIMD kernel (ca_attn4.hip; 4 waves x 64 rows = 256 rows per workgroup
  // as well, same numerics contract).  Every call that passes a scale runs on the two-waves-per-SIMD kernel below
  // (diagnostic builds: CA_ATTN_KERNEL=8 sends pre-scaled q there too).
  static const bool want4 = ca_ab_env("CA_ATTN_KERNEL", 4) != 8;
  const bool use4 = (pre && want4) || qk_f16;   // (half-precision q / k exist in the one-wave-per-SIMD kernel only)
  const int qrows = use4 ? 256 : nw * 32;
  AttnLaunch L = {};
  L.num_heads = num_heads;
  L.n_problems = n_problems;
  L.scale_log2 = scale * 1.4426950408889634f;
  static const bool no_reref = ca_ab_env("CA_ATTN_REREF", 1) == 0;      // (diagnostic builds: tools/attn_peaky.py)
  static const bool limit60 = ca_ab_env("CA_ATTN_LIMIT60", 0) == 1;
  L.flags = (no_reref ? 1 : 0) | (limit60 ? 2 : 0);
  const int hx = (num_heads + 7) / 8;  // heads per XCD group
  int total = 0;
  for (int i = 0; i < n_problems; ++i) {
    const ca_attn_problem &p = problems[i];
    if (!p.q || !p.out || !p.k0 || !p.v0 || p.nq < 1 || p.n0 < 1 || p.n1 < 0 || (p.n1 > 0 && (!p.k1 || !p.v1))) {
      ca_set_error("ca_attn_fwd_bf16[%d]: null pointer or empty shape (nq=%d n0=%d n1=%d)", i, p.nq, p.n0, p.n1);
      return CA_ERR_ARG;
    }
    if (p.nq0 < 0 || p.nq0 > p.nq || (p.nq0 > 0 && p.nq0 < p.nq && (!p.q1 || !p.out1))) {
      ca_set_error("ca_attn_fwd_bf16[%d]: two query segments need 0 < nq0 < nq, q1 and out1", i);
      return CA_ERR_ARG;
    }
    if (p.ldq % 8 || p.ldo % 8 || p.ldkv % 8 || p.ldq < num_heads * 128 || p.ldo < num_heads * 128 ||
        p.ldkv < num_heads * 128) {
      ca_set_error("ca_attn_fwd_bf16[%d]: row strides must be >= num_heads*128 and multiples of 8", i);
      return CA_ERR_ARG;
    }
    if (p.hm_con || p.hm_part) {
      if (!p.hm_con || !p.hm_part || p.hm_C < 1 || p.hm_C > 8 || p.ldhc % 4 || p.ldhc < num_heads * 128 ||
          (((uintptr_t)p.hm_con | (uintptr_t)p.hm_part) & 15) || !use4 || p.nq0 <= 0 || p.nq0 >= p.nq) {
        ca_set_error("ca_attn_fwd_bf16[%d]: hm_con / hm_part need each other, 1 <= hm_C <= 8, ldhc >= num_heads*128 (%% 4), "
                     "16-byte alignment, two query segments and the pre-scaled-q kernel", i);
        return CA_ERR_ARG;
      }
    }
    if (p.out_f32 && (p.ldo32 % 4 || p.ldo32 < num_heads * 128)) {
      ca_set_error("ca_attn_fwd_bf16[%d]: ldo32 must be >= num_heads*128 and a multiple of 4", i);
      return CA_ERR_ARG;
    }
    if (((uintptr_t)p.q | (uintptr_t)p.out | (uintptr_t)p.k0 | (uintptr_t)p.v0 | (uintptr_t)p.k1 |
         (uintptr_t)p.v1 | (uintptr_t)p.out_f32 | (uintptr_t)p.q1 | (uintptr_t)p.out1) & 15) {
      ca_set_error("ca_attn_fwd_bf16[%d]: pointers must be 16-byte aligned", i);
      return CA_ERR_ARG;
    }
    L.p[i] = p;
    if (p.n1 == 0) {  // keep segment-1 pointers dereferenceable for the (never selected) arm
      L.p[i].k1 = p.k0;
      L.p[i].v1 = p.v0;
    }
    if (p.nq0 == 0 || p.nq0 == p.nq) {  // one query segment (nq0 = 0 is the pre-batch spelling of "all rows in q")
      L.p[i].nq0 = p.nq;
      L.p[i].q1 = p.q;
      L.p[i].out1 = p.out;
    }
    L.nqb[i] = (p.nq + qrows - 1) / qrows;
    total += 8 * hx * L.nqb[i];
    L.blk_end[i] = total;
  }
  if (use4) {
    const int rc = ca_attn4_launch(L, total, qk_f16, (hipStream_t)stream);
    if (rc != CA_OK) return rc;
    const hipError_t e4 = hipGetLastError();
    if (e4 != hipSuccess) {
      ca_set_error("ca_attn_fwd_bf16: launch failed: %s", hipGetErrorString(e4));
      return CA_ERR_LAUNCH;
    }
    return CA_OK;
  }
  static std::atomic<unsigned long long> attr_done{0};  // one bit per device: the attribute is per device
  const unsigned long long dev_bit = ca_device_bit();
  if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
    hipError_t e = hipSuccess;
    for (const void *fn : {(const void *)ca_attn_kernel<8>, (const void *)ca_attn_kernel<8, true>})
      if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS);
    if (e != hipSuccess) {
      ca_set_error("ca_attn_fwd_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return CA_ERR_LAUNCH;
    }
    attr_done.fetch_or(dev_bit, std::memory_order_release);  // idempotent: a race only repeats the call
  }
  if (pre)
    hipLaunchKernelGGL((ca_attn_kernel<8, true>), dim3(total), dim3(512), ATTN_LDS, (hipStream_t)stream, L);
  else
    hipLaunchKernelGGL((ca_attn_kernel<8, false>), dim3(total), dim3(512), ATTN_LDS, (hipStream_t)stream, L);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ca_set_error("ca_attn_fwd_bf16: launch failed: %s", hipGetErrorString(e));
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}

extern "C" int ca_attn_stats(unsigned long long *counters, int32_t reset) {
  if (!counters) {
    ca_set_error("ca_attn_stats: null pointer");
    return CA_ERR_ARG;
  }
  return ca_attn4_read_counters(counters, reset);
}
