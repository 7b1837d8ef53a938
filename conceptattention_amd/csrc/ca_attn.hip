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

#include "ca_common.h"

namespace {

struct AttnLaunch {
  ca_attn_problem p[2];
  int32_t nqb[2];  // 256-row query blocks per head
  int32_t num_heads;
  int32_t blocks_p1;  // workgroups of problem 1 (they come first in the grid)
  float scale_log2;   // softmax scale * log2(e)
};

#ifdef CA_ATTN_STAMP
// Diagnostic build only (tools/stamp_attn.sh): per-phase cycle sums of waves 0 and 4 of workgroup 0.
__device__ unsigned long long ca_attn_dbg[32];
#define CA_STAMP(VAR)                                                              \
  __builtin_amdgcn_sched_barrier(0);                                               \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(VAR)::"memory");       \
  __builtin_amdgcn_sched_barrier(0)
#define CA_ACC(SLOT, T0, T1) dbg_acc[SLOT] += (T1) - (T0)
#else
#define CA_STAMP(VAR)
#define CA_ACC(SLOT, T0, T1)
#endif

// priority policy of the ping-pong kernel: 0 none, 1 raise around every matrix phase, 2 static for group 1
#ifndef CA_ATTN_PRIO_MODE
#define CA_ATTN_PRIO_MODE 1
#endif
#if CA_ATTN_PRIO_MODE == 1
#define CA_PRIO_HI() __builtin_amdgcn_s_setprio(1)
#define CA_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#elif CA_ATTN_PRIO_MODE == 3  // matrix phase always outranks the partner's vector phase: g0 1/0, g1 2/1
#define CA_PRIO_HI() if (grp) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1)
#define CA_PRIO_LO() if (grp) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0)
#elif CA_ATTN_PRIO_MODE == 4  // vector phase outranks the matrix phase
#define CA_PRIO_HI() __builtin_amdgcn_s_setprio(0)
#define CA_PRIO_LO() __builtin_amdgcn_s_setprio(2)
#else
#define CA_PRIO_HI()
#define CA_PRIO_LO()
#endif

constexpr int KV_TILE = 64;
constexpr float RESCALE_LOG2 = 8.0f;  // deferred-rescale threshold of the online softmax, in log2 units
constexpr int TILE_BYTES = KV_TILE * 256;  // one K or V tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;
constexpr int ATTN_LDS = 2 * BUF_BYTES;
#ifndef CA_PP_WINDOW
#define CA_PP_WINDOW 6
#endif
constexpr int PP_WINDOW = CA_PP_WINDOW;             // LDS operand prefetch distance (MFMA steps)
constexpr int PP_RING = 3;                          // K and V ring depth of the ping-pong kernel
constexpr int ATTN_PP_LDS = 2 * PP_RING * TILE_BYTES;

__device__ __forceinline__ bf16x8 pack8(const f32x16 &s, int base) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)s[base + j];
  return r;
}

template <int NW>  // waves per workgroup: 8 (256 query rows, 1 workgroup per CU) or 4 (128 rows, 2 per CU)
__global__ __launch_bounds__(NW * 64, 2) void ca_attn_kernel(const AttnLaunch L) {
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
  const int qrow0 = qb * (NW * 32) + wave * 32;
  const bool active = qrow0 < nq;  // wave-uniform
  const int qrow = min(qrow0 + ql, nq - 1);

  // ---- Q fragments (B operand: lane holds Q[q = lane&31][d = 16*ks + 8*h + j])
  bf16x8 qf[8];
  {
    const bf16 *qp = (const bf16 *)P.q + (size_t)qrow * P.ldq + head * 128 + h * 8;
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
  const float sl2 = L.scale_log2;

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
  auto tile_body = [&](int t, auto cur_tag, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr int cur = decltype(cur_tag)::value;
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
    if constexpr (MASKED) {  // key = 64*t + 32*kb + (r&3) + 8*(r>>2) + 4*h
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
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * sl2;
    // The reference maximum of a row is only raised (and O^T, l rescaled) when some row of the wave exceeds
    // it by more than RESCALE_LOG2: until then p = exp2(s - m_run) is at most 2^RESCALE_LOG2 instead of 1,
    // which fp32 sums and bf16 P fragments carry without loss, and O/l is unchanged.  After the first
    // tiles this is rare: no 64 accumulator multiplies per tile.
    if (__builtin_amdgcn_ballot_w64(mx > m_run + RESCALE_LOG2) != 0) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sl2, -m_run));
        s[kb][r] = p;
        rs += p;
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
  auto iteration = [&](int t, auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    if (t + 1 < nt) stage_tile(t + 1, cur ^ 1);
    if (active) tile_body(t, cur_tag, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has landed (this wave's pieces)
    __syncthreads();
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  int t = 0;
  for (; t + 1 < nt_full; t += 2) {
    iteration(t, B0{});
    iteration(t + 1, B1{});
  }
  if (t < nt_full) iteration(t, B0{});
  if (ragged && active) {
    if ((nt - 1) & 1) tile_body(nt - 1, B1{}, std::true_type{});
    else tile_body(nt - 1, B0{}, std::true_type{});
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
      bf16 *op = (bf16 *)P.out + (size_t)min(qrow0 + ql, nq - 1) * P.ldo + head * 128 + 8 * h;
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

// =============================================================================================
// Ping-pong variant (default).  Same tiles, fragments and LDS images as above, but the 8 waves run
// as two groups of 4 (one wave of each group per SIMD) that are kept one barrier apart:
//     phase X(t): O^T += V(t-1)^T P(t-1)^T ; S^T = K(t) Q^T          (32 MFMAs + LDS reads)
//     phase Y(t): online softmax of S^T -> P(t), rescale O^T          (VALU) + LDS staging
// so on every SIMD one wave is in its matrix phase while the other one is in its vector phase
// (the matrix and vector pipes run concurrently only across waves).  Group 0's threads stage every
// K tile, group 1's every V tile: in Y(t) a group writes tile t+1 of its matrix (loaded one tile
// earlier into registers) and issues the loads of tile t+2.  With the one-barrier lag this gives
//     K(t+1): written in interval 2t+1, read in 2t+2 / 2t+3, its buffer last read in 2t-1
//     V(t+1): written in interval 2t+2, read in 2t+4 / 2t+5, its buffer last read in 2t+1
// (2-deep rings, one s_barrier per phase, writers wait lgkmcnt(0) before the barrier).
__global__ __launch_bounds__(512, 2) void ca_attn_pp_kernel(const AttnLaunch L) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;  // 0: stages K, 1: stages V

  int bid = blockIdx.x;
  const int prob = bid < L.blocks_p1 ? 1 : 0;
  if (!prob) bid -= L.blocks_p1;
  const int nqb = L.nqb[prob];
  const int xg = bid & 7, idx = bid >> 3;
  const int head = xg + 8 * (idx / nqb);
  const int qb = idx % nqb;
  if (head >= L.num_heads) return;
  const ca_attn_problem &P = L.p[prob];
  const int nq = P.nq, n0 = P.n0, nkeys = P.n0 + P.n1;
  const int ldkv = P.ldkv;

  const int h = lane >> 5;
  const int ql = lane & 31;
  const int qrow0 = qb * 256 + wave * 32;
  const bool active = qrow0 < nq;
  const int qrow = min(qrow0 + ql, nq - 1);

  bf16x8 qf[8];
  {
    const bf16 *qp = (const bf16 *)P.q + (size_t)qrow * P.ldq + head * 128 + h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8 *)(qp + ks * 16);
  }

  // ---- staging by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip): a group's 4 waves move
  // one 64x128 tile of "their" matrix per K/V tile, 4 wave-instructions of 1 KiB (= 4 key rows) each.
  // The LDS image is lane-linear, so the XOR swizzle goes on the per-lane SOURCE chunk.
  const int gw = wave & 3;
  const int st_row = lane >> 4, st_cp = lane & 15;  // row within the 4-row piece, LDS chunk
  const bf16 *src0 = (const bf16 *)(grp ? P.v0 : P.k0) + head * 128;
  const bf16 *src1 = (const bf16 *)(grp ? P.v1 : P.k1) + head * 128;
  auto stage_tile = [&](int tile) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = gw * 4 + j;              // 1-KiB piece of the tile
      const int r = 4 * q + st_row;          // key row inside the tile
      const int swz = grp ? (((r & 3) << 2) | ((r >> 2) & 3)) : (r & 15);
      const int kk = min(tile * KV_TILE + r, nkeys - 1);
      const bool s0 = kk < n0;
      const bf16 *src = (s0 ? src0 : src1) + (size_t)(s0 ? kk : kk - n0) * ldkv + ((st_cp ^ swz) << 3);
      ca_glds16_asm(src, smem + (grp ? PP_RING * TILE_BYTES : 0) + (tile % PP_RING) * TILE_BYTES + q * 1024);
    }
  };
#define CA_ATTN_DMA_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  const int k_lane = ql * 256 + (((h ^ (ql & 15)) & 15) << 4);
  const int qq = (lane & 15) >> 2;
  const int c_lane = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
  int v_lane[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int x = (qq << 2) | ((2 * jj + h) & 3);
    v_lane[jj] = (4 * h + qq) * 256 + (((c_lane ^ x) & 15) << 4) + 8 * (lane & 1);
  }

  f32x16 o[4], s[2];
  bf16x8 pf[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sl2 = L.scale_log2;
  const int nt = (nkeys + KV_TILE - 1) / KV_TILE;

  // ---- matrix phase as ONE operand stream: step m < 16 is the PV product (kb = m>>3, sk = (m>>2)&1,
  // d-block m&3, operand = V^T fragment by two transposed reads), step m >= 16 is the QK^T product
  // (kb = (m-16)>>3, ks = (m-16)&7, operand = K fragment).  The fragment of step m+PP_WINDOW is requested
  // right after the MFMA of step m is issued, so the matrix pipe does not wait on LDS latency; the
  // first PP_WINDOW V fragments are requested at the end of the preceding vector phase.
  constexpr int W = PP_WINDOW;  // operand fragments requested ahead of the MFMA that consumes them
  bf16x8 fr[W];
  auto frag = [&](int m, int tile_v, int tile_k) -> bf16x8 {
    if (m < 16) {
      const int kb = m >> 3, sk = (m >> 2) & 1, db = m & 3;
      const char *vrow = smem + PP_RING * TILE_BYTES + (tile_v % PP_RING) * TILE_BYTES + (32 * kb + 16 * sk) * 256;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4 *)(vrow + (v_lane[0] ^ (db << 6))));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4 *)(vrow + 8 * 256 + (v_lane[1] ^ (db << 6))));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
      const int kb = (m - 16) >> 3, ks = (m - 16) & 7;
      return *(const bf16x8 *)(smem + (tile_k % PP_RING) * TILE_BYTES + kb * 8192 + (k_lane ^ (ks << 5)));
    }
  };
#ifdef CA_ATTN_NO_LDS_READS  // timing experiment only: operands stay whatever the registers hold
#define CA_FRAG(M, TV, TK) fr[(M) % W]
#else
#define CA_FRAG(M, TV, TK) frag((M), (TV), (TK))
#endif
#define CA_ATTN_STREAM(M_BEGIN, M_END, TILE_V, TILE_K)                                              \
  _Pragma("unroll") for (int m = (M_BEGIN); m < (M_END); ++m) {                                      \
    const bf16x8 cur = fr[m % W];                                                                    \
    if (m < 16) {                                                                                    \
      o[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, pf[m >> 2], o[m & 3], 0, 0, 0);        \
    } else {                                                                                         \
      const int kb_ = (m - 16) >> 3, ks_ = (m - 16) & 7;                                             \
      if (ks_ == 0)                                                                                  \
        s[kb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                            \
            cur, qf[0], (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
      else                                                                                           \
        s[kb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, qf[ks_], s[kb_], 0, 0, 0);             \
    }                                                                                                \
    if (m + W < (M_END)) fr[m % W] = CA_FRAG(m + W, (TILE_V), (TILE_K));                                \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  }
  auto softmax = [&](int t) {
#ifdef CA_ATTN_NO_SOFTMAX  // timing experiment only: keep the data flow, drop the vector work
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) pf[2 * kb + sk] = pack8(s[kb], 8 * sk);
    l_run += s[0][0];
    return;
#endif
    if (t == nt - 1 && (nkeys & (KV_TILE - 1))) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KV_TILE + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (key >= nkeys) s[kb][r] = -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * sl2;
    // rescale only when some row's running maximum grows (exact: alpha == 1 otherwise)
    if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sl2, -m_run));
        s[kb][r] = p;
        rs += p;
      }
    l_run += rs;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) pf[2 * kb + sk] = pack8(s[kb], 8 * sk);
  };
#define CA_ATTN_SYNC()                     \
  __builtin_amdgcn_sched_barrier(0);       \
  __builtin_amdgcn_s_barrier();            \
  __builtin_amdgcn_sched_barrier(0)
#define CA_ATTN_LDS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

  // ---- prologue: K(0) (group 0) and V(0) (group 1) land before the first barrier
  stage_tile(0);
  // vmcnt(0) through the BUILTIN (0x0F70 = vmcnt 0, expcnt/lgkmcnt untouched): besides retiring the DMA it
  // tells hipcc's wait-count pass that the Q-fragment loads above have landed, so it inserts no
  // vmcnt waits for them inside the tile loop (where they would drain the hand-counted DMA).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  CA_ATTN_SYNC();
  if (grp == 1) { CA_ATTN_SYNC(); }  // stagger: group 1 runs one barrier behind group 0
#if CA_ATTN_PRIO_MODE == 2
  if (grp == 1) __builtin_amdgcn_s_setprio(1);  // static: the younger half wins arbitration throughout
#endif

  // Staging schedule, identical for both groups on their own matrix (3-deep rings): tile t+1 is
  // DMA-issued at the start of the group's X(t) and retired (vmcnt 0) at the end of its Y(t).
  // In intervals between barriers (group 0: X(t) = 2t, Y(t) = 2t+1; group 1 one later):
  //   K(t+1): in flight 2t .. 2t+1, first read 2t+2;   its buffer held K(t-2), last read 2t-3
  //   V(t+1): in flight 2t+1 .. 2t+2, first read at the end of 2t+3 (fragment prefetch of group 0);
  //           its buffer held V(t-2), last read (PV(t-2) in X(t-1)) in interval 2t-1
  if (nt > 1) stage_tile(1);
  if (active) {  // X(0): QK(0) only
#pragma unroll
    for (int m = 16; m < 16 + W; ++m) fr[m % W] = frag(m, 0, 0);
    CA_PRIO_HI();
    CA_ATTN_STREAM(16, 32, 0, 0)
    CA_PRIO_LO();
  }
  CA_ATTN_SYNC();
#ifdef CA_ATTN_STAMP
  unsigned long long dbg_acc[6] = {0, 0, 0, 0, 0, 0}, ts0, ts1, ts2, ts3, ts4, ts5;
#endif
  for (int t = 0; t < nt - 1; ++t) {  // every tile that has a successor
    // ---- Y(t): vector phase; request the first V(t) fragments of the next matrix phase
    CA_STAMP(ts0);
    if (active) {
      softmax(t);
#pragma unroll
      for (int m = 0; m < W; ++m) fr[m] = frag(m, t, 0);
    }
    CA_STAMP(ts1);
    CA_ATTN_DMA_DONE();
    CA_ATTN_LDS_DONE();
    CA_STAMP(ts2);
    CA_ATTN_SYNC();
    CA_STAMP(ts3);
    // ---- X(t+1): matrix phase  PV(t) + QK(t+1)
    if (t + 2 < nt) stage_tile(t + 2);
    if (active) {
      CA_PRIO_HI();
      CA_ATTN_STREAM(0, 32, t, t + 1)
      CA_PRIO_LO();
    }
    CA_STAMP(ts4);
    CA_ATTN_SYNC();
    CA_STAMP(ts5);
    CA_ACC(0, ts0, ts1);  // Y compute
    CA_ACC(1, ts1, ts2);  // Y retire waits
    CA_ACC(2, ts2, ts3);  // barrier after Y
    CA_ACC(3, ts3, ts4);  // X stream
    CA_ACC(4, ts4, ts5);  // barrier after X
  }
#ifdef CA_ATTN_STAMP
  if (blockIdx.x == 200 && lane == 0 && (wave == 0 || wave == 4)) {
    for (int i = 0; i < 5; ++i) ca_attn_dbg[(wave ? 8 : 0) + i] = dbg_acc[i];
    ca_attn_dbg[(wave ? 8 : 0) + 5] = nt - 1;
  }
#endif
  {  // last tile (kept out of the loop so the accumulators never merge across two code paths)
    const int t = nt - 1;
    if (active) {
      softmax(t);
#pragma unroll
      for (int m = 0; m < W; ++m) fr[m] = frag(m, t, 0);
    }
    CA_ATTN_LDS_DONE();
    CA_ATTN_SYNC();
    if (active) {
      CA_PRIO_HI();
      CA_ATTN_STREAM(0, 16, t, t)
      CA_PRIO_LO();
    }
    CA_ATTN_SYNC();
  }
  if (grp == 0) { CA_ATTN_SYNC(); }
  CA_ATTN_DMA_DONE();

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
  // Default: 8-wave workgroups (256 query rows, one per CU), K/V tiles by LDS-DMA: 295 us for
  // 4352x4352x24 heads on MI355X.  A/B aids (same numerics): CA_ATTN_WAVES=4 = two independent 128-row
  // workgroups per CU (321 us), CA_ATTN_PP=1 = the two-group ping-pong schedule below (350 us).
  // What bounds all three (tools/micro/coissue.hip, tools/stamp_attn.py): per 64-key tile a wave issues
  // 32 MFMAs (1024 cycles) and ~180 vector instructions of softmax (~850 cycles), and a vector wave next
  // to a saturated MFMA wave on the same SIMD runs at only ~47 % of its solo speed.
  static const bool use_pp = getenv("CA_ATTN_PP") != nullptr;
  static const int nw = (getenv("CA_ATTN_WAVES") && atoi(getenv("CA_ATTN_WAVES")) == 4) ? 4 : 8;
  const int qrows = use_pp ? 256 : nw * 32;
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
    L.nqb[i] = (p.nq + qrows - 1) / qrows;
    total += 8 * hx * L.nqb[i];
  }
  if (n_problems == 1) {
    L.p[1] = L.p[0];
    L.nqb[1] = 1;
    L.blocks_p1 = 0;
  } else {
    L.blocks_p1 = 8 * hx * L.nqb[1];
  }
  static unsigned long long attr_done = 0;  // one bit per device: the attribute is per device
  const unsigned long long dev_bit = ca_device_bit();
  if (!(attr_done & dev_bit)) {
    hipError_t e = hipFuncSetAttribute((const void *)ca_attn_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       ATTN_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)ca_attn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)ca_attn_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_PP_LDS);
    if (e != hipSuccess) {
      ca_set_error("ca_attn_fwd_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return CA_ERR_LAUNCH;
    }
    attr_done |= dev_bit;  // idempotent; a race only repeats the call
  }
  if (use_pp)
    hipLaunchKernelGGL(ca_attn_pp_kernel, dim3(total), dim3(512), ATTN_PP_LDS, (hipStream_t)stream, L);
  else if (nw == 8)
    hipLaunchKernelGGL(ca_attn_kernel<8>, dim3(total), dim3(512), ATTN_LDS, (hipStream_t)stream, L);
  else
    hipLaunchKernelGGL(ca_attn_kernel<4>, dim3(total), dim3(256), ATTN_LDS, (hipStream_t)stream, L);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ca_set_error("ca_attn_fwd_bf16: launch failed: %s", hipGetErrorString(e));
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}

#ifdef CA_ATTN_STAMP
extern "C" int ca_debug_read_attn(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ca_attn_dbg), sizeof(unsigned long long) * 32);
}
#endif
