// Grouped bf16 GEMM with fused epilogues for gfx950 (MI355X).
//
//   out[m,n] = epi( sum_k A[m,k] * W[n,k] + bias[n] )        A:[M,K]  W:[N,K] (nn.Linear layout)
//
// Design (MI355X-first, see DESIGN.md "GEMM"):
//  * one workgroup = 8 waves (2 along M x 4 along N), 1 workgroup per CU, tile BM x BN x 64 with
//    BM = 256, BN = 64*N_REP; fp32 accumulators live in registers (v_mfma_f32_16x16x32_bf16).
//  * operands are staged HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip) into a
//    2-deep ring; the LDS image is lane-linear, so the bank-conflict swizzle
//    (16-byte chunk ^= (row>>1)&7 on 128-byte rows) is applied to the per-lane SOURCE address and
//    again on the ds_read_b128 side.
//  * the MFMA is issued "transposed" (W fragment as the A operand, activation fragment as B) and
//    the W rows of a wave are staged in a permuted order, so that every lane ends up holding
//    4*N_REP CONTIGUOUS output columns of one row: the epilogue (bias, GELU, gate*x+residual)
//    runs in registers and stores 16-byte pieces, 128 contiguous bytes per 4 lanes.
//  * several problems share one launch (image stream + text/concept stream of a double block),
//    tile ids are remapped so that the 32 workgroups that share an XCD (same blockIdx % 8) work
//    on a compact patch of tiles and reuse each other's panels in that XCD's L2 (ping-pong kernel:
//    group_m row tiles per group, the XCDs sharing each round's 256 consecutive tiles -- pick_group_m).
#include <stdlib.h>

#include <type_traits>

#include <atomic>

#include "ca_common.h"

namespace {

struct GemmLaunch {
  ca_gemm_problem p[CA_GEMM_MAX_PROBLEMS];
  int32_t ntiles[CA_GEMM_MAX_PROBLEMS];
  int32_t mt[CA_GEMM_MAX_PROBLEMS];
  int32_t nt[CA_GEMM_MAX_PROBLEMS];
  int32_t persist_tiles;  // ping-pong kernel: 0 = one workgroup per tile; else total tiles, walked by a CU-sized grid
  int32_t persist_tiles_grid;  // host only: workgroups of the persistent grid (= CUs, a multiple of 8)
  int32_t xcd_interleave; // ping-pong kernel: the XCDs share each full round's 256 consecutive tiles (tile order below)
  int32_t group_m;        // ping-pong kernel: row tiles per group of the tile order (the 32 concurrent tiles of an XCD are
                          // a group_m x 32/group_m patch); chosen per launch shape by the host (pick_group_m)
  // ping-pong kernel, tile order: the tiles of a problem's LAST row tile go to the end of the walk when that row tile
  // is thin (few valid rows), all other ("main") tiles keep the XCD-patch order among themselves
  int32_t main_total;                        // main tiles of all problems
  int32_t ntiles_main[CA_GEMM_MAX_PROBLEMS]; // main tiles per problem = mt_main * nt
  int32_t mt_main[CA_GEMM_MAX_PROBLEMS];     // row tiles in the main order (mt, or mt - 1 with a thin last row tile)
  int32_t nthin[CA_GEMM_MAX_PROBLEMS];       // thin tiles per problem (nt or 0)
  // thin-row kernel (a thin last row tile under the 256x256 bf16 ping-pong tile): 32 x 128 tiles
  int32_t thin_row0[CA_GEMM_MAX_PROBLEMS];   // first row of the problem's thin part
  int32_t thin_nt[CA_GEMM_MAX_PROBLEMS];     // its 128-column tiles (N / 128), 0 = none
};

// A last row tile with at most this many valid rows is "thin": its MFMAs on row fragments past M are skipped (the
// K loop is then paced by the staging and the barriers, about half a tile time), and it is walked last so that it
// is the thin tiles that spill into a partial last round.  Why: the [concept | text] stream of a 5-item double
// block has 1300 rows = 5 row tiles + 20 rows; those 20 rows cost every launch of the block one more round of
// full-price tiles (proj and mlp.2: 1032 tiles = 4.03 rounds -> 5), 9 % of the block (tools/remainder_probe.py).
#ifndef CA_GEMM_THIN_ROWS
#define CA_GEMM_THIN_ROWS 128
#endif

template <int M_REP, int N_REP>
struct Cfg {
  static constexpr int BM = 2 * 16 * M_REP;
  static constexpr int BN = 4 * 16 * N_REP;
  static constexpr int BK = 64;
  static constexpr int ROW_BYTES = BK * 2;
  static constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static constexpr int NLOAD = (BM + BN) / 64;  // global_load_lds per wave per K tile
  static_assert(BM % 64 == 0, "A/W boundary must be wave-instruction aligned");
};

#ifndef CA_GEMM_TWO_PHASE
#define CA_GEMM_TWO_PHASE 1
#endif

constexpr int GROUP_M = 8;

// Gate vector of output row m (GATE_RESIDUAL).  Rows < gate_rows use `gate`, the others `gate2` (or `gate` when
// gate2 is NULL).  With gate_stride != 0 each of the two row ranges is a sequence of work items of
// gate_item_rows / gate2_item_rows rows, and item i of a range uses its base vector + i * gate_stride floats
// (batched forward: the items of a batch have their own adaLN gates).
__device__ __forceinline__ const float *ca_gate_of(const ca_gemm_problem &P, int m) {
  const bool first = m < P.gate_rows;
  const float *g = (first || !P.gate2) ? P.gate : P.gate2;
  if (P.gate_stride) {
    const int r = first ? m : m - P.gate_rows;
    g += (size_t)(r / (first ? P.gate_item_rows : P.gate2_item_rows)) * P.gate_stride;
  }
  return g;
}

template <int M_REP, int N_REP>
__global__ __launch_bounds__(512, 2) void ca_gemm_kernel(const GemmLaunch L) {
  using C = Cfg<M_REP, N_REP>;
  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- tile id: XCD-contiguous remap (bijective for any grid size), then grouped-M order
  const int nblk = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
  int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int prob = (lid >= L.ntiles[0]) ? 1 : 0;
  if (prob) lid -= L.ntiles[0];
  const ca_gemm_problem &P = L.p[prob];
  const int MT = L.mt[prob], NT = L.nt[prob];
  const int grp = lid / (GROUP_M * NT);
  const int first_m = grp * GROUP_M;
  const int gm = min(GROUP_M, MT - first_m);
  const int in_grp = lid - grp * GROUP_M * NT;
  const int m0 = (first_m + in_grp % gm) * C::BM;
  const int n0 = (in_grp / gm) * C::BN;

  const int M = P.M, K = P.K;
  const char *Ab = (const char *)P.A;
  const char *Wb = (const char *)P.W;

  // ---- per-lane source byte offsets of the staging loads (k = 0)
  uint32_t src_off[C::NLOAD];
#pragma unroll
  for (int i = 0; i < C::NLOAD; ++i) {
    const int rr = i * 64 + wave * 8 + (lane >> 3);
    const int cp = lane & 7;
    if (i * 64 < C::BM) {
      const int rho = rr;
      const int c = cp ^ ((rho >> 1) & 7);
      const int row = min(m0 + rho, M - 1);
      src_off[i] = ((uint32_t)row * (uint32_t)P.lda + (uint32_t)c * 8u) * 2u;
    } else {
      const int rho = rr - C::BM;
      const int c = cp ^ ((rho >> 1) & 7);
      const int wloc = rho % (16 * N_REP);
      const int wbase = rho - wloc;
      const int j = wloc >> 4, a = (wloc >> 2) & 3, b = wloc & 3;
      const int n = n0 + wbase + 4 * N_REP * a + 4 * j + b;
      src_off[i] = ((uint32_t)n * (uint32_t)P.ldw + (uint32_t)c * 8u) * 2u;
    }
  }

  auto stage = [&](int buf, int kt) {
    const uint32_t kb = (uint32_t)kt * (C::BK * 2);
    char *base = smem + buf * C::STAGE_BYTES + wave * 8 * C::ROW_BYTES;
#pragma unroll
    for (int i = 0; i < C::NLOAD; ++i) {
      const char *src = ((i * 64 < C::BM) ? Ab : Wb) + src_off[i] + kb;
      ca_glds16(src, base + i * 64 * C::ROW_BYTES);
    }
  };

  f32x4 acc[M_REP][N_REP];
#pragma unroll
  for (int i = 0; i < M_REP; ++i)
#pragma unroll
    for (int j = 0; j < N_REP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row (lane&15), 16-byte chunk ((lane>>4) ^ row>>1), k-step s flips bit 6
  const int lane_off = (lane & 15) * C::ROW_BYTES + ((((lane >> 4) ^ ((lane & 15) >> 1)) & 7) << 4);
  const int a_off = wm * 16 * M_REP * C::ROW_BYTES + lane_off;
  const int w_off = (C::BM + wn * 16 * N_REP) * C::ROW_BYTES + lane_off;

  auto compute = [&](int buf) {
    const char *sb = smem + buf * C::STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 wf[N_REP], af[M_REP];
#pragma unroll
      for (int j = 0; j < N_REP; ++j)
        wf[j] = *(const bf16x8 *)(sb + ((w_off + j * 16 * C::ROW_BYTES) ^ (s * 64)));
#pragma unroll
      for (int i = 0; i < M_REP; ++i)
        af[i] = *(const bf16x8 *)(sb + ((a_off + i * 16 * C::ROW_BYTES) ^ (s * 64)));
#pragma unroll
      for (int i = 0; i < M_REP; ++i)
#pragma unroll
        for (int j = 0; j < N_REP; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop: 2-deep LDS ring, next tile's global_load_lds in flight under the MFMAs
  const int nk = K / C::BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk - 1; ++kt) {
    stage(cur ^ 1, kt + 1);
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  compute(cur);

  // ---- epilogue.  acc[i][j][r] = C[m][n]:  m = m0 + wm*16*M_REP + 16*i + (lane&15)
  //                                          n = n0 + wn*16*N_REP + 4*N_REP*(lane>>4) + 4*j + r
  const int g = lane >> 4;
  const int nb = n0 + wn * 16 * N_REP + 4 * N_REP * g;
  int epi = P.epilogue;
  char *outb = (char *)P.out;
  int ldo = P.ldc;
  int ncol = nb;
  if (epi == CA_EPI_SPLIT_GELU) {
    if (n0 >= P.n_split) {
      epi = CA_EPI_GELU_TANH;
      outb = (char *)P.out2;
      ldo = P.ld2;
      ncol = nb - P.n_split;
    } else {
      epi = CA_EPI_BIAS;
    }
  }
  float bias[4 * N_REP];
#pragma unroll
  for (int t = 0; t < 4 * N_REP; ++t) bias[t] = 0.f;
  if (P.bias) {
    const bf16 *bp = (const bf16 *)P.bias + nb;
#pragma unroll
    for (int j = 0; j < N_REP; ++j) {
      const bf16x4 b4 = *(const bf16x4 *)(bp + 4 * j);
#pragma unroll
      for (int r = 0; r < 4; ++r) bias[4 * j + r] = (float)b4[r];
    }
  }
  const int mrow0 = m0 + wm * 16 * M_REP + (lane & 15);

  if (epi == CA_EPI_GATE_RESIDUAL) {
    float gate_a[4 * N_REP], gate_b[4 * N_REP];
    const char *resb = (const char *)P.resid;
#pragma unroll
    for (int i = 0; i < M_REP; ++i) {
      const int m = mrow0 + 16 * i;
      if (m < M) {
        {  // this row's gate vector (per row: a tile may span items and both row ranges)
          const float *gr = ca_gate_of(P, m) + nb;
#pragma unroll
          for (int j = 0; j < N_REP; ++j) {
            const f32x4 ga = *(const f32x4 *)(gr + 4 * j);
#pragma unroll
            for (int r = 0; r < 4; ++r) gate_a[4 * j + r] = gate_b[4 * j + r] = ga[r];
          }
        }
        const bool first = true;
        if (P.out_f32) {  // fp32 residual stream: 16 bytes per 4 columns, read and written in place
          const f32x4 *rp = (const f32x4 *)(resb + ((size_t)m * P.ldr + nb) * 4);
          f32x4 *op = (f32x4 *)(outb + ((size_t)m * ldo + ncol) * 4);
          f32x4 res[N_REP];
#pragma unroll
          for (int j = 0; j < N_REP; ++j) res[j] = rp[j];
#pragma unroll
          for (int j = 0; j < N_REP; ++j) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float gt = first ? gate_a[4 * j + r] : gate_b[4 * j + r];
              v[r] = res[j][r] + gt * (acc[i][j][r] + bias[4 * j + r]);
            }
            op[j] = v;
          }
          continue;
        }
        const uint2 *rp = (const uint2 *)(resb + ((size_t)m * P.ldr + nb) * 2);
        uint2 *op = (uint2 *)(outb + ((size_t)m * ldo + ncol) * 2);
        uint2 res[N_REP];
#pragma unroll
        for (int j = 0; j < N_REP; ++j) res[j] = rp[j];
#pragma unroll
        for (int j = 0; j < N_REP; ++j) {
          float v[4];
          const bf16x4 r4 = __builtin_bit_cast(bf16x4, res[j]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gt = first ? gate_a[4 * j + r] : gate_b[4 * j + r];
            v[r] = (float)r4[r] + gt * (acc[i][j][r] + bias[4 * j + r]);
          }
          op[j] = make_uint2(ca_pack2(v[0], v[1]), ca_pack2(v[2], v[3]));
        }
      }
    }
  } else {
    const bool f32o = P.out_f32 && epi == CA_EPI_BIAS;
#pragma unroll
    for (int i = 0; i < M_REP; ++i) {
      const int m = mrow0 + 16 * i;
      if (m < M) {
        uint2 *op = (uint2 *)(outb + ((size_t)m * ldo + ncol) * 2);
        f32x4 *op32 = (f32x4 *)(outb + ((size_t)m * ldo + ncol) * 4);
#pragma unroll
        for (int j = 0; j < N_REP; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][j][r] + bias[4 * j + r];
            if (epi == CA_EPI_GELU_TANH) v[r] = ca_gelu_tanh(v[r]);
          }
          if (f32o) op32[j] = f32x4{v[0], v[1], v[2], v[3]};
          else op[j] = make_uint2(ca_pack2(v[0], v[1]), ca_pack2(v[2], v[3]));
        }
      }
    }
  }
}

// =============================================================================================
// Ping-pong variant (the fast path for BN = 256 / 128).
//
// The 8 waves form two groups of 4 (one wave of each group per SIMD).  A K tile is consumed in 4
// phases, one C quadrant each:  (A-lo x W-lo), (A-lo x W-hi), (A-hi x W-hi), (A-hi x W-lo), where
// lo/hi are the 128-row halves of the block's A tile and the BN/2-row halves of its W tile, and
// every wave owns 64 rows of each A half and 16*NH rows of each W half.  Per phase a wave
//   reads the fragments that phase introduces (ds_read_b128), issues the global_load_lds of ONE
//   half tile of a later K tile, s_barrier, 16*NH MFMAs, counted s_waitcnt vmcnt, s_barrier.
// Group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA segment while
// the other one reads LDS / issues DMA: the matrix pipe never waits for ds_read latency, and the
// DMA of 2-3 half tiles stays in flight across the barriers (no vmcnt(0) in the loop).
// Hand-off rules (MI355X_MICROARCH.md, LDS-DMA visibility): a half tile staged in phase q is first
// read in phase q+4 or later; every wave waits `vmcnt(cntA+cntW)` (all but the two youngest half
// tiles) before the closing barrier of each phase, which retires everything staged <= q-2 in its
// own group and, one barrier later, in the other group.  Restaging a buffer happens >= 3 phases
// after its last ds_read.  K tiles past the end are staged from the clamped last tile into a
// buffer nobody reads again, so the wait counts stay uniform without a loop tail.
typedef int ca_v4i __attribute__((ext_vector_type(4)));
typedef int ca_v8i __attribute__((ext_vector_type(8)));
// two 16-byte LDS chunks -> the 32-byte fp8 operand of the K=128 MFMA
__device__ __forceinline__ ca_v8i ca_cat32(bf16x8 lo, bf16x8 hi) {
  const ca_v4i a = __builtin_bit_cast(ca_v4i, lo), b = __builtin_bit_cast(ca_v4i, hi);
  return ca_v8i{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

#ifdef CA_GEMM_STAMP
// Diagnostic build only (tools/stamp_gemm.py, tools/stamp_gemm_persist.py): s_memtime of wave 0 of every tile
// (index = tile id of the walk) at tile entry, after the prologue barrier, after the K loop and after the epilogue.
__device__ unsigned long long ca_gemm_dbg[8 * 2048];  // 8 stamps per tile: 0-3 the tile, 4-7 inside the fused qkv epilogue
#define CA_GSTAMP(SLOT)                                                                 \
  {                                                                                     \
    unsigned long long ts_;                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    if (tid == 0 && bid < 2048) ca_gemm_dbg[bid * 8 + (SLOT)] = ts_;                    \
  }
#else
#define CA_GSTAMP(SLOT)
#endif

template <int NL, int NHI>  // 16-column fragments per wave in the lo / hi half of the W tile
struct PPCfg {
  static constexpr int BM = 256, BN = 64 * (NL + NHI), BK = 64, ROW_BYTES = 128;
  static constexpr int A_HALF = 128 * ROW_BYTES;      // 16 KB
  static constexpr int WL_BYTES = 64 * NL * ROW_BYTES;
  static constexpr int WH_BYTES = 64 * NHI * ROW_BYTES;
  static constexpr int OFF_AL = 0, OFF_AH = A_HALF, OFF_WL = 2 * A_HALF, OFF_WH = 2 * A_HALF + WL_BYTES;
  static constexpr int BUF_BYTES = 2 * A_HALF + WL_BYTES + WH_BYTES;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;
  static constexpr int CNT_A = 2;  // global_load_lds per thread per A half tile (W halves: NL, NHI)
};

template <int N>
__device__ __forceinline__ void ca_wait_vmcnt() {
  static_assert(N >= 0 && N <= 6, "unsupported vmcnt");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

// FP8: A and W hold OCP e4m3 bytes; a 128-byte LDS row is then 128 k-elements and the two 16-byte chunks a
// lane reads per fragment form the 32-byte operand of ONE v_mfma_scale_f32_16x16x128_f8f6f4 (E8M0 scales fixed
// at 2^0; the lane->k assignment is the same for both operands, which is all an unscaled product needs:
// tools/micro/mfma_fp8_probe.hip).  Same staging, barriers and wait counts; half the MFMA instructions, each
// twice as long, so a K tile costs the same cycles and carries twice the FLOPs.  Row scales of A and W
// (per-token / per-output-channel absmax quantisation) multiply the accumulators before the epilogue.
template <int NL, int NHI, bool FP8>
__device__ __forceinline__ void ca_gemm_pp_tile(const GemmLaunch &L, char *smem, const int bid, const int nblk) {
  using C = PPCfg<NL, NHI>;
  constexpr int NT_ = NL + NHI;
  constexpr uint32_t ES = FP8 ? 1u : 2u;  // bytes per operand element
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  CA_GSTAMP(0);

  int prob, mtile, ntile;
  if (bid < L.main_total) {
    const int xcd = bid & 7;
    int lid;
    if (L.xcd_interleave && bid < (L.main_total & ~255)) {
      // the 8 XCDs share one run of 256 consecutive tiles per full round (XCD x: tiles 32 x .. 32 x + 31 of it); the
      // partial last round keeps the XCD-contiguous split (any bijection does: its tiles are a few per XCD)
      lid = (bid & ~255) + 32 * xcd + ((bid >> 3) & 31);
    } else {
      const int base = L.xcd_interleave ? (L.main_total & ~255) : 0, tot = L.main_total - base, b = bid - base;
      const int q8 = tot >> 3, r8 = tot & 7;
      lid = base + (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    }
    prob = (lid >= L.ntiles_main[0]) ? 1 : 0;
    if (prob) lid -= L.ntiles_main[0];
    const int MT = L.mt_main[prob], NT = L.nt[prob];
    const int GM = L.group_m;
    const int grp = lid / (GM * NT);
    const int first_m = grp * GM;
    const int gm = min(GM, MT - first_m);
    const int in_grp = lid - grp * GM * NT;
    mtile = first_m + in_grp % gm;
    ntile = in_grp / gm;
  } else {  // thin tiles, after all main tiles: the last row tile of a problem, one tile per column block
    int u = bid - L.main_total;
    prob = (u >= L.nthin[0]) ? 1 : 0;
    if (prob) u -= L.nthin[0];
    mtile = L.mt[prob] - 1;
    ntile = u;
  }
  // a copy (scalars in SGPRs), not a reference into the by-value kernel argument: with field loads inside the
  // unrolled epilogue loops hipcc stops promoting the argument and indexes a SCRATCH copy of the whole descriptor
  // (seen as ScratchSize 392 with zero spills in -Rpass-analysis=kernel-resource-usage)
  const ca_gemm_problem P = L.p[prob];
  const int m0 = mtile * C::BM;
  const int n0 = ntile * C::BN;
  const int M = P.M;
  // 16-row fragments of this wave with at least one valid row, in the lo / hi half of the A tile (4 / 4 but in a
  // last row tile): the MFMAs of the others are skipped
  const int rows_valid = min(M - m0, C::BM);
  const int nv_lo = max(0, min(4, (rows_valid - wm * 64 + 15) >> 4));
  const int nv_hi = max(0, min(4, (rows_valid - 128 - wm * 64 + 15) >> 4));
  const char *Ab = (const char *)P.A;
  const char *Wb = (const char *)P.W;
  const int nk = (int)((uint32_t)P.K * ES / C::ROW_BYTES);

  // ---- per-lane source offsets (bytes, k = 0) of the staging loads of each half tile
  uint32_t offA[2][C::CNT_A], offWL[NL], offWH[NHI];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < C::CNT_A; ++i) {
      const int rr = i * 64 + wave * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((rr >> 1) & 7);
      const int row = min(m0 + h * 128 + rr, M - 1);
      offA[h][i] = (uint32_t)row * (uint32_t)P.lda * ES + (uint32_t)c * 16u;
    }
  auto w_src = [&](int rr, int nfrag, int nbase) {
    // LDS row rr of a W half holds weight row nbase + perm(rr): within each wave's 16*nfrag rows the
    // order is permuted so that a lane's accumulators are 4*nfrag contiguous output columns
    const int c = (lane & 7) ^ ((rr >> 1) & 7);
    const int wloc = rr % (16 * nfrag);
    const int j = wloc >> 4, a = (wloc >> 2) & 3, b = wloc & 3;
    const int n = nbase + (rr - wloc) + 4 * nfrag * a + 4 * j + b;
    return (uint32_t)n * (uint32_t)P.ldw * ES + (uint32_t)c * 16u;
  };
#pragma unroll
  for (int i = 0; i < NL; ++i) offWL[i] = w_src(i * 64 + wave * 8 + (lane >> 3), NL, n0);
#pragma unroll
  for (int i = 0; i < NHI; ++i) offWH[i] = w_src(i * 64 + wave * 8 + (lane >> 3), NHI, n0 + 64 * NL);

  auto stageA = [&](int buf, int h, int kt) {
    const uint32_t kb = (uint32_t)min(kt, nk - 1) * (C::BK * 2);
    char *base = smem + buf * C::BUF_BYTES + (h ? C::OFF_AH : C::OFF_AL) + wave * 8 * C::ROW_BYTES;
#pragma unroll
    for (int i = 0; i < C::CNT_A; ++i) ca_glds16(Ab + offA[h][i] + kb, base + i * 64 * C::ROW_BYTES);
  };
  auto stageWL = [&](int buf, int kt) {
    const uint32_t kb = (uint32_t)min(kt, nk - 1) * (C::BK * 2);
    char *base = smem + buf * C::BUF_BYTES + C::OFF_WL + wave * 8 * C::ROW_BYTES;
#pragma unroll
    for (int i = 0; i < NL; ++i) ca_glds16(Wb + offWL[i] + kb, base + i * 64 * C::ROW_BYTES);
  };
  auto stageWH = [&](int buf, int kt) {
    const uint32_t kb = (uint32_t)min(kt, nk - 1) * (C::BK * 2);
    char *base = smem + buf * C::BUF_BYTES + C::OFF_WH + wave * 8 * C::ROW_BYTES;
#pragma unroll
    for (int i = 0; i < NHI; ++i) ca_glds16(Wb + offWH[i] + kb, base + i * 64 * C::ROW_BYTES);
  };

  f32x4 acc[8][NT_];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NT_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lane_off = (lane & 15) * C::ROW_BYTES + ((((lane >> 4) ^ ((lane & 15) >> 1)) & 7) << 4);
  const int a_off = wm * 64 * C::ROW_BYTES + lane_off;  // + half offset + mi*16 rows

  bf16x8 af[4][2], wl[NL][2], wh[NHI][2];
  auto readA = [&](int buf, int h, int nv = 4) {  // nv: fragments to read (thin tiles: the valid ones)
    const char *b = smem + buf * C::BUF_BYTES + (h ? C::OFF_AH : C::OFF_AL);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      if (mi < nv)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          af[mi][ks] = *(const bf16x8 *)(b + ((a_off + mi * 16 * C::ROW_BYTES) ^ (ks * 64)));
  };
  auto readWL = [&](int buf) {
    const char *b = smem + buf * C::BUF_BYTES + C::OFF_WL;
    const int w_off = wn * 16 * NL * C::ROW_BYTES + lane_off;
#pragma unroll
    for (int nj = 0; nj < NL; ++nj)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wl[nj][ks] = *(const bf16x8 *)(b + ((w_off + nj * 16 * C::ROW_BYTES) ^ (ks * 64)));
  };
  auto readWH = [&](int buf) {
    const char *b = smem + buf * C::BUF_BYTES + C::OFF_WH;
    const int w_off = wn * 16 * NHI * C::ROW_BYTES + lane_off;
#pragma unroll
    for (int nj = 0; nj < NHI; ++nj)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wh[nj][ks] = *(const bf16x8 *)(b + ((w_off + nj * 16 * C::ROW_BYTES) ^ (ks * 64)));
  };
#define CA_PP_MMA_BODY(MI0, NJ0, WF, NW, GUARD)                                                             \
  if constexpr (FP8) {                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) if (GUARD) _Pragma("unroll") for (int nj = 0; nj < (NW); ++nj) \
        acc[(MI0) + mi][(NJ0) + nj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                     \
            ca_cat32(WF[nj][0], WF[nj][1]), ca_cat32(af[mi][0], af[mi][1]), acc[(MI0) + mi][(NJ0) + nj],    \
            0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);                                                            \
  } else {                                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)       \
        if (GUARD) _Pragma("unroll") for (int nj = 0; nj < (NW); ++nj) acc[(MI0) + mi][(NJ0) + nj] =        \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[nj][ks], af[mi][ks], acc[(MI0) + mi][(NJ0) + nj],    \
                                                    0, 0, 0);                                               \
  }
// NV = valid 16-row fragments of this wave in the A half; only the THIN instantiation of the K loop tests it
#define CA_PP_MMA(MI0, NJ0, WF, NW, NV)                                                                     \
  {                                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    if constexpr (THIN) {                                                                                   \
      CA_PP_MMA_BODY(MI0, NJ0, WF, NW, mi < (NV))                                                           \
    } else {                                                                                                \
      CA_PP_MMA_BODY(MI0, NJ0, WF, NW, true)                                                                \
    }                                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                          \
  }
#define CA_PP_SYNC()                        \
  __builtin_amdgcn_sched_barrier(0);        \
  __builtin_amdgcn_s_barrier();             \
  __builtin_amdgcn_sched_barrier(0)
#define CA_PP_WAIT_READS()                               \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
  __builtin_amdgcn_sched_barrier(0)

#if CA_GEMM_TWO_PHASE
  // ---- TWO phases per K tile: (A-lo x W-lo, W-hi) then (A-hi x W-lo, W-hi), 32 MFMAs per wave and phase, so the
  // loop passes 4 barriers per K tile instead of 8 (the barrier hand-off between the two wave groups, not LDS
  // or MFMA issue, is what the 4-phase loop loses: 2458 vs 2048 cycles per K tile).  Staging and hand-off:
  //   read segment of phase 1 (tile t): issue W-hi(t+1), A-hi(t+1) into the other buffer -- their tile t-1
  //     versions were last read by the other group one interval ago, behind a barrier;
  //   read segment of phase 2: issue A-lo(t+2), W-lo(t+2) into THIS buffer -- both groups read their tile t
  //     versions in phase 1;
  //   every read segment ends with vmcnt(what it just issued): everything a wave issued in its PREVIOUS read
  //     segment (two intervals = ~1000 cycles earlier) has then landed, and the barrier that follows publishes
  //     it.  First reader of a half tile: W-hi / A-hi(t+1) in phase 1 / 2 of tile t+1, i.e. 4 / 6 intervals after
  //     the early group issued and 3 / 5 after the late group did, whose retiring read segment is interval 3 / 3:
  //     one barrier before the read at the closest.  A-lo / W-lo(t+2): 6 intervals.
  // Prologue: tile 0 complete, A-lo(1), W-lo(1) in flight (retired by phase 1 of tile 0).
  stageA(0, 0, 0);
  stageWL(0, 0);
  stageWH(0, 0);
  stageA(0, 1, 0);
  stageA(1, 0, 1);
  stageWL(1, 1);
  ca_wait_vmcnt<C::CNT_A + NL>();
  CA_PP_SYNC();
  CA_GSTAMP(1);
  if (wm == 1) { CA_PP_SYNC(); }  // stagger: group 1 runs one barrier behind group 0
  // The loop exists twice: as written for full tiles, and for thin tiles (a last row tile with few valid rows,
  // walked last) with the LDS reads and MFMAs of row fragments past M left out -- same staging, waits and barriers,
  // so the hand-off rules above hold unchanged; what remains paces at the staging (two DMA round trips per K tile):
  // about 0.75 of a full tile's time.  (Streaming a <= 32-row tile's operands straight into MFMA registers, four K
  // tiles in flight and no barrier, was built too: bit-identical, but no faster at K = 3072 and 1.6x slower at
  // K = 12288 -- 16 rows x 64 bytes per load instruction is a poor pattern for the vector L1.)
  auto kloop = [&](auto thin_tag) {
    constexpr bool THIN = decltype(thin_tag)::value;
    const int rlo = THIN ? nv_lo : 4, rhi = THIN ? nv_hi : 4;
    for (int t = 0; t < nk; ++t) {
      const int b = t & 1;
      // phase 1
      readA(b, 0, rlo);
      if (!THIN || rlo + rhi > 0) {
        readWL(b);
        readWH(b);
      }
      stageWH(b ^ 1, t + 1);
      stageA(b ^ 1, 1, t + 1);
      CA_PP_WAIT_READS();
      ca_wait_vmcnt<NHI + C::CNT_A>();
      CA_PP_SYNC();
      CA_PP_MMA(0, 0, wl, NL, rlo);
      CA_PP_MMA(0, NL, wh, NHI, rlo);
      CA_PP_SYNC();
      // phase 2
      readA(b, 1, rhi);
      stageA(b, 0, t + 2);
      stageWL(b, t + 2);
      CA_PP_WAIT_READS();
      ca_wait_vmcnt<C::CNT_A + NL>();
      CA_PP_SYNC();
      CA_PP_MMA(4, 0, wl, NL, rhi);
      CA_PP_MMA(4, NL, wh, NHI, rhi);
      CA_PP_SYNC();
    }
  };
  if (rows_valid <= CA_GEMM_THIN_ROWS) kloop(std::true_type{});
  else kloop(std::false_type{});
#else
  constexpr bool THIN = false;  // (the 4-phase loop has no thin variant)
  // ---- prologue: AL(0) WL(0) WH(0) AH(0) WL(1).  Group 1 executes no retire-wait between this
  // barrier and group 0's first read of W-hi(0), so everything but AH(0), WL(1) must have landed.
  stageA(0, 0, 0);
  stageWL(0, 0);
  stageWH(0, 0);
  stageA(0, 1, 0);
  stageWL(1, 1);
  ca_wait_vmcnt<C::CNT_A + NL>();
  CA_PP_SYNC();
  CA_GSTAMP(1);
  if (wm == 1) { CA_PP_SYNC(); }  // stagger: group 1 runs one barrier behind group 0

  // retire-waits leave the two youngest half tiles in flight: phase 1: WL(t+2 of the previous
  // tile's phase 4) + AL; phase 2: AL + WH; phase 3: WH + AH; phase 4: AH + WL
  for (int t = 0; t < nk; ++t) {
    const int b = t & 1;
    // phase 1: (A-lo, W-lo)
    readA(b, 0);
    readWL(b);
    stageA(b ^ 1, 0, t + 1);
    CA_PP_WAIT_READS();
    CA_PP_SYNC();
    CA_PP_MMA(0, 0, wl, NL, nv_lo);
    ca_wait_vmcnt<NL + C::CNT_A>();
    CA_PP_SYNC();
    // phase 2: (A-lo, W-hi)
    readWH(b);
    stageWH(b ^ 1, t + 1);
    CA_PP_WAIT_READS();
    CA_PP_SYNC();
    CA_PP_MMA(0, NL, wh, NHI, nv_lo);
    ca_wait_vmcnt<C::CNT_A + NHI>();
    CA_PP_SYNC();
    // phase 3: (A-hi, W-hi)
    readA(b, 1);
    stageA(b ^ 1, 1, t + 1);
    CA_PP_WAIT_READS();
    CA_PP_SYNC();
    CA_PP_MMA(4, NL, wh, NHI, nv_hi);
    ca_wait_vmcnt<NHI + C::CNT_A>();
    CA_PP_SYNC();
    // phase 4: (A-hi, W-lo); W-lo of tile t+2 goes into the buffer whose W-lo was consumed in phase 1
    stageWL(b, t + 2);
    CA_PP_SYNC();
    CA_PP_MMA(4, 0, wl, NL, nv_hi);
    ca_wait_vmcnt<C::CNT_A + NL>();
    CA_PP_SYNC();
  }
#endif
  if (wm == 0) { CA_PP_SYNC(); }
  // No LDS-DMA may be outstanding when the workgroup retires or the epilogue reuses the LDS.  Through the BUILTIN
  // (0x0F70 = vmcnt 0): hipcc's wait-count pass does not see waits in inline asm, keeps believing that the loop's
  // global_load_lds are in flight, and then puts an s_waitcnt vmcnt(0) in front of every LDS read of the epilogue --
  // which there waits for the acknowledgement of every global store issued so far (the fused QK-norm + RoPE epilogue
  // read its row sums fragment by fragment between stores: 940 cycles per fragment).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  CA_GSTAMP(2);
  if constexpr (FP8) {
    // dequantise: acc[m][n] *= a_scale[m] * w_scale[n] (column order of the accumulators: see below)
    float sa[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
      sa[mi] = P.a_scale[min(m0 + (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15), M - 1)];
    // (two passes, row scale then column scale: with the product sa[mi] * sw[r] written out hipcc builds the whole
    // 8 x 16 table of products next to the 128 accumulators and spills 44 registers around it)
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int nj = 0; nj < NT_; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][nj][r] *= sa[mi];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nj = 0; nj < NT_; ++nj) {
      const int n = nj < NL ? n0 + wn * 16 * NL + 4 * NL * (lane >> 4) + 4 * nj
                            : n0 + 64 * NL + wn * 16 * NHI + 4 * NHI * (lane >> 4) + 4 * (nj - NL);
      const f32x4 sw = *(const f32x4 *)(P.w_scale + n);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][nj][r] *= sw[r];
    }
    // keep the epilogue's loads (bias, gates, residual rows) below this point: hoisted above the scaling they
    // overlap its temporaries with all 128 accumulators and spill
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue.  acc[mi][nj][r] = C[m][n]:
  //   m = m0 + (mi>>2)*128 + wm*64 + 16*(mi&3) + (lane&15)
  //   lo half (nj <  NL): n = n0 +          wn*16*NL  + 4*NL *(lane>>4) + 4*nj      + r
  //   hi half (nj >= NL): n = n0 + 64*NL +  wn*16*NHI + 4*NHI*(lane>>4) + 4*(nj-NL) + r
  const int g = lane >> 4;
  int epi = P.epilogue;
  char *outb = (char *)P.out;
  int ldo = P.ldc;
  int col_shift = 0;
  if constexpr (NL == 2 && NHI == 2) {
    // ---- fused QK-RMSNorm + RoPE on the q and k thirds of a qkv projection (one head = one 128-column
    // half of the tile).  Per row and head the sum of squares is reduced over the lane's 8 columns, the 4
    // lane groups (shuffles) and the 4 column-waves (through LDS, which the main loop no longer uses).
    if (epi == CA_EPI_QKV_NORM_ROPE && n0 < (P.n_split / 3) * 2) {
      const int hd = P.n_split / 3;               // heads * 128
      const bool is_q = n0 < hd;
      // (launch-uniform; bf16 operands only: the low-plane projection never runs in e4m3, and the fp8 instantiation
      // has no registers for it -- gemm_impl rejects the combination)
      const bool add_q = !FP8 && is_q && P.q_prerope && P.qpre_f32 == 3;
      const bf16 *nscale = (const bf16 *)(is_q ? P.norm_q : P.norm_k);
      float *part = (float *)smem;                // [2 halves][256 rows][4 column-waves]
      float x[8][2][8];
      // RoPE values of this lane's 8 rows x 8 columns-in-head (the same for both heads of the tile) and the norm
      // scales: every global load of this epilogue is issued here, before its first store, and waited for ONCE below.
      // (Loaded next to their use, each load -- and, through hipcc's vmcnt(0) in front of every use, each fragment --
      // waited for the acknowledgement of the previous fragment's stores: vmcnt retires in order and counts stores.
      // Stamps of wave 0, per head of 8 fragments: 7500 -> 3700 cycles.)  Round 3: issued before the row sums are formed,
      // so that their latency (4 000 cycles between issue and use on wave 0's stamps) runs beside that work.
      // (the bias of both heads first: vmcnt retires in order, so the wait for a bias loaded behind the RoPE values
      // would be a wait for all of them)
      const int cih = wn * 32 + 8 * g;             // column inside the head
      bf16x8 b8h[2] = {};
      if (P.bias) {
        b8h[0] = *(const bf16x8 *)((const bf16 *)P.bias + n0 + cih);
        b8h[1] = *(const bf16x8 *)((const bf16 *)P.bias + n0 + 128 + cih);
      }
      const float qos = P.q_out_scale == 0.0f ? 1.0f : P.q_out_scale;
      f32x4 rope0[8], rope1[8];
      bf16x8 s8;                                   // norm scales of this lane's 8 columns-in-head (both heads)
      auto load_rope = [&]() {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const int m = min(m0 + (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15), M - 1);
          const float *rp = P.rope + (size_t)m * 128 + cih;  // [64 pairs][cos,sin]; pairs cih/2 .. cih/2+3
          rope0[mi] = *(const f32x4 *)rp;
          rope1[mi] = *(const f32x4 *)(rp + 4);
        }
        s8 = *(const bf16x8 *)(nscale + cih);
      };
      if (add_q) {
        // qpre_f32 = 3: the hi plane's raw projection, left in q_prerope by the main qkv launch, joins the accumulators
        // ((acc + raw) + bias in both kernels) -- before the RoPE values are fetched: their 64 registers and these 8 do
        // not fit beside the 128 accumulators
#pragma unroll
        for (int hn = 0; hn < 2; ++hn)
#pragma unroll
          for (int mi = 0; mi < 8; ++mi) {
            const int m = min(m0 + (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15), M - 1);
            const float *qp = (const float *)P.q_prerope + (size_t)m * P.ldp + n0 + hn * 128 + cih;
            const f32x4 a0 = *(const f32x4 *)qp, a1 = *(const f32x4 *)(qp + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[mi][hn * 2][r] = acc[mi][hn * 2][r] + a0[r];
              acc[mi][hn * 2 + 1][r] = acc[mi][hn * 2 + 1][r] + a1[r];
            }
          }
      }
      if constexpr (!FP8) load_rope();   // (the fp8 instantiation has no registers for it here: behind the row sums)
      __syncthreads();  // every wave has retired its own LDS-DMA (vmcnt 0 above): the LDS is reusable
#pragma unroll
      for (int hn = 0; hn < 2; ++hn) {
        float bias[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) bias[t] = P.bias ? (float)b8h[hn][t] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          float sq = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = acc[mi][hn * 2 + j][r] + bias[4 * j + r];
              x[mi][hn][4 * j + r] = v;
              sq = __builtin_fmaf(v, v, sq);  // (explicit: the thin-row kernel must round as this one does)
            }
          sq += __shfl_xor(sq, 16);
          sq += __shfl_xor(sq, 32);
          const int rl = (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15);
          if (g == 0) part[(hn * 256 + rl) * 4 + wn] = sq;
        }
      }
      if constexpr (FP8) load_rope();
      CA_GSTAMP(4);
      __syncthreads();
      // every load of this epilogue has been issued: wait for them HERE, once, through the builtin.  Otherwise hipcc
      // covers each use of the RoPE values below with s_waitcnt vmcnt(0), which by then also waits for the stores of
      // the fragments before it (vmcnt counts stores and retires in order).
      __builtin_amdgcn_s_waitcnt(0x0F70);
      CA_GSTAMP(5);
#pragma unroll
      for (int hn = 0; hn < 2; ++hn) {
        if (hn == 1) { CA_GSTAMP(6); }
        const int head_col = (n0 - (is_q ? 0 : hd)) + hn * 128;  // first column of this head in its third
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const int rl = (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15);
          const int m = m0 + rl;
          if (m >= M) continue;
          const f32x4 p4 = *(const f32x4 *)(part + (hn * 256 + rl) * 4);
          const float rrms = rsqrtf(__builtin_fmaf(p4[0] + p4[1] + p4[2] + p4[3], 1.0f / 128.0f, 1e-6f));
          float y[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) y[t] = x[mi][hn][t] * rrms * (float)s8[t];
          if (is_q && P.q_prerope) {
            if (P.qpre_f32 == 2) {   // the projection itself (bias included), before the norm: ca_qpre_finish_f32
              float *qp = (float *)P.q_prerope + (size_t)m * P.ldp + head_col + cih;
              *(f32x4 *)qp = f32x4{x[mi][hn][0], x[mi][hn][1], x[mi][hn][2], x[mi][hn][3]};
              *(f32x4 *)(qp + 4) = f32x4{x[mi][hn][4], x[mi][hn][5], x[mi][hn][6], x[mi][hn][7]};
            } else if (P.qpre_f32) {
              float *qp = (float *)P.q_prerope + (size_t)m * P.ldp + head_col + cih;
              *(f32x4 *)qp = f32x4{y[0], y[1], y[2], y[3]};
              *(f32x4 *)(qp + 4) = f32x4{y[4], y[5], y[6], y[7]};
            } else {
              *(uint4 *)((bf16 *)P.q_prerope + (size_t)m * P.ldp + head_col + cih) =
                  make_uint4(ca_pack2(y[0], y[1]), ca_pack2(y[2], y[3]), ca_pack2(y[4], y[5]), ca_pack2(y[6], y[7]));
            }
          }
          const f32x4 r0 = rope0[mi], r1 = rope1[mi];
          const float cs[4] = {r0[0], r0[2], r1[0], r1[2]}, sn[4] = {r0[1], r0[3], r1[1], r1[3]};
          float z[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            z[2 * i] = __builtin_fmaf(cs[i], y[2 * i], -(sn[i] * y[2 * i + 1]));
            z[2 * i + 1] = __builtin_fmaf(sn[i], y[2 * i], cs[i] * y[2 * i + 1]);
          }
          if (is_q && qos != 1.0f) {  // (uniform branch; explicit multiply: the thin-row kernel must round alike)
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = z[i] * qos;
          }
          // (qk_f16: the rotated q / k as IEEE half for ca_attn_fwd_qk16 -- a launch-uniform branch)
          *(uint4 *)(outb + ((size_t)m * ldo + n0 + hn * 128 + cih) * 2) =
              P.qk_f16 ? make_uint4(ca_pack2_f16(z[0], z[1]), ca_pack2_f16(z[2], z[3]), ca_pack2_f16(z[4], z[5]),
                                    ca_pack2_f16(z[6], z[7]))
                       : make_uint4(ca_pack2(z[0], z[1]), ca_pack2(z[2], z[3]), ca_pack2(z[4], z[5]), ca_pack2(z[6], z[7]));
        }
      }
      return;
    }
  }
  if (epi == CA_EPI_QKV_NORM_ROPE) epi = CA_EPI_SPLIT_GELU;  // v third: plain; columns past n_split: GELU -> out2
  if (epi == CA_EPI_SPLIT_GELU) {
    if (n0 >= P.n_split) {
      epi = CA_EPI_GELU_TANH;
      outb = (char *)P.out2;
      ldo = P.ld2;
      col_shift = -P.n_split;
    } else {
      epi = CA_EPI_BIAS;
    }
  }
  // Bias and gate vectors of BOTH column halves are fetched here, before the first store of the epilogue, and waited
  // for once through the builtin.  The stores below are predicated (m < M), which leaves hipcc's wait-count pass
  // unable to count them: while any load is pending in its books it covers the next use with s_waitcnt vmcnt(0), and
  // after the first store that is a wait for the store's acknowledgement (fetched per half, the second half's bias
  // waited for all of the first half's stores): -0.5...-0.8 % per launch.  (One vmcnt(0) per half remains, right after
  // its first predicated store; a build with unpredicated stores has none and is no faster.)
  const int nb_lo = n0 + wn * 16 * NL + 4 * NL * g, nb_hi = n0 + 64 * NL + wn * 16 * NHI + 4 * NHI * g;
  bf16x4 braw_lo[NL], braw_hi[NHI];
  f32x4 graw_lo[NL], graw_hi[NHI];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    braw_lo[j] = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    graw_lo[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < NHI; ++j) {
    braw_hi[j] = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    graw_hi[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (P.bias) {
#pragma unroll
    for (int j = 0; j < NL; ++j) braw_lo[j] = *(const bf16x4 *)((const bf16 *)P.bias + nb_lo + 4 * j);
#pragma unroll
    for (int j = 0; j < NHI; ++j) braw_hi[j] = *(const bf16x4 *)((const bf16 *)P.bias + nb_hi + 4 * j);
  }
  if (epi == CA_EPI_GATE_RESIDUAL) {
    const float *gv = ca_gate_of(P, m0);
#pragma unroll
    for (int j = 0; j < NL; ++j) graw_lo[j] = *(const f32x4 *)(gv + nb_lo + 4 * j);
#pragma unroll
    for (int j = 0; j < NHI; ++j) graw_hi[j] = *(const f32x4 *)(gv + nb_hi + 4 * j);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing of this tile has been stored yet
  auto half_epilogue = [&](auto nfrag_tag, int nj0, int nb, const auto &braw, const auto &graw) {
    constexpr int NF = decltype(nfrag_tag)::value;
    float bias[4 * NF], gate_a[4 * NF], gate_b[4 * NF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bias[4 * j + r] = (float)braw[j][r];
        gate_a[4 * j + r] = gate_b[4 * j + r] = graw[j][r];
      }
    // Gates: the usual tile lies inside one work item and one row range, so its rows share ONE gate vector, loaded
    // once (gate_a; `straddle` false).  A tile that spans items or the gate_rows boundary (the concept | text tile)
    // fetches the vector per row fragment instead.
    bool straddle = false;
    if (epi == CA_EPI_GATE_RESIDUAL) {
      // (rows are monotonic in (range, item), so equal end points mean one vector for the whole tile; comparing the
      // two POINTERS would not do: text rows of other items can lie between two rows that share a vector)
      const int m_hi = min(m0 + C::BM, M) - 1;
      const float *g_lo = ca_gate_of(P, m0);
      straddle = (m0 < P.gate_rows) != (m_hi < P.gate_rows) || g_lo != ca_gate_of(P, m_hi);  // workgroup-uniform
    }
    auto row_gate = [&](int m) {  // straddling tile only: this lane's row m has its own vector
      const float *gr = ca_gate_of(P, min(m, M - 1)) + nb;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const f32x4 ga = *(const f32x4 *)(gr + 4 * j);
#pragma unroll
        for (int r = 0; r < 4; ++r) gate_a[4 * j + r] = ga[r];
      }
    };
    // the residual rows are fetched up front for all 8 row fragments (rows clamped, only the store is
    // predicated): one round trip to memory instead of eight dependent ones
    if (P.out_f32 && (epi == CA_EPI_GATE_RESIDUAL || epi == CA_EPI_BIAS)) {
      // fp32 output (the residual stream): a lane's 4*NF columns are 16*NF bytes; GATE_RESIDUAL reads the old
      // value of the same element first.  Four row fragments are fetched up front (32*NF bytes per lane each).
      const bool gated = epi == CA_EPI_GATE_RESIDUAL;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        f32x4 r32[4][NF];
        if (gated) {
#pragma unroll
          for (int mq = 0; mq < 4; ++mq) {
            const int mi = mh * 4 + mq;
            const int m = min(m0 + (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15), M - 1);
            const f32x4 *rp = (const f32x4 *)((const char *)P.resid + ((size_t)m * P.ldr + nb) * 4);
#pragma unroll
            for (int j = 0; j < NF; ++j) r32[mq][j] = rp[j];
          }
        }
#pragma unroll
        for (int mq = 0; mq < 4; ++mq) {
          const int mi = mh * 4 + mq;
          const int m = m0 + (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15);
          const bool first = true;
          if (gated && straddle) row_gate(m);
          f32x4 *op = (f32x4 *)(outb + ((size_t)m * ldo + nb + col_shift) * 4);
#pragma unroll
          for (int j = 0; j < NF; ++j) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[mi][nj0 + j][r] + bias[4 * j + r];
              if (gated) v[r] = __builtin_fmaf(first ? gate_a[4 * j + r] : gate_b[4 * j + r], v[r], r32[mq][j][r]);
            }
            if (m < M) op[j] = v;
          }
        }
      }
      return;
    }
    uint2 res[8][NF];
    if (epi == CA_EPI_GATE_RESIDUAL) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int m = min(m0 + (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15), M - 1);
        const uint2 *rp = (const uint2 *)((const char *)P.resid + ((size_t)m * P.ldr + nb) * 2);
#pragma unroll
        for (int j = 0; j < NF; ++j) res[mi][j] = rp[j];
      }
    }
    // one loop per epilogue kind under a workgroup-uniform branch: with the kind tested per element hipcc
    // if-converts the test and every bias-only tile pays the 64 v_exp_f32 + 64 v_rcp_f32 per wave and half of a GELU
    auto rows = [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int m = m0 + (mi >> 2) * 128 + wm * 64 + 16 * (mi & 3) + (lane & 15);
        uint2 *op = (uint2 *)(outb + ((size_t)m * ldo + nb + col_shift) * 2);
        const bool first = true;
        if (KIND == CA_EPI_GATE_RESIDUAL && straddle) row_gate(m);
        uint2 o[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[mi][nj0 + j][r] + bias[4 * j + r];
            if (KIND == CA_EPI_GELU_TANH) v[r] = ca_gelu_tanh(v[r]);
          }
          if (KIND == CA_EPI_GATE_RESIDUAL) {
            const bf16x4 r4 = __builtin_bit_cast(bf16x4, res[mi][j]);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(first ? gate_a[4 * j + r] : gate_b[4 * j + r], v[r], (float)r4[r]);
          }
          o[j] = make_uint2(ca_pack2(v[0], v[1]), ca_pack2(v[2], v[3]));
        }
        if (m < M) {
          if constexpr (NF == 2) {
            *(uint4 *)op = make_uint4(o[0].x, o[0].y, o[1].x, o[1].y);
          } else {
            op[0] = o[0];
          }
        }
      }
    };
    if (epi == CA_EPI_GELU_TANH) rows(std::integral_constant<int, CA_EPI_GELU_TANH>{});
    else if (epi == CA_EPI_GATE_RESIDUAL) rows(std::integral_constant<int, CA_EPI_GATE_RESIDUAL>{});
    else rows(std::integral_constant<int, CA_EPI_BIAS>{});
  };
  half_epilogue(std::integral_constant<int, NL>{}, 0, nb_lo, braw_lo, graw_lo);
  half_epilogue(std::integral_constant<int, NHI>{}, NL, nb_hi, braw_hi, graw_hi);
#ifdef CA_GEMM_STAMP
  if (CA_GEMM_STAMP != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // 2: stores left in flight (the walk as shipped)
  CA_GSTAMP(3);
#endif
}

// One workgroup per tile, or (persist_tiles > 0) a grid of one workgroup per CU walking the tiles with a stride of
// the grid size: tile v keeps the XCD class of its workgroup (grid % 8 == 0), the next tile's prologue DMA
// is issued while the previous tile's stores are still draining, and there is no workgroup dispatch between
// rounds.  The barrier between tiles protects the QK-norm partial sums, which live in the first tile buffer.
template <int NL, int NHI, bool FP8 = false>
__global__ __launch_bounds__(512, 2) void ca_gemm_pp_kernel(const GemmLaunch L) {
  extern __shared__ __attribute__((aligned(128))) char smem[];
  if (L.persist_tiles > 0) {
    for (int v = blockIdx.x; v < L.persist_tiles; v += gridDim.x) {
      ca_gemm_pp_tile<NL, NHI, FP8>(L, smem, v, L.persist_tiles);
      __syncthreads();
    }
  } else {
    ca_gemm_pp_tile<NL, NHI, FP8>(L, smem, blockIdx.x, gridDim.x);
  }
}

// =============================================================================================
// Thin-row kernel: the last row tile of a problem when it is thin (<= CA_GEMM_THIN_ROWS rows: the 5 x 4 concept rows
// that the [concept | text] stream of a 5-item double block carries past its 5 full row tiles; the 4 rows a single
// item leaves over), 32 rows per workgroup (grid y).  As 256-column tiles of the ping-pong kernel those rows cost a launch 12 full-price tiles -- one
// CU streams 256 weight rows per tile through its vector L1 whatever the row count -- and with them a fifth round
// (section "the 20 rows" of DESIGN.md).  Here they are 32 x 128 tiles, one workgroup of 4 waves each, so the weight
// rows of the thin part are spread over N / 128 CUs; a wave owns 32 columns = the two 16-column fragments and the
// weight-row permutation of a ping-pong wave, so a lane again holds 8 contiguous columns of row lane&15 and the
// epilogues are the ping-pong kernel's, fragment for fragment.  Same MFMA, same operand roles, same k order per
// accumulator and the same expression order in every epilogue: results are bit-identical to the same rows inside a
// full ping-pong tile (tests/test_kernels_gpu.py::test_gemm_thin_last_row_tile_is_bit_identical).
// Staging: a K tile is 32 activation rows + 128 weight rows of 128 bytes = 20 pieces of 1 KB by LDS-DMA, 5 per wave,
// into a ring of 4 slots; the DMA of tile t+3 is issued right after the barrier of iteration t (its slot held tile
// t-1, read before that barrier) and every iteration opens with vmcnt(10) = "all but my two youngest tiles have
// landed" followed by the barrier that publishes tile t: one barrier per K tile, three tiles in flight.
// MF = 16-row fragments per workgroup: 2 (32 rows: the concept rows of a batch) or 4 (64 rows: longer thin parts and
// the 2 x items x steps conditioning vectors of the modulation GEMM, so that the weights stream once per 64 rows)
// NW = waves per workgroup (round 5): 4 (128 columns = one head: the fused QK-norm + RoPE epilogue reduces a head's row
// sums over its 4 column-waves) or 1 (32 columns, every other epilogue).  What bounds a thin launch is how many CUs
// stream weight rows: with 128-column workgroups N = 3072 gave 24 of them -- 24 CUs each pulling 128 rows x K through
// one vector L1 (mlp.2, K = 12288: 3.1 MB per CU) -- with 32-column workgroups it is 96.  Same fragments, same k order
// per accumulator: a row's bits do not depend on NW.
// A one-wave workgroup's K loop is a latency chain (wait for the tile staged SLOTS - 1 iterations ago, barrier, 8 MFMAs);
// its LDS stage is small (8 / 12 KB), so its ring is deeper: 8 / 6 slots = 7 / 5 K tiles in flight (56 / 60 of the 63
// loads a wave may have outstanding).  The four-wave form keeps 4 slots: 6 / 7 were measured and change nothing (its
// launches -- the modulation GEMM's 8 126 workgroups at 4.6-5.6 TB/s -- are not short of bytes in flight;
// tools/thin_ab.py, round 5).
constexpr int THIN_N = 128;
template <int MF, int NW>
struct ThinCfg {
  static constexpr int M = 16 * MF, N = 32 * NW, STAGE = (M + N) * 128;
  static constexpr int PIECES = (M + N) / 8 / NW;  // 1 KB pieces per wave and K tile (NW = 4: 5 or 6; NW = 1: 8 or 12)
  static constexpr int SLOTS = NW == 4 ? 4 : (MF == 2 ? 8 : 6);
  static constexpr int LDS = SLOTS * STAGE;
  static_assert(PIECES * (SLOTS - 1) <= 63, "vmcnt counts at most 63 outstanding loads");
};
template <int N>
__device__ __forceinline__ void ca_wait_vmcnt_imm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MF, int NW>
__global__ __launch_bounds__(64 * NW) void ca_gemm_thin_kernel(const GemmLaunch L) {
  using TC = ThinCfg<MF, NW>;
  constexpr int THIN_M = TC::M, THIN_STAGE = TC::STAGE, NP = TC::PIECES, THIN_N = TC::N, THIN_SLOTS = TC::SLOTS;
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = 32-column group
  int u = blockIdx.x;
  const int nt0 = L.thin_nt[0] * (4 / NW);   // (thin_nt counts 128-column tiles)
  const int prob = (u >= nt0) ? 1 : 0;
  if (prob) u -= nt0;
  const ca_gemm_problem P = L.p[prob];
  const int m0 = L.thin_row0[prob] + THIN_M * (int)blockIdx.y, n0 = u * THIN_N, M = P.M;
  if (m0 >= M) return;  // (two problems with different thin row counts share a grid: whole workgroup)
  const char *Ab = (const char *)P.A;
  const char *Wb = (const char *)P.W;
  const int nk = P.K / 64;

  // ---- staging sources: piece q = 5*wave + i covers LDS rows 8q .. 8q+7 (rows 0..31: A, 32..159: W)
  const char *src[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int rr = 8 * (NP * wn + i) + (lane >> 3);
    if (rr < THIN_M) {
      const int c = (lane & 7) ^ ((rr >> 1) & 7);
      src[i] = Ab + (size_t)min(m0 + rr, M - 1) * P.lda * 2 + c * 16;
    } else {
      const int wr = rr - THIN_M;  // LDS row of the W image: wave wr/32, fragment order as in the ping-pong kernel
      const int c = (lane & 7) ^ ((wr >> 1) & 7);
      const int wloc = wr & 31;
      const int j = wloc >> 4, a = (wloc >> 2) & 3, b = wloc & 3;
      src[i] = Wb + (size_t)(n0 + (wr - wloc) + 8 * a + 4 * j + b) * P.ldw * 2 + c * 16;
    }
  }
  auto stage = [&](int slot, int kt) {
    const int kb = min(kt, nk - 1) * 128;  // (tiles past the end re-stage the last one into a slot nobody reads)
    char *base = smem + slot * THIN_STAGE + NP * wn * 1024;
#pragma unroll
    for (int i = 0; i < NP; ++i) ca_glds16(src[i] + kb, base + i * 1024);
  };
  const int lane_off = (lane & 15) * 128 + ((((lane >> 4) ^ ((lane & 15) >> 1)) & 7) << 4);
  f32x4 acc[MF][2];
#pragma unroll
  for (int mi = 0; mi < MF; ++mi)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[mi][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < THIN_SLOTS - 1; ++i) stage(i, i);
  for (int t = 0; t < nk; ++t) {
    ca_wait_vmcnt_imm<NP * (THIN_SLOTS - 2)>();  // all but my SLOTS - 2 youngest K tiles have landed: tile t is there
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stage((t + THIN_SLOTS - 1) % THIN_SLOTS, t + THIN_SLOTS - 1);
    const char *sb = smem + (t % THIN_SLOTS) * THIN_STAGE;
    bf16x8 af[MF][2], wf[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int mi = 0; mi < MF; ++mi) af[mi][ks] = *(const bf16x8 *)(sb + ((mi * 16 * 128 + lane_off) ^ (ks * 64)));
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[j][ks] = *(const bf16x8 *)(sb + (((THIN_M + wn * 32 + j * 16) * 128 + lane_off) ^ (ks * 64)));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], af[mi][ks], acc[mi][j], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // no LDS-DMA outstanding, every wave done reading: the LDS is free for the row sums

  // ---- epilogues: acc[mi][j][r] = C[m][n], m = m0 + 16*mi + (lane&15), n = n0 + wn*32 + 8*(lane>>4) + 4*j + r
  const int g = lane >> 4, l15 = lane & 15;
  const int nb = n0 + wn * 32 + 8 * g;
  int epi = P.epilogue;
  char *outb = (char *)P.out;
  int ldo = P.ldc, col_shift = 0;
  float bias[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) bias[t] = 0.f;
  if (P.bias) {
    const bf16x8 b8 = *(const bf16x8 *)((const bf16 *)P.bias + nb);
#pragma unroll
    for (int t = 0; t < 8; ++t) bias[t] = (float)b8[t];
  }
  if (NW == 4 && epi == CA_EPI_QKV_NORM_ROPE && n0 < (P.n_split / 3) * 2) {  // this tile = one head of q or k
    const int hd = P.n_split / 3;
    const bool is_q = n0 < hd;
    const bool add_q = is_q && P.q_prerope && P.qpre_f32 == 3;
    const bf16 *nscale = (const bf16 *)(is_q ? P.norm_q : P.norm_k);
    float *part = (float *)smem;  // [rows][4 column-waves]
    float x[MF][8];
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      float sq = 0.f;
      float qa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (add_q) {
        const float *qp = (const float *)P.q_prerope + (size_t)min(m0 + 16 * mi + l15, M - 1) * P.ldp + n0 + wn * 32 + 8 * g;
        const f32x4 a0 = *(const f32x4 *)qp, a1 = *(const f32x4 *)(qp + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) { qa[t] = a0[t]; qa[4 + t] = a1[t]; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = (add_q ? acc[mi][j][r] + qa[4 * j + r] : acc[mi][j][r]) + bias[4 * j + r];
          x[mi][4 * j + r] = v;
          sq = __builtin_fmaf(v, v, sq);  // (explicit: the thin-row kernel must round as this one does)
        }
      sq += __shfl_xor(sq, 16);
      sq += __shfl_xor(sq, 32);
      if (g == 0) part[(16 * mi + l15) * 4 + wn] = sq;
    }
    const int cih = wn * 32 + 8 * g;
    const float qos = P.q_out_scale == 0.0f ? 1.0f : P.q_out_scale;
    f32x4 rope0[MF], rope1[MF];
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      const float *rp = P.rope + (size_t)min(m0 + 16 * mi + l15, M - 1) * 128 + cih;
      rope0[mi] = *(const f32x4 *)rp;
      rope1[mi] = *(const f32x4 *)(rp + 4);
    }
    __syncthreads();
    const int head_col = n0 - (is_q ? 0 : hd);
    const bf16x8 s8 = *(const bf16x8 *)(nscale + cih);
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      const int m = m0 + 16 * mi + l15;
      if (m >= M) continue;
      const f32x4 p4 = *(const f32x4 *)(part + (16 * mi + l15) * 4);
      const float rrms = rsqrtf(__builtin_fmaf(p4[0] + p4[1] + p4[2] + p4[3], 1.0f / 128.0f, 1e-6f));
      float y[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) y[t] = x[mi][t] * rrms * (float)s8[t];
      if (is_q && P.q_prerope) {
        if (P.qpre_f32 == 2) {
          float *qp = (float *)P.q_prerope + (size_t)m * P.ldp + head_col + cih;
          *(f32x4 *)qp = f32x4{x[mi][0], x[mi][1], x[mi][2], x[mi][3]};
          *(f32x4 *)(qp + 4) = f32x4{x[mi][4], x[mi][5], x[mi][6], x[mi][7]};
        } else if (P.qpre_f32) {
          float *qp = (float *)P.q_prerope + (size_t)m * P.ldp + head_col + cih;
          *(f32x4 *)qp = f32x4{y[0], y[1], y[2], y[3]};
          *(f32x4 *)(qp + 4) = f32x4{y[4], y[5], y[6], y[7]};
        } else {
          *(uint4 *)((bf16 *)P.q_prerope + (size_t)m * P.ldp + head_col + cih) =
              make_uint4(ca_pack2(y[0], y[1]), ca_pack2(y[2], y[3]), ca_pack2(y[4], y[5]), ca_pack2(y[6], y[7]));
        }
      }
      const f32x4 r0 = rope0[mi], r1 = rope1[mi];
      const float cs[4] = {r0[0], r0[2], r1[0], r1[2]}, sn[4] = {r0[1], r0[3], r1[1], r1[3]};
      float z[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        z[2 * i] = __builtin_fmaf(cs[i], y[2 * i], -(sn[i] * y[2 * i + 1]));
        z[2 * i + 1] = __builtin_fmaf(sn[i], y[2 * i], cs[i] * y[2 * i + 1]);
      }
      if (is_q && qos != 1.0f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = z[i] * qos;
      }
      *(uint4 *)(outb + ((size_t)m * ldo + n0 + cih) * 2) =
          P.qk_f16 ? make_uint4(ca_pack2_f16(z[0], z[1]), ca_pack2_f16(z[2], z[3]), ca_pack2_f16(z[4], z[5]),
                                ca_pack2_f16(z[6], z[7]))
                   : make_uint4(ca_pack2(z[0], z[1]), ca_pack2(z[2], z[3]), ca_pack2(z[4], z[5]), ca_pack2(z[6], z[7]));
    }
    return;
  }
  if (epi == CA_EPI_QKV_NORM_ROPE) epi = CA_EPI_SPLIT_GELU;
  if (epi == CA_EPI_SPLIT_GELU) {
    if (n0 >= P.n_split) {
      epi = CA_EPI_GELU_TANH;
      outb = (char *)P.out2;
      ldo = P.ld2;
      col_shift = -P.n_split;
    } else {
      epi = CA_EPI_BIAS;
    }
  }
#pragma unroll
  for (int mi = 0; mi < MF; ++mi) {
    const int m = m0 + 16 * mi + l15;
    if (m >= M) continue;
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = acc[mi][t >> 2][t & 3] + bias[t];
    if (epi == CA_EPI_GELU_TANH) {
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = ca_gelu_tanh(v[t]);
    }
    if (epi == CA_EPI_GATE_RESIDUAL) {
      const float *gr = ca_gate_of(P, m) + nb;
      const f32x4 g0 = *(const f32x4 *)gr, g1 = *(const f32x4 *)(gr + 4);
      const float gt[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      if (P.out_f32) {
        const f32x4 *rp = (const f32x4 *)((const char *)P.resid + ((size_t)m * P.ldr + nb) * 4);
        const f32x4 ra = rp[0], rb = rp[1];
        const float res[8] = {ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]};
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = __builtin_fmaf(gt[t], v[t], res[t]);
      } else {
        const bf16x8 r8 = *(const bf16x8 *)((const char *)P.resid + ((size_t)m * P.ldr + nb) * 2);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = __builtin_fmaf(gt[t], v[t], (float)r8[t]);
      }
    }
    if (P.out_f32 && (epi == CA_EPI_GATE_RESIDUAL || epi == CA_EPI_BIAS)) {
      f32x4 *op = (f32x4 *)(outb + ((size_t)m * ldo + nb + col_shift) * 4);
      op[0] = f32x4{v[0], v[1], v[2], v[3]};
      op[1] = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      *(uint4 *)(outb + ((size_t)m * ldo + nb + col_shift) * 2) =
          make_uint4(ca_pack2(v[0], v[1]), ca_pack2(v[2], v[3]), ca_pack2(v[4], v[5]), ca_pack2(v[6], v[7]));
    }
  }
}

template <int MF, int NW>
int launch_thin_mf(const GemmLaunch &L, int groups, hipStream_t stream) {
  using TC = ThinCfg<MF, NW>;
  static std::atomic<unsigned long long> attr_done{0};  // one bit per device: the attribute is per device
  const unsigned long long dev_bit = ca_device_bit();
  if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
    hipError_t e = hipFuncSetAttribute((const void *)ca_gemm_thin_kernel<MF, NW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, TC::LDS);
    if (e != hipSuccess) {
      ca_set_error("ca_gemm_bf16: hipFuncSetAttribute(%d bytes LDS): %s", TC::LDS, hipGetErrorString(e));
      return CA_ERR_LAUNCH;
    }
    attr_done.fetch_or(dev_bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((ca_gemm_thin_kernel<MF, NW>), dim3((L.thin_nt[0] + L.thin_nt[1]) * (4 / NW), groups), dim3(64 * NW),
                     TC::LDS, stream, L);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ca_set_error("ca_gemm_bf16: thin-row launch failed: %s", hipGetErrorString(e));
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}

int launch_thin(const GemmLaunch &L, hipStream_t stream) {
  // (diagnostic builds, TIMING ONLY: CA_GEMM_KO_THIN=1 skips every thin-row launch -- wrong results; the ceiling of
  // anything that could still be done about those rows, tools/env_ab.py)
  static const int ko = ca_ab_env("CA_GEMM_KO_THIN", 0);
  if (ko) return CA_OK;
  int rows = 1;  // rows of the longest thin part
  bool heads = false;   // the fused QK-norm + RoPE epilogue needs a head's 128 columns in one workgroup
  for (int i = 0; i < CA_GEMM_MAX_PROBLEMS; ++i)
    if (L.thin_nt[i]) {
      rows = max(rows, L.p[i].M - L.thin_row0[i]);
      heads |= L.p[i].epilogue == CA_EPI_QKV_NORM_ROPE;
    }
  // one-wave workgroups (4 x the CUs streaming weight rows) for the rows of a batch's concept tokens; the 64-row form
  // -- the conditioning vectors of the modulation GEMM, longer thin parts -- keeps 4 waves (the weights stream once per
  // 64 rows either way, and its N is a million columns: the chip is full)
  // ... where the launch is short of workgroups (N / 128 <= 256), not for the modulation GEMM's million columns
  static const int narrow_env = ca_ab_env("CA_GEMM_THIN_NARROW", 1);
  const bool narrow = narrow_env && !heads && L.thin_nt[0] + L.thin_nt[1] <= 256;
  if (rows <= 32) return narrow ? launch_thin_mf<2, 1>(L, 1, stream) : launch_thin_mf<2, 4>(L, 1, stream);
  if (narrow) return launch_thin_mf<4, 1>(L, (rows + 63) / 64, stream);   // (flux-dev's 5 x 8 concept rows)
  return launch_thin_mf<4, 4>(L, (rows + 63) / 64, stream);
}

template <int NL, int NHI, bool FP8 = false>
int launch_pp(const GemmLaunch &L, int total_tiles, hipStream_t stream) {
  using C = PPCfg<NL, NHI>;
  static std::atomic<unsigned long long> attr_done{0};  // one bit per device: the attribute is per device
  const unsigned long long dev_bit = ca_device_bit();
  if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
    hipError_t e = hipFuncSetAttribute((const void *)ca_gemm_pp_kernel<NL, NHI, FP8>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      ca_set_error("ca_gemm_bf16: hipFuncSetAttribute(%d bytes LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
      return CA_ERR_LAUNCH;
    }
    attr_done.fetch_or(dev_bit, std::memory_order_release);  // idempotent: a race only repeats the call
  }
  const int grid = L.persist_tiles > 0 ? min(total_tiles, L.persist_tiles_grid) : total_tiles;
  hipLaunchKernelGGL((ca_gemm_pp_kernel<NL, NHI, FP8>), dim3(grid), dim3(512), C::LDS_BYTES, stream, L);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ca_set_error("ca_gemm_bf16: launch failed: %s", hipGetErrorString(e));
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}

template <int M_REP, int N_REP>
int launch(const GemmLaunch &L, int total_tiles, hipStream_t stream) {
  using C = Cfg<M_REP, N_REP>;
  static std::atomic<unsigned long long> attr_done{0};  // one bit per device: the attribute is per device
  const unsigned long long dev_bit = ca_device_bit();
  if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
    hipError_t e = hipFuncSetAttribute((const void *)ca_gemm_kernel<M_REP, N_REP>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      ca_set_error("ca_gemm_bf16: hipFuncSetAttribute(%d bytes LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
      return CA_ERR_LAUNCH;
    }
    attr_done.fetch_or(dev_bit, std::memory_order_release);  // idempotent: a race only repeats the call
  }
  hipLaunchKernelGGL((ca_gemm_kernel<M_REP, N_REP>), dim3(total_tiles), dim3(512), C::LDS_BYTES, stream, L);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ca_set_error("ca_gemm_bf16: launch failed: %s", hipGetErrorString(e));
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}

int tile_n_of(int tile) {
  switch (tile) {
    case CA_TILE_PP_256x256: return 256;
    case CA_TILE_PP_256x192: return 192;
    case CA_TILE_PP_256x128: return 128;
    case CA_TILE_256x256: return 256;
    case CA_TILE_256x192: return 192;
    case CA_TILE_256x128: return 128;
    case CA_TILE_256x64: return 64;
    default: return 0;
  }
}

// Row tiles per group of the ping-pong kernel's tile order.  The 32 workgroups an XCD runs at a time work on 32
// consecutive tiles of that order: a group_m x (32 / group_m) patch of the output, i.e. group_m A panels and 32 / group_m
// W panels through the XCD's L2, and rows of 32 / group_m x 512 contiguous output bytes in the round's store burst.
// It had been 8 everywhere (a square-ish patch: the fewest panels).  Round 4 measured the model's six grouped launches with
// 1 .. 32 (tools/gemm_group_m.py, profiles/r04_gemm_group_m.txt): at 5 items per forward every launch is 2-4 % faster with
// FEWER row tiles per group (wider rows of output per XCD and round), and which count is best depends on the number of
// column tiles -- 12 (N = 3072: proj, mlp.2, linear2): 1;  36-48 (qkv, mlp.0): 4;  84 (linear1): 2-3.  With one item per
// forward (17 row tiles) the choice matters less than 1 %, except that the N = 3072 launches prefer 2-4 to 1.
// CA_GEMM_GROUP_M overrides (A/B aid).  The order of the tiles changes nothing about any tile's result.
int pick_group_m(int nt, int mt, bool interleave) {
  static const int env = ca_ab_env("CA_GEMM_GROUP_M", 0);
  if (env > 0) return env;
  // (with the XCDs sharing each run of 256 tiles the choice is flat over 4 .. 8; one item per forward -- 17 row tiles --
  // prefers 4 for the widest launch: linear1 414 vs 423 us)
  if (interleave) return (nt <= 16 || (nt >= 64 && mt < 48)) ? 4 : 6;
  if (nt <= 16) return mt >= 48 ? 1 : 4;
  if (nt >= 64) return 3;
  return 4;
}

// Do the thin last-row tiles of a launch ride in the ping-pong walk (in the CUs the main tiles leave idle in their last
// round) instead of getting the thin-row kernel's own launch?  ONE predicate for the tile chooser and for gemm_impl
// (CA_GEMM_THIN_INWALK=0: never; round 3's behaviour, A/B aid).
bool thin_rides_in_walk(long main_all, long thin_all, int n_cu) {
  static const int inwalk_env = ca_ab_env("CA_GEMM_THIN_INWALK", 1);
  return inwalk_env && n_cu > 0 && thin_all > 0 && main_all % n_cu != 0 && main_all % n_cu + thin_all <= n_cu;
}

bool thin_kernel_enabled() {   // CA_GEMM_THIN_KERNEL=0: thin rows stay 256-column tiles of the ping-pong walk (A/B aid)
  static const int env = ca_ab_env("CA_GEMM_THIN_KERNEL", 1);
  return env != 0;
}

// Pick the tile that minimises (rounds over the CUs) x (time of one round).  Round times are
// per 48 K-steps, measured on MI355X (tools/bench_kernels.py): they only need to rank the choices.
int auto_tile(const ca_gemm_problem *p, int n) {
  const int n_cu = ca_cu_count() > 0 ? ca_cu_count() : 256;
  static const struct { int tile; double round_us; } cands[] = {
      {CA_TILE_PP_256x256, 80.0}, {CA_TILE_PP_256x192, 70.5}, {CA_TILE_PP_256x128, 51.0}, {CA_TILE_256x64, 35.0}};
  int best = 0;
  double best_cost = 1e300;
  for (const auto &c : cands) {
    const int bn = tile_n_of(c.tile);
    bool ok = true;
    double work = 0;  // tile-rounds weighted by K (a workgroup's time is proportional to its K loop)
    long tiles = 0;
    int kmax = 0, thin_k = 0;
    for (int i = 0; i < n; ++i) {
      if (p[i].N % bn) ok = false;
      if (p[i].epilogue == CA_EPI_SPLIT_GELU && p[i].n_split % bn) ok = false;
      if (p[i].epilogue == CA_EPI_QKV_NORM_ROPE && c.tile != CA_TILE_PP_256x256) ok = false;
      long mt = (p[i].M + 255) / 256;
      const int rem = p[i].M % 256;
      // a thin last row tile leaves the 256x256 ping-pong walk for the thin-row kernel (a fraction of a round); the
      // same predicate as gemm_impl's `own`, M <= 128 (no ping-pong tile at all) included
      if (c.tile == CA_TILE_PP_256x256 && rem > 0 && rem <= CA_GEMM_THIN_ROWS) {
        --mt;
        if (p[i].K > thin_k) thin_k = p[i].K;
      }
      const long t = mt * (p[i].N / bn);
      tiles += t;
      work += (double)t * p[i].K;
      if (p[i].K > kmax) kmax = p[i].K;
    }
    if (!ok) continue;
    // makespan estimate: at least one longest tile, at least the K-weighted work spread over the CUs
    double rounds = work / kmax / (double)n_cu;
    rounds = work == 0 ? 0.0 : rounds <= 1.0 ? 1.0 : (double)(long)(rounds + 0.999);
    // the thin-row launch behind the main one: measured 0.3 of a 256x256 round at its K (profiles/r02_remainder_probe.txt)
    // -- unless the thin tiles fit into the CUs the main tiles leave idle in their last round (gemm_impl's thin_fits):
    // then they ride in the walk and cost nothing
    double thin_cost = 0.3 * c.round_us * thin_k;
    if (c.tile == CA_TILE_PP_256x256 && thin_k > 0) {
      long main_t = 0, thin_t = 0;
      for (int i = 0; i < n; ++i) {
        const int rem = p[i].M % 256;
        const int thin = (rem > 0 && rem <= CA_GEMM_THIN_ROWS) ? 1 : 0;
        main_t += ((p[i].M + 255) / 256 - thin) * (long)(p[i].N / 256);
        thin_t += thin * (long)(p[i].N / 256);
      }
      if (thin_rides_in_walk(main_t, thin_t, n_cu)) thin_cost = 0.0;
      else if (!thin_kernel_enabled()) thin_cost = 0.75 * c.round_us * thin_k;   // full-width thin tiles after the walk
    }
    const double cost = rounds * c.round_us * kmax + thin_cost;
    if (cost < best_cost) {
      best_cost = cost;
      best = c.tile;
    }
    (void)tiles;
  }
  return best;
}

}  // namespace

extern "C" int ca_gemm_auto_tile(const ca_gemm_problem *problems, int32_t n_problems) {
  if (!problems || n_problems < 1 || n_problems > CA_GEMM_MAX_PROBLEMS) {
    ca_set_error("ca_gemm_auto_tile: n_problems=%d out of range", n_problems);
    return CA_ERR_ARG;
  }
  const int t = auto_tile(problems, n_problems);
  if (!t) {
    ca_set_error("ca_gemm_auto_tile: no tile width divides N (need N %% 64 == 0)");
    return CA_ERR_ARG;
  }
  return t;
}

namespace {

// shared argument checking + launch of ca_gemm_bf16 / ca_gemm_fp8 (FN = the entry point's name for messages)
int gemm_impl(const ca_gemm_problem *problems, int32_t n_problems, int32_t tile, ca_stream_t stream, bool fp8,
              const char *FN) {
  if (!problems || n_problems < 1 || n_problems > CA_GEMM_MAX_PROBLEMS) {
    ca_set_error("%s: n_problems=%d out of range [1,%d]", FN, n_problems, CA_GEMM_MAX_PROBLEMS);
    return CA_ERR_ARG;
  }
  if (fp8) {
    if (tile != CA_TILE_AUTO && tile != CA_TILE_PP_256x256) {
      ca_set_error("%s: only the 256x256 ping-pong tile has an fp8 variant (tile=%d)", FN, tile);
      return CA_ERR_ARG;
    }
    tile = CA_TILE_PP_256x256;
  }
  if (tile == CA_TILE_AUTO) tile = auto_tile(problems, n_problems);
  const int bn = tile_n_of(tile);
  const int kq = fp8 ? 128 : 64, ldq = fp8 ? 16 : 8, es = fp8 ? 1 : 2;
  if (!bn) {
    ca_set_error("%s: no tile configuration fits (tile=%d)", FN, tile);
    return CA_ERR_ARG;
  }
  GemmLaunch L = {};
  int total = 0;
  for (int i = 0; i < n_problems; ++i) {
    const ca_gemm_problem &p = problems[i];
    if (!p.A || !p.W || !p.out || p.M < 1 || p.N < 1 || p.K < 64) {
      ca_set_error("%s[%d]: null pointer or empty shape (M=%d N=%d K=%d)", FN, i, p.M, p.N, p.K);
      return CA_ERR_ARG;
    }
    if (p.K % kq || p.N % bn || p.lda % ldq || p.ldw % ldq || (p.ldc % 8 && !p.out_f32)) {
      ca_set_error("%s[%d]: need K%%%d==0, N%%%d==0, lda/ldw%%%d==0, ldc%%8==0 (M=%d N=%d K=%d lda=%d ldw=%d ldc=%d)",
                   FN, i, kq, bn, ldq, p.M, p.N, p.K, p.lda, p.ldw, p.ldc);
      return CA_ERR_ARG;
    }
    if (fp8 && (!p.a_scale || !p.w_scale || ((uintptr_t)p.w_scale & 15) || ((uintptr_t)p.a_scale & 3))) {
      ca_set_error("%s[%d]: fp8 operands need a_scale (fp32 [M]) and w_scale (fp32 [N], 16-byte aligned)", FN, i);
      return CA_ERR_ARG;
    }
    if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out | (uintptr_t)p.bias) & 15) {
      ca_set_error("%s[%d]: pointers must be 16-byte aligned", FN, i);
      return CA_ERR_ARG;
    }
    if (p.lda < p.K || p.ldw < p.K) {
      ca_set_error("%s[%d]: lda/ldw smaller than K", FN, i);
      return CA_ERR_ARG;
    }
    if ((uint64_t)p.M * p.lda * es >= (1ull << 32) || (uint64_t)p.N * p.ldw * es >= (1ull << 32)) {
      ca_set_error("%s[%d]: operand larger than 4 GiB", FN, i);
      return CA_ERR_ARG;
    }
    if (p.out_f32 && ((p.epilogue != CA_EPI_BIAS && p.epilogue != CA_EPI_GATE_RESIDUAL) || p.ldc % 4 ||
                      (p.epilogue == CA_EPI_GATE_RESIDUAL && p.ldr % 4))) {
      ca_set_error("%s[%d]: out_f32 needs the BIAS or GATE_RESIDUAL epilogue and ldc/ldr %% 4 == 0", FN, i);
      return CA_ERR_ARG;
    }
    switch (p.epilogue) {
      case CA_EPI_BIAS:
      case CA_EPI_GELU_TANH:
        if (p.ldc < p.N) { ca_set_error("%s[%d]: ldc < N", FN, i); return CA_ERR_ARG; }
        break;
      case CA_EPI_GATE_RESIDUAL:
        if (!p.resid || !p.gate || (p.ldr % 8 && !p.out_f32) || p.ldc < p.N || ((uintptr_t)p.resid & 15) ||
            ((uintptr_t)p.gate & 15) || ((uintptr_t)p.gate2 & 15) || (p.gate_rows < p.M && !p.gate2)) {
          ca_set_error("%s[%d]: GATE_RESIDUAL needs resid, gate (and gate2 when gate_rows < M), 16-byte aligned", FN, i);
          return CA_ERR_ARG;
        }
        if (p.gate_stride && (p.gate_stride % 4 || p.gate_stride < 0 || (p.gate_rows > 0 && p.gate_item_rows < 1) ||
                              (p.gate_rows < p.M && p.gate2_item_rows < 1))) {
          ca_set_error("%s[%d]: per-item gates need gate_stride %% 4 == 0 and gate_item_rows / gate2_item_rows >= 1", FN, i);
          return CA_ERR_ARG;
        }
        break;
      case CA_EPI_QKV_NORM_ROPE:
        // (N == n_split / 3: the q third alone -- the low-plane q projection of a captured layer, qpre_f32 = 3)
        if (tile != CA_TILE_PP_256x256 || p.n_split <= 0 || p.n_split % 768 ||
            (p.n_split > p.N && p.N != p.n_split / 3) || !p.norm_q ||
            !p.norm_k || !p.rope || (p.n_split < p.N && (!p.out2 || p.ld2 % 8 || p.ld2 < p.N - p.n_split)) ||
            p.ldc < (p.N < p.n_split ? p.N : p.n_split) ||
            (p.q_prerope && (p.ldp % (p.qpre_f32 ? 4 : 8) || p.ldp < p.n_split / 3 || p.qpre_f32 < 0 || p.qpre_f32 > 3)) ||
            (p.qpre_f32 == 3 && (!p.q_prerope || fp8)) ||
            (p.qk_f16 != 0 && p.qk_f16 != 1) ||
            (((uintptr_t)p.norm_q | (uintptr_t)p.norm_k | (uintptr_t)p.rope | (uintptr_t)p.q_prerope |
              (uintptr_t)p.out2) & 15)) {
          ca_set_error("%s[%d]: QKV_NORM_ROPE needs the 256x256 ping-pong tile, n_split = 3*heads*128 <= N, "
                       "norm_q/norm_k/rope (16-byte aligned) and out2 when N > n_split", FN, i);
          return CA_ERR_ARG;
        }
        break;
      case CA_EPI_SPLIT_GELU:
        if (!p.out2 || p.n_split <= 0 || p.n_split >= p.N || p.n_split % bn || p.ld2 % 8 ||
            ((uintptr_t)p.out2 & 15) || p.ldc < p.n_split || p.ld2 < p.N - p.n_split) {
          ca_set_error("%s[%d]: SPLIT_GELU needs out2 and 0 < n_split < N, n_split %% %d == 0", FN, i, bn);
          return CA_ERR_ARG;
        }
        break;
      default:
        ca_set_error("%s[%d]: unknown epilogue %d", FN, i, p.epilogue);
        return CA_ERR_ARG;
    }
    L.p[i] = p;
    L.mt[i] = (p.M + 255) / 256;
    L.nt[i] = p.N / bn;
    L.ntiles[i] = L.mt[i] * L.nt[i];
    total += L.ntiles[i];
  }
  if (n_problems == 1) {
    L.ntiles[1] = 0;
    L.mt[1] = L.nt[1] = 1;
    L.p[1] = L.p[0];
  }
  const bool thin_kernel_env = thin_kernel_enabled();
  // the 8 XCDs share each run of 256 consecutive tiles of the order (XCD x: its tiles 32 x .. 32 x + 31) instead of
  // owning one eighth of the order each: all of them stream the same group of A rows at a time (CA_GEMM_XCD_INTERLEAVE=0:
  // rounds 1-3's contiguous ranges)
  static const int xil_env = ca_ab_env("CA_GEMM_XCD_INTERLEAVE", 1);
  const int n_cu = ca_cu_count();
  // (the order's "round" is 256 tiles = 32 per XCD, which is what runs together only on a 256-CU part: elsewhere the
  // contiguous ranges, whose mapping does not assume a grid size)
  const int xil = xil_env && n_cu == 256;
  L.xcd_interleave = xil;
  L.group_m = pick_group_m(L.nt[0], L.mt[0], xil != 0);
  // Thin last row tiles (<= CA_GEMM_THIN_ROWS valid rows) under the bf16 256x256 tile normally get their own launch of
  // 32 x 128 tiles behind the main one (ca_gemm_thin_kernel).  But when the main tiles leave enough CUs idle in their
  // last round for every thin tile -- the one-item forward: 204 + 12 tiles on 256 CUs -- the thin tiles ride in the walk
  // (they are walked last, cost 0.75 of a full tile and finish inside the round that runs anyway): no second launch.
  // All three forms give the same bits per row (tests/test_kernels_gpu.py: thin rows in-walk / own kernel / full tile).
  int main_all = 0, thin_all = 0;
  for (int i = 0; i < n_problems; ++i) {
    const int rem = L.p[i].M % 256;
    const int thin = (rem > 0 && rem <= CA_GEMM_THIN_ROWS) ? 1 : 0;
    main_all += (L.mt[i] - thin) * L.nt[i];
    thin_all += thin * L.nt[i];
  }
  const bool thin_fits = thin_rides_in_walk(main_all, thin_all, n_cu);
  int total_pp = 0;
  for (int i = 0; i < CA_GEMM_MAX_PROBLEMS; ++i) {  // tile order of the ping-pong kernel: thin last row tiles go last
    const int rem = i < n_problems ? L.p[i].M % 256 : 0;
    const int thin = (rem > 0 && rem <= CA_GEMM_THIN_ROWS) ? 1 : 0;
    // under the bf16 256x256 tile: their own launch of 32 x 128 tiles instead (ca_gemm_thin_kernel), one grid row
    // per 32 rows
    const bool own = thin && !fp8 && tile == CA_TILE_PP_256x256 && thin_kernel_env && !thin_fits;
    L.mt_main[i] = i < n_problems ? L.mt[i] - thin : 1;
    L.ntiles_main[i] = i < n_problems ? L.mt_main[i] * L.nt[i] : 0;
    L.nthin[i] = (thin && !own) ? L.nt[i] : 0;
    L.thin_row0[i] = own ? (L.mt[i] - 1) * 256 : 0;
    L.thin_nt[i] = own ? L.p[i].N / THIN_N : 0;
    L.main_total += L.ntiles_main[i];
    total_pp += L.ntiles_main[i] + L.nthin[i];
  }
  const bool pp_tile = fp8 || tile == CA_TILE_PP_256x256 || tile == CA_TILE_PP_256x192 || tile == CA_TILE_PP_256x128;
  if (pp_tile) total = total_pp;  // (the simple kernel keeps every row tile)
  hipStream_t s = (hipStream_t)stream;
  {  // persistent walk of the tiles when there is more than one round of them (CA_GEMM_PERSIST=0 disables)
    static const int persist_env = ca_ab_env("CA_GEMM_PERSIST", 1);
    const int n = n_cu;
    if (persist_env && n > 0 && n % 8 == 0 && total > n) {
      L.persist_tiles = total;
      L.persist_tiles_grid = n;
    }
  }
  if (fp8) return launch_pp<2, 2, true>(L, total, s);
  if (tile == CA_TILE_PP_256x256 && L.thin_nt[0] + L.thin_nt[1] > 0) {
    if (total > 0) {
      const int rc = launch_pp<2, 2>(L, total, s);
      if (rc != CA_OK) return rc;
    }
    return launch_thin(L, s);
  }
  switch (tile) {
    case CA_TILE_PP_256x256: return launch_pp<2, 2>(L, total, s);
    case CA_TILE_PP_256x192: return launch_pp<2, 1>(L, total, s);
    case CA_TILE_PP_256x128: return launch_pp<1, 1>(L, total, s);
    case CA_TILE_256x256: return launch<8, 4>(L, total, s);
    case CA_TILE_256x192: return launch<8, 3>(L, total, s);
    case CA_TILE_256x128: return launch<8, 2>(L, total, s);
    default: return launch<8, 1>(L, total, s);
  }
}

}  // namespace

extern "C" int ca_gemm_bf16(const ca_gemm_problem *problems, int32_t n_problems, int32_t tile,
                            ca_stream_t stream) {
  return gemm_impl(problems, n_problems, tile, stream, false, "ca_gemm_bf16");
}

extern "C" int ca_gemm_fp8(const ca_gemm_problem *problems, int32_t n_problems, ca_stream_t stream) {
  return gemm_impl(problems, n_problems, CA_TILE_PP_256x256, stream, true, "ca_gemm_fp8");
}

#ifdef CA_GEMM_STAMP
extern "C" int ca_debug_read_gemm(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ca_gemm_dbg), sizeof(unsigned long long) * 8 * 2048);
}
#endif
