// The one-wave-per-SIMD attention kernel (own translation unit: it owns the AGPR file by hand, and is compiled with
// -mllvm -amdgpu-spill-vgpr-to-agpr=0 so that hipcc can never park a VGPR in one of those registers; build.py also
// checks the emitted code for compiler-written AGPR accesses).
#include <stdlib.h>

#include <atomic>

#include "ca_attn_common.h"

namespace {
using namespace ca_attn_detail;

// =================================================================================================================
// ca_attn4_kernel: the same attention for pre-scaled q (CA_ATTN_Q_PRESCALED) as ONE wave per SIMD.
//
// 4 waves x 64 query rows per workgroup (two 32-row query blocks per wave), the whole 512-entry register file per
// wave: O^T of both query blocks (128), the Q fragments (64) and 8-fragment rings for the K and V operands live in
// hand-owned AGPRs, the scores / probabilities, -reference and the packed P fragments in compiler-allocated VGPRs
// (an MFMA's C and D share a register class, A and B are free; v_exp_f32 cannot read AGPRs).  Every hot
// instruction is its own `asm volatile` statement: hipcc keeps volatile asm statements in program order, so the
// stream below IS the schedule -- a tile is 64 MFMAs with the LDS reads, the LDS-DMA pieces, the exponentials, the
// row-sum adds and the bf16 packs placed in the gaps between them by tools/gen_attn4_schedule.py (ca_attn4_sched.inc;
// the generator asserts its placement rules, build.py audits the emitted code for the hazards hipcc does not pad):
//     slots  0..31  S(t+1) = K(t+1) Q^T, the two query blocks' chains interleaved (no MFMA waits for its predecessor)
//     slots 32..63  O^T += V(t)^T P(t)^T
//     gaps   1..15  (odd) the wave's 4 + 4 LDS-DMA pieces of K(t+3) / V(t+1): buffer descriptor + scalar tile offset
//     gaps   2..32  (even) the K(t+1) key-block-1 and K(t+2) key-block-0 fragment reads; 17..49 (odd) the V(t) reads
//     gaps  16..63 and 0..15 of the NEXT iteration: one exponential of S(t+1) and one row-sum add each; the packs of
//                   P(t+1) behind the last P.V MFMA that reads the fragment they overwrite
// K/V tiles arrive by LDS-DMA into 3-slot rings (K three tiles ahead: its first fragments are read one tile before
// their MFMAs; V one tile ahead), one barrier per tile; the per-tile bookkeeping is a compare per matrix against the
// index of the next tile "event" (segment change, ragged / straddling / missing tile).  The softmax reference of a row
// is the maximum of tile 0 and is kept (see ca_attn_kernel) for as long as the row sums stay small: every third tile
// one compare looks at the wave's running sums, and a wave that finds one above 2^20 RE-REFERENCES in place
// (rereference(): every row's reference moves up by the exponent of its sum, O / l / the pending probabilities are
// scaled by the exact power of two; no key is visited twice).  What is left for the final check is a sum that
// overflowed between two compares (a score more than ~80 octaves above the running reference, inf / NaN inputs): the
// workgroup then recomputes its rows the classical way (running maximum, rescale per tile).
// The two waves of a SIMD in ca_attn_kernel run in lockstep (same program, one barrier per tile): per tile the matrix
// pipe idles while both exponentiate.  Here the single wave's own stream keeps it fed (2 265
// cycles per tile for 2 048 of MFMA issue).
#define CA_A4_HELPERS
#include "ca_attn4_sched.inc"
#undef CA_A4_HELPERS

namespace a4 {
constexpr int SLOTS = 3;
constexpr int V_BASE = SLOTS * TILE_BYTES;                 // V ring behind the K ring
constexpr int FLAG_OFF = 2 * SLOTS * TILE_BYTES;           // the "recompute" flag
constexpr int DUMP_OFF = FLAG_OFF + 1024;                  // 16 KiB nobody reads: where the tile loop's LDS-DMA pieces land
                                                           // when there is no tile for them (they then re-read a valid one)
constexpr int LDS_BYTES = DUMP_OFF + TILE_BYTES;
constexpr float L_LIMIT = 1267650600228229401496703205376.0f;   // 2^100: a finite row sum below this is exact enough to
                                                                // divide by (O <= l max|v| stays far from fp32's 2^128)
constexpr float REREF_ABOVE = 1048576.0f;                  // 2^20: running row sum that triggers the in-place re-reference
}  // namespace a4

// diagnostic counters (ca_attn_stats): [0] workgroups that went through the classical recomputation, [1] in-place
// re-reference events (per wave).  Written by the rare paths only.
__device__ unsigned long long ca_attn4_counters[2];

typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifdef CA_A4_STAMP   // diagnostic build only (tools/stamp_attn4.py): cycles per loop segment, per workgroup and wave
__device__ unsigned long long ca_a4_dbg[4 * 4 * 4096];
#define CA_A4_T(x) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory"); } while (0)
#else
#define CA_A4_T(x) do { } while (0)
#endif

__global__ __launch_bounds__(256, 1) void ca_attn4_kernel(const AttnLaunch L) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bid = blockIdx.x;
  int prob = 0;
  while (prob + 1 < L.n_problems && bid >= L.blk_end[prob]) ++prob;
  if (prob) bid -= L.blk_end[prob - 1];
  prob = __builtin_amdgcn_readfirstlane(prob);
  const int nqb = __builtin_amdgcn_readfirstlane(L.nqb[prob]);
  const int xg = bid & 7, idx = bid >> 3;
  const int head = xg + 8 * (idx / nqb);
  const int qb_wg = idx % nqb;
  if (head >= L.num_heads) return;
  // the descriptor's fields as scalars (indexing the by-value argument inside the loop would make hipcc keep a scratch
  // copy of it, and scalar loads inside the tile loop would share the LDS reads' counter)
  const ca_attn_problem &P = L.p[prob];
  // (through readfirstlane: the descriptor is indexed with a run-time problem number, and what hipcc cannot prove
  // wave-uniform it will not put into the SGPR operands of the tile loop's asm statements)
  auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  const int nq = uni(P.nq), n0 = uni(P.n0), nkeys = uni(P.n0 + P.n1), nq0 = uni(P.nq0);
  const int ldkv = uni(P.ldkv), ldq = uni(P.ldq), ldo = uni(P.ldo), ldo32 = uni(P.ldo32);
  const bf16 *q_a = (const bf16 *)P.q, *q_b = (const bf16 *)P.q1;
  bf16 *o_a = (bf16 *)P.out, *o_b = (bf16 *)P.out1;
  float *o32 = P.out_f32;

  const int nt = (nkeys + KV_TILE - 1) / KV_TILE;
  const bool ragged = (nkeys & (KV_TILE - 1)) != 0;
  const int nt_full = ragged ? nt - 1 : nt;

  const int h = lane >> 5, ql = lane & 31;
  const int qrow0 = qb_wg * 256 + wave * 64;
  const bool active = qrow0 < nq;  // wave-uniform

  asm volatile("" ::: CA_A4_AGPR_CLOBBERS);   // the AGPR file is ours: makes the kernel descriptor allocate it

  // ---- staging (LDS-DMA), 4 pieces of 1 KiB per wave and matrix; same images and source swizzles as ca_attn_kernel
  const int st_row = lane >> 4, st_cp = lane & 15;
  const bf16 *k0p = (const bf16 *)P.k0 + head * 128, *v0p = (const bf16 *)P.v0 + head * 128;
  const bf16 *k1p = (const bf16 *)P.k1 + head * 128, *v1p = (const bf16 *)P.v1 + head * 128;
  uint32_t koff[4], voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 4 * (wave * 4 + j) + st_row;
    koff[j] = ((uint32_t)r * (uint32_t)ldkv + ((st_cp ^ (r & 15)) << 3)) * 2u;
    voff[j] = ((uint32_t)r * (uint32_t)ldkv + ((st_cp ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 3)) * 2u;
  }
  auto stage = [&](int tile, int slot, bool is_v) {
    char *dst = smem + (is_v ? a4::V_BASE : 0) + slot * TILE_BYTES;
    const bf16 *p0 = is_v ? v0p : k0p, *p1 = is_v ? v1p : k1p;
    const int lo = tile * KV_TILE;
    const bool in0 = lo + KV_TILE <= n0, in1 = lo >= n0 && lo + KV_TILE <= nkeys;
    if (in0 || in1) {
      const bf16 *base = (in0 ? p0 : p1) + (size_t)(in0 ? lo : lo - n0) * ldkv;
#pragma unroll
      for (int j = 0; j < 4; ++j) ca_glds16_asm_s(base, is_v ? voff[j] : koff[j], dst + (wave * 4 + j) * 1024);
      return;
    }
    // (rare path, inlined at every tile event of the loop: its per-lane values are derived from a laundered lane id so
    // that hipcc cannot hoist them out of the tile loop, where they would cost registers the stream needs)
    int lane_l = lane;
    asm volatile("" : "+v"(lane_l));
    const int st_row = lane_l >> 4, st_cp = lane_l & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * (wave * 4 + j) + st_row;
      const int kk = min(lo + r, nkeys - 1);
      const bool s0 = kk < n0;
      const size_t ro = (size_t)(s0 ? kk : kk - n0) * ldkv;
      const int ch = is_v ? (st_cp ^ (((r & 3) << 2) | ((r >> 2) & 3))) : (st_cp ^ (r & 15));
      ca_glds16_asm((s0 ? p0 : p1) + ro + (ch << 3), dst + (wave * 4 + j) * 1024);
    }
  };
  // The tile loop issues its LDS-DMA from inside the instruction stream (one 1-KiB piece per odd MFMA gap 1..15, no
  // branch): buffer_load ... lds with a buffer descriptor per matrix, the tile's byte offset as the scalar offset and
  // the lane offsets koff / voff.  Descriptor of a key segment: base = the address key index 0 WOULD have (segment 1:
  // k1 - n0 rows), so that the scalar offset of tile i is i x 64 rows x row bytes in both segments and advances by one
  // s_add per tile; the record count is unlimited (every tile sent this way lies inside its segment).  A tile that
  // straddles the two segments or is ragged is staged the general way (stage()), a tile past the end not at all: for
  // those the stream's pieces get the NULL descriptor (0 records: nothing is read) and land in the dump page.  The
  // per-tile bookkeeping is a compare against the index of the next tile at which any of this changes (EVK / EVV);
  // everything else happens in the rare k_event / v_event.
  const uint32_t koff0 = koff[0], koff1 = koff[1], koff2 = koff[2], koff3 = koff[3];
  const uint32_t voff0 = voff[0], voff1 = voff[1], voff2 = voff[2], voff3 = voff[3];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(ca_lptr)smem;
  auto uni32 = [&](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
  auto make_desc = [&](const bf16 *key0_addr, uint32_t records) {
    const uint64_t b = (uint64_t)(uintptr_t)key0_addr;
    return i32x4{(int)uni32((uint32_t)b), (int)(uni32((uint32_t)(b >> 32)) & 0xffffu), (int)records, 0x00020000};
  };
  const uint32_t row_bytes = (uint32_t)ldkv * 2u, tile_step = 64u * row_bytes;
  const i32x4 dsk0 = make_desc(k0p, 0xffffffffu), dsv0 = make_desc(v0p, 0xffffffffu);
  const i32x4 dsk1 = make_desc(k1p - (size_t)n0 * ldkv, 0xffffffffu), dsv1 = make_desc(v1p - (size_t)n0 * ldkv, 0xffffffffu);
  const i32x4 ds_null = make_desc(k0p, 0u);
  const int t_straddle = (n0 & (KV_TILE - 1)) && n0 < nkeys ? n0 / KV_TILE : -1;   // the tile with keys of both segments
  const int t_seg1 = (n0 + KV_TILE - 1) / KV_TILE;                                  // first tile inside segment 1
  const uint32_t lw = lds0 + wave * 4096, lw_dump = lds0 + a4::DUMP_OFF + wave * 4096;
  i32x4 DSK = ds_null, DSV = ds_null;
  uint32_t SOK = 0, SOV = 0, LWK = lw, LWV = lw;
  int EVK = 3, EVV = 1;     // the first iteration's tiles are events (they set the state up)
  // tile `tile` is about to be sent into ring slot `slot` (dst_off = the slot's byte offset as the stream's M0 immediate
  // has it): sets descriptor / offset / destination for the stream's pieces, stages the tile here if it is not a
  // full tile inside one segment, and returns the next tile index at which to come back
  auto dma_event = [&](int tile, int slot, bool is_v, uint32_t dst_off, i32x4 &DS, uint32_t &SO, uint32_t &LW) -> int {
    const bool fast = tile < nt_full && tile != t_straddle;
    if (!fast) {
      if (tile < nt) stage(tile, slot, is_v);
      DS = ds_null, SO = 0, LW = lw_dump - dst_off;
      return tile + 1;
    }
    const bool seg1 = tile >= t_seg1;
    DS = is_v ? (seg1 ? dsv1 : dsv0) : (seg1 ? dsk1 : dsk0);
    SO = (uint32_t)tile * tile_step;
    LW = lw;
    int next = nt_full;
    if (t_straddle > tile) next = min(next, t_straddle);
    if (t_seg1 > tile) next = min(next, t_seg1);
    return next;
  };
#define CA_A4_BOOKKEEPING(R_)                                                                                          \
  do {                                                                                                                 \
    if (__builtin_expect(t + 3 == EVK, 0))                                                                        \
      EVK = dma_event(t + 3, (R_), false, (uint32_t)((R_) * TILE_BYTES), DSK, SOK, LWK);                               \
    if (__builtin_expect(t + 1 == EVV, 0))                                                                        \
      EVV = dma_event(t + 1, ((R_) + 1) % 3, true, (uint32_t)(a4::V_BASE + (((R_) + 1) % 3) * TILE_BYTES), DSV, SOV, LWV); \
  } while (0)
#define CA_A4_ADVANCE() do { SOK += tile_step; SOV += tile_step; } while (0)
  // one compare per three tiles: has any partial row sum of this wave left the comfortable range?
  const bool reref_on = active && !(L.flags & 1);
#define CA_A4_REREF_CHECK()                                                                                            \
  do {                                                                                                                 \
    /* (the sums are >= 0, so their bit patterns order like the values; a NaN's pattern is above every finite one) */ \
    if (reref_on && __builtin_expect(__builtin_amdgcn_ballot_w64(                                                      \
                        max(max(__float_as_int(l0), __float_as_int(l0b)),                                              \
                            max(__float_as_int(l1), __float_as_int(l1b))) > __float_as_int(a4::REREF_ABOVE)) != 0, 0)) \
      rereference();                                                                                                   \
  } while (0)
  auto stage_pieces = [&](uint32_t kdst_off, uint32_t vdst_off) {   // the same 8 pieces from a wave that computes nothing
    auto piece = [&](const i32x4 &ds, uint32_t so, uint32_t off, uint32_t dst) {   // (its query rows do not exist)
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                   : : "v"(off), "s"(ds), "s"(so), "s"(dst) : "memory");
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      piece(DSK, SOK, koff[j], LWK + kdst_off + 1024 * j);
      piece(DSV, SOV, voff[j], LWV + vdst_off + 1024 * j);
    }
  };
  auto drain_and_barrier = [&]() {   // this wave's DMA has landed, its LDS reads have returned; then everyone's
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // ---- fragment addresses (bytes): K relative to the K ring, V with the V ring's base included
  const uint32_t k_lane = ql * 256 + (((h ^ (ql & 15)) & 15) << 4);
  const uint32_t ka0 = k_lane, ka1 = k_lane ^ (1 << 5), ka2 = k_lane ^ (2 << 5), ka3 = k_lane ^ (3 << 5),
                 ka4 = k_lane ^ (4 << 5), ka5 = k_lane ^ (5 << 5), ka6 = k_lane ^ (6 << 5), ka7 = k_lane ^ (7 << 5);
  const int qq = (lane & 15) >> 2;
  const int c_lane = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
  uint32_t v_lane[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int x = (qq << 2) | ((2 * jj + h) & 3);
    v_lane[jj] = a4::V_BASE + (4 * h + qq) * 256 + (((c_lane ^ x) & 15) << 4) + 8 * (lane & 1);
  }
  const uint32_t va00 = v_lane[0], va01 = v_lane[0] ^ (1 << 6), va02 = v_lane[0] ^ (2 << 6), va03 = v_lane[0] ^ (3 << 6);
  const uint32_t va10 = v_lane[1], va11 = v_lane[1] ^ (1 << 6), va12 = v_lane[1] ^ (2 << 6), va13 = v_lane[1] ^ (3 << 6);

  f32x16 S00, S01, S10, S11, NM0, NM1;
  i32x4 P000, P001, P010, P011, P100, P101, P110, P111;   // P[kb][qb][sk]
  float l0 = 0.f, l1 = 0.f, l0b = 0.f, l1b = 0.f, m0 = -1e30f, m1 = -1e30f;   // row sum of block b = lb + lbb
#pragma unroll
  for (int r = 0; r < 16; ++r) NM0[r] = NM1[r] = 0.f, S00[r] = S01[r] = S10[r] = S11[r] = 0.f;
  P000 = P001 = P010 = P011 = P100 = P101 = P110 = P111 = i32x4{0, 0, 0, 0};


  // key of S[kb][*][r] in tile t: 64 t + 32 kb + (r&3) + 8 (r>>2) + 4 h
  auto mask_tail = [&](int t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = t * KV_TILE + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (key >= nkeys) S00[r] = S01[r] = -INFINITY;
      if (key + 32 >= nkeys) S10[r] = S11[r] = -INFINITY;
    }
  };
  auto pack_one = [&](i32x4 &Pf, const f32x16 &Sv, int base) {
#pragma unroll
    for (int j = 0; j < 4; ++j) Pf[j] = (int)ca_pack2(Sv[base + 2 * j], Sv[base + 2 * j + 1]);
  };
  auto pack_all = [&]() {
    pack_one(P000, S00, 0), pack_one(P001, S00, 8), pack_one(P010, S01, 0), pack_one(P011, S01, 8);
    pack_one(P100, S10, 0), pack_one(P101, S10, 8), pack_one(P110, S11, 0), pack_one(P111, S11, 8);
  };
  // S holds scores minus the reference (or raw scores with *mref = the reference to subtract): P = exp2, row sums
  auto exp_sum = [&](float sub0, float sub1) {
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S00[r] = __builtin_amdgcn_exp2f(S00[r] - sub0), S10[r] = __builtin_amdgcn_exp2f(S10[r] - sub0);
      S01[r] = __builtin_amdgcn_exp2f(S01[r] - sub1), S11[r] = __builtin_amdgcn_exp2f(S11[r] - sub1);
      r0 += S00[r] + S10[r];
      r1 += S01[r] + S11[r];
    }
    l0 += r0;
    l1 += r1;
  };
  auto row_max = [&](float &x0, float &x1) {
    x0 = S00[0], x1 = S01[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      x0 = fmaxf(x0, fmaxf(S00[r], S10[r]));
      x1 = fmaxf(x1, fmaxf(S01[r], S11[r]));
    }
    x0 = fmaxf(x0, __shfl_xor(x0, 32));
    x1 = fmaxf(x1, __shfl_xor(x1, 32));
  };
  // In-place re-reference (rare; top of an iteration, see the state described at "tile 0" below): row i's reference
  // goes up by e_i = floor(log2(row sum so far)) >= 0.  Everything that carries the old reference is scaled by the
  // exact power of two 2^-e_i -- O^T, the four partial sums, the exponentiated scores of the tile in flight (S00,
  // S01, S10; their packed fragments are re-packed) -- or shifted by e_i (S11: still raw score - reference; NM).
  // Per-row and a function of the row's own keys only, so an item's bits do not depend on the launch it shares.
  auto rereference = [&]() {
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(S00), "+v"(S01), "+v"(S10), "+v"(S11));   // MFMA results -> VALU
    float t0 = l0 + l0b, t1 = l1 + l1b;
    t0 += __shfl_xor(t0, 32), t1 += __shfl_xor(t1, 32);
    // exponent field (t >= 0; inf / NaN -> 255: clamped, the final check then sends the workgroup to the recomputation)
    const int e0 = min(max(((__float_as_int(t0) >> 23) & 0xff) - 127, 0), 126);
    const int e1 = min(max(((__float_as_int(t1) >> 23) & 0xff) - 127, 0), 126);
    const float f0 = __int_as_float((127 - e0) << 23), f1 = __int_as_float((127 - e1) << 23);
    const float d0 = (float)e0, d1 = (float)e1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S00[r] *= f0, S10[r] *= f0, S01[r] *= f1, S11[r] -= d1;
      NM0[r] -= d0, NM1[r] -= d1;
    }
    l0 *= f0, l0b *= f0, l1 *= f1, l1b *= f1;
    pack_one(P000, S00, 0), pack_one(P001, S00, 8), pack_one(P010, S01, 0), pack_one(P011, S01, 8);
    pack_one(P100, S10, 0);
    CA_A4_SCALE_O(f0, f1);
    asm volatile("s_nop 7" : "+v"(NM0), "+v"(NM1), "+v"(P000), "+v"(P001), "+v"(P010), "+v"(P011), "+v"(P100));
    if (lane == 0) atomicAdd(&ca_attn4_counters[1], 1ull);
  };
  auto set_reference = [&](float r0, float r1) {
    m0 = r0, m1 = r1;
#pragma unroll
    for (int r = 0; r < 16; ++r) NM0[r] = -r0, NM1[r] = -r1;
    asm volatile("s_nop 7" : "+v"(NM0), "+v"(NM1));   // VALU write -> MFMA C operand (the asm MFMAs are opaque to hipcc)
  };

  // ---- prologue: K(0), V(0), K(1), K(2) are requested first, the Q rows behind them (one memory round trip for both)
  stage(0, 0, false);
  stage(0, 0, true);
  if (nt > 1) stage(1, 1, false);
  if (nt > 2) stage(2, 2, false);
  // Q fragments of both query blocks -> AGPRs
  {
    uint32_t qw[2][8][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int qrow = min(qrow0 + 32 * b + ql, nq - 1);
      const bf16 *qp = (qrow < nq0 ? q_a + (size_t)qrow * ldq : q_b + (size_t)(qrow - nq0) * ldq) + head * 128 + h * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const uint4 v = *(const uint4 *)(qp + ks * 16);
        qw[b][ks][0] = v.x, qw[b][ks][1] = v.y, qw[b][ks][2] = v.z, qw[b][ks][3] = v.w;
      }
    }
    CA_A4_WRITE_Q(qw);
  }
  CA_A4_ZERO_O();

  drain_and_barrier();

#ifdef CA_A4_PLAIN   // bisecting aid: every tile the plain way (no pipelined stream), same prologue and epilogue
  if (active) {
    CA_A4_QK_PLAIN_ZERO(0u);
    if (nt_full == 0) mask_tail(0);
    float x0, x1;
    row_max(x0, x1);
    set_reference(x0, x1);
    exp_sum(x0, x1);
    pack_all();
    CA_A4_PV_PLAIN(0u);
  }
  for (int t = 1; t < nt; ++t) {
    drain_and_barrier();
    stage(t, 0, false);
    stage(t, 0, true);
    drain_and_barrier();
    if (active) {
      CA_A4_QK_PLAIN_NEGM(0u);
      if (ragged && t == nt - 1) mask_tail(t);
      exp_sum(0.f, 0.f);
      pack_all();
      CA_A4_PV_PLAIN(0u);
    }
  }
#else
  // ---- tile 0 sets the reference (the only tile whose maximum is computed)
  if (active) {
    CA_A4_QK_PLAIN_ZERO(0u);
    if (nt_full == 0) mask_tail(0);
    float x0, x1;
    row_max(x0, x1);
    set_reference(x0, x1);
    // ... and leaves the tile in the state every iteration of the loop below starts from (the loop's softmax runs one
    // exponential per MFMA gap and laps into the next iteration, ca_attn4_sched.inc): S00, S01, S10 exponentiated,
    // summed (except S10[14], S10[15]: their adds ride in the next iteration's first two gaps) and packed except P101;
    // S11 still raw (score - reference), P110 / P111 not packed.  finish_pending() completes that state behind the loop.
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S00[r] = __builtin_amdgcn_exp2f(S00[r] - x0), S10[r] = __builtin_amdgcn_exp2f(S10[r] - x0);
      S01[r] = __builtin_amdgcn_exp2f(S01[r] - x1);
      r0 += S00[r] + (r < 14 ? S10[r] : 0.f);
      r1 += S01[r];
      S11[r] = S11[r] - x1;
    }
    l0 += r0, l1 += r1;
    pack_one(P000, S00, 0), pack_one(P001, S00, 8), pack_one(P010, S01, 0), pack_one(P011, S01, 8);
    pack_one(P100, S10, 0);
  }
  drain_and_barrier();   // every wave is done with K(0) before iteration 0 lets the DMA overwrite its slot

  // ---- pipelined tiles: iteration t = K(t+1) Q^T + softmax(t+1) beside O^T += V(t)^T P(t)^T.  The ring slots of an
  // iteration are instruction immediates (t % 3), so the loop body is three iterations in a row.
  const int T = nt_full > 0 ? nt_full - 1 : 0;
  if (T > 0 && active) CA_A4_PRELOAD_K0((uint32_t)TILE_BYTES);
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, acc0 = 0, acc1 = 0, acc2 = 0;
  (void)ts0, (void)ts1, (void)ts2, (void)ts3, (void)acc0, (void)acc1, (void)acc2;
#ifdef CA_A4_STAMP
#define CA_A4_ACC() do { acc0 += ts1 - ts0; acc1 += ts2 - ts1; acc2 += ts3 - ts2; } while (0)
#else
#define CA_A4_ACC() do { } while (0)
#endif
  int t = 0;
  while (t + 3 <= T) {
    { constexpr int R = 0; (void)R; CA_A4_T(ts0); CA_A4_BOOKKEEPING(0); CA_A4_REREF_CHECK(); CA_A4_T(ts1);
      if (active) {
#define CA_A4_SCHEDULE
#include "ca_attn4_sched.inc"
#undef CA_A4_SCHEDULE
      } else {   // a wave whose query rows do not exist: its share of the staging only
        stage_pieces((uint32_t)(0 * TILE_BYTES), (uint32_t)(a4::V_BASE + ((0 + 1) % 3) * TILE_BYTES));
      }
      CA_A4_ADVANCE(); CA_A4_T(ts2); drain_and_barrier(); CA_A4_T(ts3); CA_A4_ACC(); ++t; }
    { constexpr int R = 1; (void)R; CA_A4_T(ts0); CA_A4_BOOKKEEPING(1); CA_A4_T(ts1);
      if (active) {
#define CA_A4_SCHEDULE
#include "ca_attn4_sched.inc"
#undef CA_A4_SCHEDULE
      } else {   // a wave whose query rows do not exist: its share of the staging only
        stage_pieces((uint32_t)(1 * TILE_BYTES), (uint32_t)(a4::V_BASE + ((1 + 1) % 3) * TILE_BYTES));
      }
      CA_A4_ADVANCE(); CA_A4_T(ts2); drain_and_barrier(); CA_A4_T(ts3); CA_A4_ACC(); ++t; }
    { constexpr int R = 2; (void)R; CA_A4_T(ts0); CA_A4_BOOKKEEPING(2); CA_A4_T(ts1);
      if (active) {
#define CA_A4_SCHEDULE
#include "ca_attn4_sched.inc"
#undef CA_A4_SCHEDULE
      } else {   // a wave whose query rows do not exist: its share of the staging only
        stage_pieces((uint32_t)(2 * TILE_BYTES), (uint32_t)(a4::V_BASE + ((2 + 1) % 3) * TILE_BYTES));
      }
      CA_A4_ADVANCE(); CA_A4_T(ts2); drain_and_barrier(); CA_A4_T(ts3); CA_A4_ACC(); ++t; }
  }
  if (t < T) {   // (t % 3 == 0 here) one or two iterations left
    { constexpr int R = 0; (void)R; CA_A4_T(ts0); CA_A4_BOOKKEEPING(0); CA_A4_T(ts1);
      if (active) {
#define CA_A4_SCHEDULE
#include "ca_attn4_sched.inc"
#undef CA_A4_SCHEDULE
      } else {   // a wave whose query rows do not exist: its share of the staging only
        stage_pieces((uint32_t)(0 * TILE_BYTES), (uint32_t)(a4::V_BASE + ((0 + 1) % 3) * TILE_BYTES));
      }
      CA_A4_ADVANCE(); CA_A4_T(ts2); drain_and_barrier(); CA_A4_T(ts3); CA_A4_ACC(); ++t; }
    if (t < T) { constexpr int R = 1; (void)R; CA_A4_T(ts0); CA_A4_BOOKKEEPING(1); CA_A4_T(ts1);
      if (active) {
#define CA_A4_SCHEDULE
#include "ca_attn4_sched.inc"
#undef CA_A4_SCHEDULE
      } else {   // a wave whose query rows do not exist: its share of the staging only
        stage_pieces((uint32_t)(1 * TILE_BYTES), (uint32_t)(a4::V_BASE + ((1 + 1) % 3) * TILE_BYTES));
      }
      CA_A4_ADVANCE(); CA_A4_T(ts2); drain_and_barrier(); CA_A4_T(ts3); CA_A4_ACC(); ++t; }
  }
#ifdef CA_A4_STAMP
  if (lane == 0 && blockIdx.x < 4096) {
    unsigned long long *d = ca_a4_dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
    d[0] = acc0, d[1] = acc1, d[2] = acc2, d[3] = (unsigned long long)T;
  }
#endif
  // softmax done through tile T, P.V through tile T-1
  if (active) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(S00), "+v"(S01), "+v"(S10), "+v"(S11));
    // finish_pending: what the next iteration would have done in its first gaps (see the first tile above)
    {
      l0 += S10[14] + S10[15];
      float r1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        S11[r] = __builtin_amdgcn_exp2f(S11[r]);
        r1 += S11[r];
      }
      l1 += r1;
      pack_one(P101, S10, 8), pack_one(P110, S11, 0), pack_one(P111, S11, 8);
    }
    asm volatile("s_nop 3" : "+v"(P101), "+v"(P110), "+v"(P111));
    CA_A4_PV_PLAIN((uint32_t)((T % 3) * TILE_BYTES));
  }
  if (ragged && nt > 1) {   // tile nt - 1 = T + 1: its K is staged (prologue or iteration T - 2), its V is not
    stage(T + 1, (T + 1) % 3, true);
    drain_and_barrier();
    if (active) {
      CA_A4_QK_PLAIN_NEGM((uint32_t)(((T + 1) % 3) * TILE_BYTES));
      mask_tail(T + 1);
      exp_sum(0.f, 0.f);
      pack_all();
      CA_A4_PV_PLAIN((uint32_t)(((T + 1) % 3) * TILE_BYTES));
    }
  }

#endif  // CA_A4_PLAIN
  // ---- did any row leave the safe range?  (workgroup-uniform decision: the recomputation stages tiles together)
  int *flag = (int *)(smem + a4::FLAG_OFF);
  if (tid == 0) *flag = 0;
  drain_and_barrier();
  l0 += l0b, l1 += l1b;
  l0b = l1b = 0.f;
  if (active && __builtin_amdgcn_ballot_w64(!(l0 <= a4::L_LIMIT) || !(l1 <= a4::L_LIMIT)) != 0 && lane == 0) *flag = 1;
  drain_and_barrier();
  if (*flag) {
    // classical online softmax, one tile at a time, nothing overlapped (rare: a row sum that overflowed between two
    // re-reference checks, or inf / NaN inputs)
    if (tid == 0) atomicAdd(&ca_attn4_counters[0], 1ull);
    CA_A4_ZERO_O();
    l0 = l1 = 0.f;
    m0 = m1 = -1e30f;
    for (int t = 0; t < nt; ++t) {
      drain_and_barrier();
      stage(t, 0, false);
      stage(t, 0, true);
      drain_and_barrier();
      if (active) {
        CA_A4_QK_PLAIN_ZERO(0u);
        if (ragged && t == nt - 1) mask_tail(t);
        float x0, x1;
        row_max(x0, x1);
        const float n0_ = fmaxf(m0, x0), n1_ = fmaxf(m1, x1);
        const float al0 = __builtin_amdgcn_exp2f(m0 - n0_), al1 = __builtin_amdgcn_exp2f(m1 - n1_);
        CA_A4_SCALE_O(al0, al1);
        l0 *= al0, l1 *= al1;
        m0 = n0_, m1 = n1_;
        exp_sum(m0, m1);
        pack_all();
        CA_A4_PV_PLAIN(0u);
      }
    }
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l, as ca_attn_kernel (16-byte stores via v_permlane32_swap)
  if (active) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f32x16 of1[4];   // one query block at a time: 64 registers
      if (b == 0) CA_A4_READ_O0(of1); else CA_A4_READ_O1(of1);
      const float lr = b ? l1 : l0;
      const float inv = 1.0f / (lr + __shfl_xor(lr, 32));
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const int row = qrow0 + 32 * b + ql;
      const int orow = min(row, nq - 1);
      bf16 *op = (orow < nq0 ? o_a + (size_t)orow * ldo : o_b + (size_t)(orow - nq0) * ldo) + head * 128 + 8 * h;
      const bool row_ok = row < nq;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          const f32x16 &o = of1[db];
          const uint32_t ax = ca_pack2(o[4 * g] * inv, o[4 * g + 1] * inv);
          const uint32_t ay = ca_pack2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
          const uint32_t bx = ca_pack2(o[4 * g + 4] * inv, o[4 * g + 5] * inv);
          const uint32_t by = ca_pack2(o[4 * g + 6] * inv, o[4 * g + 7] * inv);
          const u32x2 sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
          const u32x2 sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
          if (row_ok) *(uint4 *)(op + 32 * db + 8 * g) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      if (row_ok && o32) {
        float *fp = o32 + (size_t)row * ldo32 + head * 128 + 4 * h;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x16 &o = of1[db];
            *(f32x4 *)(fp + 32 * db + 8 * g) =
                f32x4{o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
          }
      }
    }
  }
}


}  // namespace

#ifdef CA_A4_STAMP
extern "C" int ca_debug_read_attn4(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ca_a4_dbg), sizeof(unsigned long long) * 4 * 4 * 4096);
}
#endif

int ca_attn4_read_counters(unsigned long long *out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(ca_attn4_counters), 2 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) {
    const unsigned long long z[2] = {0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(ca_attn4_counters), z, sizeof(z));
  }
  if (e != hipSuccess) {
    ca_set_error("ca_attn_stats: %s", hipGetErrorString(e));
    return CA_ERR_LAUNCH;
  }
  return CA_OK;
}

int ca_attn4_launch(const AttnLaunch &L, int total, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  const unsigned long long dev_bit = ca_device_bit();
  if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
    const hipError_t e = hipFuncSetAttribute((const void *)ca_attn4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             a4::LDS_BYTES);
    if (e != hipSuccess) {
      ca_set_error("ca_attn_fwd_bf16: hipFuncSetAttribute(ca_attn4_kernel): %s", hipGetErrorString(e));
      return CA_ERR_LAUNCH;
    }
    attr_done.fetch_or(dev_bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(ca_attn4_kernel, dim3(total), dim3(256), a4::LDS_BYTES, stream, L);
  return CA_OK;
}
