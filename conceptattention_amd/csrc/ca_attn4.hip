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
// one compare looks at the wave's running sums, and a wave that finds one above 2^64 RE-REFERENCES in place
// (rereference(): every row's reference moves up by the exponent of its sum, O / l / the pending probabilities are
// scaled by the exact power of two; no key is visited twice).  What is left for the final check is a sum that
// passed 2^100 or overflowed between two compares (a score more than ~36 octaves above a running sum that was just
// short of 2^64, i.e. ~100 octaves above the row's current reference within three tiles; inf / NaN inputs): the
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
constexpr float L_LIMIT_R3 = 1152921504606846976.0f;       // 2^60: round 3's limit (CA_ATTN_LIMIT60=1, an A/B aid for tools/attn_peaky.py)
constexpr float REREF_ABOVE = 18446744073709551616.0f;     // 2^64: running row sum that triggers the in-place re-reference
                                                           // (36 octaves of headroom per three tiles up to L_LIMIT; measured
                                                           // with 2^20: 1.7 / 7.7 events per wave at logit std 8 / 16 nats cost
                                                           // the launch 3.4 % / 9.4 %, and bought nothing -- fp32 is scale-free)
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

#define CA_A4_KERNEL ca_attn4_kernel
#define CA_A4_QK_T "bf16"
#include "ca_attn4_kernel.inc"
#undef CA_A4_KERNEL
#undef CA_A4_QK_T

// The same kernel for q / k rows stored as IEEE half (ca_gemm_problem.qk_f16: the layers whose heat maps are
// requested).  The output-space maps' error is the bf16 rounding of the rotated q and k (tests/tools/error_budget.py,
// profiles/r04_error_budget_output_space_*.json: 9e-4 -> 2.5e-4 per single map with 11-bit q / k; v and P do not
// matter); v_mfma_f32_32x32x16_f16 has the bf16 form's rate and register layout, so nothing else changes.
#define CA_A4_KERNEL ca_attn4_qk16_kernel
#define CA_A4_QK_T "f16"
#include "ca_attn4_kernel.inc"
#undef CA_A4_KERNEL
#undef CA_A4_QK_T

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

int ca_attn4_launch(const AttnLaunch &L, int total, bool qk_f16, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_done{0};
  const unsigned long long dev_bit = ca_device_bit();
  if (!(attr_done.load(std::memory_order_acquire) & dev_bit)) {
    for (const void *fn : {(const void *)ca_attn4_kernel, (const void *)ca_attn4_qk16_kernel}) {
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, a4::LDS_BYTES);
      if (e != hipSuccess) {
        ca_set_error("ca_attn_fwd_bf16: hipFuncSetAttribute(ca_attn4_kernel): %s", hipGetErrorString(e));
        return CA_ERR_LAUNCH;
      }
    }
    attr_done.fetch_or(dev_bit, std::memory_order_release);
  }
  // more units than CUs: one workgroup per CU walks them (no workgroup dispatch between units; CA_ATTN_PERSIST=0: one
  // workgroup per unit, round 3's launch)
  static const bool persist = ca_ab_env("CA_ATTN_PERSIST", 1) != 0;
  const int n_cu = ca_cu_count();
  AttnLaunch LL = L;
  LL.total_units = total;
  const int grid = (persist && n_cu > 0 && n_cu % 8 == 0 && total > n_cu) ? n_cu : total;
  if (qk_f16)
    hipLaunchKernelGGL(ca_attn4_qk16_kernel, dim3(grid), dim3(256), a4::LDS_BYTES, stream, LL);
  else
    hipLaunchKernelGGL(ca_attn4_kernel, dim3(grid), dim3(256), a4::LDS_BYTES, stream, LL);
  return CA_OK;
}
