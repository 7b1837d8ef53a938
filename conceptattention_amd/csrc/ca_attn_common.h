// Launch descriptor and tile constants shared by the two attention kernels (ca_attn.hip, ca_attn4.hip).
#pragma once
#include "ca_common.h"

namespace ca_attn_detail {
struct AttnLaunch {
  ca_attn_problem p[CA_ATTN_MAX_PROBLEMS];
  int32_t nqb[CA_ATTN_MAX_PROBLEMS];      // 256-row query blocks per head
  int32_t blk_end[CA_ATTN_MAX_PROBLEMS];  // workgroups of problems 0..i (problems are laid out one after another)
  int32_t n_problems;
  int32_t num_heads;
  float scale_log2;   // softmax scale * log2(e)
  int32_t total_units;   // ca_attn4_kernel: workgroup-sized units of the launch (= its grid unless the walk is persistent)
  int32_t flags;      // bit 0: ca_attn4_kernel without its in-place re-reference (CA_ATTN_REREF=0), bit 1: with round 3's
                      // row-sum limit of 2^60 instead of 2^100 (CA_ATTN_LIMIT60=1); both A/B aids of tools/attn_peaky.py
};

constexpr int KV_TILE = 64;
constexpr float REDO_LIMIT = 1073741824.0f;  // 2^30: a row sum above this sends the tile through the max-tracking path
constexpr int TILE_BYTES = KV_TILE * 256;  // one K or V tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;
constexpr int ATTN_LDS = 2 * BUF_BYTES;

}  // namespace ca_attn_detail

// ca_attn4.hip: the one-wave-per-SIMD kernel for pre-scaled q (host side: attribute once per device, launch)
int ca_attn4_launch(const ca_attn_detail::AttnLaunch &L, int total_workgroups, bool qk_f16, hipStream_t stream);
// counters of its two rare paths: out[0] = recomputed workgroups, out[1] = re-reference events (synchronous copy)
int ca_attn4_read_counters(unsigned long long *out, int reset);
