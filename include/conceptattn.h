/*
 * conceptattn.h -- C ABI of libconceptattn.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * ConceptAttention inference hot path.
 *
 * The reference (manuragkhullar/ConceptAttention) is pure PyTorch and has no FFI; what this
 * library replaces are the PyTorch op sequences on the path.  Each
 * entry point cites the reference lines (relative to the reference root) whose arithmetic it
 * implements.  The reference-side binding a maintainer would add is the ctypes stub shown in
 * INTEGRATION.md; conceptattention_amd/_lib.py is that stub.
 *
 * Contract (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer borrowed from the caller (PyTorch
 *     tensors); the library never allocates, frees or retains device memory;
 *   - every call is asynchronous on the hipStream_t passed as `stream` (a void* here) and does
 *     no host synchronisation, so calls may be captured into a hipGraph;
 *   - returns 0 on success, a negative CA_ERR_* code on a rejected argument (nothing is
 *     launched in that case); ca_last_error() gives a thread-local message;
 *   - activations and weights are bf16 row-major; modulation vectors, RoPE tables, logits and
 *     heat-map accumulators are fp32; matrix products accumulate in fp32 (MFMA).
 *   - head_dim is fixed at 128 (Flux geometry: concept_attention/flux/src/flux/util.py:34-47).
 */
#ifndef CONCEPTATTN_H
#define CONCEPTATTN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CA_VERSION 125 /* 0.1.2: ca_gemm_problem.qpre_f32 / q_out_scale, fp32 image vectors in ca_heatmap_logits_bf16,
                          CA_ATTN_Q_PRESCALED; .1: ca_axpy_f32, ca_split_bf16; .2: ca_attn_stats; .3: ca_gemm_problem.qk_f16,
                          ca_attn_fwd_qk16; .4: ca_qpre_finish_rope_f32; .5: ca_heatmap_fused */

#define CA_OK 0
#define CA_ERR_ARG (-1)    /* bad shape / null pointer / misalignment */
#define CA_ERR_LAUNCH (-2) /* hip launch failure */
#define CA_ERR_ARCH (-3)   /* device is not gfx950 */

typedef void *ca_stream_t; /* hipStream_t */

int ca_version(void);
const char *ca_last_error(void);
/* 0 if the current device is gfx950, CA_ERR_ARCH otherwise. */
int ca_check_device(void);

/* ------------------------------------------------------------------------------------------
 * Grouped GEMM with fused epilogue:  out[m,n] = epi( sum_k A[m,k] * W[n,k] + bias[n] )
 * Replaces nn.Linear on the path: qkv / proj / mlp of ModifiedDoubleStreamBlock
 * (concept_attention/modified_double_stream_block.py:90,96,102,194-202), linear1 / linear2 of
 * ModifiedSingleStreamBlock (concept_attention/modified_single_stream_block.py:49-54), img_in /
 * txt_in / final_layer.linear (concept_attention/modified_flux_dit.py:98,105,120,159).
 * Up to CA_GEMM_MAX_PROBLEMS problems share one launch (image stream + text/concept stream of a
 * double block), so the grid fills all 256 CUs.
 * Requirements: K % 64 == 0, N % tile_n == 0, lda/ldw/ldc/ldr/ld2 % 8 == 0, 16-byte aligned
 * pointers.  M is arbitrary (rows are masked).  A call is one kernel launch on `stream`, or two:
 * under the 256x256 ping-pong tile a problem's last row tile with at most 128 rows (M % 256 in
 * [1, 128]: the concept rows a [concept | text] stream carries past its full row tiles) runs as
 * 32 x 128 tiles of a second kernel queued right behind the first (bit-identical results; set
 * CA_GEMM_THIN_KERNEL=0 in the environment to keep those rows in the first kernel's tile walk).
 */
enum {
  CA_EPI_BIAS = 0,          /* out = acc + bias */
  CA_EPI_GELU_TANH = 1,     /* out = gelu_tanh(acc + bias)          (mlp.0, :196)          */
  CA_EPI_GATE_RESIDUAL = 2, /* out = resid + gate[n]*(acc + bias)   (:194-202)             */
  CA_EPI_SPLIT_GELU = 3,    /* n <  n_split: out  = acc + bias                             */
                            /* n >= n_split: out2 = gelu_tanh(acc+bias), column n-n_split  */
                            /* (single block linear1 -> qkv | mlp, single_stream_block:49) */
  CA_EPI_QKV_NORM_ROPE = 4  /* qkv projection with QKNorm + RoPE fused (256x256 ping-pong tile only):      */
                            /* columns [0, n_split) are q|k|v thirds of heads*128 columns each (or, with    */
                            /* N = n_split / 3, the q third alone);                                          */
                            /* q, k: out = rope(rmsnorm(acc+bias) * norm_{q,k}), optional q_prerope store;   */
                            /* v: out = acc + bias; columns >= n_split: out2 = gelu_tanh(acc+bias)           */
                            /* (flux/modules/layers.py:63-84, flux/math.py:25-30, double_stream_block:189)  */
};

/* tile = block tile M x N; the PP ("ping-pong") kernels are the pipelined fast path */
enum { CA_TILE_AUTO = 0, CA_TILE_256x256 = 1, CA_TILE_256x192 = 2, CA_TILE_256x128 = 3, CA_TILE_256x64 = 4,
       CA_TILE_PP_256x256 = 5, CA_TILE_PP_256x128 = 6, CA_TILE_PP_256x192 = 7 };

#define CA_GEMM_MAX_PROBLEMS 2

typedef struct {
  const void *A;     /* bf16 [M,K], row stride lda (elements) */
  const void *W;     /* bf16 [N,K], row stride ldw: nn.Linear weight (out,in) */
  const void *bias;  /* bf16 [N] or NULL */
  void *out;         /* bf16 [M,N] (or [M,n_split] for SPLIT_GELU), row stride ldc */
  const void *resid; /* bf16 [M,N], row stride ldr; GATE_RESIDUAL only; may alias out */
  const float *gate; /* fp32 [N]; GATE_RESIDUAL: gate for rows <  gate_rows */
  const float *gate2;/* fp32 [N]; GATE_RESIDUAL: gate for rows >= gate_rows (may be NULL if gate_rows >= M) */
  void *out2;        /* bf16 [M,N-n_split], row stride ld2; SPLIT_GELU / QKV_NORM_ROPE tail */
  const void *norm_q;/* bf16 [128] query_norm.scale; QKV_NORM_ROPE only */
  const void *norm_k;/* bf16 [128] key_norm.scale */
  const float *rope; /* fp32 [M,64,2] (cos,sin) for this problem's rows */
  void *q_prerope;   /* optional [M, heads*128], row stride ldp: normalised pre-RoPE q; bf16, or fp32 if qpre_f32 */
  const float *a_scale; /* ca_gemm_fp8 only: fp32 [M], dequantisation scale of each row of A */
  const float *w_scale; /* ca_gemm_fp8 only: fp32 [N], dequantisation scale of each row of W */
  int32_t M, N, K;
  int32_t lda, ldw, ldc, ldr, ld2;
  int32_t n_split;   /* SPLIT_GELU: multiple of the tile width; QKV_NORM_ROPE: 3*heads*128 */
  int32_t gate_rows;
  int32_t epilogue;  /* CA_EPI_* */
  int32_t ldp;       /* row stride of q_prerope */
  int32_t out_f32;   /* 1: `out` (and `resid`) hold fp32 elements (ldc/ldr still in elements, % 4 == 0); BIAS and
                        GATE_RESIDUAL only.  The fp32 residual stream: img_in / txt_in write it, the attention-
                        and MLP-output projections update it in place; 0: bf16 as above */
  int32_t gate_stride;     /* GATE_RESIDUAL, batched forward: 0 = one gate vector per row range (above); else the */
  int32_t gate_item_rows;  /* rows < gate_rows are consecutive work items of gate_item_rows rows, the others of     */
  int32_t gate2_item_rows; /* gate2_item_rows rows, and item i of a range uses its base vector + i * gate_stride   */
                           /* floats (gate_stride % 4 == 0): every item of a batch has its own adaLN gates         */
  int32_t qpre_f32;        /* QKV_NORM_ROPE: 1 = q_prerope holds fp32 elements (ldp in elements, % 4 == 0): the    */
                           /* cross-attention-space vectors of modified_double_stream_block.py:189-190 without     */
                           /* their bf16 rounding (the heat-map logits are then formed from fp32 q on both sides); */
                           /* 2 = fp32 too, but the q projection (acc + bias) BEFORE its RMS norm: the caller adds */
                           /* a correction and normalises with ca_qpre_finish_f32;                                 */
                           /* 3 = (round 5) that correction fused: q_prerope holds such a raw projection on ENTRY, */
                           /* it is added to acc + bias before the norm and overwritten with the normalised vector */
                           /* -- the low-plane q projection of a captured layer (N = n_split / 3: the q third      */
                           /* alone, `out` = the attention's q rows): one launch instead of GEMM + finish kernel   */
  float q_out_scale;       /* QKV_NORM_ROPE: the rotated q is multiplied by this in fp32 before its ONE rounding   */
                           /* to bf16 (0 = 1.0; k, v and q_prerope are not scaled).  With softmax_scale * log2(e)  */
                           /* here, ca_attn_fwd_bf16(scale = CA_ATTN_Q_PRESCALED) needs no per-score multiply      */
  int32_t qk_f16;          /* QKV_NORM_ROPE: 1 = the rotated q and k are stored as IEEE half (fp16, 11-bit         */
                           /* mantissa; |values| stay far below 65504 behind an RMS norm) in the same 2-byte       */
                           /* elements of `out`; v stays bf16.  Read by ca_attn_fwd_qk16.  0 = bf16                */
  int32_t _pad;
} ca_gemm_problem;

int ca_gemm_bf16(const ca_gemm_problem *problems, int32_t n_problems, int32_t tile, ca_stream_t stream);
/* The CA_TILE_* value CA_TILE_AUTO resolves to for these problems (> 0), or CA_ERR_ARG. No launch. */
int ca_gemm_auto_tile(const ca_gemm_problem *problems, int32_t n_problems);

/* The same grouped GEMM and epilogues on OCP fp8 (e4m3fn) operands, for the reduced-precision sweep mode
 * (BASELINE.json configs[4]; no counterpart in the reference, which is bf16 throughout -- SURVEY.md 8f-2):
 *   out[m,n] = epi( a_scale[m] * w_scale[n] * sum_k A8[m,k] * W8[n,k] + bias[n] )
 * A8/W8 are e4m3 bytes produced by ca_quantize_rows_fp8 (absmax per row), lda/ldw in elements (= bytes),
 * K % 128 == 0, lda/ldw % 16 == 0; the accumulation is fp32 (v_mfma_scale_f32_16x16x128_f8f6f4 with unit
 * block scales), out/resid/bias/gates stay bf16/fp32 as above.  256x256 ping-pong tile only (N % 256 == 0).
 */
int ca_gemm_fp8(const ca_gemm_problem *problems, int32_t n_problems, ca_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Flash attention forward, head_dim 128, no mask:  out = softmax(q k^T * scale) v  per head.
 * Replaces F.scaled_dot_product_attention at modified_double_stream_block.py:112-116 (text+image
 * rows) and :162-168 (concept rows; only the C concept query rows are evaluated, which equals
 * rows [:C] of the reference's (C+L)x(C+L) product), and flux/math.py:6-12 for the single blocks.
 * q/k/v are read in place from a [rows, >=3*H*128] projection buffer: head h of a row lives at
 * column h*128 of the pointer given; the output is written head-concatenated ("B H L D -> B L (H D)",
 * modified_double_stream_block.py:170-176).  The key/value set of a problem is the concatenation
 * of two row segments (segment 1 may be empty), e.g. [concept rows ; image rows].
 * Up to CA_ATTN_MAX_PROBLEMS problems share one launch (per work item of a batch: its text+image rows, and its
 * concept rows); their workgroups are laid out in the order given (put the small concept problems first).
 * The query rows of a problem (and the output rows with them) may themselves come in two row segments: rows
 * [0, nq0) at q / out, rows [nq0, nq) at q1 / out1 -- in a batched forward an item's text rows and image rows are
 * not adjacent.  nq0 = 0 or nq0 = nq: one segment (q1 / out1 unused).
 */
#define CA_ATTN_MAX_PROBLEMS 16
typedef struct {
  const void *q; /* bf16, row stride ldq */
  void *out;     /* bf16, row stride ldo */
  const void *k0, *v0; /* segment 0: n0 rows, row stride ldkv */
  const void *k1, *v1; /* segment 1: n1 rows, row stride ldkv */
  float *out_f32; /* optional fp32 copy of the output rows [nq, heads*128] (row stride ldo32; indexed by the
                     problem's row number, also with two query segments), or NULL: used for the C concept rows and,
                     in the layers whose maps are requested, for the image rows, so that the heat-map products
                     (concept_attention_pipeline.py:57-62) see neither side's bf16 rounding */
  const void *q1; /* second query segment (rows nq0..nq-1), row stride ldq; or NULL */
  void *out1;     /* its output rows, row stride ldo */
  const float *hm_con; /* (round 5; pre-scaled-q kernels only) optional fp32 [hm_C, heads*128], row stride ldhc: the C   */
                       /* concept rows' attention outputs of this work item, complete BEFORE this launch starts        */
  float *hm_part;      /* with hm_con: fp32 [heads][nq - nq0][8] -- per head the partial output-space heat-map logits  */
                       /* of the second query segment's rows (the image rows):                                         */
                       /*   hm_part[(head * (nq - nq0) + r) * 8 + c] = <out row nq0 + r of that head, in fp32 before   */
                       /*   its rounding to bf16, hm_con[c, head*128 ..]>                                              */
                       /* i.e. the dot products of concept_attention_pipeline.py:57-61 split by head, formed from the  */
                       /* accumulators instead of an fp32 copy of every row (out_f32: 53 MB per item); summed over the */
                       /* heads and weighted by ca_heatmap_fused (img_f32 = 2)                                         */
  int32_t nq, n0, n1;
  int32_t ldq, ldo, ldkv, ldo32;
  int32_t nq0;    /* query rows in the first segment; 0 or nq = all */
  int32_t hm_C;   /* concepts in hm_con (1..8) */
  int32_t ldhc;   /* row stride of hm_con (elements, % 4 == 0) */
  int32_t _pad;
} ca_attn_problem;

/* scale = the softmax scale (1/sqrt(128) at the reference's call sites), or CA_ATTN_Q_PRESCALED when the q rows were
 * written with softmax_scale * log2(e) already folded in (ca_gemm_problem.q_out_scale): p = exp2(q.k - reference)
 * then costs no multiply per score. */
#define CA_ATTN_Q_PRESCALED 0.0f
int ca_attn_fwd_bf16(const ca_attn_problem *problems, int32_t n_problems, int32_t num_heads,
                     float scale, ca_stream_t stream);
/* ca_attn_fwd_bf16(scale = CA_ATTN_Q_PRESCALED) for q and k rows that hold IEEE half (fp16) instead of bf16 -- written
 * so by the qkv epilogue with ca_gemm_problem.qk_f16 = 1; v, the probabilities and the outputs stay bf16 / fp32.  Used
 * for the layers whose heat maps are requested: the bf16 rounding of the rotated q and k is what bounds the output-
 * space maps (modified_double_stream_block.py:185-191 on :112-116), and an 11-bit mantissa costs the MFMA nothing. */
int ca_attn_fwd_qk16(const ca_attn_problem *problems, int32_t n_problems, int32_t num_heads, ca_stream_t stream);
/* Diagnostics of the pre-scaled-q kernel's two rare paths on the current device, since the last reset:
 * counters[0] = workgroups whose rows were recomputed with a running maximum (a row sum passed 2^100 or
 * overflowed: a score > 100 octaves above its row's first-tile maximum with no check in between, or inf / NaN inputs),
 * counters[1] = in-place re-reference events (per wave: a running row sum passed 2^64 and the reference was moved up).  Both are 0 on the data the
 * reference's synthetic weights produce.  A blocking device-to-host copy: not for the hot path; reset != 0 zeroes them. */
int ca_attn_stats(unsigned long long *counters, int32_t reset);

/* ------------------------------------------------------------------------------------------
 * LayerNorm (no affine, eps) followed by adaLN modulation:  out = (1 + scale) * LN(x) + shift
 * modified_double_stream_block.py:88-89,94-95,100-101,196,199,202; single_stream_block:48;
 * LastLayer flux/modules/layers.py:250-251.  Row ranges may use different modulation vectors
 * (concept rows / text rows / image rows): segment i covers rows [row_end[i-1], row_end[i]).
 */
#define CA_MAX_SEGMENTS 16
typedef struct {
  int32_t row_end;
  int32_t _pad;
  const float *shift; /* fp32 [H] */
  const float *scale; /* fp32 [H] */
} ca_mod_segment;

int ca_ln_modulate_bf16(const void *x, int32_t ldx, void *out, int32_t ldo, int32_t M, int32_t H,
                        const ca_mod_segment *segs, int32_t n_segs, float eps, ca_stream_t stream);
/* The same two with an fp32 input x (row stride ldx elements, % 4 == 0): the residual stream kept in fp32
 * (its per-block bf16 rounding is what makes the heat-map error grow with depth). */
int ca_ln_modulate_f32in(const float *x, int32_t ldx, void *out, int32_t ldo, int32_t M, int32_t H,
                         const ca_mod_segment *segs, int32_t n_segs, float eps, ca_stream_t stream);
int ca_ln_modulate_f32in_fp8(const float *x, int32_t ldx, void *out8, int32_t ldo, float *out_scale, int32_t M,
                             int32_t H, const ca_mod_segment *segs, int32_t n_segs, float eps, ca_stream_t stream);
/* ca_ln_modulate_f32in with a second bf16 plane out_lo = bf16(y - float(bf16(y))) (row stride ldlo): what the bf16
 * rounding of the modulated row y drops.  out + out_lo carry y to ~16 mantissa bits. */
int ca_ln_modulate_f32in_split(const float *x, int32_t ldx, void *out, int32_t ldo, void *out_lo, int32_t ldlo,
                               int32_t M, int32_t H, const ca_mod_segment *segs, int32_t n_segs, float eps,
                               ca_stream_t stream);
/* Same, with the result quantised for ca_gemm_fp8: out8 = e4m3 bytes (row stride ldo bytes, % 16),
 * out_scale[row] = absmax(row) / 448 (fp32 [M]). */
int ca_ln_modulate_fp8(const void *x, int32_t ldx, void *out8, int32_t ldo, float *out_scale, int32_t M,
                       int32_t H, const ca_mod_segment *segs, int32_t n_segs, float eps, ca_stream_t stream);
/* Row-wise absmax quantisation bf16 [M,K] -> e4m3 [M,K] + fp32 scale per row (weights once per model;
 * activations between two fp8 GEMMs). */
int ca_quantize_rows_fp8(const void *x, int32_t ldx, void *out8, int32_t ldo, float *out_scale, int32_t M,
                         int32_t K, ca_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * QK-RMSNorm + RoPE, in place on the q and k thirds of a [M, 3*H*128] projection buffer.
 * QKNorm: flux/modules/layers.py:63-84 (fp32, eps 1e-6, per-head scale[128]);
 * apply_rope: flux/math.py:25-30 (interleaved pairs (2i,2i+1)); rope table [M,64,2] fp32 =
 * (cos,sin) of rope() flux/math.py:15-22 computed by the host in float64.
 * Optionally stores the normalised, PRE-RoPE q (what the reference captures as
 * cross_attention_{image,concept}_vectors, modified_double_stream_block.py:189-190) to q_prerope.
 */
typedef struct {
  int32_t row_end;
  int32_t _pad;
  const void *q_scale; /* bf16 [128] */
  const void *k_scale; /* bf16 [128] */
} ca_norm_segment;

int ca_qknorm_rope_bf16(void *qkv, int32_t ld, int32_t M, int32_t num_heads,
                        const ca_norm_segment *segs, int32_t n_segs, const float *rope_cos_sin,
                        void *q_prerope, int32_t ldp, ca_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Small-batch matrix-vector products (weight streaming, HBM-bound):
 *   out[v,n] (+)= sum_k f(x[v,k]) * W[n,k] + bias[n],  v < nv <= 8,  f = SiLU or identity.
 * Modulation (flux/modules/layers.py:113-126), MLPEmbedder (:52-60), LastLayer.adaLN (:246,249).
 */
int ca_gemv_bf16(const float *x, int32_t nv, int32_t ldx, const void *W, const void *bias, float *out,
                 int32_t ldo, int32_t N, int32_t K, int32_t silu_input, int32_t accumulate,
                 ca_stream_t stream);

/* Cross-attention-space query vectors (post-QKNorm, pre-RoPE q; modified_double_stream_block.py:189-190) from the
 * UNROUNDED LayerNorm output: x fp32 [M, heads*128] = the q projection of bf16(y) before its norm (ca_gemm_problem
 * qpre_f32 = 2), d fp32 (or NULL) = the same weights applied to the low plane of ca_ln_modulate_f32in_split;
 * in place  x <- RMSNorm_128(x + d) * norm_scale  per head (flux/modules/layers.py:63-72).  The rounding of the
 * GEMM operand is ~90 % of the cross-space heat-map error of a bf16 MFMA path. */
int ca_qpre_finish_f32(float *x, int32_t ldx, const float *d, int32_t ldd, const void *norm_scale, int32_t M,
                       int32_t heads, ca_stream_t stream);
/* The same, and with q_out != NULL the normalised vector is also rotated (rope fp32 [M,64,2] for these rows:
 * apply_rope, flux/math.py:25-30), multiplied by q_out_scale (0 = 1) and stored as 2-byte elements (bf16, or IEEE half
 * with q_f16 = 1; row stride ldq elements) -- i.e. it REPLACES the attention's q rows that the qkv epilogue formed from
 * bf16(y) by the ones formed from the unrounded y.  The bf16 rounding of that GEMM operand, through q, is what bounds a
 * single output-space heat map (modified_double_stream_block.py:112-116 feeding :185-188): 7.7e-4 of 7.8e-4. */
int ca_qpre_finish_rope_f32(float *x, int32_t ldx, const float *d, int32_t ldd, const void *norm_scale,
                            const float *rope, void *q_out, int32_t ldq, float q_out_scale, int32_t q_f16,
                            int32_t M, int32_t heads, ca_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Concept heat maps, one (timestep, layer) at a time (compute_heatmaps_from_vectors,
 * concept_attention/concept_attention_pipeline.py:29-91, softmax branch):
 *   logits[c,p] = <img_vec[p,:], con_vec[c,:]>                                   (:57-61)
 *   acc[c,p]   += weight * softmax_c(logits[:,p])        (weight = 1/(|timesteps|*|layers|), :64-82)
 * fp32 accumulation (the reference does this in bf16; see DESIGN.md "tolerance").
 */
/* con_is_f32 is a bit set: bit 0 = con_vec is fp32 [C,dim] (else bf16), bit 1 = img_vec is fp32 [L,dim] (else bf16;
 * needs bit 0 as well); ldi / ldc in elements, multiples of 8 / 4. */
int ca_heatmap_logits_bf16(const void *img_vec, int32_t ldi, const void *con_vec, int32_t ldc,
                           int32_t con_is_f32, int32_t L, int32_t C, int32_t dim, float *logits,
                           ca_stream_t stream);
int ca_heatmap_softmax_accumulate(const float *logits, int32_t C, int32_t L, float weight, float *acc,
                                  ca_stream_t stream);
/* The same accumulation with the other two weightings across concepts the reference offers
 * (`attention_norm`, concept_attention_pipeline.py:33,64-71): norm = CA_NORM_SOFTMAX (as above),
 * CA_NORM_SPARSEMAX or CA_NORM_ENTMAX15.  The reference takes the latter two from the third-party `entmax`
 * package, which it neither pins nor vendors: they are implemented here from the published algorithms
 * (Martins & Astudillo 2016; Peters, Niculae & Martins 2019) and their parity with that package is UNPINNED.
 * The sparse norms need C <= 16. */
#define CA_NORM_SOFTMAX 0
#define CA_NORM_SPARSEMAX 1
#define CA_NORM_ENTMAX15 2
int ca_heatmap_norm_accumulate(const float *logits, int32_t C, int32_t L, int32_t norm, float weight, float *acc,
                               ca_stream_t stream);

/* The three steps above for up to CA_HEATMAP_MAX_PROBLEMS (work item, space) pairs of one layer in ONE launch
 * (round 5): per problem  logits[c,p] = <img_vec[p,:], con_vec[c,:]>  for all C concepts of a patch in one pass over
 * the image vectors, then  acc[c,p] += weight * norm_c(logits[:,p])  and, if acc2 != NULL,
 * acc2[c,p] += weight2 * norm_c(logits[:,p])  (the per-layer table row of the per-layer x noise-level sweep,
 * experiments/per_layer_segmentation/test_segmentations_per_layer.py:104-114); the logits never reach memory unless
 * `logits` != NULL.  Per patch and concept the arithmetic is that of ca_heatmap_logits_bf16 followed by
 * ca_heatmap_norm_accumulate (same k order, same expressions): the results are bit-identical to the three-launch form.
 * All problems of a call share L, C, dim and norm, and no two of them may name the same accumulator (the updates are
 * plain read-modify-writes by different workgroups).  Needs C <= 8 and C * dim * 4 bytes of LDS (<= 96 KB);
 * CA_ERR_ARG otherwise (the caller then uses the three-launch form). */
#define CA_HEATMAP_MAX_PROBLEMS 16
typedef struct {
  const void *img_vec; /* [L, dim] bf16, or fp32 if img_f32 = 1; row stride ldi elements (% 8 bf16, % 4 fp32).        */
                       /* img_f32 = 2: instead fp32 [ldi heads][L][8], the per-head partial logits an attention       */
                       /* launch left in ca_attn_problem.hm_part: logits[c,p] = their sum over the heads in head      */
                       /* order; con_vec is then unused (may be NULL)                                                 */
  const void *con_vec; /* [C, dim] bf16, or fp32 if con_f32; row stride ldc elements */
  float *acc;          /* fp32 [C, L] contiguous, or NULL */
  float *acc2;         /* fp32 [C, L] contiguous, or NULL */
  float *logits;       /* fp32 [C, L] contiguous, or NULL: the raw logits as well */
  int32_t ldi, ldc;
  int32_t img_f32, con_f32;
  float weight, weight2;
} ca_heatmap_problem;
int ca_heatmap_fused(const ca_heatmap_problem *problems, int32_t n_problems, int32_t L, int32_t C, int32_t dim,
                     int32_t norm, ca_stream_t stream);

/* Sinusoidal timestep embedding (timestep_embedding, flux/modules/layers.py:28-49):
 * out[v, 0:dim/2] = cos(time_factor*t[v]*f_i), out[v, dim/2:] = sin(...), f_i = max_period^(-i/(dim/2)). */
int ca_timestep_embedding_f32(const float *t, int32_t nt, float *out, int32_t dim, float time_factor,
                              float max_period, ca_stream_t stream);

/* Euler step of denoise(): x = x + a*y  (flux/sampling.py:141), bf16 in/out, fp32 math. */
int ca_axpy_bf16(void *x, const void *y, float a, int64_t n, ca_stream_t stream);
/* The same update with the latent kept in fp32 between the steps: x fp32 [n] += a * y (y bf16 [n], or fp32 [n] if
 * y_is_f32: the model's prediction unrounded).  The reference's loop (flux/sampling.py:141 on bf16 tensors) re-rounds
 * the running latent after every step -- 2^-9 relative each time, accumulating; on this path the model's img_in then
 * takes the fp32 value as two bf16 planes (ca_split_bf16). */
int ca_axpy_f32(float *x, const void *y, int32_t y_is_f32, float a, int64_t n, ca_stream_t stream);
/* x fp32 [rows, K] (row stride ldx) -> hi = bf16(x), lo = bf16(x - hi) (row stride ldo), K % 4 == 0: the two planes of
 * a GEMM operand that must not lose its low bits (ca_silu_split_bf16 without the silu): the latent into img_in
 * (modified_flux_dit.py:98). */
int ca_split_bf16(const float *x, int32_t ldx, void *hi, void *lo, int32_t ldo, int32_t rows, int32_t K, ca_stream_t stream);

/* silu(x) of the conditioning vectors (Modulation: lin(silu(vec)), flux/modules/layers.py:113-126) split into two
 * bf16 planes, hi = bf16(s), lo = bf16(s - hi): hi + lo carries s to ~16 mantissa bits, so that the adaLN modulation
 * of every block -- [2 * items * steps, H] x [sum N, H]^T, the weights streamed ONCE -- can run as two bf16 MFMA GEMMs
 * (ca_gemm_bf16, the second one accumulating) instead of one weight pass per 4 vectors (ca_gemv_bf16).
 * x fp32 [rows, K] (row stride ldx), hi / lo bf16 [rows, K] (row stride ldo), K % 4 == 0. */
int ca_silu_split_bf16(const float *x, int32_t ldx, void *hi, void *lo, int32_t ldo, int32_t rows, int32_t K,
                       ca_stream_t stream);

/* The two planes' products of a single-pass modulation GEMM ([hi; lo] stacked as 2*nv rows, no bias) folded into the
 * modulation: out[v,n] = (pair[v,n] + bias[n]) + pair[nv + v,n] -- the roundings of the two-launch form (first GEMM
 * with bias, second one accumulating), so both forms give the same bits.  pair fp32 [2*nv, N] (row stride ldp),
 * bias bf16 [N] or NULL, out fp32 [nv, N] (row stride ldo), N % 4 == 0. */
int ca_modulation_combine_f32(const float *pair, int32_t ldp, const void *bias, float *out, int32_t ldo, int32_t nv,
                              int32_t N, ca_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CONCEPTATTN_H */
