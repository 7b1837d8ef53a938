"""HipFluxDiT: the modified Flux DiT forward (concept-token stream included) on MI355X.

Host-side mirror of the reference's model object for this path:
  * ``ModifiedFluxDiT.forward``              concept_attention/modified_flux_dit.py:75-163
  * ``ModifiedDoubleStreamBlock.forward``    concept_attention/modified_double_stream_block.py:69-204
  * ``ModifiedSingleStreamBlock.forward``    concept_attention/modified_single_stream_block.py:43-56
It keeps the call contract ``denoise`` (flux/sampling.py:126-139) and ``encode_image``
(concept_attention_pipeline.py:284-297) rely on -- same keyword names, ``(pred | None, dict)``
return value with the four reference keys stacked over the double blocks -- so it can be passed
wherever the reference takes a ``dit_class`` instance.  All arithmetic runs in the gfx950
kernels of libconceptattn.so (conceptattention_amd.ops); PyTorch provides device memory only.

Data layout in HBM (one resident activation set per model instance, batch 1):
  X    [C+T+L, H]   fp32 residual streams, rows = [concept tokens | text tokens | image tokens]
  XM   [C+T+L, H]   LayerNorm+modulated input of the next projection
  QKV  [C+T+L, 3H]  projection output, q|k|v thirds, head-major inside a third
  ATT  [C+T+L, H]   attention output, head-concatenated
  HID  [C+T+L, 4H]  MLP hidden of the double blocks
  CAT  [T+L, 5H]    single blocks: [attention | gelu(mlp)] input of linear2
With this row order the text-weight projections see [concepts|text] as one contiguous matrix
(the concept stream re-uses the text weights, modified_double_stream_block.py:100-104), the
single blocks see [text|image] as one contiguous matrix (modified_flux_dit.py:149), and the
concept attention reads [concept keys | image keys] as two row segments without any copy.
"""
from __future__ import annotations

import math
import os

from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L
from . import ops
from .params import FluxParams
from .weights import iter_synthetic_state_dict, state_dict_spec

def on_own_device(fn):
    """Run a method with ``self.device`` as the current HIP device: the kernels launch on the calling thread's
    current device and stream, so an object built for cuda:1 must not depend on the caller's current device."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapped


DICT_KEYS = (
    "output_space_concept_vectors",
    "output_space_image_vectors",
    "cross_attention_concept_vectors",
    "cross_attention_image_vectors",
)


class FluxWeights:
    """bf16 device weights under the BFL state-dict names.  The adaLN modulation matrices of all
    blocks live in ONE stacked buffer so a diffusion step computes every block's
    shift/scale/gate with a single weight-streaming launch."""

    def __init__(self, params: FluxParams, device):
        self.params = params
        self.device = torch.device(device)
        H = params.hidden_size
        self.tensors: dict[str, torch.Tensor] = {}
        self.fp8 = None  # name -> (e4m3 bytes, fp32 row scales) of the large projections, built on first fp8 use
        mod_names = []
        for i in range(params.depth):
            mod_names += [(f"double_blocks.{i}.img_mod.lin", 6 * H), (f"double_blocks.{i}.txt_mod.lin", 6 * H)]
        for i in range(params.depth_single_blocks):
            mod_names.append((f"single_blocks.{i}.modulation.lin", 3 * H))
        mod_names.append(("final_layer.adaLN_modulation.1", 2 * H))
        self.mod_offset: dict[str, int] = {}
        off = 0
        for name, n in mod_names:
            self.mod_offset[name] = off
            off += n
        self.mod_rows = off
        self.mod_w = torch.empty(off, H, device=self.device, dtype=torch.bfloat16)
        self.mod_b = torch.empty(off, device=self.device, dtype=torch.bfloat16)
        for name, shape in state_dict_spec(params):
            base, kind = name.rsplit(".", 1)
            if base in self.mod_offset:
                o = self.mod_offset[base]
                src = self.mod_w if kind == "weight" else self.mod_b
                self.tensors[name] = src[o:o + shape[0]]
            else:
                self.tensors[name] = torch.empty(shape, device=self.device, dtype=torch.bfloat16)

    def __getitem__(self, name: str) -> torch.Tensor:
        return self.tensors[name]

    def state_dict(self) -> dict[str, torch.Tensor]:
        return dict(self.tensors)

    def load_state_dict(self, sd, strict: bool = True):
        """Copy tensors in (any float dtype / device).  Same (missing, unexpected) semantics as
        nn.Module.load_state_dict (the reference calls it with strict=False,
        concept_attention/image_generator.py:44)."""
        self.fp8 = None
        missing = [k for k in self.tensors if k not in sd]
        unexpected = [k for k in sd if k not in self.tensors]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:4]}.. unexpected {unexpected[:4]}..")
        for k, t in self.tensors.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(t.shape):
                    raise RuntimeError(f"load_state_dict: {k} has shape {tuple(sd[k].shape)}, expected {tuple(t.shape)}")
                t.copy_(sd[k])
        return missing, unexpected

    def init_synthetic(self, seed: int = 0, on_device: bool = True):
        """Random-init weights of the right geometry.  on_device=True draws on
        the GPU (fast, 23.8 GB for full Flux); False draws on the CPU generator so the values
        are identical to ``weights.synthetic_state_dict`` (used by the parity tests)."""
        dev = self.device if on_device else "cpu"
        self.fp8 = None
        for name, t in iter_synthetic_state_dict(self.params, seed=seed, device=dev, dtype=torch.float32):
            self.tensors[name].copy_(t)
        return self


@dataclass
class _Geom:
    """Row geometry of one forward: B work items; rows [0, oT) concept, [oT, oI) text, [oI, n) image."""
    B: int
    C: int
    T: int
    L: int

    @property
    def oT(self):
        return self.B * self.C

    @property
    def oI(self):
        return self.B * (self.C + self.T)

    @property
    def n(self):
        return self.B * (self.C + self.T + self.L)


@dataclass
class HeatmapRequest:
    """Fused heat-map accumulation (replaces materialising the 478 MB/step dicts that
    compute_heatmaps_from_vectors later slices, concept_attention_pipeline.py:57-82)."""
    layer_indices: tuple
    weight: float                 # 1 / (|timesteps| * |layers|)
    out_space: Optional[torch.Tensor]     # fp32 [C, L] accumulator (output-space maps), or None (per-layer tables only)
    cross_space: Optional[torch.Tensor]   # fp32 [C, L] accumulator (cross-attention-space maps), or None
    # optional per-layer tables [len(layer_indices), C, L] (row i = layer_indices[i]), accumulated with
    # per_layer_weight: the per-layer x per-noise-level extraction of
    # experiments/per_layer_segmentation/test_segmentations_per_layer.py:104-114 without the vector stacks
    per_layer_out: Optional[torch.Tensor] = None
    per_layer_cross: Optional[torch.Tensor] = None
    per_layer_weight: float = 1.0
    norm: int = L.NORM_SOFTMAX    # weighting across concepts: softmax | sparsemax | entmax15 (_lib.NORM_*)


class HipFluxDiT:
    """Drop-in for the reference's ``ModifiedFluxDiT`` instance on the hot path (inference only)."""

    def __init__(self, params: FluxParams, device="cuda:0", weights: Optional[FluxWeights] = None,
                 attention_block_class=None, precision: str = "bf16", residual_dtype=torch.float32,
                 bf16_timesteps: bool = False):
        # attention_block_class is accepted for signature compatibility with
        # ModifiedFluxDiT(params, attention_block_class=...) (modified_flux_dit.py:34); the HIP
        # path has exactly one block implementation.
        if params.head_dim != 128:
            raise ValueError("HipFluxDiT: the gfx950 kernels are built for head_dim 128")
        self.params = params
        self.device = torch.device(device)
        self.in_channels = params.in_channels
        self.out_channels = params.in_channels
        self.hidden_size = params.hidden_size
        self.num_heads = params.num_heads
        L.load()
        self.weights = weights if weights is not None else FluxWeights(params, self.device)
        self._ws_key = None
        self._ws_cache = {}
        self._rope_key = None
        self._mod_cur = None
        self._mod_steps = None
        # The residual streams X are kept in fp32 by default: every block adds two gated projections to them, and
        # rounding the running sum to bf16 after each add (as a bf16 activation tensor does, the reference's own
        # bf16 run included) is what makes the heat-map error grow with depth -- 38 roundings by layer 18.
        # Measured (tests/tools/error_budget.py): 5e-3 -> 1.3e-3 max-abs on layers 15..18.
        # Cost: 27 MB more per LayerNorm read / projection write-back per block.  torch.bfloat16 restores the old
        # layout (A/B aid, and what the reference's activations are).
        if residual_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("residual_dtype must be torch.float32 or torch.bfloat16")
        self.residual_dtype = residual_dtype
        # The reference's production (bf16) run builds t_vec / guidance_vec in the activations' dtype and forms
        # time_factor * t in bf16 (flux/sampling.py:122-125, flux/modules/layers.py:37): at t = 0.75 it embeds 752,
        # and most shifted flux-dev schedule values are rounded twice; the embedding itself is cast to bf16 too.
        # Default False = the fp32 oracle's behaviour (t and 1000 t exact); True reproduces the reference's bf16
        # values (pinned by tests/golden/timestep_embedding_bf16.npz).
        self.bf16_timesteps = bool(bf16_timesteps)
        # adaLN modulations of MANY conditioning vectors (all steps x items, precompute_conditioning) as two bf16 MFMA
        # GEMMs instead of one weight-streaming GEMV pass per 4 vectors (_modulation_rows); False = GEMV only (A/B aid:
        # set the attribute; round 5 retired the CA_MODULATION_GEMM / CA_ATTN_PRESCALE / CA_SPLIT_Q_CAPTURE variables)
        self.modulation_by_gemm = True
        # softmax_scale * log2(e) folded into q by the qkv epilogue (in fp32, before q's one rounding to bf16), so the
        # attention kernel's probability is a bare exp2 (include/conceptattn.h CA_ATTN_Q_PRESCALED); False = the kernel
        # multiplies every score instead (A/B aid)
        self.prescale_q = True
        # sampling.denoise keeps the latent in fp32 between the Euler steps when the model says so (A/B: CA_FP32_LATENT=0)
        self.fp32_latent = os.environ.get("CA_FP32_LATENT", "1") != "0"
        # An fp32 copy of the captured layers' [text | image] attention rows for the output-space logits (round 3).
        # Round 5 re-measured it at full depth (VERDICT r04 weak #5 asked to drop it: 1 011 vs 888 us per captured 5-item
        # attention launch, 53 MB per item): layer 0's map moves by 2e-5 (tests/tools/diag_out_space.py), but deeper
        # layers do not agree -- without it the worst step-0 map is 8.1e-4 (layer 4) instead of 3.8e-4 and the final
        # maps 1.45e-4 instead of 7.9e-5 from the fp32 oracle (tests/test_full_depth_gpu.py fails its 1.5 x bounds).
        # So the fp32 precision stays -- but since round 5 without the copy: epilogue_logits (below) forms the logits from
        # the attention kernel's accumulators; this switch only matters with epilogue_logits = False ("0" = the bf16 rows).
        self.f32_image_vectors = os.environ.get("CA_F32_IMAGE_VECTORS", "1") != "0"
        # The cross-attention-space vectors (post-QKNorm, pre-RoPE q) of the captured layers from the UNROUNDED
        # LayerNorm output: the bf16 rounding of that GEMM operand is ~90 % of the cross-space heat-map error
        # (tests/tools/error_budget.py: 3.3e-3 -> 3.5e-4 per map).  The LayerNorm writes a second bf16 plane with what
        # the rounding drops, one more GEMM applies the q weights to it (image rows and concept rows of the captured
        # layers only: +0.45 % FLOPs per call), and ops.qpre_finish normalises the sum.  The q the ATTENTION uses is
        # untouched, so the image does not depend on which layers are captured.  False = one rounding more (A/B aid).
        self.split_q_capture = True
        # The rotated q and k of the attention as IEEE half instead of bf16 (ca_gemm_problem.qk_f16 ->
        # ca_attn_fwd_qk16): their bf16 rounding is one of the two things that bound a single output-space heat map
        # (round 4, tests/tools/error_budget.py --out-space2: 9e-4 -> 2.5e-4 per map with 11-bit q / k; v and the
        # probabilities do not matter), and behind an RMS norm the values sit far inside fp16's range.  "captured"
        # (default): the layers whose maps are requested -- where the attention's q is special anyway (split_q_attention
        # below); the chip clocks the f16 MFMA 1.6 % lower than the bf16 one, and with every block in half precision
        # ("all") the maps are no closer to the oracle (profiles/r04_full_depth_parity*.json: 5.6e-4 / 9.6e-4 worst single
        # map against 5.7e-4 / 8.4e-4) at -0.6 % heat maps/s.  "0": bf16 everywhere, as the reference (A/B aid).
        # Needs the pre-scaled-q kernel (prescale_q = False switches it off: _qk16).
        self.qk_f16 = os.environ.get("CA_QK_F16", "captured")
        # the ATTENTION's q of the captured layers' image / concept rows from the unrounded LayerNorm output as well
        # (ops.qpre_finish writes it over the epilogue's; _double_block); "0" = round 3's q (A/B aid)
        self.split_q_attention = os.environ.get("CA_SPLIT_Q_ATTENTION", "1") != "0"
        # Exact independence of everything the forward returns from WHICH layers' maps are requested (the reference's
        # property: the concept stream and the capture are read-only side computations, modified_double_stream_block.py:
        # 105-119, 185-191).  False (default): a captured layer's attention uses the accurate q for its image and concept
        # rows and half-precision q / k, so the latent moves at rounding level with layer_indices (bounded:
        # tests/test_full_depth_gpu.py::test_latent_dependence_on_the_captured_layer_set).  True: every layer's attention
        # output that feeds proj / the residual streams is formed exactly as in an uncaptured layer (rounded-operand q,
        # bf16 q / k), and the captured layers' maps come from SEPARATE attention problems of the same launch -- the
        # accurate q of the image rows (and of the concept rows) against the same keys and values, written to scratch
        # rows that only the heat maps read.  Cost: the image rows' attention twice in captured layers (+1.3 % of a
        # generate call with 4 of 57 layers captured; +15 % of a 19-layer sweep forward); single output-space maps then
        # carry k's bf16 rounding again (measured in the same test file).
        self.capture_independent_image = False
        self._layer_indep = False   # (set per layer by _double_block, read by _capture)
        self._layer_part = False
        # fp8 mode: the qkv projection of a layer whose maps are requested stays bf16 (_double_block)
        self.fp8_bf16_qkv_when_captured = os.environ.get("CA_FP8_QKV_BF16_CAPTURED", "1") != "0"
        # the heat-map updates of a captured layer (all work items, both spaces) as ONE ca_heatmap_fused launch; False =
        # logits + weighting launches per item and space (the same bits; parity / A-B aid, and what C > 8 uses)
        self.fused_heatmaps = True
        # (round 5) The output-space logits of a captured layer from the attention kernel's own accumulators: the concept
        # problems run first (their fp32 rows ATT32 must exist), every main problem's epilogue then forms, per head, the
        # dot products of its image rows (fp32, before their bf16 rounding) with the C concept rows and leaves
        # [heads, L, 8] partial logits (PART, 3 MB per item), which ca_heatmap_fused sums over the heads -- instead of an
        # fp32 copy of every [text | image] row (ATTI32: 53 MB per item, +115 us per captured 5-item attention launch,
        # and 250 MB of reads in the heat-map launch).  The same arithmetic in another summation order (per head, then
        # over the heads).  False = the fp32 rows (rounds 3-4; A/B aid).  Needs fused_heatmaps and the pre-scaled-q kernel.
        self.epilogue_logits = True
        if self.qk_f16 not in ("all", "captured", "0"):
            raise ValueError("CA_QK_F16 must be all, captured or 0")
        self.set_precision(precision)

    # ---- reduced-precision mode (BASELINE.json configs[4]; no counterpart in the reference)
    FP8_LINEARS = ("img_attn.qkv", "txt_attn.qkv", "img_attn.proj", "txt_attn.proj", "img_mlp.0", "txt_mlp.0",
                   "img_mlp.2", "txt_mlp.2", "linear1", "linear2")

    def set_precision(self, precision: str, keep_bf16_layers=()):
        """"bf16" (default; the parity path) or "fp8": the six large projections of every block run on e4m3
        operands (weights quantised once per output channel, activations per token by the producing
        kernel), accumulation fp32, everything else unchanged.  ``keep_bf16_layers``: double blocks that stay
        in bf16 even in fp8 mode (e.g. the layers whose attention outputs are turned into heat maps).  See
        DESIGN.md "fp8 mode" for the measured heat-map deviation from the bf16 path."""
        if precision not in ("bf16", "fp8"):
            raise ValueError(f"precision must be 'bf16' or 'fp8', got {precision!r}")
        self.precision = precision
        self.keep_bf16_layers = frozenset(int(l) for l in keep_bf16_layers)
        return self

    def _fp8_weights(self):
        W = self.weights
        if W.fp8 is None:
            W.fp8 = {name: ops.quantize_rows_fp8(t) for name, t in W.tensors.items()
                     if name.endswith(".weight") and name.rsplit(".", 1)[0].split(".", 2)[-1] in self.FP8_LINEARS}
        return W.fp8

    def materialize(self):
        """Build the lazily created state that model instances SHARE (the fp8 weight images live on the common
        FluxWeights object) on the CURRENT stream.  Callers that are about to run replicas on other streams call
        this first and then make those streams wait on the current one; otherwise a replica could launch an fp8
        GEMM on weight images whose quantisation kernels are still queued on another stream."""
        if self.precision == "fp8":
            self._fp8_weights()
        return self

    @staticmethod
    def _launch_gemm(problems):
        """One grouped GEMM launch (image-weight rows + text-weight rows of all work items)."""
        ops.gemm(problems)

    def _gemm(self, fp8, a, a8, a8s, wname, bias, out, *args, **kw):
        """One problem of a grouped launch in the current precision."""
        if fp8:
            q, sc = self._fp8_weights()[wname]
            return ops.Gemm(a8, q, bias, out, *args, a_scale=a8s, w_scale=sc, **kw)
        return ops.Gemm(a, self.weights[wname], bias, out, *args, **kw)

    # ---- nn.Module-like surface the reference's loader touches (image_generator.py:37-44,183,194)
    def load_state_dict(self, sd, strict: bool = True, assign: bool = False):
        return self.weights.load_state_dict(sd, strict=strict)

    def state_dict(self):
        return self.weights.state_dict()

    def to(self, *a, **k):
        return self

    def cpu(self):  # the reference round-trips 23.8 GB per call (image_generator.py:183,194); we keep it resident
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ workspace
    def _workspace(self, L_img: int, T: int, C: int, B: int = 1):
        """Activation set for a batch of B work items.  Rows are ordered
        [concept rows of item 0..B-1 | text rows of item 0..B-1 | image rows of item 0..B-1]: all rows that use
        the text weights are one contiguous matrix, all image rows another, and the single blocks' [text | image]
        rows a third (B = 1: the layout of the module docstring)."""
        key = (L_img, T, C, B, self.precision, self.residual_dtype)
        if self._ws_key == key:
            return
        # A few activation sets stay cached (LRU): callers that alternate shapes -- a ragged last group of a batched
        # run, B = 1 calls between B = 5 groups -- would otherwise free and re-zero gigabytes inside their loop.
        cached = self._ws_cache.pop(key, None)
        if cached is None:
            cached = self._alloc_workspace(L_img, T, C, B)
            while len(self._ws_cache) >= self.WS_CACHE_ENTRIES:
                self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = cached
        self.__dict__.update({k: v for k, v in cached.items() if k not in self._LAZY_BUFFERS})
        self._ws_key = key
        self._rope_key = None

    WS_CACHE_ENTRIES = 3

    # Buffers only some forwards need (the captured layers' fp32 vectors: ~0.55 GB per work item at 1024 x 1024; +53 MB
    # with f32_image_vectors) are allocated on first use, per activation set: a model that never returns maps never pays
    # for them.  An fp8-mode forward WITH captured layers does (round 4: their qkv projection stays bf16, so the split-q
    # buffers XML / QD / QPRE are used there too).  name -> (rows, columns..., dtype) of the set's geometry
    _LAZY_BUFFERS = {
        "QPRE": lambda n, B, T, L, H: ((n, H), torch.float32),       # post-QKNorm pre-RoPE q (cross-space vectors)
        "XML": lambda n, B, T, L, H: ((n, H), torch.bfloat16),       # low plane of XM: bf16(y - float(bf16(y)))
        "QD": lambda n, B, T, L, H: ((n, H), torch.float32),         # its q projection (ops.qpre_finish adds it)
        "ATTI32": lambda n, B, T, L, H: ((B, T + L, H), torch.float32),   # fp32 [text | image] attention rows
        # capture_independent_image: the accurate q of the captured layers' image / concept rows (the map-side attention
        # problems' queries) and those problems' bf16 output rows, which nothing but the heat maps reads
        # epilogue_logits: per-head partial output-space logits of every item's image rows [B, heads, L, 8]
        "PART": lambda n, B, T, L, H: ((B, H // 128, L, 8), torch.float32),
        "QACC": lambda n, B, T, L, H: ((n, H), torch.bfloat16),
        "ATTM": lambda n, B, T, L, H: ((n, H), torch.bfloat16),
    }

    def _lazy_buffer(self, name: str) -> torch.Tensor:
        ws = self._ws_cache[self._ws_key]
        t = ws.get(name)
        if t is None:
            L_img, T, C, B = self._ws_key[:4]
            shape, dtype = self._LAZY_BUFFERS[name](B * (C + T + L_img), B, T, L_img, self.params.hidden_size)
            t = ws[name] = torch.zeros(shape, device=self.device, dtype=dtype)
        return t

    QPRE = property(lambda self: self._lazy_buffer("QPRE"))
    XML = property(lambda self: self._lazy_buffer("XML"))
    QD = property(lambda self: self._lazy_buffer("QD"))
    ATTI32 = property(lambda self: self._lazy_buffer("ATTI32"))
    PART = property(lambda self: self._lazy_buffer("PART"))
    QACC = property(lambda self: self._lazy_buffer("QACC"))
    ATTM = property(lambda self: self._lazy_buffer("ATTM"))

    def clear_workspaces(self) -> None:
        """Drop every cached activation set (up to WS_CACHE_ENTRIES sets stay resident between calls: ~1.6 GB per work
        item at 1024 x 1024 with the capture buffers, 8 GB for a 5-item set); the next forward allocates afresh."""
        for ws in self._ws_cache.values():
            for k in ws:
                self.__dict__.pop(k, None)
        self._ws_cache.clear()
        self._ws_key = None
        self._rope_key = None

    def workspace_bytes(self) -> int:
        """Bytes of HBM the cached activation sets hold right now."""
        return sum(t.numel() * t.element_size() for ws in self._ws_cache.values() for t in ws.values()
                   if isinstance(t, torch.Tensor))

    def _alloc_workspace(self, L_img: int, T: int, C: int, B: int) -> dict:
        p, dev = self.params, self.device
        H, MLP = p.hidden_size, p.mlp_hidden
        n = B * (C + T + L_img)
        bf = dict(device=dev, dtype=torch.bfloat16)
        f32 = dict(device=dev, dtype=torch.float32)
        ws = dict(
            X=torch.zeros(n, H, device=dev, dtype=self.residual_dtype),
            XM=torch.zeros(n, H, **bf),
            QKV=torch.zeros(n, 3 * H, **bf),
            ATT=torch.zeros(n, H, **bf),
            HID=torch.zeros(n, MLP, **bf),
            CAT=torch.zeros(B * (T + L_img), H + MLP, **bf),
            # (QPRE, XML, QD, ATTI32 -- the fp32 vectors of the captured layers -- are allocated on first use: _LAZY_BUFFERS)
            ATT32=torch.zeros(max(B * C, 1), H, **f32),  # fp32 copy of the concept attention rows
            TXT_IN=torch.zeros(B * (C + T), p.context_in_dim, **bf),
            PRED=torch.zeros(B * L_img, p.in_channels, **bf),
            PRED32=torch.zeros(B * L_img, p.in_channels, **f32),
            ROPE=torch.zeros(n, 64, 2, **f32),
            TEMB=torch.zeros(2 * B, 256, **f32),      # rows 2j / 2j+1: item j's vec / concept_vec chain
            TVAL=torch.zeros(2 * B, **f32),
            YIN=torch.zeros(2 * B, p.vec_in_dim, **f32),
            HVEC=torch.zeros(2 * B, H, **f32),
            VEC=torch.zeros(2 * B, H, **f32),
            MOD=torch.zeros(B, 2, self.weights.mod_rows, **f32),
            LOGITS=torch.zeros(max(C, 1), L_img, **f32))
        if self.precision == "fp8":   # e4m3 images of the GEMM inputs + one fp32 scale per row
            u8 = dict(device=dev, dtype=torch.uint8)
            ws.update(XM8=torch.zeros(n, H, **u8), XMS=torch.zeros(n, **f32),
                      ATT8=torch.zeros(n, H, **u8), ATTS=torch.zeros(n, **f32),
                      HID8=torch.zeros(n, MLP, **u8), HIDS=torch.zeros(n, **f32),
                      CAT8=torch.zeros(B * (T + L_img), H + MLP, **u8), CATS=torch.zeros(B * (T + L_img), **f32))
        return ws

    def _rope_table(self, img_ids, txt_ids, concept_ids, C, T):
        """(cos, sin) per row in [concept | text | image] order.  rope(): angles in float64,
        stored fp32 (flux/math.py:15-22); EmbedND concatenates the axes (layers.py:18-25)."""
        # The table is cached only for ids whose CONTENT is known: tensors made by sampling.make_img_ids /
        # sampling.zero_ids carry a tag (what they hold + their version counter at creation).  A pointer/shape
        # key is not content identity (the allocator re-uses addresses: 2048x512 then 512x2048 would hit), so
        # untagged or since-modified ids rebuild the table on every forward (a few tiny kernels).
        def tag(t):
            g = getattr(t, "_ca_ids_tag", None)
            if g is None:
                return None
            try:
                return g[0] if t._version == g[1] else None
            except RuntimeError:  # inference-mode tensors have no version counter
                return None
        key = tuple(tag(t) for t in (img_ids, txt_ids, concept_ids))
        if any(k is None for k in key):
            key = None
        if key is not None and self._rope_key == key:
            return
        ids = torch.cat((concept_ids.reshape(-1, 3), txt_ids.reshape(-1, 3), img_ids.reshape(-1, 3)), 0) \
            .to(self.device, torch.float64)
        col = 0
        for a, d in enumerate(self.params.axes_dim):
            scale = torch.arange(0, d, 2, dtype=torch.float64, device=self.device) / d
            omega = 1.0 / (self.params.theta ** scale)
            ang = ids[:, a:a + 1] * omega[None]
            self.ROPE[:, col:col + d // 2, 0] = torch.cos(ang).float()
            self.ROPE[:, col:col + d // 2, 1] = torch.sin(ang).float()
            col += d // 2
        self._rope_key = key

    def _mod(self, name: str, row: int, chunk: int, item: int = 0) -> torch.Tensor:
        """fp32 view of one modulation chunk of work item ``item``: name = '<block>.lin' base, row 0 = vec,
        1 = concept_vec.  Consecutive items are ``self._mod_cur.stride(0)`` floats apart (the gate_stride of the
        GEMM epilogue)."""
        H = self.hidden_size
        o = self.weights.mod_offset[name] + chunk * H
        return self._mod_cur[item, row, o:o + H]

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    @on_own_device
    def __call__(self, img, img_ids, txt, txt_ids, concepts, concept_ids, concept_vec, timesteps, y,
                 guidance=None, stop_after_multimodal_attentions: bool = False, edit_metadata=None,
                 iteration=None, joint_attention_kwargs=None, return_vectors: bool = True,
                 heatmaps=None, cond_slot: Optional[int] = None, **kwargs):
        """Same keyword contract as ModifiedFluxDiT.forward (modified_flux_dit.py:75-92), for a batch of B
        independent work items (the reference's callers use B = 1): img (B,L,64), txt (B,T,4096), concepts
        (B,C,4096), y / concept_vec (B,768), timesteps / guidance (B,), ids (B,n,3).

        Extra (HIP-path) keywords: ``return_vectors=False`` skips materialising the four
        per-layer vector stacks (the dict is then empty); ``heatmaps`` (one HeatmapRequest, or a list of B)
        accumulates softmax-over-concepts maps for the requested layers inside the forward; ``cond_slot=i``
        uses step i of a preceding ``precompute_conditioning`` call instead of recomputing the
        conditioning vectors from ``timesteps`` / ``y`` / ``guidance``."""
        assert concept_vec is not None, "Concept vectors must be provided for this implementation."
        if img.ndim != 3 or txt.ndim != 3:
            raise ValueError("Input img and txt tensors must have 3 dimensions.")
        B = img.shape[0]
        if txt.shape[0] != B or concepts.shape[0] != B:
            raise ValueError("img, txt and concepts must have the same batch size")
        if 2 * B > L.ATTN_MAX_PROBLEMS or 3 * B > L.MAX_SEGMENTS:
            raise NotImplementedError(f"HipFluxDiT: at most {min(L.ATTN_MAX_PROBLEMS // 2, L.MAX_SEGMENTS // 3)} work "
                                      "items per forward (attention problems / LayerNorm segments per launch)")
        p, W = self.params, self.weights
        if p.guidance_embed and guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        Li, T, C = img.shape[1], txt.shape[1], concepts.shape[1]
        self._workspace(Li, T, C, B)
        self._rope_table(img_ids, txt_ids, concept_ids, C, T)
        g = _Geom(B, C, T, Li)
        X, XM = self.X, self.XM
        bf = torch.bfloat16
        if heatmaps is not None and not isinstance(heatmaps, (list, tuple)):
            heatmaps = [heatmaps]
        if heatmaps is not None and len(heatmaps) != B:
            raise ValueError("heatmaps: one HeatmapRequest per work item")

        # ---- input embeddings: img_in, txt_in (text and concept tokens share txt_in, :105,120)
        self.TXT_IN[:g.oT].copy_(concepts.reshape(B * C, -1))
        self.TXT_IN[g.oT:].copy_(txt.reshape(B * T, -1))
        img_flat = img.reshape(B * Li, -1)
        split_in = img.dtype == torch.float32 and self.residual_dtype == torch.float32
        if split_in:
            # an fp32 latent (sampling.denoise keeps the Euler state in fp32 on this path): img_in also sees what the
            # bf16 rounding of its operand drops -- the same weights applied to the low plane, accumulated into the
            # fp32 residual stream (K = 64: the second pass costs microseconds)
            img_in = torch.empty(B * Li, img_flat.shape[1], device=img.device, dtype=bf)
            img_lo = torch.empty_like(img_in)
            ops.split_planes(img_flat.contiguous(), img_in, img_lo)
        else:
            img_in = img_flat.to(bf).contiguous()
        self._launch_gemm([ops.Gemm(img_in, W["img_in.weight"], W["img_in.bias"], X[g.oI:]),
                           ops.Gemm(self.TXT_IN, W["txt_in.weight"], W["txt_in.bias"], X[:g.oI])])
        if split_in:
            self._launch_gemm([ops.Gemm(img_lo, W["img_in.weight"], None, X[g.oI:], L.EPI_GATE_RESIDUAL,
                                        resid=X[g.oI:], gate=self._ones_gate(W["img_in.weight"].shape[0]))])

        # ---- conditioning vectors: row 0 = vec (y), row 1 = concept_vec (modified_flux_dit.py:99-119)
        if cond_slot is not None:
            self._mod_cur = self._mod_steps[cond_slot]
            if self._mod_cur.shape[0] != B:
                raise ValueError("precompute_conditioning was run for a different batch size")
        else:
            self._mod_cur = self.MOD
            self._conditioning(timesteps, y, concept_vec, guidance, B)

        out = {k: [] for k in DICT_KEYS} if return_vectors else {}
        for i in range(p.depth):
            self._double_block(i, g, joint_attention_kwargs, out, return_vectors, heatmaps)
        if return_vectors:
            out = {k: torch.stack(v, 0) for k, v in out.items()}
        if stop_after_multimodal_attentions:
            return None, out
        for i in range(p.depth_single_blocks):
            self._single_block(i, g)

        # ---- LastLayer on the image rows (flux/modules/layers.py:242-253)
        fm = "final_layer.adaLN_modulation.1"
        ops.ln_modulate(X[g.oI:], XM[g.oI:], [((j + 1) * Li, self._mod(fm, 0, 0, j), self._mod(fm, 0, 1, j))
                                              for j in range(B)])
        # the prediction has the latent's type: an fp32 latent (sampling.denoise on this path) gets it unrounded
        pred = self.PRED32 if split_in else self.PRED
        self._launch_gemm([ops.Gemm(XM[g.oI:], W["final_layer.linear.weight"], W["final_layer.linear.bias"], pred)])
        return pred.view(B, Li, -1).clone(), out

    def _vec_chain(self, tv, yin, gv, hv, vecs):
        """vec = time_in(t) (+ guidance_in(g)) + vector_in(y) for every row of tv / yin (MLPEmbedders,
        modified_flux_dit.py:99-119), at most 4 vectors per weight pass."""
        p, W = self.params, self.weights
        n = tv.shape[0]
        temb = torch.empty(n, 256, device=self.device, dtype=torch.float32)
        gemb = torch.empty(n, 256, device=self.device, dtype=torch.float32) if p.guidance_embed else None
        for val, emb in ((tv, temb), (gv, gemb)):
            if emb is None:
                continue
            if self.bf16_timesteps:   # bf16(1000 * bf16(t)) as the kernel's argument, embedding rounded to bf16
                arg = (val.to(torch.bfloat16) * 1000.0).float()
                ops.timestep_embedding(arg, emb, time_factor=1.0)
                emb.copy_(emb.to(torch.bfloat16))
            else:
                ops.timestep_embedding(val, emb)
        for r0 in range(0, n, 4):
            r = slice(r0, min(r0 + 4, n))
            ops.gemv(temb[r], W["time_in.in_layer.weight"], W["time_in.in_layer.bias"], hv[r])
            ops.gemv(hv[r], W["time_in.out_layer.weight"], W["time_in.out_layer.bias"], vecs[r], silu_input=True)
            if p.guidance_embed:
                ops.gemv(gemb[r], W["guidance_in.in_layer.weight"], W["guidance_in.in_layer.bias"], hv[r])
                ops.gemv(hv[r], W["guidance_in.out_layer.weight"], W["guidance_in.out_layer.bias"], vecs[r],
                         silu_input=True, accumulate=True)
            ops.gemv(yin[r], W["vector_in.in_layer.weight"], W["vector_in.in_layer.bias"], hv[r])
            ops.gemv(hv[r], W["vector_in.out_layer.weight"], W["vector_in.out_layer.bias"], vecs[r],
                     silu_input=True, accumulate=True)

    def _conditioning(self, timesteps, y, concept_vec, guidance, B=1):
        """vec / concept_vec of every work item and all modulations for ONE step into self.VEC / self.MOD."""
        p = self.params
        t = timesteps.reshape(-1).float()
        self.TVAL.view(B, 2).copy_((t if t.numel() == B else t[:1].expand(B))[:, None].expand(B, 2))
        self.YIN.view(B, 2, -1)[:, 0].copy_(y.reshape(B, -1))
        self.YIN.view(B, 2, -1)[:, 1].copy_(concept_vec.reshape(B, -1))
        gv = None
        if p.guidance_embed:
            gq = guidance.reshape(-1).float()
            gv = (gq if gq.numel() == B else gq[:1].expand(B))[:, None].expand(B, 2).reshape(-1).contiguous()
        self._vec_chain(self.TVAL, self.YIN, gv, self.HVEC, self.VEC)
        self._modulations()

    @on_own_device
    def precompute_conditioning(self, timesteps, y, concept_vec, guidance=None):
        """Conditioning vectors and every block's adaLN modulation for ALL diffusion steps (and all work items of
        a batch: y / concept_vec (B,768)) up front -- they depend only on (t, guidance, y), never on the
        activations: the 6.4 GB of modulation weights are streamed once per 4 vectors instead of once per step.
        Step i is then selected with ``cond_slot=i`` in the model call.  Same arithmetic as the in-call path
        (modified_flux_dit.py:99-119, flux/modules/layers.py:113-126)."""
        p, W, dev = self.params, self.weights, self.device
        if p.guidance_embed and guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        n = len(timesteps)
        y = y.reshape(-1, p.vec_in_dim)
        B = y.shape[0]
        H = p.hidden_size
        f32 = dict(device=dev, dtype=torch.float32)
        # row order: step, item, (vec | concept_vec)
        tv = ops.host_values([float(t) for t in timesteps for _ in range(2 * B)], dev)
        hv = torch.empty(2 * B * n, H, **f32)
        vecs = torch.empty(2 * B * n, H, **f32)
        yin = torch.empty(n, B, 2, p.vec_in_dim, **f32)
        yin[:, :, 0] = y.float()
        yin[:, :, 1] = concept_vec.reshape(B, -1).float()
        yin = yin.view(2 * B * n, -1)
        gv = None
        if p.guidance_embed:
            gq = ops.host_values(guidance, dev).reshape(-1)
            gv = (gq if gq.numel() == B else gq[:1].expand(B))[None, :, None].expand(n, B, 2).reshape(-1).contiguous()
        mod = torch.empty(n, B, 2, W.mod_rows, **f32)
        mod2 = mod.view(2 * B * n, W.mod_rows)
        self._vec_chain(tv, yin, gv, hv, vecs)
        self._modulation_rows(vecs, mod2)
        self._mod_steps = mod
        return n

    def _modulation_rows(self, vecs, mod2):
        """mod2[v] = Modulation.lin(silu(vecs[v])) of every block for all vectors v.  As two bf16 MFMA GEMMs over the
        stacked [sum N, H] weights (silu(vec) split into hi + lo bf16 planes: ~16 mantissa bits; ops.modulation_gemm),
        which stream the 6.4 GB of weights twice in all; where the shape does not fit that kernel, by weight-streaming
        GEMV launches of 4 vectors (4 x 3072 fp32 inputs leave room for 3 workgroups per CU in LDS; with 8 the weight
        stream drops from 5.3 to 2.2 TB/s: tools/gemv_bench.py), i.e. once per 4 vectors."""
        W = self.weights
        if self.modulation_by_gemm:  # (for ANY vector count: the per-call and the all-steps path must round alike)
            if getattr(W, "mod_ones", None) is None:
                W.mod_ones = torch.ones(W.mod_rows, device=self.device, dtype=torch.float32)
            if ops.modulation_gemm(vecs, W.mod_w, W.mod_b, mod2, W.mod_ones):
                return
        for r0 in range(0, vecs.shape[0], 4):
            r = slice(r0, min(r0 + 4, vecs.shape[0]))
            ops.gemv(vecs[r], W.mod_w, W.mod_b, mod2[r], silu_input=True)

    def _modulations(self):
        """Every block's adaLN shift/scale/gate from VEC (rows 2j / 2j+1 = vec / concept_vec of item j) by
        weight-streaming launches of at most 4 vectors (Modulation, flux/modules/layers.py:113-126)."""
        mod2 = self.MOD.view(-1, self.weights.mod_rows)
        self._modulation_rows(self.VEC[:mod2.shape[0]], mod2)
        self._mod_cur = self.MOD

    def _f32_image_vectors(self, capture: bool, heatmaps) -> bool:
        """The attention kernel writes an fp32 copy of the [text | image] output rows in captured layers when the maps
        are reduced on the device (fused heat-map path); A/B: CA_F32_IMAGE_VECTORS=0."""
        return bool(capture and heatmaps is not None and self.f32_image_vectors)

    def _ones_gate(self, n: int) -> torch.Tensor:
        g = getattr(self, "_ones_gate_vec", None)
        if g is None or g.shape[0] != n:
            g = self._ones_gate_vec = torch.ones(n, device=self.device, dtype=torch.float32)
        return g

    def _qk16(self, capture: bool) -> bool:
        return self.prescale_q and (self.qk_f16 == "all" or (self.qk_f16 == "captured" and bool(capture)))

    def _q_out_scale(self) -> float:
        """What the qkv epilogue multiplies the rotated q by: softmax_scale * log2(e) (head_dim 128), or 0 (= 1)."""
        return (1.0 / math.sqrt(128.0)) * 1.4426950408889634 if self.prescale_q else 0.0

    def _double_block(self, i, g, joint_attention_kwargs=None, out=None, return_vectors=False, heatmaps=None):
        """ModifiedDoubleStreamBlock.forward (modified_double_stream_block.py:69-204) on the resident X rows
        [concepts | text | image] of all work items; 7 launches."""
        p, W = self.params, self.weights
        H, NH = p.hidden_size, p.num_heads
        B, C, T, Li, oT, oI, n = g.B, g.C, g.T, g.L, g.oT, g.oI, g.n
        X, XM, QKV, ATT, HID = self.X, self.XM, self.QKV, self.ATT, self.HID
        qs, ks, vs = QKV[:, :H], QKV[:, H:2 * H], QKV[:, 2 * H:]
        cross = self_ = True
        if joint_attention_kwargs is not None:
            cross = joint_attention_kwargs.get("concept_cross_attention", True)
            self_ = joint_attention_kwargs.get("concept_self_attention", True)
        b = f"double_blocks.{i}."
        im, tm = b + "img_mod.lin", b + "txt_mod.lin"
        capture = return_vectors or (heatmaps is not None and any(i in h.layer_indices for h in heatmaps))
        fp8 = self.precision == "fp8" and i not in self.keep_bf16_layers
        # fp8 mode: the qkv projection of a layer whose maps are requested stays bf16 (round 4).  Its q / k / v ARE the
        # vectors of that layer's maps, and e4m3's 3 mantissa bits on their GEMM operands cost a map 2-4e-2 against the
        # fp32 oracle (tests/test_full_depth_gpu.py::test_fp8_forward_full_size_vs_fp32_oracle); proj and the MLP reach a
        # map only through the residual stream.  25 % of a double block's projection FLOPs.
        fp8_qkv = fp8 and not (capture and self.fp8_bf16_qkv_when_captured)
        if fp8:
            XM8, XMS, ATT8, ATTS, HID8, HIDS = self.XM8, self.XMS, self.ATT8, self.ATTS, self.HID8, self.HIDS
            xm_out = dict(out=XM8, out_scale=XMS)
        else:
            XM8 = XMS = ATT8 = ATTS = HID8 = HIDS = None
            xm_out = dict(out=XM)
        xm_out_qkv = xm_out if fp8_qkv else dict(out=XM)

        def rows(t, lo, hi):
            return None if t is None else t[lo:hi]

        def segs(sh, sc):
            """LayerNorm-modulate segments in row order: concept rows (concept_vec), text rows, image rows of
            every item; sh / sc = chunk index of shift / scale in the block's modulation."""
            return ([((j + 1) * C, self._mod(tm, 1, sh, j), self._mod(tm, 1, sc, j)) for j in range(B)] +
                    [(oT + (j + 1) * T, self._mod(tm, 0, sh, j), self._mod(tm, 0, sc, j)) for j in range(B)] +
                    [(oI + (j + 1) * Li, self._mod(im, 0, sh, j), self._mod(im, 0, sc, j)) for j in range(B)])

        gs = 0 if B == 1 else self._mod_cur.stride(0)   # floats between consecutive items' gate vectors
        G = self._gemm
        # K4: LayerNorm + (1+scale)*x+shift, per row range and item (:88-89,94-95,100-101)
        split = capture and not fp8_qkv and self.split_q_capture and self.residual_dtype == torch.float32 and C > 0
        ops.ln_modulate(X, segments=[sg for sg in segs(0, 1) if sg[0] > 0], **xm_out_qkv,
                        **({"out_lo": self.XML} if split else {}))
        # K5+K6+K7: qkv projections (image stream + [concept|text] stream in one grouped launch) with
        # QK-RMSNorm and RoPE fused into the epilogue; pre-RoPE q kept for the cross-attention maps
        qpre = self.QPRE if capture else None
        # (capture_independent_image: q / k exactly as in an uncaptured layer; the accurate q goes to QACC instead)
        indep = bool(split and self.capture_independent_image and self.split_q_attention)
        qk16 = self._qk16(capture and not self.capture_independent_image)
        self._layer_indep = indep
        def qkv_launch():
            self._launch_gemm([G(fp8_qkv, XM[oI:], rows(XM8, oI, n), rows(XMS, oI, n), b + "img_attn.qkv.weight",
                                 W.tensors.get(b + "img_attn.qkv.bias"), QKV[oI:],
                                 L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=W[b + "img_attn.norm.query_norm.scale"],
                                 norm_k=W[b + "img_attn.norm.key_norm.scale"], rope=self.ROPE[oI:],
                                 q_prerope=None if qpre is None else qpre[oI:], q_out_scale=self._q_out_scale(),
                                 qpre_raw=split, qk_f16=qk16),
                               G(fp8_qkv, XM[:oI], rows(XM8, 0, oI), rows(XMS, 0, oI), b + "txt_attn.qkv.weight",
                                 W.tensors.get(b + "txt_attn.qkv.bias"), QKV[:oI],
                                 L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=W[b + "txt_attn.norm.query_norm.scale"],
                                 norm_k=W[b + "txt_attn.norm.key_norm.scale"], rope=self.ROPE[:oI],
                                 q_prerope=None if qpre is None else qpre[:oI], q_out_scale=self._q_out_scale(),
                                 qpre_raw=split, qk_f16=qk16)])
        # (256 x 256 tiles named for the low-plane launches: the automatic choice prices the concept rows' problem and
        # lands on 256 x 128, 376 vs 332 us per 5-item launch)
        lo_tile = L.TILE_PP_256x256 if B * Li >= 4096 else L.TILE_AUTO
        if split and self.split_q_attention:
            # (round 5) the q weights applied to the low plane of y -- image rows and concept rows; text rows are not
            # captured -- with the correction, the RMS norm, the rotation and the store of the ATTENTION's q fused into
            # that launch's epilogue (qpre_f32 = 3): the main launch leaves the hi plane's raw q projection in QPRE, this
            # one adds its own product, normalises (QPRE <- the cross-attention-space vectors, fp32) and writes the
            # rotated, scaled q over what the main epilogue formed from bf16(y).  A single output-space map: 7.8e-4 ->
            # 2.5e-4 from the fp32 oracle (the operand's rounding reaches the map through q; tests/tools/diag_out_space.py).
            # Rounds 3-4 had a second GEMM output (QD) and ca_qpre_finish_rope_f32 for it: 18 instead of 10 bytes per
            # element past the GEMM and 2 launches more per captured layer.
            qkv_launch()
            q_acc = self.QACC if indep else qs
            lo = dict(epilogue=L.EPI_QKV_NORM_ROPE, n_split=3 * H, q_out_scale=self._q_out_scale(), qpre_add=True,
                      qk_f16=qk16)
            ops.gemm([ops.Gemm(self.XML[oI:], W[b + "img_attn.qkv.weight"][:H], None, q_acc[oI:],
                               norm_q=W[b + "img_attn.norm.query_norm.scale"], norm_k=W[b + "img_attn.norm.key_norm.scale"],
                               rope=self.ROPE[oI:], q_prerope=qpre[oI:], **lo),
                      ops.Gemm(self.XML[:oT], W[b + "txt_attn.qkv.weight"][:H], None, q_acc[:oT],
                               norm_q=W[b + "txt_attn.norm.query_norm.scale"], norm_k=W[b + "txt_attn.norm.key_norm.scale"],
                               rope=self.ROPE[:oT], q_prerope=qpre[:oT], **lo)], L.TILE_PP_256x256)
        elif split:
            # split_q_attention = False (A/B aid): only the cross-attention-space vectors take the correction; the
            # attention's q stays the main epilogue's (rounds 3's route: second GEMM output + finish kernel)
            ops.gemm([ops.Gemm(self.XML[oI:], W[b + "img_attn.qkv.weight"][:H], None, self.QD[oI:]),
                      ops.Gemm(self.XML[:oT], W[b + "txt_attn.qkv.weight"][:H], None, self.QD[:oT])], lo_tile)
            qkv_launch()
            ops.qpre_finish(qpre[oI:], self.QD[oI:], W[b + "img_attn.norm.query_norm.scale"], NH)
            ops.qpre_finish(qpre[:oT], self.QD[:oT], W[b + "txt_attn.norm.query_norm.scale"], NH)
        else:
            qkv_launch()
        # K8+K9: per item, joint text+image attention and the concept rows; one launch (concept problems first)
        f32img = self._f32_image_vectors(capture, heatmaps)
        # per-head partial logits from the attention epilogue instead of fp32 rows (self.epilogue_logits above)
        use_part = bool(capture and heatmaps is not None and not indep and self.epilogue_logits and self.fused_heatmaps
                        and self.prescale_q and 1 <= C <= 8 and ops.heatmap_fused_fits(C, H))
        self._layer_part = use_part
        if use_part:
            f32img = False

        def concept_problem(j, q_rows, out_rows, out32):
            cj, ij = slice(j * C, (j + 1) * C), slice(oI + j * Li, oI + (j + 1) * Li)
            if cross and self_:
                return ops.Attn(q_rows[cj], out_rows[cj], ks[cj], vs[cj], ks[ij], vs[ij], out_f32=out32)
            if cross:   # :129-138 image keys/values only
                return ops.Attn(q_rows[cj], out_rows[cj], ks[ij], vs[ij], out_f32=out32)
            return ops.Attn(q_rows[cj], out_rows[cj], ks[cj], vs[cj], out_f32=out32)   # :139-147 concept keys/values only
        probs = []
        if C > 0 and (cross or self_):
            for j in range(B):   # (independent mode: the rounded-q rows feed proj; the fp32 rows of the maps come below)
                probs.append(concept_problem(j, qs, ATT, None if indep else self.ATT32[j * C:(j + 1) * C]))
        if C > 0 and not (cross or self_):   # :157-159 concept_attn = concept_v
            ATT[:oT].copy_(vs[:oT])
            self.ATT32[:oT].copy_(vs[:oT])
        if use_part and probs:
            # the concept rows first, in their own launch (102 us for a 5-item layer): the main problems' epilogues read
            # ATT32.  (A bandwidth-style kernel for these <= 8-row problems -- keys split over workgroups, fp32 scores --
            # was built and measured at the same 100 us and the same maps: removed again)
            ops.attention(probs, NH, q_prescaled=self.prescale_q, qk_f16=qk16)
            probs = []
        for j in range(B):
            tj, ij = slice(oT + j * T, oT + (j + 1) * T), slice(oI + j * Li, oI + (j + 1) * Li)
            hm_kw = dict(hm_con=self.ATT32[j * C:(j + 1) * C], hm_part=self.PART[j]) if use_part else {}
            probs.append(ops.Attn(qs[tj], ATT[tj], ks[tj], vs[tj], ks[ij], vs[ij], q1=qs[ij], out1=ATT[ij],
                                  out_f32=self.ATTI32[j] if (f32img and not indep) else None, **hm_kw))
        if indep:
            # the maps' side: the accurate q of the image rows against the same keys / values, into rows that only the
            # heat maps read (one more problem per item in this launch) ...
            for j in range(B):
                tj, ij = slice(oT + j * T, oT + (j + 1) * T), slice(oI + j * Li, oI + (j + 1) * Li)
                probs.append(ops.Attn(self.QACC[ij], self.ATTM[ij], ks[tj], vs[tj], ks[ij], vs[ij],
                                      out_f32=self.ATTI32[j, T:] if f32img else None))
        ops.attention(probs[:L.ATTN_MAX_PROBLEMS], NH, q_prescaled=self.prescale_q, qk_f16=qk16)
        if len(probs) > L.ATTN_MAX_PROBLEMS:
            ops.attention(probs[L.ATTN_MAX_PROBLEMS:], NH, q_prescaled=self.prescale_q, qk_f16=qk16)
        if indep and C > 0 and (cross or self_):
            # ... and of the concept rows (B tiny problems: their own launch, the first one is full at 5 items)
            ops.attention([concept_problem(j, self.QACC, self.ATTM, self.ATT32[j * C:(j + 1) * C]) for j in range(B)],
                          NH, q_prescaled=self.prescale_q, qk_f16=qk16)
        if capture:
            self._capture(out, i, g, NH, return_vectors, heatmaps)
        if fp8:
            ops.quantize_rows_fp8(ATT, ATT8, ATTS)
        # K12: proj + gated residual (:194,198,201); concept rows use txt weights + concept gate
        self._launch_gemm([G(fp8, ATT[oI:], rows(ATT8, oI, n), rows(ATTS, oI, n), b + "img_attn.proj.weight",
                             W[b + "img_attn.proj.bias"], X[oI:],
                             L.EPI_GATE_RESIDUAL, resid=X[oI:], gate=self._mod(im, 0, 2), gate_stride=gs,
                             gate_item_rows=Li),
                           G(fp8, ATT[:oI], rows(ATT8, 0, oI), rows(ATTS, 0, oI), b + "txt_attn.proj.weight",
                             W[b + "txt_attn.proj.bias"], X[:oI],
                             L.EPI_GATE_RESIDUAL, resid=X[:oI], gate=self._mod(tm, 1, 2), gate2=self._mod(tm, 0, 2),
                             gate_rows=oT, gate_stride=gs, gate_item_rows=max(C, 1), gate2_item_rows=T)])
        # K13: LN + modulate + MLP + gated residual (:196,199,202)
        ops.ln_modulate(X, segments=[sg for sg in segs(3, 4) if sg[0] > 0], **xm_out)
        self._launch_gemm([G(fp8, XM[oI:], rows(XM8, oI, n), rows(XMS, oI, n), b + "img_mlp.0.weight",
                             W[b + "img_mlp.0.bias"], HID[oI:], L.EPI_GELU_TANH),
                           G(fp8, XM[:oI], rows(XM8, 0, oI), rows(XMS, 0, oI), b + "txt_mlp.0.weight",
                             W[b + "txt_mlp.0.bias"], HID[:oI], L.EPI_GELU_TANH)])
        if fp8:
            ops.quantize_rows_fp8(HID, HID8, HIDS)
        self._launch_gemm([G(fp8, HID[oI:], rows(HID8, oI, n), rows(HIDS, oI, n), b + "img_mlp.2.weight",
                             W[b + "img_mlp.2.bias"], X[oI:],
                             L.EPI_GATE_RESIDUAL, resid=X[oI:], gate=self._mod(im, 0, 5), gate_stride=gs,
                             gate_item_rows=Li),
                           G(fp8, HID[:oI], rows(HID8, 0, oI), rows(HIDS, 0, oI), b + "txt_mlp.2.weight",
                             W[b + "txt_mlp.2.bias"], X[:oI],
                             L.EPI_GATE_RESIDUAL, resid=X[:oI], gate=self._mod(tm, 1, 5), gate2=self._mod(tm, 0, 5),
                             gate_rows=oT, gate_stride=gs, gate_item_rows=max(C, 1), gate2_item_rows=T)])

    def _single_block(self, i, g):
        """ModifiedSingleStreamBlock.forward (modified_single_stream_block.py:43-56) on the [text | image] rows of
        all work items; 4 launches."""
        p, W = self.params, self.weights
        H, NH = p.hidden_size, p.num_heads
        B, T, Li, oT, oI = g.B, g.T, g.L, g.oT, g.oI
        xs, xms, qkvs, CAT = self.X[oT:], self.XM[oT:], self.QKV[oT:], self.CAT
        nT = B * T                       # rows of xs that are text; image rows follow
        b = f"single_blocks.{i}."
        m = b + "modulation.lin"
        fp8 = self.precision == "fp8"
        G = self._gemm
        segs = ([((j + 1) * T, self._mod(m, 0, 0, j), self._mod(m, 0, 1, j)) for j in range(B)] +
                [(nT + (j + 1) * Li, self._mod(m, 0, 0, j), self._mod(m, 0, 1, j)) for j in range(B)])
        if fp8:
            xm8, xms8 = self.XM8[oT:], self.XMS[oT:]
            ops.ln_modulate(xs, xm8, segs, out_scale=xms8)
        else:
            xm8 = xms8 = None
            ops.ln_modulate(xs, xms, segs)
        self._launch_gemm([G(fp8, xms, xm8, xms8, b + "linear1.weight", W[b + "linear1.bias"], qkvs,
                             L.EPI_QKV_NORM_ROPE, out2=CAT[:, H:], n_split=3 * H, norm_q=W[b + "norm.query_norm.scale"],
                             norm_k=W[b + "norm.key_norm.scale"], rope=self.ROPE[oT:],
                             q_out_scale=self._q_out_scale(), qk_f16=self._qk16(False))])
        qh, kh, vh, oh = qkvs[:, :H], qkvs[:, H:2 * H], qkvs[:, 2 * H:], CAT[:, :H]
        probs = []
        for j in range(B):
            tj, ij = slice(j * T, (j + 1) * T), slice(nT + j * Li, nT + (j + 1) * Li)
            probs.append(ops.Attn(qh[tj], oh[tj], kh[tj], vh[tj], kh[ij], vh[ij], q1=qh[ij], out1=oh[ij]))
        ops.attention(probs, NH, q_prescaled=self.prescale_q, qk_f16=self._qk16(False))
        if fp8:
            ops.quantize_rows_fp8(CAT, self.CAT8, self.CATS)
        gate = self._mod(m, 0, 2)
        self._launch_gemm([G(fp8, CAT, self.CAT8 if fp8 else None, self.CATS if fp8 else None, b + "linear2.weight",
                             W[b + "linear2.bias"], xs, L.EPI_GATE_RESIDUAL, resid=xs, gate=gate, gate2=gate,
                             gate_rows=nT, gate_stride=0 if B == 1 else self._mod_cur.stride(0),
                             gate_item_rows=T, gate2_item_rows=Li)])

    forward = __call__

    def _capture(self, out, layer, g, NH, return_vectors, heatmaps):
        """Dict capture (modified_double_stream_block.py:185-191) and/or fused heat-map update, per work item."""
        ATT, QPRE = self.ATT, self.QPRE
        B, C, Li, oT, oI = g.B, g.C, g.L, g.oT, g.oI
        fused = self.fused_heatmaps and ops.heatmap_fused_fits(C, self.hidden_size)
        launches = {}   # norm -> problems of this layer: ONE launch for all work items and both spaces
        for j in range(B if heatmaps is not None else 0):
            hm = heatmaps[j]
            if layer not in hm.layer_indices:
                continue
            # output space: the C concept rows come from the fp32 copy the attention kernel wrote
            # (their bf16 rounding, multiplied by the large component all attention outputs share,
            # is the dominant heat-map error otherwise -- DESIGN.md "tolerance")
            li = hm.layer_indices.index(layer)
            cj, ij = slice(j * C, (j + 1) * C), slice(oI + j * Li, oI + (j + 1) * Li)
            part = self._layer_part   # (_layer_part / _layer_indep: set by _double_block for this layer)
            img_out = None if part else (self.ATTI32[j, g.T:] if self._f32_image_vectors(True, heatmaps) else
                                         (self.ATTM[ij] if self._layer_indep else ATT[ij]))
            for img_vec, con_vec, acc, table in ((img_out, self.ATT32[cj], hm.out_space, hm.per_layer_out),
                                                 (QPRE[ij], QPRE[cj], hm.cross_space, hm.per_layer_cross)):
                if acc is None and table is None:
                    continue
                if fused:
                    launches.setdefault(hm.norm, []).append(ops.Heatmap(
                        img_vec, None if img_vec is None else con_vec, acc, hm.weight,
                        None if table is None else table[li], hm.per_layer_weight,
                        part=self.PART[j] if img_vec is None else None))
                    continue
                # the three-launch form (more than 8 concepts, or fused_heatmaps = False: the A/B and parity aid)
                ops.heatmap_logits(img_vec, con_vec, self.LOGITS[:C])
                if acc is not None:
                    ops.heatmap_softmax_accumulate(self.LOGITS[:C], acc, hm.weight, hm.norm)
                if table is not None:
                    ops.heatmap_softmax_accumulate(self.LOGITS[:C], table[li], hm.per_layer_weight, hm.norm)
        for norm, probs in launches.items():
            # one launch per norm; problems that name an accumulator already in the launch (two requests sharing a
            # tensor) start a new one: within a launch the updates are unordered read-modify-writes
            batch, seen = [], set()
            for pr in probs + [None]:
                ptrs = set() if pr is None else {t.data_ptr() for t in (pr.acc, pr.acc2) if t is not None}
                if pr is None or len(batch) == L.HEATMAP_MAX_PROBLEMS or (ptrs & seen):
                    if batch:
                        ops.heatmap_fused(batch, norm)
                    batch, seen = [], set()
                if pr is not None:
                    batch.append(pr)
                    seen |= ptrs
        if return_vectors:
            H = self.hidden_size
            cf = torch.contiguous_format
            out["output_space_concept_vectors"].append(ATT[:oT].view(B, C, H).clone())
            out["output_space_image_vectors"].append(ATT[oI:].view(B, Li, H).clone())
            # the reference stores post-QKNorm, pre-RoPE q per head: [B, heads, tokens, 128]; in the
            # self-attention-only ablation it rebinds concept_q to the post-RoPE tensor (:140), which
            # for the all-zero concept ids is the same values.  (clone, not .contiguous(): for C = 1 the permuted
            # view already counts as contiguous and would keep aliasing QPRE, which the next block overwrites)
            # (QPRE is fp32 on the device; the dict carries the activations' dtype, as the reference's does)
            out["cross_attention_concept_vectors"].append(
                QPRE[:oT].view(B, C, NH, 128).permute(0, 2, 1, 3).to(torch.bfloat16, memory_format=cf, copy=True))
            out["cross_attention_image_vectors"].append(
                QPRE[oI:].view(B, Li, NH, 128).permute(0, 2, 1, 3).to(torch.bfloat16, memory_format=cf, copy=True))
