"""Tensor-level wrappers over the C ABI (include/conceptattn.h).

PyTorch is used only for device memory and the stream; every operator below runs a hand-written
gfx950 kernel from libconceptattn.so.  Arguments are validated here for dtype/device and in the
library for shapes/alignment (ValueError on a rejected argument, nothing is launched).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib as L


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA(HIP) tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")
    return t


def host_values(values, device, dtype=torch.float32) -> torch.Tensor:
    """A few host numbers (timesteps, guidance) as a device tensor WITHOUT blocking the host: ``torch.tensor(list,
    device=...)`` copies from pageable memory, which waits for everything queued on the stream before it -- in a loop of
    forwards the host then cannot run ahead and the GPU idles while each forward's first launches are enqueued.  A pinned
    staging tensor and a stream-ordered copy keep the queue full (the caching host allocator recycles the staging block
    only after the copy has run)."""
    t = torch.as_tensor(values, dtype=dtype)
    if t.device.type != "cpu":
        return t.to(device)
    return t.reshape(-1).pin_memory().to(device, non_blocking=True).reshape(t.shape)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


@dataclass
class Gemm:
    """One problem of a grouped GEMM launch: out = epi(a @ w.T + bias)."""
    a: torch.Tensor                      # [M,K] bf16 (row stride free)
    w: torch.Tensor                      # [N,K] bf16
    bias: Optional[torch.Tensor]         # [N] bf16
    out: torch.Tensor                    # [M,N] bf16 (row stride free)
    epilogue: int = L.EPI_BIAS
    resid: Optional[torch.Tensor] = None  # [M,N] bf16, may be `out`
    gate: Optional[torch.Tensor] = None   # [N] fp32
    gate2: Optional[torch.Tensor] = None  # [N] fp32 for rows >= gate_rows
    gate_rows: Optional[int] = None
    out2: Optional[torch.Tensor] = None   # SPLIT_GELU / QKV_NORM_ROPE second output [M,N-n_split]
    n_split: int = 0
    norm_q: Optional[torch.Tensor] = None  # QKV_NORM_ROPE: bf16 [128] scales, fp32 [M,64,2] rope table,
    norm_k: Optional[torch.Tensor] = None  # optional bf16 [M, heads*128] pre-RoPE q output
    rope: Optional[torch.Tensor] = None
    q_prerope: Optional[torch.Tensor] = None   # bf16 or fp32 [M, heads*128]
    q_out_scale: float = 0.0                   # QKV_NORM_ROPE: rotated q times this before rounding (0 = 1)
    qpre_raw: bool = False                     # fp32 q_prerope receives the projection BEFORE its norm (qpre_finish)
    qpre_add: bool = False                     # fp32 q_prerope HOLDS such a raw projection: added before the norm, then
                                               # overwritten with the normalised vector (qpre_f32 = 3; N = n_split / 3)
    qk_f16: bool = False                       # QKV_NORM_ROPE: rotated q / k stored as fp16 bits (attention(qk_f16=True))
    a_scale: Optional[torch.Tensor] = None  # fp8 mode: a, w are uint8 (e4m3 bytes) with fp32 row scales
    w_scale: Optional[torch.Tensor] = None  # ([M] and [N]); the launch then goes to ca_gemm_fp8
    # batched forward: rows < gate_rows are items of gate_item_rows rows, the others items of gate2_item_rows rows;
    # item i of a range uses gate (gate2) + i * gate_stride floats.  gate_stride = 0: one vector per range.
    gate_stride: int = 0
    gate_item_rows: int = 0
    gate2_item_rows: int = 0


def gemm(problems: Sequence[Gemm], tile: int = L.TILE_AUTO) -> None:
    lib = L.load()
    arr = (L.GemmProblem * len(problems))()
    fp8 = problems[0].a.dtype == torch.uint8
    op_dtype = torch.uint8 if fp8 else torch.bfloat16
    for i, g in enumerate(problems):
        f32o = g.out.dtype == torch.float32  # the fp32 residual stream (BIAS / GATE_RESIDUAL epilogues)
        a, w, out = _chk(g.a, op_dtype, "a"), _chk(g.w, op_dtype, "w"), _chk(g.out, g.out.dtype if f32o else torch.bfloat16, "out")
        if fp8:
            if g.a_scale is None or g.w_scale is None:
                raise ValueError(f"gemm[{i}]: fp8 operands need a_scale and w_scale")
            sa, sw = _chk(g.a_scale, torch.float32, "a_scale"), _chk(g.w_scale, torch.float32, "w_scale")
            if sa.numel() != a.shape[0] or sw.numel() != w.shape[0] or not (sa.is_contiguous() and sw.is_contiguous()):
                raise ValueError(f"gemm[{i}]: a_scale must be contiguous [M], w_scale contiguous [N]")
            arr[i].a_scale, arr[i].w_scale = sa.data_ptr(), sw.data_ptr()
        if a.dim() != 2 or w.dim() != 2 or out.dim() != 2 or a.shape[1] != w.shape[1] or a.shape[0] != out.shape[0]:
            raise ValueError(f"gemm[{i}]: shape mismatch a{tuple(a.shape)} w{tuple(w.shape)} out{tuple(out.shape)}")
        p = arr[i]
        p.A, p.W, p.out = a.data_ptr(), w.data_ptr(), out.data_ptr()
        p.bias = _ptr(None if g.bias is None else _chk(g.bias, torch.bfloat16, "bias"))
        p.M, p.N, p.K = a.shape[0], w.shape[0], a.shape[1]
        p.lda, p.ldw, p.ldc = a.stride(0), w.stride(0), out.stride(0)
        p.epilogue = g.epilogue
        p.out_f32 = int(f32o)
        p.gate_rows = p.M if g.gate_rows is None else g.gate_rows
        if g.epilogue == L.EPI_GATE_RESIDUAL:
            if g.resid is None or g.gate is None:
                raise ValueError(f"gemm[{i}]: GATE_RESIDUAL needs resid and gate")
            p.resid, p.ldr = _chk(g.resid, out.dtype, "resid").data_ptr(), g.resid.stride(0)
            p.gate = _chk(g.gate, torch.float32, "gate").data_ptr()
            p.gate2 = _ptr(None if g.gate2 is None else _chk(g.gate2, torch.float32, "gate2"))
            p.gate_stride, p.gate_item_rows, p.gate2_item_rows = g.gate_stride, g.gate_item_rows, g.gate2_item_rows
        elif g.epilogue == L.EPI_QKV_NORM_ROPE:
            if g.norm_q is None or g.norm_k is None or g.rope is None:
                raise ValueError(f"gemm[{i}]: QKV_NORM_ROPE needs norm_q, norm_k, rope")
            p.norm_q = _chk(g.norm_q, torch.bfloat16, "norm_q").data_ptr()
            p.norm_k = _chk(g.norm_k, torch.bfloat16, "norm_k").data_ptr()
            rope = _chk(g.rope, torch.float32, "rope")
            if tuple(rope.shape) != (p.M, 64, 2) or not rope.is_contiguous():
                raise ValueError(f"gemm[{i}]: rope must be contiguous [M,64,2]")
            p.rope, p.n_split = rope.data_ptr(), g.n_split
            p.q_out_scale = float(g.q_out_scale)
            p.qk_f16 = int(bool(g.qk_f16))
            if g.q_prerope is not None:
                if g.q_prerope.dtype not in (torch.bfloat16, torch.float32):
                    raise ValueError(f"gemm[{i}]: q_prerope must be bf16 or fp32")
                p.q_prerope, p.ldp = _chk(g.q_prerope, g.q_prerope.dtype, "q_prerope").data_ptr(), g.q_prerope.stride(0)
                p.qpre_f32 = int(g.q_prerope.dtype == torch.float32)
                if g.qpre_raw or g.qpre_add:
                    if not p.qpre_f32 or (g.qpre_raw and g.qpre_add):
                        raise ValueError(f"gemm[{i}]: qpre_raw / qpre_add (one of them) need an fp32 q_prerope")
                    p.qpre_f32 = 2 if g.qpre_raw else 3
            if g.out2 is not None:
                p.out2, p.ld2 = _chk(g.out2, torch.bfloat16, "out2").data_ptr(), g.out2.stride(0)
        elif g.epilogue == L.EPI_SPLIT_GELU:
            if g.out2 is None:
                raise ValueError(f"gemm[{i}]: SPLIT_GELU needs out2")
            p.out2, p.ld2, p.n_split = _chk(g.out2, torch.bfloat16, "out2").data_ptr(), g.out2.stride(0), g.n_split
    if fp8:
        if tile not in (L.TILE_AUTO, L.TILE_PP_256x256):
            raise ValueError("gemm: fp8 operands run on the 256x256 ping-pong tile only")
        if _gemm_hook is not None:
            _gemm_hook(arr, L.TILE_PP_256x256,
                       lambda: L.check(lib.ca_gemm_fp8(arr, len(problems), _stream()), "ca_gemm_fp8"))
            return
        L.check(lib.ca_gemm_fp8(arr, len(problems), _stream()), "ca_gemm_fp8")
        return
    if _gemm_hook is not None:
        if tile == L.TILE_AUTO:
            tile = lib.ca_gemm_auto_tile(arr, len(problems))
            if tile <= 0:
                L.check(tile, "ca_gemm_auto_tile")
        _gemm_hook(arr, tile, lambda: L.check(lib.ca_gemm_bf16(arr, len(problems), tile, _stream()), "ca_gemm_bf16"))
        return
    L.check(lib.ca_gemm_bf16(arr, len(problems), tile, _stream()), "ca_gemm_bf16")


# Optional instrumentation used by bench.py: hook(problem_array, tile, launch) must call launch().
_gemm_hook = None


def set_gemm_hook(hook) -> None:
    global _gemm_hook
    _gemm_hook = hook


def linear(a, w, bias, out=None, epilogue=L.EPI_BIAS, tile=L.TILE_AUTO, **kw):
    if out is None:
        out = torch.empty(a.shape[0], w.shape[0], device=a.device, dtype=torch.bfloat16)
    gemm([Gemm(a, w, bias, out, epilogue, **kw)], tile)
    return out


@dataclass
class Attn:
    """out = softmax(q k^T * scale) v per head; keys/values = rows of (k0,v0) then rows of (k1,v1).
    q/k/v are 2-D views [rows, num_heads*128] (row stride free), head h at columns h*128.."""
    q: torch.Tensor
    out: torch.Tensor
    k0: torch.Tensor
    v0: torch.Tensor
    k1: Optional[torch.Tensor] = None
    v1: Optional[torch.Tensor] = None
    out_f32: Optional[torch.Tensor] = None  # optional fp32 copy of the output rows
    q1: Optional[torch.Tensor] = None       # second query-row segment (its rows follow q's) and its output rows
    out1: Optional[torch.Tensor] = None
    # per-head partial heat-map logits of the second segment's rows (q_prescaled kernels): hm_con fp32 [C <= 8, heads*128]
    # = the concept rows' attention outputs (complete before this launch), hm_part fp32 [heads, rows of q1, 8]
    hm_con: Optional[torch.Tensor] = None
    hm_part: Optional[torch.Tensor] = None


def attention(problems: Sequence[Attn], num_heads: int, scale: Optional[float] = None,
              q_prescaled: bool = False, qk_f16: bool = False) -> None:
    """``q_prescaled``: the q rows already carry softmax_scale * log2(e) (Gemm.q_out_scale; CA_ATTN_Q_PRESCALED).
    ``qk_f16`` (with q_prescaled): the q and k rows hold IEEE half bits in their bf16-typed tensors (Gemm.qk_f16)."""
    lib = L.load()
    arr = (L.AttnProblem * len(problems))()
    for i, a in enumerate(problems):
        for n in ("q", "out", "k0", "v0"):
            _chk(getattr(a, n), torch.bfloat16, n)
        p = arr[i]
        p.q, p.out, p.k0, p.v0 = a.q.data_ptr(), a.out.data_ptr(), a.k0.data_ptr(), a.v0.data_ptr()
        p.nq, p.n0 = a.q.shape[0], a.k0.shape[0]
        p.ldq, p.ldo, p.ldkv = a.q.stride(0), a.out.stride(0), a.k0.stride(0)
        if a.v0.stride(0) != p.ldkv or a.v0.shape[0] != p.n0 or a.out.shape[0] != p.nq:
            raise ValueError(f"attention[{i}]: k0/v0/out row mismatch")
        p.nq0 = p.nq
        if a.q1 is not None and a.q1.shape[0] > 0:
            _chk(a.q1, torch.bfloat16, "q1"), _chk(a.out1, torch.bfloat16, "out1")
            if a.q1.stride(0) != p.ldq or a.out1.stride(0) != p.ldo or a.out1.shape[0] != a.q1.shape[0]:
                raise ValueError(f"attention[{i}]: both query / output segments must share one row stride")
            p.q1, p.out1 = a.q1.data_ptr(), a.out1.data_ptr()
            p.nq = p.nq0 + a.q1.shape[0]
        if a.k1 is not None and a.k1.shape[0] > 0:
            _chk(a.k1, torch.bfloat16, "k1"), _chk(a.v1, torch.bfloat16, "v1")
            if a.k1.stride(0) != p.ldkv or a.v1.stride(0) != p.ldkv or a.v1.shape[0] != a.k1.shape[0]:
                raise ValueError(f"attention[{i}]: both key/value segments must share one row stride")
            p.k1, p.v1, p.n1 = a.k1.data_ptr(), a.v1.data_ptr(), a.k1.shape[0]
        if a.hm_con is not None or a.hm_part is not None:
            if a.hm_con is None or a.hm_part is None or a.q1 is None:
                raise ValueError(f"attention[{i}]: hm_con, hm_part and a second query segment go together")
            _chk(a.hm_con, torch.float32, "hm_con"), _chk(a.hm_part, torch.float32, "hm_part")
            if a.hm_con.dim() != 2 or a.hm_con.shape[1] != num_heads * 128 or not 1 <= a.hm_con.shape[0] <= 8 or \
                    tuple(a.hm_part.shape) != (num_heads, a.q1.shape[0], 8) or not a.hm_part.is_contiguous():
                raise ValueError(f"attention[{i}]: hm_con must be fp32 [C <= 8, heads*128], hm_part contiguous fp32 "
                                 "[heads, rows of q1, 8]")
            p.hm_con, p.hm_part = a.hm_con.data_ptr(), a.hm_part.data_ptr()
            p.hm_C, p.ldhc = a.hm_con.shape[0], a.hm_con.stride(0)
        if a.out_f32 is not None:
            _chk(a.out_f32, torch.float32, "out_f32")
            if a.out_f32.shape[0] != p.nq:
                raise ValueError(f"attention[{i}]: out_f32 row mismatch")
            p.out_f32, p.ldo32 = a.out_f32.data_ptr(), a.out_f32.stride(0)
    if qk_f16 and not q_prescaled:
        raise ValueError("attention: qk_f16 needs q_prescaled (ca_attn_fwd_qk16)")
    if q_prescaled:
        if scale is not None:
            raise ValueError("attention: give either scale or q_prescaled")
        scale = L.ATTN_Q_PRESCALED
    elif scale is None:
        scale = 1.0 / math.sqrt(128.0)
    elif not scale > 0:
        raise ValueError("attention: scale must be > 0")
    if qk_f16:
        def launch():
            L.check(lib.ca_attn_fwd_qk16(arr, len(problems), num_heads, _stream()), "ca_attn_fwd_qk16")
    else:
        def launch():
            L.check(lib.ca_attn_fwd_bf16(arr, len(problems), num_heads, scale, _stream()), "ca_attn_fwd_bf16")
    if _attn_hook is not None:
        _attn_hook(arr, num_heads, launch)
        return
    launch()


def attention_stats(reset: bool = False) -> dict:
    """Counters of ca_attn4_kernel's rare softmax paths on the current device (blocking copy; diagnostics only)."""
    import ctypes
    c = (ctypes.c_ulonglong * 2)()
    L.check(L.load().ca_attn_stats(c, int(reset)), "ca_attn_stats")
    return {"recomputed_workgroups": int(c[0]), "rereference_events": int(c[1])}


# Optional instrumentation used by bench.py: hook(problem_array, num_heads, launch) must call launch().
_attn_hook = None


def set_attn_hook(hook) -> None:
    global _attn_hook
    _attn_hook = hook


def ln_modulate(x, out, segments, eps: float = 1e-6, out_scale=None, out_lo=None) -> None:
    """segments: [(row_end, shift fp32[H], scale fp32[H]), ...] covering all rows of x.
    With ``out`` uint8 and ``out_scale`` fp32 [M] the result is quantised to e4m3 per row (ca_ln_modulate_fp8).
    ``out_lo`` (bf16, fp32 input only): second plane bf16(y - float(bf16(y))) (ca_ln_modulate_f32in_split)."""
    lib = L.load()
    fp8 = out.dtype == torch.uint8
    x32 = x.dtype == torch.float32   # fp32 residual stream
    _chk(x, torch.float32 if x32 else torch.bfloat16, "x"), _chk(out, torch.uint8 if fp8 else torch.bfloat16, "out")
    if fp8 and (out_scale is None or out_scale.dtype != torch.float32 or out_scale.numel() != x.shape[0]
                or not out_scale.is_contiguous()):
        raise ValueError("ln_modulate: fp8 output needs a contiguous fp32 out_scale [M]")
    if len(segments) > L.MAX_SEGMENTS:
        raise ValueError("ln_modulate: too many segments")
    arr = (L.ModSegment * len(segments))()
    for i, (row_end, shift, scale) in enumerate(segments):
        arr[i].row_end = row_end
        arr[i].shift = _chk(shift, torch.float32, "shift").data_ptr()
        arr[i].scale = _chk(scale, torch.float32, "scale").data_ptr()
    if fp8:
        fn = lib.ca_ln_modulate_f32in_fp8 if x32 else lib.ca_ln_modulate_fp8
        L.check(fn(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), out_scale.data_ptr(),
                   x.shape[0], x.shape[1], arr, len(segments), eps, _stream()), "ca_ln_modulate_fp8")
        return
    if out_lo is not None:
        if not x32:
            raise ValueError("ln_modulate: out_lo needs an fp32 input")
        _chk(out_lo, torch.bfloat16, "out_lo")
        if out_lo.shape != out.shape:
            raise ValueError("ln_modulate: out_lo must have out's shape")
        L.check(lib.ca_ln_modulate_f32in_split(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0),
                                               out_lo.data_ptr(), out_lo.stride(0), x.shape[0], x.shape[1], arr,
                                               len(segments), eps, _stream()), "ca_ln_modulate_f32in_split")
        return
    fn = lib.ca_ln_modulate_f32in if x32 else lib.ca_ln_modulate_bf16
    L.check(fn(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1],
               arr, len(segments), eps, _stream()), "ca_ln_modulate_bf16")


def qpre_finish(x, d, norm_scale, num_heads: int, rope=None, q_out=None, q_out_scale: float = 0.0,
                q_f16: bool = False) -> None:
    """x <- RMSNorm_128(x + d) * norm_scale per (row, head), in place; x, d fp32 [M, heads*128] (d may be None).
    With ``rope`` (fp32 [M,64,2]) and ``q_out`` (bf16-typed [M, heads*128] view, row stride free) the result is also
    rotated, multiplied by ``q_out_scale`` and stored there as the attention's q (IEEE half bits with ``q_f16``)."""
    lib = L.load()
    _chk(x, torch.float32, "x"), _chk(norm_scale, torch.bfloat16, "norm_scale")
    if d is not None:
        _chk(d, torch.float32, "d")
        if d.shape != x.shape:
            raise ValueError("qpre_finish: d must have x's shape")
    if x.dim() != 2 or x.shape[1] != num_heads * 128 or norm_scale.numel() != 128:
        raise ValueError("qpre_finish: x must be [M, heads*128], norm_scale [128]")
    if q_out is None:
        L.check(lib.ca_qpre_finish_f32(x.data_ptr(), x.stride(0), _ptr(d), 0 if d is None else d.stride(0),
                                       norm_scale.data_ptr(), x.shape[0], num_heads, _stream()), "ca_qpre_finish_f32")
        return
    _chk(q_out, torch.bfloat16, "q_out"), _chk(rope, torch.float32, "rope")
    if tuple(q_out.shape) != tuple(x.shape) or tuple(rope.shape) != (x.shape[0], 64, 2) or not rope.is_contiguous():
        raise ValueError("qpre_finish: q_out must have x's shape, rope must be contiguous [M,64,2]")
    L.check(lib.ca_qpre_finish_rope_f32(x.data_ptr(), x.stride(0), _ptr(d), 0 if d is None else d.stride(0),
                                        norm_scale.data_ptr(), rope.data_ptr(), q_out.data_ptr(), q_out.stride(0),
                                        float(q_out_scale), int(bool(q_f16)), x.shape[0], num_heads, _stream()),
            "ca_qpre_finish_rope_f32")


def quantize_rows_fp8(x, out=None, out_scale=None):
    """Row-wise absmax quantisation bf16 [M,K] -> (uint8 e4m3 [M,K], fp32 scale [M])."""
    lib = L.load()
    _chk(x, torch.bfloat16, "x")
    if x.dim() != 2:
        raise ValueError("quantize_rows_fp8: expected a 2-D tensor")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    if out_scale is None:
        out_scale = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
    _chk(out, torch.uint8, "out"), _chk(out_scale, torch.float32, "out_scale")
    if tuple(out.shape) != tuple(x.shape) or out_scale.numel() != x.shape[0] or not out_scale.is_contiguous():
        raise ValueError("quantize_rows_fp8: out must match x, out_scale must be contiguous [M]")
    L.check(lib.ca_quantize_rows_fp8(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), out_scale.data_ptr(),
                                     x.shape[0], x.shape[1], _stream()), "ca_quantize_rows_fp8")
    return out, out_scale


def qknorm_rope(qkv, num_heads, segments, rope_cos_sin, q_prerope=None) -> None:
    """In place on the q,k thirds of qkv [M, 3*num_heads*128].
    segments: [(row_end, q_scale bf16[128], k_scale bf16[128])]; rope_cos_sin fp32 [M,64,2]."""
    lib = L.load()
    _chk(qkv, torch.bfloat16, "qkv"), _chk(rope_cos_sin, torch.float32, "rope_cos_sin")
    M = qkv.shape[0]
    if tuple(rope_cos_sin.shape) != (M, 64, 2) or not rope_cos_sin.is_contiguous():
        raise ValueError(f"qknorm_rope: rope table must be contiguous [M,64,2], got {tuple(rope_cos_sin.shape)}")
    arr = (L.NormSegment * len(segments))()
    for i, (row_end, qs, ks) in enumerate(segments):
        arr[i].row_end = row_end
        arr[i].q_scale = _chk(qs, torch.bfloat16, "q_scale").data_ptr()
        arr[i].k_scale = _chk(ks, torch.bfloat16, "k_scale").data_ptr()
    pp, ldp = (None, 0) if q_prerope is None else (_chk(q_prerope, torch.bfloat16, "q_prerope").data_ptr(),
                                                   q_prerope.stride(0))
    L.check(lib.ca_qknorm_rope_bf16(qkv.data_ptr(), qkv.stride(0), M, num_heads, arr, len(segments),
                                    rope_cos_sin.data_ptr(), pp, ldp, _stream()), "ca_qknorm_rope_bf16")


def gemv(x, w, bias, out, silu_input=False, accumulate=False) -> None:
    """out[v,:] (+)= f(x[v,:]) @ w.T + bias; x fp32 [nv,K] (nv<=8), w bf16 [N,K], out fp32 [nv,N]."""
    lib = L.load()
    _chk(x, torch.float32, "x"), _chk(w, torch.bfloat16, "w"), _chk(out, torch.float32, "out")
    if not w.is_contiguous():
        raise ValueError("gemv: w must be contiguous")
    b = _ptr(None if bias is None else _chk(bias, torch.bfloat16, "bias"))
    L.check(lib.ca_gemv_bf16(x.data_ptr(), x.shape[0], x.stride(0), w.data_ptr(), b, out.data_ptr(), out.stride(0),
                             w.shape[0], w.shape[1], int(silu_input), int(accumulate), _stream()), "ca_gemv_bf16")


def silu_split(x, hi, lo) -> None:
    """hi + lo = silu(x) to ~16 mantissa bits, as two bf16 planes (x fp32 [rows,K]; hi, lo bf16 [rows,K])."""
    lib = L.load()
    _chk(x, torch.float32, "x"), _chk(hi, torch.bfloat16, "hi"), _chk(lo, torch.bfloat16, "lo")
    if x.dim() != 2 or hi.shape != x.shape or lo.shape != x.shape or hi.stride(0) != lo.stride(0):
        raise ValueError("silu_split: x, hi, lo must be [rows,K] of one shape (hi / lo of one row stride)")
    L.check(lib.ca_silu_split_bf16(x.data_ptr(), x.stride(0), hi.data_ptr(), lo.data_ptr(), hi.stride(0), x.shape[0],
                                   x.shape[1], _stream()), "ca_silu_split_bf16")


def modulation_gemm(vecs, w, bias, out, ones) -> bool:
    """out[v,:] = silu(vecs[v,:]) @ w.T + bias for any number of vectors through the thin-row GEMM kernel (ca_gemv
    streams the weights once per 4 vectors): silu(vecs) as two bf16 planes hi + lo.  Up to 32 vectors: ONE weight pass
    over the stacked rows [hi; lo] (64 rows per workgroup), the two planes' products folded by ca_modulation_combine;
    more: two passes, the second plane accumulating into the fp32 output (gate = ones).  Both forms round alike
    ((hi.w + bias) + lo.w), so a vector's result does not depend on the count.  vecs fp32 [nv,K], w bf16 [N,K], bias
    bf16 [N], out fp32 [nv,N], ones fp32 [N].  Returns False (nothing launched) when the shape does not fit that kernel
    (N % 256, K % 64): the caller then uses gemv."""
    nv, K = vecs.shape
    N = w.shape[0]
    if N % 256 or K % 64 or nv < 1 or not w.is_contiguous():
        return False
    if nv > 128:  # the thin-row kernel takes at most 128 rows per problem: 128 vectors per pair of weight passes
        for r0 in range(0, nv, 128):
            modulation_gemm(vecs[r0:r0 + 128], w, bias, out[r0:r0 + 128], ones)
        return True
    lib = L.load()
    planes = torch.empty(2 * nv, K, device=vecs.device, dtype=torch.bfloat16)
    hi, lo = planes[:nv], planes[nv:]
    silu_split(vecs, hi, lo)
    # column chunks of fewer than 2^32 weight bytes (the GEMM's 32-bit operand offsets), multiples of 256 columns
    n_chunks = -(-N * K * 2 // ((1 << 32) - 1))
    step = -(-(N // 256) // n_chunks) * 256
    pair = torch.empty(2 * nv, N, device=vecs.device, dtype=torch.float32) if nv <= 32 else None
    for c0 in range(0, N, step):
        c1 = min(c0 + step, N)
        if pair is not None:
            gemm([Gemm(planes, w[c0:c1], None, pair[:, c0:c1], L.EPI_BIAS)], L.TILE_PP_256x256)
            continue
        o = out[:, c0:c1]
        gemm([Gemm(hi, w[c0:c1], None if bias is None else bias[c0:c1], o, L.EPI_BIAS)], L.TILE_PP_256x256)
        gemm([Gemm(lo, w[c0:c1], None, o, L.EPI_GATE_RESIDUAL, resid=o, gate=ones[c0:c1])], L.TILE_PP_256x256)
    if pair is not None:
        _chk(out, torch.float32, "out")
        L.check(lib.ca_modulation_combine_f32(pair.data_ptr(), pair.stride(0), _ptr(bias), out.data_ptr(), out.stride(0),
                                              nv, N, _stream()), "ca_modulation_combine_f32")
    return True


def heatmap_logits(img_vec, con_vec, logits) -> None:
    """logits[c,p] = <img_vec[p,:], con_vec[c,:]>; img bf16|fp32 [L,dim], con bf16|fp32 [C,dim] -> fp32 [C,L]
    (fp32 image vectors go with fp32 concept vectors)."""
    lib = L.load()
    if img_vec.dtype not in (torch.bfloat16, torch.float32) or con_vec.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("heatmap_logits: img_vec and con_vec must be bf16 or fp32")
    _chk(img_vec, img_vec.dtype, "img_vec")
    _chk(con_vec, con_vec.dtype, "con_vec")
    _chk(logits, torch.float32, "logits")
    Lp, dim, Cc = img_vec.shape[0], img_vec.shape[1], con_vec.shape[0]
    if tuple(logits.shape) != (Cc, Lp) or not logits.is_contiguous() or con_vec.shape[1] != dim:
        raise ValueError("heatmap_logits: shape mismatch")
    L.check(lib.ca_heatmap_logits_bf16(img_vec.data_ptr(), img_vec.stride(0), con_vec.data_ptr(), con_vec.stride(0),
                                       int(con_vec.dtype == torch.float32) | 2 * int(img_vec.dtype == torch.float32),
                                       Lp, Cc, dim, logits.data_ptr(),
                                       _stream()), "ca_heatmap_logits_bf16")


def heatmap_softmax_accumulate(logits, acc, weight: float, norm: int = L.NORM_SOFTMAX) -> None:
    """acc += weight * norm_over_concepts(logits); norm: L.NORM_SOFTMAX | NORM_SPARSEMAX | NORM_ENTMAX15."""
    lib = L.load()
    _chk(logits, torch.float32, "logits"), _chk(acc, torch.float32, "acc")
    if logits.shape != acc.shape or not logits.is_contiguous() or not acc.is_contiguous():
        raise ValueError("heatmap_softmax_accumulate: logits/acc must be contiguous [C,L]")
    if norm == L.NORM_SOFTMAX:
        L.check(lib.ca_heatmap_softmax_accumulate(logits.data_ptr(), logits.shape[0], logits.shape[1], weight,
                                                  acc.data_ptr(), _stream()), "ca_heatmap_softmax_accumulate")
        return
    L.check(lib.ca_heatmap_norm_accumulate(logits.data_ptr(), logits.shape[0], logits.shape[1], int(norm), weight,
                                           acc.data_ptr(), _stream()), "ca_heatmap_norm_accumulate")


@dataclass
class Heatmap:
    """One (work item, space) problem of a fused heat-map launch: logits = img_vec @ con_vec.T per patch, then
    acc += weight * norm_c(logits) and / or acc2 += weight2 * norm_c(logits); ``logits`` (optional) receives the raw
    logits.  img_vec bf16|fp32 [L,dim], con_vec bf16|fp32 [C,dim], acc / acc2 / logits fp32 [C,L] contiguous."""
    img_vec: Optional[torch.Tensor]
    con_vec: Optional[torch.Tensor]
    acc: Optional[torch.Tensor] = None
    weight: float = 0.0
    acc2: Optional[torch.Tensor] = None
    weight2: float = 0.0
    logits: Optional[torch.Tensor] = None
    # instead of the two vector sets: fp32 [heads, L, 8] per-head partial logits (Attn.hm_part): logits = their sum over
    # the heads, in head order
    part: Optional[torch.Tensor] = None


def heatmap_fused_fits(C: int, dim: int) -> bool:
    """Does ca_heatmap_fused take this geometry (all C concept vectors of a problem as fp32 in LDS)?"""
    return 1 <= C <= 8 and dim % 8 == 0 and 8 <= dim <= 4096


def heatmap_fused(problems: Sequence[Heatmap], norm: int = L.NORM_SOFTMAX) -> None:
    """All heat-map updates of one layer in ONE launch (ca_heatmap_fused): bit-identical to heatmap_logits followed by
    heatmap_softmax_accumulate per problem and accumulator."""
    lib = L.load()
    if not 1 <= len(problems) <= L.HEATMAP_MAX_PROBLEMS:
        raise ValueError(f"heatmap_fused: 1..{L.HEATMAP_MAX_PROBLEMS} problems per launch")
    arr = (L.HeatmapProblem * len(problems))()
    some = next((t for h in problems for t in (h.acc, h.acc2, h.logits) if t is not None), None)
    if some is None:
        raise ValueError("heatmap_fused: every problem needs acc, acc2 or logits")
    Cc, Lp = some.shape
    vec = next((h for h in problems if h.part is None), None)
    dim = 8 if vec is None else vec.img_vec.shape[1]
    for i, h in enumerate(problems):
        p = arr[i]
        if h.part is not None:
            _chk(h.part, torch.float32, "part")
            if h.part.dim() != 3 or tuple(h.part.shape[1:]) != (Lp, 8) or not h.part.is_contiguous():
                raise ValueError(f"heatmap_fused[{i}]: part must be contiguous fp32 [heads, L, 8]")
            p.img_vec, p.ldi, p.img_f32 = h.part.data_ptr(), h.part.shape[0], 2
        else:
            if h.img_vec.dtype not in (torch.bfloat16, torch.float32) or h.con_vec.dtype not in (torch.bfloat16, torch.float32):
                raise ValueError(f"heatmap_fused[{i}]: img_vec and con_vec must be bf16 or fp32")
            _chk(h.img_vec, h.img_vec.dtype, "img_vec"), _chk(h.con_vec, h.con_vec.dtype, "con_vec")
            if tuple(h.img_vec.shape) != (Lp, dim) or tuple(h.con_vec.shape) != (Cc, dim):
                raise ValueError(f"heatmap_fused[{i}]: all problems of a launch share L, C and dim")
            p.img_vec, p.con_vec, p.ldi, p.ldc = h.img_vec.data_ptr(), h.con_vec.data_ptr(), h.img_vec.stride(0), h.con_vec.stride(0)
            p.img_f32, p.con_f32 = int(h.img_vec.dtype == torch.float32), int(h.con_vec.dtype == torch.float32)
        for name in ("acc", "acc2", "logits"):
            t = getattr(h, name)
            if t is not None:
                _chk(t, torch.float32, name)
                if tuple(t.shape) != (Cc, Lp) or not t.is_contiguous():
                    raise ValueError(f"heatmap_fused[{i}]: {name} must be contiguous fp32 [C,L]")
                setattr(p, name, t.data_ptr())
        p.weight, p.weight2 = float(h.weight), float(h.weight2)
    L.check(lib.ca_heatmap_fused(arr, len(problems), Lp, Cc, dim, int(norm), _stream()), "ca_heatmap_fused")


def axpy(x, y, a: float) -> None:
    """x += a*y (bf16, fp32 math)."""
    lib = L.load()
    _chk(x, torch.bfloat16, "x"), _chk(y, torch.bfloat16, "y")
    if not (x.is_contiguous() and y.is_contiguous()) or x.numel() != y.numel():
        raise ValueError("axpy: x,y must be contiguous with equal sizes")
    L.check(lib.ca_axpy_bf16(x.data_ptr(), y.data_ptr(), float(a), x.numel(), _stream()), "ca_axpy_bf16")


def axpy_f32(x, y, a: float) -> None:
    """x (fp32) += a*y (bf16 or fp32): the Euler update with the latent kept in fp32 between the steps."""
    lib = L.load()
    _chk(x, torch.float32, "x")
    if y.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("axpy_f32: y must be bf16 or fp32")
    _chk(y, y.dtype, "y")
    if not (x.is_contiguous() and y.is_contiguous()) or x.numel() != y.numel():
        raise ValueError("axpy_f32: x,y must be contiguous with equal sizes")
    L.check(lib.ca_axpy_f32(x.data_ptr(), y.data_ptr(), int(y.dtype == torch.float32), float(a), x.numel(), _stream()),
            "ca_axpy_f32")


def split_planes(x, hi, lo) -> None:
    """hi = bf16(x), lo = bf16(x - hi): x fp32 [rows,K] as two bf16 planes (~16 mantissa bits)."""
    lib = L.load()
    _chk(x, torch.float32, "x"), _chk(hi, torch.bfloat16, "hi"), _chk(lo, torch.bfloat16, "lo")
    if x.dim() != 2 or hi.shape != x.shape or lo.shape != x.shape or hi.stride(0) != lo.stride(0):
        raise ValueError("split_planes: x, hi, lo must be [rows,K] of one shape (hi / lo of one row stride)")
    L.check(lib.ca_split_bf16(x.data_ptr(), x.stride(0), hi.data_ptr(), lo.data_ptr(), hi.stride(0), x.shape[0],
                              x.shape[1], _stream()), "ca_split_bf16")


def timestep_embedding(t, out, time_factor: float = 1000.0, max_period: float = 10000.0) -> None:
    """out[v,:] = [cos(tf*t[v]*f), sin(tf*t[v]*f)]; t fp32 [nt], out fp32 [nt, dim] contiguous."""
    lib = L.load()
    _chk(t, torch.float32, "t"), _chk(out, torch.float32, "out")
    if out.dim() != 2 or out.shape[0] != t.shape[0] or not out.is_contiguous():
        raise ValueError("timestep_embedding: out must be contiguous [nt, dim]")
    L.check(lib.ca_timestep_embedding_f32(t.data_ptr(), t.shape[0], out.data_ptr(), out.shape[1], time_factor,
                                          max_period, _stream()), "ca_timestep_embedding_f32")
