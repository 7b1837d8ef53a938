"""ctypes binding of libconceptattn.so (the C ABI declared in include/conceptattn.h).

The library is the product path: there is no Python/PyTorch fallback.  If the shared object is
missing or does not export a symbol, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CA_LIB_PATH: another build of the same library (A/B and diagnostic builds under tools/ab); still no fallback
LIB_PATH = os.environ.get("CA_LIB_PATH") or os.path.join(_HERE, "libconceptattn.so")

CA_VERSION = 125
EPI_BIAS, EPI_GELU_TANH, EPI_GATE_RESIDUAL, EPI_SPLIT_GELU, EPI_QKV_NORM_ROPE = 0, 1, 2, 3, 4
TILE_AUTO, TILE_256x256, TILE_256x192, TILE_256x128, TILE_256x64 = 0, 1, 2, 3, 4
TILE_PP_256x256, TILE_PP_256x128, TILE_PP_256x192 = 5, 6, 7
NORM_SOFTMAX, NORM_SPARSEMAX, NORM_ENTMAX15 = 0, 1, 2
NORMS = {"softmax": NORM_SOFTMAX, "sparsemax": NORM_SPARSEMAX, "entmax15": NORM_ENTMAX15}
MAX_SEGMENTS = 16
GEMM_MAX_PROBLEMS = 2
ATTN_MAX_PROBLEMS = 16
HEATMAP_MAX_PROBLEMS = 16
ATTN_Q_PRESCALED = 0.0   # ca_attn_fwd_bf16(scale=...): the q rows already carry softmax_scale * log2(e)


class GemmProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
                ("resid", C.c_void_p), ("gate", C.c_void_p), ("gate2", C.c_void_p), ("out2", C.c_void_p),
                ("norm_q", C.c_void_p), ("norm_k", C.c_void_p), ("rope", C.c_void_p), ("q_prerope", C.c_void_p),
                ("a_scale", C.c_void_p), ("w_scale", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldw", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
                ("ld2", C.c_int32), ("n_split", C.c_int32), ("gate_rows", C.c_int32),
                ("epilogue", C.c_int32), ("ldp", C.c_int32), ("out_f32", C.c_int32), ("gate_stride", C.c_int32),
                ("gate_item_rows", C.c_int32), ("gate2_item_rows", C.c_int32),
                ("qpre_f32", C.c_int32), ("q_out_scale", C.c_float), ("qk_f16", C.c_int32), ("_pad", C.c_int32)]


class AttnProblem(C.Structure):
    _fields_ = [("q", C.c_void_p), ("out", C.c_void_p), ("k0", C.c_void_p), ("v0", C.c_void_p),
                ("k1", C.c_void_p), ("v1", C.c_void_p), ("out_f32", C.c_void_p),
                ("q1", C.c_void_p), ("out1", C.c_void_p), ("hm_con", C.c_void_p), ("hm_part", C.c_void_p),
                ("nq", C.c_int32), ("n0", C.c_int32), ("n1", C.c_int32),
                ("ldq", C.c_int32), ("ldo", C.c_int32), ("ldkv", C.c_int32), ("ldo32", C.c_int32),
                ("nq0", C.c_int32), ("hm_C", C.c_int32), ("ldhc", C.c_int32), ("_pad", C.c_int32)]


class HeatmapProblem(C.Structure):
    _fields_ = [("img_vec", C.c_void_p), ("con_vec", C.c_void_p), ("acc", C.c_void_p), ("acc2", C.c_void_p),
                ("logits", C.c_void_p), ("ldi", C.c_int32), ("ldc", C.c_int32), ("img_f32", C.c_int32),
                ("con_f32", C.c_int32), ("weight", C.c_float), ("weight2", C.c_float)]


class ModSegment(C.Structure):
    _fields_ = [("row_end", C.c_int32), ("_pad", C.c_int32), ("shift", C.c_void_p), ("scale", C.c_void_p)]


class NormSegment(C.Structure):
    _fields_ = [("row_end", C.c_int32), ("_pad", C.c_int32), ("q_scale", C.c_void_p), ("k_scale", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/conceptattn.h declares
SIGNATURES = {
    "ca_version": (C.c_int, []),
    "ca_last_error": (C.c_char_p, []),
    "ca_check_device": (C.c_int, []),
    "ca_gemm_bf16": (C.c_int, [C.POINTER(GemmProblem), C.c_int32, C.c_int32, C.c_void_p]),
    "ca_gemm_auto_tile": (C.c_int, [C.POINTER(GemmProblem), C.c_int32]),
    "ca_gemm_fp8": (C.c_int, [C.POINTER(GemmProblem), C.c_int32, C.c_void_p]),
    "ca_attn_fwd_bf16": (C.c_int, [C.POINTER(AttnProblem), C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "ca_attn_fwd_qk16": (C.c_int, [C.POINTER(AttnProblem), C.c_int32, C.c_int32, C.c_void_p]),
    "ca_attn_stats": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int32]),
    "ca_ln_modulate_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(ModSegment), C.c_int32, C.c_float, C.c_void_p]),
    "ca_ln_modulate_fp8": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                     C.POINTER(ModSegment), C.c_int32, C.c_float, C.c_void_p]),
    "ca_ln_modulate_f32in": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.POINTER(ModSegment), C.c_int32, C.c_float, C.c_void_p]),
    "ca_ln_modulate_f32in_fp8": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                           C.c_int32, C.POINTER(ModSegment), C.c_int32, C.c_float, C.c_void_p]),
    "ca_ln_modulate_f32in_split": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                             C.c_int32, C.c_int32, C.POINTER(ModSegment), C.c_int32, C.c_float,
                                             C.c_void_p]),
    "ca_qpre_finish_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_void_p]),
    "ca_qpre_finish_rope_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ca_quantize_rows_fp8": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p]),
    "ca_qknorm_rope_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(NormSegment),
                                      C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ca_gemv_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                               C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ca_heatmap_logits_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ca_heatmap_softmax_accumulate": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                                C.c_void_p]),
    "ca_heatmap_norm_accumulate": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                             C.c_void_p]),
    "ca_heatmap_fused": (C.c_int, [C.POINTER(HeatmapProblem), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p]),
    "ca_timestep_embedding_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                            C.c_void_p]),
    "ca_axpy_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    "ca_silu_split_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p]),
    "ca_axpy_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int64, C.c_void_p]),
    "ca_split_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                C.c_void_p]),
    "ca_modulation_combine_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_void_p]),
}

_lib = None


class ConceptAttnError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen libconceptattn.so and bind every entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own libamdhip64; whichever HIP runtime is mapped first serves the whole process.  Import
    # torch before the dlopen so that the tensors' runtime is also the one the kernels are launched through
    # (the other order leaves this library on /opt/rocm's runtime, which then reports "no ROCm-capable device").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ConceptAttnError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ConceptAttnError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.ca_version() != CA_VERSION:
        raise ConceptAttnError(f"libconceptattn version {lib.ca_version()} != binding {CA_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ca_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise ConceptAttnError(f"{what}: rc={rc}: {msg}")
