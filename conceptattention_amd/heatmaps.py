"""Concept heat maps from stacked vectors: the HIP counterpart of
``compute_heatmaps_from_vectors`` (concept_attention/concept_attention_pipeline.py:29-91).

Semantics kept: optional head merge (:43-51), dot products over the feature axis (:57-61),
softmax / sparsemax / entmax15 ACROSS concepts per patch (:64-71), select timesteps then layers (:76-77), mean (:78-82),
reshape to the patch grid (:85-90).  Differences, both deliberate and documented in DESIGN.md:
the (t, layer) pairs that are not selected are never computed (the reference computes all and
slices), and products/softmax/mean are fp32 (the reference runs them in the activations' bf16).
The grid side is sqrt(patches) instead of the reference's hard-coded 64.
"""
from __future__ import annotations

import math

import torch

from . import _lib as L
from . import ops


def resolve_norm(softmax: bool, attention_norm: str) -> int:
    """The reference's branch order (concept_attention_pipeline.py:64-71): softmax if ``softmax`` or
    attention_norm == "softmax", else entmax15 / sparsemax, else ValueError."""
    if softmax or attention_norm == "softmax":
        return L.NORM_SOFTMAX
    if attention_norm in ("entmax15", "sparsemax"):
        return L.NORMS[attention_norm]
    raise ValueError(f"Unknown attention_norm={attention_norm}")


def linear_normalization(x: torch.Tensor, dim: int) -> torch.Tensor:
    """concept_attention/utils.py:35-44 (host-side, tiny: C x dim)."""
    x_min = torch.min(x, dim=dim, keepdim=True)[0]
    x_shifted = x - x_min
    x_sum = torch.sum(x_shifted, dim=dim, keepdim=True)
    x_sum = torch.where(x_sum == 0, torch.ones_like(x_sum), x_sum)
    return x_shifted / x_sum


def compute_heatmaps_from_vectors(image_vectors, concept_vectors, layer_indices, timesteps=list(range(4)),
                                  softmax: bool = True, normalize_concepts: bool = False,
                                  attention_norm: str = "sparsemax"):
    """image_vectors [t, layers, 1, patches, dim] (or [t, layers, 1, heads, patches, 128]),
    concept_vectors likewise with concepts in place of patches -> fp32 [1, concepts, side, side]."""
    # entmax15 / sparsemax: the reference calls the third-party `entmax` package, which it neither pins nor
    # vendors; the kernels implement the published algorithms (parity with that package UNPINNED)
    norm = resolve_norm(softmax, attention_norm)
    if image_vectors.dim() == 6:
        t, l, b, h, n, d = image_vectors.shape
        image_vectors = image_vectors.permute(0, 1, 2, 4, 3, 5).reshape(t, l, b, n, h * d)
        c = concept_vectors.shape[4]
        concept_vectors = concept_vectors.permute(0, 1, 2, 4, 3, 5).reshape(t, l, b, c, h * d)
    if image_vectors.shape[2] != 1:
        raise NotImplementedError("batch size 1 only")
    if normalize_concepts:
        concept_vectors = linear_normalization(concept_vectors.float(), dim=-2).to(torch.bfloat16)
    # heatmaps[timesteps][:, layer_indices]: python ints index dim 0 / dim 1 (floats truncate as in
    # torch indexing, concept_attention_pipeline.py:76)
    ts = [int(t) for t in timesteps]
    ls = [int(l) for l in layer_indices]
    n_patches, C = image_vectors.shape[3], concept_vectors.shape[3]
    dev = image_vectors.device
    acc = torch.zeros(C, n_patches, device=dev, dtype=torch.float32)
    logits = torch.empty(C, n_patches, device=dev, dtype=torch.float32)
    w = 1.0 / (len(ts) * len(ls))
    for t in ts:
        for l in ls:
            iv = image_vectors[t, l, 0].to(torch.bfloat16).contiguous()
            cv = concept_vectors[t, l, 0].to(torch.bfloat16).contiguous()
            ops.heatmap_logits(iv, cv, logits)
            ops.heatmap_softmax_accumulate(logits, acc, w, norm)
    side = int(round(math.sqrt(n_patches)))
    if side * side != n_patches:
        raise ValueError(f"{n_patches} patches do not form a square grid")
    return acc.view(1, C, side, side)
