"""Sampler-side host logic of the path: noise, schedule, patchify/ids, Euler loop, unpack.

Mirrors concept_attention/flux/src/flux/sampling.py (get_noise :12-29, prepare :31-65 minus the
T5/CLIP calls, get_schedule :67-94, denoise :96-152, unpack :154-162) and
concept_attention/utils.py:6-33 (embed_concepts' shape/zero contract).  Pure index arithmetic
and scalar maths; the per-step model call and the Euler update run in the HIP kernels.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import ops
from .flux_dit import DICT_KEYS, HeatmapRequest


def get_noise(num_samples: int, height: int, width: int, device, dtype, seed: int):
    """flux/sampling.py:12-29 (device generator, so values depend on the device type)."""
    return torch.randn(num_samples, 16, 2 * math.ceil(height / 16), 2 * math.ceil(width / 16), device=device,
                       dtype=dtype, generator=torch.Generator(device=device).manual_seed(seed))


def time_shift(mu: float, sigma: float, t):
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


def get_lin_function(x1: float = 256, y1: float = 0.5, x2: float = 4096, y2: float = 1.15):
    m = (y2 - y1) / (x2 - x1)
    b = y1 - m * x1
    return lambda x: m * x + b


def get_schedule(num_steps: int, image_seq_len: int, base_shift: float = 0.5, max_shift: float = 1.15,
                 shift: bool = True) -> list[float]:
    """flux/sampling.py:78-94."""
    timesteps = torch.linspace(1, 0, num_steps + 1)
    if shift:
        mu = get_lin_function(y1=base_shift, y2=max_shift)(image_seq_len)
        timesteps = time_shift(mu, 1.0, timesteps)
    return timesteps.tolist()


def patchify(x: torch.Tensor) -> torch.Tensor:
    """'b c (h ph) (w pw) -> b (h w) (c ph pw)', ph=pw=2 (flux/sampling.py:36).
    Image-token index = row * (w/2) + col."""
    b, c, h, w = x.shape
    return x.view(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(b, (h // 2) * (w // 2), c * 4)


def unpack(x: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """flux/sampling.py:154-162."""
    h, w = math.ceil(height / 16), math.ceil(width / 16)
    b, _, cpp = x.shape
    c = cpp // 4
    return x.view(b, h, w, c, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(b, c, h * 2, w * 2)


def _tag_ids(t: torch.Tensor, what: tuple) -> torch.Tensor:
    """Mark an id tensor with what it holds, so HipFluxDiT may cache the RoPE table by content (the tag is void
    once the tensor's version counter moves)."""
    try:
        t._ca_ids_tag = (what, t._version)
    except RuntimeError:  # inference mode: no version counter -> no tag -> table rebuilt per forward
        pass
    return t


def make_img_ids(h2: int, w2: int, device=None, batch: int = 1) -> torch.Tensor:
    """img_ids of prepare() (flux/sampling.py:40-43): [0, row, col] per token, repeated over the batch."""
    ids = torch.zeros(h2, w2, 3, device=device)
    ids[..., 1] = torch.arange(h2, device=device)[:, None]
    ids[..., 2] = torch.arange(w2, device=device)[None, :]
    return _tag_ids(ids.reshape(1, h2 * w2, 3).repeat(batch, 1, 1), ("img", h2, w2, batch))


def zero_ids(n: int, device=None, batch: int = 1) -> torch.Tensor:
    """txt_ids of prepare() (flux/sampling.py:50) and concept_ids of embed_concepts (utils.py:26): all zero."""
    return _tag_ids(torch.zeros(batch, n, 3, device=device), ("zero", n, batch))


def prepare_from_embeddings(img: torch.Tensor, txt: torch.Tensor, vec: torch.Tensor) -> dict:
    """prepare() with the T5/CLIP outputs supplied by the caller (they are out of scope here:
    SURVEY.md §2 row 9).  img: latent (1,16,h,w)."""
    bs, c, h, w = img.shape
    if txt.shape[0] != bs or vec.shape[0] != bs:
        raise ValueError("img, txt and vec must have the same batch size")
    return {
        "img": patchify(img),
        "img_ids": make_img_ids(h // 2, w // 2, img.device, bs),
        "txt": txt.to(img.device),
        "txt_ids": zero_ids(txt.shape[1], img.device, bs),
        "vec": vec.to(img.device),
    }


def concept_inputs(concept_embeddings: torch.Tensor, vec_like: torch.Tensor):
    """embed_concepts' output contract (concept_attention/utils.py:6-33): first-token embeddings
    (B,C,4096), all-zero ids (B,C,3) and an all-ZERO pooled vector."""
    b, c = concept_embeddings.shape[:2]
    return (concept_embeddings, zero_ids(c, concept_embeddings.device, b), torch.zeros_like(vec_like))


@torch.no_grad()
def denoise(model, img, img_ids, txt, txt_ids, vec, timesteps: list[float], guidance: float = 4.0,
            concepts=None, concept_ids=None, concept_vec=None, return_intermediate_images: bool = True,
            joint_attention_kwargs=None, return_vectors: bool = True,
            heatmaps: Optional[HeatmapRequest] = None, heatmap_timesteps=None):
    """Sequential Euler loop of flux/sampling.py:96-152.  Returns (img, intermediates, dict) with
    each dict entry stacked over time.  HIP-path extras: ``return_vectors=False`` +
    ``heatmaps`` (one HeatmapRequest per work item of the batch) / ``heatmap_timesteps`` accumulate the concept
    maps inside the model call for the selected (step, layer) pairs instead of stacking the vectors."""
    gen = denoise_steps(model, img, img_ids, txt, txt_ids, vec, timesteps, guidance, concepts, concept_ids,
                        concept_vec, return_intermediate_images, joint_attention_kwargs, return_vectors, heatmaps,
                        heatmap_timesteps)
    while True:
        try:
            next(gen)
        except StopIteration as stop:
            return stop.value


def denoise_steps(model, img, img_ids, txt, txt_ids, vec, timesteps, guidance=4.0, concepts=None, concept_ids=None,
                  concept_vec=None, return_intermediate_images=True, joint_attention_kwargs=None,
                  return_vectors=True, heatmaps=None, heatmap_timesteps=None):
    """Generator form of ``denoise``: yields after the launches of each diffusion step have been
    enqueued (nothing is synchronised), so a caller can interleave several independent work items on
    different HIP streams; the generator's return value is denoise's (img, intermediates, dict)."""
    img = img.to(torch.bfloat16).contiguous().clone()
    # HIP path: the Euler state is kept in fp32 between the steps (the reference re-rounds the running latent to bf16
    # after every step, flux/sampling.py:141 on bf16 tensors: 2^-9 relative per step, accumulating; the fp32 oracle
    # does not), and the model's img_in sees the fp32 value (HipFluxDiT: hi + lo planes).  Returned latents are bf16.
    lat32 = img.float() if getattr(model, "fp32_latent", False) and getattr(model, "residual_dtype", None) == torch.float32 else None
    intermediates = [img.clone()] if return_intermediate_images else []
    out = {k: [] for k in DICT_KEYS} if return_vectors else {}
    guidance_vec = torch.full((img.shape[0],), guidance, device=img.device, dtype=torch.float32)
    sel = None if heatmap_timesteps is None else set(heatmap_timesteps)
    # HIP path: all steps' conditioning vectors / adaLN modulations in one pass over the weights
    slots = hasattr(model, "precompute_conditioning")
    if slots:
        model.precompute_conditioning(timesteps[:-1], vec, concept_vec, guidance)
    for it, (t_curr, t_prev) in enumerate(zip(timesteps[:-1], timesteps[1:])):
        t_vec = torch.full((img.shape[0],), t_curr, dtype=torch.float32, device=img.device)
        hm = heatmaps if (heatmaps is not None and (sel is None or it in sel)) else None
        pred, d = model(img=img if lat32 is None else lat32, img_ids=img_ids, txt=txt, txt_ids=txt_ids, concepts=concepts,
                        concept_ids=concept_ids, concept_vec=concept_vec, y=vec, timesteps=t_vec,
                        guidance=guidance_vec, iteration=it, joint_attention_kwargs=joint_attention_kwargs,
                        return_vectors=return_vectors, heatmaps=hm, **({"cond_slot": it} if slots else {}))
        if lat32 is None:
            ops.axpy(img, pred.contiguous(), t_prev - t_curr)  # img = img + (t_prev - t_curr) * pred  (:141)
        else:
            ops.axpy_f32(lat32, pred.contiguous(), t_prev - t_curr)
            if return_intermediate_images or it == len(timesteps) - 2:
                img = lat32.to(torch.bfloat16)      # what the caller sees is bf16, as in the reference
        if return_intermediate_images:
            intermediates.append(img.clone())
        for k in out:
            out[k].append(d[k])
        yield it
    out = {k: torch.stack(v, 0) for k, v in out.items()}
    return img, intermediates, out
