"""State-dict layout and seeded synthetic weights / inputs.

The state-dict keys and shapes are the BFL Flux names the reference loads with
``load_state_dict(strict=False, assign=True)`` (concept_attention/image_generator.py:43-44;
full list in SURVEY.md §8b), so a real ``flux1-*.safetensors`` drops in unchanged.

Synthetic mode (no network: BASELINE.json asks for random-init weights): every tensor is drawn
from its own generator seeded by ``crc32(name) ^ seed`` so the result does not depend on
construction order and is identical on every rank.
"""
from __future__ import annotations

import math
import zlib
from typing import Iterator

import torch

from .params import FluxParams


def state_dict_spec(p: FluxParams) -> list[tuple[str, tuple[int, ...]]]:
    """[(name, shape)] in BFL order; Linear weights are (out, in) row-major."""
    H, MLP, D = p.hidden_size, p.mlp_hidden, p.head_dim
    spec: list[tuple[str, tuple[int, ...]]] = []

    def lin(name, out, inp, bias=True):
        spec.append((name + ".weight", (out, inp)))
        if bias:
            spec.append((name + ".bias", (out,)))

    lin("img_in", H, p.in_channels)
    for emb, inp in (("time_in", 256), ("vector_in", p.vec_in_dim)) + (
            (("guidance_in", 256),) if p.guidance_embed else ()):
        lin(f"{emb}.in_layer", H, inp)
        lin(f"{emb}.out_layer", H, H)
    lin("txt_in", H, p.context_in_dim)
    for i in range(p.depth):
        for s in ("img", "txt"):
            b = f"double_blocks.{i}.{s}"
            lin(f"{b}_mod.lin", 6 * H, H)
            lin(f"{b}_attn.qkv", 3 * H, H, bias=p.qkv_bias)
            spec.append((f"{b}_attn.norm.query_norm.scale", (D,)))
            spec.append((f"{b}_attn.norm.key_norm.scale", (D,)))
            lin(f"{b}_attn.proj", H, H)
            lin(f"{b}_mlp.0", MLP, H)
            lin(f"{b}_mlp.2", H, MLP)
    for i in range(p.depth_single_blocks):
        b = f"single_blocks.{i}"
        lin(f"{b}.linear1", 3 * H + MLP, H)
        lin(f"{b}.linear2", H, H + MLP)
        spec.append((f"{b}.norm.query_norm.scale", (D,)))
        spec.append((f"{b}.norm.key_norm.scale", (D,)))
        lin(f"{b}.modulation.lin", 3 * H, H)
    lin("final_layer.linear", p.in_channels, H)
    lin("final_layer.adaLN_modulation.1", 2 * H, H)
    return spec


def _gen(name: str, seed: int, device) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed(((zlib.crc32(name.encode()) << 20) ^ (seed * 0x9E3779B1)) & 0x7FFF_FFFF_FFFF_FFFF)
    return g


def synth_tensor(name: str, shape: tuple[int, ...], fan_in: int, seed: int = 0,
                 device="cpu", dtype=torch.float32) -> torch.Tensor:
    """One synthetic parameter.

    * Linear weight / bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (PyTorch's default init bound;
      keeps activations O(1)).
    * QK-norm scales: query 0.15*(1±0.1), key 6.0*(1±0.1).  With unit scales the
      cross-attention-space logits <img_q, concept_q> over 24x128 unit-RMS dims have std ~55
      and the concept softmax saturates (SURVEY.md §7 "hard parts"); these values keep both the
      attention logits (q.k/sqrt(128), std ~1) and the cross-attention logits (std ~1.3)
      in a range where a numerical comparison is meaningful.
    """
    g = _gen(name, seed, device)
    u = torch.rand(shape, generator=g, device=device, dtype=torch.float32) * 2 - 1
    if name.endswith("query_norm.scale"):
        out = 0.15 * (1 + 0.1 * u)
    elif name.endswith("key_norm.scale"):
        out = 6.0 * (1 + 0.1 * u)
    else:
        out = u * (1.0 / math.sqrt(fan_in))
    return out.to(dtype)


def iter_synthetic_state_dict(p: FluxParams, seed: int = 0, device="cpu", dtype=torch.float32,
                              prefix: str | None = None) -> Iterator[tuple[str, torch.Tensor]]:
    """Yield (name, tensor) lazily (a full flux model is 23.8 GB in bf16)."""
    fan_in = {}
    spec = state_dict_spec(p)
    for name, shape in spec:
        if name.endswith(".weight"):
            fan_in[name[: -len(".weight")]] = shape[1]
    for name, shape in spec:
        if prefix is not None and not name.startswith(prefix):
            continue
        base = name.rsplit(".", 1)[0]
        yield name, synth_tensor(name, shape, fan_in.get(base, 1), seed, device, dtype)


def synthetic_state_dict(p: FluxParams, seed: int = 0, device="cpu", dtype=torch.float32,
                         prefix: str | None = None) -> dict[str, torch.Tensor]:
    return dict(iter_synthetic_state_dict(p, seed, device, dtype, prefix))


def synthetic_inputs(p: FluxParams, height: int, width: int, n_txt: int, n_concepts: int,
                     seed: int = 0, device="cpu", dtype=torch.float32) -> dict[str, torch.Tensor]:
    """Synthetic stand-ins for what get_noise/prepare/embed_concepts hand to the model
    (SURVEY.md §8d): latent ~N(0,1) (1,16,h/8,w/8), txt ~N(0,1) (1,T,4096),
    concepts ~N(0,1) (1,C,4096), vec ~N(0,1) (1,768), concept_vec = 0, ids as in
    flux/sampling.py:40-50 and concept_attention/utils.py:26."""
    def n(name, shape):
        return torch.randn(shape, generator=_gen("input." + name, seed, device), device=device,
                           dtype=torch.float32).to(dtype)
    h8, w8 = 2 * math.ceil(height / 16), 2 * math.ceil(width / 16)
    h2, w2 = h8 // 2, w8 // 2
    ids = torch.zeros(h2, w2, 3, device=device)
    ids[..., 1] = torch.arange(h2, device=device)[:, None]
    ids[..., 2] = torch.arange(w2, device=device)[None, :]
    return {
        "latent": n("latent", (1, p.in_channels // 4, h8, w8)),
        "img_ids": ids.reshape(1, h2 * w2, 3),
        "txt": n("txt", (1, n_txt, p.context_in_dim)),
        "txt_ids": torch.zeros(1, n_txt, 3, device=device),
        "vec": n("vec", (1, p.vec_in_dim)),
        "concepts": n("concepts", (1, n_concepts, p.context_in_dim)),
        "concept_ids": torch.zeros(1, n_concepts, 3, device=device),
        "concept_vec": torch.zeros(1, p.vec_in_dim, device=device, dtype=dtype),
    }
