"""Model geometry for the Flux DiT the ConceptAttention path runs on.

Mirrors ``FluxParams`` (reference concept_attention/modified_flux_dit.py:13-26) and the two
``configs[*].params`` entries of concept_attention/flux/src/flux/util.py:28-93.  The only
difference between flux-schnell and flux-dev is ``guidance_embed`` (util.py:46 vs :78).
"""
from __future__ import annotations

from dataclasses import dataclass, replace


@dataclass(frozen=True)
class FluxParams:
    in_channels: int = 64
    vec_in_dim: int = 768
    context_in_dim: int = 4096
    hidden_size: int = 3072
    mlp_ratio: float = 4.0
    num_heads: int = 24
    depth: int = 19
    depth_single_blocks: int = 38
    axes_dim: tuple = (16, 56, 56)
    theta: int = 10_000
    qkv_bias: bool = True
    guidance_embed: bool = False

    def __post_init__(self):
        # same guards as ModifiedFluxDiT.__init__ (modified_flux_dit.py:40-46)
        if self.hidden_size % self.num_heads != 0:
            raise ValueError(
                f"Hidden size {self.hidden_size} must be divisible by num_heads {self.num_heads}")
        pe_dim = self.hidden_size // self.num_heads
        if sum(self.axes_dim) != pe_dim:
            raise ValueError(f"Got {list(self.axes_dim)} but expected positional dim {pe_dim}")

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def mlp_hidden(self) -> int:
        return int(self.hidden_size * self.mlp_ratio)


configs = {
    "flux-schnell": FluxParams(guidance_embed=False),
    "flux-dev": FluxParams(guidance_embed=True),
}

# T5 sequence length per model (reference concept_attention/image_generator.py:57)
T5_TOKENS = {"flux-schnell": 256, "flux-dev": 512}


def tiny_params(**kw) -> FluxParams:
    """Small geometry for CPU-checkable tests; keeps head_dim == sum(axes_dim) == 128, which
    the HIP kernels (and the reference's EmbedND check) require."""
    base = dict(hidden_size=256, num_heads=2, depth=2, depth_single_blocks=2)
    base.update(kw)
    return replace(FluxParams(), **base)
