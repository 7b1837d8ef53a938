"""Seeded inputs of the full-size single-block parity case.  TEST INFRASTRUCTURE ONLY.

Shared by oracle/make_goldens.py (reference run, build container), the CPU oracle test and the
GPU parity test so that all three see bit-identical, bf16-representable inputs
(H=3072, L=4096 image tokens, T=256 text tokens, C=4 concepts: BASELINE.json configs[1]).
"""
from __future__ import annotations

import torch


def full_block_inputs(p, L_side: int = 64, T: int = 256, C: int = 4, seed: int = 7):
    H = p.hidden_size
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).bfloat16().float()

    L = L_side * L_side
    ids = torch.zeros(L_side, L_side, 3)
    ids[..., 1] = torch.arange(L_side)[:, None]
    ids[..., 2] = torch.arange(L_side)[None, :]
    return {
        "img": rn(1, L, H), "txt": rn(1, T, H), "concepts": rn(1, C, H),
        "vec": rn(1, H, scale=0.7), "concept_vec": rn(1, H, scale=0.7),
        "img_ids": ids.reshape(1, L, 3), "txt_ids": torch.zeros(1, T, 3),
        "concept_ids": torch.zeros(1, C, 3),
        "sample_rows": torch.arange(5, L, 67)[:61],
    }
