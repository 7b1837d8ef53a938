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


# Peaky-logit variants of the full-size double-block case (round 5: every other model-level parity number lives on the
# synthetic weights' logit std of ~1 nat).  Applied to the block's state dict (names without the block prefix, values
# already bf16-representable); the results stay bf16-representable so that the reference, the oracle and the HIP path
# see the same numbers.
#   iid8      key_norm scales x 8: the joint-attention logits q.k/sqrt(128) get std ~8 nats (the query scales -- which
#             are also the cross-attention-space vectors' -- stay, so the cross-space maps keep their conditioning);
#             the v third of both qkv projections x 0.25: a peaky attention row returns single value vectors instead of
#             their average, and without this the output-space logits (dot products of two such rows over 3072 dims)
#             saturate the softmax over the concepts (maps exactly 0 / 1: nothing left to compare)
#   coldtext  a common direction u (unit RMS over a head's 128 dims) enters every image query (+beta u on the q third of
#             img_attn.qkv.bias) and every text / concept key (-beta u on the k third of txt_attn.qkv.bias), key scales
#             x 2.5: for an image query row the whole text tile sits ~20 nats below its image keys (structured, as a
#             trained head with a cold text tile; the attention kernel keeps tile 0's maximum as softmax reference)
PEAKY_CASES = {"iid8": dict(key_gain=8.0, beta=0.0, v_gain=0.25), "coldtext": dict(key_gain=2.5, beta=1.2, v_gain=1.0)}


def peaky_state_dict(sd: dict, case: str, hidden: int) -> dict:
    cfg = PEAKY_CASES[case]
    out = dict(sd)

    def rb(t):
        return t.bfloat16().float()
    for stream in ("img", "txt"):
        k = f"{stream}_attn.norm.key_norm.scale"
        out[k] = rb(sd[k] * cfg["key_gain"])
    if cfg["v_gain"] != 1.0:
        for stream in ("img", "txt"):
            for kind in ("weight", "bias"):
                t = out[f"{stream}_attn.qkv.{kind}"].clone()
                t[2 * hidden:] *= cfg["v_gain"]          # (a power of two: stays bf16-representable)
                out[f"{stream}_attn.qkv.{kind}"] = t
    if cfg["beta"]:
        g = torch.Generator().manual_seed(1234)
        u = torch.randn(128, generator=g)
        u = u / u.pow(2).mean().sqrt()
        H = hidden
        b = out["img_attn.qkv.bias"].clone()
        b[:H] += cfg["beta"] * u.repeat(H // 128)
        out["img_attn.qkv.bias"] = rb(b)
        b = out["txt_attn.qkv.bias"].clone()
        b[H:2 * H] -= cfg["beta"] * u.repeat(H // 128)
        out["txt_attn.qkv.bias"] = rb(b)
    return out
