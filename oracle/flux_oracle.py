"""CPU fp32 oracle for the ConceptAttention hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*: a functional, state-dict driven restatement (plain
torch fp32 on the CPU) of the reference algorithm behind
``ConceptAttentionFluxPipeline.generate_image()``.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package ``conceptattention_amd`` never does.

Parity status: PINNED.  ``oracle/make_goldens.py`` imports the reference's own
modules (in the build container only) with a seeded state-dict and writes
``tests/golden/*.npz``; ``tests/test_oracle_vs_golden.py`` checks every function
below against those vectors.

Every function cites the reference lines (relative to /root/reference) it
follows.  Weights are looked up in a flat ``sd`` mapping that uses the BFL Flux
state-dict names, so the same seeded weights drive the
reference, this oracle and the HIP path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

DICT_KEYS = (
    "output_space_concept_vectors",
    "output_space_image_vectors",
    "cross_attention_concept_vectors",
    "cross_attention_image_vectors",
)


@dataclass
class Geometry:
    """Model geometry; mirrors FluxParams (concept_attention/modified_flux_dit.py:13-26)."""
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


# --------------------------------------------------------------------------- primitives
def linear(sd, name, x):
    """nn.Linear with (out,in) row-major weight."""
    b = sd.get(name + ".bias")
    return F.linear(x, sd[name + ".weight"].float(), None if b is None else b.float())


def layer_norm(x):
    """nn.LayerNorm(elementwise_affine=False, eps=1e-6); modified_double_stream_block.py:51,53."""
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + 1e-6)


def rms_norm(x, scale):
    """RMSNorm, flux/modules/layers.py:63-72."""
    rrms = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
    return x * rrms * scale.float()


def timestep_embedding(t, dim=256, max_period=10000, time_factor=1000.0):
    """flux/modules/layers.py:28-49 (cos first, then sin)."""
    t = time_factor * t.float()
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def mlp_embedder(sd, name, x):
    """MLPEmbedder, flux/modules/layers.py:52-60."""
    return linear(sd, name + ".out_layer", F.silu(linear(sd, name + ".in_layer", x)))


def modulation(sd, name, vec, n):
    """Modulation, flux/modules/layers.py:113-126: lin(silu(vec)).chunk(n) -> (shift, scale, gate)*."""
    return linear(sd, name + ".lin", F.silu(vec))[:, None, :].chunk(n, dim=-1)


def rope_cos_sin(ids, axes_dim, theta):
    """EmbedND + rope, flux/modules/layers.py:18-25 and flux/math.py:15-22.

    ids (n,3) -> cos,sin (n, sum(axes_dim)/2) float32 (angles computed in float64)."""
    cs, sn = [], []
    for a, d in enumerate(axes_dim):
        scale = torch.arange(0, d, 2, dtype=torch.float64) / d
        omega = 1.0 / (theta ** scale)
        ang = ids[:, a].double()[:, None] * omega[None]
        cs.append(torch.cos(ang))
        sn.append(torch.sin(ang))
    return torch.cat(cs, -1).float(), torch.cat(sn, -1).float()


def apply_rope(x, cos, sin):
    """apply_rope, flux/math.py:25-30 on (B,H,n,D): pairs (2i,2i+1) rotated by angle(pos,i)."""
    x0, x1 = x[..., 0::2], x[..., 1::2]
    o0 = cos * x0 - sin * x1
    o1 = sin * x0 + cos * x1
    return torch.stack([o0, o1], dim=-1).reshape(x.shape)


def sdpa(q, k, v):
    """softmax(q k^T / sqrt(D)) v; explicit form of modified_double_stream_block.py:21-41."""
    w = torch.softmax((q @ k.transpose(-2, -1)) / math.sqrt(q.shape[-1]), dim=-1)
    return w @ v


def _split_heads(x, nh):
    """einops 'B L (K H D) -> K B H L D', K=3 (modified_double_stream_block.py:91)."""
    B, L, _ = x.shape
    x = x.view(B, L, 3, nh, -1).permute(2, 0, 3, 1, 4)
    return x[0], x[1], x[2]


def _merge_heads(x):
    """'B H L D -> B L (H D)' (modified_double_stream_block.py:170-176)."""
    B, H, L, D = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, H * D)


# --------------------------------------------------------------------------- blocks
def double_block(sd, p, nh, img, txt, vec, rope_ti, concepts, concept_vec, rope_ci,
                 joint_attention_kwargs=None):
    """ModifiedDoubleStreamBlock.forward, modified_double_stream_block.py:69-204.

    rope_ti / rope_ci = (cos,sin) for [txt;img] and [concepts;img].  The concept attention is
    evaluated for the C concept query rows only (equal to the first C rows of
    the reference's full (C+L)x(C+L) SDPA)."""
    T, C = txt.shape[1], concepts.shape[1]
    im1 = modulation(sd, p + "img_mod", vec, 6)
    tm1 = modulation(sd, p + "txt_mod", vec, 6)
    cm1 = modulation(sd, p + "txt_mod", concept_vec, 6)

    def pre_attn(x, mod, stream):
        xm = (1 + mod[1]) * layer_norm(x) + mod[0]
        q, k, v = _split_heads(linear(sd, p + stream + "_attn.qkv", xm), nh)
        q = rms_norm(q, sd[p + stream + "_attn.norm.query_norm.scale"])
        k = rms_norm(k, sd[p + stream + "_attn.norm.key_norm.scale"])
        return q, k, v

    img_q, img_k, img_v = pre_attn(img, im1, "img")
    txt_q, txt_k, txt_v = pre_attn(txt, tm1, "txt")
    con_q, con_k, con_v = pre_attn(concepts, cm1, "txt")  # concept stream re-uses txt weights

    q = apply_rope(torch.cat((txt_q, img_q), 2), *rope_ti)
    k = apply_rope(torch.cat((txt_k, img_k), 2), *rope_ti)
    attn = sdpa(q, k, torch.cat((txt_v, img_v), 2))
    txt_attn, img_attn = attn[:, :, :T], attn[:, :, T:]

    cq = apply_rope(torch.cat((con_q, img_q), 2), *rope_ci)
    ck = apply_rope(torch.cat((con_k, img_k), 2), *rope_ci)
    cv = torch.cat((con_v, img_v), 2)
    cross = self_ = True
    if joint_attention_kwargs is not None:
        cross = joint_attention_kwargs.get("concept_cross_attention", True)
        self_ = joint_attention_kwargs.get("concept_self_attention", True)
    stored_con_q = con_q
    if cross and self_:
        con_attn = sdpa(cq[:, :, :C], ck, cv)
    elif cross:
        con_attn = sdpa(cq[:, :, :C], ck[:, :, C:], img_v)
    elif self_:
        stored_con_q = cq[:, :, :C]  # reference rebinds concept_q to the post-RoPE tensor (:140)
        con_attn = sdpa(cq[:, :, :C], ck[:, :, :C], con_v)
    else:
        con_attn = con_v

    txt_attn, img_attn, con_attn = map(_merge_heads, (txt_attn, img_attn, con_attn))
    d = {
        "output_space_concept_vectors": con_attn,
        "output_space_image_vectors": img_attn,
        "cross_attention_concept_vectors": stored_con_q,
        "cross_attention_image_vectors": img_q,
    }

    def post_attn(x, a, mod, stream):
        x = x + mod[2] * linear(sd, p + stream + "_attn.proj", a)
        h = (1 + mod[4]) * layer_norm(x) + mod[3]
        h = linear(sd, p + stream + "_mlp.2", F.gelu(linear(sd, p + stream + "_mlp.0", h), approximate="tanh"))
        return x + mod[5] * h

    img = post_attn(img, img_attn, im1, "img")
    txt = post_attn(txt, txt_attn, tm1, "txt")
    concepts = post_attn(concepts, con_attn, cm1, "txt")
    return img, txt, concepts, d


def single_block(sd, p, nh, x, vec, rope_ti):
    """ModifiedSingleStreamBlock.forward, modified_single_stream_block.py:43-56."""
    H = x.shape[-1]
    shift, scale, gate = modulation(sd, p + "modulation", vec, 3)
    xm = (1 + scale) * layer_norm(x) + shift
    y = linear(sd, p + "linear1", xm)
    qkv, mlp = y[..., : 3 * H], y[..., 3 * H:]
    q, k, v = _split_heads(qkv, nh)
    q = apply_rope(rms_norm(q, sd[p + "norm.query_norm.scale"]), *rope_ti)
    k = apply_rope(rms_norm(k, sd[p + "norm.key_norm.scale"]), *rope_ti)
    attn = _merge_heads(sdpa(q, k, v))
    out = linear(sd, p + "linear2", torch.cat((attn, F.gelu(mlp, approximate="tanh")), 2))
    return x + gate * out


# --------------------------------------------------------------------------- model
def dit_forward(sd, g: Geometry, img, img_ids, txt, txt_ids, concepts, concept_ids, concept_vec,
                timesteps, y, guidance=None, stop_after_multimodal_attentions=False,
                joint_attention_kwargs=None, collect_block_outputs=False):
    """ModifiedFluxDiT.forward, modified_flux_dit.py:75-163.  Batch size 1.

    Returns (pred | None, dict) with each dict value stacked over the ``depth`` double blocks."""
    if img.ndim != 3 or txt.ndim != 3:
        raise ValueError("Input img and txt tensors must have 3 dimensions.")
    img, txt, concepts = img.float(), txt.float(), concepts.float()
    nh = g.num_heads
    img = linear(sd, "img_in", img)
    temb = timestep_embedding(timesteps)
    vec = mlp_embedder(sd, "time_in", temb)
    cvec = mlp_embedder(sd, "time_in", temb)
    if g.guidance_embed:
        if guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        gemb = mlp_embedder(sd, "guidance_in", timestep_embedding(guidance))
        vec, cvec = vec + gemb, cvec + gemb
    vec = vec + mlp_embedder(sd, "vector_in", y.float())
    cvec = cvec + mlp_embedder(sd, "vector_in", concept_vec.float())
    txt = linear(sd, "txt_in", txt)
    concepts = linear(sd, "txt_in", concepts)
    rope_ti = rope_cos_sin(torch.cat((txt_ids[0], img_ids[0]), 0), g.axes_dim, g.theta)
    rope_ci = rope_cos_sin(torch.cat((concept_ids[0], img_ids[0]), 0), g.axes_dim, g.theta)

    out = {k: [] for k in DICT_KEYS}
    block_outs = []
    for i in range(g.depth):
        img, txt, concepts, d = double_block(sd, f"double_blocks.{i}.", nh, img, txt, vec, rope_ti,
                                             concepts, cvec, rope_ci, joint_attention_kwargs)
        for k in DICT_KEYS:
            out[k].append(d[k])
        if collect_block_outputs:
            block_outs.append((img, txt, concepts))
    out = {k: torch.stack(v, 0) for k, v in out.items()}
    if collect_block_outputs:
        out["_block_outputs"] = block_outs
    if stop_after_multimodal_attentions:
        return None, out
    x = torch.cat((txt, img), 1)
    for i in range(g.depth_single_blocks):
        x = single_block(sd, f"single_blocks.{i}.", nh, x, vec, rope_ti)
    x = x[:, txt.shape[1]:]
    # LastLayer, flux/modules/layers.py:242-253
    shift, scale = linear(sd, "final_layer.adaLN_modulation.1", F.silu(vec)).chunk(2, dim=1)
    x = (1 + scale[:, None, :]) * layer_norm(x) + shift[:, None, :]
    return linear(sd, "final_layer.linear", x), out


# --------------------------------------------------------------------------- heatmaps
def heatmap_logits(image_vectors, concept_vectors):
    """Head-merge + einsum of compute_heatmaps_from_vectors, concept_attention_pipeline.py:43-61.

    -> [time, layers, batch, concepts, patches]"""
    if image_vectors.ndim == 6:
        t, l, b, h, n, d = image_vectors.shape
        image_vectors = image_vectors.permute(0, 1, 2, 4, 3, 5).reshape(t, l, b, n, h * d)
        c = concept_vectors.shape[4]
        concept_vectors = concept_vectors.permute(0, 1, 2, 4, 3, 5).reshape(t, l, b, c, h * d)
    return torch.einsum("tlbpd,tlbcd->tlbcp", image_vectors.float(), concept_vectors.float())


def linear_normalization(x, dim):
    """concept_attention/utils.py:35-44."""
    x = x - x.min(dim=dim, keepdim=True)[0]
    s = x.sum(dim=dim, keepdim=True)
    return x / torch.where(s == 0, torch.ones_like(s), s)


def compute_heatmaps(image_vectors, concept_vectors, layer_indices, timesteps, softmax=True,
                     normalize_concepts=False, side=None):
    """compute_heatmaps_from_vectors (softmax branch), concept_attention_pipeline.py:29-91, in fp32.

    ``side`` generalises the reference's hard-coded 64x64 grid (:85-90) to sqrt(patches)."""
    if not softmax:
        raise NotImplementedError("entmax15/sparsemax come from the un-pinned third-party `entmax` "
                                  "package (parity unpinned, SURVEY.md §8c)")
    if normalize_concepts and concept_vectors.ndim == 6:
        t, l, b, h, c, d = concept_vectors.shape
        cm = concept_vectors.permute(0, 1, 2, 4, 3, 5).reshape(t, l, b, c, h * d)
        cm = linear_normalization(cm.float(), dim=-2)
        t_, l_, b_, h_, n_, d_ = image_vectors.shape
        im = image_vectors.permute(0, 1, 2, 4, 3, 5).reshape(t_, l_, b_, n_, h_ * d_)
        logits = torch.einsum("tlbpd,tlbcd->tlbcp", im.float(), cm)
    elif normalize_concepts:
        logits = torch.einsum("tlbpd,tlbcd->tlbcp", image_vectors.float(),
                              linear_normalization(concept_vectors.float(), dim=-2))
    else:
        logits = heatmap_logits(image_vectors, concept_vectors)
    hm = torch.softmax(logits, dim=-2)
    hm = hm[list(timesteps)][:, list(layer_indices)]
    hm = hm.mean(dim=(0, 1))
    n = hm.shape[-1]
    side = side or int(round(math.sqrt(n)))
    return hm.reshape(hm.shape[0], hm.shape[1], side, side)


# --------------------------------------------------------------------------- sampler
def time_shift(mu, sigma, t):
    """flux/sampling.py:67-68."""
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


def get_schedule(num_steps, image_seq_len, base_shift=0.5, max_shift=1.15, shift=True):
    """flux/sampling.py:78-94."""
    ts = torch.linspace(1, 0, num_steps + 1)
    if shift:
        m = (max_shift - base_shift) / (4096 - 256)
        mu = m * image_seq_len + (base_shift - m * 256)
        ts = time_shift(mu, 1.0, ts)
    return ts.tolist()


def patchify(x):
    """prepare(): 'b c (h ph) (w pw) -> b (h w) (c ph pw)', flux/sampling.py:36."""
    b, c, hh, ww = x.shape
    x = x.view(b, c, hh // 2, 2, ww // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(b, (hh // 2) * (ww // 2), c * 4)


def unpack(x, height, width):
    """flux/sampling.py:154-162."""
    h, w = math.ceil(height / 16), math.ceil(width / 16)
    b, _, cpp = x.shape
    c = cpp // 4
    return x.view(b, h, w, c, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(b, c, h * 2, w * 2)


def make_img_ids(h2, w2):
    """prepare(): img_ids[...,1]=row, [...,2]=col; token index = row*w2+col (flux/sampling.py:40-43)."""
    ids = torch.zeros(h2, w2, 3)
    ids[..., 1] = torch.arange(h2)[:, None]
    ids[..., 2] = torch.arange(w2)[None, :]
    return ids.reshape(1, h2 * w2, 3)


def denoise(sd, g, img, img_ids, txt, txt_ids, vec, timesteps, guidance, concepts, concept_ids,
            concept_vec, joint_attention_kwargs=None):
    """denoise, flux/sampling.py:96-152: sequential Euler loop; dict entries stacked over time."""
    out = {k: [] for k in DICT_KEYS}
    gvec = torch.full((img.shape[0],), float(guidance))
    for t_curr, t_prev in zip(timesteps[:-1], timesteps[1:]):
        t_vec = torch.full((img.shape[0],), float(t_curr))
        pred, d = dit_forward(sd, g, img, img_ids, txt, txt_ids, concepts, concept_ids, concept_vec,
                              t_vec, vec, gvec, joint_attention_kwargs=joint_attention_kwargs)
        img = img + (t_prev - t_curr) * pred
        for k in DICT_KEYS:
            out[k].append(d[k])
    return img, {k: torch.stack(v, 0) for k, v in out.items()}
