"""Generate tests/golden/*.npz by running the REFERENCE's own modules.  TEST INFRASTRUCTURE ONLY.

Runs in the build container only (needs /root/reference, which never travels to the GPU box):

    python -B oracle/make_goldens.py

The reference has no tests or golden vectors of its own, so parity is pinned by
importing its hot-path modules here (recipe: SURVEY.md §8c), driving them with the seeded
state-dict / inputs of ``conceptattention_amd.weights`` and committing inputs' checksums and
outputs as data.  Nothing from the reference's source is written to the fixtures.

Fixtures
  tiny_schnell.npz / tiny_dev.npz   full ModifiedFluxDiT forward, H=256 NH=2 depth 2/2, 16x16 patches
  tiny_ablation.npz                 joint_attention_kwargs branches of the double block
  block_full.npz                    ONE full-size double block (H=3072, L=4096, T=256, C=4)
  single_full.npz                   ONE full-size single block
  block_full_dev.npz                ONE full-size double block at the flux-dev token counts (T=512, C=8)
  block_full_peaky.npz              the block_full case with peaky joint-attention logits (std ~8 nats; a cold text tile)
  tiny_peaky.npz                    the tiny FULL model (2 double + 2 single blocks) with every key-norm scale x 8
  heatmap_kat.npz                   compute_heatmaps_from_vectors known answers (softmax branch)
  sampler.npz                       get_schedule / prepare-patchify / unpack / denoise (tiny, 2 steps)
  metrics.npz                       segmentation scores of concept_attention/utils.py on seeded masks / maps
  timestep_embedding_bf16.npz       the reference's timestep_embedding on bf16 timesteps (its production dtype)
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np
import torch


def _import_reference():
    """Model-level recipe (i) + pipeline-level stubs (ii) of SURVEY.md §8c."""
    import transformers  # noqa: F401  (must be imported before the stubs below)
    pkg = types.ModuleType("concept_attention")
    pkg.__path__ = [REF + "/concept_attention"]
    sys.modules["concept_attention"] = pkg

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("entmax", entmax15=None, sparsemax=None)

    class _WM:
        def set_watermark(self, *a, **k):
            pass

        def encode(self, x, *a, **k):
            return x
    stub("imwatermark", WatermarkEncoder=_WM)
    stub("fire", Fire=lambda *a, **k: None)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms", Compose=None, ToTensor=None, Lambda=None)
    tv.transforms.functional = stub("torchvision.transforms.functional")
    from concept_attention.modified_flux_dit import ModifiedFluxDiT, FluxParams
    from concept_attention.modified_double_stream_block import ModifiedDoubleStreamBlock
    from concept_attention.modified_single_stream_block import ModifiedSingleStreamBlock
    from concept_attention.concept_attention_pipeline import compute_heatmaps_from_vectors
    from concept_attention.flux.src.flux import sampling
    return dict(ModifiedFluxDiT=ModifiedFluxDiT, FluxParams=FluxParams,
                ModifiedDoubleStreamBlock=ModifiedDoubleStreamBlock,
                ModifiedSingleStreamBlock=ModifiedSingleStreamBlock,
                compute_heatmaps_from_vectors=compute_heatmaps_from_vectors, sampling=sampling)


def _checksum(t: torch.Tensor) -> np.ndarray:
    t = t.double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * torch.arange(t.numel()) % 7).sum().item()])


def _ref_params(ref, p):
    return ref["FluxParams"](in_channels=p.in_channels, vec_in_dim=p.vec_in_dim,
                             context_in_dim=p.context_in_dim, hidden_size=p.hidden_size,
                             mlp_ratio=p.mlp_ratio, num_heads=p.num_heads, depth=p.depth,
                             depth_single_blocks=p.depth_single_blocks, axes_dim=list(p.axes_dim),
                             theta=p.theta, qkv_bias=p.qkv_bias, guidance_embed=p.guidance_embed)


def tiny_model(ref, out_dir, guidance_embed, name):
    from conceptattention_amd.params import tiny_params
    from conceptattention_amd.weights import synthetic_state_dict, synthetic_inputs
    from oracle.flux_oracle import patchify
    p = tiny_params(guidance_embed=guidance_embed)
    sd = synthetic_state_dict(p, seed=1)
    model = ref["ModifiedFluxDiT"](_ref_params(ref, p)).eval()
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    inp = synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=3, seed=2)
    img = patchify(inp["latent"])
    t = torch.tensor([0.75])
    gvec = torch.tensor([3.5])
    with torch.no_grad():
        pred, d = model(img=img, img_ids=inp["img_ids"], txt=inp["txt"], txt_ids=inp["txt_ids"],
                        concepts=inp["concepts"], concept_ids=inp["concept_ids"],
                        concept_vec=inp["concept_vec"], y=inp["vec"], timesteps=t, guidance=gvec)
        none_pred, d2 = model(img=img, img_ids=inp["img_ids"], txt=inp["txt"], txt_ids=inp["txt_ids"],
                              concepts=inp["concepts"], concept_ids=inp["concept_ids"],
                              concept_vec=inp["concept_vec"], y=inp["concept_vec"], timesteps=t,
                              guidance=gvec, stop_after_multimodal_attentions=True)
    assert none_pred is None
    arrays = {"pred": pred.numpy(), "timestep": t.numpy(), "guidance": gvec.numpy(),
              "in_checksum_img": _checksum(img), "in_checksum_txt": _checksum(inp["txt"]),
              "w_checksum": _checksum(sd["double_blocks.1.img_mlp.2.weight"])}
    for k, v in d.items():
        arrays[k] = v.numpy()
    for k, v in d2.items():
        arrays["stop_" + k] = v.numpy()
    np.savez_compressed(os.path.join(out_dir, name), **arrays)
    print(name, {k: v.shape for k, v in arrays.items()})


def tiny_peaky_state_dict(sd):
    """Every key_norm.scale of the model x 8 (double blocks: both streams; single blocks), rounded to bf16: the joint-
    attention logits of all 4 blocks at std ~8 nats; the v third of every qkv / linear1 projection x 0.25 (as in
    full_block_case.PEAKY_CASES["iid8"]: peaky rows return single value vectors, and without it the output-space logits
    -- std 6.8, up to 44 -- saturate the softmax over the concepts).  Shared with the tests through this module."""
    out = {}
    for k, v in sd.items():
        v = v.bfloat16().float()
        if k.endswith("key_norm.scale"):
            v = (v * 8.0).bfloat16().float()
        elif k.endswith(("_attn.qkv.weight", "_attn.qkv.bias", "linear1.weight", "linear1.bias")):
            H = v.shape[1] if v.dim() == 2 else (v.shape[0] // 3 if "qkv" in k else None)
            if H is None:   # linear1.bias: [3H + mlp]; H from the matching weight
                H = sd[k.replace(".bias", ".weight")].shape[1]
            v = v.clone()
            v[2 * H:3 * H] *= 0.25
        out[k] = v
    return out


def tiny_peaky(ref, out_dir):
    """Model-level counterpart of block_full_peaky: the whole ModifiedFluxDiT forward (double AND single blocks, final
    layer) at peaky logits, fp32, on bf16-representable weights and inputs."""
    from conceptattention_amd.params import tiny_params
    from conceptattention_amd.weights import synthetic_state_dict, synthetic_inputs
    from oracle.flux_oracle import patchify
    p = tiny_params()
    sd = tiny_peaky_state_dict(synthetic_state_dict(p, seed=1))
    model = ref["ModifiedFluxDiT"](_ref_params(ref, p)).eval()
    model.load_state_dict(sd, strict=True)
    inp = {k: (v.bfloat16().float() if v.is_floating_point() else v)
           for k, v in synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=3, seed=2).items()}
    img = patchify(inp["latent"])
    t = torch.tensor([0.75])
    with torch.no_grad():
        pred, d = model(img=img, img_ids=inp["img_ids"], txt=inp["txt"], txt_ids=inp["txt_ids"],
                        concepts=inp["concepts"], concept_ids=inp["concept_ids"],
                        concept_vec=inp["concept_vec"], y=inp["vec"], timesteps=t, guidance=torch.tensor([0.0]))
    # (the reference's compute_heatmaps_from_vectors is hard-wired to 64 x 64 patches: the tests form the maps of these
    # vectors with the oracle's reduction, which heatmap_kat.npz pins)
    arrays = {"pred": pred.numpy(), "timestep": t.numpy(),
              "key_scale_checksum": _checksum(sd["single_blocks.1.norm.key_norm.scale"])}
    for k, v in d.items():
        arrays[k] = v.numpy()
    np.savez_compressed(os.path.join(out_dir, "tiny_peaky.npz"), **arrays)
    print("tiny_peaky: pred max", pred.abs().max().item())


def tiny_ablation(ref, out_dir):
    from conceptattention_amd.params import tiny_params
    from conceptattention_amd.weights import synthetic_state_dict, synthetic_inputs
    from oracle.flux_oracle import patchify
    p = tiny_params(depth=1, depth_single_blocks=0)
    sd = synthetic_state_dict(p, seed=1)
    model = ref["ModifiedFluxDiT"](_ref_params(ref, p)).eval()
    model.load_state_dict(sd, strict=True)
    inp = synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=3, seed=2)
    img = patchify(inp["latent"])
    arrays = {}
    for cross in (True, False):
        for self_ in (True, False):
            with torch.no_grad():
                _, d = model(img=img, img_ids=inp["img_ids"], txt=inp["txt"], txt_ids=inp["txt_ids"],
                             concepts=inp["concepts"], concept_ids=inp["concept_ids"],
                             concept_vec=inp["concept_vec"], y=inp["vec"], timesteps=torch.tensor([0.5]),
                             stop_after_multimodal_attentions=True,
                             joint_attention_kwargs={"concept_cross_attention": cross,
                                                     "concept_self_attention": self_})
            tag = f"cross{int(cross)}_self{int(self_)}_"
            arrays[tag + "output_space_concept_vectors"] = d["output_space_concept_vectors"].numpy()
            arrays[tag + "cross_attention_concept_vectors"] = d["cross_attention_concept_vectors"].numpy()
    np.savez_compressed(os.path.join(out_dir, "tiny_ablation.npz"), **arrays)
    print("tiny_ablation", list(arrays))


def full_blocks(ref, out_dir):
    """One full-size double block and one full-size single block (fp32, seeded bf16-representable
    weights and inputs so the HIP bf16 path sees the *same* numbers)."""
    from conceptattention_amd.params import FluxParams
    from conceptattention_amd.weights import synthetic_state_dict
    from oracle.full_block_case import full_block_inputs
    p = FluxParams()
    H, NH = p.hidden_size, p.num_heads
    case = full_block_inputs(p)
    # ---------------- double block
    sd = synthetic_state_dict(p, seed=0, prefix="double_blocks.0.")
    sd = {k[len("double_blocks.0."):]: v.bfloat16().float() for k, v in sd.items()}
    blk = ref["ModifiedDoubleStreamBlock"](H, NH, mlp_ratio=p.mlp_ratio, qkv_bias=True).eval()
    blk.load_state_dict(sd, strict=True)
    from concept_attention.flux.src.flux.modules.layers import EmbedND
    emb = EmbedND(dim=128, theta=p.theta, axes_dim=list(p.axes_dim))
    pe = emb(torch.cat((case["txt_ids"], case["img_ids"]), 1))
    cpe = emb(torch.cat((case["concept_ids"], case["img_ids"]), 1))
    with torch.no_grad():
        img, txt, con, d = blk(img=case["img"], txt=case["txt"], vec=case["vec"], pe=pe,
                               concepts=case["concepts"], concept_vec=case["concept_vec"], concept_pe=cpe)
    hm_fn = ref["compute_heatmaps_from_vectors"]
    st = {k: v[None, None] for k, v in d.items()}  # [time=1, layers=1, batch, ...]
    hm_out = hm_fn(st["output_space_image_vectors"], st["output_space_concept_vectors"],
                   layer_indices=[0], timesteps=[0], softmax=True)
    hm_cross = hm_fn(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"],
                     layer_indices=[0], timesteps=[0], softmax=True)
    logits_out = torch.einsum("bpd,bcd->bcp", d["output_space_image_vectors"], d["output_space_concept_vectors"])
    ciq = d["cross_attention_image_vectors"].permute(0, 2, 1, 3).reshape(1, -1, H)
    ccq = d["cross_attention_concept_vectors"].permute(0, 2, 1, 3).reshape(1, -1, H)
    logits_cross = torch.einsum("bpd,bcd->bcp", ciq, ccq)
    rows = case["sample_rows"]
    arrays = dict(
        heatmap_output_space=hm_out.numpy(), heatmap_cross_attention=hm_cross.numpy(),
        logits_output_space=logits_out.numpy(), logits_cross_attention=logits_cross.numpy(),
        concept_attn=d["output_space_concept_vectors"].numpy(),
        concept_q=d["cross_attention_concept_vectors"].numpy(),
        img_attn_rows=d["output_space_image_vectors"][0, rows].numpy(),
        img_q_rows=d["cross_attention_image_vectors"][0, :, rows].numpy(),
        img_out_rows=img[0, rows].numpy(), txt_out=txt[0, ::8].numpy(), concepts_out=con.numpy(),
        sample_rows=rows.numpy(),
        img_out_checksum=_checksum(img), in_checksum=_checksum(case["img"]),
        w_checksum=_checksum(sd["img_mlp.2.weight"]),
    )
    np.savez_compressed(os.path.join(out_dir, "block_full.npz"), **arrays)
    print("block_full heat range", hm_out.min().item(), hm_out.max().item(),
          "cross", hm_cross.min().item(), hm_cross.max().item())
    del blk, sd
    # ---------------- single block
    sd = synthetic_state_dict(p, seed=0, prefix="single_blocks.0.")
    sd = {k[len("single_blocks.0."):]: v.bfloat16().float() for k, v in sd.items()}
    sblk = ref["ModifiedSingleStreamBlock"](H, NH, mlp_ratio=p.mlp_ratio).eval()
    sblk.load_state_dict(sd, strict=True)
    x = torch.cat((case["txt"], case["img"]), 1)
    with torch.no_grad():
        y = sblk(x, vec=case["vec"], pe=pe)
    srows = torch.cat((torch.arange(0, 256, 16), 256 + rows))
    np.savez_compressed(os.path.join(out_dir, "single_full.npz"), out_rows=y[0, srows].numpy(),
                        sample_rows=srows.numpy(), out_checksum=_checksum(y))
    print("single_full done")


def full_block_dev(ref, out_dir):
    """Full-size double block at the flux-dev token counts (BASELINE.json configs[2]): T=512, C=8."""
    from conceptattention_amd.params import FluxParams
    from conceptattention_amd.weights import synthetic_state_dict
    from oracle.full_block_case import full_block_inputs
    from concept_attention.flux.src.flux.modules.layers import EmbedND
    p = FluxParams(guidance_embed=True)
    H, NH = p.hidden_size, p.num_heads
    case = full_block_inputs(p, T=512, C=8, seed=8)
    sd = synthetic_state_dict(p, seed=0, prefix="double_blocks.0.")
    sd = {k[len("double_blocks.0."):]: v.bfloat16().float() for k, v in sd.items()}
    blk = ref["ModifiedDoubleStreamBlock"](H, NH, mlp_ratio=p.mlp_ratio, qkv_bias=True).eval()
    blk.load_state_dict(sd, strict=True)
    emb = EmbedND(dim=128, theta=p.theta, axes_dim=list(p.axes_dim))
    pe = emb(torch.cat((case["txt_ids"], case["img_ids"]), 1))
    cpe = emb(torch.cat((case["concept_ids"], case["img_ids"]), 1))
    with torch.no_grad():
        img, txt, con, d = blk(img=case["img"], txt=case["txt"], vec=case["vec"], pe=pe,
                               concepts=case["concepts"], concept_vec=case["concept_vec"], concept_pe=cpe)
    st = {k: v[None, None] for k, v in d.items()}
    fn = ref["compute_heatmaps_from_vectors"]
    hm_out = fn(st["output_space_image_vectors"], st["output_space_concept_vectors"], layer_indices=[0],
                timesteps=[0], softmax=True)
    hm_cross = fn(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"], layer_indices=[0],
                  timesteps=[0], softmax=True)
    rows = case["sample_rows"]
    np.savez_compressed(os.path.join(out_dir, "block_full_dev.npz"),
                        heatmap_output_space=hm_out.numpy(), heatmap_cross_attention=hm_cross.numpy(),
                        concept_attn=d["output_space_concept_vectors"].numpy(),
                        img_attn_rows=d["output_space_image_vectors"][0, rows].numpy(),
                        img_out_rows=img[0, rows].numpy(), txt_out=txt[0, ::16].numpy(), concepts_out=con.numpy(),
                        sample_rows=rows.numpy())
    print("block_full_dev heat range", hm_out.min().item(), hm_out.max().item())


def full_block_peaky(ref, out_dir):
    """block_full's geometry and inputs with the QK-norm scales / qkv biases of oracle/full_block_case.PEAKY_CASES:
    the joint-attention logits at std ~8 nats, and a structured case with the text tile ~20 nats below the image
    keys.  Same outputs as block_full.npz per case (key prefix '<case>_'), plus statistics of the joint-attention
    logits the reference's own q / k produce (recomputed here from the block's submodules)."""
    from conceptattention_amd.params import FluxParams
    from conceptattention_amd.weights import synthetic_state_dict
    from oracle.full_block_case import PEAKY_CASES, full_block_inputs, peaky_state_dict
    from concept_attention.flux.src.flux.modules.layers import EmbedND
    from concept_attention.flux.src.flux.math import apply_rope
    from einops import rearrange
    p = FluxParams()
    H, NH = p.hidden_size, p.num_heads
    case = full_block_inputs(p)
    base = synthetic_state_dict(p, seed=0, prefix="double_blocks.0.")
    base = {k[len("double_blocks.0."):]: v.bfloat16().float() for k, v in base.items()}
    emb = EmbedND(dim=128, theta=p.theta, axes_dim=list(p.axes_dim))
    pe = emb(torch.cat((case["txt_ids"], case["img_ids"]), 1))
    cpe = emb(torch.cat((case["concept_ids"], case["img_ids"]), 1))
    hm_fn = ref["compute_heatmaps_from_vectors"]
    rows = case["sample_rows"]
    arrays = {"sample_rows": rows.numpy()}
    for name in PEAKY_CASES:
        sd = peaky_state_dict(base, name, H)
        blk = ref["ModifiedDoubleStreamBlock"](H, NH, mlp_ratio=p.mlp_ratio, qkv_bias=True).eval()
        blk.load_state_dict(sd, strict=True)
        with torch.no_grad():
            img, txt, con, d = blk(img=case["img"], txt=case["txt"], vec=case["vec"], pe=pe,
                                   concepts=case["concepts"], concept_vec=case["concept_vec"], concept_pe=cpe)
            # statistics of the joint-attention logits (heads 0, 7, 23; every 16th image query row), from the block's own
            # submodules exactly as its forward forms q and k (modified_double_stream_block.py:88-111)
            im1, _ = blk.img_mod(case["vec"])
            tm1, _ = blk.txt_mod(case["vec"])
            xi = (1 + im1.scale) * blk.img_norm1(case["img"]) + im1.shift
            xt = (1 + tm1.scale) * blk.txt_norm1(case["txt"]) + tm1.shift
            iq, ik, iv = rearrange(blk.img_attn.qkv(xi), "B L (K H D) -> K B H L D", K=3, H=NH)
            iq, ik = blk.img_attn.norm(iq, ik, iv)
            tq, tk, tv = rearrange(blk.txt_attn.qkv(xt), "B L (K H D) -> K B H L D", K=3, H=NH)
            tq, tk = blk.txt_attn.norm(tq, tk, tv)
            q, k = apply_rope(torch.cat((tq, iq), 2), torch.cat((tk, ik), 2), pe)
            T = case["txt"].shape[1]
            lg = (q[0, [0, 7, 23]][:, T::16] @ k[0, [0, 7, 23]].transpose(-1, -2)) / 128 ** 0.5   # [3, 256, T+L] nats
            stats = np.array([lg.std().item(), (lg.max(-1).values - lg[..., :64].max(-1).values).max().item(),
                              (lg[..., T:].mean() - lg[..., :T].mean()).item(), lg.abs().max().item()])
        st = {k_: v[None, None] for k_, v in d.items()}
        hm_out = hm_fn(st["output_space_image_vectors"], st["output_space_concept_vectors"],
                       layer_indices=[0], timesteps=[0], softmax=True)
        hm_cross = hm_fn(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"],
                         layer_indices=[0], timesteps=[0], softmax=True)
        logits_out = torch.einsum("bpd,bcd->bcp", d["output_space_image_vectors"], d["output_space_concept_vectors"])
        ciq = d["cross_attention_image_vectors"].permute(0, 2, 1, 3).reshape(1, -1, H)
        ccq = d["cross_attention_concept_vectors"].permute(0, 2, 1, 3).reshape(1, -1, H)
        logits_cross = torch.einsum("bpd,bcd->bcp", ciq, ccq)
        arrays.update({
            f"{name}_heatmap_output_space": hm_out.numpy(), f"{name}_heatmap_cross_attention": hm_cross.numpy(),
            f"{name}_logits_output_space": logits_out.numpy(), f"{name}_logits_cross_attention": logits_cross.numpy(),
            f"{name}_concept_attn": d["output_space_concept_vectors"].numpy(),
            f"{name}_concept_q": d["cross_attention_concept_vectors"].numpy(),
            f"{name}_img_attn_rows": d["output_space_image_vectors"][0, rows].numpy(),
            f"{name}_img_q_rows": d["cross_attention_image_vectors"][0, :, rows].numpy(),
            f"{name}_img_out_rows": img[0, rows].numpy(), f"{name}_txt_out": txt[0, ::8].numpy(),
            f"{name}_concepts_out": con.numpy(),
            f"{name}_joint_logit_stats": stats,   # [std, max over rows of (row max - max of its first 64 keys), mean image key - mean text key, max |logit|] in nats
            f"{name}_key_scale_checksum": _checksum(sd["img_attn.norm.key_norm.scale"]),
        })
        # the yardstick of DESIGN.md section 2: the REFERENCE's own block in bf16 (its production dtype) on the same
        # weights and inputs, its vectors reduced in fp32 -- how far a bf16 run is from the fp32 values above
        bf = torch.bfloat16
        blk16 = ref["ModifiedDoubleStreamBlock"](H, NH, mlp_ratio=p.mlp_ratio, qkv_bias=True).eval()
        blk16.load_state_dict(sd, strict=True)
        blk16 = blk16.to(bf)
        with torch.no_grad():
            i16, t16, c16, d16 = blk16(img=case["img"].to(bf), txt=case["txt"].to(bf), vec=case["vec"].to(bf), pe=pe,
                                       concepts=case["concepts"].to(bf), concept_vec=case["concept_vec"].to(bf),
                                       concept_pe=cpe)
        st16 = {k_: v.float()[None, None] for k_, v in d16.items()}
        h16o = hm_fn(st16["output_space_image_vectors"], st16["output_space_concept_vectors"], layer_indices=[0],
                     timesteps=[0], softmax=True)
        h16c = hm_fn(st16["cross_attention_image_vectors"], st16["cross_attention_concept_vectors"], layer_indices=[0],
                     timesteps=[0], softmax=True)
        arrays[f"{name}_refbf16_err"] = np.array([
            (h16o - hm_out).abs().max().item(), (h16c - hm_cross).abs().max().item(),
            (d16["output_space_concept_vectors"].float() - d["output_space_concept_vectors"]).abs().max().item(),
            (d16["output_space_image_vectors"].float() - d["output_space_image_vectors"]).abs().max().item(),
            (i16.float() - img).abs().max().item()])   # [map out, map cross, concept_attn, img_attn, img_out] max-abs
        print(f"block_full_peaky[{name}]: reference in bf16 vs fp32: ", arrays[f"{name}_refbf16_err"])
        del blk16
        print(f"block_full_peaky[{name}]: joint logits std {stats[0]:.2f} nats, late-max gap {stats[1]:.1f}, image - text "
              f"{stats[2]:.1f}, max |logit| {stats[3]:.1f}; heat range {hm_out.min().item():.4f} .. {hm_out.max().item():.4f}, "
              f"cross {hm_cross.min().item():.4f} .. {hm_cross.max().item():.4f}")
        del blk
    np.savez_compressed(os.path.join(out_dir, "block_full_peaky.npz"), **arrays)


def heatmap_kat(ref, out_dir):
    fn = ref["compute_heatmaps_from_vectors"]
    g = torch.Generator().manual_seed(11)
    iv = torch.randn(3, 5, 1, 4096, 16, generator=g)
    cv = torch.randn(3, 5, 1, 4, 16, generator=g)
    a = fn(iv, cv, layer_indices=[1, 3, 4], timesteps=[0, 2], softmax=True)
    b = fn(iv, cv, layer_indices=[2], timesteps=[1], softmax=True, normalize_concepts=True)
    iv6 = torch.randn(2, 2, 1, 2, 4096, 8, generator=g)
    cv6 = torch.randn(2, 2, 1, 2, 5, 8, generator=g)
    c = fn(iv6, cv6, layer_indices=[0, 1], timesteps=[0, 1], softmax=True)
    np.savez_compressed(os.path.join(out_dir, "heatmap_kat.npz"),
                        iv=iv.half().numpy(), cv=cv.half().numpy(), iv6=iv6.half().numpy(),
                        cv6=cv6.half().numpy(),
                        out_a=fn(iv.half().float(), cv.half().float(), layer_indices=[1, 3, 4],
                                 timesteps=[0, 2], softmax=True).numpy(),
                        out_b=fn(iv.half().float(), cv.half().float(), layer_indices=[2], timesteps=[1],
                                 softmax=True, normalize_concepts=True).numpy(),
                        out_c=fn(iv6.half().float(), cv6.half().float(), layer_indices=[0, 1],
                                 timesteps=[0, 1], softmax=True).numpy())
    print("heatmap_kat", a.shape, b.shape, c.shape)


def sampler(ref, out_dir):
    from conceptattention_amd.params import tiny_params
    from conceptattention_amd.weights import synthetic_state_dict, synthetic_inputs
    s = ref["sampling"]
    arrays = {
        "schedule_schnell_4": np.array(s.get_schedule(4, 4096, shift=False)),
        "schedule_dev_50_4096": np.array(s.get_schedule(50, 4096, shift=True)),
        "schedule_dev_28_1024": np.array(s.get_schedule(28, 1024, shift=True)),
    }
    x = torch.arange(16 * 8 * 12, dtype=torch.float32).reshape(1, 16, 8, 12)
    from einops import rearrange
    packed = rearrange(x, "b c (h ph) (w pw) -> b (h w) (c ph pw)", ph=2, pw=2)
    arrays["patchify_in"] = x.numpy()
    arrays["patchify_out"] = packed.numpy()
    arrays["unpack_out"] = s.unpack(packed, 64, 96).numpy()
    # tiny 2-step denoise + heatmaps through the reference sampler (64x64 patches so that the
    # reference's hard-coded 64x64 reshape, concept_attention_pipeline.py:85-90, is valid)
    p = tiny_params(depth=2, depth_single_blocks=1)
    sd = synthetic_state_dict(p, seed=3)
    model = ref["ModifiedFluxDiT"](_ref_params(ref, p)).eval()
    model.load_state_dict(sd, strict=True)
    inp = synthetic_inputs(p, 1024, 1024, n_txt=8, n_concepts=3, seed=4)
    img = rearrange(inp["latent"], "b c (h ph) (w pw) -> b (h w) (c ph pw)", ph=2, pw=2)
    ts = s.get_schedule(2, img.shape[1], shift=False)
    with torch.no_grad():
        out, _, d = s.denoise(model, img=img, img_ids=inp["img_ids"], txt=inp["txt"],
                              txt_ids=inp["txt_ids"], vec=inp["vec"], timesteps=ts, guidance=0.0,
                              concepts=inp["concepts"], concept_ids=inp["concept_ids"],
                              concept_vec=inp["concept_vec"])
    fn = ref["compute_heatmaps_from_vectors"]
    arrays["denoise_img_rows"] = out[0, ::64].numpy()
    arrays["denoise_img_checksum"] = _checksum(out)
    arrays["denoise_heatmaps"] = fn(d["output_space_image_vectors"], d["output_space_concept_vectors"],
                                    layer_indices=[0, 1], timesteps=[0, 1], softmax=True).numpy()
    arrays["denoise_cross_maps"] = fn(d["cross_attention_image_vectors"],
                                      d["cross_attention_concept_vectors"],
                                      layer_indices=[1], timesteps=[0, 1], softmax=True).numpy()
    np.savez_compressed(os.path.join(out_dir, "sampler.npz"), **arrays)
    print("sampler done", arrays["schedule_schnell_4"])


def timestep_embedding_bf16(ref, out_dir):
    """The reference's production run keeps the timestep in the activations' bf16 (t_vec / guidance_vec are built
    in img.dtype, flux/sampling.py:122-125) and `time_factor * t` is then a bf16 product (layers.py:37): at
    t = 0.75 it embeds 752, not 750.  Golden for HipFluxDiT(bf16_timesteps=True): the reference's own
    timestep_embedding on bf16 timesteps (both schedules + a guidance value)."""
    from concept_attention.flux.src.flux.modules.layers import timestep_embedding
    s = ref["sampling"]
    ts = s.get_schedule(4, 4096, shift=False)[:-1] + s.get_schedule(50, 4096, shift=True)[:-1] + [3.5, 0.0]
    t_bf = torch.tensor(ts, dtype=torch.bfloat16)
    emb = timestep_embedding(t_bf, 256)
    np.savez_compressed(os.path.join(out_dir, "timestep_embedding_bf16.npz"), t=np.array(ts, np.float64),
                        emb=emb.float().numpy(), t_bf16=t_bf.float().numpy(), arg_bf16=(1000.0 * t_bf).float().numpy())
    print("timestep_embedding_bf16", emb.shape)


def metrics(ref, out_dir):
    """pixel accuracy / IoU areas / AP of the reference's utils on seeded 2-class cases, driven the way
    experiments/imagenet_segmentation/run_experiment.py:205-219 drives them."""
    from concept_attention import utils as U
    g = torch.Generator().manual_seed(21)
    arrays = {}
    for i, (h, w) in enumerate([(224, 224), (64, 64), (17, 9)]):
        label = (torch.rand(h, w, generator=g) > 0.6).float()
        coeff = (0.6 * label + 0.8 * torch.rand(h, w, generator=g)).clamp(0, 1)
        coeff = (coeff * 16).round() / 16            # many tied scores: exercises the threshold grouping of AP
        mask = (coeff > coeff.mean()).float()
        m2, y2 = torch.stack((1 - mask, mask)), torch.stack((1 - label, label))
        cor, lab = U.batch_pix_accuracy(m2, y2)
        inter, union = U.batch_intersection_union(m2, y2, nclass=2)
        ap = U.get_ap_scores(torch.stack((1 - coeff, coeff)).unsqueeze(0), label.unsqueeze(0))
        arrays.update({f"label{i}": label.numpy().astype(np.uint8), f"coeff{i}": coeff.numpy(),
                       f"mask{i}": mask.numpy().astype(np.uint8), f"pix{i}": np.array([cor, lab]),
                       f"inter{i}": inter, f"union{i}": union, f"ap{i}": np.array(ap)})
    # multi-class label map with an ignored (-1) region, for get_ap_scores' ignore_index handling
    tgt = torch.randint(-1, 3, (12, 10), generator=g)
    pred = torch.rand(3, 12, 10, generator=g)
    arrays["mc_target"] = tgt.numpy()
    arrays["mc_pred"] = pred.numpy()
    arrays["mc_ap"] = np.array(U.get_ap_scores(pred.unsqueeze(0), tgt.unsqueeze(0)))
    x = torch.randn(3, 5, 7, generator=g)
    arrays["linnorm_in"] = x.numpy()
    arrays["linnorm_out"] = U.linear_normalization(x, dim=-2).numpy()
    np.savez_compressed(os.path.join(out_dir, "metrics.npz"), **arrays)
    print("metrics", arrays["pix0"], arrays["ap0"], arrays["mc_ap"])


def main():
    torch.set_num_threads(8)
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    ref = _import_reference()
    which = sys.argv[1:] or ["tiny", "ablation", "heatmap", "sampler", "metrics", "full", "fulldev", "temb", "peaky", "tinypeaky"]
    if "tiny" in which:
        tiny_model(ref, out_dir, False, "tiny_schnell.npz")
        tiny_model(ref, out_dir, True, "tiny_dev.npz")
    if "ablation" in which:
        tiny_ablation(ref, out_dir)
    if "heatmap" in which:
        heatmap_kat(ref, out_dir)
    if "sampler" in which:
        sampler(ref, out_dir)
    if "metrics" in which:
        metrics(ref, out_dir)
    if "temb" in which:
        timestep_embedding_bf16(ref, out_dir)
    if "full" in which:
        full_blocks(ref, out_dir)
    if "fulldev" in which:
        full_block_dev(ref, out_dir)
    if "peaky" in which:
        full_block_peaky(ref, out_dir)
    if "tinypeaky" in which:
        tiny_peaky(ref, out_dir)


if __name__ == "__main__":
    main()
