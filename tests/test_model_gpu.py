"""The HIP model path against the CPU oracle / the reference's golden vectors (GPU box).

Tolerance semantics (SURVEY.md §7 "hard parts"): the HIP path computes in bf16
with fp32 accumulation; the oracle is fp32 on the SAME bf16-representable weights and inputs.
Heat maps (softmax over concepts, values in [0,1]) must agree to <= 1e-3 max-abs per block;
activations to a few bf16 ulps of their magnitude."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conceptattention_amd import ops  # noqa: E402
from conceptattention_amd.flux_dit import DICT_KEYS, HeatmapRequest, HipFluxDiT, _Geom  # noqa: E402
from conceptattention_amd.heatmaps import compute_heatmaps_from_vectors  # noqa: E402
from conceptattention_amd.params import FluxParams, tiny_params  # noqa: E402
from conceptattention_amd.weights import synthetic_inputs, synthetic_state_dict  # noqa: E402
from oracle import flux_oracle as O  # noqa: E402

DEV = "cuda:0"


def bf(x):
    return x.bfloat16().float()


def maxabs(a, b):
    return (torch.as_tensor(a).float().cpu() - torch.as_tensor(b).float().cpu()).abs().max().item()


def tiny_case(guidance_embed=False, depth=2, singles=2, C=3, T=8, side=256):
    p = tiny_params(guidance_embed=guidance_embed, depth=depth, depth_single_blocks=singles)
    sd = {k: bf(v) for k, v in synthetic_state_dict(p, seed=1).items()}
    inp = {k: (bf(v) if v.is_floating_point() else v)
           for k, v in synthetic_inputs(p, side, side, n_txt=T, n_concepts=C, seed=2).items()}
    return p, sd, inp


def run_hip(p, sd, inp, t, guidance=None, **kw):
    m = HipFluxDiT(p, DEV)
    m.load_state_dict(sd)
    d = {k: v.to(DEV) for k, v in inp.items()}
    img = O.patchify(inp["latent"]).to(DEV)
    return m, m(img=img, img_ids=d["img_ids"], txt=d["txt"], txt_ids=d["txt_ids"], concepts=d["concepts"],
                concept_ids=d["concept_ids"], concept_vec=d["concept_vec"], y=d["vec"],
                timesteps=torch.tensor([t], device=DEV),
                guidance=None if guidance is None else torch.tensor([guidance], device=DEV), **kw)


def test_prediction_follows_the_latent_dtype():
    """bf16 latent in -> bf16 prediction out (the reference's call); fp32 latent in -> img_in over its hi + lo planes and
    an fp32 prediction.  For a bf16-representable latent the low plane is zero, so the two calls agree bit for bit
    after the one rounding."""
    p, sd, inp = tiny_case(False)
    lat = inp["latent"].bfloat16()
    _, (pred_b, _) = run_hip(p, sd, dict(inp, latent=lat), 0.75, 3.5)
    _, (pred_f, _) = run_hip(p, sd, dict(inp, latent=lat.float()), 0.75, 3.5)
    assert pred_b.dtype == torch.bfloat16 and pred_f.dtype == torch.float32
    assert torch.equal(pred_f.bfloat16(), pred_b)


@pytest.mark.parametrize("guidance_embed", [False, True])
def test_tiny_model_matches_oracle(guidance_embed):
    p, sd, inp = tiny_case(guidance_embed)
    t, g = 0.75, 3.5
    pred_o, d_o = O.dit_forward(sd, p, O.patchify(inp["latent"]), inp["img_ids"], inp["txt"], inp["txt_ids"],
                                inp["concepts"], inp["concept_ids"], inp["concept_vec"], torch.tensor([t]),
                                inp["vec"], torch.tensor([g]))
    _, (pred, d) = run_hip(p, sd, inp, t, g)
    # the prediction has the latent's type: fp32 in (the fp32 Euler state of sampling.denoise: hi + lo planes into
    # img_in, unrounded prediction out), bf16 in -> bf16 out (the reference's call)
    assert pred.shape == pred_o.shape and pred.dtype == torch.float32
    for k in DICT_KEYS:
        assert tuple(d[k].shape) == tuple(d_o[k].shape), k
        scale = d_o[k].abs().max().item()
        assert maxabs(d[k], d_o[k]) < 2e-2 * max(scale, 1.0), (k, maxabs(d[k], d_o[k]), scale)
    assert maxabs(pred, pred_o) < 3e-2 * max(pred_o.abs().max().item(), 1.0)
    # heat maps from the HIP vectors through the HIP reduction vs the oracle's fp32 maps
    st = {k: v[None] for k, v in d.items()}
    st_o = {k: v[None] for k, v in d_o.items()}
    hm = compute_heatmaps_from_vectors(st["output_space_image_vectors"], st["output_space_concept_vectors"],
                                       layer_indices=[0, 1], timesteps=[0])
    hm_o = O.compute_heatmaps(st_o["output_space_image_vectors"], st_o["output_space_concept_vectors"], [0, 1], [0])
    assert hm.shape == hm_o.shape == (1, 3, 16, 16)
    # two chained blocks (layer 1 sees layer 0's bf16 residual stream, not teacher-forced): 3e-3
    assert maxabs(hm, hm_o) < 3e-3


def test_tiny_model_peaky_logits_vs_reference_golden(golden):
    """Model level at peaky logits (round 5): the whole tiny model -- double blocks, single blocks (linear1's fused
    QK-norm + RoPE epilogue, the [text | image] attention), final layer -- with every key-norm scale x 8, against the
    REFERENCE's fp32 forward (tests/golden/tiny_peaky.npz).  Same relative gates as the std-1 tiny test; the heat maps
    of the HIP vectors against the maps of the reference's vectors (both through reductions pinned elsewhere)."""
    from oracle.make_goldens import tiny_peaky_state_dict
    g = golden("tiny_peaky.npz")
    p = tiny_params()
    sd = tiny_peaky_state_dict(synthetic_state_dict(p, seed=1))
    inp = {k: (bf(v) if v.is_floating_point() else v)
           for k, v in synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=3, seed=2).items()}
    for qk16 in ("captured", "0"):
        m = HipFluxDiT(p, DEV)
        m.qk_f16 = qk16
        m.load_state_dict(sd)
        d_in = {k: v.to(DEV) for k, v in inp.items()}
        pred, d = m(img=O.patchify(inp["latent"]).to(DEV), img_ids=d_in["img_ids"], txt=d_in["txt"], txt_ids=d_in["txt_ids"],
                    concepts=d_in["concepts"], concept_ids=d_in["concept_ids"], concept_vec=d_in["concept_vec"],
                    y=d_in["vec"], timesteps=torch.tensor([float(g["timestep"][0])], device=DEV),
                    guidance=torch.zeros(1, device=DEV))
        ref_pred = torch.from_numpy(g["pred"])
        e_pred = maxabs(pred, ref_pred)
        errs = {}
        for k in DICT_KEYS:
            ref = torch.from_numpy(g[k])
            assert tuple(d[k].shape) == tuple(ref.shape), k
            errs[k] = maxabs(d[k], ref) / max(ref.abs().max().item(), 1.0)
        st = {k: v[None] for k, v in d.items()}
        st_o = {k: torch.from_numpy(g[k])[None] for k in DICT_KEYS}
        hm = compute_heatmaps_from_vectors(st["output_space_image_vectors"], st["output_space_concept_vectors"],
                                           layer_indices=[0, 1], timesteps=[0])
        hm_o = O.compute_heatmaps(st_o["output_space_image_vectors"], st_o["output_space_concept_vectors"], [0, 1], [0])
        e_hm = maxabs(hm, hm_o)
        print(f"\n[measured] tiny model, key scales x 8, qk_f16={qk16}: pred {e_pred:.3e} (max |pred| "
              f"{ref_pred.abs().max().item():.2f}), vectors (rel.) {errs}, output-space maps {e_hm:.3e}")
        assert e_pred < 3e-2 * max(ref_pred.abs().max().item(), 1.0)
        assert all(v < 2e-2 for v in errs.values()), errs
        # measured on MI355X: 1.8e-3 with half-precision q / k, 4.2e-3 with bf16 (std-1 tiny test: < 3e-3 allowed); a
        # peaky row passes v's and P's 2^-9 on unaveraged, and the dict route carries bf16 vectors on both sides
        assert e_hm < 6.3e-3


def test_tiny_stop_after_multimodal_and_fused_heatmaps():
    p, sd, inp = tiny_case()
    _, d_o = O.dit_forward(sd, p, O.patchify(inp["latent"]), inp["img_ids"], inp["txt"], inp["txt_ids"],
                           inp["concepts"], inp["concept_ids"], inp["concept_vec"], torch.tensor([0.5]),
                           inp["concept_vec"], stop_after_multimodal_attentions=True)
    C, Lp = 3, 256
    req = HeatmapRequest((1,), 1.0, torch.zeros(C, Lp, device=DEV), torch.zeros(C, Lp, device=DEV))
    m = HipFluxDiT(p, DEV)
    m.load_state_dict(sd)
    d = {k: v.to(DEV) for k, v in inp.items()}
    pred, dd = m(img=O.patchify(inp["latent"]).to(DEV), img_ids=d["img_ids"], txt=d["txt"], txt_ids=d["txt_ids"],
                 concepts=d["concepts"], concept_ids=d["concept_ids"], concept_vec=d["concept_vec"],
                 y=d["concept_vec"], timesteps=torch.tensor([0.5], device=DEV),
                 stop_after_multimodal_attentions=True, return_vectors=False, heatmaps=req)
    assert pred is None and dd == {}
    st = {k: v[None] for k, v in d_o.items()}
    ho = O.compute_heatmaps(st["output_space_image_vectors"], st["output_space_concept_vectors"], [1], [0])
    hc = O.compute_heatmaps(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"], [1], [0])
    assert maxabs(req.out_space.view(1, C, 16, 16), ho) < 3e-3  # layer 1 of a chained pair (see above)
    assert maxabs(req.cross_space.view(1, C, 16, 16), hc) < 5e-3  # cross logits have std > 1: bf16 q rounding


@pytest.mark.parametrize("guidance_embed", [False, True])
def test_precomputed_conditioning_equals_per_step(guidance_embed):
    """precompute_conditioning (all steps' modulations in one pair of weight passes) is bit-identical to the
    per-call path: a vector's modulation does not depend on how many vectors share the launch."""
    p, sd, inp = tiny_case(guidance_embed, depth=1, singles=1)
    m = HipFluxDiT(p, DEV)
    m.load_state_dict(sd)
    d = {k: v.to(DEV) for k, v in inp.items()}
    img = O.patchify(inp["latent"]).to(DEV)
    ts = [1.0, 0.8, 0.6, 0.4, 0.2]
    kw = dict(img=img, img_ids=d["img_ids"], txt=d["txt"], txt_ids=d["txt_ids"], concepts=d["concepts"],
              concept_ids=d["concept_ids"], concept_vec=d["concept_vec"], y=d["vec"],
              guidance=torch.tensor([2.0], device=DEV))
    ref = [m(timesteps=torch.tensor([t], device=DEV), **kw) for t in ts]
    m.precompute_conditioning(ts, d["vec"], d["concept_vec"], 2.0)
    for i, t in enumerate(ts):
        pred, dd = m(timesteps=torch.tensor([123.0], device=DEV), cond_slot=i, **kw)  # timesteps ignored
        assert torch.equal(pred, ref[i][0])
        for k in DICT_KEYS:
            assert torch.equal(dd[k], ref[i][1][k]), (i, k)


def test_tiny_ablation_branches():
    p, sd, inp = tiny_case(depth=1, singles=0)
    for cross in (True, False):
        for self_ in (True, False):
            jak = {"concept_cross_attention": cross, "concept_self_attention": self_}
            _, d_o = O.dit_forward(sd, p, O.patchify(inp["latent"]), inp["img_ids"], inp["txt"], inp["txt_ids"],
                                   inp["concepts"], inp["concept_ids"], inp["concept_vec"], torch.tensor([0.5]),
                                   inp["vec"], stop_after_multimodal_attentions=True, joint_attention_kwargs=jak)
            _, (_, d) = run_hip(p, sd, inp, 0.5, stop_after_multimodal_attentions=True, joint_attention_kwargs=jak)
            for k in ("output_space_concept_vectors", "cross_attention_concept_vectors"):
                assert maxabs(d[k], d_o[k]) < 2e-2 * max(1.0, d_o[k].abs().max().item()), (jak, k)


def test_argument_errors_match_reference():
    p, sd, inp = tiny_case(guidance_embed=True, depth=1, singles=0)
    m = HipFluxDiT(p, DEV)
    m.load_state_dict(sd)
    d = {k: v.to(DEV) for k, v in inp.items()}
    img = O.patchify(inp["latent"]).to(DEV)
    kw = dict(img_ids=d["img_ids"], txt=d["txt"], txt_ids=d["txt_ids"], concepts=d["concepts"],
              concept_ids=d["concept_ids"], concept_vec=d["concept_vec"], y=d["vec"],
              timesteps=torch.tensor([0.5], device=DEV))
    with pytest.raises(ValueError):  # modified_flux_dit.py:102
        m(img=img, guidance=None, **kw)
    with pytest.raises(ValueError):  # modified_flux_dit.py:94-95
        m(img=img[0], guidance=torch.tensor([1.0]), **kw)
    with pytest.raises(ValueError):
        tiny_params(num_heads=3)  # hidden % heads
    with pytest.raises(RuntimeError):
        m.load_state_dict({"nope": torch.zeros(1)})


def _full_block_model(prefix_kind, T=256, C=4, seed=7, sd_transform=None):
    """A HipFluxDiT holding ONE full-size block (H=3072) with the golden case's weights (``sd_transform``: applied to
    the block's state dict, names without the block prefix)."""
    from oracle.full_block_case import full_block_inputs
    p = FluxParams(depth=1 if prefix_kind == "double" else 0, depth_single_blocks=0 if prefix_kind == "double" else 1)
    case = full_block_inputs(p, T=T, C=C, seed=seed)
    m = HipFluxDiT(p, DEV)
    pref = "double_blocks.0." if prefix_kind == "double" else "single_blocks.0."
    sd = synthetic_state_dict(p, seed=0, prefix=pref)
    if sd_transform is not None:
        sd = {pref + k: v for k, v in sd_transform({k[len(pref):]: v.bfloat16().float() for k, v in sd.items()}).items()}
    m.load_state_dict(sd, strict=False)
    L = 4096
    m._workspace(L, T, C)
    m._rope_table(case["img_ids"], case["txt_ids"], case["concept_ids"], C, T)
    m.X[:C].copy_(case["concepts"][0])
    m.X[C:C + T].copy_(case["txt"][0])
    m.X[C + T:].copy_(case["img"][0])
    m.VEC[0].copy_(case["vec"][0])
    m.VEC[1].copy_(case["concept_vec"][0])
    m._modulations()
    return m, case, (L, T, C)


def test_full_size_double_block_vs_reference_golden(golden):
    """BASELINE.json configs[1] geometry, one block, teacher-forced inputs: the reference's own
    outputs (tests/golden/block_full.npz) are the expected values."""
    g = golden("block_full.npz")
    m, case, (L, T, C) = _full_block_model("double")
    out = {k: [] for k in DICT_KEYS}
    req = HeatmapRequest((0,), 1.0, torch.zeros(C, L, device=DEV), torch.zeros(C, L, device=DEV))
    m._double_block(0, _Geom(1, C, T, L), None, out, True, [req])
    torch.cuda.synchronize()
    rows = torch.from_numpy(g["sample_rows"]).to(DEV)
    CT = C + T
    # attention outputs (pre-proj): SURVEY measured 1.5e-3 bf16-vs-fp32 on these
    assert maxabs(m.ATT[:C], g["concept_attn"][0]) < 4e-3
    assert maxabs(m.ATT[CT:][rows], g["img_attn_rows"]) < 8e-3
    # normalised pre-RoPE q (|q| up to ~0.6 with the synthetic 0.15 scale)
    cq = m.QPRE[:C].view(C, 24, 128).permute(1, 0, 2)
    assert maxabs(cq, g["concept_q"][0]) < 4e-3
    iq = m.QPRE[CT:].view(L, 24, 128).permute(1, 0, 2)[:, rows]
    assert maxabs(iq, g["img_q_rows"]) < 4e-3
    # pre-softmax logits pin the reduction itself (random-init maps are nearly uniform)
    ops.heatmap_logits(m.ATT[CT:], m.ATT32[:C], m.LOGITS[:C])
    ref_lo = torch.from_numpy(g["logits_output_space"][0])
    assert maxabs(m.LOGITS[:C], ref_lo) < 2e-2 * max(1.0, ref_lo.abs().max().item())
    # heat maps: <= 1e-3 max-abs (output space), the north-star tolerance
    hm = req.out_space.view(C, 64, 64)
    print(f"\n[measured] full-size block: output-space heat map max-abs vs reference golden "
          f"{maxabs(hm, g['heatmap_output_space'][0]):.3e}; concept attention rows "
          f"{maxabs(m.ATT32[:C], g['concept_attn'][0]):.3e}")
    assert maxabs(hm, g["heatmap_output_space"][0]) < 1e-3
    assert abs(hm.sum(0) - 1).max().item() < 1e-5
    # cross-attention-space maps are ill-conditioned (logit std ~3 even with the scaled synthetic
    # norms): stated bound 2e-2 max-abs, and the per-patch winner must agree almost everywhere
    cm = req.cross_space.view(C, 64, 64).cpu()
    ref_cm = torch.from_numpy(g["heatmap_cross_attention"][0])
    assert maxabs(cm, ref_cm) < 2e-2
    agree = (cm.argmax(0) == ref_cm.argmax(0)).float().mean().item()
    assert agree > 0.99, agree
    # residual streams after the block (values O(1..5); SURVEY: 2.7e-2 bf16-vs-fp32)
    assert maxabs(m.X[CT:][rows], g["img_out_rows"]) < 6e-2
    assert maxabs(m.X[C:CT][::8], g["txt_out"]) < 6e-2
    assert maxabs(m.X[:C], g["concepts_out"][0]) < 6e-2
    # dict capture shapes = the reference's (modified_double_stream_block.py:185-191)
    assert tuple(out["output_space_image_vectors"][0].shape) == (1, L, 3072)
    assert tuple(out["cross_attention_image_vectors"][0].shape) == (1, 24, L, 128)


PEAKY_MEASURED = {}
# max-abs vs the reference's fp32 block, <= 1.5 x measured on MI355X (profiles/r05_peaky_block_parity.json; the worse of
# the two q / k storage types: half precision / bf16).  Measured, output-space map: iid8 6.3e-3 / 8.8e-3, coldtext
# 1.9e-3 / 3.1e-3; cross-space map 5e-6; the reference's own bf16 run: iid8 2.3e-2 / 8.6e-3 (cross), coldtext 8.4e-3 / 4.7e-3.
PEAKY_BOUNDS = {"iid8": {"heatmap_out": 1.32e-2, "heatmap_cross": 1e-4, "concept_attn_f32": 9.6e-3, "img_attn_rows": 1.8e-2,
                         "img_out_rows": 2.5e-3},
                "coldtext": {"heatmap_out": 4.7e-3, "heatmap_cross": 1e-4, "concept_attn_f32": 4.9e-3,
                             "img_attn_rows": 1.31e-2, "img_out_rows": 1.9e-3}}


@pytest.mark.parametrize("qk_f16", ["captured", "0"])
@pytest.mark.parametrize("case_name", ["iid8", "coldtext"])
def test_full_size_double_block_peaky_logits_vs_reference_golden(golden, case_name, qk_f16):
    """The block_full case at peaky joint-attention logits (round 5; VERDICT r04 weak #2: every other model-level parity
    number lives on logits of std ~1 nat): tests/golden/block_full_peaky.npz is the REFERENCE's block on the same
    weights with the key-norm scales x 8 (std 7.3 nats, row maxima up to 22 nats above the first key tile's) and a
    structured case (every image query sees the whole text tile ~15 nats below its image keys).  Here the fused QK-norm
    epilogue, the pre-scaled q, the half-precision q / k of a captured layer (|k| up to ~200) and the fp32 concept rows
    all see other magnitudes than anywhere else in the suite.  Same gates as the std-1 case above."""
    from oracle.full_block_case import peaky_state_dict
    g = golden("block_full_peaky.npz")
    G = lambda k: g[f"{case_name}_{k}"]   # noqa: E731
    m, case, (L, T, C) = _full_block_model("double", sd_transform=lambda sd: peaky_state_dict(sd, case_name, 3072))
    m.qk_f16 = qk_f16
    out = {k: [] for k in DICT_KEYS}
    req = HeatmapRequest((0,), 1.0, torch.zeros(C, L, device=DEV), torch.zeros(C, L, device=DEV))
    ops.attention_stats(reset=True)
    m._double_block(0, _Geom(1, C, T, L), None, out, True, [req])
    torch.cuda.synchronize()
    stats = ops.attention_stats()
    rows = torch.from_numpy(g["sample_rows"]).to(DEV)
    CT = C + T
    e = {"concept_attn_f32": maxabs(m.ATT32[:C], G("concept_attn")[0]),
         "img_attn_rows": maxabs(m.ATT[CT:][rows], G("img_attn_rows")),
         "concept_q": maxabs(m.QPRE[:C].view(C, 24, 128).permute(1, 0, 2), G("concept_q")[0]),
         "img_q_rows": maxabs(m.QPRE[CT:].view(L, 24, 128).permute(1, 0, 2)[:, rows], G("img_q_rows")),
         "heatmap_out": maxabs(req.out_space.view(C, 64, 64), G("heatmap_output_space")[0]),
         "heatmap_cross": maxabs(req.cross_space.view(C, 64, 64), G("heatmap_cross_attention")[0]),
         "img_out_rows": maxabs(m.X[CT:][rows], G("img_out_rows")),
         "txt_out": maxabs(m.X[C:CT][::8], G("txt_out")), "concepts_out": maxabs(m.X[:C], G("concepts_out")[0]),
         "recomputed_workgroups": stats["recomputed_workgroups"], "rereference_events": stats["rereference_events"],
         "ref_max_abs_img_attn": float(np.abs(G("img_attn_rows")).max())}
    PEAKY_MEASURED[f"{case_name}/qk_f16={qk_f16}"] = e
    print(f"\n[measured] peaky block {case_name} qk_f16={qk_f16}: {e}")
    if len(PEAKY_MEASURED) == 4:
        import json
        import os
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(PEAKY_MEASURED, open("gpurun_out/r05_peaky_block_parity.json", "w"), indent=1)
    # Gates.  A peaky attention row returns (almost) single value vectors, so the 2^-9 relative rounding of v, of P and of
    # the rotated k no longer averages out over 4 352 keys as it does at std 1: NO bf16 path holds 1e-3 on this map.  The
    # yardstick is DESIGN.md section 2's: the REFERENCE's own block run in bf16 on the same numbers (stored beside the
    # fp32 values by oracle/make_goldens.py: output map, cross map, concept rows, image rows, residual rows), and the
    # HIP path must be at least as close to fp32 as that, plus absolute bounds <= 1.5 x measured (PEAKY_BOUNDS).
    y = dict(zip(("heatmap_out", "heatmap_cross", "concept_attn_f32", "img_attn_rows", "img_out_rows"), G("refbf16_err")))
    e["reference_bf16"] = {k: float(v) for k, v in y.items()}
    for k, yref in y.items():
        assert e[k] <= max(1e-3, float(yref)), (k, e[k], float(yref))
        assert e[k] <= PEAKY_BOUNDS[case_name][k], (k, e[k])
    assert e["concept_q"] < 4e-3 and e["img_q_rows"] < 4e-3
    cm = req.cross_space.view(C, 64, 64).cpu()
    ref_cm = torch.from_numpy(G("heatmap_cross_attention")[0])
    assert (cm.argmax(0) == ref_cm.argmax(0)).float().mean().item() > 0.99
    assert e["txt_out"] < 6e-2 and e["concepts_out"] < 6e-2
    assert stats["recomputed_workgroups"] == 0  # the in-place re-reference takes these distributions (ca_attn_stats)


def test_full_size_single_block_vs_reference_golden(golden):
    g = golden("single_full.npz")
    m, case, (L, T, C) = _full_block_model("single")
    m._single_block(0, _Geom(1, C, T, L))
    rows = torch.from_numpy(g["sample_rows"]).to(DEV)
    assert maxabs(m.X[C:][rows], g["out_rows"]) < 6e-2


def test_full_size_double_block_dev_token_counts(golden):
    """flux-dev token counts (T=512, C=8: two passes of the 4-concept logits kernel, 8 concept query
    rows in the attention launch) against the reference's golden for that case."""
    g = golden("block_full_dev.npz")
    m, case, (L, T, C) = _full_block_model("double", T=512, C=8, seed=8)
    req = HeatmapRequest((0,), 1.0, torch.zeros(C, L, device=DEV), torch.zeros(C, L, device=DEV))
    m._double_block(0, _Geom(1, C, T, L), None, None, False, [req])
    rows = torch.from_numpy(g["sample_rows"]).to(DEV)
    CT = C + T
    assert maxabs(m.ATT32[:C], g["concept_attn"][0]) < 1e-3   # fp32 copy: no output rounding
    assert maxabs(m.ATT[CT:][rows], g["img_attn_rows"]) < 8e-3
    hm = req.out_space.view(C, 64, 64)
    assert maxabs(hm, g["heatmap_output_space"][0]) < 1e-3
    assert abs(hm.sum(0) - 1).max().item() < 1e-5
    cm = req.cross_space.view(C, 64, 64).cpu()
    ref_cm = torch.from_numpy(g["heatmap_cross_attention"][0])
    assert maxabs(cm, ref_cm) < 2e-2 and (cm.argmax(0) == ref_cm.argmax(0)).float().mean().item() > 0.985
    assert maxabs(m.X[CT:][rows], g["img_out_rows"]) < 6e-2
    assert maxabs(m.X[C:CT][::16], g["txt_out"]) < 6e-2
    assert maxabs(m.X[:C], g["concepts_out"][0]) < 6e-2


def test_tiny_model_fp8_mode_tracks_bf16():
    """fp8 mode (e4m3 projections): no reference exists, so the check is against the bf16 path of the same
    kernels on the tiny model -- same shapes and keys, heat maps within 3e-2 (e4m3 has a 2^-4 relative
    step), pred within 10 % rms; and switching back restores the bf16 result bit for bit."""
    p = tiny_params()
    m = HipFluxDiT(p, DEV)
    m.weights.init_synthetic(3, on_device=False)
    inp = synthetic_inputs(p, 256, 256, 8, 3, seed=5, dtype=torch.bfloat16)
    from conceptattention_amd import sampling
    x = {k: v.to(DEV) for k, v in inp.items()}
    kw = dict(img=sampling.patchify(x["latent"]), img_ids=x["img_ids"], txt=x["txt"], txt_ids=x["txt_ids"],
              concepts=x["concepts"], concept_ids=x["concept_ids"], concept_vec=x["concept_vec"], y=x["vec"],
              timesteps=torch.full((1,), 0.7, device=DEV), guidance=torch.zeros(1, device=DEV))
    pred_a, d_a = m(**kw)
    m.set_precision("fp8")
    pred_b, d_b = m(**kw)
    m.set_precision("bf16")
    pred_c, d_c = m(**kw)
    assert torch.equal(pred_a, pred_c) and all(torch.equal(d_a[k], d_c[k]) for k in d_a)
    assert d_b.keys() == d_a.keys() and all(d_b[k].shape == d_a[k].shape for k in d_a)
    rel = ((pred_b.float() - pred_a.float()).norm() / pred_a.float().norm()).item()
    assert 0 < rel < 0.10, rel
    from conceptattention_amd.heatmaps import compute_heatmaps_from_vectors
    layers = list(range(p.depth))
    ha = compute_heatmaps_from_vectors(d_a["output_space_image_vectors"][None], d_a["output_space_concept_vectors"][None],
                                       layers, [0])
    hb = compute_heatmaps_from_vectors(d_b["output_space_image_vectors"][None], d_b["output_space_concept_vectors"][None],
                                       layers, [0])
    assert (ha - hb).abs().max().item() < 3e-2


def test_full_size_block_image_stream_ignores_the_concepts():
    """Size-independent property of the reference's block at the full geometry: the concept stream only READS
    the image keys/values (modified_double_stream_block.py:121-123,162-166), so the image and text rows
    after the block must not change by one bit when the concept tokens change or their count does; and
    permuting the concepts permutes their own outputs."""
    outs = {}
    _, base, _ = _full_block_model("double", C=4, seed=7)     # one set of image / text / vec values for all runs
    for tag, C, pick in (("c4", 4, [0, 1, 2, 3]), ("c4_perm", 4, [2, 0, 3, 1]), ("c1", 1, [0])):
        m, _, (L, T, C_) = _full_block_model("double", C=C, seed=7)
        m.X[:C].copy_(base["concepts"][0][pick])
        m.X[C:C + T].copy_(base["txt"][0])
        m.X[C + T:].copy_(base["img"][0])
        m.VEC[0].copy_(base["vec"][0])
        m.VEC[1].copy_(base["concept_vec"][0])
        m._modulations()
        m._double_block(0, _Geom(1, C, T, L), None, None, False, None)
        torch.cuda.synchronize()
        outs[tag] = (m.X[C:C + T].clone(), m.X[C + T:].clone(), m.X[:C].clone(), m.ATT32[:C].clone())
    for tag in ("c4_perm", "c1"):
        assert torch.equal(outs[tag][0], outs["c4"][0]), f"text rows changed ({tag})"
        assert torch.equal(outs[tag][1], outs["c4"][1]), f"image rows changed ({tag})"
    # concept i of the permuted run is concept perm[i] of the first run (summation order over the 4 concept
    # keys differs, so equality is to fp32 / bf16 rounding, not bitwise)
    perm = [2, 0, 3, 1]
    assert maxabs(outs["c4_perm"][3], outs["c4"][3][perm]) < 1e-4
    assert maxabs(outs["c4_perm"][2], outs["c4"][2][perm]) < 4e-2


def test_forward_is_hip_graph_capturable():
    """Every entry point is stream-ordered and allocates nothing, so a whole forward can be captured into a
    HIP graph and replayed; the replay reproduces the eager result bit for bit."""
    p, sd, inp = tiny_case()
    m = HipFluxDiT(p, DEV)
    m.load_state_dict(sd)
    d = {k: v.to(DEV) for k, v in inp.items()}
    kw = dict(img=O.patchify(inp["latent"]).to(DEV), img_ids=d["img_ids"], txt=d["txt"], txt_ids=d["txt_ids"],
              concepts=d["concepts"], concept_ids=d["concept_ids"], concept_vec=d["concept_vec"], y=d["vec"],
              timesteps=torch.tensor([0.5], device=DEV), return_vectors=False)
    eager = m(**kw)[0].clone()          # also warms up (workspace, LDS opt-in attributes)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m(**kw)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(**kw)[0]
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
