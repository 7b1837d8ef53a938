"""The CPU oracle (oracle/flux_oracle.py) against the vectors the REFERENCE produced
(tests/golden/*.npz, written by oracle/make_goldens.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conceptattention_amd.params import FluxParams, tiny_params
from conceptattention_amd.weights import synthetic_inputs, synthetic_state_dict
from oracle import flux_oracle as O

TOL = 2e-5  # fp32 vs fp32, different accumulation order only


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _maxabs(a, b):
    return (torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max().item()


@pytest.mark.parametrize("name,guidance_embed", [("tiny_schnell.npz", False), ("tiny_dev.npz", True)])
def test_tiny_model_forward(golden, name, guidance_embed):
    g = golden(name)
    p = tiny_params(guidance_embed=guidance_embed)
    sd = synthetic_state_dict(p, seed=1)
    inp = synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=3, seed=2)
    img = O.patchify(inp["latent"])
    t, gv = _t(g["timestep"]), _t(g["guidance"])
    pred, d = O.dit_forward(sd, p, img, inp["img_ids"], inp["txt"], inp["txt_ids"], inp["concepts"],
                            inp["concept_ids"], inp["concept_vec"], t, inp["vec"], gv)
    assert _maxabs(pred, g["pred"]) < TOL
    for k in O.DICT_KEYS:
        assert d[k].shape == g[k].shape, k
        assert _maxabs(d[k], g[k]) < TOL, k
    none_pred, d2 = O.dit_forward(sd, p, img, inp["img_ids"], inp["txt"], inp["txt_ids"], inp["concepts"],
                                  inp["concept_ids"], inp["concept_vec"], t, inp["concept_vec"], gv,
                                  stop_after_multimodal_attentions=True)
    assert none_pred is None
    for k in O.DICT_KEYS:
        assert _maxabs(d2[k], g["stop_" + k]) < TOL, k


def test_guidance_required_for_dev():
    p = tiny_params(guidance_embed=True, depth=1, depth_single_blocks=0)
    sd = synthetic_state_dict(p, seed=1)
    inp = synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=2, seed=2)
    with pytest.raises(ValueError):
        O.dit_forward(sd, p, O.patchify(inp["latent"]), inp["img_ids"], inp["txt"], inp["txt_ids"],
                      inp["concepts"], inp["concept_ids"], inp["concept_vec"], torch.tensor([0.5]),
                      inp["vec"], None)


def test_ablation_branches(golden):
    g = golden("tiny_ablation.npz")
    p = tiny_params(depth=1, depth_single_blocks=0)
    sd = synthetic_state_dict(p, seed=1)
    inp = synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=3, seed=2)
    img = O.patchify(inp["latent"])
    for cross in (True, False):
        for self_ in (True, False):
            _, d = O.dit_forward(sd, p, img, inp["img_ids"], inp["txt"], inp["txt_ids"], inp["concepts"],
                                 inp["concept_ids"], inp["concept_vec"], torch.tensor([0.5]), inp["vec"],
                                 stop_after_multimodal_attentions=True,
                                 joint_attention_kwargs={"concept_cross_attention": cross,
                                                         "concept_self_attention": self_})
            tag = f"cross{int(cross)}_self{int(self_)}_"
            for k in ("output_space_concept_vectors", "cross_attention_concept_vectors"):
                assert _maxabs(d[k], g[tag + k]) < TOL, tag + k


def test_heatmap_known_answers(golden):
    g = golden("heatmap_kat.npz")
    iv, cv = _t(g["iv"]).float(), _t(g["cv"]).float()
    a = O.compute_heatmaps(iv, cv, [1, 3, 4], [0, 2])
    assert a.shape == (1, 4, 64, 64) and _maxabs(a, g["out_a"]) < 1e-6
    b = O.compute_heatmaps(iv, cv, [2], [1], normalize_concepts=True)
    assert _maxabs(b, g["out_b"]) < 1e-6
    c = O.compute_heatmaps(_t(g["iv6"]).float(), _t(g["cv6"]).float(), [0, 1], [0, 1])
    assert c.shape == (1, 5, 64, 64) and _maxabs(c, g["out_c"]) < 1e-6
    assert abs(a.sum(1) - 1).max() < 1e-5  # softmax over the concept axis
    with pytest.raises(NotImplementedError):
        O.compute_heatmaps(iv, cv, [0], [0], softmax=False)


def test_sampler_pieces(golden):
    g = golden("sampler.npz")
    assert np.allclose(O.get_schedule(4, 4096, shift=False), g["schedule_schnell_4"], atol=1e-7)
    assert np.allclose(O.get_schedule(50, 4096, shift=True), g["schedule_dev_50_4096"], atol=1e-6)
    assert np.allclose(O.get_schedule(28, 1024, shift=True), g["schedule_dev_28_1024"], atol=1e-6)
    x = _t(g["patchify_in"])
    assert np.array_equal(O.patchify(x).numpy(), g["patchify_out"])  # exact: index mapping only
    assert np.array_equal(O.unpack(_t(g["patchify_out"]), 64, 96).numpy(), g["unpack_out"])
    ids = O.make_img_ids(4, 6)
    assert ids[0, 2 * 6 + 5].tolist() == [0.0, 2.0, 5.0]  # token index = row*w + col


def test_tiny_denoise_and_heatmaps(golden):
    g = golden("sampler.npz")
    p = tiny_params(depth=2, depth_single_blocks=1)
    sd = synthetic_state_dict(p, seed=3)
    inp = synthetic_inputs(p, 1024, 1024, n_txt=8, n_concepts=3, seed=4)
    img = O.patchify(inp["latent"])
    ts = O.get_schedule(2, img.shape[1], shift=False)
    out, d = O.denoise(sd, p, img, inp["img_ids"], inp["txt"], inp["txt_ids"], inp["vec"], ts, 0.0,
                       inp["concepts"], inp["concept_ids"], inp["concept_vec"])
    assert _maxabs(out[0, ::64], g["denoise_img_rows"]) < 5e-5
    hm = O.compute_heatmaps(d["output_space_image_vectors"], d["output_space_concept_vectors"], [0, 1], [0, 1])
    assert _maxabs(hm, g["denoise_heatmaps"]) < 1e-5
    cm = O.compute_heatmaps(d["cross_attention_image_vectors"], d["cross_attention_concept_vectors"], [1], [0, 1])
    assert _maxabs(cm, g["denoise_cross_maps"]) < 1e-4


def test_full_size_blocks(golden):
    """One full-size double block + single block; ~10 s on 8 cores."""
    from oracle.full_block_case import full_block_inputs
    g = golden("block_full.npz")
    p = FluxParams()
    case = full_block_inputs(p)
    sd = {k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=0, prefix="double_blocks.0.").items()}
    rope_ti = O.rope_cos_sin(torch.cat((case["txt_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    rope_ci = O.rope_cos_sin(torch.cat((case["concept_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    img, txt, con, d = O.double_block(sd, "double_blocks.0.", p.num_heads, case["img"], case["txt"], case["vec"],
                                      rope_ti, case["concepts"], case["concept_vec"], rope_ci)
    rows = _t(g["sample_rows"])
    assert _maxabs(d["output_space_concept_vectors"], g["concept_attn"]) < 1e-5
    assert _maxabs(d["cross_attention_concept_vectors"], g["concept_q"]) < 1e-5
    assert _maxabs(d["output_space_image_vectors"][0, rows], g["img_attn_rows"]) < 1e-5
    assert _maxabs(d["cross_attention_image_vectors"][0, :, rows], g["img_q_rows"]) < 1e-5
    assert _maxabs(img[0, rows], g["img_out_rows"]) < 2e-4
    assert _maxabs(txt[0, ::8], g["txt_out"]) < 2e-4
    assert _maxabs(con, g["concepts_out"]) < 2e-4
    st = {k: v[None, None] for k, v in d.items()}
    lo = O.heatmap_logits(st["output_space_image_vectors"], st["output_space_concept_vectors"])[0, 0]
    assert _maxabs(lo, g["logits_output_space"]) < 1e-4
    hm = O.compute_heatmaps(st["output_space_image_vectors"], st["output_space_concept_vectors"], [0], [0])
    assert _maxabs(hm, g["heatmap_output_space"]) < 1e-5
    lc = O.heatmap_logits(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"])[0, 0]
    assert _maxabs(lc, g["logits_cross_attention"]) < 1e-4
    cm = O.compute_heatmaps(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"], [0], [0])
    assert _maxabs(cm, g["heatmap_cross_attention"]) < 1e-4
    # single block
    gs = golden("single_full.npz")
    sds = {k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=0, prefix="single_blocks.0.").items()}
    y = O.single_block(sds, "single_blocks.0.", p.num_heads, torch.cat((case["txt"], case["img"]), 1),
                       case["vec"], rope_ti)
    assert _maxabs(y[0, _t(gs["sample_rows"])], gs["out_rows"]) < 2e-4


def test_tiny_model_peaky_logits(golden):
    """The whole tiny model (2 double + 2 single blocks, final layer) with every key-norm scale x 8, made by the
    reference's ModifiedFluxDiT (oracle/make_goldens.py tinypeaky): the oracle at peaky logits through every block type."""
    from oracle.make_goldens import tiny_peaky_state_dict
    g = golden("tiny_peaky.npz")
    p = tiny_params()
    sd = tiny_peaky_state_dict(synthetic_state_dict(p, seed=1))
    inp = {k: (v.bfloat16().float() if v.is_floating_point() else v)
           for k, v in synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=3, seed=2).items()}
    pred, d = O.dit_forward(sd, p, O.patchify(inp["latent"]), inp["img_ids"], inp["txt"], inp["txt_ids"], inp["concepts"],
                            inp["concept_ids"], inp["concept_vec"], _t(g["timestep"]), inp["vec"], torch.tensor([0.0]))
    assert _maxabs(pred, g["pred"]) < 5e-5
    for k in O.DICT_KEYS:
        assert d[k].shape == g[k].shape and _maxabs(d[k], g[k]) < 5e-5, k


@pytest.mark.parametrize("case_name", ["iid8", "coldtext"])
def test_full_size_block_peaky_logits(golden, case_name):
    """block_full's case with peaky joint-attention logits (oracle/full_block_case.PEAKY_CASES: std ~7 nats; a text tile
    ~15 nats below the image keys), made by the reference's own block (oracle/make_goldens.py peaky)."""
    from oracle.full_block_case import full_block_inputs, peaky_state_dict
    g = golden("block_full_peaky.npz")
    p = FluxParams()
    case = full_block_inputs(p)
    base = {k[len("double_blocks.0."):]: v.bfloat16().float()
            for k, v in synthetic_state_dict(p, seed=0, prefix="double_blocks.0.").items()}
    sd = {"double_blocks.0." + k: v for k, v in peaky_state_dict(base, case_name, p.hidden_size).items()}
    rope_ti = O.rope_cos_sin(torch.cat((case["txt_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    rope_ci = O.rope_cos_sin(torch.cat((case["concept_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    img, txt, con, d = O.double_block(sd, "double_blocks.0.", p.num_heads, case["img"], case["txt"], case["vec"],
                                      rope_ti, case["concepts"], case["concept_vec"], rope_ci)
    rows = _t(g["sample_rows"])
    G = lambda k: g[f"{case_name}_{k}"]   # noqa: E731
    stats = G("joint_logit_stats")
    assert stats[0] > 4.0 and stats[1] > 20.0          # the case IS peaky: std, late-maximum gap in nats
    assert _maxabs(d["output_space_concept_vectors"], G("concept_attn")) < 2e-5
    assert _maxabs(d["cross_attention_concept_vectors"], G("concept_q")) < 1e-5
    assert _maxabs(d["output_space_image_vectors"][0, rows], G("img_attn_rows")) < 2e-5
    assert _maxabs(d["cross_attention_image_vectors"][0, :, rows], G("img_q_rows")) < 1e-5
    assert _maxabs(img[0, rows], G("img_out_rows")) < 2e-4
    assert _maxabs(txt[0, ::8], G("txt_out")) < 2e-4
    assert _maxabs(con, G("concepts_out")) < 2e-4
    st = {k: v[None, None] for k, v in d.items()}
    lo = O.heatmap_logits(st["output_space_image_vectors"], st["output_space_concept_vectors"])[0, 0]
    assert _maxabs(lo, G("logits_output_space")) < 2e-4
    hm = O.compute_heatmaps(st["output_space_image_vectors"], st["output_space_concept_vectors"], [0], [0])
    assert _maxabs(hm, G("heatmap_output_space")) < 2e-5
    lc = O.heatmap_logits(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"])[0, 0]
    assert _maxabs(lc, G("logits_cross_attention")) < 1e-4
    cm = O.compute_heatmaps(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"], [0], [0])
    assert _maxabs(cm, G("heatmap_cross_attention")) < 1e-4


def test_full_size_block_dev_token_counts(golden):
    """T=512 text tokens, C=8 concepts (BASELINE.json configs[2] geometry), one full-size double block."""
    from oracle.full_block_case import full_block_inputs
    g = golden("block_full_dev.npz")
    p = FluxParams(guidance_embed=True)
    case = full_block_inputs(p, T=512, C=8, seed=8)
    sd = {k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=0, prefix="double_blocks.0.").items()}
    rope_ti = O.rope_cos_sin(torch.cat((case["txt_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    rope_ci = O.rope_cos_sin(torch.cat((case["concept_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    img, txt, con, d = O.double_block(sd, "double_blocks.0.", p.num_heads, case["img"], case["txt"], case["vec"],
                                      rope_ti, case["concepts"], case["concept_vec"], rope_ci)
    rows = _t(g["sample_rows"])
    assert _maxabs(d["output_space_concept_vectors"], g["concept_attn"]) < 1e-5
    assert _maxabs(img[0, rows], g["img_out_rows"]) < 2e-4
    st = {k: v[None, None] for k, v in d.items()}
    hm = O.compute_heatmaps(st["output_space_image_vectors"], st["output_space_concept_vectors"], [0], [0])
    assert hm.shape == (1, 8, 64, 64) and _maxabs(hm, g["heatmap_output_space"]) < 1e-5
