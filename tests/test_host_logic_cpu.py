"""Host-side logic of the product package on CPU (no kernels run): sampler arithmetic against the
reference's golden vectors, state-dict layout, argument checks, colour mapping."""
import numpy as np
import pytest
import torch

from conceptattention_amd import sampling
from conceptattention_amd.params import FluxParams, configs, tiny_params
from conceptattention_amd.weights import state_dict_spec, synthetic_inputs, synthetic_state_dict


def test_schedule_patchify_unpack_match_reference(golden):
    g = golden("sampler.npz")
    assert np.allclose(sampling.get_schedule(4, 4096, shift=False), g["schedule_schnell_4"], atol=1e-7)
    assert np.allclose(sampling.get_schedule(50, 4096, shift=True), g["schedule_dev_50_4096"], atol=1e-6)
    assert np.allclose(sampling.get_schedule(28, 1024, shift=True), g["schedule_dev_28_1024"], atol=1e-6)
    x = torch.from_numpy(g["patchify_in"])
    assert np.array_equal(sampling.patchify(x).numpy(), g["patchify_out"])
    assert np.array_equal(sampling.unpack(torch.from_numpy(g["patchify_out"]), 64, 96).numpy(), g["unpack_out"])


def test_image_token_index_is_row_major():
    ids = sampling.make_img_ids(64, 64)
    assert ids.shape == (1, 4096, 3)
    for tok in (0, 63, 64, 4095, 1234):
        assert ids[0, tok].tolist() == [0.0, float(tok // 64), float(tok % 64)]


def test_state_dict_layout_matches_flux():
    """Key names / shapes / parameter count of the BFL checkpoints (11.90 B params)."""
    spec = dict(state_dict_spec(configs["flux-schnell"]))
    assert sum(int(np.prod(s)) for s in spec.values()) == 11_891_178_560
    assert spec["double_blocks.18.txt_attn.qkv.weight"] == (9216, 3072)
    assert spec["single_blocks.37.linear1.weight"] == (21504, 3072)
    assert spec["single_blocks.0.linear2.weight"] == (3072, 15360)
    assert spec["final_layer.adaLN_modulation.1.weight"] == (6144, 3072)
    assert "guidance_in.in_layer.weight" not in spec
    dev = dict(state_dict_spec(configs["flux-dev"]))
    assert dev["guidance_in.in_layer.weight"] == (3072, 256)
    assert sum(int(np.prod(s)) for s in dev.values()) == 11_901_408_320  # 11.90 B


def test_synthetic_tensors_are_order_independent_and_seeded():
    p = tiny_params()
    a = synthetic_state_dict(p, seed=1)
    b = synthetic_state_dict(p, seed=1, prefix="double_blocks.1.")
    for k, v in b.items():
        assert torch.equal(a[k], v)
    c = synthetic_state_dict(p, seed=2, prefix="double_blocks.1.")
    assert not torch.equal(c["double_blocks.1.img_mlp.0.weight"], a["double_blocks.1.img_mlp.0.weight"])
    inp = synthetic_inputs(p, 256, 256, 8, 3, seed=0)
    assert inp["latent"].shape == (1, 16, 32, 32) and inp["concept_vec"].abs().max() == 0
    assert inp["concept_ids"].shape == (1, 3, 3) and inp["concept_ids"].abs().max() == 0


def test_params_validation_matches_reference_errors():
    with pytest.raises(ValueError):
        FluxParams(hidden_size=3072, num_heads=23)
    with pytest.raises(ValueError):
        FluxParams(axes_dim=(16, 56, 55))
    assert configs["flux-dev"].guidance_embed and not configs["flux-schnell"].guidance_embed


def test_heatmap_norm_selection_follows_the_reference_branch_order():
    """concept_attention_pipeline.py:64-71: softmax if `softmax` or attention_norm == "softmax", else entmax15 /
    sparsemax, else ValueError (raised before anything touches the GPU)."""
    from conceptattention_amd import _lib
    from conceptattention_amd.heatmaps import compute_heatmaps_from_vectors, linear_normalization, resolve_norm
    assert resolve_norm(True, "sparsemax") == _lib.NORM_SOFTMAX      # the reference's defaults
    assert resolve_norm(False, "softmax") == _lib.NORM_SOFTMAX
    assert resolve_norm(False, "sparsemax") == _lib.NORM_SPARSEMAX
    assert resolve_norm(False, "entmax15") == _lib.NORM_ENTMAX15
    iv, cv = torch.zeros(1, 1, 1, 4, 8), torch.zeros(1, 1, 1, 2, 8)
    with pytest.raises(ValueError):
        compute_heatmaps_from_vectors(iv, cv, [0], [0], softmax=False, attention_norm="nope")
    x = torch.tensor([[1.0, 3.0], [2.0, 2.0]])
    n = linear_normalization(x, dim=0)
    assert torch.allclose(n.sum(0), torch.tensor([1.0, 1.0]))


def test_colorize_uses_global_minmax():
    from conceptattention_amd.pipeline import colorize_heatmaps
    hm = np.stack([np.full((4, 4), 0.2, np.float32), np.full((4, 4), 0.4, np.float32)])
    hm[1, 0, 0] = 0.6
    imgs = colorize_heatmaps(hm, "plasma")
    assert len(imgs) == 2 and imgs[0].size == (4, 4) and imgs[0].mode == "RGB"
    import matplotlib.pyplot as plt
    lo = (np.array(plt.get_cmap("plasma")(0.0)[:3]) * 255).astype(np.uint8)
    hi = (np.array(plt.get_cmap("plasma")(1.0)[:3]) * 255).astype(np.uint8)
    assert tuple(np.asarray(imgs[0])[1, 1]) == tuple(lo)      # 0.2 is the global minimum
    assert tuple(np.asarray(imgs[1])[0, 0]) == tuple(hi)      # 0.6 is the global maximum


def test_batched_prepare_and_id_tags():
    """prepare / embed_concepts contracts for a batch of work items (flux/sampling.py:31-65, utils.py:6-33 per item)
    and the content tags HipFluxDiT keys its RoPE-table cache on."""
    lat = torch.arange(2 * 16 * 4 * 6, dtype=torch.float32).reshape(2, 16, 4, 6)
    txt, vec = torch.zeros(2, 5, 32), torch.zeros(2, 8)
    inp = sampling.prepare_from_embeddings(lat, txt, vec)
    assert inp["img"].shape == (2, 6, 64) and inp["img_ids"].shape == (2, 6, 3) and inp["txt_ids"].shape == (2, 5, 3)
    for b in range(2):   # every item is patchified and indexed like a batch of one
        one = sampling.prepare_from_embeddings(lat[b:b + 1], txt[b:b + 1], vec[b:b + 1])
        assert torch.equal(inp["img"][b], one["img"][0]) and torch.equal(inp["img_ids"][b], one["img_ids"][0])
    with pytest.raises(ValueError):
        sampling.prepare_from_embeddings(lat, txt[:1], vec)
    con, con_ids, con_vec = sampling.concept_inputs(torch.ones(2, 3, 32), vec)
    assert con_ids.shape == (2, 3, 3) and con_ids.abs().max() == 0 and con_vec.abs().max() == 0
    a, b = sampling.make_img_ids(4, 2), sampling.make_img_ids(2, 4)
    assert a._ca_ids_tag[0] == ("img", 4, 2, 1) and b._ca_ids_tag[0] == ("img", 2, 4, 1)   # same shape, different content
    assert sampling.zero_ids(3, batch=2)._ca_ids_tag[0] == ("zero", 3, 2)
    v = a._version
    a[0, 0, 1] = 7.0                                      # an in-place write moves the version: the tag is void
    assert a._version != v and a._ca_ids_tag[1] == v


def test_row_geometry_of_a_batched_forward():
    from conceptattention_amd.flux_dit import _Geom
    g = _Geom(B=5, C=4, T=256, L=4096)
    assert (g.oT, g.oI, g.n) == (20, 1300, 21780)
    # the single blocks' [text | image] rows of five items are exactly 85 row tiles of 256 (1020 tiles at N = 3072)
    assert (g.n - g.oT) % 256 == 0 and (g.n - g.oT) // 256 == 85


def test_every_attribute_the_model_reads_is_assigned_somewhere():
    import os
    """HipFluxDiT cannot be constructed without a GPU, so a deleted `self.x = ...` line in its constructor would only
    show on the GPU box (it did once, round 4).  Static check: every `self.<name>` the class reads is assigned, defined
    as a method / property, or a workspace buffer."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conceptattention_amd",
                            "flux_dit.py")).read()
    cls = src[src.index("class HipFluxDiT"):]
    used = set(re.findall(r"self\.([A-Za-z_][A-Za-z_0-9]*)\b", cls))
    assigned = (set(re.findall(r"self\.([A-Za-z_][A-Za-z_0-9]*)\s*=[^=]", cls)) |
                set(re.findall(r"def ([A-Za-z_][A-Za-z_0-9]*)\(", cls)) |
                set(re.findall(r"^    ([A-Za-z_][A-Za-z_0-9]*) = ", cls, re.M)) |
                set(re.findall(r"^\s+([A-Z][A-Z0-9_]*)=torch\.", cls, re.M)) |          # _alloc_workspace's buffers
                set(re.findall(r'"([A-Z][A-Z0-9_]*)":\s*lambda', cls)) |                 # _LAZY_BUFFERS
                set(re.findall(r"([A-Z][A-Z0-9_]*)=torch\.zeros", cls)))
    missing = sorted(used - assigned - {"__dict__"})
    assert missing == [], missing


def _pp_tile_of(bid, main_total, ntiles_main, mt_main, nt, group_m, xcd_interleave, nthin, mt):
    """Python model of ca_gemm_pp_tile's tile order (conceptattention_amd/csrc/ca_gemm.hip: bid -> (problem, row tile,
    column tile)); kept beside the kernel's arithmetic line for line."""
    if bid < main_total:
        xcd = bid & 7
        if xcd_interleave and bid < (main_total & ~255):
            lid = (bid & ~255) + 32 * xcd + ((bid >> 3) & 31)
        else:
            base = (main_total & ~255) if xcd_interleave else 0
            tot, b = main_total - base, bid - base
            q8, r8 = tot >> 3, tot & 7
            lid = base + (xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3)
        prob = 1 if lid >= ntiles_main[0] else 0
        if prob:
            lid -= ntiles_main[0]
        MT, NT = mt_main[prob], nt[prob]
        grp = lid // (group_m * NT)
        first_m = grp * group_m
        gm = min(group_m, MT - first_m)
        in_grp = lid - grp * group_m * NT
        return prob, first_m + in_grp % gm, in_grp // gm
    u = bid - main_total
    prob = 1 if u >= nthin[0] else 0
    if prob:
        u -= nthin[0]
    return prob, mt[prob] - 1, u


def test_gemm_tile_order_is_a_bijection_for_any_tile_count():
    """the XCD-interleaved order assumes runs of 256 tiles; the mapping must stay a bijection onto the
    (problem, row tile, column tile) set for every main_total (multiples of 256 or not), group height, column count,
    with and without thin tiles, interleave on or off -- whatever grid walks it."""
    import itertools
    for (mt0, mt1), NT, gm, xil, thin in itertools.product(
            [(17, 2), (80, 6), (5, 0), (1, 1), (33, 7), (86, 0)], [1, 12, 36, 48, 84], [1, 3, 4, 6, 8], [0, 1], [0, 1]):
        mt = [mt0, max(mt1, 1)]
        has1 = mt1 > 0
        nthin = [0, NT if (thin and has1) else 0]
        mt_main = [mt[0], (mt[1] - (1 if nthin[1] else 0)) if has1 else 1]
        ntiles_main = [mt_main[0] * NT, mt_main[1] * NT if has1 else 0]
        main_total = sum(ntiles_main)
        total = main_total + sum(nthin)
        seen = {_pp_tile_of(b, main_total, ntiles_main, mt_main, [NT, NT], gm, xil, nthin, mt) for b in range(total)}
        want = {(0, m, n) for m in range(mt_main[0]) for n in range(NT)}
        if has1:
            want |= {(1, m, n) for m in range(mt_main[1]) for n in range(NT)}
            if nthin[1]:
                want |= {(1, mt[1] - 1, n) for n in range(NT)}
        assert seen == want, (mt0, mt1, NT, gm, xil, thin, len(seen), len(want))
