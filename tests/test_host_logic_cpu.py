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
