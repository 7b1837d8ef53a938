"""Round-2 behaviour on the GPU: the real-weights seams (safetensors loader, injectable encoders / autoencoder),
sparsemax / 1.5-entmax heat maps, RoPE table caching by content, fp8 shared state across streams."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conceptattention_amd import ConceptAttentionFluxPipeline, _lib, ops  # noqa: E402
from conceptattention_amd.flux_dit import HipFluxDiT  # noqa: E402
from conceptattention_amd.heatmaps import compute_heatmaps_from_vectors  # noqa: E402
from conceptattention_amd.image_generator import load_flow_model  # noqa: E402
from conceptattention_amd.params import tiny_params  # noqa: E402
from conceptattention_amd.weights import synthetic_inputs, synthetic_state_dict  # noqa: E402
from oracle import flux_oracle as O  # noqa: E402
from oracle import sparse_norms as SN  # noqa: E402

DEV = "cuda:0"


# ------------------------------------------------------------------ real-weights seams
def _write_bfl_checkpoint(tmp_path, p, seed=1, drop=(), extra=()):
    """A tiny-geometry checkpoint under the BFL state-dict names, as flux1-*.safetensors files hold them
    (bf16 tensors; concept_attention/image_generator.py:36-44 loads such a file with load_sft + strict=False)."""
    from safetensors.torch import save_file
    sd = {k: v.bfloat16().contiguous() for k, v in synthetic_state_dict(p, seed=seed).items() if k not in drop}
    for k in extra:
        sd[k] = torch.zeros(3, dtype=torch.bfloat16)
    path = os.path.join(tmp_path, "flux1-tiny.safetensors")
    save_file(sd, path)
    return path, sd


def test_safetensors_checkpoint_loads_through_env_and_path(tmp_path, monkeypatch):
    p = tiny_params()
    path, sd = _write_bfl_checkpoint(str(tmp_path), p)
    # (1) explicit path
    m1 = load_flow_model("flux-schnell", DEV, weights=path, params=p)
    # (2) FLUX_SCHNELL env var, as flux/util.py:33 (the reference's configs read the same variable)
    monkeypatch.setenv("FLUX_SCHNELL", path)
    m2 = load_flow_model("flux-schnell", DEV, weights="synthetic", params=p)
    for k, v in sd.items():
        assert torch.equal(m1.state_dict()[k].cpu(), v), k
        assert torch.equal(m2.state_dict()[k].cpu(), v), k
    # (3) the pipeline constructor takes the same path and the loaded weights drive the forward: identical maps
    # to a pipeline built from the state dict itself
    monkeypatch.delenv("FLUX_SCHNELL")
    a = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights=path, params=p, n_text_tokens=8)
    b = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights={k: v.float() for k, v in sd.items()},
                                     params=p, n_text_tokens=8)
    kw = dict(prompt="a dog", concepts=["dog", "tree"], width=256, height=256, layer_indices=[0, 1],
              num_inference_steps=2, return_pil_heatmaps=False, seed=5)
    ra, rb = a.generate_image(**kw), b.generate_image(**kw)
    assert np.array_equal(ra.concept_heatmaps, rb.concept_heatmaps)


def test_load_state_dict_strict_false_reports_missing_and_unexpected(tmp_path):
    p = tiny_params()
    drop = ("double_blocks.0.img_attn.proj.bias",)
    path, sd = _write_bfl_checkpoint(str(tmp_path), p, drop=drop, extra=("some.other.tensor",))
    from safetensors.torch import load_file
    m = HipFluxDiT(p, DEV)
    before = m.state_dict()[drop[0]].clone()
    missing, unexpected = m.load_state_dict(load_file(path), strict=False, assign=True)
    assert list(missing) == list(drop) and list(unexpected) == ["some.other.tensor"]
    assert torch.equal(m.state_dict()[drop[0]], before)  # a missing key leaves the tensor untouched
    with pytest.raises(RuntimeError):
        m.load_state_dict(load_file(path), strict=True)
    bad = dict(load_file(path))
    bad["img_in.weight"] = torch.zeros(5, 5)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=False)
    # load_flow_model's file branch is strict=False as the reference (image_generator.py:44): it must not raise
    load_flow_model("flux-schnell", DEV, weights=path, params=p)


class _StubEncoder:
    """Stands where HFEmbedder does (flux/modules/conditioner.py:6-37): t5(text) -> (1,T,4096), clip(text) -> (1,768)."""

    def __init__(self, T, ctx, vec):
        self.T, self.ctx, self.vec, self.calls = T, ctx, vec, []

    def _v(self, text, n):
        g = torch.Generator().manual_seed(sum(text.encode()) + n)
        return torch.randn(n, generator=g)

    def t5(self, text):
        self.calls.append(("t5", text))
        return self._v(text, self.T * self.ctx).view(1, self.T, self.ctx).to(DEV, torch.bfloat16)

    def clip(self, text):
        self.calls.append(("clip", text))
        return self._v(text, self.vec).view(1, self.vec).to(DEV, torch.bfloat16)


class _StubAutoEncoder:
    """Stands where flux's AutoEncoder does (flux/modules/autoencoder.py): encode (1,3,H,W) -> (1,16,H/8,W/8),
    decode back; plain average pooling / nearest upsampling so that the plumbing is checkable."""

    def encode(self, x):
        z = torch.nn.functional.avg_pool2d(x, 8)                       # (1,3,h,w)
        return z.repeat(1, 6, 1, 1)[:, :16]

    def decode(self, z):
        return torch.nn.functional.interpolate(z[:, :3].float(), scale_factor=8, mode="nearest")


def test_injected_text_encoder_and_autoencoder_end_to_end():
    import PIL.Image
    p = tiny_params()
    enc, ae = _StubEncoder(8, p.context_in_dim, p.vec_in_dim), _StubAutoEncoder()
    pipe = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights="synthetic", params=p,
                                        text_encoder=enc, autoencoder=ae)
    out = pipe.generate_image("a cat on grass", ["cat", "grass"], width=256, height=256, layer_indices=[0, 1],
                              num_inference_steps=2, seed=1)
    # prompt through t5 + clip, each concept through t5 (first token only: concept_attention/utils.py:17-20)
    assert ("t5", "a cat on grass") in enc.calls and ("clip", "a cat on grass") in enc.calls
    assert ("t5", "cat") in enc.calls and ("t5", "grass") in enc.calls
    assert isinstance(out.image, PIL.Image.Image) and out.image.size == (256, 256)
    assert len(out.concept_heatmaps) == 2 and isinstance(out.concept_heatmaps[0], PIL.Image.Image)
    # encode_image(PIL): resize -> autoencoder.encode -> noised forward of the double blocks
    enc.calls.clear()
    img = PIL.Image.fromarray((np.random.default_rng(0).random((200, 300, 3)) * 255).astype(np.uint8))
    e = pipe.encode_image(img, ["cat", "grass"], prompt="a cat", width=256, height=256, layer_indices=[0, 1],
                          num_steps=2, noise_timestep=1, return_pil_heatmaps=False)
    assert e.concept_heatmaps.shape == (2, 16, 16) and np.abs(e.concept_heatmaps.sum(0) - 1).max() < 1e-5
    assert ("t5", "a cat") in enc.calls
    # the same latent handed over directly gives the same maps
    arr = torch.from_numpy(np.asarray(img.convert("RGB"))).permute(2, 0, 1).float() / 255.0
    lat = ae.encode(torch.nn.functional.interpolate((2.0 * arr - 1.0)[None].to(DEV), (256, 256))).to(torch.bfloat16)
    e2 = pipe.encode_image(lat, ["cat", "grass"], prompt="a cat", width=256, height=256, layer_indices=[0, 1],
                           num_steps=2, noise_timestep=1, return_pil_heatmaps=False)
    assert np.array_equal(e.concept_heatmaps, e2.concept_heatmaps)


# ------------------------------------------------------------------ sparsemax / entmax15 (parity unpinned)
@pytest.mark.parametrize("C", [1, 2, 4, 8, 11, 16])
@pytest.mark.parametrize("norm", ["sparsemax", "entmax15"])
def test_sparse_norm_kernel_matches_published_algorithm(C, norm):
    g = torch.Generator().manual_seed(C)
    L = 1000
    logits = (torch.randn(C, L, generator=g) * 2.5).to(DEV)
    acc = torch.full((C, L), 0.25, device=DEV)
    ops.heatmap_softmax_accumulate(logits, acc, 0.5, _lib.NORMS[norm])
    want = 0.25 + 0.5 * getattr(SN, norm)(logits.cpu().numpy(), axis=0)
    assert np.abs(acc.cpu().numpy() - want).max() < 2e-6
    # known answers through the kernel
    if C == 4:
        z = torch.tensor([[1.0, 3.0, 0.0], [0.5, 0.0, 0.0], [-1.0, 0.0, 0.0], [-9.0, -9.0, -9.0]], device=DEV)
        a = torch.zeros(4, 3, device=DEV)
        ops.heatmap_softmax_accumulate(z.contiguous(), a, 1.0, _lib.NORM_SPARSEMAX)
        assert torch.allclose(a[:, 0].cpu(), torch.tensor([0.75, 0.25, 0.0, 0.0]), atol=1e-6)
        assert torch.allclose(a[:, 1].cpu(), torch.tensor([1.0, 0.0, 0.0, 0.0]), atol=1e-6)
        assert torch.allclose(a[:, 2].cpu(), torch.tensor([1 / 3, 1 / 3, 1 / 3, 0.0]), atol=1e-6)


def test_sparse_norm_argument_errors():
    logits, acc = torch.zeros(17, 64, device=DEV), torch.zeros(17, 64, device=DEV)
    with pytest.raises(ValueError):
        ops.heatmap_softmax_accumulate(logits, acc, 1.0, _lib.NORM_SPARSEMAX)   # C > 16
    with pytest.raises(ValueError):
        ops.heatmap_softmax_accumulate(logits[:4].contiguous(), acc[:4].contiguous(), 1.0, 7)  # unknown norm


def test_pipeline_attention_norm_branches():
    """softmax=False selects attention_norm as the reference does (concept_attention_pipeline.py:64-71): fused and
    stacked routes agree, maps are sparse and sum to one, an unknown name is a ValueError."""
    p = tiny_params()
    pipe = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights="synthetic", params=p, n_text_tokens=8)
    kw = dict(prompt="a cat", concepts=["cat", "sky", "tree"], width=256, height=256, layer_indices=[0, 1],
              num_inference_steps=2, seed=2, return_pil_heatmaps=False)
    soft = pipe.generate_image(**kw)
    assert np.array_equal(soft.concept_heatmaps, pipe.generate_image(softmax=True, attention_norm="entmax15", **kw)
                          .concept_heatmaps)  # softmax=True wins, whatever attention_norm says
    for norm in ("sparsemax", "entmax15"):
        a = pipe.generate_image(softmax=False, attention_norm=norm, **kw)
        b = pipe.generate_image(softmax=False, attention_norm=norm, fused=False, **kw)
        assert np.abs(a.concept_heatmaps.sum(0) - 1).max() < 1e-5
        assert np.abs(a.cross_attention_maps - b.cross_attention_maps).max() < 5e-3   # fp32 vs bf16 q vectors
        assert np.abs(a.concept_heatmaps - b.concept_heatmaps).max() < 5e-3  # fp32 vs bf16 concept rows
        assert not np.array_equal(a.concept_heatmaps, soft.concept_heatmaps)
    with pytest.raises(ValueError):
        pipe.generate_image(softmax=False, attention_norm="nope", **kw)
    # stacked route against the oracle-side restatement on the same vectors
    lat = torch.randn(1, 16, 32, 32, generator=torch.Generator().manual_seed(1)).to(DEV, torch.bfloat16)
    _, d = pipe.flux_generator.generate_image(width=256, height=256, num_steps=2, guidance=0.0, seed=0,
                                              prompt="a cat", concepts=["cat", "sky", "tree"], latent=lat)
    hm = compute_heatmaps_from_vectors(d["output_space_image_vectors"], d["output_space_concept_vectors"],
                                       layer_indices=[1], timesteps=[0, 1], softmax=False, attention_norm="sparsemax")
    lg = O.heatmap_logits(d["output_space_image_vectors"].float().cpu(),
                          d["output_space_concept_vectors"].float().cpu()).numpy()
    want = SN.sparsemax(lg, axis=-2)[[0, 1]][:, [1]].mean((0, 1)).reshape(1, 3, 16, 16)
    assert np.abs(hm.cpu().numpy() - want).max() < 1e-4


# ------------------------------------------------------------------ RoPE table caching is by content
def _forward(m, p, h2, w2, inp, ids, tagged=False):
    from conceptattention_amd import sampling
    img = inp["latent"].reshape(1, 16, h2 * 2, w2 * 2)
    d = {k: v.to(DEV) for k, v in inp.items()}
    if tagged:  # ids whose content the model knows: the cached-table route
        d["txt_ids"], d["concept_ids"] = sampling.zero_ids(8, DEV), sampling.zero_ids(2, DEV)
    return m(img=O.patchify(img).to(DEV), img_ids=ids.to(DEV), txt=d["txt"], txt_ids=d["txt_ids"],
             concepts=d["concepts"], concept_ids=d["concept_ids"], concept_vec=d["concept_vec"], y=d["vec"],
             timesteps=torch.tensor([0.5], device=DEV))[0]


def test_rope_table_follows_the_ids_across_orientations():
    """32x8 and 8x32 token grids have the same token count and tensor shapes; the positional table must follow
    the ids of each call (tagged ids from sampling.make_img_ids and plain untagged tensors alike)."""
    from conceptattention_amd import sampling
    p = tiny_params(depth=1, depth_single_blocks=1)
    sd = {k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=1).items()}
    inp = {k: (v.bfloat16().float() if v.is_floating_point() else v)
           for k, v in synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=2, seed=3).items()}
    fresh = {}
    for h2, w2 in ((32, 8), (8, 32)):
        m = HipFluxDiT(p, DEV)
        m.load_state_dict(sd)
        fresh[(h2, w2)] = _forward(m, p, h2, w2, inp, O.make_img_ids(h2, w2))
    assert not torch.equal(fresh[(32, 8)], fresh[(8, 32)])
    m = HipFluxDiT(p, DEV)
    m.load_state_dict(sd)
    for tagged, make in ((True, lambda h, w: sampling.make_img_ids(h, w, DEV)), (False, lambda h, w: O.make_img_ids(h, w))):
        for h2, w2 in ((32, 8), (8, 32), (32, 8), (32, 8)):
            assert torch.equal(_forward(m, p, h2, w2, inp, make(h2, w2), tagged), fresh[(h2, w2)]), (h2, w2)
            assert (m._rope_key is not None) == tagged
    # a tagged tensor modified in place loses its tag (version counter) -> recomputed, not served stale
    ids = sampling.make_img_ids(32, 8, DEV)
    assert torch.equal(_forward(m, p, 32, 8, inp, ids, True), fresh[(32, 8)])
    ids.copy_(O.make_img_ids(8, 32).to(DEV))
    assert torch.equal(_forward(m, p, 8, 32, inp, ids, True), fresh[(8, 32)])


# ------------------------------------------------------------------ fp8: shared weight images, per-call keep list
def test_fp8_cold_start_two_streams_equals_sequential_and_keep_layers_restored():
    p = tiny_params()
    items = []
    for j in range(4):
        inp = synthetic_inputs(p, 256, 256, 8, 3, seed=70 + j, dtype=torch.bfloat16)
        items.append({k: inp[k].to(DEV) for k in ("latent", "txt", "vec", "concepts")})
    kw = dict(layer_indices=[1], num_inference_steps=2)
    cold = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights="synthetic", params=p, n_text_tokens=8,
                                        precision="fp8")
    assert cold.model.weights.fp8 is None            # nothing quantised yet: the two-stream call starts cold
    par = cold.generate_many_on_device(items, n_streams=2, **kw)
    torch.cuda.synchronize()
    warm = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights="synthetic", params=p, n_text_tokens=8,
                                        precision="fp8")
    seq = [warm.generate_on_device(i["latent"], i["txt"], i["vec"], i["concepts"], **kw) for i in items]
    torch.cuda.synchronize()
    for a, b in zip(par, seq):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # generate keeps the heat-map layers in bf16 for THIS call only
    for m in cold._replicas + [warm.model]:
        assert m.precision == "fp8" and m.keep_bf16_layers == frozenset()
    # ... so a sweep afterwards runs every block in fp8, exactly like on a pipeline that never generated
    i0 = items[0]
    a = warm.layer_noise_sweep_on_device(i0["latent"], i0["txt"], i0["vec"], i0["concepts"], noise_levels=[1],
                                         num_steps=2)
    never = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights="synthetic", params=p, n_text_tokens=8,
                                         precision="fp8")
    b = never.layer_noise_sweep_on_device(i0["latent"], i0["txt"], i0["vec"], i0["concepts"], noise_levels=[1],
                                          num_steps=2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_timestep_indices_are_validated_like_fancy_indexing():
    p = tiny_params()
    pipe = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights="synthetic", params=p, n_text_tokens=8)
    kw = dict(prompt="a cat", concepts=["cat", "sky"], width=256, height=256, layer_indices=[0],
              num_inference_steps=2, seed=2, return_pil_heatmaps=False)
    with pytest.raises(IndexError):
        pipe.generate_image(timesteps=[0, 2], **kw)       # the reference's heatmaps[timesteps] raises here too
    last = pipe.generate_image(timesteps=[1], **kw)
    neg = pipe.generate_image(timesteps=[-1], **kw)        # negative indices count from the end
    assert np.array_equal(last.concept_heatmaps, neg.concept_heatmaps)


# ------------------------------------------------------------------ fp32 residual stream (kernels)
@pytest.mark.parametrize("tile", [_lib.TILE_PP_256x256, _lib.TILE_PP_256x192, _lib.TILE_PP_256x128, _lib.TILE_256x64])
@pytest.mark.parametrize("M", [260, 1000])
def test_gemm_fp32_output_bias_and_gate_residual(tile, M):
    """out_f32: img_in / txt_in write the fp32 residual stream (BIAS), proj / mlp.2 / linear2 update it in place
    (GATE_RESIDUAL with two gate vectors); compared with an fp64 evaluation of the same bf16 operands."""
    g = torch.Generator().manual_seed(M + tile)
    N, K = 768, 512
    a = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV, torch.bfloat16)
    b = torch.randn(N, generator=g).to(DEV, torch.bfloat16)
    prod = a.double() @ w.double().T + b.double()
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm([ops.Gemm(a, w, b, out)], tile)
    assert (out.double() - prod).abs().max().item() < 1e-4 * prod.abs().max().item()      # fp32 accumulate, no rounding
    x = torch.randn(M, N, generator=g).to(DEV)                                             # fp32 residual
    g1, g2 = torch.randn(N, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    want = x.double().clone()
    want[:4] += g1.double() * prod[:4]
    want[4:] += g2.double() * prod[4:]
    ops.gemm([ops.Gemm(a, w, b, x, _lib.EPI_GATE_RESIDUAL, resid=x, gate=g1, gate2=g2, gate_rows=4)], tile)
    assert (x.double() - want).abs().max().item() < 1e-4 * want.abs().max().item()
    # strided fp32 output (a row slice of a wider buffer) and rejection of the epilogues that have no fp32 form
    wide = torch.zeros(M, N + 64, device=DEV)
    ops.gemm([ops.Gemm(a, w, b, wide[:, 32:32 + N])], tile)
    assert torch.equal(wide[:, 32:32 + N], out) and wide[:, :32].abs().max() == 0
    with pytest.raises(ValueError):
        ops.gemm([ops.Gemm(a, w, b, out, _lib.EPI_GELU_TANH)], tile)


def test_ln_modulate_fp32_input_matches_bf16_input_on_representable_rows():
    g = torch.Generator().manual_seed(5)
    M, H = 777, 3072
    xb = torch.randn(M, H, generator=g).to(DEV, torch.bfloat16)
    sh, sc = torch.randn(H, generator=g).to(DEV), torch.randn(H, generator=g).to(DEV) * 0.2
    sh2, sc2 = torch.randn(H, generator=g).to(DEV), torch.randn(H, generator=g).to(DEV) * 0.2
    segs = [(100, sh, sc), (M, sh2, sc2)]
    o16, o32 = torch.empty(M, H, device=DEV, dtype=torch.bfloat16), torch.empty(M, H, device=DEV, dtype=torch.bfloat16)
    ops.ln_modulate(xb, o16, segs)
    ops.ln_modulate(xb.float(), o32, segs)
    assert torch.equal(o16, o32)              # same values in, same arithmetic
    # genuinely fp32 rows against the oracle's formula
    xf = torch.randn(M, H, generator=g).to(DEV) * 3 + 0.1234567
    ops.ln_modulate(xf, o32, segs)
    mu, var = xf.double().mean(-1, keepdim=True), xf.double().var(-1, unbiased=False, keepdim=True)
    ln = (xf.double() - mu) / torch.sqrt(var + 1e-6)
    want = torch.cat(((1 + sc.double()) * ln[:100] + sh.double(), (1 + sc2.double()) * ln[100:] + sh2.double()))
    assert (o32.double() - want).abs().max().item() < 2 ** -7 * want.abs().max().item()
    # fp8 output from an fp32 input: identical bytes / scales to the bf16-input kernel on representable rows
    q16, s16 = torch.empty(M, H, device=DEV, dtype=torch.uint8), torch.empty(M, device=DEV)
    q32, s32 = torch.empty(M, H, device=DEV, dtype=torch.uint8), torch.empty(M, device=DEV)
    ops.ln_modulate(xb, q16, segs, out_scale=s16)
    ops.ln_modulate(xb.float(), q32, segs, out_scale=s32)
    assert torch.equal(q16, q32) and torch.equal(s16, s32)


def test_fp32_residual_is_closer_to_the_oracle_than_bf16_residual():
    """Tiny model, 2 + 2 blocks: both layouts run the same kernels; the fp32 stream must not be further from the
    fp32 oracle than the bf16 one on `pred`, and both stay within the model test's bounds."""
    p = tiny_params()
    sd = {k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=1).items()}
    inp = {k: (v.bfloat16().float() if v.is_floating_point() else v)
           for k, v in synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=3, seed=2).items()}
    pred_o, _ = O.dit_forward(sd, p, O.patchify(inp["latent"]), inp["img_ids"], inp["txt"], inp["txt_ids"],
                              inp["concepts"], inp["concept_ids"], inp["concept_vec"], torch.tensor([0.75]), inp["vec"])
    errs = {}
    for dt in (torch.float32, torch.bfloat16):
        m = HipFluxDiT(p, DEV, residual_dtype=dt)
        m.load_state_dict(sd)
        d = {k: v.to(DEV) for k, v in inp.items()}
        pred, _ = m(img=O.patchify(inp["latent"]).to(DEV), img_ids=d["img_ids"], txt=d["txt"], txt_ids=d["txt_ids"],
                    concepts=d["concepts"], concept_ids=d["concept_ids"], concept_vec=d["concept_vec"], y=d["vec"],
                    timesteps=torch.tensor([0.75], device=DEV))
        assert m.X.dtype == dt
        errs[dt] = (pred.float().cpu() - pred_o).pow(2).mean().sqrt().item()
    assert errs[torch.float32] <= errs[torch.bfloat16] * 1.05, errs


# ------------------------------------------------------------------ batched forward (B work items per launch)
@pytest.mark.parametrize("guidance_embed", [False, True])
def test_batched_forward_is_bit_identical_to_one_item_at_a_time(guidance_embed):
    """B items through one forward (rows [concepts of all | text of all | image of all], per-item adaLN vectors and
    gates, 2B attention problems per launch) give, per item, the bits of a B = 1 forward: pred, all four vector
    stacks and the fused maps.  Items differ in every input, including the timestep."""
    from conceptattention_amd.flux_dit import HeatmapRequest
    p = tiny_params(guidance_embed=guidance_embed)
    sd = {k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=1).items()}
    B, T, C = 3, 8, 3
    inps = [{k: (v.bfloat16() if v.is_floating_point() else v).to(DEV)
             for k, v in synthetic_inputs(p, 256, 256, n_txt=T, n_concepts=C, seed=20 + j).items()} for j in range(B)]
    ts = torch.tensor([0.9, 0.5, 0.2], device=DEV)
    gd = torch.tensor([1.0, 2.5, 4.0], device=DEV)
    m = HipFluxDiT(p, DEV)
    m.load_state_dict(sd)

    def call(idx, heat):
        cat = lambda k: torch.cat([inps[j][k] for j in idx], 0)  # noqa: E731
        img = torch.cat([O.patchify(inps[j]["latent"].float().cpu()).to(DEV, torch.bfloat16) for j in idx], 0)
        return m(img=img, img_ids=cat("img_ids"), txt=cat("txt"), txt_ids=cat("txt_ids"), concepts=cat("concepts"),
                 concept_ids=cat("concept_ids"), concept_vec=cat("concept_vec"), y=cat("vec"), timesteps=ts[idx],
                 guidance=gd[idx], heatmaps=heat)
    mk = lambda: HeatmapRequest((0, 1), 0.5, torch.zeros(C, 256, device=DEV), torch.zeros(C, 256, device=DEV))  # noqa: E731
    one = []
    for j in range(B):
        h = mk()
        pred, d = call([j], [h])
        one.append((pred.clone(), {k: v.clone() for k, v in d.items()}, h))
    hs = [mk() for _ in range(B)]
    pred, d = call(list(range(B)), hs)
    torch.cuda.synchronize()
    assert pred.shape == (B, 256, p.in_channels)
    assert d["output_space_image_vectors"].shape == (p.depth, B, 256, p.hidden_size)
    assert d["cross_attention_concept_vectors"].shape == (p.depth, B, p.num_heads, C, 128)
    for j in range(B):
        assert torch.equal(pred[j], one[j][0][0]), j
        for k in d:
            assert torch.equal(d[k][:, j], one[j][1][k][:, 0]), (j, k)
        assert torch.equal(hs[j].out_space, one[j][2].out_space) and torch.equal(hs[j].cross_space, one[j][2].cross_space)


def test_generate_many_batched_equals_one_by_one_including_fp8():
    p = tiny_params()
    items = []
    for j in range(7):
        inp = synthetic_inputs(p, 256, 256, 8, 3, seed=90 + j, dtype=torch.bfloat16)
        items.append({k: inp[k].to(DEV) for k in ("latent", "txt", "vec", "concepts")})
    kw = dict(layer_indices=[0, 1], num_inference_steps=2)
    for precision in ("bf16", "fp8"):
        pipe = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights="synthetic", params=p, n_text_tokens=8,
                                            precision=precision)
        seq = [pipe.generate_on_device(i["latent"], i["txt"], i["vec"], i["concepts"], **kw) for i in items]
        for batch, streams in ((3, 1), (5, 1), (2, 2)):
            par = pipe.generate_many_on_device(items, n_streams=streams, batch=batch, **kw)
            torch.cuda.synchronize()
            assert len(par) == len(items)
            for a, b in zip(seq, par):
                for x, y in zip(a, b):
                    assert x.shape == y.shape and torch.equal(x, y), (precision, batch, streams)


# ------------------------------------------------------------------ RCCL (one rank: all a 1-GPU box allows)
def test_rccl_one_rank_group_runs_the_device_collectives():
    """backend "nccl" IS RCCL on ROCm.  A one-GPU box cannot host two RCCL ranks, but a one-rank group still goes
    through RCCL's communicator setup and its all_gather / all_reduce kernels on device tensors: the branch of
    gather_heatmaps / allreduce_sum_ / max_over_ranks that the multi-GPU bench takes."""
    import subprocess
    import sys
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CA_ROOT"])
from conceptattention_amd import distributed as D
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
x = torch.arange(3 * 2 * 4 * 8 * 8, dtype=torch.float32, device="cuda:0").view(3, 2, 4, 8, 8)
full = D.gather_heatmaps(x, 3, 0, 1, force_collective=True)
assert full.is_cuda and torch.equal(full, x)
acc = torch.full((4, 64), 2.5, device="cuda:0")
assert torch.equal(D.allreduce_sum_(acc.clone(), force_collective=True), acc)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
ev = D.collective_evidence(torch.device("cuda:0"), force_collective=True)   # the record an N > 1 bench line carries
assert ev["backend"].startswith("rccl") and ev["ranks_seen"] == 1 and ev["devices"] == [0] and ev["distinct_devices"] == 1, ev
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
"""
    env = dict(os.environ, CA_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_bf16_timesteps_switch_reproduces_the_reference_embedding(golden):
    """HipFluxDiT(bf16_timesteps=True) embeds what the reference's bf16 run embeds (752 for t = 0.75): checked on the
    reference's own timestep_embedding output for both schedules; the default keeps t exact (fp32 oracle)."""
    g = golden("timestep_embedding_bf16.npz")
    t = torch.from_numpy(g["t"]).float().to(DEV)
    arg = (t.to(torch.bfloat16) * 1000.0).float()
    assert np.array_equal(arg.cpu().numpy(), g["arg_bf16"])          # the argument the reference forms
    assert float(g["arg_bf16"][1]) == 752.0                           # t = 0.75
    emb = torch.empty(t.numel(), 256, device=DEV)
    ops.timestep_embedding(arg, emb, time_factor=1.0)
    got = emb.to(torch.bfloat16).float().cpu().numpy()
    # cos / sin of arguments up to 3500 rad in fp32: the device and host libm agree to ~1e-4 before the bf16 cast,
    # so all but a handful of values land on the same bf16 and none is more than one bf16 step of 1.0 away
    d = np.abs(got - g["emb"])
    assert d.max() <= 2 ** -7 and (d > 0).mean() < 0.02
    # through the model: the switch changes the conditioning at t = 0.75 (750 vs 752), not at t = 0.5 (exact)
    p = tiny_params(depth=1, depth_single_blocks=0)
    sd = {k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=1).items()}
    vec = torch.randn(1, p.vec_in_dim, device=DEV)
    mods = {}
    for sw in (False, True):
        m = HipFluxDiT(p, DEV, bf16_timesteps=sw)
        m.load_state_dict(sd)
        m.precompute_conditioning([0.75, 0.5], vec, torch.zeros_like(vec))
        mods[sw] = m._mod_steps.clone()
    assert not torch.equal(mods[False][0], mods[True][0])
    assert (mods[False][1] - mods[True][1]).abs().max() < 2e-2 * mods[False][1].abs().max()  # only the bf16 cast of the embedding


def test_capture_buffers_are_lazy_and_workspaces_can_be_cleared():
    """The fp32 capture buffers (QPRE, XML, QD, ATTI32: ~0.6 GB per work item at 1024 x 1024) exist only once a forward
    asked for maps, and clear_workspaces() drops every cached activation set."""
    from conceptattention_amd.weights import synthetic_inputs, synthetic_state_dict
    p = tiny_params()
    sd = {k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=1).items()}
    pl = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights=sd, params=p, n_text_tokens=8)
    m = pl.model
    inp = synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=2, seed=2, dtype=torch.bfloat16)
    from conceptattention_amd import sampling as S
    con, con_ids, con_vec = S.concept_inputs(inp["concepts"].to(DEV), inp["vec"].to(DEV))
    prep = S.prepare_from_embeddings(inp["latent"].to(DEV), inp["txt"].to(DEV), inp["vec"].to(DEV))
    kw = dict(img=prep["img"], img_ids=prep["img_ids"], txt=prep["txt"], txt_ids=prep["txt_ids"], concepts=con,
              concept_ids=con_ids, concept_vec=con_vec, y=prep["vec"], timesteps=torch.ones(1, device=DEV),
              guidance=torch.zeros(1, device=DEV))
    m(**kw, return_vectors=False)                       # no maps asked for
    ws = m._ws_cache[m._ws_key]
    assert not any(k in ws for k in ("QPRE", "XML", "QD", "ATTI32"))
    before = m.workspace_bytes()
    pred, d = m(**kw)                                   # the reference's call: all vectors returned
    assert "QPRE" in ws and m.workspace_bytes() > before
    assert d["cross_attention_image_vectors"].shape[0] == p.depth
    m.clear_workspaces()
    assert m.workspace_bytes() == 0 and m._ws_key is None
    pred2, d2 = m(**kw)                                 # allocates afresh, same bits
    assert torch.equal(pred, pred2)
    assert torch.equal(d["output_space_image_vectors"], d2["output_space_image_vectors"])
