"""Full-size, full-depth goldens (BASELINE.json configs[1] and configs[0] shapes).  TEST INFRASTRUCTURE ONLY.

Runs in the build container only (imports the reference from /root/reference for the bf16 yardstick):

    python -B oracle/make_full_goldens.py fp32        # ~20 min: the pinned fp32 oracle, 4 full-size steps
    python -B oracle/make_full_goldens.py refbf16     # ~6 min: the REFERENCE's own modules in bf16, same weights
    python -B oracle/make_full_goldens.py cfg1        # config 1 shape (256x256, C=1, 1 step) at H=3072
    python -B oracle/make_full_goldens.py dev         # ~15 min: flux-dev geometry (guidance 3.5, T=512, C=8), first 2
                                                      #   steps of the shifted 50-step schedule, all 57 blocks, fp32
    python -B oracle/make_full_goldens.py devref      # the REFERENCE's modules in bf16 on the same (yardstick)
    python -B oracle/make_full_goldens.py encode      # encode_image path at H=3072, C=2: add_noise_to_image + ONE
                                                      #   stop_after_multimodal_attentions forward (19 blocks), fp32
    python -B oracle/make_full_goldens.py encoderef   # the reference in bf16 on the same (yardstick)

Writes tests/golden/full_depth_schnell.npz (fp32 maps), tests/golden/full_depth_refbf16.npz (how far the
reference's own bf16 run is from those maps: the yardstick of tests/test_full_depth_gpu.py) and
tests/golden/cfg1_full_hidden.npz, and full_depth_dev.npz / full_depth_dev_refbf16.npz / encode_full.npz /
encode_full_refbf16.npz for BASELINE.json configs[2] and configs[3].  Only numbers are stored: heat maps, sample rows, error statistics.

Setup shared with the GPU test: flux-schnell geometry (19 double + 38 single blocks, H=3072), weights =
``weights.synth_tensor(name, seed=0)`` rounded to bf16, inputs = ``weights.synthetic_inputs(seed=5)`` rounded
to bf16, 1024x1024 (4096 image tokens), 256 text tokens, 4 concepts, schedule [1, .75, .5, .25, 0], guidance 0.
"""
from __future__ import annotations

import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from conceptattention_amd.params import configs
from conceptattention_amd.weights import state_dict_spec, synth_tensor, synthetic_inputs
from oracle import flux_oracle as O

GOLD = os.path.join(ROOT, "tests", "golden")
SEED_W, SEED_IN = 0, 5
SAMPLE_ROWS = np.arange(5, 4096, 67)[:61]


class LazySD(dict):
    """Generates each bf16-rounded tensor on access and keeps only the last few (a full fp32 model is 47.6 GB)."""

    def __init__(self, p, seed=SEED_W, keep=8):
        super().__init__()
        self.spec = dict(state_dict_spec(p))
        self.fan = {n[:-7]: s[1] for n, s in self.spec.items() if n.endswith(".weight")}
        self.seed, self.keep, self.cache = seed, keep, {}

    def make(self, name):
        return synth_tensor(name, self.spec[name], self.fan.get(name.rsplit(".", 1)[0], 1), seed=self.seed).bfloat16()

    def __getitem__(self, name):
        if name not in self.cache:
            if len(self.cache) >= self.keep:
                self.cache.pop(next(iter(self.cache)))
            self.cache[name] = self.make(name).float()
        return self.cache[name]

    def get(self, name, default=None):
        return self[name] if name in self.spec else default


def inputs(p, size, T, C):
    return {k: (v.bfloat16().float() if v.is_floating_point() else v)
            for k, v in synthetic_inputs(p, size, size, T, C, seed=SEED_IN).items()}


def layer_maps(d):
    """fp32 softmax-over-concepts maps [C, L] of one layer's dict (both spaces), computed the way
    oracle.compute_heatmaps does for a single (t, layer) pair."""
    st = {k: v[None, None].float() for k, v in d.items()}
    side = int(round(st["output_space_image_vectors"].shape[-2] ** 0.5))
    ho = O.compute_heatmaps(st["output_space_image_vectors"], st["output_space_concept_vectors"], [0], [0], side=side)
    hc = O.compute_heatmaps(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"], [0], [0],
                            side=side)
    return ho[0].reshape(ho.shape[1], -1), hc[0].reshape(hc.shape[1], -1)


def oracle_step(sd, p, img, inp, t, log, on_layer, guidance=None, y=None, stop_after=False):
    """One DiT step of the fp32 oracle, block by block (weights streamed); on_layer(i, dict) per double block.
    ``guidance``: embedded when the geometry has guidance_in (modified_flux_dit.py:100-103,114-117); ``y``: the
    pooled vector of the image/text stream (encode_image passes the all-zero concept_vec,
    concept_attention_pipeline.py:291); ``stop_after``: return after the double blocks (:152-153)."""
    nh = p.num_heads
    T = inp["txt"].shape[1]
    x_img = O.linear(sd, "img_in", img)
    temb = O.timestep_embedding(torch.tensor([t]))
    vec, cvec = O.mlp_embedder(sd, "time_in", temb), O.mlp_embedder(sd, "time_in", temb)
    if p.guidance_embed:
        gemb = O.mlp_embedder(sd, "guidance_in", O.timestep_embedding(torch.tensor([float(guidance)])))
        vec, cvec = vec + gemb, cvec + gemb
    vec = vec + O.mlp_embedder(sd, "vector_in", inp["vec"] if y is None else y)
    cvec = cvec + O.mlp_embedder(sd, "vector_in", inp["concept_vec"])
    x_txt = O.linear(sd, "txt_in", inp["txt"])
    x_con = O.linear(sd, "txt_in", inp["concepts"])
    rope_ti = O.rope_cos_sin(torch.cat((inp["txt_ids"][0], inp["img_ids"][0])), p.axes_dim, p.theta)
    rope_ci = O.rope_cos_sin(torch.cat((inp["concept_ids"][0], inp["img_ids"][0])), p.axes_dim, p.theta)
    for i in range(p.depth):
        x_img, x_txt, x_con, od = O.double_block(sd, f"double_blocks.{i}.", nh, x_img, x_txt, vec, rope_ti, x_con,
                                                 cvec, rope_ci)
        on_layer(i, od)
        log(f"double {i}")
    if stop_after:
        return None
    x = torch.cat((x_txt, x_img), 1)
    for i in range(p.depth_single_blocks):
        x = O.single_block(sd, f"single_blocks.{i}.", nh, x, vec, rope_ti)
        if i % 8 == 7:
            log(f"single {i}")
    x = x[:, T:]
    shift, scale = O.linear(sd, "final_layer.adaLN_modulation.1", torch.nn.functional.silu(vec)).chunk(2, dim=1)
    x = (1 + scale[:, None, :]) * O.layer_norm(x) + shift[:, None, :]
    return O.linear(sd, "final_layer.linear", x)


def run_fp32(name="full_depth_schnell.npz", size=1024, T=256, C=4, steps=4):
    p = configs["flux-schnell"]
    t0 = time.time()

    def log(msg):
        print(f"[{time.time() - t0:6.0f}s] {msg}", flush=True)
    inp = inputs(p, size, T, C)
    sd = LazySD(p)
    img = O.patchify(inp["latent"])
    L = img.shape[1]
    ts = O.get_schedule(steps, L, shift=False)
    out = np.zeros((steps, p.depth, C, L), np.float32)
    cross = np.zeros((steps, p.depth, C, L), np.float32)
    rows = SAMPLE_ROWS[SAMPLE_ROWS < L]
    pred_rows, attn_absmax = [], np.zeros((steps, p.depth), np.float32)
    for s, (tc, tp) in enumerate(zip(ts[:-1], ts[1:])):
        def on_layer(i, od, s=s):
            ho, hc = layer_maps(od)
            out[s, i], cross[s, i] = ho.numpy(), hc.numpy()
            attn_absmax[s, i] = od["output_space_image_vectors"].abs().max().item()
        pred = oracle_step(sd, p, img, inp, tc, lambda m, s=s: log(f"step {s} {m}"), on_layer)
        pred_rows.append(pred[0, rows].numpy())
        img = img + (tp - tc) * pred
        np.save(os.path.join("/tmp", f"full_fp32_partial_{name}.npy"), out)  # survives an interrupted run
    arrays = dict(
        schedule=np.array(ts), sample_rows=rows, pred_rows=np.stack(pred_rows), final_img_rows=img[0, rows].numpy(),
        final_img_absmean=np.array(img.abs().mean().item()),
        # step 0: every layer; later steps: the default layers 15..18 only (fixture size)
        out_step0=out[0], cross_step0=cross[0], out_late=out[1:, 15:19], cross_late=cross[1:, 15:19],
        final_out=out[:, 15:19].mean((0, 1)), final_cross=cross[:, 15:19].mean((0, 1)), attn_absmax=attn_absmax,
        weight_seed=np.array(SEED_W), input_seed=np.array(SEED_IN))
    np.savez_compressed(os.path.join(GOLD, name), **arrays)
    np.save("/tmp/full_fp32_all_out.npy", out)      # complete per-(step, layer) tables for run_refbf16 (not committed)
    np.save("/tmp/full_fp32_all_cross.npy", cross)
    log(f"wrote {name}")


def _import_reference():
    import types
    pkg = types.ModuleType("concept_attention")
    pkg.__path__ = ["/root/reference/concept_attention"]
    sys.modules["concept_attention"] = pkg
    from concept_attention.modified_flux_dit import ModifiedFluxDiT, FluxParams
    from concept_attention.flux.src.flux import sampling
    return ModifiedFluxDiT, FluxParams, sampling


def run_refbf16(name="full_depth_refbf16.npz", size=1024, T=256, C=4, steps=4):
    """The reference's own ModifiedFluxDiT + denoise in bf16 (its production dtype) on the same seeded weights and
    inputs; stores how far ITS heat maps are from the fp32 maps of run_fp32 -- the yardstick for the HIP path."""
    p = configs["flux-schnell"]
    t0 = time.time()

    def log(msg):
        print(f"[{time.time() - t0:6.0f}s] {msg}", flush=True)
    ModifiedFluxDiT, FluxParams, sampling = _import_reference()
    rp = FluxParams(in_channels=p.in_channels, vec_in_dim=p.vec_in_dim, context_in_dim=p.context_in_dim,
                    hidden_size=p.hidden_size, mlp_ratio=p.mlp_ratio, num_heads=p.num_heads, depth=p.depth,
                    depth_single_blocks=p.depth_single_blocks, axes_dim=list(p.axes_dim), theta=p.theta,
                    qkv_bias=p.qkv_bias, guidance_embed=p.guidance_embed)
    with torch.device("meta"):
        model = ModifiedFluxDiT(rp)
    lazy = LazySD(p)
    sd = {n: lazy.make(n) for n in lazy.spec}
    model.load_state_dict(sd, strict=True, assign=True)
    model.eval()
    log("reference model in bf16 built")
    inp = inputs(p, size, T, C)
    bf = torch.bfloat16
    img = O.patchify(inp["latent"]).to(bf)
    L = img.shape[1]
    ts = sampling.get_schedule(steps, L, shift=False)
    with torch.no_grad():
        x, _, d = sampling.denoise(model, img=img, img_ids=inp["img_ids"], txt=inp["txt"].to(bf),
                                   txt_ids=inp["txt_ids"], vec=inp["vec"].to(bf), timesteps=ts, guidance=0.0,
                                   concepts=inp["concepts"].to(bf), concept_ids=inp["concept_ids"],
                                   concept_vec=inp["concept_vec"].to(bf))
    log("reference bf16 denoise done")
    gold_out, gold_cross = np.load("/tmp/full_fp32_all_out.npy"), np.load("/tmp/full_fp32_all_cross.npy")
    err_out = np.zeros((steps, p.depth), np.float32)
    err_cross = np.zeros((steps, p.depth), np.float32)
    agree_cross = np.zeros((steps, p.depth), np.float32)
    maps_out = np.zeros((steps, p.depth, C, L), np.float32)
    maps_cross = np.zeros((steps, p.depth, C, L), np.float32)
    for s in range(steps):
        for i in range(p.depth):
            ho, hc = layer_maps({k: v[s, i] for k, v in d.items()})   # fp32 reduction of the bf16 vectors
            maps_out[s, i], maps_cross[s, i] = ho.numpy(), hc.numpy()
            err_out[s, i] = np.abs(maps_out[s, i] - gold_out[s, i]).max()
            err_cross[s, i] = np.abs(maps_cross[s, i] - gold_cross[s, i]).max()
            agree_cross[s, i] = (maps_cross[s, i].argmax(0) == gold_cross[s, i].argmax(0)).mean()
    # the reference's own all-bf16 reduction for the default layers / timesteps (concept_attention_pipeline.py:29-91
    # hard-codes a 64x64 grid, which is this case)
    sys.modules.pop("concept_attention.concept_attention_pipeline", None)
    from oracle.make_goldens import _import_reference as import_pipeline_level
    fn = import_pipeline_level()["compute_heatmaps_from_vectors"]
    with torch.no_grad():
        own_out = fn(d["output_space_image_vectors"], d["output_space_concept_vectors"],
                     layer_indices=list(range(15, 19)), timesteps=list(range(steps)), softmax=True)
        own_cross = fn(d["cross_attention_image_vectors"], d["cross_attention_concept_vectors"],
                       layer_indices=list(range(15, 19)), timesteps=list(range(steps)), softmax=True)
    g_final_out, g_final_cross = gold_out[:, 15:19].mean((0, 1)), gold_cross[:, 15:19].mean((0, 1))
    arrays = dict(
        err_out=err_out, err_cross=err_cross, agree_cross=agree_cross,
        final_err_out_fp32reduce=np.array(np.abs(maps_out[:, 15:19].mean((0, 1)) - g_final_out).max()),
        final_err_cross_fp32reduce=np.array(np.abs(maps_cross[:, 15:19].mean((0, 1)) - g_final_cross).max()),
        final_err_out_own_bf16_reduce=np.array(np.abs(own_out.float().numpy().reshape(C, L) - g_final_out).max()),
        final_err_cross_own_bf16_reduce=np.array(np.abs(own_cross.float().numpy().reshape(C, L) - g_final_cross).max()),
        final_img_rows=x[0, SAMPLE_ROWS].float().numpy())
    np.savez_compressed(os.path.join(GOLD, name), **arrays)
    log(f"wrote {name}")
    print("reference bf16 vs fp32 oracle, output-space max-abs per (step, layer 15..18):\n", err_out[:, 15:19])
    print("cross-space:\n", err_cross[:, 15:19])
    print({k: float(v) for k, v in arrays.items() if v.ndim == 0})


def run_cfg1(name="cfg1_full_hidden.npz"):
    """BASELINE.json configs[0] shape at the full hidden size: 256x256 (L=256), 256 text tokens, C=1, ONE step,
    all 57 blocks, fp32 oracle.  (The reference itself cannot run this size: its reduction hard-codes 64x64.)"""
    p = configs["flux-schnell"]
    t0 = time.time()

    def log(msg):
        print(f"[{time.time() - t0:6.0f}s] cfg1 {msg}", flush=True)
    inp = inputs(p, 256, 256, 1)
    sd = LazySD(p)
    img = O.patchify(inp["latent"])
    L = img.shape[1]
    out = np.zeros((p.depth, 1, L), np.float32)
    cross = np.zeros((p.depth, 1, L), np.float32)
    vec_out, vec_con = [], []

    def on_layer(i, od):
        # C = 1: the softmax over concepts is identically 1, so the maps carry no information; keep the raw
        # vectors' sample rows and logits instead
        logits = od["output_space_image_vectors"][0] @ od["output_space_concept_vectors"][0].T
        out[i] = logits.T.numpy()
        iq = od["cross_attention_image_vectors"][0].permute(1, 0, 2).reshape(L, -1)
        cq = od["cross_attention_concept_vectors"][0].permute(1, 0, 2).reshape(1, -1)
        cross[i] = (iq @ cq.T).T.numpy()
        vec_out.append(od["output_space_image_vectors"][0, ::16].numpy())
        vec_con.append(od["output_space_concept_vectors"][0].numpy())
    pred = oracle_step(sd, p, img, inp, 1.0, log, on_layer)
    np.savez_compressed(os.path.join(GOLD, name), logits_out=out, logits_cross=cross, img_attn_rows=np.stack(vec_out),
                        concept_attn=np.stack(vec_con), pred=pred[0].numpy(), weight_seed=np.array(SEED_W),
                        input_seed=np.array(SEED_IN))
    log(f"wrote {name}")


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2] (flux-dev, 8 concepts, shifted 50-step schedule) and configs[3] (encode_image path)
DEV_T, DEV_C, DEV_GUIDANCE, DEV_STEPS, DEV_OF = 512, 8, 3.5, 2, 50
DEV_KEEP0 = (0, 9)            # step-0 layers kept besides 15..18 (fixture size: 131 KB per layer and space at C = 8)
ENC_C, ENC_STEPS, ENC_NOISE_TIMESTEP, ENC_NOISE_SEED = 2, 4, 2, 77


def _logger(tag):
    t0 = time.time()

    def log(msg):
        print(f"[{time.time() - t0:6.0f}s] {tag} {msg}", flush=True)
    return log


def run_dev_fp32(name="full_depth_dev.npz", size=1024):
    """flux-dev geometry (guidance_in present), guidance 3.5, T = 512 text tokens, C = 8 concepts, the FIRST TWO steps of
    the shifted 50-step schedule (get_schedule(50, 4096, shift=True), flux/sampling.py:78-94), all 57 blocks, fp32."""
    p = configs["flux-dev"]
    log = _logger("dev")
    inp = inputs(p, size, DEV_T, DEV_C)
    sd = LazySD(p)
    img = O.patchify(inp["latent"])
    L = img.shape[1]
    ts = O.get_schedule(DEV_OF, L, shift=True)[:DEV_STEPS + 1]
    out = np.zeros((DEV_STEPS, p.depth, DEV_C, L), np.float32)
    cross = np.zeros_like(out)
    rows = SAMPLE_ROWS[SAMPLE_ROWS < L]
    pred_rows = []
    for s, (tc, tp) in enumerate(zip(ts[:-1], ts[1:])):
        def on_layer(i, od, s=s):
            ho, hc = layer_maps(od)
            out[s, i], cross[s, i] = ho.numpy(), hc.numpy()
        pred = oracle_step(sd, p, img, inp, tc, lambda m, s=s: log(f"step {s} {m}"), on_layer, guidance=DEV_GUIDANCE)
        pred_rows.append(pred[0, rows].numpy())
        img = img + (tp - tc) * pred
    keep0 = list(DEV_KEEP0) + list(range(15, 19))
    arrays = dict(schedule=np.array(ts), sample_rows=rows, pred_rows=np.stack(pred_rows),
                  final_img_rows=img[0, rows].numpy(), layers_step0=np.array(keep0),
                  out_step0=out[0, keep0], cross_step0=cross[0, keep0], out_step1=out[1, 15:19],
                  cross_step1=cross[1, 15:19], final_out=out[:, 15:19].mean((0, 1)),
                  final_cross=cross[:, 15:19].mean((0, 1)), guidance=np.array(DEV_GUIDANCE),
                  weight_seed=np.array(SEED_W), input_seed=np.array(SEED_IN))
    np.savez_compressed(os.path.join(GOLD, name), **arrays)
    np.save("/tmp/dev_fp32_all_out.npy", out)
    np.save("/tmp/dev_fp32_all_cross.npy", cross)
    log(f"wrote {name}")


def _reference_model(p, log):
    ModifiedFluxDiT, FluxParams, sampling = _import_reference()
    rp = FluxParams(in_channels=p.in_channels, vec_in_dim=p.vec_in_dim, context_in_dim=p.context_in_dim,
                    hidden_size=p.hidden_size, mlp_ratio=p.mlp_ratio, num_heads=p.num_heads, depth=p.depth,
                    depth_single_blocks=p.depth_single_blocks, axes_dim=list(p.axes_dim), theta=p.theta,
                    qkv_bias=p.qkv_bias, guidance_embed=p.guidance_embed)
    with torch.device("meta"):
        model = ModifiedFluxDiT(rp)
    lazy = LazySD(p)
    model.load_state_dict({n: lazy.make(n) for n in lazy.spec}, strict=True, assign=True)
    model.eval()
    log("reference model in bf16 built")
    return model, sampling


def _yardstick(d, gold_out, gold_cross, steps, depth, C, L):
    err_out = np.zeros((steps, depth), np.float32)
    err_cross = np.zeros((steps, depth), np.float32)
    agree_cross = np.zeros((steps, depth), np.float32)
    maps_out = np.zeros((steps, depth, C, L), np.float32)
    maps_cross = np.zeros((steps, depth, C, L), np.float32)
    for s in range(steps):
        for i in range(depth):
            ho, hc = layer_maps({k: v[s, i] for k, v in d.items()})   # fp32 reduction of the bf16 vectors
            maps_out[s, i], maps_cross[s, i] = ho.numpy(), hc.numpy()
            err_out[s, i] = np.abs(maps_out[s, i] - gold_out[s, i]).max()
            err_cross[s, i] = np.abs(maps_cross[s, i] - gold_cross[s, i]).max()
            agree_cross[s, i] = (maps_cross[s, i].argmax(0) == gold_cross[s, i].argmax(0)).mean()
    return dict(err_out=err_out, err_cross=err_cross, agree_cross=agree_cross,
                final_err_out_fp32reduce=np.array(np.abs(maps_out[:, 15:19].mean((0, 1))
                                                         - gold_out[:, 15:19].mean((0, 1))).max()),
                final_err_cross_fp32reduce=np.array(np.abs(maps_cross[:, 15:19].mean((0, 1))
                                                           - gold_cross[:, 15:19].mean((0, 1))).max()))


def run_dev_refbf16(name="full_depth_dev_refbf16.npz", size=1024):
    """The reference's own ModifiedFluxDiT + denoise in bf16 on the flux-dev case of run_dev_fp32 (yardstick)."""
    p = configs["flux-dev"]
    log = _logger("devref")
    model, sampling = _reference_model(p, log)
    inp = inputs(p, size, DEV_T, DEV_C)
    bf = torch.bfloat16
    img = O.patchify(inp["latent"]).to(bf)
    L = img.shape[1]
    ts = sampling.get_schedule(DEV_OF, L, shift=True)[:DEV_STEPS + 1]
    with torch.no_grad():
        x, _, d = sampling.denoise(model, img=img, img_ids=inp["img_ids"], txt=inp["txt"].to(bf),
                                   txt_ids=inp["txt_ids"], vec=inp["vec"].to(bf), timesteps=ts, guidance=DEV_GUIDANCE,
                                   concepts=inp["concepts"].to(bf), concept_ids=inp["concept_ids"],
                                   concept_vec=inp["concept_vec"].to(bf))
    log("reference bf16 denoise done")
    arrays = _yardstick(d, np.load("/tmp/dev_fp32_all_out.npy"), np.load("/tmp/dev_fp32_all_cross.npy"), DEV_STEPS,
                        p.depth, DEV_C, L)
    arrays["final_img_rows"] = x[0, SAMPLE_ROWS].float().numpy()
    np.savez_compressed(os.path.join(GOLD, name), **arrays)
    log(f"wrote {name}")
    print("dev: reference bf16 vs fp32 oracle, output-space max-abs per (step, layer 15..18):\n", arrays["err_out"][:, 15:19])
    print("cross-space:\n", arrays["err_cross"][:, 15:19])
    print({k: float(v) for k, v in arrays.items() if v.ndim == 0})


def encode_case(p, size=1024, T=256):
    """Inputs of the full-size encode_image case: the seeded latent is the 'encoded image'; the noise is drawn on the
    HOST generator (the GPU test must form the same tensor; get_noise's device stream differs per platform) and mixed
    as add_noise_to_image does (concept_attention/segmentation.py:85-113) at schedule[2] of 4 schnell steps (t = 0.5),
    in fp32 with ONE rounding to bf16 -- the tensor conceptattention_amd.pipeline._encode_maps forms."""
    inp = inputs(p, size, T, ENC_C)
    noise = torch.randn(inp["latent"].shape, generator=torch.Generator().manual_seed(ENC_NOISE_SEED)).bfloat16()
    t = O.get_schedule(ENC_STEPS, (size // 16) ** 2, shift=False)[ENC_NOISE_TIMESTEP]
    x = (t * noise.float() + (1.0 - t) * inp["latent"]).bfloat16().float()
    return inp, noise, x, t


def run_encode_fp32(name="encode_full.npz"):
    """encode_image (concept_attention_pipeline.py:204-357) at full size: ONE forward of the 19 double blocks
    (stop_after_multimodal_attentions) on the noised latent, y = concept_vec = 0 (:291), two concepts."""
    p = configs["flux-schnell"]
    log = _logger("encode")
    inp, noise, x, t = encode_case(p)
    sd = LazySD(p)
    img = O.patchify(x)
    L = img.shape[1]
    out = np.zeros((1, p.depth, ENC_C, L), np.float32)
    cross = np.zeros_like(out)

    def on_layer(i, od):
        ho, hc = layer_maps(od)
        out[0, i], cross[0, i] = ho.numpy(), hc.numpy()
    oracle_step(sd, p, img, inp, t, log, on_layer, y=inp["concept_vec"], stop_after=True)
    arrays = dict(t=np.array(t), noise_seed=np.array(ENC_NOISE_SEED), noise_checksum=np.array(noise.float().sum().item()),
                  x_rows=x[0, :, ::16, ::16].numpy(), out_layers=out[0, 15:19], cross_layers=cross[0, 15:19],
                  out_early=out[0, [0, 9]], cross_early=cross[0, [0, 9]],
                  final_out=out[0, 15:19].mean(0), final_cross=cross[0, 15:19].mean(0),
                  weight_seed=np.array(SEED_W), input_seed=np.array(SEED_IN))
    np.savez_compressed(os.path.join(GOLD, name), **arrays)
    np.save("/tmp/encode_fp32_all_out.npy", out)
    np.save("/tmp/encode_fp32_all_cross.npy", cross)
    log(f"wrote {name}")


def run_encode_refbf16(name="encode_full_refbf16.npz"):
    """The reference's model called the way its encode_image calls it (concept_attention_pipeline.py:284-297) in
    bf16 on the case of run_encode_fp32 (yardstick)."""
    p = configs["flux-schnell"]
    log = _logger("encoderef")
    model, _ = _reference_model(p, log)
    inp, noise, x, t = encode_case(p)
    bf = torch.bfloat16
    with torch.no_grad():
        _, d = model(img=O.patchify(x).to(bf), img_ids=inp["img_ids"], txt=inp["txt"].to(bf), txt_ids=inp["txt_ids"],
                     concepts=inp["concepts"].to(bf), concept_ids=inp["concept_ids"],
                     concept_vec=inp["concept_vec"].to(bf), y=inp["concept_vec"].to(bf),
                     timesteps=torch.full((1,), t, dtype=bf), guidance=torch.zeros(1, dtype=bf),
                     stop_after_multimodal_attentions=True, joint_attention_kwargs=None)
    log("reference bf16 forward done")
    arrays = _yardstick({k: v[None] for k, v in d.items()}, np.load("/tmp/encode_fp32_all_out.npy"),
                        np.load("/tmp/encode_fp32_all_cross.npy"), 1, p.depth, ENC_C, x.shape[-1] * x.shape[-2] // 4)
    np.savez_compressed(os.path.join(GOLD, name), **arrays)
    log(f"wrote {name}")
    print("encode: reference bf16 vs fp32 oracle, layers 15..18 out / cross:", arrays["err_out"][:, 15:19],
          arrays["err_cross"][:, 15:19])
    print({k: float(v) for k, v in arrays.items() if v.ndim == 0})


if __name__ == "__main__":
    torch.set_num_threads(8)
    for what in sys.argv[1:] or ["fp32", "refbf16", "cfg1"]:
        {"fp32": run_fp32, "refbf16": run_refbf16, "cfg1": run_cfg1, "dev": run_dev_fp32, "devref": run_dev_refbf16,
         "encode": run_encode_fp32, "encoderef": run_encode_refbf16}[what]()
