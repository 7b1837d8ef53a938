"""Full-size, full-depth goldens (BASELINE.json configs[1] and configs[0] shapes).  TEST INFRASTRUCTURE ONLY.

Runs in the build container only (imports the reference from /root/reference for the bf16 yardstick):

    python -B oracle/make_full_goldens.py fp32        # ~20 min: the pinned fp32 oracle, 4 full-size steps
    python -B oracle/make_full_goldens.py refbf16     # ~6 min: the REFERENCE's own modules in bf16, same weights
    python -B oracle/make_full_goldens.py cfg1        # config 1 shape (256x256, C=1, 1 step) at H=3072

Writes tests/golden/full_depth_schnell.npz (fp32 maps), tests/golden/full_depth_refbf16.npz (how far the
reference's own bf16 run is from those maps: the yardstick of tests/test_full_depth_gpu.py) and
tests/golden/cfg1_full_hidden.npz.  Only numbers are stored: heat maps, sample rows, error statistics.

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


def oracle_step(sd, p, img, inp, t, log, on_layer):
    """One DiT step of the fp32 oracle, block by block (weights streamed); on_layer(i, dict) per double block."""
    nh = p.num_heads
    T = inp["txt"].shape[1]
    x_img = O.linear(sd, "img_in", img)
    temb = O.timestep_embedding(torch.tensor([t]))
    vec = O.mlp_embedder(sd, "time_in", temb) + O.mlp_embedder(sd, "vector_in", inp["vec"])
    cvec = O.mlp_embedder(sd, "time_in", temb) + O.mlp_embedder(sd, "vector_in", inp["concept_vec"])
    x_txt = O.linear(sd, "txt_in", inp["txt"])
    x_con = O.linear(sd, "txt_in", inp["concepts"])
    rope_ti = O.rope_cos_sin(torch.cat((inp["txt_ids"][0], inp["img_ids"][0])), p.axes_dim, p.theta)
    rope_ci = O.rope_cos_sin(torch.cat((inp["concept_ids"][0], inp["img_ids"][0])), p.axes_dim, p.theta)
    for i in range(p.depth):
        x_img, x_txt, x_con, od = O.double_block(sd, f"double_blocks.{i}.", nh, x_img, x_txt, vec, rope_ti, x_con,
                                                 cvec, rope_ci)
        on_layer(i, od)
        log(f"double {i}")
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


if __name__ == "__main__":
    torch.set_num_threads(8)
    for what in sys.argv[1:] or ["fp32", "refbf16", "cfg1"]:
        {"fp32": run_fp32, "refbf16": run_refbf16, "cfg1": run_cfg1}[what]()
