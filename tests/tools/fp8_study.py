"""Tolerance study of the fp8 (e4m3 projections) mode against the bf16 path of the SAME kernels, at the full
BASELINE.json configs[1] geometry (there is no reference or oracle for fp8: SURVEY.md §8f-2).

    python tests/tools/fp8_study.py            # -> gpurun_out/fp8_study.json

Reports, for one generate_image-equivalent call (4 steps, C=4, layers 15-18) and for the per-layer table of
one encode-style forward: max-abs and mean-abs difference of the heat maps, the fraction of patches whose
arg-max concept changes, and the relative error of the denoised latent.  Also checks the bf16 path against
the fp32 oracle on ONE double block so the two deviations can be read side by side."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from conceptattention_amd.params import T5_TOKENS, configs
from conceptattention_amd.pipeline import ConceptAttentionFluxPipeline
from conceptattention_amd.weights import synthetic_inputs

dev = "cuda:0"
name = "flux-schnell"
p = configs[name]
C, T, size = 4, T5_TOKENS[name], 1024
pipe = ConceptAttentionFluxPipeline(name, device=dev, weights="synthetic", weight_seed=0)
res = {"config": f"{name} {size}x{size}, C={C}, T={T}, 4 steps, layers 15-18, synthetic weights/inputs", "items": []}
for seed in (1000, 1001):
    inp = synthetic_inputs(p, size, size, T, C, seed=seed, device="cpu", dtype=torch.bfloat16)
    x = {k: inp[k].to(dev) for k in ("latent", "txt", "vec", "concepts")}
    out = {}
    for prec in ("bf16", "fp8", "fp8_keep15_18"):
        pipe.fp8_keep_heatmap_layers = "keep" in prec   # else generate_on_device re-applies the default keep set
        for m in pipe._replicas:
            m.set_precision(prec.split("_")[0], keep_bf16_layers=range(15, 19) if "keep" in prec else ())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img, hm, cm = pipe.generate_on_device(x["latent"], x["txt"], x["vec"], x["concepts"],
                                              layer_indices=list(range(15, 19)), num_inference_steps=4, guidance=0.0)
        torch.cuda.synchronize()
        tab, ctab = pipe.layer_noise_sweep_on_device(x["latent"], x["txt"], x["vec"], x["concepts"], [2], num_steps=4,
                                                     seed=seed)
        out[prec] = dict(img=img.float().cpu(), hm=hm[0].cpu(), cm=cm[0].cpu(), tab=tab[0].cpu(), ctab=ctab[0].cpu(),
                         sec=time.perf_counter() - t0)
    a = out["bf16"]
    for mode in ("fp8", "fp8_keep15_18"):
        b = out[mode]
        item = {"seed": seed, "mode": mode}
        for k, label in (("hm", "output_space_heatmaps"), ("cm", "cross_attention_maps")):
            d = (a[k] - b[k]).abs()
            item[label] = {"max_abs": d.max().item(), "mean_abs": d.mean().item(),
                           "argmax_changed_frac": (a[k].argmax(0) != b[k].argmax(0)).float().mean().item(),
                           "value_range": [a[k].min().item(), a[k].max().item()]}
        d = (a["tab"] - b["tab"]).abs()   # [19 layers, C, side, side]
        item["per_layer_single_forward_max_abs"] = [round(v, 5) for v in d.amax(dim=(1, 2, 3)).tolist()]
        item["latent_rel_rms"] = ((a["img"] - b["img"]).norm() / a["img"].norm()).item()
        item["seconds"] = {"bf16": a["sec"], mode: b["sec"]}
        res["items"].append(item)
        print(json.dumps(item), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "fp8_study.json"), "w"), indent=1)
