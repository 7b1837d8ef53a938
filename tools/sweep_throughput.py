"""Per-layer x per-noise-level sweep (BASELINE.json configs[4] shape of work: 50 noise levels, maps of all 19 double
blocks, 1024x1024, 4 concepts) on one GPU in bf16 and in fp8 mode (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd.params import configs
from conceptattention_amd.pipeline import ConceptAttentionFluxPipeline
from conceptattention_amd.weights import synthetic_inputs

dev = "cuda:0"
p = configs["flux-schnell"]
pipe = ConceptAttentionFluxPipeline("flux-schnell", device=dev)
inp = synthetic_inputs(p, 1024, 1024, 256, 4, seed=3, device="cpu", dtype=torch.bfloat16)
x = {k: inp[k].to(dev) for k in ("latent", "txt", "vec", "concepts")}
levels = list(range(50))
ref = None
for prec in ("bf16", "fp8", "bf16", "fp8"):
    pipe.model.set_precision(prec)
    pipe.layer_noise_sweep_on_device(x["latent"], x["txt"], x["vec"], x["concepts"], levels[:2], num_steps=50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, cross = pipe.layer_noise_sweep_on_device(x["latent"], x["txt"], x["vec"], x["concepts"], levels, num_steps=50)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    msg = f"{prec}: {dt:.3f} s for {len(levels)} levels x 19 layers x 2 spaces ({dt/len(levels)*1e3:.1f} ms per level)"
    if prec == "bf16":
        ref = out
    else:
        msg += f"; max-abs deviation of the output-space table from bf16: {(out - ref).abs().max().item():.3e}"
    print(msg, flush=True)
