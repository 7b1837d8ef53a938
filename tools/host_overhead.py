"""How long does the host need to ENQUEUE one generate_image-equivalent call (no synchronisation)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd.pipeline import ConceptAttentionFluxPipeline
from conceptattention_amd.params import configs
from conceptattention_amd.weights import synthetic_inputs
dev = "cuda:0"
p = configs["flux-schnell"]
pipe = ConceptAttentionFluxPipeline("flux-schnell", device=dev, weights="synthetic")
inp = synthetic_inputs(p, 1024, 1024, 256, 4, seed=1, dtype=torch.bfloat16)
x = {k: inp[k].to(dev) for k in ("latent", "txt", "vec", "concepts")}
pipe.generate_on_device(x["latent"], x["txt"], x["vec"], x["concepts"]); torch.cuda.synchronize()
t0 = time.perf_counter()
pipe.generate_on_device(x["latent"], x["txt"], x["vec"], x["concepts"])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
