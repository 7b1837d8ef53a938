"""Images/s of the encode path (19 double blocks per image, BASELINE.json configs[3] shape of work: 2 concepts per
image) on one GPU, one vs two streams (development aid; bench.py measures the headline generate path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd.params import configs
from conceptattention_amd.pipeline import ConceptAttentionFluxPipeline
from conceptattention_amd.weights import synthetic_inputs

dev = "cuda:0"
p = configs["flux-schnell"]
pipe = ConceptAttentionFluxPipeline("flux-schnell", device=dev)
items = []
for j in range(20):
    inp = synthetic_inputs(p, 1024, 1024, 256, 2, seed=50 + j, device="cpu", dtype=torch.bfloat16)
    items.append({k: inp[k].to(dev) for k in ("latent", "txt", "vec", "concepts")})
for ns, nb in ((1, 1), (2, 1), (1, 5), (1, 1), (2, 1), (1, 5)):
    pipe.encode_many_on_device(items[:max(2, nb)], n_streams=ns, batch=nb, layer_indices=list(range(19)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.encode_many_on_device(items, n_streams=ns, batch=nb, layer_indices=list(range(19)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams {ns} images per forward {nb}: {len(items)/dt:.1f} images/s ({dt/len(items)*1e3:.1f} ms per image, "
          "19 layers x 2 spaces of maps)", flush=True)
