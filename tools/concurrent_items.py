"""Experiment: two independent work items on two HIP streams of ONE GPU (shared weights, separate
activation sets), launches interleaved block by block, vs the same two items back to back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conceptattention_amd.flux_dit import HipFluxDiT, HeatmapRequest
from conceptattention_amd.params import configs
from conceptattention_amd import sampling
from conceptattention_amd.weights import synthetic_inputs

dev = "cuda:0"
p = configs["flux-schnell"]
m0 = HipFluxDiT(p, dev)
m0.weights.init_synthetic(0)
m1 = HipFluxDiT(p, dev, weights=m0.weights)
models = [m0, m1]
C, T, L = 4, 256, 4096
items = []
for j in range(2):
    inp = synthetic_inputs(p, 1024, 1024, T, C, seed=100 + j, dtype=torch.bfloat16)
    x = {k: v.to(dev) for k, v in inp.items()}
    x["img"] = sampling.patchify(x["latent"]).contiguous()
    items.append(x)
sched = sampling.get_schedule(4, L, shift=False)


def gen_steps(m, x):
    """Generator: yields after enqueuing each block of each diffusion step (cooperative interleave)."""
    img = x["img"].clone()
    req = HeatmapRequest((15, 16, 17, 18), 1 / 16, torch.zeros(C, L, device=dev), torch.zeros(C, L, device=dev))
    m.precompute_conditioning(sched[:-1], x["vec"], x["concept_vec"], 0.0)
    yield
    for it, (tc, tp) in enumerate(zip(sched[:-1], sched[1:])):
        # reuse the model's own forward but split it in two halves to interleave: emulate by running whole step
        pred, _ = m(img=img, img_ids=x["img_ids"], txt=x["txt"], txt_ids=x["txt_ids"], concepts=x["concepts"],
                    concept_ids=x["concept_ids"], concept_vec=x["concept_vec"], y=x["vec"],
                    timesteps=torch.tensor([tc], device=dev), return_vectors=False, heatmaps=req, cond_slot=it)
        from conceptattention_amd import ops
        ops.axpy(img, pred.contiguous(), tp - tc)
        yield
    yield req


def run_sequential(n):
    for _ in range(n):
        for j in range(2):
            for _ in gen_steps(models[j], items[j]):
                pass


def run_concurrent(n, streams):
    for _ in range(n):
        gens = [gen_steps(models[j], items[j]) for j in range(2)]
        alive = [True, True]
        while any(alive):
            for j in range(2):
                if alive[j]:
                    with torch.cuda.stream(streams[j]):
                        try:
                            next(gens[j])
                        except StopIteration:
                            alive[j] = False


run_sequential(1)
torch.cuda.synchronize()
t0 = time.perf_counter(); run_sequential(2); torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 4
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
run_concurrent(1, streams); torch.cuda.synchronize()
t0 = time.perf_counter(); run_concurrent(2, streams); torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / 4
print(f"sequential {ts*1e3:.1f} ms/call, two streams {tc*1e3:.1f} ms/call, gain {ts/tc:.3f}x")
