"""Does replaying one DiT forward from a captured HIP graph beat launching its ~330 kernels one by one?
(development aid; measures GPU time of an eager forward vs a graph replay of the same forward)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd import sampling
from conceptattention_amd.params import configs
from conceptattention_amd.pipeline import ConceptAttentionFluxPipeline
from conceptattention_amd.weights import synthetic_inputs

dev = "cuda:0"
p = configs["flux-schnell"]
pipe = ConceptAttentionFluxPipeline("flux-schnell", device=dev)
m = pipe.model
inp = synthetic_inputs(p, 1024, 1024, 256, 4, seed=1, device="cpu", dtype=torch.bfloat16)
x = {k: inp[k].to(dev) for k in ("latent", "txt", "vec", "concepts")}
con, con_ids, con_vec = sampling.concept_inputs(x["concepts"], x["vec"])
pi = sampling.prepare_from_embeddings(x["latent"], x["txt"], x["vec"])
m.precompute_conditioning([1.0], pi["vec"], con_vec, None)
kw = dict(img=pi["img"], img_ids=pi["img_ids"], txt=pi["txt"], txt_ids=pi["txt_ids"], concepts=con, concept_ids=con_ids,
          concept_vec=con_vec, y=pi["vec"], timesteps=torch.ones(1, device=dev), guidance=torch.zeros(1, device=dev),
          return_vectors=False, cond_slot=0)


def run():
    return m(**kw)[0]


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


eager = timed(run)
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    run()
torch.cuda.current_stream().wait_stream(st)
with torch.cuda.graph(g):
    out = run()
replay = timed(g.replay)
ref = run()
g.replay(); torch.cuda.synchronize()
print(f"eager forward {eager:.3f} ms, graph replay {replay:.3f} ms ({(eager/replay-1)*100:+.2f} %), "
      f"outputs equal: {torch.equal(out, ref)}")
