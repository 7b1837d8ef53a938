"""Where is a forward host-bound?  For one 5-item forward of a workload, the host's enqueue clock and the GPU's event
clock at the same points (forward start, after every double block, forward end).  If the GPU reaches a point right
after the host enqueued it, the host is the limit there; if the host is far ahead, the GPU is.

    python tools/host_gpu_timeline.py --workload sweep|encode|generate [--batch 5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd.params import configs
from conceptattention_amd.pipeline import ConceptAttentionFluxPipeline
from conceptattention_amd.weights import synthetic_inputs

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sweep")
ap.add_argument("--batch", type=int, default=5)
ap.add_argument("--out", default=None)
ap.add_argument("--repeat", type=int, default=0, help="also time this many back-to-back forwards (events per forward)")
a = ap.parse_args()
dev = "cuda:0"
p = configs["flux-schnell"]
C = 2 if a.workload == "encode" else 4
pipe = ConceptAttentionFluxPipeline("flux-schnell", device=dev)
items = []
for j in range(a.batch):
    inp = synthetic_inputs(p, 1024, 1024, 256, C, seed=1000 + j, device="cpu", dtype=torch.bfloat16)
    items.append({k: inp[k].to(dev) for k in ("latent", "txt", "vec", "concepts")})


def run():
    if a.workload == "sweep":
        x = items[0]
        pipe.layer_noise_sweep_on_device(x["latent"], x["txt"], x["vec"], x["concepts"], list(range(a.batch)),
                                         num_steps=50, batch=a.batch)
    elif a.workload == "encode":
        pipe.encode_many_on_device(items, batch=a.batch)
    else:
        pipe.generate_many_on_device(items, batch=a.batch)


run()
torch.cuda.synchronize()
m = pipe.model
marks = []   # (label, host time, event)


def mark(label):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((label, time.perf_counter(), e))


orig_d, orig_s = m._double_block, m._single_block


def dbl(i, *k, **kw):
    orig_d(i, *k, **kw)
    mark(f"d{i}")


def sgl(i, *k, **kw):
    orig_s(i, *k, **kw)
    if i % 8 == 7:
        mark(f"s{i}")


m._double_block, m._single_block = dbl, sgl
torch.cuda.synchronize()
mark("start")
run()
mark("end")
t_enq = time.perf_counter()
torch.cuda.synchronize()
t_done = time.perf_counter()
m._double_block, m._single_block = orig_d, orig_s
h0, e0 = marks[0][1], marks[0][2]
rows = []
prev_h = prev_g = 0.0
for label, h, e in marks:
    hh, gg = (h - h0) * 1e3, e0.elapsed_time(e)
    rows.append({"at": label, "host_ms": round(hh, 2), "gpu_ms": round(gg, 2), "host_d": round(hh - prev_h, 2),
                 "gpu_d": round(gg - prev_g, 2), "gpu_behind_host_ms": round(gg - hh, 2)})
    prev_h, prev_g = hh, gg
for r in rows:
    print(r)
doc = {"workload": a.workload, "batch": a.batch, "host_enqueue_ms": (t_enq - h0) * 1e3, "total_ms": (t_done - h0) * 1e3,
       "marks": rows}
print(json.dumps({k: v for k, v in doc.items() if k != "marks"}))
if a.repeat:
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.repeat + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(a.repeat):
        run()
        evs[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    per = [round(evs[i].elapsed_time(evs[i + 1]), 2) for i in range(a.repeat)]
    doc["repeat_ms"] = per
    doc["repeat_wall_ms"] = wall
    print("back-to-back forwards, GPU ms each:", per, "wall", round(wall, 1))
if a.out:
    json.dump(doc, open(a.out, "w"), indent=1)
