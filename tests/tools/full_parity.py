"""One-off end-to-end parity run at FULL size (flux-schnell geometry, 1024x1024, C=4): ONE complete DiT
step (19 double + 38 single blocks) on the HIP path vs the fp32 CPU oracle with identical seeded
weights, NOT teacher-forced, so it shows how the bf16 error grows with depth.  Writes
gpurun_out/full_parity.json (copied to profiles/).  Takes ~10 min of host CPU for the oracle."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from conceptattention_amd.flux_dit import HipFluxDiT, HeatmapRequest, DICT_KEYS
from conceptattention_amd.params import configs
from conceptattention_amd.weights import state_dict_spec, synth_tensor, synthetic_inputs
from oracle import flux_oracle as O

MODEL = sys.argv[1] if len(sys.argv) > 1 else "flux-schnell"   # or flux-dev (T=512, C=8, guidance embedding)
p = configs[MODEL]
T_TXT, C = (512, 8) if MODEL == "flux-dev" else (256, 4)
GUIDANCE = 3.5 if p.guidance_embed else None
dev = "cuda:0"
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
spec = dict(state_dict_spec(p))
fan = {n[:-7]: s[1] for n, s in spec.items() if n.endswith(".weight")}


class LazySD(dict):
    """Generates each (bf16-rounded) tensor on access; keeps only the last few."""
    def __init__(self):
        self.cache = {}
    def _make(self, name):
        return synth_tensor(name, spec[name], fan.get(name.rsplit(".", 1)[0], 1), seed=0).bfloat16().float()
    def __getitem__(self, name):
        if name not in self.cache:
            if len(self.cache) > 6:
                self.cache.pop(next(iter(self.cache)))
            self.cache[name] = self._make(name)
        return self.cache[name]
    def get(self, name, default=None):
        return self[name] if name in spec else default

t0 = time.time()
m = HipFluxDiT(p, dev)
for name in spec:  # identical weights on the device
    m.weights.tensors[name].copy_(synth_tensor(name, spec[name], fan.get(name.rsplit(".", 1)[0], 1), seed=0))
print(f"weights on device {time.time()-t0:.0f}s", flush=True)
inp = {k: (v.bfloat16().float() if v.is_floating_point() else v)
       for k, v in synthetic_inputs(p, 1024, 1024, T_TXT, C, seed=5).items()}
img = O.patchify(inp["latent"])
L = 4096
tval = 1.0
# HIP: all 19 layers' maps individually + the default 15..18 mean
d = {k: v.to(dev) for k, v in inp.items()}
per_layer = []
for layer in range(19):
    per_layer.append(HeatmapRequest((layer,), 1.0, torch.zeros(C, L, device=dev), torch.zeros(C, L, device=dev)))
# one forward per request would be 19 forwards; instead capture vectors for all layers once
pred, dd = m(img=img.to(dev), img_ids=d["img_ids"], txt=d["txt"], txt_ids=d["txt_ids"], concepts=d["concepts"],
             concept_ids=d["concept_ids"], concept_vec=d["concept_vec"], y=d["vec"],
             timesteps=torch.tensor([tval], device=dev), return_vectors=True,
             guidance=None if GUIDANCE is None else torch.tensor([GUIDANCE], device=dev))
torch.cuda.synchronize()
hip = {k: v.float().cpu() for k, v in dd.items()}
pred = pred.float().cpu()
print(f"hip forward done {time.time()-t0:.0f}s", flush=True)

# oracle, block by block with progress output
sd = LazySD()
nh = p.num_heads
x_img = O.linear(sd, "img_in", img)
temb = O.timestep_embedding(torch.tensor([tval]))
vec = O.mlp_embedder(sd, "time_in", temb) + O.mlp_embedder(sd, "vector_in", inp["vec"])
cvec = O.mlp_embedder(sd, "time_in", temb) + O.mlp_embedder(sd, "vector_in", inp["concept_vec"])
if GUIDANCE is not None:  # modified_flux_dit.py:100-103,113-116: both conditioning vectors get the guidance embedding
    gemb = O.mlp_embedder(sd, "guidance_in", O.timestep_embedding(torch.tensor([GUIDANCE])))
    vec, cvec = vec + gemb, cvec + gemb
x_txt = O.linear(sd, "txt_in", inp["txt"])
x_con = O.linear(sd, "txt_in", inp["concepts"])
rope_ti = O.rope_cos_sin(torch.cat((inp["txt_ids"][0], inp["img_ids"][0])), p.axes_dim, p.theta)
rope_ci = O.rope_cos_sin(torch.cat((inp["concept_ids"][0], inp["img_ids"][0])), p.axes_dim, p.theta)
res = {"layers": []}
for i in range(p.depth):
    x_img, x_txt, x_con, od = O.double_block(sd, f"double_blocks.{i}.", nh, x_img, x_txt, vec, rope_ti, x_con, cvec, rope_ci)
    st = {k: v[None, None] for k, v in od.items()}
    ho = O.compute_heatmaps(st["output_space_image_vectors"], st["output_space_concept_vectors"], [0], [0])[0]
    hc = O.compute_heatmaps(st["cross_attention_image_vectors"], st["cross_attention_concept_vectors"], [0], [0])[0]
    hs = {k: hip[k][i][None, None] for k in DICT_KEYS}
    hho = O.compute_heatmaps(hs["output_space_image_vectors"], hs["output_space_concept_vectors"], [0], [0])[0]
    hhc = O.compute_heatmaps(hs["cross_attention_image_vectors"], hs["cross_attention_concept_vectors"], [0], [0])[0]
    e_attn = (hip["output_space_image_vectors"][i] - od["output_space_image_vectors"]).abs().max().item()
    rec = {"layer": i, "heatmap_out_maxabs": (hho - ho).abs().max().item(), "heatmap_cross_maxabs": (hhc - hc).abs().max().item(),
           "cross_argmax_agree": (hhc.argmax(0) == hc.argmax(0)).float().mean().item(),
           "img_attn_maxabs": e_attn, "img_attn_ref_absmax": od["output_space_image_vectors"].abs().max().item()}
    res["layers"].append(rec)
    print(f"[{time.time()-t0:.0f}s] double {i}: {rec}", flush=True)
x = torch.cat((x_txt, x_img), 1)
for i in range(p.depth_single_blocks):
    x = O.single_block(sd, f"single_blocks.{i}.", nh, x, vec, rope_ti)
    if i % 6 == 5:
        print(f"[{time.time()-t0:.0f}s] single {i}", flush=True)
x = x[:, T_TXT:]
shift, scale = O.linear(sd, "final_layer.adaLN_modulation.1", torch.nn.functional.silu(vec)).chunk(2, dim=1)
x = (1 + scale[:, None, :]) * O.layer_norm(x) + shift[:, None, :]
pred_o = O.linear(sd, "final_layer.linear", x)
e = (pred - pred_o).abs()
res["pred"] = {"maxabs": e.max().item(), "rms_err": e.pow(2).mean().sqrt().item(), "ref_rms": pred_o.pow(2).mean().sqrt().item(),
               "ref_absmax": pred_o.abs().max().item()}
res["default_layers_15_18_heatmap_out_maxabs"] = max(r["heatmap_out_maxabs"] for r in res["layers"][15:19])
res["note"] = f"one full {MODEL}-geometry DiT step, t=1.0, L=4096 T={T_TXT} C={C}, guidance={GUIDANCE}, random-init weights seed 0, HIP bf16 vs fp32 oracle, not teacher-forced; heat maps recomputed in fp32 from each path's captured bf16/fp32 vectors"
print(json.dumps(res["pred"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"full_parity_{MODEL}.json"), "w"), indent=1)
