"""Where the end-to-end heat-map error of the bf16 path comes from (CPU study, build container or GPU box host).

Emulates the HIP path's storage roundings on top of the fp32 oracle -- same seeded weights / inputs as
oracle/make_full_goldens.py -- with each rounding switchable, runs the 19 double blocks of step 0 at full size
for several variants side by side (weights are generated once per block) and reports, per variant, the
max-abs difference of the per-layer output-space / cross-space maps from the fp32 maps.

    python -B tests/tools/error_budget.py            # needs /tmp/full_fp32_all_{out,cross}.npy or the golden

Roundings of the HIP path: residual stream X (bf16 | fp32), XM = LayerNorm-modulate output
(GEMM operand), q/k/v after norm+RoPE, attention output rows (image rows bf16, concept rows fp32 for the maps),
softmax probabilities P (bf16 operand of P V), MLP hidden.
"""
import json
import os
import sys
import time
from dataclasses import dataclass

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
import math

import numpy as np
import torch
import torch.nn.functional as F

from conceptattention_amd.params import configs
from oracle import flux_oracle as O
from oracle.make_full_goldens import LazySD, inputs, layer_maps


@dataclass
class Rnd:
    name: str
    resid: bool = True      # residual stream stored in bf16
    xm: bool = True         # LN-modulate output in bf16
    qkv: bool = True        # q, k (post norm+rope) and v in bf16
    p: bool = True          # softmax numerators in bf16 before P V
    attn: bool = True       # attention output (image/text rows) in bf16
    hid: bool = True        # MLP hidden in bf16
    lin_out: bool = False   # projection outputs rounded before the gated add (the HIP epilogue does NOT: fp32 acc)
    qpre_img: bool = True   # stored post-QKNorm pre-RoPE image q (cross-space image vectors) in bf16
    qpre_con: bool = True   # the same for the C concept rows
    xm_q: bool = True       # the LN-modulate output AS SEEN BY the q third of the qkv projection (a hi+lo bf16 split of
                            # that operand would make it effectively fp32)
    qk_mode: str = ""       # "" = as `qkv` says; else the storage type of the rotated q / k: "bf16" | "fp16" | "fp32"
    v_mode: str = ""        # the same for v ("split" = hi + lo bf16 planes, ~16 mantissa bits)


def r(x, on):
    return x.bfloat16().float() if on else x


def rm(x, mode, default_on):
    if mode == "":
        return r(x, default_on)
    if mode == "fp16":
        return x.half().float()
    if mode == "split":
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    return r(x, mode == "bf16")


def sdpa_r(q, k, v, cfg):
    s = (q @ k.transpose(-2, -1)) / math.sqrt(q.shape[-1])
    m = s.amax(-1, keepdim=True)
    p = torch.exp(s - m)
    l = p.sum(-1, keepdim=True)
    return (r(p, cfg.p) @ v) / l


def double_block(sd, pfx, nh, img, txt, vec, rope_ti, con, cvec, rope_ci, cfg):
    T, C = txt.shape[1], con.shape[1]
    im, tm, cm = (O.modulation(sd, pfx + "img_mod", vec, 6), O.modulation(sd, pfx + "txt_mod", vec, 6),
                  O.modulation(sd, pfx + "txt_mod", cvec, 6))

    def pre(x, mod, s):
        xm32 = (1 + mod[1]) * O.layer_norm(x) + mod[0]
        xm = r(xm32, cfg.xm)
        q, k, v = O._split_heads(O.linear(sd, pfx + s + "_attn.qkv", xm), nh)
        if cfg.xm and not cfg.xm_q:
            q = O._split_heads(O.linear(sd, pfx + s + "_attn.qkv", xm32), nh)[0]
        return (O.rms_norm(q, sd[pfx + s + "_attn.norm.query_norm.scale"]),
                O.rms_norm(k, sd[pfx + s + "_attn.norm.key_norm.scale"]), rm(v, cfg.v_mode, cfg.qkv))
    iq, ik, iv = pre(img, im, "img")
    tq, tk, tv = pre(txt, tm, "txt")
    cq, ck, cv = pre(con, cm, "txt")
    q = rm(O.apply_rope(torch.cat((tq, iq), 2), *rope_ti), cfg.qk_mode, cfg.qkv)
    k = rm(O.apply_rope(torch.cat((tk, ik), 2), *rope_ti), cfg.qk_mode, cfg.qkv)
    attn = sdpa_r(q, k, torch.cat((tv, iv), 2), cfg)
    t_attn, i_attn = r(attn[:, :, :T], cfg.attn), r(attn[:, :, T:], cfg.attn)
    qc = rm(O.apply_rope(torch.cat((cq, iq), 2), *rope_ci), cfg.qk_mode, cfg.qkv)
    kc = rm(O.apply_rope(torch.cat((ck, ik), 2), *rope_ci), cfg.qk_mode, cfg.qkv)
    c_attn32 = sdpa_r(qc[:, :, :C], kc, torch.cat((cv, iv), 2), cfg)   # fp32 copy feeds the maps
    t_attn, i_attn, c_attn32 = map(O._merge_heads, (t_attn, i_attn, c_attn32))
    d = {"output_space_concept_vectors": c_attn32, "output_space_image_vectors": i_attn,
         "cross_attention_concept_vectors": r(cq, cfg.qkv and cfg.qpre_con),
         "cross_attention_image_vectors": r(iq, cfg.qkv and cfg.qpre_img)}

    def post(x, a, mod, s):
        x = r(x + mod[2] * r(O.linear(sd, pfx + s + "_attn.proj", a), cfg.lin_out), cfg.resid)
        h = r((1 + mod[4]) * O.layer_norm(x) + mod[3], cfg.xm)
        h = r(F.gelu(O.linear(sd, pfx + s + "_mlp.0", h), approximate="tanh"), cfg.hid)
        return r(x + mod[5] * r(O.linear(sd, pfx + s + "_mlp.2", h), cfg.lin_out), cfg.resid)
    return (post(img, i_attn, im, "img"), post(txt, t_attn, tm, "txt"), post(con, r(c_attn32, cfg.attn), cm, "txt"), d)


def main():
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    p = configs["flux-schnell"]
    inp = inputs(p, 1024, 256, 4)
    sd = LazySD(p, keep=16)
    img0 = O.patchify(inp["latent"])
    if os.path.exists("/tmp/full_fp32_all_out.npy"):
        gold_out, gold_cross = np.load("/tmp/full_fp32_all_out.npy")[0], np.load("/tmp/full_fp32_all_cross.npy")[0]
    else:
        g = np.load(os.path.join(ROOT, "tests", "golden", "full_depth_schnell.npz"))
        gold_out, gold_cross = g["out_step0"], g["cross_step0"]
    variants = [Rnd("hip_today(all bf16 storage)"),
                Rnd("fp32_residual", resid=False),
                Rnd("fp32_residual+fp32_xm", resid=False, xm=False),
                Rnd("fp32_residual+fp32_p", resid=False, p=False),
                Rnd("only_residual_bf16", xm=False, qkv=False, p=False, attn=False, hid=False),
                Rnd("fp32_residual+fp32_qpre_con", resid=False, qpre_con=False),
                Rnd("fp32_residual+fp32_qpre", resid=False, qpre_con=False, qpre_img=False),
                Rnd("fp32_residual+fp32_qpre+split_xm_q", resid=False, qpre_con=False, qpre_img=False, xm_q=False)]
    # round 4: the OUTPUT space, starting from what the HIP path stores today (fp32 residual, fp32 attention rows in the
    # captured layers -- emulated for every layer here --, fp32 cross-space vectors); one rounding lifted at a time
    now = dict(resid=False, attn=False, qpre_con=False, qpre_img=False, xm_q=False)
    out_space = [Rnd("r4_today", **now),
                 Rnd("r4_today+fp32_p", **now, p=False),
                 Rnd("r4_today+fp32_qkv", **now, qkv=False),
                 Rnd("r4_today+fp32_xm", **dict(now, xm=False)),
                 Rnd("r4_today+fp32_xm+fp32_qkv", **dict(now, xm=False), qkv=False),
                 Rnd("r4_today+fp32_hid", **now, hid=False),
                 Rnd("r4_today+fp32_xm+fp32_qkv+fp32_p", **dict(now, xm=False), qkv=False, p=False)]
    out_space2 = [Rnd("r4_today", **now),
                  Rnd("r4_fp32_qk", **now, qk_mode="fp32"),
                  Rnd("r4_fp32_v", **now, v_mode="fp32"),
                  Rnd("r4_fp16_qk", **now, qk_mode="fp16"),
                  Rnd("r4_fp16_qk+split_v", **now, qk_mode="fp16", v_mode="split"),
                  Rnd("r4_fp16_qk+fp16_v", **now, qk_mode="fp16", v_mode="fp16"),
                  Rnd("r4_split_v", **now, v_mode="split")]
    if len(sys.argv) > 1 and sys.argv[1] == "--out-space":
        variants, sys.argv = out_space, sys.argv[:1] + sys.argv[2:]
    if len(sys.argv) > 1 and sys.argv[1] == "--out-space2":
        variants, sys.argv = out_space2, sys.argv[:1] + sys.argv[2:]
    if len(sys.argv) > 1:
        variants = [v for v in variants if v.name.split("(")[0] in sys.argv[1:]]
    nh = p.num_heads
    temb = O.timestep_embedding(torch.tensor([1.0]))
    vec = O.mlp_embedder(sd, "time_in", temb) + O.mlp_embedder(sd, "vector_in", inp["vec"])
    cvec = O.mlp_embedder(sd, "time_in", temb) + O.mlp_embedder(sd, "vector_in", inp["concept_vec"])
    rope_ti = O.rope_cos_sin(torch.cat((inp["txt_ids"][0], inp["img_ids"][0])), p.axes_dim, p.theta)
    rope_ci = O.rope_cos_sin(torch.cat((inp["concept_ids"][0], inp["img_ids"][0])), p.axes_dim, p.theta)
    state = {}
    for v in variants:
        state[v.name] = [r(O.linear(sd, "img_in", img0), v.resid), r(O.linear(sd, "txt_in", inp["txt"]), v.resid),
                         r(O.linear(sd, "txt_in", inp["concepts"]), v.resid)]
    res = {v.name: {"out": [], "cross": []} for v in variants}
    mean_cross = {v.name: 0.0 for v in variants}   # mean map over layers 15..18 (what generate_image averages, one step)
    t0 = time.time()
    for i in range(p.depth):
        for v in variants:
            x_img, x_txt, x_con = state[v.name]
            x_img, x_txt, x_con, d = double_block(sd, f"double_blocks.{i}.", nh, x_img, x_txt, vec, rope_ti, x_con,
                                                  cvec, rope_ci, v)
            state[v.name] = [x_img, x_txt, x_con]
            ho, hc = layer_maps(d)
            res[v.name]["out"].append(float(np.abs(ho.numpy() - gold_out[i]).max()))
            res[v.name]["cross"].append(float(np.abs(hc.numpy() - gold_cross[i]).max()))
            if 15 <= i < 19:
                mean_cross[v.name] = mean_cross[v.name] + (hc.numpy() - gold_cross[i]) / 4.0
        print(f"[{time.time() - t0:5.0f}s] layer {i}: " +
              "  ".join(f"{v.name.split('(')[0]} {res[v.name]['out'][-1]:.2e}/{res[v.name]['cross'][-1]:.2e}"
                        for v in variants), flush=True)
    out = {"note": "max-abs difference of per-layer maps (output space / cross space) from the fp32 oracle, step 0 "
                   "(t=1.0), full size, CPU emulation of the storage roundings", "variants": res,
           "layers_15_18_out": {k: max(v["out"][15:19]) for k, v in res.items()},
           "layers_15_18_cross": {k: max(v["cross"][15:19]) for k, v in res.items()},
           "mean_map_layers_15_18_cross": {k: float(np.abs(v).max()) for k, v in mean_cross.items()}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", os.environ.get("CA_BUDGET_OUT", "error_budget.json")), "w"),
              indent=1)
    print(json.dumps(out["layers_15_18_out"], indent=1))
    print(json.dumps(out["layers_15_18_cross"], indent=1))
    print(json.dumps(out["mean_map_layers_15_18_cross"], indent=1))


if __name__ == "__main__":
    main()
