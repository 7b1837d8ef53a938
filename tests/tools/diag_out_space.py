"""Where a single output-space heat map's error enters the HIP path (full size, layer 0 of step 0; GPU box).

The CPU emulation of the storage roundings (tests/tools/error_budget.py --out-space2) puts a layer's output-space map
at 2.3-3.4e-4 from the fp32 oracle once q and k carry 11 mantissa bits; the HIP path measured 7.7e-4-1.05e-3.  This
tool stops the real forward at layer 0's attention, copies what the kernels produced (XM, QKV, the fp32 attention
rows) and re-derives the map from each stage with fp32 torch ops on the GPU, so the stage that adds the difference
shows:

    A   the map as the path returns it (HIP heat-map kernels on the fp32 attention rows)
    A2  the same reduction in torch from the HIP attention rows                     (isolates the heat-map kernels)
    B   exact fp32 attention from the HIP q / k / v                                 (isolates the attention kernel)
    C2  fp32 projection + norm + RoPE from the HIP XM, q / k rounded to half, v to bf16, exact attention
                                                                                    (isolates the GEMM + epilogue)
    C   the same without rounding q / k / v                                         (what XM's own rounding costs)

    python -B tests/tools/diag_out_space.py        -> gpurun_out/diag_out_space.json
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import numpy as np
import torch

from conceptattention_amd import flux_dit, ops, sampling
from conceptattention_amd.flux_dit import HeatmapRequest
from conceptattention_amd.pipeline import ConceptAttentionFluxPipeline
from conceptattention_amd.weights import synthetic_inputs

DEV = "cuda:0"
H, NH, D = 3072, 24, 128
SL2 = (1.0 / math.sqrt(128.0)) * 1.4426950408889634


def exact_attention(q, k, v, rows_per=1024):
    """softmax(q k^T / sqrt(128)) v per head in fp32; q [nq, H], k / v [nk, H] fp32."""
    out = torch.empty(q.shape[0], H, device=q.device)
    for h in range(NH):
        c = slice(h * D, (h + 1) * D)
        kh, vh = k[:, c], v[:, c]
        for r in range(0, q.shape[0], rows_per):
            s = q[r:r + rows_per, c] @ kh.t() / math.sqrt(D)
            out[r:r + rows_per, c] = torch.softmax(s, dim=-1) @ vh
    return out


def maps(o_con, o_img):
    return torch.softmax(o_con @ o_img.t(), dim=0)      # [C, L]: softmax over the concepts


def main():
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    pl = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights=None)
    pl.model.weights.init_synthetic(seed=0, on_device=False)
    m, p = pl.model, pl.params
    m.epilogue_logits = False   # (round 5: this stage-by-stage diagnosis reads the fp32 rows of the attention output)
    inp = {k: (v.bfloat16().float() if v.is_floating_point() else v)
           for k, v in synthetic_inputs(p, 1024, 1024, 256, 4, seed=5).items()}
    d = {k: v.to(DEV) for k, v in inp.items()}
    x = d["latent"].to(torch.bfloat16)
    con, con_ids, con_vec = sampling.concept_inputs(d["concepts"].to(torch.bfloat16), d["vec"].to(torch.bfloat16))
    prep = sampling.prepare_from_embeddings(x, d["txt"].to(torch.bfloat16), d["vec"].to(torch.bfloat16))
    img = prep["img"].to(torch.bfloat16).contiguous()
    L_, C, T = img.shape[1], con.shape[1], prep["txt"].shape[1]
    ts = sampling.get_schedule(4, L_, shift=False)
    out = torch.zeros(p.depth, C, L_, device=DEV)
    cross = torch.zeros(p.depth, C, L_, device=DEV)
    m.precompute_conditioning(ts[:-1], prep["vec"], con_vec, 0.0)
    snap = {}
    real_attention = ops.attention

    def spy(probs, nh, **kw):
        first = "QKV" not in snap
        if first:
            torch.cuda.synchronize()
            snap["XM"], snap["QKV"] = m.XM.clone(), m.QKV.clone()
            snap["X"] = m.X.clone()                      # (layer 0's input: nothing has updated it yet)
            b0 = "double_blocks.0."
            snap["mod"] = {(s_, r_, c_): m._mod(b0 + s_ + "_mod.lin", r_, c_).clone()
                           for s_, r_ in (("img", 0), ("txt", 0), ("txt", 1)) for c_ in (0, 1)}
            snap["qk_f16"] = bool(kw.get("qk_f16"))
        real_attention(probs, nh, **kw)
        if first:
            torch.cuda.synchronize()
            snap["ATT32"], snap["ATTI32"] = m.ATT32.clone(), m.ATTI32.clone()
    flux_dit.ops.attention = spy
    req = HeatmapRequest(tuple(range(p.depth)), 0.0, torch.zeros(C, L_, device=DEV), torch.zeros(C, L_, device=DEV),
                         per_layer_out=out, per_layer_cross=cross, per_layer_weight=1.0)
    m(img=img.float() if getattr(m, "fp32_latent", False) else img, img_ids=prep["img_ids"], txt=prep["txt"],
      txt_ids=prep["txt_ids"], concepts=con, concept_ids=con_ids, concept_vec=con_vec, y=prep["vec"],
      timesteps=torch.full((1,), ts[0], device=DEV), guidance=torch.zeros(1, device=DEV), return_vectors=False,
      heatmaps=req, cond_slot=0, stop_after_multimodal_attentions=True)
    torch.cuda.synchronize()
    flux_dit.ops.attention = real_attention
    gold = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "full_depth_schnell.npz"))["out_step0"][0]).to(DEV)
    oT, oI = C, C + T
    res = {"qk_f16": snap["qk_f16"]}

    def err(name, mp):
        res[name] = float((mp - gold).abs().max())
        print(f"{name:4s} max |map - fp32 oracle| = {res[name]:.3e}", flush=True)

    err("A", out[0])
    o_img_hip, o_con_hip = snap["ATTI32"][0, T:], snap["ATT32"][:C]
    err("A2", maps(o_con_hip, o_img_hip))
    # ---- B: exact attention from the HIP q / k / v
    QKV = snap["QKV"]
    def qk_view(t):
        return t.contiguous().view(torch.float16).float() if snap["qk_f16"] else t.float()
    q, k, v = qk_view(QKV[:, :H]) / SL2, qk_view(QKV[:, H:2 * H]), QKV[:, 2 * H:].float()
    o_img = exact_attention(q[oI:], k[oT:], v[oT:])
    o_con = exact_attention(q[:C], torch.cat((k[:C], k[oI:])), torch.cat((v[:C], v[oI:])))
    err("B", maps(o_con, o_img))
    res["attn_rows_img_hip_vs_exact_from_hip_qkv"] = float((o_img_hip - o_img).abs().max())
    res["attn_rows_img_rms"] = float(o_img.pow(2).mean().sqrt())
    # ---- C / C2: projection + norm + RoPE in fp32 torch from the HIP XM
    W = m.weights
    XM = snap["XM"].float()
    rope = m.ROPE                                     # [rows, 64, 2]
    def project(rows, stream):
        b = f"double_blocks.0.{stream}_attn."
        y = XM[rows] @ W[b + "qkv.weight"].float().t() + W[b + "qkv.bias"].float()
        n = y.shape[0]
        def norm_rope(t, scale):
            t = t.view(n, NH, D)
            t = t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6) * scale.float()
            cs, sn = rope[rows][:, None, :, 0], rope[rows][:, None, :, 1]
            te, to = t[..., 0::2], t[..., 1::2]
            return torch.stack((cs * te - sn * to, sn * te + cs * to), -1).reshape(n, H)
        return (norm_rope(y[:, :H], W[b + "norm.query_norm.scale"]), norm_rope(y[:, H:2 * H], W[b + "norm.key_norm.scale"]),
                y[:, 2 * H:])
    qi, ki, vi = project(slice(oI, None), "img")
    qc, kc, vc = project(slice(0, oI), "txt")         # concept + text rows (txt weights)
    qf, kf, vf = torch.cat((qc, qi)), torch.cat((kc, ki)), torch.cat((vc, vi))
    def run(qq, kk, vv, name):
        oi = exact_attention(qq[oI:], kk[oT:], vv[oT:])
        oc = exact_attention(qq[:C], torch.cat((kk[:C], kk[oI:])), torch.cat((vv[:C], vv[oI:])))
        err(name, maps(oc, oi))
    run(qf, kf, vf, "C")
    xm_hip = XM
    q2 = (qf * SL2).half().float() / SL2 if snap["qk_f16"] else (qf * SL2).bfloat16().float() / SL2
    k2 = kf.half().float() if snap["qk_f16"] else kf.bfloat16().float()
    run(q2, k2, vf.bfloat16().float(), "C2")
    res["q_hip_vs_torch_from_xm_maxabs"] = float((q - q2).abs().max())
    res["k_hip_vs_torch_from_xm_maxabs"] = float((k - k2).abs().max())
    res["v_hip_vs_torch_from_xm_maxabs"] = float((v - vf.bfloat16().float()).abs().max())
    # ---- D: LayerNorm + modulation in fp32 torch from the HIP residual rows and the HIP modulation vectors
    from oracle import flux_oracle as O
    X = snap["X"].float()
    def ln_mod(rows, shift, scale):
        return (1 + scale) * O.layer_norm(X[rows]) + shift
    md = snap["mod"]
    def xm_from(mods, Xsrc=None):
        parts = [((slice(0, C)), mods[("txt", 1, 0)], mods[("txt", 1, 1)]), (slice(C, oI), mods[("txt", 0, 0)], mods[("txt", 0, 1)]),
                 (slice(oI, None), mods[("img", 0, 0)], mods[("img", 0, 1)])]
        return torch.cat([(1 + sc) * O.layer_norm((X if Xsrc is None else Xsrc)[r]) + sh for r, sh, sc in parts])
    def downstream(xm, name, round_xm=True):
        nonlocal XM
        XM = xm.bfloat16().float() if round_xm else xm
        a, b_, c_ = project(slice(oI, None), "img")
        d_, e_, f_ = project(slice(0, oI), "txt")
        run(torch.cat((d_, a)), torch.cat((e_, b_)), torch.cat((f_, c_)), name)
    xm_d = xm_from(md)
    res["xm_hip_vs_torch_ln_of_hip_x_and_mod_maxabs"] = float((xm_hip - xm_d.bfloat16().float()).abs().max())
    res["xm_rms"] = float(xm_d.pow(2).mean().sqrt())
    downstream(xm_d, "D")
    # ---- E: the modulation vectors from the fp32 oracle (time_in / vector_in / Modulation on the CPU), HIP X
    sd = {}
    for kname in ("time_in.in_layer", "time_in.out_layer", "vector_in.in_layer", "vector_in.out_layer",
                  "double_blocks.0.img_mod.lin", "double_blocks.0.txt_mod.lin", "img_in", "txt_in"):
        for sfx in (".weight", ".bias"):
            sd[kname + sfx] = W[kname + sfx].float().cpu()
    temb = O.timestep_embedding(torch.tensor([ts[0]]))
    vec = O.mlp_embedder(sd, "time_in", temb) + O.mlp_embedder(sd, "vector_in", inp["vec"])
    cvec = O.mlp_embedder(sd, "time_in", temb) + O.mlp_embedder(sd, "vector_in", con_vec.float().cpu())
    im, tm, cm = (O.modulation(sd, "double_blocks.0.img_mod", vec, 6), O.modulation(sd, "double_blocks.0.txt_mod", vec, 6),
                  O.modulation(sd, "double_blocks.0.txt_mod", cvec, 6))
    mo = {("img", 0, 0): im[0][0, 0], ("img", 0, 1): im[1][0, 0], ("txt", 0, 0): tm[0][0, 0], ("txt", 0, 1): tm[1][0, 0],
          ("txt", 1, 0): cm[0][0, 0], ("txt", 1, 1): cm[1][0, 0]}
    mo = {k_: v_.to(DEV) for k_, v_ in mo.items()}
    for k_ in mo:
        res["mod_err_" + "_".join(map(str, k_))] = float((mo[k_] - md[k_]).abs().max())
    downstream(xm_from(mo), "E")
    # ---- F: the residual rows from fp32 torch too (img_in / txt_in on the bf16-representable inputs)
    Xo = torch.cat((O.linear(sd, "txt_in", inp["concepts"])[0], O.linear(sd, "txt_in", inp["txt"])[0],
                    O.linear(sd, "img_in", O.patchify(inp["latent"]))[0])).to(DEV)
    res["x_hip_vs_oracle_maxabs"] = float((X - Xo).abs().max())
    res["x_rms"] = float(Xo.pow(2).mean().sqrt())
    downstream(xm_from(mo, Xo), "F")
    downstream(xm_from(mo, Xo), "G", round_xm=False)   # ... and no rounding of XM at all: the fp32 oracle itself
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "diag_out_space.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
