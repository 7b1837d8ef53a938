"""Time the REFERENCE's own CPU path for the hot path's two block types.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

Build container only (imports /root/reference by the recipe of SURVEY.md section 8c, which never travels to the GPU box):

    python -B oracle/time_reference_cpu.py            # -> profiles/r05_reference_cpu_baseline.json

One full-size ModifiedDoubleStreamBlock.forward (concept_attention/modified_double_stream_block.py:69-204) and one
ModifiedSingleStreamBlock.forward (concept_attention/modified_single_stream_block.py:43-56) at BASELINE.json configs[1]'s
geometry (H = 3072, 4096 image tokens, 256 text tokens, 4 concepts), seeded synthetic weights and inputs
(oracle/full_block_case.py), fp32 and bf16 (the reference's production dtype), thread count stated; the per-call
figure is the blocks' time x (19, 38) x 4 steps -- the DiT only, no text encoders / VAE, which is also all the MI355X
figure covers.  The port (oracle/flux_oracle.py, what bench.py's cpu_baseline times on the GPU box's host) is timed
beside it on the same machine so the two CPU figures can be related.
"""
from __future__ import annotations

import json
import os
import platform
import subprocess
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def _time(fn, reps):
    fn()   # warm
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    from oracle.make_goldens import _import_reference
    from oracle.full_block_case import full_block_inputs
    from oracle import flux_oracle as O
    from conceptattention_amd.params import FluxParams
    from conceptattention_amd.weights import synthetic_state_dict
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    ref = _import_reference()
    from concept_attention.flux.src.flux.modules.layers import EmbedND
    p = FluxParams()
    H, NH = p.hidden_size, p.num_heads
    case = full_block_inputs(p)
    emb = EmbedND(dim=128, theta=p.theta, axes_dim=list(p.axes_dim))
    pe = emb(torch.cat((case["txt_ids"], case["img_ids"]), 1))
    cpe = emb(torch.cat((case["concept_ids"], case["img_ids"]), 1))
    sd_d = {k[len("double_blocks.0."):]: v.bfloat16().float()
            for k, v in synthetic_state_dict(p, seed=0, prefix="double_blocks.0.").items()}
    sd_s = {k[len("single_blocks.0."):]: v.bfloat16().float()
            for k, v in synthetic_state_dict(p, seed=0, prefix="single_blocks.0.").items()}
    res = {}
    for dt_name, dt, reps in (("fp32", torch.float32, 2), ("bf16", torch.bfloat16, 3)):
        dbl = ref["ModifiedDoubleStreamBlock"](H, NH, mlp_ratio=p.mlp_ratio, qkv_bias=True).eval()
        dbl.load_state_dict(sd_d, strict=True)
        dbl = dbl.to(dt)
        sgl = ref["ModifiedSingleStreamBlock"](H, NH, mlp_ratio=p.mlp_ratio).eval()
        sgl.load_state_dict(sd_s, strict=True)
        sgl = sgl.to(dt)
        c = {k: (v.to(dt) if v.is_floating_point() and k not in ("img_ids", "txt_ids", "concept_ids") else v)
             for k, v in case.items()}
        x = torch.cat((c["txt"], c["img"]), 1)
        with torch.no_grad():
            td = _time(lambda: dbl(img=c["img"], txt=c["txt"], vec=c["vec"], pe=pe, concepts=c["concepts"],
                                   concept_vec=c["concept_vec"], concept_pe=cpe), reps)
            ts = _time(lambda: sgl(x, vec=c["vec"], pe=pe), reps)
        call = 4 * (19 * td + 38 * ts)
        res[dt_name] = {"double_block_s": td, "single_block_s": ts, "call_s_extrapolated": call,
                        "concept_heatmaps_per_s": 4 / call, "reps": reps}
        print(f"reference {dt_name}: double {td:.2f} s, single {ts:.2f} s -> {call:.0f} s per call", flush=True)
        del dbl, sgl
    # the port on the same machine (what bench.py's cpu_baseline leg times on the GPU box's host cores)
    sd = {"double_blocks.0." + k: v for k, v in sd_d.items()}
    sd.update({"single_blocks.0." + k: v for k, v in sd_s.items()})
    rope_ti = O.rope_cos_sin(torch.cat((case["txt_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    rope_ci = O.rope_cos_sin(torch.cat((case["concept_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    x = torch.cat((case["txt"], case["img"]), 1)
    with torch.no_grad():
        td = _time(lambda: O.double_block(sd, "double_blocks.0.", NH, case["img"], case["txt"], case["vec"], rope_ti,
                                          case["concepts"], case["concept_vec"], rope_ci), 2)
        ts = _time(lambda: O.single_block(sd, "single_blocks.0.", NH, x, case["vec"], rope_ti), 2)
    call = 4 * (19 * td + 38 * ts)
    res["port_fp32"] = {"double_block_s": td, "single_block_s": ts, "call_s_extrapolated": call,
                        "concept_heatmaps_per_s": 4 / call, "reps": 2,
                        "what": "oracle/flux_oracle.py (concept attention for the C query rows only)"}
    print(f"port fp32: double {td:.2f} s, single {ts:.2f} s -> {call:.0f} s per call", flush=True)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    doc = {"what": "the reference's own modules (ModifiedDoubleStreamBlock / ModifiedSingleStreamBlock) on the host CPU of "
                   "the BUILD container, full-size blocks, warm runs; per call = 4 steps x (19 double + 38 single), DiT only",
           "threads": threads, "cpu": cpu, "machine": platform.machine(), "torch": torch.__version__,
           "geometry": {"hidden": H, "image_tokens": 4096, "text_tokens": 256, "concepts": 4},
           "git_head": head, "results": res,
           "command": "python -B oracle/time_reference_cpu.py"}
    out = os.path.join(ROOT, "profiles", "r05_reference_cpu_baseline.json")
    json.dump(doc, open(out, "w"), indent=1)
    print("written", out)


if __name__ == "__main__":
    main()
