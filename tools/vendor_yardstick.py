"""Same-box yardstick: the vendor kernels the REFERENCE would run on this GPU against this repo's kernels, shape by shape.

The reference runs its linears through `nn.Linear` (hipBLASLt under PyTorch-ROCm; `modified_double_stream_block.py:90-116,
194-202`, `modified_single_stream_block.py:49-54`) and its attention through `F.scaled_dot_product_attention`
(`flux/math.py:6-12`).  This tool times exactly those two library calls, bf16, at the six projection shapes of the path
for a 5-item and a 1-item forward and at the attention shape (4352 x 4352 x 24 heads x 128), next to `ca_gemm_bf16` /
`ca_attn_fwd_bf16` on the same data, same warm-up and event timing as tools/bench_kernels.py, alternating and keeping
the best of three rounds for each side.  Tools only: nothing in the product path calls a vendor GEMM or SDPA.

    python tools/vendor_yardstick.py [out.json]      (on the GPU box; default gpurun_out/vendor_yardstick.json)

Per shape: the image-stream problem alone (plain bias epilogue on both sides: kernel against kernel) and the launch as
the model makes it (image + [concept|text] stream grouped in one ca_gemm_bf16 call with the model's fused epilogue)
against what the reference makes of it (two F.linear calls plus the elementwise ops its modules run afterwards).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from conceptattention_amd import _lib as L
from conceptattention_amd import ops

dev = "cuda"
H, NH, D = 3072, 24, 128
L_IMG, T_TXT, C = 4096, 256, 4


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).bfloat16()


def best(fns, rounds=3):
    """fns: {name: callable}; alternate them, keep each one's best time (the chip's clock moves with load history)."""
    out = {k: float("inf") for k in fns}
    for _ in range(rounds):
        for k, fn in fns.items():
            out[k] = min(out[k], timeit(fn))
    return out


def gemm_shape(name, B, rows_img, rows_ctx, N, K, epi):
    """rows_ctx = 0: single-block shapes (one [text|image] matrix)."""
    Mi, Mc = B * rows_img, B * rows_ctx
    a_i, w_i, b_i = rnd(Mi, K), rnd(N, K, scale=0.02), rnd(N)
    o_i = torch.empty(Mi, N, device=dev, dtype=torch.bfloat16)
    res = {"shape": name, "items": B, "M_image_stream": Mi, "M_concept_text_stream": Mc, "N": N, "K": K}
    # (a) kernel against kernel: the image-stream problem, bias epilogue
    t = best({"vendor": lambda: F.linear(a_i, w_i, b_i),
              "ours": lambda: ops.gemm([ops.Gemm(a_i, w_i, b_i, o_i, L.EPI_BIAS)])})
    fl = 2.0 * Mi * N * K
    res["plain"] = {"vendor_us": t["vendor"] * 1e6, "ours_us": t["ours"] * 1e6, "vendor_tflops": fl / t["vendor"] / 1e12,
                    "ours_tflops": fl / t["ours"] / 1e12, "ours_over_vendor": t["vendor"] / t["ours"]}
    # (b) the launch as the model makes it against the reference's op sequence for the same rows
    probs, vend = [], []
    for M in ([Mi, Mc] if Mc else [Mi]):
        a, w, b = (a_i, w_i, b_i) if M == Mi else (rnd(M, K), rnd(N, K, scale=0.02), rnd(N))
        if epi == "gate":    # x + gate * linear(.)  (modified_double_stream_block.py:194-202): fp32 residual stream here
            x32 = torch.randn(M, N, device=dev)
            xb = x32.bfloat16()
            g32, gb = torch.randn(N, device=dev), rnd(N)
            probs.append(ops.Gemm(a, w, b, x32, L.EPI_GATE_RESIDUAL, resid=x32, gate=g32))
            vend.append(lambda a=a, w=w, b=b, xb=xb, gb=gb: xb + gb * F.linear(a, w, b))
        elif epi == "gelu":  # mlp.0 + GELU(tanh)
            o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            probs.append(ops.Gemm(a, w, b, o, L.EPI_GELU_TANH))
            vend.append(lambda a=a, w=w, b=b: F.gelu(F.linear(a, w, b), approximate="tanh"))
        else:
            o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            probs.append(ops.Gemm(a, w, b, o, L.EPI_BIAS))
            vend.append(lambda a=a, w=w, b=b: F.linear(a, w, b))
    t = best({"vendor": lambda: [f() for f in vend], "ours": lambda: ops.gemm(probs)})
    fl = sum(2.0 * p.a.shape[0] * N * K for p in probs)
    res["as_in_model"] = {"epilogue": epi, "vendor_us": t["vendor"] * 1e6, "ours_us": t["ours"] * 1e6,
                          "vendor_tflops": fl / t["vendor"] / 1e12, "ours_tflops": fl / t["ours"] / 1e12,
                          "ours_over_vendor": t["vendor"] / t["ours"],
                          "vendor_ops": "F.linear per stream" + {"gate": " + gate * y + x (bf16 elementwise)",
                                                                  "gelu": " + F.gelu(tanh)", "bias": ""}[epi]}
    print(json.dumps(res), flush=True)
    return res


def attn_shape(B):
    n = L_IMG + T_TXT
    bufs = [rnd(n, 3 * H) for _ in range(B)]
    outs = [torch.empty(n, H, device=dev, dtype=torch.bfloat16) for _ in range(B)]
    probs = [ops.Attn(b[:, :H], o, b[:, H:2 * H], b[:, 2 * H:]) for b, o in zip(bufs, outs)]
    # the vendor side gets its favourite layout, [B, heads, L, D] contiguous (the reference builds it with a rearrange)
    q, k, v = (torch.stack([b[:, i * H:(i + 1) * H].reshape(n, NH, D).transpose(0, 1) for b in bufs]).contiguous()
               for i in range(3))
    scale = D ** -0.5
    fns = {"ours": lambda: ops.attention(probs, NH, scale=scale)}
    backends = {}
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        for nm, be in (("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION)):
            def run(be=be):
                with sdpa_kernel(be):
                    return F.scaled_dot_product_attention(q, k, v)
            try:
                run()
                torch.cuda.synchronize()
                backends[nm] = run
            except Exception as ex:   # backend not built for this arch
                print(f"sdpa backend {nm}: unavailable ({type(ex).__name__}: {str(ex)[:120]})", flush=True)
    except ImportError:
        pass
    backends["default_dispatch"] = lambda: F.scaled_dot_product_attention(q, k, v)
    fns.update({"vendor_" + k_: f for k_, f in backends.items()})
    # the model path: q pre-scaled, one-wave-per-SIMD kernel
    fns["ours_prescaled_q(model path)"] = lambda: ops.attention(probs, NH, q_prescaled=True)
    t = best(fns)
    fl = 4.0 * n * n * D * NH * B
    res = {"shape": "attention", "items": B, "rows": n, "heads": NH, "head_dim": D,
           "us": {k_: v_ * 1e6 for k_, v_ in t.items()}, "tflops": {k_: fl / v_ / 1e12 for k_, v_ in t.items()}}
    vend_best = min(v_ for k_, v_ in t.items() if k_.startswith("vendor"))
    res["ours_over_best_vendor"] = vend_best / t["ours_prescaled_q(model path)"]
    # same numbers? (the yardstick is only meaningful if both sides compute the same thing)
    ref = F.scaled_dot_product_attention(q[:1].float(), k[:1].float(), v[:1].float())[0].transpose(0, 1).reshape(n, H)
    ops.attention(probs[:1], NH, scale=scale)
    torch.cuda.synchronize()
    res["max_abs_diff_ours_vs_fp32_sdpa"] = float((outs[0].float() - ref).abs().max())
    res["max_abs_diff_vendor_vs_fp32_sdpa"] = float(
        (backends["default_dispatch"]()[0].transpose(0, 1).reshape(n, H).float() - ref).abs().max())
    print(json.dumps(res), flush=True)
    return res


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/vendor_yardstick.json"
    doc = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "hip": torch.version.hip,
           "method": "best of 3 alternating rounds, 20 launches each after 3 warm-ups, HIP events; bf16; "
                     "vendor GEMM = torch.nn.functional.linear (hipBLASLt / rocBLAS as PyTorch dispatches it), vendor "
                     "attention = F.scaled_dot_product_attention",
           "gemm": [], "attention": []}
    n = L_IMG + T_TXT
    for B in (5, 1):
        doc["gemm"].append(gemm_shape("qkv", B, L_IMG, T_TXT + C, 3 * H, H, "bias"))
        doc["gemm"].append(gemm_shape("proj", B, L_IMG, T_TXT + C, H, H, "gate"))
        doc["gemm"].append(gemm_shape("mlp.0", B, L_IMG, T_TXT + C, 4 * H, H, "gelu"))
        doc["gemm"].append(gemm_shape("mlp.2", B, L_IMG, T_TXT + C, H, 4 * H, "gate"))
        doc["gemm"].append(gemm_shape("linear1", B, n, 0, 7 * H, H, "bias"))
        doc["gemm"].append(gemm_shape("linear2", B, n, 0, H, 5 * H, "gate"))
        torch.cuda.empty_cache()
    for B in (5, 1):
        doc["attention"].append(attn_shape(B))
    wins = [g for g in doc["gemm"] if g["plain"]["ours_over_vendor"] < 0.97]
    doc["vendor_wins_by_more_than_3pct"] = [f"{g['shape']} x{g['items']}" for g in wins]
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(doc, open(out, "w"), indent=1)
    print("written", out)


if __name__ == "__main__":
    main()
