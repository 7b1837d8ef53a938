"""Tile-order sweep of the ping-pong GEMM: the model's six grouped launches (image + [concept | text] stream, fused epilogue
kinds) at 5 items and at 1 item per forward, with CA_GEMM_GROUP_M = 1, 2, 3, 4, 6, 8 (one fresh process each: the
override is read once) and with the library's own choice ("auto").  Best of 3 x 20 launches, same box.

    python tools/gemm_group_m.py            -> table on stdout (profiles/r04_gemm_group_m.txt)
"""
import os
# the switches this tool flips exist in the diagnostic build only: python -m conceptattention_amd.csrc.build --ab
_AB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ab", "switches", "libca.so")
if os.path.exists(_AB):
    os.environ.setdefault("CA_LIB_PATH", _AB)
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = 3072
SHAPES = [("qkv", 3 * H, H, "bias"), ("proj", H, H, "gate"), ("mlp.0", 4 * H, H, "gelu"), ("mlp.2", H, 4 * H, "gate"),
          ("linear1", 7 * H, H, "bias"), ("linear2", H, 5 * H, "gate")]

if len(sys.argv) > 1 and sys.argv[1] == "--run":
    sys.path.insert(0, ROOT)
    import torch
    from conceptattention_amd import _lib as L
    from conceptattention_amd import ops
    from tools.bench_kernels import rnd, timeit
    for B in (5, 1):
        for name, N, K, epi in SHAPES:
            single = name.startswith("linear")
            Ms = [B * 4352] if single else [B * 4096, B * 260]
            probs = []
            for M in Ms:
                a, w, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
                if epi == "gate":
                    x32 = torch.randn(M, N, device="cuda")
                    probs.append(ops.Gemm(a, w, b, x32, L.EPI_GATE_RESIDUAL, resid=x32, gate=torch.randn(N, device="cuda")))
                else:
                    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                    probs.append(ops.Gemm(a, w, b, o, L.EPI_GELU_TANH if epi == "gelu" else L.EPI_BIAS))
            t = min(timeit(lambda: ops.gemm(probs, L.TILE_PP_256x256)) for _ in range(3))
            print(f"RESULT {B} {name} {t * 1e6:.1f}", flush=True)
            del probs
            torch.cuda.empty_cache()
    sys.exit(0)

res = {}
modes = ["auto", "1", "2", "3", "4", "6", "8", "12", "16"]
for g in modes:
    env = dict(os.environ)
    env.pop("CA_GEMM_GROUP_M", None)
    if g != "auto":
        env["CA_GEMM_GROUP_M"] = g
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--run"], env=env, capture_output=True, text=True).stdout
    for l in out.splitlines():
        if l.startswith("RESULT"):
            _, B, name, us = l.split()
            res[(int(B), name, g)] = float(us)
    print("done", g, flush=True)
for B in (5, 1):
    print(f"\n{B} item(s) per forward: us per grouped launch")
    print(f"{'launch':10s} " + " ".join(f"{g:>8s}" for g in modes) + "   best")
    for name, *_ in SHAPES:
        row = [res.get((B, name, g), float('nan')) for g in modes]
        best = min((v, g) for v, g in zip(row[1:], modes[1:]))
        print(f"{name:10s} " + " ".join(f"{v:8.1f}" for v in row) + f"   {best[1]} ({best[0]:.1f})")
