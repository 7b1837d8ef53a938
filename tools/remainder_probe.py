"""What the 20 leftover rows of the [concept | text] stream cost a double block's grouped launches (5 work items:
image stream 20480 rows = 80 row tiles, [concept | text] stream 1300 rows = 5 row tiles + 20 rows): the same launch
with 1280 rows in the second problem, on the 256x256 ping-pong tile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from conceptattention_amd import _lib as L, ops
from tools.bench_kernels import timeit, rnd

PP = L.TILE_PP_256x256


def launch(N, K, kind, m_txt):
    probs = []
    for M in (20480, m_txt):
        a, w, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
        if kind == "gate32":
            x = torch.randn(M, N, device="cuda")
            probs.append(ops.Gemm(a, w, b, x, L.EPI_GATE_RESIDUAL, resid=x, gate=torch.randn(N, device="cuda")))
        elif kind == "gelu":
            probs.append(ops.Gemm(a, w, b, torch.empty(M, N, device="cuda", dtype=torch.bfloat16), L.EPI_GELU_TANH))
        else:
            qkv = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            table = torch.randn(M, 64, 2, device="cuda")
            s128 = torch.ones(128, device="cuda", dtype=torch.bfloat16)
            probs.append(ops.Gemm(a, w, b, qkv, L.EPI_QKV_NORM_ROPE, n_split=N, norm_q=s128, norm_k=s128, rope=table))
    return timeit(lambda: ops.gemm(probs, PP))


tot = {1300: 0.0, 1280: 0.0}
for rep in range(2):
    for (N, K, kind, name) in [(9216, 3072, "qkvrope", "qkv"), (3072, 3072, "gate32", "proj"), (12288, 3072, "gelu", "mlp.0"),
                               (3072, 12288, "gate32", "mlp.2")]:
        t = {m: launch(N, K, kind, m) for m in (1300, 1280)}
        tiles = {m: (80 + (m + 255) // 256) * (N // 256) for m in t}
        print(f"{name:6s} N={N:5d} K={K:5d}: 1300 rows {t[1300]*1e6:7.1f} us ({tiles[1300]} tiles = {tiles[1300]/256:.2f} rounds)   "
              f"1280 rows {t[1280]*1e6:7.1f} us ({tiles[1280]} tiles = {tiles[1280]/256:.2f} rounds)   difference {1e6*(t[1300]-t[1280]):6.1f} us", flush=True)
        if rep:
            for m in t:
                tot[m] += t[m]
print(f"four launches of a double block: {tot[1300]*1e6:.0f} us with the 20 rows, {tot[1280]*1e6:.0f} us without ({100*(tot[1300]/tot[1280]-1):.1f} %)")
