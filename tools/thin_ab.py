"""Thin-row GEMM launches alone (the 5 x 4 concept rows of a batch; the modulation GEMM of 40 / 10 conditioning vectors):
microseconds per launch for whichever library CA_LIB_PATH names.  usage: [CA_LIB_PATH=tools/ab/.../libca.so] python tools/thin_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd import _lib as L, ops

dev = "cuda"
torch.manual_seed(0)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("library:", L.LIB_PATH)
for name, N, K, epi in (("proj-like  N3072  K3072  gate", 3072, 3072, L.EPI_GATE_RESIDUAL), ("mlp.0-like N12288 K3072  gelu", 12288, 3072, L.EPI_GELU_TANH),
                        ("mlp.2-like N3072  K12288 gate", 3072, 12288, L.EPI_GATE_RESIDUAL)):
    for M in (20, 40):
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        if epi == L.EPI_GATE_RESIDUAL:
            x = torch.zeros(M, N, device=dev)
            gate = torch.ones(N, device=dev)
            g = ops.Gemm(a, w, b, x, epi, resid=x, gate=gate)
        else:
            g = ops.Gemm(a, w, b, torch.zeros(M, N, device=dev, dtype=torch.bfloat16), epi)
        t = timeit(lambda: ops.gemm([g], L.TILE_PP_256x256))
        print(f"{name}  M={M:3d}: {t:7.1f} us  ({N * K * 2 / t / 1e6:5.2f} TB/s of weights)")
H, NM = 3072, 19 * 12 * 3072 + 38 * 3 * 3072 + 2 * 3072
w = (torch.randn(NM, H, device=dev) * 0.02).bfloat16()
bias = torch.randn(NM, device=dev).bfloat16()
ones = torch.ones(NM, device=dev)
for nv in (10, 40):
    vecs = torch.randn(nv, H, device=dev)
    out = torch.zeros(nv, NM, device=dev)
    t = timeit(lambda: ops.modulation_gemm(vecs, w, bias, out, ones), 10)
    print(f"modulation GEMM, {nv} vectors ({NM} x {H} weights = {NM * H * 2 / 1e9:.2f} GB): {t / 1e3:6.2f} ms")
