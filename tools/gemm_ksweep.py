"""K sweep of the ping-pong GEMM at a fixed tile count: time(K) = overhead + iters * t_iter.
Separates the per-tile prologue/epilogue cost from the main loop (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from conceptattention_amd import _lib as L
from tools.bench_kernels import bench_gemm

for (M, N, epi, name) in [(4096, 3072, L.EPI_BIAS, "192t-bias"), (4096, 3072, L.EPI_GATE_RESIDUAL, "192t-gate"),
                          (4096, 4096, L.EPI_BIAS, "256t-bias"), (4096, 8192, L.EPI_BIAS, "512t-bias"),
                          (4096, 12288, L.EPI_GELU_TANH, "768t-gelu")]:
    ks = [256, 512, 1024, 2048, 3072, 6144, 12288]
    ts = [bench_gemm(M, N, K, L.TILE_PP_256x256, epi, name=name) for K in ks]
    it = np.array(ks) / 64.0
    A = np.stack([np.ones_like(it), it], 1)
    (o, t), *_ = np.linalg.lstsq(A[2:], np.array(ts[2:]) * 1e6, rcond=None)
    print(f"  fit {name}: overhead {o:.1f} us per launch-round, {t*1000:.1f} ns per K-iteration (K>=1024)", flush=True)
