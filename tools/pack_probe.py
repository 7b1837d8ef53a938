"""Do two kernels from two HIP streams share the chip's CUs?  Times grouped GEMM launches of 128 / 192 / 216 tiles
(256x256, K = 3072) back to back on one stream and concurrently on two.  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from conceptattention_amd import _lib as L
from conceptattention_amd import ops
from tools.bench_kernels import rnd

dev = "cuda:0"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def mk(M, N, K=3072):
    a, w, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
    return ops.Gemm(a, w, b, torch.empty(M, N, device=dev, dtype=torch.bfloat16))


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for tiles, (M, N) in {128: (2048, 4096), 192: (4096, 3072), 216: (4608, 3072), 256: (4096, 4096)}.items():
    g1, g2 = mk(M, N), mk(M, N)

    def seq():
        ops.gemm([g1], L.TILE_PP_256x256)
        ops.gemm([g2], L.TILE_PP_256x256)

    def par():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ops.gemm([g1], L.TILE_PP_256x256)
        with torch.cuda.stream(s2):
            ops.gemm([g2], L.TILE_PP_256x256)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    one = timed(lambda: ops.gemm([g1], L.TILE_PP_256x256))
    print(f"{tiles:4d} tiles: one launch {one:7.1f} us | two back to back {timed(seq):7.1f} us | two on two streams {timed(par):7.1f} us",
          flush=True)
