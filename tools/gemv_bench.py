"""GEMV (modulation weight streaming) micro-benchmark: vectors per pass vs achieved TB/s (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from conceptattention_amd import ops
dev = "cuda"
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
N = 1056768
w = (torch.randn(N, 3072, device=dev) * 0.02).bfloat16()
for nv in (2, 4, 8):
    xv = torch.randn(nv, 3072, device=dev); ov = torch.empty(nv, N, device=dev)
    t = timeit(lambda: ops.gemv(xv, w, None, ov, silu_input=True))
    print(f"gemv {nv}x3072 -> {N}: {t*1e6:.0f} us  {w.numel()*2/t/1e9:.0f} GB/s", flush=True)
