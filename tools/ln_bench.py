"""LN-modulate micro-benchmark at the Flux shapes (development aid): one item with a bf16 stream, and the bench's
5-item launch on the fp32 residual stream (15 segments).  usage: python tools/ln_bench.py [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from conceptattention_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from conceptattention_amd import ops
from tools.bench_kernels import rnd, timeit
x = rnd(4356, 3072)
o = torch.empty_like(x)
sh = [torch.randn(3072, device="cuda") for _ in range(15)]
sc = [torch.randn(3072, device="cuda") for _ in range(15)]
for _ in range(2):
    t = timeit(lambda: ops.ln_modulate(x, o, [(4, sh[0], sc[0]), (260, sh[1], sc[1]), (4356, sh[2], sc[2])]), iters=50)
    print(f"ln_modulate 4356x3072 bf16 in (3 segments): {t*1e6:.1f} us  {2*x.numel()*2/t/1e9:.0f} GB/s", flush=True)
B, C, T, Li = 5, 4, 256, 4096
n = B * (C + T + Li)
x32 = torch.randn(n, 3072, device="cuda")
o5 = torch.empty(n, 3072, device="cuda", dtype=torch.bfloat16)
ends = [(j + 1) * C for j in range(B)] + [B * C + (j + 1) * T for j in range(B)] + [B * (C + T) + (j + 1) * Li for j in range(B)]
segs = [(e, sh[i], sc[i]) for i, e in enumerate(ends)]
for _ in range(3):
    t = timeit(lambda: ops.ln_modulate(x32, o5, segs), iters=50)
    print(f"ln_modulate {n}x3072 fp32 in, 15 segments (5 items): {t*1e6:.1f} us  {x32.numel()*6/t/1e9:.0f} GB/s", flush=True)
q = torch.empty(4356, 3072, device="cuda", dtype=torch.uint8); s = torch.empty(4356, device="cuda")
t = timeit(lambda: ops.ln_modulate(x, q, [(4356, sh[2], sc[2])], out_scale=s), iters=50)
print(f"ln_modulate fp8 out: {t*1e6:.1f} us", flush=True)
