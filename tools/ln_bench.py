"""LN-modulate micro-benchmark at the Flux shapes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd import ops
from tools.bench_kernels import rnd, timeit
x = rnd(4356, 3072)
o = torch.empty_like(x)
sh = [torch.randn(3072, device="cuda") for _ in range(3)]
sc = [torch.randn(3072, device="cuda") for _ in range(3)]
for _ in range(3):
    t = timeit(lambda: ops.ln_modulate(x, o, [(4, sh[0], sc[0]), (260, sh[1], sc[1]), (4356, sh[2], sc[2])]), iters=50)
    print(f"ln_modulate 4356x3072 (3 segments): {t*1e6:.1f} us  {2*x.numel()*2/t/1e9:.0f} GB/s", flush=True)
q = torch.empty(4356, 3072, device="cuda", dtype=torch.uint8); s = torch.empty(4356, device="cuda")
t = timeit(lambda: ops.ln_modulate(x, q, [(4356, sh[2], sc[2])], out_scale=s), iters=50)
print(f"ln_modulate fp8 out: {t*1e6:.1f} us", flush=True)
