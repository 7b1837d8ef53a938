import os, sys
sys.path.insert(0, "/root/repo")
if len(sys.argv) > 1:
    from conceptattention_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from conceptattention_amd import ops
from tools.bench_kernels import rnd, timeit
img = rnd(4096, 3072)
for C, dt in ((4, torch.float32), (4, torch.bfloat16), (8, torch.float32)):
    con = torch.randn(C, 3072, device="cuda").to(dt)
    lg = torch.empty(C, 4096, device="cuda")
    t = timeit(lambda: ops.heatmap_logits(img, con, lg), iters=100)
    print(f"heatmap_logits 4096x3072 C={C} {dt}: {t*1e6:.1f} us  {img.numel()*2/t/1e9:.0f} GB/s", flush=True)
