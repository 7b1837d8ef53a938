"""A/B of the two LayerNorm + modulation kernels on the model's 5-item launch: time and bit equality.
usage: python tools/ln_rows_ab.py   (runs itself twice: CA_LN_ROWS=0 / 1)"""
import os, subprocess, sys
# the switches this tool flips exist in the diagnostic build only: python -m conceptattention_amd.csrc.build --ab
_AB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ab", "switches", "libca.so")
if os.path.exists(_AB):
    os.environ.setdefault("CA_LIB_PATH", _AB)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "CA_LN_ROWS" not in os.environ:
    for v in ("0", "1", "0", "1"):
        subprocess.check_call([sys.executable, __file__], env=dict(os.environ, CA_LN_ROWS=v))
    sys.exit(0)
sys.path.insert(0, ROOT)
import hashlib
import torch
from conceptattention_amd import ops
from tools.bench_kernels import timeit
torch.manual_seed(0)
B, C, T, Li, H = 5, 4, 256, 4096, 3072
n = B * (C + T + Li)
x = torch.randn(n, H, device="cuda") * 2
vec = [torch.randn(H, device="cuda") * 0.3 for _ in range(6 * B)]
segs, r = [], 0
for kind, rows in (("c", C), ("t", T), ("i", Li)):
    for j in range(B):
        r += rows
        segs.append((r, vec[len(segs) * 2 % len(vec)], vec[(len(segs) * 2 + 1) % len(vec)]))
out = torch.empty(n, H, device="cuda", dtype=torch.bfloat16)
lo = torch.empty_like(out)
t = timeit(lambda: ops.ln_modulate(x, out, segs))
t2 = timeit(lambda: ops.ln_modulate(x, out, segs, out_lo=lo))
torch.cuda.synchronize()
h = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes() + lo.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
print(f"CA_LN_ROWS={os.environ['CA_LN_ROWS']}: {t*1e6:.1f} us ({n*H*6/t/1e12:.2f} TB/s), with low plane {t2*1e6:.1f} us, sha1 {h}", flush=True)
