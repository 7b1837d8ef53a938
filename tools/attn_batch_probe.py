"""Attention launch of a batched forward: 5 items' main problems (text+image rows) with and without their 5 concept
problems (C rows each, keys = concept + image rows).  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from conceptattention_amd import ops
from tools.bench_kernels import rnd, timeit

B, C, T, Li, NH = 5, 4, 256, 4096, 24
H = NH * 128
oT, oI, n = B * C, B * (C + T), B * (C + T + Li)
qkv = rnd(n, 3 * H)
att = torch.empty(n, H, device="cuda", dtype=torch.bfloat16)
att32 = torch.empty(B * C, H, device="cuda")
qs, ks, vs = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
main, con = [], []
for j in range(B):
    cj, tj, ij = slice(j * C, (j + 1) * C), slice(oT + j * T, oT + (j + 1) * T), slice(oI + j * Li, oI + (j + 1) * Li)
    con.append(ops.Attn(qs[cj], att[cj], ks[cj], vs[cj], ks[ij], vs[ij], out_f32=att32[cj]))
    main.append(ops.Attn(qs[tj], att[tj], ks[tj], vs[tj], ks[ij], vs[ij], q1=qs[ij], out1=att[ij]))
fl = B * 4.0 * (T + Li) ** 2 * 128 * NH
for rep in range(2):
    for name, probs in (("main only (2040 workgroups)", main), ("concept + main (2160)", con + main),
                        ("concept only (120)", con)):
        t = timeit(lambda: ops.attention(probs, NH))
        print(f"{name:32s} {t*1e6:8.1f} us   {fl/t/1e12 if 'only (120' not in name else 0:7.1f} TF/s", flush=True)
