"""Race / hazard screen for ca_attn4_kernel at the benchmarked launch shape: 5 work items in one launch (per item its
text + image rows as two query and two key segments, and its concept rows against [concept keys ; image keys]) against
the SAME items launched one at a time in the single-item layout ([concept | text | image] rows adjacent: one query and
one key segment for the main problem).  Every output must be bit-identical between the two forms and from launch to
launch; fresh random data every few iterations, launches alternating between two streams.
usage: python tools/soak_attn4.py [iterations=60]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from conceptattention_amd import ops

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B, C, T, Li, NH = 5, 4, 256, 4096, 24
H = NH * 128
sl2 = 1.4426950408889634 / 128 ** 0.5
oT, oI, n = B * C, B * (C + T), B * (C + T + Li)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad_form = bad_repeat = 0
t0 = time.time()
for it in range(iters):
    if it % 4 == 0:
        g = torch.Generator(device=dev).manual_seed(1000 + it)
        qkv = (torch.randn(n, 3 * H, device=dev, generator=g) * (1.0 + 0.5 * (it % 3))).bfloat16()
        qkv[:, :H] = (qkv[:, :H].float() * sl2).bfloat16()
        first = None
        torch.cuda.synchronize()      # the data is written on the default stream, the launches run on s1 / s2
    qs, ks, vs = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    with torch.cuda.stream(s1 if it % 2 == 0 else s2):
        att = torch.zeros(n, H, device=dev, dtype=torch.bfloat16)
        att32 = torch.zeros(B * C, H, device=dev)
        probs = []
        for j in range(B):
            cj, tj = slice(j * C, (j + 1) * C), slice(oT + j * T, oT + (j + 1) * T)
            ij = slice(oI + j * Li, oI + (j + 1) * Li)
            probs.append(ops.Attn(qs[cj], att[cj], ks[cj], vs[cj], ks[ij], vs[ij], out_f32=att32[cj]))
            probs.append(ops.Attn(qs[tj], att[tj], ks[tj], vs[tj], ks[ij], vs[ij], q1=qs[ij], out1=att[ij]))
        ops.attention(probs, NH, q_prescaled=True)
    with torch.cuda.stream(s2 if it % 2 == 0 else s1):
        singles = []
        for j in range(B):          # the single-item layout: rows [concept | text | image] of item j, adjacent
            rows = torch.cat((torch.arange(j * C, (j + 1) * C), torch.arange(oT + j * T, oT + (j + 1) * T),
                              torch.arange(oI + j * Li, oI + (j + 1) * Li))).to(dev)
            one = qkv[rows].contiguous()
            q1, k1, v1 = one[:, :H], one[:, H:2 * H], one[:, 2 * H:]
            o1 = torch.zeros(C + T + Li, H, device=dev, dtype=torch.bfloat16)
            o32 = torch.zeros(C, H, device=dev)
            ops.attention([ops.Attn(q1[:C], o1[:C], k1[:C], v1[:C], k1[C + T:], v1[C + T:], out_f32=o32),
                           ops.Attn(q1[C:], o1[C:], k1[C:], v1[C:])], NH, q_prescaled=True)
            singles.append((rows, o1, o32))
    torch.cuda.synchronize()
    for j, (rows, o1, o32) in enumerate(singles):
        if not (torch.equal(att[rows], o1) and torch.equal(att32[j * C:(j + 1) * C], o32)):
            bad_form += 1
    if first is None:
        first = att.clone()
    elif not torch.equal(att, first):
        bad_repeat += 1
    if it % 20 == 19:
        print(f"iteration {it + 1}: batched != single {bad_form}, launch != first launch {bad_repeat} ({time.time() - t0:.0f} s)",
              flush=True)
print("RESULT", "clean" if not (bad_form or bad_repeat) else f"MISMATCHES form {bad_form} repeat {bad_repeat}")
sys.exit(1 if (bad_form or bad_repeat) else 0)
