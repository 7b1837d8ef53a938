"""Development aid for the one-wave-per-SIMD attention kernel: accuracy against the fp32 reference on a ladder of
shapes (pre-scaled q, one / two key segments, ragged tails, inactive waves), then the timing of the Flux shape and of
the 5-item launch.  Run once with CA_ATTN_KERNEL=4 and once without (the switch is read once per process).
usage: [CA_ATTN_KERNEL=4] python tools/attn4_check.py [--no-time] [--only-time]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from conceptattention_amd import ops

dev = "cuda"
sl2 = (1.0 / math.sqrt(128.0)) * 1.4426950408889634


def ref(q, k, v, nh):
    qh = q.float().view(q.shape[0], nh, 128).transpose(0, 1)
    kh = k.float().view(k.shape[0], nh, 128).transpose(0, 1)
    vh = v.float().view(v.shape[0], nh, 128).transpose(0, 1)
    w = torch.softmax(qh @ kh.transpose(1, 2) * math.log(2.0), dim=-1)
    return (w @ vh).transpose(0, 1).reshape(q.shape[0], nh * 128)


def one(nq, n0, n1, nh, qscale=1.0):
    g = torch.Generator().manual_seed(nq * 7 + n0 * 3 + n1)
    nk = n0 + n1
    q = (torch.randn(nq, nh * 128, generator=g) * sl2 * qscale).to(dev).bfloat16()
    k = torch.randn(nk + 8, nh * 128, generator=g).to(dev).bfloat16()
    v = torch.randn(nk + 8, nh * 128, generator=g).to(dev).bfloat16()
    out = torch.zeros(nq, nh * 128, device=dev, dtype=torch.bfloat16)
    if n1:
        # the two key segments are not adjacent in memory (8 rows between them)
        a = ops.Attn(q, out, k[:n0], v[:n0], k[n0 + 8:], v[n0 + 8:])
        kk, vv = torch.cat((k[:n0], k[n0 + 8:])), torch.cat((v[:n0], v[n0 + 8:]))
    else:
        a = ops.Attn(q, out, k[:n0], v[:n0])
        kk, vv = k[:n0], v[:n0]
    ops.attention([a], nh, q_prescaled=True)
    torch.cuda.synchronize()
    r = ref(q, kk, vv, nh)
    e = (out.float() - r).abs()
    bad = (e.amax(1) > 2e-2).nonzero().flatten().tolist()
    print(f"nq={nq:5d} n0={n0:5d} n1={n1:5d} nh={nh:2d}: max err {e.max().item():.3e} nan "
          f"{int(torch.isnan(out.float()).sum())} bad rows {len(bad)} first {bad[:6]}", flush=True)
    return len(bad) == 0 and not torch.isnan(out.float()).any()


if "--only-time" not in sys.argv:
    ok = True
    for shape in [(64, 64, 0, 1), (64, 128, 0, 1), (64, 192, 0, 1), (64, 256, 0, 1), (64, 320, 0, 1), (64, 200, 0, 1),
                  (64, 40, 0, 1), (64, 100, 0, 1), (40, 264, 0, 1), (300, 1000, 0, 2), (256, 512, 0, 3),
                  (4, 4, 260, 2), (4, 4, 256, 2), (264, 8, 256, 2), (264, 3, 509, 1), (100, 70, 70, 1),
                  (4352, 4352, 0, 2), (4, 4, 4096, 24), (2112, 256, 4096, 3)]:
        ok = one(*shape) and ok
    print("ACCURACY", "OK" if ok else "FAILED", flush=True)

if "--no-time" not in sys.argv:
    from tools.bench_kernels import rnd, timeit
    nh, n = 24, 4352
    H = nh * 128
    buf = rnd(n, 3 * H)
    buf[:, :H] *= sl2
    out = torch.empty(n, H, device=dev, dtype=torch.bfloat16)
    probs = [ops.Attn(buf[:, :H], out, buf[:, H:2 * H], buf[:, 2 * H:])]
    for _ in range(2):
        t = timeit(lambda: ops.attention(probs, nh, q_prescaled=True))
        print(f"attn {n}x{n}x{nh}: {t*1e6:8.1f} us  {4*n*n*128*nh/t/1e12:7.1f} TF/s", flush=True)
    B, C, T, Li = 5, 4, 256, 4096
    oT, oI, nn = B * C, B * (C + T), B * (C + T + Li)
    qkv = rnd(nn, 3 * H)
    qkv[:, :H] *= sl2
    att = torch.empty(nn, H, device=dev, dtype=torch.bfloat16)
    att32 = torch.empty(B * C, H, device=dev)
    qs, ks, vs = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    main, con = [], []
    for j in range(B):
        cj, tj = slice(j * C, (j + 1) * C), slice(oT + j * T, oT + (j + 1) * T)
        ij = slice(oI + j * Li, oI + (j + 1) * Li)
        con.append(ops.Attn(qs[cj], att[cj], ks[cj], vs[cj], ks[ij], vs[ij], out_f32=att32[cj]))
        main.append(ops.Attn(qs[tj], att[tj], ks[tj], vs[tj], ks[ij], vs[ij], q1=qs[ij], out1=att[ij]))
    fl = B * 4.0 * (T + Li) ** 2 * 128 * nh
    for name, probs in (("5 items, main only", main), ("5 items, concept + main", con + main),
                        ("5 items, main + concept", main + con), ("concept only", con)):
        t = timeit(lambda: ops.attention(probs, nh, q_prescaled=True))
        print(f"{name:28s} {t*1e6:8.1f} us   {fl/t/1e12 if 'only' != name[-4:] or 'main' in name else 0:7.1f} TF/s",
              flush=True)
