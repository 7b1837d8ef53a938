"""Diagnostic: the spike test's case through the pre-scaled kernel; which rows / columns are wrong."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conceptattention_amd import ops
from test_kernels_gpu import rnd, attn_ref, attn_forms
for gain, nq, nk, spikes in [(4.0, 300, 640, ((500, 17), (70, 290))), (4.0, 64, 640, ((500, 17),)), (4.0, 64, 640, ((70, 17),)),
                             (4.0, 64, 128, ((70, 17),)), (4.0, 64, 64, ((30, 17),)), (4.0, 64, 192, ((130, 17),))]:
    q, k, v = rnd(nq, 128), rnd(nk, 128, seed=3), rnd(nk, 128, seed=4)
    for kp, qr in spikes:
        k[kp] = (q[qr].float() * gain).bfloat16()
    out = torch.zeros(nq, 128, device="cuda", dtype=torch.bfloat16)
    qk, qe, kw = attn_forms(q, True)
    ops.attention([ops.Attn(qk, out, k, v)], 1, **kw)
    torch.cuda.synchronize()
    ref = attn_ref(qe, k, v, 1)
    e = (out.float() - ref).abs()
    nanrows = torch.isnan(out.float()).any(1).nonzero().flatten().tolist()
    bad = (e.nan_to_num(1e9).amax(1) > 1e-2).nonzero().flatten().tolist()
    print(f"gain {gain} nq {nq} nk {nk} spikes {spikes}: nan rows {len(nanrows)} {nanrows[:40]} bad rows {len(bad)} {bad[:10]}", flush=True)
