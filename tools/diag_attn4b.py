"""Diagnostic: which wrong formula does the broken query block follow (plain-path build of attn4)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd import ops
dev = "cuda"
sl2 = (1.0 / math.sqrt(128.0)) * 1.4426950408889634
nq, nk, nh = 64, 128, 1
g = torch.Generator().manual_seed(nq + nk)
q = (torch.randn(nq, 128, generator=g) * sl2).to(dev).bfloat16()
k = torch.randn(nk, 128, generator=g).to(dev).bfloat16()
v = torch.randn(nk, 128, generator=g).to(dev).bfloat16()
out = torch.zeros(nq, 128, device=dev, dtype=torch.bfloat16)
ops.attention([ops.Attn(q, out, k, v)], nh, q_prescaled=True)
torch.cuda.synchronize()
s = q.float() @ k.float().t()                  # log2-domain scores [64, 128]
m0 = s[:, :64].amax(1, keepdim=True)
p_ok = torch.exp2(s - m0)
cands = {
    "correct": p_ok,
    "tile0 only": torch.cat((p_ok[:, :64], torch.zeros_like(p_ok[:, 64:])), 1),
    "tile1 only": torch.cat((torch.zeros_like(p_ok[:, :64]), p_ok[:, 64:]), 1),
    "tile1 without the reference": torch.cat((p_ok[:, :64], torch.exp2(s[:, 64:])), 1),
    "tile1 with +reference": torch.cat((p_ok[:, :64], torch.exp2(s[:, 64:] + m0)), 1),
    "tile1 with block B's reference": torch.cat((p_ok[:, :64], torch.exp2(s[:, 64:] - m0.roll(32, 0))), 1),
    "tile1 scores of block B's queries": torch.cat((p_ok[:, :64], torch.exp2(s.roll(32, 0)[:, 64:] - m0)), 1),
}
o = out.float()
for name, p in cands.items():
    r = (p / p.sum(1, keepdim=True)) @ v.float()
    ea, eb = (o[:32] - r[:32]).abs().max().item(), (o[32:] - r[32:]).abs().max().item()
    print(f"{name:36s} block A err {ea:.3e}   block B err {eb:.3e}")
print("out block A sample", o[0, :6].tolist(), "nan", int(torch.isnan(o).sum()), "absmax A", o[:32].abs().max().item())
