import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd import ops
dev = "cuda"
torch.manual_seed(0)
nh, n, C, T = 24, 4352, 4, 256
H = nh * 128
buf = torch.randn(n + C, 3 * H, device=dev)
buf[:, :H] *= 0.15
buf[:, H:2 * H] *= 6.0
buf = buf.bfloat16()
q, k, v = buf[:, :H], buf[:, H:2 * H], buf[:, 2 * H:]
out = torch.zeros(n + C, H, device=dev, dtype=torch.bfloat16)
ops.attention([ops.Attn(q[C:], out[C:], k[C:], v[C:]), ops.Attn(q[:C], out[:C], k[:C], v[:C], k[C + T:], v[C + T:])], nh)

def ref(qq, kk, vv):
    qh = qq.double().view(-1, nh, 128).transpose(0, 1)
    kh = kk.double().view(-1, nh, 128).transpose(0, 1)
    vh = vv.double().view(-1, nh, 128).transpose(0, 1)
    w = torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(128), -1)
    return (w @ vh).transpose(0, 1).reshape(qq.shape[0], H)

def stats(name, o, r):
    e = o.double() - r
    print(f"{name}: rel L2 {e.norm().item() / r.norm().item():.3e}  max {e.abs().max().item():.3e} mean err {e.mean().item():.3e}  ref rms {r.pow(2).mean().sqrt().item():.3e}")

r_main = ref(q[C:C + 512], k[C:], v[C:])
stats("hip main rows", out[C:C + 512], r_main)
stats("  (bf16 rounding of exact)", r_main.bfloat16(), r_main)
r_con = ref(q[:C], torch.cat((k[:C], k[C + T:])), torch.cat((v[:C], v[C + T:])))
stats("hip concept rows", out[:C], r_con)
stats("  (bf16 rounding of exact)", r_con.bfloat16(), r_con)
# torch's own bf16 SDPA for comparison
qh = q[C:C + 512].view(-1, nh, 128).transpose(0, 1)[None]
kh = k[C:].view(-1, nh, 128).transpose(0, 1)[None]
vh = v[C:].view(-1, nh, 128).transpose(0, 1)[None]
o_t = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh)[0].transpose(0, 1).reshape(512, H)
stats("torch bf16 sdpa main rows", o_t, r_main)
