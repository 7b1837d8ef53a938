"""Stage-by-stage error of the HIP double block vs the fp32 oracle (full size)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from conceptattention_amd import ops
from conceptattention_amd.flux_dit import HipFluxDiT, HeatmapRequest, DICT_KEYS, _Geom
from conceptattention_amd.params import FluxParams
from conceptattention_amd.weights import synthetic_state_dict
from oracle import flux_oracle as O
from oracle.full_block_case import full_block_inputs

DEV = "cuda:0"
p = FluxParams(depth=1, depth_single_blocks=0)
case = full_block_inputs(p)
sd = {k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=0, prefix="double_blocks.0.").items()}
m = HipFluxDiT(p, DEV)
m.load_state_dict(sd, strict=False)
L, T, C = 4096, 256, 4
CT = C + T
m._workspace(L, T, C)
m._rope_table(case["img_ids"], case["txt_ids"], case["concept_ids"], C, T)
m.X[:C].copy_(case["concepts"][0]); m.X[C:CT].copy_(case["txt"][0]); m.X[CT:].copy_(case["img"][0])
m.VEC[0].copy_(case["vec"][0]); m.VEC[1].copy_(case["concept_vec"][0])
m._modulations()
out = {k: [] for k in DICT_KEYS}
req = HeatmapRequest((0,), 1.0, torch.zeros(C, L, device=DEV), torch.zeros(C, L, device=DEV))

def st(name, hip, ref):
    hip, ref = hip.float().cpu(), ref.float().cpu()
    e = hip - ref
    print(f"{name:28s} rel L2 {e.norm().item()/ref.norm().item():.3e} max {e.abs().max().item():.3e} ref rms {ref.pow(2).mean().sqrt().item():.3e}", flush=True)

# oracle pieces
pfx = "double_blocks.0."
im = O.modulation(sd, pfx + "img_mod", case["vec"], 6)
tm = O.modulation(sd, pfx + "txt_mod", case["vec"], 6)
cm = O.modulation(sd, pfx + "txt_mod", case["concept_vec"], 6)
H = 3072
mo = m.weights.mod_offset
st("mod img shift1", m.MOD[0, mo[pfx+"img_mod.lin"]:mo[pfx+"img_mod.lin"]+H], im[0][0,0])
st("mod con gate1", m.MOD[1, mo[pfx+"txt_mod.lin"]+2*H:mo[pfx+"txt_mod.lin"]+3*H], cm[2][0,0])
xm_img = (1 + im[1]) * O.layer_norm(case["img"]) + im[0]
xm_con = (1 + cm[1]) * O.layer_norm(case["concepts"]) + cm[0]
xm_txt = (1 + tm[1]) * O.layer_norm(case["txt"]) + tm[0]

m._double_block(0, _Geom(1, C, T, L), None, out, True, [req])
torch.cuda.synchronize()
st("XM(mod2) n/a", m.XM[:C], m.XM[:C])
qkv_img = O.linear(sd, pfx + "img_attn.qkv", xm_img)
qkv_con = O.linear(sd, pfx + "txt_attn.qkv", xm_con)
qkv_txt = O.linear(sd, pfx + "txt_attn.qkv", xm_txt)
st("V img", m.QKV[CT:, 2*H:], qkv_img[0, :, 2*H:])
st("V con", m.QKV[:C, 2*H:], qkv_con[0, :, 2*H:])
iq, ik, iv = O._split_heads(qkv_img, 24); cq, ck, cv = O._split_heads(qkv_con, 24); tq, tk, tv = O._split_heads(qkv_txt, 24)
iqn = O.rms_norm(iq, sd[pfx+"img_attn.norm.query_norm.scale"]); ikn = O.rms_norm(ik, sd[pfx+"img_attn.norm.key_norm.scale"])
cqn = O.rms_norm(cq, sd[pfx+"txt_attn.norm.query_norm.scale"]); ckn = O.rms_norm(ck, sd[pfx+"txt_attn.norm.key_norm.scale"])
st("QPRE img", m.QPRE[CT:], O._merge_heads(iqn)[0])
st("QPRE con", m.QPRE[:C], O._merge_heads(cqn)[0])
rope_ci = O.rope_cos_sin(torch.cat((case["concept_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
k_ci = O.apply_rope(torch.cat((ckn, ikn), 2), *rope_ci)
q_ci = O.apply_rope(torch.cat((cqn, iqn), 2), *rope_ci)
st("K img roped", m.QKV[CT:, H:2*H], O._merge_heads(k_ci[:, :, C:])[0])
st("Q con roped", m.QKV[:C, :H], O._merge_heads(q_ci[:, :, :C])[0])
con_attn = O._merge_heads(O.sdpa(q_ci[:, :, :C], k_ci, torch.cat((cv, iv), 2)))[0]
st("ATT con (exact inputs)", m.ATT[:C], con_attn)
# concept attention recomputed in fp64 from the HIP path's OWN q,k,v  -> isolates the attention kernel
qq = m.QKV[:C, :H].double().view(C, 24, 128).transpose(0, 1)
kk = torch.cat((m.QKV[:C, H:2*H], m.QKV[CT:, H:2*H])).double().view(-1, 24, 128).transpose(0, 1)
vv = torch.cat((m.QKV[:C, 2*H:], m.QKV[CT:, 2*H:])).double().view(-1, 24, 128).transpose(0, 1)
w = torch.softmax(qq @ kk.transpose(1, 2) / math.sqrt(128), -1)
st("ATT con (hip q,k,v inputs)", m.ATT[:C], (w @ vv).transpose(0, 1).reshape(C, H))
# logits from mixed sources
import numpy as np
g = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", "block_full.npz"))
ref_lo = torch.from_numpy(g["logits_output_space"][0])
lo = torch.empty(C, L, device=DEV)
ops.heatmap_logits(m.ATT[CT:], m.ATT[:C], lo)
e = (lo.cpu() - ref_lo)
print("logits err: max", e.abs().max().item(), "per-concept mean err", e.mean(1).tolist(), "std over p", e.std(1).tolist())
lo2 = (m.ATT[CT:].float() @ con_attn.to(DEV).t()).t()
e2 = (lo2.cpu() - ref_lo)
print("logits (hip img, exact con): max", e2.abs().max().item(), "mean", e2.mean(1).tolist(), "std", e2.std(1).tolist())
