"""Race screen for the hand-synchronised kernels (LDS-DMA hand-offs behind counted waits and barriers): the same
launches are repeated many times, on two streams at once so that neighbours and timing vary, and every result
is compared bit for bit with the first one.  A protocol slip shows up as a rare differing tile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd import _lib as L, ops
from tools.bench_kernels import rnd

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
nh, C, n = 24, 4, 4352
buf = rnd(C + n, 9216)
q, k, v = buf[:, :3072], buf[:, 3072:6144], buf[:, 6144:]
a0, a1 = rnd(4096, 3072), rnd(260, 3072)
w0, w1, b0 = rnd(12288, 3072, scale=0.02), rnd(12288, 3072, scale=0.02), rnd(12288)
al, wl = rnd(4352, 15360), rnd(3072, 15360, scale=0.02)
gate = torch.randn(3072, device=dev)
qa, sa = ops.quantize_rows_fp8(a0)
qw, sw = ops.quantize_rows_fp8(w0)


def attn():
    o = torch.zeros(C + n, 3072, device=dev, dtype=torch.bfloat16)
    ops.attention([ops.Attn(q[C:], o[C:], k[C:], v[C:]), ops.Attn(q[:C], o[:C], k[:C], v[:C], k[C + 256:], v[C + 256:])], nh)
    return o


q4 = (q.float() * (1.4426950408889634 / 128 ** 0.5)).bfloat16()


def attn4():
    o = torch.zeros(C + n, 3072, device=dev, dtype=torch.bfloat16)
    ops.attention([ops.Attn(q4[C:], o[C:], k[C:], v[C:]), ops.Attn(q4[:C], o[:C], k[:C], v[:C], k[C + 256:], v[C + 256:])],
                  nh, q_prescaled=True)
    return o


def mlp0():
    o0 = torch.empty(4096, 12288, device=dev, dtype=torch.bfloat16)
    o1 = torch.empty(260, 12288, device=dev, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(a0, w0, b0, o0, L.EPI_GELU_TANH), ops.Gemm(a1, w1, b0, o1, L.EPI_GELU_TANH)], L.TILE_PP_256x256)
    return torch.cat((o0, o1))


def linear2():
    x = torch.zeros(4352, 3072, device=dev, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(al, wl, None, x, L.EPI_GATE_RESIDUAL, resid=x, gate=gate)], L.TILE_PP_256x256)
    return x


def fp8():
    o = torch.empty(4096, 12288, device=dev, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(qa, qw, b0, o, a_scale=sa, w_scale=sw)])
    return o


at, wt = rnd(1300, 12288), rnd(3072, 12288, scale=0.02)
xt0 = torch.randn(1300, 3072, device=dev)


def thin_rows():  # 5 full row tiles in the ping-pong walk + 20 rows in the thin-row kernel (4-slot LDS-DMA ring, K = 12288)
    x = xt0.clone()
    ops.gemm([ops.Gemm(at, wt, None, x, L.EPI_GATE_RESIDUAL, resid=x, gate=gate)], L.TILE_PP_256x256)
    return x


# (round 5) the one-wave thin tiles with their 8-slot ring at K = 3072 with a GELU epilogue, the low-plane q projection with
# the finish fused into its epilogue (full tiles + a thin last tile + the 4-wave thin form for the head epilogue), and the
# fused heat-map launch (no hand-off of its own; here for the record)
ag, wg, bg = rnd(1300, 3072), rnd(12288, 3072, scale=0.02), rnd(12288)


def thin_rows_gelu():
    o = torch.empty(1300, 12288, device=dev, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(ag, wg, bg, o, L.EPI_GELU_TANH)], L.TILE_PP_256x256)
    return o


Mq = 4096 + 20
lo_plane, wq = rnd(Mq, 3072, scale=0.01), rnd(3072, 3072, scale=0.02)
nq = (0.5 + torch.rand(128, device=dev)).bfloat16()
rope = torch.randn(Mq, 64, 2, device=dev)
rope = (rope / rope.norm(dim=-1, keepdim=True)).contiguous()
qraw0 = torch.randn(Mq, 3072, device=dev)


def lo_q_fused():
    qpre = qraw0.clone()
    qout = torch.zeros(Mq, 3072, device=dev, dtype=torch.bfloat16)
    kw = dict(epilogue=L.EPI_QKV_NORM_ROPE, n_split=9216, norm_q=nq, norm_k=nq, q_out_scale=0.1275, qpre_add=True, qk_f16=True)
    ops.gemm([ops.Gemm(lo_plane[:4096], wq, None, qout[:4096], rope=rope[:4096], q_prerope=qpre[:4096], **kw),
              ops.Gemm(lo_plane[4096:], wq, None, qout[4096:], rope=rope[4096:], q_prerope=qpre[4096:], **kw)],
             L.TILE_PP_256x256)
    return torch.cat((qpre.view(torch.int32).flatten(), qout.view(torch.int16).flatten().int()))   # (bit patterns)


img_v, con_v = torch.randn(5, 4096, 3072, device=dev), torch.randn(5, 4, 3072, device=dev) * 0.05
img_b = img_v.bfloat16()


def heat_fused():
    acc = torch.zeros(10, 2, 4, 4096, device=dev)   # (a launch's problems must not share an accumulator)
    ops.heatmap_fused([ops.Heatmap(img_v[j], con_v[j], acc[j, 0], 0.25, acc[j, 1], 1.0) for j in range(5)] +
                      [ops.Heatmap(img_b[j], con_v[j], acc[5 + j, 0], 0.25) for j in range(5)])
    return acc


cases = {"thin rows, one-wave tiles (mlp.0 shape, GELU)": thin_rows_gelu, "low-plane q projection, fused finish": lo_q_fused,
         "fused heat maps (10 problems)": heat_fused, "thin rows (mlp.2 shape)": thin_rows, "attention": attn, "attention (pre-scaled q: ca_attn4_kernel)": attn4, "mlp0 (grouped, persistent)": mlp0, "linear2 (K=15360)": linear2, "fp8 gemm": fp8}
first = {name: fn() for name, fn in cases.items()}
torch.cuda.synchronize()
bad = {name: 0 for name in cases}
t0 = time.time()
names = list(cases)
for it in range(iters):
    outs = []
    for j, name in enumerate(names):
        st = s1 if (it + j) % 2 == 0 else s2
        with torch.cuda.stream(st):
            outs.append((name, cases[name]()))
    torch.cuda.synchronize()
    for name, o in outs:
        if not torch.equal(o, first[name]):
            bad[name] += 1
    if it % 50 == 49:
        print(f"iteration {it + 1}: mismatches so far {bad} ({time.time() - t0:.0f} s)", flush=True)
print("RESULT", "clean" if not any(bad.values()) else f"MISMATCHES {bad}")
sys.exit(1 if any(bad.values()) else 0)
