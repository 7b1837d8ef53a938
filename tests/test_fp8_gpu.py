"""The reduced-precision (fp8 e4m3) kernels behind BASELINE.json configs[4] ("fp8 MFMA").  The reference has
no fp8 path ("fp8 has no oracle -> own tolerance study"), so these tests pin the KERNELS to
exact statements of what they compute: the quantiser against torch's own float8_e4m3fn cast of the same
scaled values, the GEMM against an fp32 product of the DEQUANTISED operands (so the only difference left is
fp32 summation order and the bf16 rounding of the output)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from conceptattention_amd import _lib as L  # noqa: E402
from conceptattention_amd import ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).bfloat16()


def deq(q_u8, scale):
    """e4m3 bytes + row scales -> fp32 (decoded on the CPU by torch)."""
    return q_u8.cpu().view(torch.float8_e4m3fn).float() * scale.cpu()[:, None]


@pytest.mark.parametrize("M,K", [(5, 64), (260, 3072), (33, 15360)])
def test_quantize_rows_matches_e4m3fn_cast(M, K):
    x = rnd(M, K, scale=3.0)
    x[0, :] = 0                      # an all-zero row must not divide by zero
    x[1, 7] = 1000.0                 # an outlier sets the row scale
    q, s = ops.quantize_rows_fp8(x)
    xs = x.float().cpu()
    amax = xs.abs().amax(dim=1)
    want_s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(s.cpu(), want_s, rtol=1e-6, atol=0)
    # same bytes as torch's OCP e4m3fn round-to-nearest-even of x / scale (1/scale is formed the same way)
    want_q = (xs * (1.0 / s.cpu())[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    same = (q.cpu() == want_q)
    # +0 / -0 differ in the sign bit only
    zero_pair = ((q.cpu() & 0x7F) == 0) & ((want_q & 0x7F) == 0)
    assert bool((same | zero_pair).all()), f"{(~(same | zero_pair)).sum().item()} bytes differ"
    assert (deq(q, s) - xs).abs().max() <= amax.max() / 448.0 * 16.0  # half a step of the top binade


def test_ln_modulate_fp8_equals_quantised_bf16_path():
    M, H = 300, 3072
    x = rnd(M, H, scale=2.0)
    g = torch.Generator().manual_seed(3)
    sh = [torch.randn(H, generator=g).to(DEV) * 0.3 for _ in range(2)]
    sc = [torch.randn(H, generator=g).to(DEV) * 0.3 for _ in range(2)]
    segs = [(40, sh[0], sc[0]), (M, sh[1], sc[1])]
    q = torch.empty(M, H, device=DEV, dtype=torch.uint8)
    s = torch.empty(M, device=DEV)
    ops.ln_modulate(x, q, segs, out_scale=s)
    # fp32 reference of the modulated row, then the same absmax/e4m3 quantisation
    xf = x.float()
    y = torch.nn.functional.layer_norm(xf, (H,), eps=1e-6)
    y[:40] = (1 + sc[0]) * y[:40] + sh[0]
    y[40:] = (1 + sc[1]) * y[40:] + sh[1]
    amax = y.abs().amax(dim=1)
    assert torch.allclose(s, amax / 448.0, rtol=2e-5)
    err = (deq(q, s) - y.cpu()).abs()
    # one e4m3 step is 2^-3 relative; allow one full step (fp32 LN rounding can flip a rounding decision)
    assert bool((err <= 0.125 * y.cpu().abs() + (amax.cpu() / 448.0 * 2 ** -6)[:, None] + 1e-6).all())


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (260, 512, 384), (4352, 768, 3072)])
def test_gemm_fp8_bias_vs_dequantised_product(M, N, K):
    a, w, b = rnd(M, K), rnd(N, K, scale=0.05), rnd(N)
    qa, sa = ops.quantize_rows_fp8(a)
    qw, sw = ops.quantize_rows_fp8(w)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(qa, qw, b, out, a_scale=sa, w_scale=sw)])
    ref = deq(qa, sa).double() @ deq(qw, sw).double().t() + b.cpu().double()
    err = (out.cpu().double() - ref).abs()
    assert bool((err <= 2e-2 + 8e-3 * ref.abs()).all()), f"max err {err.max().item():.4g}"
    # and the quantisation itself costs a few percent against the bf16 operands (reported, not a bound on parity)
    full = a.cpu().double() @ w.cpu().double().t() + b.cpu().double()
    rel = (ref - full).norm() / full.norm()
    assert rel < 0.06, rel


def test_gemm_fp8_identity_asymmetric():
    """Exact small integers: catches any k-order mismatch between the two operands' fragments."""
    M, N, K = 256, 256, 256
    a = ((torch.arange(M)[:, None] * 7 + torch.arange(K)[None, :] * 3) % 5 - 2).float()
    w = ((torch.arange(N)[:, None] * 5 + torch.arange(K)[None, :] * 11) % 7 - 3).float()
    qa = a.to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    qw = w.to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    one_m, one_n = torch.ones(M, device=DEV), torch.ones(N, device=DEV)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(qa, qw, None, out, a_scale=one_m, w_scale=one_n)])
    ref = a @ w.t()
    assert torch.equal(out.float().cpu(), ref.bfloat16().float())


def test_gemm_fp8_grouped_gate_residual_and_gelu():
    M0, M1, N, K = 512, 260, 512, 256
    outs, refs = [], []
    probs = []
    for M, seed in ((M0, 1), (M1, 2)):
        a, w, b = rnd(M, K, seed=seed), rnd(N, K, scale=0.05, seed=seed), rnd(N, seed=seed)
        qa, sa = ops.quantize_rows_fp8(a)
        qw, sw = ops.quantize_rows_fp8(w)
        resid = rnd(M, N, seed=seed + 5)
        gate = torch.randn(N, device=DEV)
        gate2 = torch.randn(N, device=DEV)
        out = resid.clone()
        probs.append(ops.Gemm(qa, qw, b, out, L.EPI_GATE_RESIDUAL, resid=out, gate=gate, gate2=gate2, gate_rows=100,
                              a_scale=sa, w_scale=sw))
        lin = deq(qa, sa).double() @ deq(qw, sw).double().t() + b.cpu().double()
        g = torch.where(torch.arange(M)[:, None] < 100, gate.cpu().double()[None], gate2.cpu().double()[None])
        refs.append(resid.cpu().double() + g * lin)
        outs.append(out)
    ops.gemm(probs)
    for out, ref in zip(outs, refs):
        err = (out.cpu().double() - ref).abs()
        assert bool((err <= 3e-2 + 8e-3 * ref.abs()).all()), err.max()
    a, w, b = rnd(300, 256), rnd(512, 256, scale=0.05), rnd(512)
    qa, sa = ops.quantize_rows_fp8(a)
    qw, sw = ops.quantize_rows_fp8(w)
    out = torch.empty(300, 512, device=DEV, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(qa, qw, b, out, L.EPI_GELU_TANH, a_scale=sa, w_scale=sw)])
    lin = deq(qa, sa).double() @ deq(qw, sw).double().t() + b.cpu().double()
    ref = torch.nn.functional.gelu(lin, approximate="tanh")
    err = (out.cpu().double() - ref).abs()
    assert bool((err <= 2e-2 + 8e-3 * ref.abs()).all()), err.max()


def test_gemm_fp8_argument_errors():
    a = torch.zeros(256, 128, device=DEV, dtype=torch.uint8)
    w = torch.zeros(256, 128, device=DEV, dtype=torch.uint8)
    out = torch.empty(256, 256, device=DEV, dtype=torch.bfloat16)
    s = torch.ones(256, device=DEV)
    with pytest.raises(ValueError):
        ops.gemm([ops.Gemm(a, w, None, out)])                         # scales missing
    with pytest.raises(ValueError):
        ops.gemm([ops.Gemm(a[:, :64], w[:, :64], None, out, a_scale=s, w_scale=s)])   # K % 128
    with pytest.raises(ValueError):
        ops.gemm([ops.Gemm(a, w, None, out, a_scale=s, w_scale=s)], L.TILE_PP_256x128)
