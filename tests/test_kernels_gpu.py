"""Each gfx950 kernel (called through the C ABI) against an fp32 reference of the same op on the
same bf16-representable inputs.  Tolerances are stated per test: outputs are bf16 (8-bit mantissa,
relative quantum 2^-8 = 3.9e-3), accumulation is fp32."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from conceptattention_amd import _lib as L  # noqa: E402
from conceptattention_amd import ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).bfloat16()


def close(out, ref, atol, rtol=8e-3):
    """|out-ref| <= atol + rtol*|ref| elementwise (rtol = 2 bf16 ulps)."""
    out, ref = out.float(), ref.float()
    err = (out - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    assert not bad.any(), f"max err {err.max().item():.4g} (ref max {ref.abs().max().item():.4g}), {bad.sum().item()} bad"


def test_device_is_gfx950():
    assert L.load().ca_check_device() == 0, L.load().ca_last_error()


# ------------------------------------------------------------------------------------------ GEMM
def test_gemm_identity_asymmetric():
    """A = I with an asymmetric W catches a transposed / permuted C layout exactly."""
    K = N = 256
    a = torch.eye(256, K, device=DEV).bfloat16()
    w = (torch.arange(N, device=DEV)[:, None] * 3 + torch.arange(K, device=DEV)[None, :] * 5) % 251
    w = (w.float() - 125).bfloat16()  # integers < 256: exact in bf16
    w = torch.cat((w, w[:128] + 1))  # N = 384: divisible by every tile width
    N = 384
    for tile in (L.TILE_256x128, L.TILE_256x64, L.TILE_PP_256x192, L.TILE_PP_256x128, L.TILE_256x192):
        out = ops.linear(a, w, None, tile=tile)
        assert torch.equal(out.float(), w.float().t().contiguous()), f"tile {tile}"


@pytest.mark.parametrize("tile,N", [(L.TILE_256x256, 512), (L.TILE_256x192, 384), (L.TILE_256x128, 384),
                                    (L.TILE_256x64, 64), (L.TILE_AUTO, 768), (L.TILE_PP_256x256, 512),
                                    (L.TILE_PP_256x128, 384), (L.TILE_PP_256x192, 576)])
@pytest.mark.parametrize("M,K", [(4, 192), (260, 64), (513, 192), (300, 1024)])
def test_gemm_bias_tails(tile, N, M, K):
    a, w, b = rnd(M, K), rnd(N, K, scale=0.1), rnd(N)
    out = ops.linear(a, w, b, tile=tile)
    ref = a.float() @ w.float().t() + b.float()
    close(out, ref, atol=2e-2)


@pytest.mark.parametrize("tile,N", [(L.TILE_256x256, 512), (L.TILE_PP_256x256, 512), (L.TILE_PP_256x128, 512),
                                    (L.TILE_PP_256x192, 768)])
def test_gemm_epilogues_and_strides(tile, N):
    M, K = 300, 320
    big = rnd(M, K + 64)
    a = big[:, 32:32 + K]  # row stride K+64, 16-byte aligned offset
    w, b = rnd(N, K, scale=0.1), rnd(N)
    lin = a.float() @ w.float().t() + b.float()
    out = ops.linear(a, w, b, epilogue=L.EPI_GELU_TANH, tile=tile)
    close(out, torch.nn.functional.gelu(lin, approximate="tanh"), atol=2e-2)
    # gate * x + residual with two gate vectors (rows < 7 use gate, the rest gate2), in place
    resid = rnd(M, N)
    g1, g2 = rnd(N).float(), rnd(N, seed=5).float()
    ref = resid.float() + torch.where(torch.arange(M, device=DEV)[:, None] < 7, g1[None], g2[None]) * lin
    x = resid.clone()
    ops.gemm([ops.Gemm(a, w, b, x, L.EPI_GATE_RESIDUAL, resid=x, gate=g1, gate2=g2, gate_rows=7)], tile)
    close(x, ref, atol=3e-2)
    # split: first ns columns plain, the rest GELU into a strided destination
    ns = 384 if N == 768 else 256
    o1 = torch.zeros(M, ns, device=DEV, dtype=torch.bfloat16)
    cat = torch.zeros(M, 128 + N - ns, device=DEV, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(a, w, b, o1, L.EPI_SPLIT_GELU, out2=cat[:, 128:], n_split=ns)], tile)
    close(o1, lin[:, :ns], atol=2e-2)
    close(cat[:, 128:], torch.nn.functional.gelu(lin[:, ns:], approximate="tanh"), atol=2e-2)
    assert cat[:, :128].abs().max() == 0


@pytest.mark.parametrize("tile", [L.TILE_AUTO, L.TILE_256x256, L.TILE_PP_256x256, L.TILE_PP_256x128, L.TILE_PP_256x192])
def test_gemm_grouped_two_problems(tile):
    a0, w0, b0 = rnd(700, 256), rnd(768, 256, scale=0.1), rnd(768)
    n1 = 384 if tile == L.TILE_PP_256x192 else 512
    a1, w1, b1 = rnd(260, 128), rnd(n1, 128, scale=0.1), rnd(n1)
    o0 = torch.empty(700, 768, device=DEV, dtype=torch.bfloat16)
    o1 = torch.empty(260, n1, device=DEV, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(a0, w0, b0, o0), ops.Gemm(a1, w1, b1, o1, L.EPI_GELU_TANH)], tile)
    close(o0, a0.float() @ w0.float().t() + b0.float(), atol=2e-2)
    close(o1, torch.nn.functional.gelu(a1.float() @ w1.float().t() + b1.float(), approximate="tanh"), atol=2e-2)


@pytest.mark.parametrize("tail", [0, 512])
def test_gemm_qkv_norm_rope_epilogue(tail):
    """qkv projection with QK-RMSNorm + RoPE fused (and, for the single blocks, a GELU'd mlp tail):
    compared with the oracle's fp32 rms_norm / apply_rope on the fp32 projection."""
    from oracle import flux_oracle as O
    nh, M, K = 2, 300, 192
    Hd = nh * 128
    N = 3 * Hd + tail
    a, w, b = rnd(M, K), rnd(N, K, scale=0.1), rnd(N)
    ids = torch.zeros(M, 3)
    ids[20:, 1] = torch.arange(M - 20) // 16
    ids[20:, 2] = torch.arange(M - 20) % 16
    cos, sin = O.rope_cos_sin(ids, (16, 56, 56), 10000)
    table = torch.stack((cos, sin), -1).contiguous().to(DEV)
    nq, nk = (0.5 + torch.rand(128)).bfloat16().to(DEV), (0.5 + torch.rand(128)).bfloat16().to(DEV)
    out = torch.zeros(M, 3 * Hd, device=DEV, dtype=torch.bfloat16)
    pre = torch.zeros(M, Hd, device=DEV, dtype=torch.bfloat16)
    out2 = torch.zeros(M, max(tail, 8), device=DEV, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(a, w, b, out, L.EPI_QKV_NORM_ROPE, n_split=3 * Hd, norm_q=nq, norm_k=nk, rope=table,
                       q_prerope=pre, out2=out2 if tail else None)])
    lin = (a.float() @ w.float().t() + b.float()).cpu()
    qkv = lin[:, :3 * Hd].view(M, 3, nh, 128)
    for which, sc in ((0, nq), (1, nk)):
        normed = O.rms_norm(qkv[:, which], sc.float().cpu())
        if which == 0:
            close(pre.cpu(), normed.reshape(M, -1), atol=1e-2)
        roped = O.apply_rope(normed.permute(1, 0, 2)[None], cos, sin)[0].permute(1, 0, 2)
        close(out.cpu().view(M, 3, nh, 128)[:, which], roped, atol=1e-2)
    close(out.cpu().view(M, 3, nh, 128)[:, 2], qkv[:, 2], atol=2e-2)
    if tail:
        close(out2.cpu(), torch.nn.functional.gelu(lin[:, 3 * Hd:], approximate="tanh"), atol=2e-2)
    # fp32 pre-RoPE q (the cross-attention-space vectors without their bf16 rounding) and the rotated q pre-multiplied
    # by softmax_scale * log2(e) in fp32 before its one rounding; k, v and the bf16 form of q_prerope unchanged.
    # M = 300 = one full row tile + 44 rows: both the ping-pong and the thin-row kernel's epilogue are exercised.
    qs = 0.088388347648 * 1.4426950408889634
    out_s = torch.zeros_like(out)
    pre32 = torch.zeros(M, Hd, device=DEV, dtype=torch.float32)
    ops.gemm([ops.Gemm(a, w, b, out_s, L.EPI_QKV_NORM_ROPE, n_split=3 * Hd, norm_q=nq, norm_k=nk, rope=table,
                       q_prerope=pre32, out2=out2 if tail else None, q_out_scale=qs)])
    assert torch.equal(pre32.bfloat16(), pre)                        # the same values, rounded once later
    assert (pre32.cpu() - O.rms_norm(qkv[:, 0], nq.float().cpu()).reshape(M, -1)).abs().max() < 2e-5 * 40
    assert torch.equal(out_s[:, Hd:], out[:, Hd:])                   # k and v thirds are not scaled
    roped_q = O.apply_rope(O.rms_norm(qkv[:, 0], nq.float().cpu()).permute(1, 0, 2)[None], cos, sin)[0].permute(1, 0, 2)
    close(out_s.cpu().view(M, 3, nh, 128)[:, 0], roped_q * qs, atol=2e-3)
    with pytest.raises(ValueError):  # only the 256x256 ping-pong tile carries this epilogue
        ops.gemm([ops.Gemm(a, w, b, out, L.EPI_QKV_NORM_ROPE, n_split=3 * Hd, norm_q=nq, norm_k=nk, rope=table,
                           out2=out2 if tail else None)], L.TILE_PP_256x128)


@pytest.mark.parametrize("tile,N", [(L.TILE_PP_256x256, 768), (L.TILE_PP_256x192, 384), (L.TILE_PP_256x128, 256)])
@pytest.mark.parametrize("rem", [1, 20, 32, 44, 128])
def test_gemm_thin_last_row_tile_is_bit_identical(tile, N, rem):
    """A last row tile with few valid rows (<= 128) leaves the ordinary tile walk: under the bf16 256x256 tile it goes
    to the thin-row kernel (32 x 128 tiles, its own launch), otherwise it takes a copy of the ping-pong K loop
    without the MFMAs and LDS reads of row fragments past M and is walked last.  Its rows must come out bit for bit as when the same rows sit in a FULL row tile of a longer
    problem, for every epilogue -- the k order per accumulator is the contract."""
    K, Mfull = 448, 512
    a, w, b = rnd(Mfull, K), rnd(N, K, scale=0.1), rnd(N)
    M = 256 + rem
    for epi in (L.EPI_BIAS, L.EPI_GELU_TANH):
        full = ops.linear(a, w, b, epilogue=epi, tile=tile)
        thin = ops.linear(a[:M], w, b, epilogue=epi, tile=tile)
        assert torch.equal(thin, full[:M]), f"epilogue {epi}"
    # gate * x + residual on an fp32 and on a bf16 stream, per-row gate vectors, in place
    g1, g2 = rnd(N).float(), rnd(N, seed=5).float()
    for dt in (torch.float32, torch.bfloat16):
        resid = rnd(Mfull, N).to(dt)
        xf, xt = resid.clone(), resid[:M].clone()
        ops.gemm([ops.Gemm(a, w, b, xf, L.EPI_GATE_RESIDUAL, resid=xf, gate=g1, gate2=g2, gate_rows=260)], tile)
        ops.gemm([ops.Gemm(a[:M], w, b, xt, L.EPI_GATE_RESIDUAL, resid=xt, gate=g1, gate2=g2, gate_rows=260)], tile)
        assert torch.equal(xt, xf[:M]), f"gated, {dt}"
    if tile == L.TILE_PP_256x256:  # fused QK-norm + RoPE (N = 768: two heads' worth of q, k, v)
        table = torch.randn(Mfull, 64, 2, device=DEV)
        nq, nk = (0.5 + torch.rand(128)).bfloat16().to(DEV), (0.5 + torch.rand(128)).bfloat16().to(DEV)
        outs = []
        for rows in (Mfull, M):
            out = torch.zeros(rows, N, device=DEV, dtype=torch.bfloat16)
            pre = torch.zeros(rows, N // 3, device=DEV, dtype=torch.bfloat16)
            ops.gemm([ops.Gemm(a[:rows], w, b, out, L.EPI_QKV_NORM_ROPE, n_split=N, norm_q=nq, norm_k=nk,
                               rope=table[:rows].contiguous(), q_prerope=pre)], tile)
            outs.append((out, pre))
        assert torch.equal(outs[1][0], outs[0][0][:M]) and torch.equal(outs[1][1], outs[0][1][:M])
    # and two problems in one launch, both with a thin last row tile (the walk puts all thin tiles last)
    a1, w1, b1 = rnd(rem, K), rnd(N, K, scale=0.1, seed=3), rnd(N, seed=3)
    o0 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    o1 = torch.empty(rem, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(a[:M], w, b, o0), ops.Gemm(a1, w1, b1, o1, L.EPI_GELU_TANH)], tile)
    assert torch.equal(o0, ops.linear(a, w, b, tile=tile)[:M])
    close(o1, torch.nn.functional.gelu(a1.float() @ w1.float().t() + b1.float(), approximate="tanh"), atol=2e-2)


def test_gemm_rejects_bad_arguments():
    a, w = rnd(16, 100), rnd(256, 100)
    with pytest.raises(ValueError):
        ops.linear(a, w, None)  # K % 64 != 0
    with pytest.raises(ValueError):
        ops.linear(rnd(16, 128), rnd(100, 128), None)  # N not a multiple of any tile width


def test_attention_repeatable_and_persistent_gemm_multi_round():
    """The attention kernel hands K/V tiles to its waves by LDS-DMA behind one barrier per tile, and the GEMM walks
    several rounds of tiles per workgroup: both must give the same bits launch after launch (a race shows up
    as a rare differing tile), here at the full Flux sizes with the concept rows riding along."""
    nh, C, n = 24, 4, 4352
    buf = rnd(C + n, 9216)
    q, k, v = buf[:, :3072], buf[:, 3072:6144], buf[:, 6144:]
    probs = lambda o: [ops.Attn(q[C:], o[C:], k[C:], v[C:]),
                       ops.Attn(q[:C], o[:C], k[:C], v[:C], k[C + 256:], v[C + 256:])]
    for kw in ({}, {"q_prescaled": True}):          # ca_attn_kernel, then ca_attn4_kernel (q values merely differ in scale)
        first = torch.zeros(C + n, 3072, device=DEV, dtype=torch.bfloat16)
        ops.attention(probs(first), nh, **kw)
        for i in range(10):
            out = torch.zeros_like(first)
            ops.attention(probs(out), nh, **kw)
            assert torch.equal(out, first), f"attention launch {i} differs ({kw})"
    # 864 tiles in one grouped launch = 3.4 rounds on 256 CUs (the mlp.0 launch of a double block)
    a0, a1 = rnd(4096, 3072), rnd(260, 3072, seed=1)
    w0, w1 = rnd(12288, 3072, scale=0.02), rnd(12288, 3072, scale=0.02, seed=1)
    b0, b1 = rnd(12288), rnd(12288, seed=1)
    def mlp0():
        o0 = torch.empty(4096, 12288, device=DEV, dtype=torch.bfloat16)
        o1 = torch.empty(260, 12288, device=DEV, dtype=torch.bfloat16)
        ops.gemm([ops.Gemm(a0, w0, b0, o0, L.EPI_GELU_TANH), ops.Gemm(a1, w1, b1, o1, L.EPI_GELU_TANH)],
                 L.TILE_PP_256x256)
        return o0, o1
    f0, f1 = mlp0()
    ref = torch.nn.functional.gelu(a1.float() @ w1.float().t() + b1.float(), approximate="tanh")
    close(f1, ref, atol=2e-2)
    for i in range(5):
        o0, o1 = mlp0()
        assert torch.equal(o0, f0) and torch.equal(o1, f1), f"grouped launch {i} differs"


def test_gemm_pipelined_kernel_repeatable_under_load():
    """The ping-pong kernel hands tiles between waves through LDS-DMA + counted waits; a protocol
    slip shows up as rare wrong tiles, so compare many back-to-back launches bit for bit and
    against fp32 (different shapes keep every CU busy with uneven neighbours)."""
    M, N, K = 4096, 3072, 3072
    a, w = rnd(M, K), rnd(N, K, scale=0.02)
    ref = a.float() @ w.float().t()
    first = ops.linear(a, w, None, tile=L.TILE_PP_256x256)
    close(first, ref, atol=3e-2)
    for i in range(20):
        out = ops.linear(a, w, None, tile=(L.TILE_PP_256x256, L.TILE_PP_256x128, L.TILE_PP_256x192)[i % 3])
        assert torch.equal(out, first), f"launch {i} differs"


@pytest.mark.parametrize("tile", [L.TILE_256x256, L.TILE_PP_256x256])
def test_gemm_full_size_flux_shapes(tile):
    """M=4352 (T+L), the single-block linear1 split: N=21504 -> qkv 9216 | mlp 12288, K=3072."""
    M, K = 4352, 3072
    a, w, b = rnd(M, K), rnd(21504, K, scale=0.02), rnd(21504)
    qkv = torch.empty(M, 9216, device=DEV, dtype=torch.bfloat16)
    cat = torch.empty(M, 15360, device=DEV, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(a, w, b, qkv, L.EPI_SPLIT_GELU, out2=cat[:, 3072:], n_split=9216)], tile)
    ref = a.float() @ w.float().t() + b.float()
    close(qkv, ref[:, :9216], atol=3e-2)
    close(cat[:, 3072:], torch.nn.functional.gelu(ref[:, 9216:], approximate="tanh"), atol=3e-2)


# ------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, nh):
    """fp32 softmax(q k^T / sqrt(128)) v on [rows, nh*128] views."""
    qh = q.float().view(q.shape[0], nh, 128).transpose(0, 1)
    kh = k.float().view(k.shape[0], nh, 128).transpose(0, 1)
    vh = v.float().view(v.shape[0], nh, 128).transpose(0, 1)
    w = torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(128), dim=-1)
    return (w @ vh).transpose(0, 1).reshape(q.shape[0], nh * 128)


SL2 = (1.0 / math.sqrt(128.0)) * 1.4426950408889634


def attn_forms(q, prescaled):
    """(q to hand to the kernel, q the kernel effectively uses, kwargs): with ``prescaled`` the q rows carry
    softmax_scale * log2(e) (CA_ATTN_Q_PRESCALED -> ca_attn4_kernel, the model path's kernel); otherwise the kernel
    scales (ca_attn_kernel)."""
    if not prescaled:
        return q, q, {}
    qp = (q.float() * SL2).bfloat16()
    return qp, qp.float() / SL2, {"q_prescaled": True}


PRESCALED = pytest.mark.parametrize("prescaled", [False, True], ids=["scaling-kernel", "prescaled-attn4"])


@PRESCALED
@pytest.mark.parametrize("nq,nk", [(300, 200), (32, 64), (7, 1), (513, 1000), (70, 64), (64, 129), (257, 448)])
def test_attention_ragged(nq, nk, prescaled):
    nh = 3
    qkv_q = rnd(nq, 3 * nh * 128)
    qkv_k = rnd(nk, 3 * nh * 128, seed=1)
    q, k, v = qkv_q[:, :nh * 128], qkv_k[:, nh * 128:2 * nh * 128], qkv_k[:, 2 * nh * 128:]
    out = torch.zeros(nq, nh * 128, device=DEV, dtype=torch.bfloat16)
    qk, qe, kw = attn_forms(q, prescaled)
    ops.attention([ops.Attn(qk, out, k, v)], nh, **kw)
    close(out, attn_ref(qe, k, v, nh), atol=1e-2)


def test_attention_two_segments_and_two_problems():
    """Concept rows attend to [concept keys ; image keys]; main rows to [text ; image] in one launch."""
    nh, C, T, Limg = 2, 5, 40, 333
    buf = rnd(C + T + Limg, 3 * nh * 128, scale=1.5)
    H = nh * 128
    q, k, v = buf[:, :H], buf[:, H:2 * H], buf[:, 2 * H:]
    out = torch.zeros(C + T + Limg, H, device=DEV, dtype=torch.bfloat16)
    out32 = torch.zeros(C, H, device=DEV)
    main = ops.Attn(q[C:], out[C:], k[C:], v[C:])
    con = ops.Attn(q[:C], out[:C], k[:C], v[:C], k[C + T:], v[C + T:], out_f32=out32)
    ops.attention([main, con], nh)
    close(out[C:], attn_ref(q[C:], k[C:], v[C:], nh), atol=1e-2)
    kc, vc = torch.cat((k[:C], k[C + T:])), torch.cat((v[:C], v[C + T:]))
    close(out[:C], attn_ref(q[:C], kc, vc, nh), atol=1e-2)
    assert torch.equal(out32.bfloat16(), out[:C])  # the fp32 copy rounds to the bf16 output
    # only P's bf16 rounding is left in the fp32 copy: relative 2^-9 of |O| (<= 3 here).  With the deferred
    # rescale a row's dominant weight is no longer exactly 1.0, so these peaky rows (inputs scaled by 1.5)
    # see that bound in full; at the model's own statistics the rows agree to 2.6e-4 (test_model_gpu.py)
    assert (out32 - attn_ref(q[:C], kc, vc, nh)).abs().max() < 8e-3


@PRESCALED
@pytest.mark.parametrize("gain", [4.0, 9.0])
def test_attention_online_softmax_rescale_spike(gain, prescaled):
    """A key far down the sequence dominates one query row: the running max must jump at that tile.  gain 4 = ~65
    octaves above the row's tile-0 reference: ca_attn_kernel redoes the tile, ca_attn4_kernel (no per-tile check) sees
    a row sum > 2^60 at the end and recomputes the workgroup's rows the classical way; gain 9 = ~146 octaves: the
    kept-reference pass overflows to inf on the way, which the same final check catches."""
    nh, nq, nk = 1, 300, 640
    q, k, v = rnd(nq, 128), rnd(nk, 128, seed=3), rnd(nk, 128, seed=4)
    k[500] = (q[17].float() * gain).bfloat16()  # logit ~ gain*|q|^2/sqrt(128) >> others
    k[70] = (q[290].float() * gain).bfloat16()  # a second workgroup, an early tile
    out = torch.zeros(nq, 128, device=DEV, dtype=torch.bfloat16)
    qk, qe, kw = attn_forms(q, prescaled)
    ops.attention([ops.Attn(qk, out, k, v)], nh, **kw)
    close(out, attn_ref(qe, k, v, nh), atol=1e-2)


@PRESCALED
@pytest.mark.parametrize("alpha", [1.0, 1.45, 1.9, 3.6])    # logit gap to the tile-0 reference ~ 16 / 24 / 31 / 59 octaves
@pytest.mark.parametrize("where", ["late_full_tile", "masked_tail_tile", "segment_1", "segment_1_tail"])
def test_attention_kept_reference_band_spikes(alpha, where, prescaled):
    """The softmax reference of a row is set by tile 0 and then KEPT; a tile is redone only when a partial row sum
    exceeds 2^30.  These spikes sit in the band the 65-octave spike test above does NOT reach: a late key 16-31
    octaves above the tile-0 reference (alpha * |q|^2 / sqrt(128) * log2 e), i.e. P up to ~2^29 WITHOUT a redo
    (alpha 1.0, 1.45) and just across the redo limit (1.9) -- in a full tile, in the ragged (masked) last tile, and
    in the second key segment (its full and its ragged tile).  fp32 reference, the tolerance of the other tests."""
    nh, nq = 1, 96
    q0 = rnd(nq, 128)
    q, qe, kw = attn_forms(q0, prescaled)      # (the spikes are built from the unscaled q0)
    if where in ("late_full_tile", "masked_tail_tile"):
        nk = 640 if where == "late_full_tile" else 650           # 650: tile 10 has 10 valid keys
        k, v = rnd(nk, 128, seed=3), rnd(nk, 128, seed=4)
        pos = 500 if where == "late_full_tile" else 645
        k[pos] = (q0[17].float() * alpha).bfloat16()
        k[pos - 130] = (q0[70].float() * alpha * 0.9).bfloat16()   # a second row, another wave, another tile
        probs, kk, vv = (lambda o: [ops.Attn(q, o, k, v)]), k, v
    else:
        n0, n1 = 100, 540 if where == "segment_1" else 533       # 633 keys: the last tile is ragged AND in segment 1
        k0, v0, k1, v1 = rnd(n0, 128, seed=3), rnd(n0, 128, seed=4), rnd(n1, 128, seed=5), rnd(n1, 128, seed=6)
        pos = 400 if where == "segment_1" else n1 - 3
        k1[pos] = (q0[17].float() * alpha).bfloat16()
        k1[7] = (q0[70].float() * alpha * 0.9).bfloat16()          # tile 1 = the tile that straddles the two segments
        probs, kk, vv = (lambda o: [ops.Attn(q, o, k0, v0, k1, v1)]), torch.cat((k0, k1)), torch.cat((v0, v1))
    out = torch.zeros(nq, 128, device=DEV, dtype=torch.bfloat16)
    ops.attention(probs(out), nh, **kw)
    ref = attn_ref(qe, kk, vv, nh)
    close(out, ref, atol=1e-2)
    # the spiked rows are (nearly) one-hot on the spike's value row: check them against it directly as well
    s17 = (qe[17].float() @ kk.float().t() / math.sqrt(128)).softmax(0)
    assert s17.max() > 0.95
    if s17.max() > 0.995:
        assert (out[17].float() - vv[int(s17.argmax())].float()).abs().max() < 5e-2


@pytest.mark.parametrize("C,T,Limg", [(5, 40, 333), (4, 64, 256), (1, 7, 70), (3, 8, 256), (8, 512, 1024)])
def test_attention_prescaled_q_two_segments_two_problems(C, T, Limg):
    """CA_ATTN_Q_PRESCALED: q rows that already carry softmax_scale * log2(e) (written so by the qkv epilogue); the
    kernel's probability is a bare exp2 and every tile's K Q^T chain starts from -reference.  Same result as the
    scaling kernel on the unscaled q up to q's one rounding; full, ragged and segment-straddling tiles, the concept
    problem with its fp32 output copy, and a spike that forces the redo path."""
    nh = 2
    H = nh * 128
    buf = rnd(C + T + Limg, 3 * H, scale=1.5)
    q, k, v = buf[:, :H], buf[:, H:2 * H], buf[:, 2 * H:]
    k[C + T + Limg // 2, :128] = (q[C + 3, :128].float() * 1.6).bfloat16()    # ~27 octaves above its row's reference
    sl2 = (1.0 / math.sqrt(128.0)) * 1.4426950408889634
    qp = (q.float() * sl2).bfloat16()
    out = torch.zeros(C + T + Limg, H, device=DEV, dtype=torch.bfloat16)
    out32 = torch.zeros(C, H, device=DEV)
    main = ops.Attn(qp[C:C + T], out[C:C + T], k[C:C + T], v[C:C + T], k[C + T:], v[C + T:], q1=qp[C + T:],
                    out1=out[C + T:])
    con = ops.Attn(qp[:C], out[:C], k[:C], v[:C], k[C + T:], v[C + T:], out_f32=out32)
    ops.attention([con, main], nh, q_prescaled=True)
    qe = (qp.float() / sl2)                     # the q the kernel effectively used (exact reference for ITS inputs)
    close(out[C:], attn_ref(qe[C:], k[C:], v[C:], nh), atol=1e-2)
    kc, vc = torch.cat((k[:C], k[C + T:])), torch.cat((v[:C], v[C + T:]))
    close(out[:C], attn_ref(qe[:C], kc, vc, nh), atol=1e-2)
    assert torch.equal(out32.bfloat16(), out[:C])
    # against the unscaled path on the original q: only q's rounding differs (2^-9 relative on the logits)
    out_u = torch.zeros_like(out)
    ops.attention([ops.Attn(q[:C], out_u[:C], k[:C], v[:C], k[C + T:], v[C + T:]),
                   ops.Attn(q[C:], out_u[C:], k[C:], v[C:])], nh)
    assert (out.float() - out_u.float()).abs().max() < 0.1
    with pytest.raises(ValueError):
        ops.attention([con], nh, scale=0.1, q_prescaled=True)


def test_split_q_capture_chain_matches_fp32_projection():
    """The cross-attention-space q vectors from the UNROUNDED LayerNorm output: LayerNorm with its low plane
    (ca_ln_modulate_f32in_split), the qkv projection storing q before its norm (qpre_f32 = 2), the q weights applied
    to the low plane, ca_qpre_finish_f32.  Against fp32 rmsnorm(y32 @ Wq^T + b): an order of magnitude closer than the
    one-rounding path, on full rows tiles and on a thin last row tile (M = 300)."""
    from oracle import flux_oracle as O
    nh, M, H = 2, 300, 256
    x = torch.randn(M, H, device=DEV) * 1.5
    sh, sc = torch.randn(H, device=DEV) * 0.2, torch.randn(H, device=DEV) * 0.3
    w, b = rnd(3 * H, H, scale=0.08), rnd(3 * H)
    nq, nk = (0.5 + torch.rand(128)).bfloat16().to(DEV), (0.5 + torch.rand(128)).bfloat16().to(DEV)
    table = torch.zeros(M, 64, 2, device=DEV)
    table[..., 0] = 1.0                                             # identity RoPE
    hi = torch.zeros(M, H, device=DEV, dtype=torch.bfloat16)
    lo = torch.zeros_like(hi)
    ops.ln_modulate(x, hi, [(M, sh, sc)], out_lo=lo)
    y32 = (1 + sc) * torch.nn.functional.layer_norm(x, (H,), eps=1e-6) + sh
    only = torch.zeros_like(hi)
    ops.ln_modulate(x, only, [(M, sh, sc)])
    assert torch.equal(only, hi)                                    # the high plane is the ordinary output
    assert (hi.float() + lo.float() - y32).abs().max() < 3e-5 * 4   # ~16 mantissa bits (|y| < ~8)
    out = torch.zeros(M, 3 * H, device=DEV, dtype=torch.bfloat16)
    qraw = torch.zeros(M, H, device=DEV)
    ops.gemm([ops.Gemm(hi, w, b, out, L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk, rope=table,
                       q_prerope=qraw, qpre_raw=True)])
    out1 = torch.zeros_like(out)
    q1 = torch.zeros(M, H, device=DEV)
    ops.gemm([ops.Gemm(hi, w, b, out1, L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk, rope=table,
                       q_prerope=q1)])
    assert torch.equal(out, out1)                                   # the attention's q, k, v do not change
    d = torch.zeros(M, H, device=DEV)
    ops.gemm([ops.Gemm(lo, w[:H], None, d)])
    plain = qraw.clone()
    ops.qpre_finish(plain, None, nq, nh)
    assert (plain - q1).abs().max() < 2e-5                          # without the correction: the fused epilogue's values
    ops.qpre_finish(qraw, d, nq, nh)
    ref = O.rms_norm((y32.cpu() @ w[:H].float().cpu().t() + b[:H].float().cpu()).view(M, nh, 128),
                     nq.float().cpu()).reshape(M, H)
    e_split, e_one = (qraw.cpu() - ref).abs().max().item(), (q1.cpu() - ref).abs().max().item()
    assert e_split < 2e-4 and e_split < 0.15 * e_one, (e_split, e_one)
    # round 4: the same call also writes the ATTENTION's q (rotated, scaled, half or bf16) from that vector
    rope_t = torch.randn(M, 64, 2, device=DEV)
    rope_t = (rope_t / rope_t.norm(dim=-1, keepdim=True)).contiguous()
    for f16 in (True, False):
        qraw2 = torch.zeros(M, H, device=DEV)
        outq = torch.zeros(M, 3 * H, device=DEV, dtype=torch.bfloat16)
        ops.gemm([ops.Gemm(hi, w, b, outq, L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk, rope=rope_t,
                           q_prerope=qraw2, qpre_raw=True, q_out_scale=SL2, qk_f16=f16)])
        epi_q = outq[:, :H].clone()
        kv_before = outq[:, H:].clone()
        # without the low-plane correction the result is the epilogue's own q (same expressions; the row sums are
        # reduced in another order, so an element may differ by one ulp of its storage type)
        x0 = qraw2.clone()
        ops.qpre_finish(x0, None, nq, nh, rope=rope_t, q_out=outq[:, :H], q_out_scale=SL2, q_f16=f16)
        view = (lambda t_: t_.contiguous().view(torch.float16).float()) if f16 else (lambda t_: t_.float())
        a_, b_ = view(outq[:, :H]), view(epi_q)
        assert (a_ - b_).abs().max() <= (2.0 ** -10 if f16 else 2.0 ** -7) * b_.abs().max()
        assert (a_ != b_).float().mean() < 0.02
        assert torch.equal(outq[:, H:], kv_before)                          # k and v are not touched
        # with it: the fp32 reference of the unrounded projection, rotated and scaled
        ops.qpre_finish(qraw2, d, nq, nh, rope=rope_t, q_out=outq[:, :H], q_out_scale=SL2, q_f16=f16)
        cs, sn = rope_t[..., 0].cpu()[:, None, :], rope_t[..., 1].cpu()[:, None, :]
        r3 = ref.view(M, nh, 128)
        re, ro = r3[..., 0::2], r3[..., 1::2]
        want = torch.stack((cs * re - sn * ro, sn * re + cs * ro), -1).reshape(M, H) * SL2
        got = view(outq[:, :H]).cpu()
        assert (got - want).abs().max() < (5e-4 if f16 else 4e-3) * want.abs().max() + 5e-5, (f16, (got - want).abs().max())   # half an ulp of the storage type + the projection's own 2e-4
        assert torch.equal(qraw2, qraw)                                     # the cross-space vector is the same as before
    with pytest.raises(ValueError):
        ops.qpre_finish(qraw, d, nq, nh, rope=rope_t[:5].contiguous(), q_out=outq[:, :H])


@pytest.mark.parametrize("M", [300, 20, 513])
@pytest.mark.parametrize("f16", [True, False])
def test_low_plane_q_projection_with_fused_finish(M, f16):
    """Round 5: the low-plane q projection with the correction, the RMS norm, the rotation and the store of the
    attention's q fused into its epilogue (ca_gemm_problem.qpre_f32 = 3, N = n_split / 3) against the two-step route it
    replaces (second GEMM output + ca_qpre_finish_rope_f32): the same values up to the order in which a head's 128
    squares are summed (the cross-space vector within 2e-6 relative, the stored q within one ulp of its type on < 2 %
    of the elements), and against the fp32 reference of the unrounded projection.  M = 300: full tiles + a thin last row
    tile; M = 20: the thin-row kernel alone (the concept rows); M = 513: a 1-row last tile."""
    from oracle import flux_oracle as O
    nh, H = 2, 256
    g = torch.Generator(device="cpu").manual_seed(M)
    x = (torch.randn(M, H, generator=g) * 1.5).to(DEV)
    sh, sc = (torch.randn(H, generator=g) * 0.2).to(DEV), (torch.randn(H, generator=g) * 0.3).to(DEV)
    w, b = rnd(3 * H, H, scale=0.08), rnd(3 * H)
    nq, nk = (0.5 + torch.rand(128, generator=g)).bfloat16().to(DEV), (0.5 + torch.rand(128, generator=g)).bfloat16().to(DEV)
    rope_t = torch.randn(M, 64, 2, generator=g).to(DEV)
    rope_t = (rope_t / rope_t.norm(dim=-1, keepdim=True)).contiguous()
    hi = torch.zeros(M, H, device=DEV, dtype=torch.bfloat16)
    lo = torch.zeros_like(hi)
    ops.ln_modulate(x, hi, [(M, sh, sc)], out_lo=lo)
    y32 = (1 + sc) * torch.nn.functional.layer_norm(x, (H,), eps=1e-6) + sh

    def main_launch():
        qraw = torch.zeros(M, H, device=DEV)
        out = torch.zeros(M, 3 * H, device=DEV, dtype=torch.bfloat16)
        ops.gemm([ops.Gemm(hi, w, b, out, L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk, rope=rope_t,
                           q_prerope=qraw, qpre_raw=True, q_out_scale=SL2, qk_f16=f16)], L.TILE_PP_256x256)
        return qraw, out
    # two-step route
    qa, outa = main_launch()
    d = torch.zeros(M, H, device=DEV)
    ops.gemm([ops.Gemm(lo, w[:H], None, d)], L.TILE_PP_256x256)
    ops.qpre_finish(qa, d, nq, nh, rope=rope_t, q_out=outa[:, :H], q_out_scale=SL2, q_f16=f16)
    # fused route
    qb, outb = main_launch()
    kv = outb[:, H:].clone()
    ops.gemm([ops.Gemm(lo, w[:H], None, outb[:, :H], L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk,
                       rope=rope_t, q_prerope=qb, qpre_add=True, q_out_scale=SL2, qk_f16=f16)], L.TILE_PP_256x256)
    torch.cuda.synchronize()
    assert torch.equal(outb[:, H:], kv)                                   # k and v are not touched
    assert (qa - qb).abs().max() <= 2e-6 * qa.abs().max()
    view = (lambda t_: t_.contiguous().view(torch.float16).float()) if f16 else (lambda t_: t_.float())
    a_, b_ = view(outa[:, :H]), view(outb[:, :H])
    assert (a_ - b_).abs().max() <= (2.0 ** -10 if f16 else 2.0 ** -7) * a_.abs().max()
    assert (a_ != b_).float().mean() < 0.02
    ref = O.rms_norm((y32.cpu() @ w[:H].float().cpu().t() + b[:H].float().cpu()).view(M, nh, 128),
                     nq.float().cpu()).reshape(M, H)
    assert (qb.cpu() - ref).abs().max() < 2e-4
    # a row's bits do not depend on the tile form it runs in: the same rows as the head of a longer problem
    if M == 20:
        Mb = 256 + 20
        hi2 = torch.cat((torch.zeros(256, H, device=DEV, dtype=torch.bfloat16), hi))
        lo2 = torch.cat((torch.zeros(256, H, device=DEV, dtype=torch.bfloat16), lo))
        rope2 = torch.cat((torch.zeros(256, 64, 2, device=DEV), rope_t)).contiguous()
        q2 = torch.zeros(Mb, H, device=DEV)
        out2 = torch.zeros(Mb, 3 * H, device=DEV, dtype=torch.bfloat16)
        ops.gemm([ops.Gemm(hi2, w, b, out2, L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk, rope=rope2,
                           q_prerope=q2, qpre_raw=True, q_out_scale=SL2, qk_f16=f16)], L.TILE_PP_256x256)
        ops.gemm([ops.Gemm(lo2, w[:H], None, out2[:, :H], L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk,
                           rope=rope2, q_prerope=q2, qpre_add=True, q_out_scale=SL2, qk_f16=f16)], L.TILE_PP_256x256)
        assert torch.equal(q2[256:], qb) and torch.equal(out2[256:, :H], outb[:, :H])
    with pytest.raises(ValueError):   # qpre_add needs the fp32 buffer
        ops.gemm([ops.Gemm(lo, w[:H], None, outb[:, :H], L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk,
                           rope=rope_t, q_prerope=torch.zeros(M, H, device=DEV, dtype=torch.bfloat16), qpre_add=True)],
                 L.TILE_PP_256x256)


def test_heatmap_logits_fp32_image_vectors():
    """fp32 image AND concept vectors (the fp32 q_prerope store): the same k order as the bf16-image form."""
    Lp, dim, C = 300, 3072, 6
    iv = torch.randn(Lp, dim, device=DEV)
    cv = torch.randn(C, dim, device=DEV)
    lg = torch.zeros(C, Lp, device=DEV)
    ops.heatmap_logits(iv, cv, lg)
    ref = (cv.double() @ iv.double().t()).float()
    assert (lg - ref).abs().max() < 2e-3
    lg_b = torch.zeros(C, Lp, device=DEV)
    ops.heatmap_logits(iv.bfloat16(), cv, lg_b)                 # bf16 image vectors: exactly the rounded inputs
    lg_b32 = torch.zeros(C, Lp, device=DEV)
    ops.heatmap_logits(iv.bfloat16().float(), cv, lg_b32)
    assert torch.equal(lg_b, lg_b32)
    with pytest.raises(ValueError):
        ops.heatmap_logits(iv, cv.bfloat16(), lg)               # fp32 image vectors go with fp32 concept vectors


@PRESCALED
def test_attention_full_size(prescaled):
    """24 heads, 4352 rows (T=256 + L=4096) read in place from a [rows, 9216] projection buffer."""
    nh, n = 24, 4352
    buf = rnd(n, 9216)
    if prescaled:
        buf[:, :3072] = (buf[:, :3072].float() * SL2).bfloat16()
    q, k, v = buf[:, :3072], buf[:, 3072:6144], buf[:, 6144:]
    out = torch.zeros(n, 3072, device=DEV, dtype=torch.bfloat16)
    ops.attention([ops.Attn(q, out, k, v)], nh, **({"q_prescaled": True} if prescaled else {}))
    if prescaled:
        q = q.float() / SL2
    ref = torch.cat([attn_ref(q[i:i + 1088], k, v, nh) for i in range(0, n, 1088)])
    close(out, ref, atol=5e-3)


def _as_half_bits(x):
    """fp32 -> IEEE half, returned in a bf16-typed tensor (what the qkv epilogue writes with qk_f16 = 1)."""
    return x.float().half().view(torch.bfloat16)


@pytest.mark.parametrize("nq,nk,n0", [(300, 200, 0), (513, 1000, 0), (70, 64, 0), (257, 448, 100), (4352, 4352, 256)])
def test_attention_half_precision_q_and_k(nq, nk, n0):
    """ca_attn_fwd_qk16: q (pre-scaled) and k rows hold IEEE half, v stays bf16 -- the kernel of the layers whose heat
    maps are requested.  Against the fp32 reference on the half-rounded q / k (exact statement of what it computes:
    bf16-output tolerance), and -- the point of it -- on UNROUNDED q / k the fp32 output copy is several times closer
    than the bf16-q/k kernel's (11 vs 8 mantissa bits in the logits)."""
    nh = 2
    H = nh * 128
    q32 = torch.randn(nq, H, device=DEV) * 1.3
    k32 = torch.randn(nk, H, device=DEV) * 1.3
    v = rnd(nk, H, seed=4)
    qh, kh = _as_half_bits(q32 * SL2), _as_half_bits(k32)
    out = torch.zeros(nq, H, device=DEV, dtype=torch.bfloat16)
    o32 = torch.zeros(nq, H, device=DEV)
    if n0:
        pr = ops.Attn(qh, out, kh[:n0], v[:n0], kh[n0:], v[n0:], out_f32=o32)
    else:
        pr = ops.Attn(qh, out, kh, v, out_f32=o32)
    ops.attention([pr], nh, q_prescaled=True, qk_f16=True)
    q_eff, k_eff = (q32 * SL2).half().float() / SL2, k32.half().float()
    ref = torch.cat([attn_ref(q_eff[i:i + 1088], k_eff, v, nh) for i in range(0, nq, 1088)])
    close(out, ref, atol=1e-2)
    assert torch.equal(o32.bfloat16(), out)
    # the same problem through bf16 q / k: further from the fp32 attention of the unrounded q / k
    qb, kb = (q32 * SL2).bfloat16(), k32.bfloat16()
    ob, ob32 = torch.zeros_like(out), torch.zeros_like(o32)
    prb = ops.Attn(qb, ob, kb[:n0], v[:n0], kb[n0:], v[n0:], out_f32=ob32) if n0 else ops.Attn(qb, ob, kb, v, out_f32=ob32)
    ops.attention([prb], nh, q_prescaled=True)
    ref32 = torch.cat([attn_ref(q32[i:i + 1088], k32, v, nh) for i in range(0, nq, 1088)])
    # (rms: what is left with half-precision q / k is the bf16 rounding of P, which both kernels share)
    e16, eb = (o32 - ref32).pow(2).mean().sqrt().item(), (ob32 - ref32).pow(2).mean().sqrt().item()
    assert e16 < 0.6 * eb, (e16, eb)
    assert (o32 - ref32).abs().max().item() < (ob32 - ref32).abs().max().item()
    with pytest.raises(ValueError):
        ops.attention([pr], nh, qk_f16=True)          # half-precision q / k exist for pre-scaled q only


def test_qkv_epilogue_writes_half_precision_q_and_k():
    """ca_gemm_problem.qk_f16: the fused QK-norm + RoPE epilogue stores the rotated q and k as IEEE half, v (and a
    captured pre-RoPE q) unchanged; full row tiles and a thin last row tile (M = 300) give the same bits per row."""
    nh, M, H = 2, 300, 256
    a, w, b = rnd(M, H), rnd(3 * H, H, scale=0.08), rnd(3 * H)
    nq, nk = (0.5 + torch.rand(128)).bfloat16().to(DEV), (0.5 + torch.rand(128)).bfloat16().to(DEV)
    table = torch.randn(M, 64, 2, device=DEV)
    table = table / table.norm(dim=-1, keepdim=True)                 # unit (cos, sin) pairs
    outs = {}
    for f16 in (False, True):
        out = torch.zeros(M, 3 * H, device=DEV, dtype=torch.bfloat16)
        qp = torch.zeros(M, H, device=DEV)
        ops.gemm([ops.Gemm(a, w, b, out, L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk, rope=table,
                           q_prerope=qp, q_out_scale=SL2, qk_f16=f16)])
        outs[f16] = (out, qp)
    o_b, o_h = outs[False][0], outs[True][0]
    assert torch.equal(o_b[:, 2 * H:], o_h[:, 2 * H:])               # v: bf16 either way
    assert torch.equal(outs[False][1], outs[True][1])                # the captured pre-RoPE q does not change
    qk_h = o_h[:, :2 * H].contiguous().view(torch.float16).float()   # the same bytes read as half
    qk_b = o_b[:, :2 * H].float()
    # both are roundings of the same fp32 value: they agree to bf16's quantum, and the half one has 3 more bits
    assert (qk_h - qk_b).abs().max() <= 2.0 ** -8 * qk_h.abs().max()
    from oracle import flux_oracle as O
    y = a.float().cpu() @ w.float().cpu().t() + b.float().cpu()
    qr = O.rms_norm(y[:, :H].view(M, nh, 128), nq.float().cpu())
    kr = O.rms_norm(y[:, H:2 * H].view(M, nh, 128), nk.float().cpu())
    cs, sn = table[..., 0].cpu()[:, None, :], table[..., 1].cpu()[:, None, :]
    def rope(x):
        xe, xo = x[..., 0::2], x[..., 1::2]
        return torch.stack((cs * xe - sn * xo, sn * xe + cs * xo), -1).reshape(M, H)
    ref = torch.cat((rope(qr) * SL2, rope(kr)), 1)
    eh, eb = (qk_h.cpu() - ref).abs().max().item(), (qk_b.cpu() - ref).abs().max().item()
    assert eh < 0.25 * eb, (eh, eb)
    # row-tile independence: the first 256 rows alone (a full tile) and the last 44 (thin rows) give the same bits
    out2 = torch.zeros(256, 3 * H, device=DEV, dtype=torch.bfloat16)
    ops.gemm([ops.Gemm(a[:256], w, b, out2, L.EPI_QKV_NORM_ROPE, n_split=3 * H, norm_q=nq, norm_k=nk,
                       rope=table[:256].contiguous(), q_out_scale=SL2, qk_f16=True)])
    assert torch.equal(out2, o_h[:256])


def _peaky_case(kind, nq, nk, seed=0):
    """q (unscaled), k, v with a chosen logit distribution (nats): 'std<s>' = i.i.d. logits of that spread;
    'structured' = every row sees its first 64 keys near -20 nats and a few late keys near +25 (the text tile is cold,
    a handful of image keys carry the row: what a trained head can look like)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    if kind.startswith("std"):
        s = float(kind[3:])
        a = math.sqrt(s)                     # logit = q.k / sqrt(128), q, k ~ a N(0,1): std = a^2
        q = torch.randn(nq, 128, generator=g) * a
        k = torch.randn(nk, 128, generator=g) * a
    else:
        u = torch.randn(128, generator=g)
        u = u / u.norm()
        q = u[None, :] * (math.sqrt(128.0) ** 0.5 * 3.0) + torch.randn(nq, 128, generator=g) * 0.05
        k = torch.randn(nk, 128, generator=g) * 0.05
        qs = float((q[0] @ u))
        lo, hi = (-40.0, 45.0) if kind == "structured_far" else (-20.0, 25.0)
        k[:64] += u[None, :] * (lo * math.sqrt(128.0) / qs)
        hot = torch.randperm(nk - 64, generator=g)[:5] + 64
        k[hot] += u[None, :] * (hi * math.sqrt(128.0) / qs)
    v = torch.randn(nk, 128, generator=g)
    return q.to(DEV).bfloat16(), k.to(DEV).bfloat16(), v.to(DEV).bfloat16()


@pytest.mark.parametrize("kind", ["std1", "std4", "std8", "std16", "structured", "structured_far"])
def test_attention_peaky_distributions(kind):
    """ca_attn4_kernel keeps tile 0's maximum as the softmax reference while the row sums stay below 2^64 (fp32 is
    scale-free: nothing is lost before that) and moves a wave's references up IN PLACE beyond it (exact powers of two,
    no key visited twice).  Accuracy against the fp32 reference on each distribution, and the counters say which path
    ran: no workgroup needs the classical recomputation on any of these -- round 3's kernel (limit 2^60, no
    re-reference) recomputed every workgroup of both structured cases -- and the model-like distributions never leave
    the fast path at all."""
    nh, nq, nk = 1, 300, 4352 - 13            # ragged last tile; the second workgroup has inactive waves (300 - 256 <= 64)
    q0, k, v = _peaky_case(kind, nq, nk)
    q, qe, kw = attn_forms(q0, True)
    out = torch.zeros(nq, 128, device=DEV, dtype=torch.bfloat16)
    ops.attention_stats(reset=True)
    ops.attention([ops.Attn(q, out, k, v)], nh, **kw)
    torch.cuda.synchronize()
    st = ops.attention_stats()
    ref = attn_ref(qe, k, v, nh)
    close(out, ref, atol=1e-2)
    assert st["recomputed_workgroups"] == 0, st
    if kind in ("std1", "std4", "std8"):
        assert st["rereference_events"] == 0, st
    if kind == "structured_far":              # 85 nats = 123 octaves between tile 0's maximum and the hot keys
        assert st["rereference_events"] > 0, st


@pytest.mark.parametrize("octaves", [58.0, 62.0, 95.0, 110.0, 130.0, -90.0])
def test_attention_late_spike_around_the_old_and_new_limits(octaves):
    """A single late key `octaves` above its row's tile-0 reference, in a launch with a ragged tail and inactive waves
    (nq % 256 <= 192): below / above round 3's 2^60 limit (58, 62: the kept reference simply carries them now), near
    the 2^100 limit (95, 110: the in-place re-reference at the next check, or the final check), and beyond fp32 (130:
    exp2 overflows to inf, the final check sends the workgroup through the classical recomputation).  -90 = a DRIFT:
    90 octaves at tile 2 and another 90 above THAT at tile 8 -- 180 octaves above tile 0's maximum, which no single
    fp32 reference spans -- handled in place, no recomputation.  All against the fp32 reference."""
    nh, nq, nk = 1, 256 + 130, 64 * 11 + 5
    q0, k, v = rnd(nq, 128), rnd(nk, 128, seed=3) * 0.05, rnd(nk, 128, seed=4)
    row = 256 + 70                             # second workgroup, its second wave; waves 2 and 3 of it have no rows
    qn = float(q0[row].float().pow(2).sum())
    per_oct = math.sqrt(128.0) / 1.4426950408889634 / qn
    if octaves > 0:
        spikes = [(64 * 6 + 9, octaves)]
    else:
        spikes = [(64 * 2 + 40, 90.0), (64 * 8 + 9, 180.0)]
    for pos, o in spikes:
        k[pos] = (q0[row].float() * (o * per_oct)).bfloat16()
    q, qe, kw = attn_forms(q0, True)
    out = torch.zeros(nq, 128, device=DEV, dtype=torch.bfloat16)
    ops.attention_stats(reset=True)
    ops.attention([ops.Attn(q, out, k, v)], nh, **kw)
    torch.cuda.synchronize()
    st = ops.attention_stats()
    close(out, attn_ref(qe, k, v, nh), atol=1e-2)
    assert (out[row].float() - v[spikes[-1][0]].float()).abs().max() < 5e-2     # one-hot on the last spike's value row
    if octaves in (58.0, 62.0):
        assert st == {"recomputed_workgroups": 0, "rereference_events": 0}, st
    elif octaves == 95.0:
        assert st["recomputed_workgroups"] == 0, st
    elif octaves == 130.0:
        assert st["recomputed_workgroups"] == 1, st
    elif octaves < 0:
        assert st["recomputed_workgroups"] == 0 and st["rereference_events"] >= 1, st


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_attention_nan_and_inf_keys_propagate_and_nothing_hangs(bad):
    """A NaN / inf element in K: the rows that see it come out NaN (as torch's SDPA gives), every other workgroup's
    rows are untouched, and the launch ends (every wave reaches the same barriers on the recomputation path, waves
    without query rows included)."""
    nh, nq, nk = 2, 256 + 70, 64 * 9 + 17
    H = nh * 128
    q0, k, v = rnd(nq, H), rnd(nk, H, seed=3), rnd(nk, H, seed=4)
    k[300, 5] = bad                           # head 0 only
    q, qe, kw = attn_forms(q0, True)
    out = torch.zeros(nq, H, device=DEV, dtype=torch.bfloat16)
    ops.attention([ops.Attn(q, out, k, v)], nh, **kw)
    torch.cuda.synchronize()
    ref = attn_ref(qe, k, v, nh)
    bad_rows = torch.isnan(ref[:, :128]).any(1)           # NaN: every row; inf: the rows whose q[5] makes the score +inf
    assert bad_rows.any()
    assert torch.equal(torch.isnan(out[:, :128].float()).any(1), bad_rows)
    assert torch.isnan(out[bad_rows][:, :128].float()).all()
    if (~bad_rows).any():
        close(out[~bad_rows][:, :128], ref[~bad_rows][:, :128], atol=1e-2)
    close(out[:, 128:], ref[:, 128:], atol=1e-2)          # head 1 never saw it


def test_attention_kv_base_with_bit_31_set():
    """Regression for commit 1e7c133: `readfirstlane` returns int, and a 64-bit tile base built as (hi << 32) | lo
    sign-extended a low half with bit 31 set -- a fault that came and went with the allocation addresses.  Place K / V
    (and q / out) at an address whose bit 31 is set, deterministically: somewhere inside a 2.5 GiB allocation."""
    nh, nq, nk = 2, 300, 64 * 7 + 3
    H = nh * 128
    big = torch.empty(5 * (1 << 29), dtype=torch.uint8, device=DEV)          # 2.5 GiB spans a bit-31 region
    base = big.data_ptr()
    start = base if base & 0x80000000 else ((base >> 31) + 1) << 31          # first address with bit 31 set
    start = (start + 255) & ~255
    need = (nq + 2 * nk) * H * 2 + nq * H * 2
    assert start + need + 4096 <= base + big.numel() and (start & 0x80000000)
    off = start - base
    def view(n_rows, o):
        return big[o:o + n_rows * H * 2].view(torch.bfloat16).view(n_rows, H)
    qv, kv, vv, ov = view(nq, off), view(nk, off + nq * H * 2), view(nk, off + (nq + nk) * H * 2), \
        view(nq, off + (nq + 2 * nk) * H * 2)
    assert all(t.data_ptr() & 0x80000000 for t in (qv, kv, vv, ov))
    q0, k, v = rnd(nq, H), rnd(nk, H, seed=3), rnd(nk, H, seed=4)
    q, qe, kw = attn_forms(q0, True)
    qv.copy_(q), kv.copy_(k), vv.copy_(v), ov.zero_()
    ops.attention([ops.Attn(qv, ov, kv, vv)], nh, **kw)
    torch.cuda.synchronize()
    close(ov, attn_ref(qe, k, v, nh), atol=1e-2)
    # the same through two key segments (the descriptor of segment 1 is "the address key 0 WOULD have": below the buffer)
    ov.zero_()
    ops.attention([ops.Attn(qv, ov, kv[:100], vv[:100], kv[100:], vv[100:])], nh, **kw)
    torch.cuda.synchronize()
    close(ov, attn_ref(qe, k, v, nh), atol=1e-2)


def test_attention_shape_fuzz():
    """tools/fuzz_attn4.py: 80 random launches of the pre-scaled kernel (1-3 problems, 1-3 heads, query / key row counts
    around the tile and workgroup boundaries, one or two key segments that are not adjacent in memory, one or two query
    segments, padded row strides, fp32 output copies) against the fp32 reference, elementwise tolerance
    6e-3 + 2^-7 |reference|."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_attn4.py"), "80", "3"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "RESULT clean" in r.stdout, (r.stdout[-800:], r.stderr[-800:])


# ------------------------------------------------------------------------------------------ row kernels
def test_ln_modulate_segments():
    M, H = 70, 3072
    x = rnd(M, H, scale=2.0)
    sh = [torch.randn(H, device=DEV) for _ in range(3)]
    sc = [torch.randn(H, device=DEV) * 0.3 for _ in range(3)]
    out = torch.empty_like(x)
    ops.ln_modulate(x, out, [(4, sh[0], sc[0]), (36, sh[1], sc[1]), (M, sh[2], sc[2])])
    ln = torch.nn.functional.layer_norm(x.float(), (H,), eps=1e-6)
    seg = torch.tensor([0] * 4 + [1] * 32 + [2] * (M - 36), device=DEV)
    ref = (1 + torch.stack(sc)[seg]) * ln + torch.stack(sh)[seg]
    close(out, ref, atol=1e-2)
    # small hidden size (tiny test geometry)
    x2 = rnd(9, 256)
    o2 = torch.empty_like(x2)
    ops.ln_modulate(x2, o2, [(9, sh[0][:256].contiguous(), sc[0][:256].contiguous())])
    close(o2, (1 + sc[0][:256]) * torch.nn.functional.layer_norm(x2.float(), (256,), eps=1e-6) + sh[0][:256], atol=1e-2)


def test_qknorm_rope_matches_oracle():
    from oracle import flux_oracle as O
    nh, M, C = 2, 50, 3
    qkv = rnd(M, 3 * nh * 128, scale=2.0)
    orig = qkv.clone()
    ids = torch.zeros(M, 3)
    ids[C:, 1] = torch.arange(M - C) // 8
    ids[C:, 2] = torch.arange(M - C) % 8
    cos, sin = O.rope_cos_sin(ids, (16, 56, 56), 10000)
    table = torch.stack((cos, sin), -1).contiguous().to(DEV)
    qs = [(0.5 + torch.rand(128)).bfloat16().to(DEV) for _ in range(4)]
    pre = torch.zeros(M, nh * 128, device=DEV, dtype=torch.bfloat16)
    ops.qknorm_rope(qkv, nh, [(10, qs[0], qs[1]), (M, qs[2], qs[3])], table, q_prerope=pre)
    x = orig.float().cpu().view(M, 3, nh, 128)
    for which in (0, 1):
        sc = torch.stack([qs[which].float().cpu()] * 10 + [qs[2 + which].float().cpu()] * (M - 10))  # [M,128]
        normed = O.rms_norm(x[:, which], sc[:, None, :])
        if which == 0:
            close(pre.cpu(), normed.reshape(M, -1), atol=1e-2)
        roped = O.apply_rope(normed.permute(1, 0, 2)[None], cos, sin)[0].permute(1, 0, 2)
        close(qkv.cpu().view(M, 3, nh, 128)[:, which], roped, atol=1e-2)
    assert torch.equal(qkv.view(M, 3, nh, 128)[:, 2], orig.view(M, 3, nh, 128)[:, 2])  # v untouched


@pytest.mark.parametrize("nv,K,N", [(1, 256, 3072), (2, 3072, 18432), (3, 768, 100), (4, 4096, 64), (8, 3072, 6144),
                                    (6, 256, 512)])
def test_gemv(nv, K, N):
    x = torch.randn(nv, K, device=DEV)
    w, b = rnd(N, K, scale=0.05), rnd(N)
    out = torch.zeros(nv, N, device=DEV)
    ops.gemv(x, w, b, out, silu_input=True)
    ref = torch.nn.functional.silu(x) @ w.float().t() + b.float()
    assert (out - ref).abs().max() < 2e-3
    ops.gemv(x, w, None, out, silu_input=False, accumulate=True)
    assert (out - (ref + x @ w.float().t())).abs().max() < 4e-3


@pytest.mark.parametrize("nv", [5, 40, 100, 300])
def test_modulation_gemm_matches_gemv(nv):
    """adaLN modulations of many conditioning vectors as two bf16 GEMMs (silu(vec) split into hi + lo planes, thin-row
    kernel of 64 rows per workgroup, second plane accumulating) against the fp32-input GEMV: the planes carry silu(vec)
    to ~16 mantissa bits, so the two agree to a few 1e-5 relative of the row's scale; and the result of a vector must
    not depend on which other vectors share the launch (bit for bit)."""
    K, N = 256, 1536
    vecs = torch.randn(nv, K, device=DEV)
    w, b = rnd(N, K, scale=0.1), rnd(N)
    ones = torch.ones(N, device=DEV)
    hi, lo = torch.empty(nv, K, device=DEV, dtype=torch.bfloat16), torch.empty(nv, K, device=DEV, dtype=torch.bfloat16)
    ops.silu_split(vecs, hi, lo)
    s = torch.nn.functional.silu(vecs)
    assert (hi.float() + lo.float() - s).abs().max() <= 2e-5 * s.abs().max()
    out = torch.empty(nv, N, device=DEV)
    assert ops.modulation_gemm(vecs, w, b, out, ones)
    ref = torch.empty(nv, N, device=DEV)
    for r0 in range(0, nv, 4):
        ops.gemv(vecs[r0:r0 + 4], w, b, ref[r0:r0 + 4], silu_input=True)
    assert (out - ref).abs().max() <= 3e-5 * ref.abs().max() + 1e-6
    close(out, s @ w.float().t() + b.float(), atol=1e-3, rtol=1e-4)
    alone = torch.empty(3, N, device=DEV)
    assert ops.modulation_gemm(vecs[1:4].contiguous(), w, b, alone, ones)
    assert torch.equal(alone, out[1:4])
    assert not ops.modulation_gemm(vecs, w[:1000], b[:1000], out[:, :1000], ones[:1000])  # N % 256: caller falls back


@pytest.mark.parametrize("C", [1, 4, 6])
def test_heatmap_logits_softmax_accumulate(C):
    Lp, dim = 4096, 3072
    img, con = rnd(Lp, dim), rnd(C, dim, scale=0.05)
    logits = torch.empty(C, Lp, device=DEV)
    ops.heatmap_logits(img, con, logits)
    ref = con.float() @ img.float().t()
    assert (logits - ref).abs().max() < 1e-3 * max(1.0, ref.abs().max().item())
    logits32 = torch.empty(C, Lp, device=DEV)
    ops.heatmap_logits(img, con.float() * 1.001, logits32)  # fp32 concept vectors
    assert (logits32 - 1.001 * ref).abs().max() < 1e-3 * max(1.0, ref.abs().max().item())
    acc = torch.full((C, Lp), 0.25, device=DEV)
    ops.heatmap_softmax_accumulate(logits, acc, 0.5)
    assert (acc - (0.25 + 0.5 * torch.softmax(ref, dim=0))).abs().max() < 1e-4


def test_axpy():
    x, y = rnd(4096, 64), rnd(4096, 64, seed=9)
    ref = x.float() - 0.25 * y.float()
    ops.axpy(x, y, -0.25)
    close(x, ref, atol=1e-3)
    x2, y2 = rnd(1001), rnd(1001, seed=2)
    x2 = x2[:1000].clone()  # 16-byte aligned, not a multiple of 8 elements
    ref2 = x2.float() + 2 * y2[:1000].float()
    ops.axpy(x2, y2[:1000].clone(), 2.0)
    close(x2, ref2, atol=1e-3)


def test_axpy_f32_and_split_planes():
    """The fp32 Euler state: x += a y in fp32 (exactly one fma per element), and the two bf16 planes of an fp32 operand
    (hi = bf16(x), lo = bf16(x - hi), together ~16 mantissa bits) with which img_in sees it (two GEMM passes)."""
    x = torch.randn(4096 * 64 + 5, device=DEV)[:4096 * 64].clone()
    y = rnd(4096, 64, seed=9)
    ref = torch.addcmul(x.double(), y.double().flatten(), torch.tensor(-0.25, dtype=torch.float64, device=DEV))
    ops.axpy_f32(x, y, -0.25)
    assert (x.double() - ref).abs().max() <= 2.0 ** -23 * ref.abs().max()
    x3 = torch.randn(1003, device=DEV)
    y3 = rnd(1003, seed=2)
    r3 = x3 + 2.0 * y3.float()
    ops.axpy_f32(x3, y3, 2.0)
    assert (x3 - r3).abs().max() <= 1e-6
    x4, y4 = torch.randn(70001, device=DEV), torch.randn(70001, device=DEV)       # fp32 y: the unrounded prediction
    r4 = torch.addcmul(x4.double(), y4.double(), torch.tensor(0.125, dtype=torch.float64, device=DEV))
    ops.axpy_f32(x4, y4, 0.125)
    assert (x4.double() - r4).abs().max() <= 2.0 ** -23 * r4.abs().max()
    v = torch.randn(300, 64, device=DEV) * 3
    hi, lo = torch.empty(300, 64, device=DEV, dtype=torch.bfloat16), torch.empty(300, 64, device=DEV, dtype=torch.bfloat16)
    ops.split_planes(v, hi, lo)
    assert torch.equal(hi, v.bfloat16()) and torch.equal(lo, (v - v.bfloat16().float()).bfloat16())
    assert (hi.float() + lo.float() - v).abs().max() <= 2.0 ** -16 * v.abs().max()
    # the two planes through the GEMM: W (hi + lo) to fp32 accuracy of the operand
    w, b = rnd(256, 64, scale=0.3), rnd(256)
    out = torch.zeros(300, 256, device=DEV)
    ops.gemm([ops.Gemm(hi, w, b, out)])
    ops.gemm([ops.Gemm(lo, w, None, out, L.EPI_GATE_RESIDUAL, resid=out, gate=torch.ones(256, device=DEV))])
    ref_g = v.double() @ w.double().t() + b.double()
    one_plane = hi.double() @ w.double().t() + b.double()
    assert (out.double() - ref_g).abs().max() < 0.02 * (one_plane - ref_g).abs().max() + 1e-5
    with pytest.raises(ValueError):
        ops.split_planes(v, hi, lo[:, :32])
