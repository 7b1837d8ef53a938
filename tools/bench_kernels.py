"""Micro-benchmarks of the individual gfx950 kernels at the Flux shapes (development aid;
bench.py is the judged benchmark).  Prints TFLOP/s or GB/s per kernel on random data."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from conceptattention_amd import _lib as L
from conceptattention_amd import ops

dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).bfloat16()


def bench_gemm(M, N, K, tile=L.TILE_AUTO, epi=L.EPI_BIAS, name=""):
    a, w, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = {}
    if epi == L.EPI_GATE_RESIDUAL:
        kw = dict(resid=out, gate=torch.randn(N, device=dev))
    t = timeit(lambda: ops.gemm([ops.Gemm(a, w, b, out, epi, **kw)], tile))
    print(f"gemm {name:10s} M={M:5d} N={N:5d} K={K:5d} tile={tile} epi={epi}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
    return t


def bench_attn(n, nh=24, C=0):
    buf = rnd(n + C, 3 * nh * 128)
    H = nh * 128
    q, k, v = buf[:, :H], buf[:, H:2 * H], buf[:, 2 * H:]
    out = torch.empty(n + C, H, device=dev, dtype=torch.bfloat16)
    probs = [ops.Attn(q[C:], out[C:], k[C:], v[C:])]
    if C:
        probs.append(ops.Attn(q[:C], out[:C], k[:C], v[:C], k[C + 256:], v[C + 256:]))
    t = timeit(lambda: ops.attention(probs, nh))
    fl = 4 * n * n * 128 * nh
    print(f"attn n={n} heads={nh} C={C}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)


if __name__ == "__main__" and len(sys.argv) == 1:
    print(torch.cuda.get_device_name(0))
    PP, PP1, PP2 = L.TILE_PP_256x256, L.TILE_PP_256x128, L.TILE_PP_256x192
    for tile in (PP, PP2, PP1):
        bench_gemm(4096, 9216, 3072, tile, name="qkv")
    for tile in (PP, PP2, PP1):
        bench_gemm(4096, 3072, 3072, tile, L.EPI_GATE_RESIDUAL, name="proj")
    for tile in (PP, PP2):
        bench_gemm(4096, 12288, 3072, tile, L.EPI_GELU_TANH, name="mlp0")
    for tile in (PP, PP2, PP1):
        bench_gemm(4096, 3072, 12288, tile, L.EPI_GATE_RESIDUAL, name="mlp2")
    for tile in (PP, PP2):
        bench_gemm(4352, 21504, 3072, tile, name="linear1")
    for tile in (PP, PP2, PP1):
        bench_gemm(4352, 3072, 15360, tile, L.EPI_GATE_RESIDUAL, name="linear2")
    for tile in (PP,):
        bench_gemm(8192, 8192, 8192, tile, name="8k")
    bench_gemm(260, 9216, 3072, PP, name="txt-qkv")
    bench_attn(4352)
    bench_attn(4352, C=4)
    bench_attn(4608)
    # row kernels
    x = rnd(4356, 3072)
    o = torch.empty_like(x)
    sh, sc = torch.randn(3072, device=dev), torch.randn(3072, device=dev)
    t = timeit(lambda: ops.ln_modulate(x, o, [(4356, sh, sc)]))
    print(f"ln_modulate 4356x3072: {t*1e6:.1f} us  {2*x.numel()*2/t/1e9:.0f} GB/s")
    qkv = rnd(4356, 9216)
    table = torch.zeros(4356, 64, 2, device=dev)
    table[..., 0] = 1
    s128 = torch.ones(128, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.qknorm_rope(qkv, 24, [(4356, s128, s128)], table))
    print(f"qknorm_rope 4356x6144: {t*1e6:.1f} us  {2*4356*6144*2/t/1e9:.0f} GB/s")
    w = rnd(18432 * 8, 3072, scale=0.02)
    for nv in (2, 8):
        xv = torch.randn(nv, 3072, device=dev)
        ov = torch.empty(nv, 18432 * 8, device=dev)
        t = timeit(lambda: ops.gemv(xv, w, None, ov, silu_input=True))
        print(f"gemv {nv}x3072 -> {w.shape[0]}: {t*1e6:.1f} us  {w.numel()*2/t/1e9:.0f} GB/s")
    img, con = rnd(4096, 3072), rnd(4, 3072)
    lg = torch.empty(4, 4096, device=dev)
    acc = torch.zeros(4, 4096, device=dev)
    t = timeit(lambda: (ops.heatmap_logits(img, con, lg), ops.heatmap_softmax_accumulate(lg, acc, 1.0)))
    print(f"heatmap 4096x3072 C=4: {t*1e6:.1f} us  {img.numel()*2/t/1e9:.0f} GB/s")


def bench_qkv_fusion():
    """A/B: qkv GEMM + separate QK-norm/RoPE kernel vs the fused epilogue (same process)."""
    M, K, nh = 4352, 3072, 24
    H = nh * 128
    a, w, b = rnd(M, K), rnd(3 * H + 12288, K, scale=0.02), rnd(3 * H + 12288)
    qkv = torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16)
    cat = torch.empty(M, 12288, device=dev, dtype=torch.bfloat16)
    table = torch.zeros(M, 64, 2, device=dev); table[..., 0] = 1
    s128 = torch.ones(128, device=dev, dtype=torch.bfloat16)

    def unfused():
        ops.gemm([ops.Gemm(a, w, b, qkv, L.EPI_SPLIT_GELU, out2=cat, n_split=3 * H)])
        ops.qknorm_rope(qkv, nh, [(M, s128, s128)], table)

    def fused():
        ops.gemm([ops.Gemm(a, w, b, qkv, L.EPI_QKV_NORM_ROPE, out2=cat, n_split=3 * H, norm_q=s128, norm_k=s128, rope=table)])
    for _ in range(3):
        tu, tf = timeit(unfused), timeit(fused)
        print(f"linear1 (+norm/rope): unfused {tu*1e6:.1f} us, fused {tf*1e6:.1f} us", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "qkv":
    bench_qkv_fusion()


def bench_gemm_fp8(M, N, K, epi=L.EPI_BIAS, name=""):
    a, w, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
    qa, sa = ops.quantize_rows_fp8(a)
    qw, sw = ops.quantize_rows_fp8(w)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(resid=out, gate=torch.randn(N, device=dev)) if epi == L.EPI_GATE_RESIDUAL else {}
    t = timeit(lambda: ops.gemm([ops.Gemm(qa, qw, b, out, epi, a_scale=sa, w_scale=sw, **kw)]))
    tq = timeit(lambda: ops.quantize_rows_fp8(a, qa, sa))
    print(f"gemm-fp8 {name:8s} M={M:5d} N={N:5d} K={K:5d} epi={epi}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s"
          f"   (quantising A: {tq*1e6:6.1f} us, {M*K*3/tq/1e9:6.0f} GB/s)", flush=True)
    return t


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "fp8":
    print(torch.cuda.get_device_name(0))
    for (M, N, K, epi, name) in [(4096, 9216, 3072, L.EPI_BIAS, "qkv"), (4096, 3072, 3072, L.EPI_GATE_RESIDUAL, "proj"),
                                 (4096, 12288, 3072, L.EPI_GELU_TANH, "mlp0"), (4096, 3072, 12288, L.EPI_GATE_RESIDUAL, "mlp2"),
                                 (4352, 21504, 3072, L.EPI_BIAS, "linear1"), (4352, 3072, 15360, L.EPI_GATE_RESIDUAL, "linear2"),
                                 (8192, 8192, 8192, L.EPI_BIAS, "8k")]:
        bench_gemm(M, N, K, L.TILE_PP_256x256, epi, name=name)
        bench_gemm_fp8(M, N, K, epi, name=name)
