"""Same-box A/B of the GEMM source under different -D flags: builds each into a scratch .so and times the model's
launch shapes.   usage: python tools/gemm_ab.py "-DCA_GEMM_TWO_PHASE=0" "-DCA_GEMM_TWO_PHASE=1" """
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "conceptattention_amd", "csrc")
if not (len(sys.argv) == 2 and sys.argv[1].startswith("--run=")):
    outs = []
    for i, flags in enumerate(sys.argv[1:]):
        out = f"/tmp/libca_gemm_ab_{i}.so"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               f"-I{src}", "-o", out] + flags.split() +
                              [os.path.join(src, f) for f in ("ca_api.hip", "ca_gemm.hip", "ca_attn.hip", "ca_attn4.hip", "ca_rowops.hip")])
        outs.append((flags, out))
    for rep in range(2):
        for flags, out in outs:
            print(f"== {flags}", flush=True)
            subprocess.check_call([sys.executable, __file__, f"--run={out}"])
    sys.exit(0)
sys.path.insert(0, ROOT)
from conceptattention_amd import _lib
_lib.LIB_PATH = sys.argv[1].split("=", 1)[1]
from conceptattention_amd import _lib as L  # noqa: E402
from tools.bench_kernels import bench_gemm  # noqa: E402
PP = L.TILE_PP_256x256
import torch  # noqa: E402
from tools.bench_kernels import timeit, rnd  # noqa: E402
from conceptattention_amd import ops  # noqa: E402


def bench_model_launch(M, N, K, kind, name):
    """The model's own launches of a 5-item forward, with their epilogues (fp32 residual stream, fused QK-norm + RoPE)."""
    a, w, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
    if kind == "gate32":
        x = torch.randn(M, N, device="cuda")
        g = ops.Gemm(a, w, b, x, L.EPI_GATE_RESIDUAL, resid=x, gate=torch.randn(N, device="cuda"))
    else:  # qkv (+ GELU'd mlp columns past n_split) with the fused norm + rope epilogue
        H = 3072
        qkv = torch.empty(M, 3 * H, device="cuda", dtype=torch.bfloat16)
        cat = torch.empty(M, N - 3 * H, device="cuda", dtype=torch.bfloat16) if N > 3 * H else None
        table = torch.randn(M, 64, 2, device="cuda")
        s128 = torch.ones(128, device="cuda", dtype=torch.bfloat16)
        g = ops.Gemm(a, w, b, qkv, L.EPI_QKV_NORM_ROPE, out2=cat, n_split=3 * H, norm_q=s128, norm_k=s128, rope=table)
    t = timeit(lambda: ops.gemm([g], PP))
    print(f"gemm {name:12s} M={M:5d} N={N:5d} K={K:5d} {kind:7s}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)


bench_gemm(4096, 4096, 3072, PP, name="warm")
for (M, N, K, kind, name) in [(21780, 9216, 3072, "qkvrope", "dbl qkv x5"), (21780, 3072, 3072, "gate32", "dbl proj x5"),
                              (21780, 3072, 12288, "gate32", "dbl mlp2 x5"), (21760, 21504, 3072, "qkvrope", "sgl lin1 x5"),
                              (21760, 3072, 15360, "gate32", "sgl lin2 x5")]:
    bench_model_launch(M, N, K, kind, name)
for (M, N, K, epi, name) in [(21780, 9216, 3072, L.EPI_BIAS, "qkv x5"), (21780, 3072, 3072, L.EPI_BIAS, "proj x5"),
                             (21780, 12288, 3072, L.EPI_GELU_TANH, "mlp0 x5"), (21760, 3072, 15360, L.EPI_BIAS, "lin2 x5"),
                             (4352, 3072, 3072, L.EPI_BIAS, "proj x1"), (8192, 8192, 8192, L.EPI_BIAS, "8k")]:
    bench_gemm(M, N, K, PP, epi, name=name)
