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
                              [os.path.join(src, f) for f in ("ca_api.hip", "ca_gemm.hip", "ca_attn.hip", "ca_rowops.hip")])
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
bench_gemm(4096, 4096, 3072, PP, name="warm")
for (M, N, K, epi, name) in [(21780, 9216, 3072, L.EPI_BIAS, "qkv x5"), (21780, 3072, 3072, L.EPI_BIAS, "proj x5"),
                             (21780, 12288, 3072, L.EPI_GELU_TANH, "mlp0 x5"), (21760, 3072, 15360, L.EPI_BIAS, "lin2 x5"),
                             (4352, 3072, 3072, L.EPI_BIAS, "proj x1"), (8192, 8192, 8192, L.EPI_BIAS, "8k")]:
    bench_gemm(M, N, K, PP, epi, name=name)
