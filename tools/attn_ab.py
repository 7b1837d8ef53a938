"""Same-box A/B of two attention kernel sources: builds each into a scratch .so and times the Flux shape.
usage: python tools/attn_ab.py <variant.hip | "-DFLAG=..."> [...]   (paths relative to the repo root; a -D argument
builds the in-tree ca_attn.hip with that flag)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "conceptattention_amd", "csrc")
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].startswith("--run=")):
    for v in sys.argv[1:]:
        out = f"/tmp/libca_ab_{os.path.basename(v).replace('.', '_').replace('=', '_')}.so"
        attn = [v, os.path.join(src, "ca_attn.hip")] if v.startswith("-D") else [os.path.join(ROOT, v)]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               f"-I{src}", "-o", out] + attn +
                              [os.path.join(src, f) for f in ("ca_api.hip", "ca_gemm.hip", "ca_rowops.hip")])
    for rep in range(2):
        for v in sys.argv[1:]:
            out = f"/tmp/libca_ab_{os.path.basename(v).replace('.', '_').replace('=', '_')}.so"
            subprocess.check_call([sys.executable, __file__, f"--run={out}"])
    sys.exit(0)
sys.path.insert(0, ROOT)
from conceptattention_amd import _lib
_lib.LIB_PATH = sys.argv[1].split("=", 1)[1]
from tools.bench_kernels import bench_attn, timeit  # noqa: E402
print(os.path.basename(_lib.LIB_PATH), end=": ", flush=True)
bench_attn(4352)      # warm-up
bench_attn(4352, C=4)
bench_attn(4352, C=4)
import runpy  # noqa: E402
runpy.run_path(os.path.join(ROOT, "tools", "attn_batch_probe.py"), run_name="__main__")  # the 5-item launch
