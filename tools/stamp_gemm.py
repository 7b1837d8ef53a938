"""Diagnostic: build libconceptattn with -DCA_GEMM_STAMP into a scratch .so and print where a ping-pong GEMM
workgroup's time goes (s_memtime ticks, 100 MHz): prologue / K loop / epilogue, and the spread of the
workgroup start and end times over the launch."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "conceptattention_amd", "csrc")
out = "/tmp/libca_gstamp.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-DCA_GEMM_STAMP", "-o", out] +
                      [os.path.join(src, f) for f in ("ca_api.hip", "ca_gemm.hip", "ca_attn.hip", "ca_attn4.hip", "ca_rowops.hip")])
from conceptattention_amd import _lib
_lib.LIB_PATH = out
import numpy as np
import torch
from conceptattention_amd import _lib as L, ops

lib = _lib.load()
lib.ca_debug_read_gemm.argtypes = [ctypes.c_void_p]
arr = (ctypes.c_ulonglong * (4 * 2048))()


def run(M, N, K, epi, name):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(resid=o, gate=torch.randn(N, device="cuda")) if epi == L.EPI_GATE_RESIDUAL else {}
    for _ in range(3):
        ops.gemm([ops.Gemm(a, w, b, o, epi, **kw)], L.TILE_PP_256x256)
    torch.cuda.synchronize()
    assert lib.ca_debug_read_gemm(arr) == 0
    nt = ((M + 255) // 256) * (N // 256)
    t = np.array(arr[:4 * min(nt, 2048)], dtype=np.float64).reshape(-1, 4)
    t0 = t[:, 0].min()
    tick = 10.0  # ns per s_memtime tick (100 MHz constant clock)
    pro, loop, epi_t = (t[:, 1] - t[:, 0]) * tick, (t[:, 2] - t[:, 1]) * tick, (t[:, 3] - t[:, 2]) * tick
    print(f"{name:10s} M={M} N={N} K={K} tiles={nt}: prologue {np.median(pro)/1e3:6.2f} us  loop {np.median(loop)/1e3:7.2f} us"
          f"  epilogue {np.median(epi_t)/1e3:6.2f} us | WG start spread p50/p99 {np.percentile(t[:,0]-t0,50)*tick/1e3:6.2f}/"
          f"{np.percentile(t[:,0]-t0,99)*tick/1e3:6.2f} us  last end {(t[:,3].max()-t0)*tick/1e3:7.2f} us", flush=True)


for K in (256, 3072):
    run(4096, 4096, K, L.EPI_BIAS, "bias")
    run(4096, 4096, K, L.EPI_GELU_TANH, "gelu")
    run(4096, 4096, K, L.EPI_GATE_RESIDUAL, "gate")
run(4096, 12288, 3072, L.EPI_GELU_TANH, "mlp0")
run(4352, 3072, 15360, L.EPI_GATE_RESIDUAL, "linear2")
