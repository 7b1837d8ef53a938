"""Diagnostic: per-tile timeline of the PERSISTENT ping-pong GEMM walk (build with -DCA_GEMM_STAMP=2: stores left in
flight after a tile, as shipped).  For every round of the walk: median ticks from tile entry to the end of the
prologue wait (which, vmcnt being in order, also waits for the previous tile's stores), K loop, epilogue issue, and
the spread of the tile start times inside the round.   usage: python tools/stamp_gemm_persist.py [extra -D flags]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "conceptattention_amd", "csrc")
out = "/tmp/libca_gstamp2.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-DCA_GEMM_STAMP=2", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-o", out] + sys.argv[1:] +
                      [os.path.join(src, f) for f in ("ca_api.hip", "ca_gemm.hip", "ca_attn.hip", "ca_attn4.hip", "ca_rowops.hip")])
from conceptattention_amd import _lib
_lib.LIB_PATH = out
import numpy as np
import torch
from conceptattention_amd import _lib as L, ops

lib = _lib.load()
lib.ca_debug_read_gemm.argtypes = [ctypes.c_void_p]
arr = (ctypes.c_ulonglong * (8 * 2048))()


def run(M, N, K, epi, name):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(resid=o, gate=torch.randn(N, device="cuda")) if epi == L.EPI_GATE_RESIDUAL else {}
    if epi == L.EPI_QKV_NORM_ROPE:
        s128 = torch.ones(128, device="cuda", dtype=torch.bfloat16)
        kw = dict(n_split=N, norm_q=s128, norm_k=s128, rope=torch.randn(M, 64, 2, device="cuda"))
    for _ in range(3):
        ops.gemm([ops.Gemm(a, w, b, o, epi, **kw)], L.TILE_PP_256x256)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.gemm([ops.Gemm(a, w, b, o, epi, **kw)], L.TILE_PP_256x256)
    e.record()
    torch.cuda.synchronize()
    assert lib.ca_debug_read_gemm(arr) == 0
    nt = ((M + 255) // 256) * (N // 256)
    assert nt <= 2048
    t = np.array(arr[:8 * nt], dtype=np.float64).reshape(-1, 8) * 0.01  # hundreds of shader cycles
    t0 = t[:, 0].min()
    print(f"{name} M={M} N={N} K={K} tiles={nt} ({nt/256:.2f} rounds): {s.elapsed_time(e)*100:.1f} us per launch, "
          f"last end {t[:,3].max()-t0:.1f} us", flush=True)
    if nt > 512:  # same workgroup, consecutive tiles of its walk: what lies between one tile's last stamp and the next tile's first
        gap = t[256:, 0] - t[:-256, 3]
        span = t[(nt - 1) // 256 * 256, 3] - t[0, 0]
        print(f"  between tiles (last store issued by wave 0 -> next tile entered, i.e. the other waves' epilogues + barrier): median {np.median(gap):.2f}, "
              f"p95 {np.percentile(gap, 95):.2f}; workgroup 0 first entry -> last exit {span:.1f} = {span*100/(s.elapsed_time(e)*100):.0f} ticks per us")
    for r in range((nt + 255) // 256):
        x = t[r * 256:(r + 1) * 256]
        print(f"  round {r}: n={len(x):3d}  start p5/p50/p95 {np.percentile(x[:,0]-t0,5):7.1f}/{np.percentile(x[:,0]-t0,50):7.1f}/"
              f"{np.percentile(x[:,0]-t0,95):7.1f}  prologue(+drain of previous) {np.median(x[:,1]-x[:,0]):5.2f}  "
              f"loop {np.median(x[:,2]-x[:,1]):6.2f}  epilogue issue {np.median(x[:,3]-x[:,2]):5.2f}  tile {np.median(x[:,3]-x[:,0]):6.2f}"
              f"  | epilogue: barrier {np.median(x[:,4]-x[:,2]):5.2f} bias {np.median(x[:,5]-x[:,4]):5.2f} cvt+lds-write {np.median(x[:,6]-x[:,5]):5.2f}"
              f" lds-read+store {np.median(x[:,7]-x[:,6]):5.2f} second half {np.median(x[:,3]-x[:,7]):5.2f}")


run(13068, 9216, 3072, L.EPI_BIAS, "qkv x3")
run(13056, 6144, 3072, L.EPI_QKV_NORM_ROPE, "q,k thirds x3, fused norm + rope")
pass
run(8712, 12288, 3072, L.EPI_GELU_TANH, "mlp0 x2")
