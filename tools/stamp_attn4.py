"""Cycle shares of ca_attn4_kernel's tile loop (diagnostic build with -DCA_A4_STAMP, s_memtime around the three parts
of an iteration: tile-base bookkeeping | the 64-MFMA instruction stream | drain + barrier).

    python tools/stamp_attn4.py build     # here (no GPU): tools/ab/libca_a4_stamp.so
    python tools/stamp_attn4.py           # on the GPU box

Read the SHARES, not the length: the stamps' own waits forbid overlaps the real kernel has (cdna_hip_programming.md 7)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "conceptattention_amd", "csrc")
LIB = os.environ.get("CA_A4_STAMP_LIB") or os.path.join(ROOT, "tools", "ab", "libca_a4_stamp.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DCA_A4_STAMP", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", f"-I{SRC}", "-o", LIB] +
                          [os.path.join(SRC, f) for f in ("ca_api.hip", "ca_gemm.hip", "ca_attn.hip", "ca_attn4.hip",
                                                          "ca_rowops.hip")])
    print("built", LIB)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np
import torch
from conceptattention_amd import _lib
_lib.LIB_PATH = LIB
from conceptattention_amd import ops

nh, n = 24, 4352
buf = torch.randn(n, 3 * nh * 128, device="cuda").bfloat16()
H = nh * 128
out = torch.empty(n, H, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.attention([ops.Attn(buf[:, :H], out, buf[:, H:2 * H], buf[:, 2 * H:])], nh, q_prescaled=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
ops.attention([ops.Attn(buf[:, :H], out, buf[:, H:2 * H], buf[:, 2 * H:])], nh, q_prescaled=True)
e.record()
torch.cuda.synchronize()
lib = _lib.load()
arr = (ctypes.c_ulonglong * (4 * 4 * 4096))()
lib.ca_debug_read_attn4.argtypes = [ctypes.c_void_p]
assert lib.ca_debug_read_attn4(arr) == 0
a = np.frombuffer(arr, dtype=np.uint64).reshape(4096, 4, 4).astype(np.float64)
nblk = 8 * 3 * 17
a = a[:nblk]
it = a[..., 3]
per = a[..., :3] / np.maximum(it[..., None], 1)
print(f"launch {s.elapsed_time(e) * 1e3:.0f} us (stamped build); iterations per workgroup {it.mean():.0f}")
for w in range(4):
    m = per[:, w].mean(0)
    print(f"wave {w}: bookkeeping {m[0]:7.0f}  stream {m[1]:7.0f}  drain+barrier {m[2]:7.0f}  = {m.sum():7.0f} cycles per tile")
first = per[:256].mean((0, 1))
print("first round of workgroups:", [round(x) for x in first], "last round:", [round(x) for x in per[-152:].mean((0, 1))])
