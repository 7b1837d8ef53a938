"""Diagnostic: build libconceptattn with -DCA_ATTN_STAMP into a scratch .so and print per-phase cycles."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "conceptattention_amd", "csrc")
out = "/tmp/libca_stamp.so"
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCA_ATTN_STAMP"] + extra + [
                       "-o", out] + [os.path.join(src, f) for f in ("ca_api.hip", "ca_gemm.hip", "ca_attn.hip", "ca_rowops.hip")])
from conceptattention_amd import _lib
_lib.LIB_PATH = out
import torch
from conceptattention_amd import ops
nh, n = 24, 4352
buf = torch.randn(n, 3 * nh * 128, device="cuda").bfloat16()
H = nh * 128
o = torch.empty(n, H, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.attention([ops.Attn(buf[:, :H], o, buf[:, H:2 * H], buf[:, 2 * H:])], nh)
torch.cuda.synchronize()
lib = _lib.load()
arr = (ctypes.c_ulonglong * 32)()
lib.ca_debug_read_attn.argtypes = [ctypes.c_void_p]
print("rc", lib.ca_debug_read_attn(arr))
names = ["Y compute", "Y retire", "bar after Y", "X stream", "bar after X"]
for w, base in ((0, 0), (4, 8)):
    nt = arr[base + 5]
    print(f"wave {w}: tiles {nt}: " + ", ".join(f"{names[i]} {arr[base+i]/max(nt,1):.0f}" for i in range(5)),
          " total/tile", sum(arr[base + i] for i in range(5)) / max(nt, 1))
