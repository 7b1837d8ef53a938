"""Launch one kernel shape a few times (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conceptattention_amd import ops, _lib as L
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
dev = "cuda"
torch.manual_seed(0)
if what == "attn":
    nh, n = 24, 4352
    buf = torch.randn(n, 3 * nh * 128, device=dev).bfloat16()
    H = nh * 128
    out = torch.empty(n, H, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.attention([ops.Attn(buf[:, :H], out, buf[:, H:2 * H], buf[:, 2 * H:])], nh)
elif what == "gemm":
    M, N, K = 4096, 12288, 3072
    a, w = torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm([ops.Gemm(a, w, None, out)], L.TILE_PP_256x256)
torch.cuda.synchronize()
