"""Which kernels PyTorch's F.linear / SDPA dispatch to at the path's shapes (tools only; read with rocprofv3):

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vendor_names -- python3 tools/vendor_kernel_names.py

The macro-tile / wave-group / prefetch fields of the hipBLASLt kernel names are the only view of the vendor design
available offline; tools/vendor_yardstick.py holds the timings."""
import torch
import torch.nn.functional as F

dev = "cuda"
H = 3072
shapes = [("qkv", 20480, 3 * H, H), ("proj", 20480, H, H), ("mlp0", 20480, 4 * H, H), ("mlp2", 20480, H, 4 * H),
          ("linear1", 21760, 7 * H, H), ("linear2", 21760, H, 5 * H), ("qkv1", 4096, 3 * H, H), ("lin1_1", 4352, 7 * H, H)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    for _ in range(3):
        F.linear(a, w, b)
    torch.cuda.synchronize()
q = torch.randn(1, 24, 4352, 128, device=dev).bfloat16()
for _ in range(3):
    F.scaled_dot_product_attention(q, q, q)
torch.cuda.synchronize()
