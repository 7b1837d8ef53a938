import sys, os
sys.path.insert(0, os.getcwd())
from tools.bench_kernels import bench_attn
for _ in range(3):
    bench_attn(4352)
    bench_attn(4352, C=4)
