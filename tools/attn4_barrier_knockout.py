"""Upper bound on what a one-barrier-per-two-tiles schedule of ca_attn4_kernel could gain, measured
instead of argued: a TIMING-ONLY build (-DCA_A4_KO_BARRIER2: the tile loop's s_barrier on every second tile only; its
results are wrong by construction) against the shipped kernel, the model's 5-item launch (10 problems, 2160 units) and the
one-item launch, alternating fresh processes on the same box.

    python tools/attn4_barrier_knockout.py build    # here (no GPU): tools/ab/ko_barrier2/libca.so
    python tools/attn4_barrier_knockout.py          # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KO = os.path.join(ROOT, "tools", "ab", "ko_barrier2", "libca.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    from conceptattention_amd.csrc.build import build
    print(build(defines=["CA_A4_KO_BARRIER2"], out=KO, verbose=False))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--run":
    sys.path.insert(0, ROOT)
    import torch
    from conceptattention_amd import _lib, ops
    from tools.bench_kernels import rnd, timeit
    NH, C, T, Li = 24, 4, 256, 4096
    H = NH * 128
    out = []
    for B in (5, 1):
        oT, oI, n = B * C, B * (C + T), B * (C + T + Li)
        qkv = rnd(n, 3 * H)
        att = torch.empty(n, H, device="cuda", dtype=torch.bfloat16)
        att32 = torch.empty(B * C, H, device="cuda")
        qs, ks, vs = qkv[:, :H] * 0.1275, qkv[:, H:2 * H], qkv[:, 2 * H:]
        qs = qs.bfloat16().contiguous()
        probs = []
        for j in range(B):
            cj, tj, ij = slice(j * C, (j + 1) * C), slice(oT + j * T, oT + (j + 1) * T), slice(oI + j * Li, oI + (j + 1) * Li)
            probs.append(ops.Attn(qs[cj], att[cj], ks[cj], vs[cj], ks[ij], vs[ij], out_f32=att32[cj]))
        for j in range(B):
            tj, ij = slice(oT + j * T, oT + (j + 1) * T), slice(oI + j * Li, oI + (j + 1) * Li)
            probs.append(ops.Attn(qs[tj], att[tj], ks[tj], vs[tj], ks[ij], vs[ij], q1=qs[ij], out1=att[ij]))
        t = min(timeit(lambda: ops.attention(probs, NH, q_prescaled=True)) for _ in range(3))
        out.append(f"{B} item(s): {t * 1e6:7.1f} us ({B * 4.0 * (T + Li) ** 2 * 128 * NH / t / 1e12:6.1f} TF/s)")
    print(f"{os.path.basename(os.path.dirname(_lib.LIB_PATH)):24s} " + "   ".join(out), flush=True)
    sys.exit(0)
for rep in range(3):
    for lib in ("", KO):
        env = dict(os.environ)
        if lib:
            env["CA_LIB_PATH"] = lib
        subprocess.check_call([sys.executable, __file__, "--run"], env=env)
