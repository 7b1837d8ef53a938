"""Same-box A/B of whole-path throughput between ENVIRONMENT settings of one build: runs bench.py in fresh processes,
alternating (devices of this pool differ by several per cent, so only same-box pairs compare).
(library switches need the diagnostic build: python -m conceptattention_amd.csrc.build --ab, picked up automatically)
usage: python tools/env_ab.py [--reps 3] [--bench "--batch 1 --steps 4"] "CA_X=0" "CA_X=1" ...  ("" = default env)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library's A/B switches exist in the diagnostic build only (python -m conceptattention_amd.csrc.build --ab)
AB_LIB = os.path.join(ROOT, "tools", "ab", "switches", "libca.so")
args = sys.argv[1:]
reps, bench_args = 3, ["--steps", "10", "--warmup", "1"]
while args and args[0] in ("--reps", "--bench"):
    if args[0] == "--reps":
        reps, args = int(args[1]), args[2:]
    else:
        bench_args, args = args[1].split(), args[2:]
for rep in range(reps):
    for setting in args:
        env = dict(os.environ)
        if os.path.exists(AB_LIB):
            env.setdefault("CA_LIB_PATH", AB_LIB)
        for kv in setting.split():
            k, v = kv.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + bench_args,
                             capture_output=True, text=True, cwd=ROOT, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(setting, "FAILED", out.stderr[-400:], flush=True)
            continue
        d = json.loads(line[-1])
        ra = d.get("roofline_attention") or {}
        print(f"{setting or '(default)':32s} {d['value']:9.3f} heat maps/s  {d['ms_per_step']:7.2f} ms/step  GEMM "
              f"{d['roofline'].get('achieved', 0):6.1f} TF/s ({d['roofline'].get('avg_launch_us', 0):6.1f} us/launch)  attention "
              f"{ra.get('achieved', 0):6.1f} TF/s ({ra.get('avg_launch_us', 0):6.1f} us)  equal_single {d.get('batched_equals_single')}",
              flush=True)
