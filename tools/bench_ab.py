"""Same-box A/B of whole-path throughput: runs bench.py once per given libconceptattn build, alternating, in fresh
processes (devices of this pool differ by several per cent, so only same-box pairs compare).
usage: python tools/bench_ab.py [--reps 2] libA.so libB.so ...   ("HEAD" = the in-tree library; extra bench.py
arguments through the environment: BENCH_ARGS="--precision fp8")"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
reps = 2
if args and args[0] == "--reps":
    reps, args = int(args[1]), args[2:]
code = ("import sys, runpy; sys.path.insert(0, {root!r}); from conceptattention_amd import _lib; "
        "_lib.LIB_PATH = {lib!r} if {lib!r} != 'HEAD' else _lib.LIB_PATH; "
        "sys.argv = ['bench.py', '--steps', '10', '--warmup', '1', '--no-cpu-baseline'] + {extra!r}; "
        "runpy.run_path({bench!r}, run_name='__main__')")
for rep in range(reps):
    for lib in args:
        path = lib if lib == "HEAD" else os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", code.format(root=ROOT, lib=path, bench=os.path.join(ROOT, "bench.py"), extra=os.environ.get("BENCH_ARGS", "").split())],
                             capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, "FAILED", out.stderr[-400:], flush=True)
            continue
        d = json.loads(line[-1])
        print(f"{lib:40s} {d['value']:7.3f} heat maps/s  {d['ms_per_step']:7.2f} ms/call  GEMM {d['roofline']['achieved']:6.1f} TF/s "
              f"({d['roofline']['avg_launch_us']:6.1f} us/launch)  attention {d['roofline_attention']['achieved']:6.1f} TF/s", flush=True)
