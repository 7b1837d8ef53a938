"""Same-box A/B of whole-path throughput between settings of HipFluxDiT ATTRIBUTES (the model-level switches that are not
environment variables): runs bench.py in fresh processes, alternating, with the attributes overridden after construction.
usage: python tools/attr_ab.py [--reps 3] [--bench "--workload sweep --steps 20"] "epilogue_logits=False" "epilogue_logits=True" """
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
reps, bench_args = 3, ["--steps", "10", "--warmup", "1"]
while args and args[0] in ("--reps", "--bench"):
    if args[0] == "--reps":
        reps, args = int(args[1]), args[2:]
    else:
        bench_args, args = args[1].split(), args[2:]
code = ("import sys, runpy; sys.path.insert(0, {root!r}); import conceptattention_amd.flux_dit as F; "
        "_init = F.HipFluxDiT.__init__\n"
        "def init(self, *a, **k):\n"
        "    _init(self, *a, **k)\n"
        "    for kv in {setting!r}.split():\n"
        "        n, v = kv.split('=', 1)\n"
        "        setattr(self, n, eval(v))\n"
        "F.HipFluxDiT.__init__ = init\n"
        "sys.argv = ['bench.py', '--no-cpu-baseline'] + {extra!r}\n"
        "runpy.run_path({bench!r}, run_name='__main__')")
for rep in range(reps):
    for setting in args:
        out = subprocess.run([sys.executable, "-c", code.format(root=ROOT, setting=setting, extra=bench_args,
                                                                bench=os.path.join(ROOT, "bench.py"))],
                             capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(setting, "FAILED", out.stderr[-600:], flush=True)
            continue
        d = json.loads(line[-1])
        ra = d.get("roofline_attention") or {}
        print(f"{setting or '(default)':28s} {d['value']:9.3f} heat maps/s  {d['ms_per_step']:7.2f} ms/step  GEMM "
              f"{d['roofline'].get('avg_launch_us', 0):6.1f} us/launch  attention {ra.get('avg_launch_us', 0):6.1f} us  "
              f"equal_single {d.get('batched_equals_single')}", flush=True)
