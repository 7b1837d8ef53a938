"""Timing-only knock-outs of ca_attn4_kernel's instruction stream: where do the cycles of a tile go?  Each variant is the
stamped build (tools/stamp_attn4.py) of a stream generated with one ingredient removed (tools/gen_attn4_schedule.py,
CA_A4_KO); the results of such builds are wrong by construction, only their stamps are read.

    python tools/attn4_knockouts.py build     # here (no GPU): tools/ab/ko_<variant>/libca.so
    python tools/attn4_knockouts.py           # on the GPU box: one line per variant
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "conceptattention_amd", "csrc")
AB = os.path.join(ROOT, "tools", "ab")
VARIANTS = ["none", "exp", "add", "exp,add", "cvt", "lds", "dma", "exp,add,cvt", "exp,add,cvt,lds,dma"]
if os.environ.get("CA_A4_VARIANTS"):      # e.g. "none;order:valu_first;order:mem_first" (order:* = placement inside a gap)
    VARIANTS = os.environ["CA_A4_VARIANTS"].split(";")


def lib_of(v):
    return os.path.join(AB, "ko_" + v.replace(",", "_").replace(":", "_"), "libca.so")


if len(sys.argv) > 1 and sys.argv[1] == "build":
    procs = []
    for v in VARIANTS:
        d = os.path.dirname(lib_of(v))
        os.makedirs(d, exist_ok=True)
        shutil.copy(os.path.join(SRC, "ca_attn4.hip"), d)
        shutil.copy(os.path.join(SRC, "ca_attn4_kernel.inc"), d)   # (it includes the schedule: must sit next to the variant)
        env = dict(os.environ, CA_A4_OUT=os.path.join(d, "ca_attn4_sched.inc"))
        env["CA_A4_ORDER" if v.startswith("order:") else "CA_A4_KO"] = "" if v == "none" else v.split(":")[-1]
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_attn4_schedule.py")], env=env,
                              stdout=subprocess.DEVNULL)
        obj = os.path.join(d, "ca_attn4.o")
        cmd = ("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DCA_A4_STAMP -mllvm "
               f"-amdgpu-spill-vgpr-to-agpr=0 -I{SRC} -c {d}/ca_attn4.hip -o {obj} && /opt/rocm/bin/hipcc "
               f"--offload-arch=gfx950 -shared -fPIC -o {lib_of(v)} {obj} " +
               " ".join(os.path.join(SRC, f) for f in ("ca_api.o", "ca_gemm.o", "ca_attn.o", "ca_rowops.o")))
        procs.append(subprocess.Popen(cmd, shell=True))
        if len(procs) >= 4:
            assert procs.pop(0).wait() == 0
    for p in procs:
        assert p.wait() == 0
    print("built", len(VARIANTS), "variants")
    sys.exit(0)

for v in VARIANTS:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stamp_attn4.py")],
                       env=dict(os.environ, CA_A4_STAMP_LIB=lib_of(v)), capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("launch", "wave 0"))]
    print(f"{v:24s} " + " | ".join(lines) if lines else f"{v}: FAILED {r.stderr[-300:]}", flush=True)
