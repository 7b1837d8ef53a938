"""Build libconceptattn.so (hipcc, gfx950 only) in-tree next to the package."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libconceptattn.so")
SOURCES = ["ca_api.hip", "ca_gemm.hip", "ca_attn.hip", "ca_attn4.hip", "ca_rowops.hip"]
# ca_attn4.hip owns the AGPR file by hand (literal a[...] registers in its asm statements): hipcc must never park a
# VGPR there (its default spill target), and the emitted code is audited for it below
EXTRA_FLAGS = {"ca_attn4.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-save-temps=obj"]}
HEADERS = ["ca_common.h", "ca_attn_common.h", "ca_attn4_sched.inc"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES + HEADERS]
    deps.append(os.path.join(os.path.dirname(PKG), "include", "conceptattn.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def audit_attn4(asm_path: str) -> None:
    """ca_attn4_kernel keeps O, Q and the K/V operand rings in AGPRs it names literally: any AGPR access hipcc emits
    on its own (outside #ASMSTART/#ASMEND), any VGPR spill or scratch use would silently corrupt them."""
    import re
    text = open(asm_path).read()
    inside, bad = False, []
    for line in text.split("\n"):
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        elif not inside and not line.lstrip().startswith((";", ".")) and \
                ("accvgpr" in line or re.search(r"\ba\[?\d", line.split(";")[0]) or
                 re.search(r"\bm0\b", line.split(";")[0])):   # (M0: the tile loop's LDS-DMA writes it without saving it)
            bad.append(line.strip())
    m = re.search(r"\.vgpr_spill_count:\s*(\d+)", text)
    scratch = re.search(r"\.private_segment_fixed_size:\s*(\d+)", text)
    if bad or (m and int(m.group(1))) or (scratch and int(scratch.group(1))):
        raise RuntimeError(f"ca_attn4.hip audit failed: compiler AGPR / M0 accesses {bad[:5]}, vgpr spills "
                           f"{m.group(1) if m else '?'}, scratch {scratch.group(1) if scratch else '?'} bytes")
    for tmp in os.listdir(HERE):   # -save-temps leftovers
        if tmp.startswith("ca_attn4-") and not tmp.endswith(".s"):
            os.remove(os.path.join(HERE, tmp))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:  # compile the translation units in parallel, then link
        o = os.path.join(HERE, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + EXTRA_FLAGS.get(s, []) + \
              ["-c", os.path.join(HERE, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    audit_attn4(os.path.join(HERE, "ca_attn4-hip-amdgcn-amd-amdhsa-gfx950.s"))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
