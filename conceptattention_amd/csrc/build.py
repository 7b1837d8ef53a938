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
HEADERS = ["ca_common.h", "ca_attn_common.h", "ca_attn4_sched.inc", "ca_attn4_kernel.inc"]
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
    bad += audit_sgpr_hazards(text)
    bad += audit_mfma_result_hazards(text)
    bad += audit_asm_statements(text)
    # (every kernel of the unit: the bf16 and the half-precision-q/k instantiation)
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s*(\d+)", text)]
    scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", text)]
    if bad or any(spills) or any(scratch) or not spills:
        raise RuntimeError(f"ca_attn4.hip audit failed: compiler AGPR / M0 accesses / hazards {bad[:5]}, vgpr spills "
                           f"{spills}, scratch {scratch} bytes")
    for tmp in os.listdir(HERE):   # -save-temps leftovers
        if tmp.startswith("ca_attn4-") and not tmp.endswith(".s"):
            os.remove(os.path.join(HERE, tmp))


def audit_sgpr_hazards(text: str) -> list:
    """hipcc pads no hazard of an instruction INSIDE an asm statement.  The one the tile loop's LDS-DMA pieces are
    exposed to: an SGPR written by a VALU instruction (v_readlane = the reload of a spilled SGPR, v_readfirstlane,
    v_cmp) needs 5 wait states before a VMEM instruction reads it (SALU-written SGPRs are interlocked); M0 needs one
    state between its write and the LDS-DMA.  Straight-line check of every asm VMEM instruction against the 5
    instructions in front of it (s_nop N counts N + 1)."""
    import re
    def sregs(tok):
        out = set()
        for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", tok):
            out |= set(range(int(a), int(b) + 1))
        out |= {int(x) for x in re.findall(r"\bs(\d+)\b", tok)}
        return out
    def vregs(tok):
        out = set()
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
            out |= set(range(int(a), int(b) + 1))
        out |= {int(x) for x in re.findall(r"\bv(\d+)\b", tok)}
        return out
    vhist = []                             # (wait states, VALU-written VGPRs, text): a VGPR written by a VALU instruction
                                           # needs 2 wait states before an MFMA reads it as A / B / C
    hist, bad, inside = [], [], False      # hist: (wait states, VALU-written SGPRs, writes M0, text)
    for line in text.split("\n"):
        code = line.split(";")[0].strip()
        if "#ASMSTART" in line:
            inside = True
            continue
        if "#ASMEND" in line:
            inside = False
            continue
        if not code or code.endswith(":") or code.startswith("."):
            continue
        op = code.split()[0]
        args = code[len(op):]
        if inside and op.startswith(("buffer_", "global_", "flat_")):
            need, states = sregs(args), 0
            for ws, written, wm0, txt in reversed(hist):
                if states >= 5:
                    break
                if written & need:
                    bad.append(f"{txt} -> {code} ({states} wait states)")
                if wm0 and states < 1 and " lds" in code + " ":
                    bad.append(f"{txt} -> {code} (M0, {states} wait states)")
                states += ws
        if inside and op.startswith("v_mfma"):
            need, states = vregs(",".join(args.split(",")[1:])), 0
            for ws_, written, txt in reversed(vhist):
                if states >= 2:
                    break
                if written & need:
                    bad.append(f"{txt} -> {code} ({states} wait states)")
                states += ws_
        ws = int(args.split()[0], 0) + 1 if op == "s_nop" else 1
        is_valu = op.startswith("v_") and not op.startswith(("v_mfma", "v_accvgpr_write", "v_cmp", "v_readlane",
                                                              "v_readfirstlane"))
        vhist.append((ws, vregs(args.split(",")[0]) if is_valu else set(), code))
        vhist = vhist[-6:]
        first = args.split(",")[0]
        valu_sgpr_write = sregs(first) if op.startswith("v_") else set()
        if op.startswith("v_cmp") and "_e64" not in op and "s" not in first:
            valu_sgpr_write = set()                                    # writes VCC
        hist.append((ws, valu_sgpr_write, re.search(r"\bm0\b", first) is not None, code))
        hist = hist[-12:]
    return bad


def _regs(tok: str) -> set:
    """('v' | 'a', n) for every vector / accumulator register an operand string names."""
    import re
    out = set()
    for c, a, b in re.findall(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {(c, i) for i in range(int(a), int(b) + 1)}
    out |= {(c, int(x)) for c, x in re.findall(r"\b([va])(\d+)\b", tok)}
    return out


def _mfma_passes(op: str) -> int:
    # gfx950: v_mfma_f32_32x32x16_{bf16,f16} and the 16x16x128 f8f6f4 forms hold the matrix pipe for 8 passes of 4
    # cycles, the 16x16x32 forms for 4; anything unknown is taken as the longest (16)
    if "32x32x16" in op or "16x16x128" in op:
        return 8
    if "16x16x32" in op:
        return 4
    return 16


def audit_mfma_result_hazards(text: str) -> list:
    """The hazard the other audits do not see: the RESULT of an MFMA that sits inside an asm statement, touched too early.
    gfx950 does not interlock it and hipcc pads nothing around an instruction it cannot see: a register written by an
    N-pass MFMA may be read or written by a VALU / LDS / VMEM instruction, or read by another MFMA as A or B, only
    N + 3 wait states after the MFMA issued (8 passes: 11).  An MFMA that accumulates onto it (C = D = the same
    tuple) is interlocked by the hardware.

    Time model (one unit = one wait state = one issue slot of the wave): every instruction takes 1, `s_nop N` N + 1,
    and an MFMA issues no earlier than N after the previous MFMA issued (one matrix pipe per SIMD) -- which is what
    lets the generated stream read a score two MFMA slots after the chain's last MFMA.  Checked for EVERY instruction,
    inside or outside #ASMSTART (a copy or a re-ordered instruction hipcc places between two asm statements is exactly
    the failure of commit 863ff6e).  Control flow: the scan is linear in layout order (fall-through), and the state at
    every branch is carried to the branch target as well -- the R = 2 -> R = 0 back-edge of the tile loop included.

    Second rule, for the generated stream (statements marked `; a4s` by tools/gen_attn4_schedule.py): between two marked
    statements of one basic block hipcc may place scalar instructions and s_nop only; any vector, LDS or memory
    instruction there means it moved or copied a register the schedule owns."""
    import re
    ins = []          # (op, args, inside_asm, marked, raw)
    labels = {}
    inside = marked = False
    for line in text.split("\n"):
        if "#ASMSTART" in line:
            inside, marked = True, False
            continue
        if "#ASMEND" in line:
            inside = False
            continue
        if inside and "; a4s" in line:
            marked = True
        code = line.split(";")[0].strip()
        if not code or code.startswith("."):
            m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", code)
            if m:
                labels[m.group(1)] = len(ins)
                ins.append(("label", m.group(1), False, False, code))
            continue
        m = re.match(r"^([A-Za-z_][\w.$]*):$", code)
        if m:
            labels[m.group(1)] = len(ins)
            ins.append(("label", m.group(1), False, False, code))
            continue
        op = code.split()[0]
        ins.append((op, code[len(op):], inside, inside and marked, code))
    bad = []

    def scan(start, t, pipe_free, pending, limit, snapshots):
        """pending: list of (ready_time, regs, text).  Returns nothing; appends to bad."""
        n = 0
        for i in range(start, len(ins)):
            op, args, in_asm, _, raw = ins[i]
            if op == "label":
                continue
            if limit is not None:
                n += 1
                if n > limit or not any(r > t for r, _, _ in pending):
                    return
            pending = [p for p in pending if p[0] > t]
            if op == "s_nop":
                t += int(args.split()[0], 0) + 1
                continue
            if op.startswith(("s_endpgm",)):
                return
            if op.startswith("v_mfma") or op.startswith("v_smfmac"):
                parts = [x.strip() for x in args.split(",")]
                d, a, b, c = (_regs(parts[0]), _regs(parts[1]), _regs(parts[2]), _regs(parts[3]) if len(parts) > 3 else set())
                t = max(t, pipe_free)
                for ready, regs, txt in pending:
                    if ready > t and ((a | b) & regs or ((c | d) & regs and not (c == regs and d == regs))):
                        bad.append(f"{txt} -> {raw} (MFMA result used {ready - t} wait states early)")
                passes = _mfma_passes(op)
                pipe_free = t + passes
                if in_asm:   # hipcc knows (and pads) the MFMAs it emits itself
                    pending.append((t + 1 + passes + 3, d, raw))
                t += 1
                continue
            touched = _regs(args)
            if touched:
                for ready, regs, txt in pending:
                    if ready > t and touched & regs:
                        bad.append(f"{txt} -> {raw} (MFMA result used {ready - t} wait states early)")
            if snapshots is not None and op.startswith(("s_cbranch", "s_branch")):
                snapshots.append((args.strip(), t + 1, pipe_free, list(pending)))
            t += 1

    snaps = []
    scan(0, 0, 0, [], None, snaps)
    for target, t, pipe_free, pending in snaps:
        if target in labels and any(r > t for r, _, _ in pending):
            scan(labels[target], t, pipe_free, pending, 64, None)
    # ---- nothing but scalar instructions between two stream statements of one basic block
    prev_marked = False
    between = []
    for op, args, in_asm, mk, raw in ins:
        if op == "label" or op.startswith(("s_cbranch", "s_branch")):
            prev_marked, between = False, []
            continue
        if mk:
            if prev_marked:
                bad += [f"{r} (compiler instruction inside the generated stream)" for r in between]
            prev_marked, between = True, []
            continue
        if in_asm:          # an unmarked asm statement (helpers): ends the stream region
            prev_marked, between = False, []
            continue
        if prev_marked and op.startswith(("v_", "ds_", "buffer_", "global_", "scratch_", "flat_")):
            between.append(raw)
    return sorted(set(bad))


# the asm statements ca_attn4.hip is made of, outside the generated stream (which is marked `; a4s` per statement): what
# each helper emits, instruction by instruction.  The hazard rules above were worked out for these shapes; a statement
# with an MFMA, an LDS or a memory instruction that is none of them is new code nobody has audited.
KNOWN_ASM_STATEMENTS = {
    ("v_mfma_f32_32x32x16_bf16",): "one MFMA of the plain K Q^T / P V helpers",
    ("v_mfma_f32_32x32x16_f16",): "the same, half-precision q / k",
    ("ds_read_b128",): "a K fragment read of the plain helpers / the loop's preload",
    ("ds_read_b64_tr_b16",): "a V fragment read of the plain P V helper",
    ("s_mov_b32", "s_mov_b32", "s_nop", "global_load_lds_dwordx4", "s_mov_b32"): "ca_glds16_asm: M0 saved, one LDS-DMA piece, M0 restored",
    ("s_mov_b32", "s_nop", "buffer_load_dwordx4"): "stage_pieces: one LDS-DMA piece through a buffer descriptor",
}


def audit_asm_statements(text: str) -> list:
    """Every asm statement that contains an MFMA, an LDS or a memory instruction is either part of the generated stream
    (marked) or one of KNOWN_ASM_STATEMENTS; anything else fails the build until it has been looked at and listed."""
    bad, cur = [], None
    for line in text.split("\n"):
        if "#ASMSTART" in line:
            cur = []
            continue
        if "#ASMEND" in line:
            if cur is not None and not any("; a4s" in l for l in cur):
                ops = tuple(l.split(";")[0].split()[0] for l in cur if l.split(";")[0].strip())
                hot = any(o.startswith(("v_mfma", "v_smfmac", "buffer_", "global_", "flat_", "scratch_", "ds_")) for o in ops)
                if hot and ops not in KNOWN_ASM_STATEMENTS:
                    bad.append("unclassified asm statement: " + " ; ".join(ops))
            cur = None
            continue
        if cur is not None:
            cur.append(line.strip())
    return sorted(set(bad))


def build(force: bool = False, verbose: bool = True, defines=(), out: str | None = None) -> str:
    """The product library (in-tree), or with ``defines`` / ``out`` a diagnostic build elsewhere (build_ab)."""
    if out is not None or defines:
        return _build_variant(list(defines), out or LIB, verbose)
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


AB_LIB = os.path.join(os.path.dirname(PKG), "tools", "ab", "switches", "libca.so")


def _build_variant(defines, out, verbose):
    """Objects next to ``out`` (never over the product's), same flags and the same ca_attn4 audit."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    odir = os.path.dirname(os.path.abspath(out))
    os.makedirs(odir, exist_ok=True)
    objs, procs = [], []
    for s in SOURCES:
        o = os.path.join(odir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + [f"-D{d}" for d in defines] + \
              [f for f in EXTRA_FLAGS.get(s, []) if not f.startswith("-save-temps")] + ["-c", os.path.join(HERE, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_ab(verbose: bool = True) -> str:
    """The diagnostic build with the A/B environment switches compiled in (-DCA_AB_SWITCHES): tools/ab/switches/libca.so,
    loaded through CA_LIB_PATH by tools/env_ab.py, tools/attn_peaky.py, tools/gemm_group_m.py, tools/ln_rows_ab.py.  The
    product library (libconceptattn.so) has none of these switches."""
    return build(verbose=verbose, defines=["CA_AB_SWITCHES"], out=AB_LIB)


if __name__ == "__main__":
    if "--ab" in sys.argv:
        print(build_ab())
    else:
        build(force="--force" in sys.argv)
