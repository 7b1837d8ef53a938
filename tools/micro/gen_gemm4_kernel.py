#!/usr/bin/env python3
"""Generates gemm4_kernel_probe.inc: the instruction stream of ONE 64-wide K tile of gemm4_kernel_probe.hip -- a
one-wave-per-SIMD GEMM prototype (4 waves, 128 x 128 outputs per wave in a[0:255], v_mfma_f32_16x16x32_bf16 issued
transposed: W fragment as the A operand, so a lane owns 4 consecutive output columns of one row).

K tile t lives in LDS stage t & 1 (A rows at stage * 32768, W rows at 65536 + stage * 32768; 128-byte rows, 16-byte
chunks XOR-swizzled by (row >> 1) & 7).  The stream of tile t (variant ST = t & 1, 128 MFMAs):
    k32 step 0, MFMAs 0..63    + the 16 fragment reads of step 1 (one per four MFMAs, this stage)
    k32 step 1, MFMAs 0..31
    s_waitcnt vmcnt(0) ; s_barrier     tile t+1 (requested one tile ago) has landed everywhere, and every wave has
                                       issued -- and waited for -- its last read of this stage
    k32 step 1, MFMAs 32..63   + the 16 fragment reads of tile t+1's step 0 (other stage), and this wave's 16 LDS-DMA
                                 pieces of tile t+2 into THIS stage (M0 = piece address, buffer descriptor + the tile's
                                 byte offset as scalar offset)
Fragments: double buffer FA / FW [2][8] (step parity).  Names in scope in the kernel: FA, FW, aaddr[2][8], waddr[2][8]
(byte addresses of this lane's 16 bytes per step and fragment, stage 0), aoff[8], woff[8] (lane offsets of the DMA
pieces), DSA, DSW (descriptors), SO (byte offset of tile t+2 along K), LW (LDS address of this wave's first piece)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
MFMA = "v_mfma_f32_16x16x32_bf16"
KO = set(filter(None, os.environ.get("GEMM4_KO", "").split(",")))   # timing-only knock-outs: vmcnt, barrier, dma, lds
OUTNAME = os.environ.get("GEMM4_OUT", "gemm4_kernel_probe.inc")
# "bunched" (round 3): one barrier per K tile in the middle of step 1, the 16 DMA pieces and the 16 reads of the next tile
# share the 32 slots behind it.  "spread" (round 4, after reading how hipBLASLt's hand-written 4-wave kernel does it --
# profiles/r04_vendor_kernel_names.json): once a wave's step-1 fragments are in registers the WHOLE stage is dead (its
# step-0 fragments were read a tile ago), so a barrier at the top of step 1 frees it and the 16 pieces of tile t+2 go out
# one per four MFMAs over all 64 slots of step 1; a second barrier in the middle (vmcnt(8): the 8 pieces issued so far
# may stay in flight) publishes tile t+1 for the 16 reads of its step 0.
SCHED = os.environ.get("GEMM4_SCHED", "spread")


def mfma(i, j):
    a = 4 * (8 * i + j)
    return f"{MFMA} a[{a}:{a + 3}], %{{w}}, %{{a}}, a[{a}:{a + 3}]"


def gen(st):
    cur_off, nxt_off = st * 32768, (1 - st) * 32768
    L = [f"  // ---- stage {st}"]
    allf = lambda b: ", ".join([f'"+v"(FA[{b}][{f}])' for f in range(8)] + [f'"+v"(FW[{b}][{f}])' for f in range(8)])
    # step 0: fragments in buffer 0 (read during the previous tile's second half), prefetch step 1 into buffer 1
    L.append(f'    asm volatile("s_waitcnt lgkmcnt(0)" : {allf(0)} :: "memory");')
    n = 0
    for i in range(8):
        for j in range(8):
            m = mfma(i, j)
            if n % 4 == 0:
                f = n // 4
                dst = f"FA[1][{f}]" if f < 8 else f"FW[1][{f - 8}]"
                adr = f"aaddr[1][{f}]" if f < 8 else f"waddr[1][{f - 8}]"
                L.append(f'    asm volatile("{m.format(w="1", a="2")}\\n\\tds_read_b128 %0, %3 offset:{cur_off}" : "=&v"({dst}) : '
                         f'"v"(FW[0][{j}]), "v"(FA[0][{i}]), "v"({adr}) : "memory");')
            else:
                L.append(f'    asm volatile("{m.format(w="0", a="1")}" : : "v"(FW[0][{j}]), "v"(FA[0][{i}]));')
            n += 1
    # step 1: fragments in buffer 1
    L.append(f'    asm volatile("s_waitcnt lgkmcnt(0)" : {allf(1)} :: "memory");')
    if SCHED == "spread" and "barrier" not in KO:
        L.append('    asm volatile("s_barrier" ::: "memory");   // every wave holds its step-1 fragments: this stage is free')
    n = 0
    piece = 0
    for i in range(8):
        for j in range(8):
            m = mfma(i, j)
            if n == 32:
                if SCHED == "spread":
                    sync = [] if "vmcnt" in KO else ["s_waitcnt vmcnt(8)"]
                else:
                    sync = [] if "vmcnt" in KO else ["s_waitcnt vmcnt(0)"]
                sync += [] if "barrier" in KO else ["s_barrier"]
                if sync:
                    L.append('    asm volatile("' + "\\n\\t".join(sync) + '" ::: "memory");')
            is_read = n >= 32 and n % 2 == 0 and "lds" not in KO
            if SCHED == "spread":
                is_dma = n % 4 == 1 and "dma" not in KO
            else:
                is_dma = n >= 32 and n % 2 == 1 and "dma" not in KO
            if is_read:          # a fragment read of the next tile's step 0 (other stage) -> buffer 0
                f = (n - 32) // 2
                dst = f"FA[0][{f}]" if f < 8 else f"FW[0][{f - 8}]"
                adr = f"aaddr[0][{f}]" if f < 8 else f"waddr[0][{f - 8}]"
                L.append(f'    asm volatile("{m.format(w="1", a="2")}\\n\\tds_read_b128 %0, %3 offset:{nxt_off}" : "=&v"({dst}) : '
                         f'"v"(FW[1][{j}]), "v"(FA[1][{i}]), "v"({adr}) : "memory");')
            elif is_dma:                       # an LDS-DMA piece of tile t+2 into this stage: A pieces 0..7, W pieces 0..7
                p = piece
                piece += 1
                isw = p >= 8
                dst_off = (65536 if isw else 0) + cur_off + 1024 * (p & 7)
                L.append(f'    asm volatile("s_add_u32 m0, %2, {dst_off}\\n\\t{m.format(w="0", a="1")}\\n\\t'
                         f'buffer_load_dwordx4 %3, %4, %5 offen lds" : : "v"(FW[1][{j}]), "v"(FA[1][{i}]), "s"(LW), '
                         f'"v"({"woff" if isw else "aoff"}[{p & 7}]), "s"({"DSW" if isw else "DSA"}), "s"(SO) : "memory", "scc");')
            else:
                L.append(f'    asm volatile("{m.format(w="0", a="1")}" : : "v"(FW[1][{j}]), "v"(FA[1][{i}]));')
            n += 1
    assert piece == 16 or "dma" in KO
    return L


out = ["// GENERATED by tools/micro/gen_gemm4_kernel.py -- included twice (ST = 0, 1)"]
for st in range(2):
    out.append(f"#if ST == {st}")
    out += gen(st)
    out.append("#endif")
# epilogue helper: accumulator block (i, j) -> 4 floats
out.append("#ifdef GEMM4_READ_ACC")
for b in range(64):
    out.append(f'#define GEMM4_READ_BLOCK_{b}(v) asm volatile("v_accvgpr_read_b32 %0, a{4*b}\\n\\tv_accvgpr_read_b32 %1, a{4*b+1}\\n\\t'
               f'v_accvgpr_read_b32 %2, a{4*b+2}\\n\\tv_accvgpr_read_b32 %3, a{4*b+3}" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]))')
zero = "\\n\\t".join(f"v_accvgpr_write_b32 a{i}, 0" for i in range(256))
out.append(f'#define GEMM4_ZERO() asm volatile("{zero}\\n\\ts_nop 7")')
out.append("#define GEMM4_AGPR_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(256)))
out.append("#endif")
open(os.path.join(HERE, OUTNAME), "w").write("\n".join(out) + "\n")
print("wrote", OUTNAME, len(out), "lines")
