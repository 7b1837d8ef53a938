#!/usr/bin/env python3
"""Generates conceptattention_amd/csrc/ca_attn4_sched.inc: the hand-placed instruction stream of ONE pipelined K/V tile of
the one-wave-per-SIMD attention kernel (ca_attn4_kernel in ca_attn.hip), as a sequence of single-instruction
`asm volatile` statements (hipcc keeps volatile asm statements in program order and only allocates the VGPR operands).

The stream of iteration t (64 MFMAs = one 64-key tile for the wave's 64 query rows):

    slots  0..31   S(t+1) = K(t+1) Q^T       (4 chains of 8 MFMAs; a chain's first MFMA takes -reference as C)
    slots 32..63   O^T   += V(t)^T P(t)^T    (16 V fragments, each used for both 32-row query blocks)

and in the gaps between the MFMAs, placed by the tables below:
    * K(t+1) key-block-1 fragments and K(t+2) key-block-0 fragments (ds_read_b128 into the 8-fragment AGPR ring),
      V(t) fragments (2 x ds_read_b64_tr_b16 each, 8-fragment AGPR ring), each read >= 7 MFMA slots before its use and
      one slot after the last MFMA that reads the ring entry it overwrites;
    * the softmax of S(t+1): one v_exp_f32 per score (the scores arrive as s - reference), the row-sum add one
      instruction later (transcendental -> VALU use needs a wait state), v_cvt_pk_bf16_f32 of a P fragment only after the
      last P.V MFMA of tile t that reads that fragment's registers;
    * counted s_waitcnt lgkmcnt(N): the LDS returns data in order, N = reads issued after the one the MFMA needs.

AGPR map (hand-owned, see ca_attn.hip): O a[0:127], Q a[128:191], K ring a[192:223], V ring a[224:255].
Register-class rule of the MFMA encoding: C and D share one class, A and B are free -> S (read by v_exp) and -reference
live in VGPRs, O in AGPRs.

    python tools/gen_attn4_schedule.py          # rewrites the .inc
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "conceptattention_amd", "csrc", "ca_attn4_sched.inc")

AO, AQ, AK, AV = 0, 128, 192, 224
TILE = 16384
V_BASE_IN_ADDR = True   # the V address registers already contain the V ring's base


def areg(base, n):
    return f"a[{base}:{base + n - 1}]"


class Stream:
    """One `asm volatile` statement per MFMA slot: the MFMA and every filler of the gap behind it in one statement, so
    hipcc cannot put its own padding (it adds an s_nop behind an asm statement whose output the next one reads, blind
    to what is inside) between them.  Operands are collected by C expression: written only -> "=&v" (early clobber: a
    multi-instruction statement may write an output before it has read all its inputs), read and written -> "+v"."""

    def __init__(self):
        self.lines = []
        self.lds = 0           # LDS reads issued so far (in order)
        self.tag = {}          # name -> count of reads issued when that read was issued
        self.cur = []          # instructions of the open statement: (template, [(expr, mode)])

    def ins(self, template, *ops):
        """template uses {0}, {1}.. for the operands; ops = (expr, 'r' | 'w' | 'rw')."""
        self.cur.append((template, ops))

    def flush(self):
        if not self.cur:
            return
        mode = {}
        order = []
        for _, ops in self.cur:
            for e, m in ops:
                if e not in mode:
                    mode[e] = set()
                    order.append(e)
                mode[e] |= set(m)
        outs = [e for e in order if "w" in mode[e]]
        inps = [e for e in order if "w" not in mode[e]]
        num = {e: i for i, e in enumerate(outs + inps)}
        text = []
        for tpl, ops in self.cur:
            text.append(tpl.format(*[f"%{num[e]}" for e, _ in ops]))
        o = ", ".join(f'"{"+v" if "r" in mode[e] else "=&v"}"({e})' for e in outs)
        i = ", ".join(f'"v"({e})' for e in inps)
        body = "\\n\\t".join(text)
        self.lines.append(f'    asm volatile("{body}" : {o} : {i} : "memory");')
        self.cur = []

    def comment(self, c):
        self.lines.append(f"    // {c}")

    def read_k(self, ring, off, ks, name):
        self.ins(f"ds_read_b128 {areg(AK + 4 * ring, 4)}, {{0}} offset:{off}", (f"ka{ks}", "r"))
        self.lds += 1
        self.tag[name] = self.lds

    def read_v(self, ring, vslot_off, kb, sk, db, name):
        row = (32 * kb + 16 * sk) * 256
        self.ins(f"ds_read_b64_tr_b16 {areg(AV + 4 * ring, 2)}, {{0}} offset:{vslot_off + row}", (f"va0{db}", "r"))
        self.ins(f"ds_read_b64_tr_b16 {areg(AV + 4 * ring + 2, 2)}, {{0}} offset:{vslot_off + row + 2048}",
                 (f"va1{db}", "r"))
        self.lds += 2
        self.tag[name] = self.lds

    def wait_for(self, name):
        self.ins(f"s_waitcnt lgkmcnt({self.lds - self.tag[name]})")


def S(kb, qb):
    return f"S{kb}{qb}"


def gen_iteration(r):
    """Iteration t with t % 3 == r: K(t+1) in K slot (r+1)%3, K(t+2) in (r+2)%3, V(t) in V slot r."""
    k1, k2, v0 = ((r + 1) % 3) * TILE, ((r + 2) % 3) * TILE, r * TILE
    st = Stream()
    valu = {s: [] for s in range(64)}
    # exps: S00 at slots 16..23 (2 per slot), S01 at 24..31, then S10[0..7] 32..39, S11[0..7] 40..47,
    # S10[8..15] 48..55, S11[8..15] 56..63 (1 per slot); the row-sum add of a value follows the NEXT exponential
    # (transcendental -> VALU use needs a wait state)
    order = []
    for i in range(16):
        order.append((16 + i // 2, 0, 0, i))
    for i in range(16):
        order.append((24 + i // 2, 0, 1, i))
    for i in range(8):
        order.append((32 + i, 1, 0, i))
    for i in range(8):
        order.append((40 + i, 1, 1, i))
    for i in range(8):
        order.append((48 + i, 1, 0, 8 + i))
    for i in range(8):
        order.append((56 + i, 1, 1, 8 + i))
    pending_add = None
    for slot, kb, qb, i in order:
        valu[slot].append(("exp", kb, qb, i))
        if pending_add is not None:
            valu[slot].append(pending_add)
        pending_add = ("add", kb, qb, i)
    last_add = pending_add
    # packs: P[kb][qb][sk] dword j <- (S[kb][qb][8sk+2j], S[kb][qb][8sk+2j+1]); only after the last P.V MFMA of tile t
    # that reads P[kb][qb][sk] (slot 32 + 2*(kb*8+sk*4+3) + qb)
    def cvts(slots, kb, sk):
        lst = [(kb, qb, sk, j) for qb in (0, 1) for j in range(4)]
        per = len(lst) // len(slots)
        for n, s in enumerate(slots):
            for c in lst[n * per:(n + 1) * per]:
                valu[s].append(("cvt",) + c)
    cvts([0, 1, 2, 3], 1, 1)          # of the PREVIOUS pass over S1x[8..15]: first thing in the iteration
    cvts([40, 41, 42, 43], 0, 0)
    cvts([48, 49, 50, 51], 0, 1)
    cvts([58, 59, 60, 61], 1, 0)
    reads = {s: [] for s in range(64)}
    for ks in range(8):
        reads[9 + ks].append(("k", ks, k1 + 8192, ks, f"k1_{ks}"))        # K(t+1) key block 1, fragment ks -> ring ks
    for f in range(8):
        kb, sk, db = f >> 3, (f >> 2) & 1, f & 3
        reads[17 + f].append(("v", f, v0, kb, sk, db, f"v_{f}"))          # V(t) fragments 0..7
    for ks in range(8):
        reads[25 + ks].append(("k", ks, k2, ks, f"k2_{ks}"))              # K(t+2) key block 0 (for the NEXT iteration;
                                                                          # no tile t+2: a stale slot is read, unused)
    for f in range(8, 16):
        kb, sk, db = f >> 3, (f >> 2) & 1, f & 3
        reads[34 + 2 * (f - 8)].append(("v", f - 8, v0, kb, sk, db, f"v_{f}"))
    st.comment(f"---- iteration variant r = {r}: K(t+1) slot {(r + 1) % 3}, K(t+2) slot {(r + 2) % 3}, V(t) slot {r}")
    for s in range(64):
        if s < 32:
            kb, qb, ks = s >> 4, (s >> 3) & 1, s & 7
            if kb == 1 and qb == 0:
                st.wait_for(f"k1_{ks}")
            a = areg(AK + 4 * ks, 4)
            q = areg(AQ + 4 * (qb * 8 + ks), 4)
            if ks == 0:
                st.ins(f"v_mfma_f32_32x32x16_bf16 {{0}}, {a}, {q}, {{1}}", (S(kb, qb), "w"), (f"NM{qb}", "r"))
            else:
                st.ins(f"v_mfma_f32_32x32x16_bf16 {{0}}, {a}, {q}, {{0}}", (S(kb, qb), "rw"))
        else:
            f, qb = (s - 32) >> 1, (s - 32) & 1
            kb, sk, db = f >> 3, (f >> 2) & 1, f & 3
            if qb == 0:
                st.wait_for(f"v_{f}")
            o = areg(AO + 16 * (qb * 4 + db), 16)
            v = areg(AV + 4 * (f & 7), 4)
            st.ins(f"v_mfma_f32_32x32x16_bf16 {o}, {v}, {{0}}, {o}", (f"P{kb}{qb}{sk}", "r"))
        for rd in reads[s]:
            if rd[0] == "k":
                st.read_k(rd[1], rd[2], rd[3], rd[4])
            else:
                st.read_v(rd[1], rd[2], rd[3], rd[4], rd[5], rd[6])
        for v in valu[s]:
            if v[0] == "exp":
                _, kb, qb, i = v
                st.ins("v_exp_f32 {0}, {0}", (f"{S(kb, qb)}[{i}]", "rw"))
            elif v[0] == "add":
                _, kb, qb, i = v
                st.ins("v_add_f32 {0}, {0}, {1}", (f"l{qb}", "rw"), (f"{S(kb, qb)}[{i}]", "r"))
            else:
                _, kb, qb, sk, j = v
                st.ins("v_cvt_pk_bf16_f32 {0}, {1}, {2}", (f"P{kb}{qb}{sk}[{j}]", "w"),
                       (f"{S(kb, qb)}[{8 * sk + 2 * j}]", "r"), (f"{S(kb, qb)}[{8 * sk + 2 * j + 1}]", "r"))
        if s == 63:
            _, kb, qb, i = last_add
            st.ins("s_nop 0")
            st.ins("v_add_f32 {0}, {0}, {1}", (f"l{qb}", "rw"), (f"{S(kb, qb)}[{i}]", "r"))
        st.flush()
    return st.lines


def gen_helpers():
    """Rare-path and prologue / epilogue building blocks (straight-line, not interleaved)."""
    L = []
    L.append("#define CA_A4_AGPR_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(256)))
    L.append("")
    # ---- Q fragments -> AGPR: qw[qb][ks] is a 4-dword view of the bf16x8 fragment
    L.append("#define CA_A4_WRITE_Q(qw) do { \\")
    for qb in range(2):
        for ks in range(8):
            for j in range(4):
                L.append(f'  asm volatile("v_accvgpr_write_b32 a{AQ + 4 * (qb * 8 + ks) + j}, %0" : : "v"(qw[{qb}][{ks}][{j}])); \\')
    L.append('  asm volatile("s_nop 7"); } while (0)')
    L.append("")
    L.append("#define CA_A4_ZERO_O() do { \\")
    for i in range(128):
        L.append(f'  asm volatile("v_accvgpr_write_b32 a{AO + i}, 0"); \\')
    L.append('  asm volatile("s_nop 7"); } while (0)')
    L.append("")
    # ---- O -> VGPRs (epilogue): of[qb][db] f32x16
    L.append("#define CA_A4_READ_O(of) do { asm volatile(\"s_nop 15\\n\\ts_nop 7\"); \\")
    for qb in range(2):
        for db in range(4):
            for r in range(16):
                L.append(f'  asm volatile("v_accvgpr_read_b32 %0, a{AO + 16 * (qb * 4 + db) + r}" : "=v"(of[{qb}][{db}][{r}])); \\')
    L.append("  } while (0)")
    L.append("")
    # ---- O *= alpha (safe path only): al0 / al1 per query block
    L.append("#define CA_A4_SCALE_O(al0, al1) do { float t_; asm volatile(\"s_nop 15\\n\\ts_nop 7\"); \\")
    for qb in range(2):
        for i in range(64):
            a = AO + 64 * qb + i
            L.append(f'  asm volatile("v_accvgpr_read_b32 %0, a{a}\\n\\tv_mul_f32 %0, %0, %1\\n\\tv_accvgpr_write_b32 a{a}, %0" : "=&v"(t_) : "v"(al{qb})); \\')
    L.append('  asm volatile("s_nop 7"); } while (0)')
    L.append("")
    # ---- K(kb0) fragments of a tile into the ring (loop entry): kbase = K slot byte offset (runtime)
    L.append("#define CA_A4_PRELOAD_K0(kbase) do { \\")
    for ks in range(8):
        L.append(f'  asm volatile("ds_read_b128 {areg(AK + 4 * ks, 4)}, %0" : : "v"(ka{ks} + (kbase)) : "memory"); \\')
    L.append('  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)')
    L.append("")
    # ---- plain K Q^T of one tile: ZERO = accumulators start from 0 (first tile / safe path), else from -reference
    for zero in (True, False):
        L.append(f"#define CA_A4_QK_PLAIN_{'ZERO' if zero else 'NEGM'}(kbase) do {{ \\")
        for kb in range(2):
            for ks in range(8):
                L.append(f'  asm volatile("ds_read_b128 {areg(AK + 4 * ks, 4)}, %0 offset:{kb * 8192}" : : "v"(ka{ks} + (kbase)) : "memory"); \\')
            L.append('  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \\')
            for qb in range(2):
                for ks in range(8):
                    a, q = areg(AK + 4 * ks, 4), areg(AQ + 4 * (qb * 8 + ks), 4)
                    if ks == 0 and zero:
                        L.append(f'  asm volatile("v_mfma_f32_32x32x16_bf16 %0, {a}, {q}, 0" : "=v"({S(kb, qb)})); \\')
                    elif ks == 0:
                        L.append(f'  asm volatile("v_mfma_f32_32x32x16_bf16 %0, {a}, {q}, %1" : "=&v"({S(kb, qb)}) : "v"(NM{qb})); \\')
                    else:
                        L.append(f'  asm volatile("v_mfma_f32_32x32x16_bf16 %0, {a}, {q}, %0" : "+v"({S(kb, qb)})); \\')
            if kb == 0:   # the ring is re-filled for key block 1: the MFMAs above must have read their operands
                L.append('  asm volatile("s_nop 7"); \\')
        L.append('  asm volatile("s_nop 15\\n\\ts_nop 7" : "+v"(S00), "+v"(S01), "+v"(S10), "+v"(S11)); } while (0)')
        L.append("")
    # ---- plain O^T += V^T P^T of one tile: vbase = V slot byte offset (runtime)
    L.append("#define CA_A4_PV_PLAIN(vbase) do { asm volatile(\"s_nop 7\"); \\")
    for f in range(16):
        kb, sk, db = f >> 3, (f >> 2) & 1, f & 3
        row = (32 * kb + 16 * sk) * 256
        ring = f & 7
        L.append(f'  asm volatile("ds_read_b64_tr_b16 {areg(AV + 4 * ring, 2)}, %0 offset:{row}" : : "v"(va0{db} + (vbase)) : "memory"); \\')
        L.append(f'  asm volatile("ds_read_b64_tr_b16 {areg(AV + 4 * ring + 2, 2)}, %0 offset:{row + 2048}" : : "v"(va1{db} + (vbase)) : "memory"); \\')
        L.append('  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \\')
        for qb in range(2):
            o = areg(AO + 16 * (qb * 4 + db), 16)
            L.append(f'  asm volatile("v_mfma_f32_32x32x16_bf16 {o}, {areg(AV + 4 * ring, 4)}, %0, {o}" : : "v"(P{kb}{qb}{sk})); \\')
        L.append('  asm volatile("s_nop 3"); \\')
    L.append("  } while (0)")
    L.append("")
    return L


def main():
    out = ["// GENERATED by tools/gen_attn4_schedule.py -- do not edit by hand.",
           "// Part 1 (CA_A4_HELPERS): straight-line building blocks; part 2 (CA_A4_SCHEDULE): one pipelined tile of",
           "// ca_attn4_kernel, included inside the tile loop with these names in scope:",
           "//   f32x16 S00,S01,S10,S11 (scores / fp32 P), NM0,NM1 (-reference), i32x4 P000..P111 (bf16 P fragments),",
           "//   float l0,l1 (row sums), uint32_t ka0..ka7 (K fragment addresses), va00..va13 (V fragment addresses,",
           "//   V ring base included), int R (t % 3).",
           "// A counted lgkmcnt(N) for read X: N = the reads issued after X (the LDS returns data in order).",
           "", "#ifdef CA_A4_HELPERS"]
    out += gen_helpers()
    out += ["#endif  // CA_A4_HELPERS", "", "#ifdef CA_A4_SCHEDULE"]
    for r in range(3):
        out.append(f"  {'if' if r == 0 else 'else if'} (R == {r}) {{")
        out += gen_iteration(r)
        out.append("  }")
    out.append("#endif  // CA_A4_SCHEDULE")
    open(OUT, "w").write("\n".join(out) + "\n")
    print("wrote", OUT, len(out), "lines")


if __name__ == "__main__":
    main()
