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
        cls = lambda e: "s" if "s" in mode[e] else "v"     # mode letter 's': a scalar (SGPR) operand
        o = ", ".join(f'"{"+" if "r" in mode[e] else "=&"}{cls(e)}"({e})' for e in outs)
        i = ", ".join(f'"{cls(e)}"({e})' for e in inps)
        body = "\\n\\t".join(text)
        clob = '"memory", "scc"' if any(t.startswith(("s_cmp", "s_add")) for t in text) else '"memory"'
        self.lines.append(f'    asm volatile("{body}" : {o} : {i} : {clob});')
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
        assert name in self.tag, f"{name} waited for before it was issued"
        self.ins(f"s_waitcnt lgkmcnt({min(self.lds - self.tag[name], 15)})")   # the counter field has 4 bits


def S(kb, qb):
    return f"S{kb}{qb}"


def gen_iteration(r):
    """Iteration t with t % 3 == r: K(t+1) in K slot (r+1)%3, K(t+2) in (r+2)%3, V(t) in V slot r."""
    k1, k2, v0 = ((r + 1) % 3) * TILE, ((r + 2) % 3) * TILE, r * TILE
    st = Stream()
    # ---- the exponentials: ONE per MFMA gap, 64 gaps for the tile's 64 scores per lane.  Order S00, S01, S10, S11
    # starting at slot 10 (a block is exponentiated >= 2 MFMA slots after the MFMA that completes it: S00 at slot 7,
    # S01 at 15, S10 at 23, S11 at 31); the last ten (S11[6..15]) wrap into slots 0..9 of the NEXT iteration -- S11 is
    # not written again before slot 24.  So at the top of an iteration S11 still holds the previous pass's raw
    # (score - reference) values in elements 6..15, and P110 / P111 are not packed yet: ca_attn4.hip hands the loop
    # exactly that state and completes it after the loop (CA_A4 "pending S11" in the kernel).
    order = []
    n = 10
    for kb, qb in ((0, 0), (0, 1), (1, 0), (1, 1)):
        for i in range(16):
            order.append((n, kb, qb, i))
            n += 1
    # hipcc gives every statement AT MOST ONE written element per vector variable for free (a second tied or defined
    # sub-register of the same tuple costs it a copy out and back), and pads an s_nop behind a statement whose output the
    # NEXT statement reads.  So one exponential per statement, and the row-sum add of a value rides two exponentials
    # later (also >= the one wait state a transcendental needs), into two accumulators per query block in turn
    # (l0 / l0b, l1 / l1b): consecutive statements then never read each other's output.
    exps_of = {s: [] for s in range(64)}
    for n, (slot, kb, qb, i) in enumerate(order):
        add = order[n - 2][1:] + ("" if n % 2 == 0 else "b",)      # (n < 2: the previous pass's last two values)
        exps_of[slot % 64].append(((kb, qb, i), add))
    WRAP_FIRST = 64 - 10     # order[54:] are issued at slots 0..9 of the next iteration
    # ---- packs: P[kb][qb][sk] dword j <- (S[kb][qb][8sk+2j], S[kb][qb][8sk+2j+1]); after the values' exponentials and
    # after the last P.V MFMA of tile t that reads P[kb][qb][sk] (slot 32 + 2*(kb*8+sk*4+3) + qb); at most one pack per
    # vector variable and statement.
    cvt_of = {s: [] for s in range(64)}
    def cvts(slots, kb, qb, sk):
        for j, s_ in enumerate(slots):
            cvt_of[s_ % 64].append((kb, qb, sk, j))
    cvts([40, 41, 42, 43], 0, 0, 0)    # S00[0..7]  exponentiated by slot 17; P000 last read at slot 38
    cvts([40, 41, 42, 43], 0, 1, 0)    # S01[0..7]  by slot 33;              P010 at 39
    cvts([48, 49, 50, 51], 0, 0, 1)    # S00[8..15] by 25;                   P001 at 46
    cvts([48, 49, 50, 51], 0, 1, 1)    # S01[8..15] by 41;                   P011 at 47
    cvts([56, 57, 58, 59], 1, 0, 0)    # S10[0..7]  by 49;                   P100 at 54
    cvts([64, 65, 70, 71], 1, 0, 1)    # S10[8..15] by 57;                   P101 last read at slot 62 -> next iteration
    cvts([66, 67, 68, 69], 1, 1, 0)    # S11[0..7]  by 65 (= slot 1 of the next iteration); P110 last read at slot 55
    cvts([76, 77, 78, 79], 1, 1, 1)    # S11[8..15] by 73;                   P111 at 63
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

    def exp_and_add(e):
        (kb, qb, i), add = e
        st.ins("v_exp_f32 {0}, {0}", (f"{S(kb, qb)}[{i}]", "rw"))
        akb, aqb, ai, sfx = add
        st.ins("v_add_f32 {0}, {0}, {1}", (f"l{aqb}{sfx}", "rw"), (f"{S(akb, aqb)}[{ai}]", "r"))

    st.comment(f"---- iteration variant r = {r}: K(t+1) slot {(r + 1) % 3}, K(t+2) slot {(r + 2) % 3}, V(t) slot {r}")
    for s in range(64):
        if s < 32:
            kb, qb, ks = s >> 4, (s >> 3) & 1, s & 7
            if kb == 1 and qb == 0 and ks % 2 == 0:
                st.wait_for(f"k1_{ks + 1}")     # one wait per two fragments
            a = areg(AK + 4 * ks, 4)
            q = areg(AQ + 4 * (qb * 8 + ks), 4)
            if ks == 0:
                st.ins(f"v_mfma_f32_32x32x16_bf16 {{0}}, {a}, {q}, {{1}}", (S(kb, qb), "w"), (f"NM{qb}", "r"))
            else:
                st.ins(f"v_mfma_f32_32x32x16_bf16 {{0}}, {a}, {q}, {{0}}", (S(kb, qb), "rw"))
        else:
            f, qb = (s - 32) >> 1, (s - 32) & 1
            kb, sk, db = f >> 3, (f >> 2) & 1, f & 3
            if qb == 0 and f % 2 == 0:
                st.wait_for(f"v_{f + 1}")       # one wait per two fragments
            o = areg(AO + 16 * (qb * 4 + db), 16)
            v = areg(AV + 4 * (f & 7), 4)
            st.ins(f"v_mfma_f32_32x32x16_bf16 {o}, {v}, {{0}}, {o}", (f"P{kb}{qb}{sk}", "r"))
        if s < 8:
            # LDS-DMA of this wave's 4 + 4 pieces (1 KiB each) of K(t+3) / V(t+1), one per gap: scalar tile base + lane
            # offset, M0 = the piece's LDS address.  (hipcc itself never uses M0 in this kernel: build.py checks that,
            # so it is neither saved nor restored.)  DMAK / DMAV are 64-bit tile bases, LDK / LDV the LDS address of the
            # wave's first piece; for a tile that does not exist, or that ca_attn4.hip staged the general way, they
            # point at a valid tile and at the dump page, so no branch is needed here.
            j, isv = s >> 1, s & 1
            if os.environ.get("CA_A4_GEN_SKIP"):     # bisecting aid: skip the piece when the base is 0
                st.ins("s_cmp_eq_u64 {0}, 0", ("DMAV" if isv else "DMAK", "rs"))
                st.ins("s_cbranch_scc1 .Lca4_nodma%=")
            st.ins(f"s_add_u32 m0, {{0}}, {1024 * j}", ("LDV" if isv else "LDK", "rs"))
            # M0 write -> LDS-DMA: one wait state.  And the tile base: an SGPR written by SALU or VALU needs 5 wait
            # states before a VMEM instruction may use it as its address; hipcc pads nothing inside an asm statement and
            # may write the pair right in front of ANY of these statements (scalar bookkeeping in front of the first
            # piece, a v_readlane reload of a spilled SGPR in front of any other).  With one state here the pieces
            # intermittently read through a half-updated base (address 0xffff....: a fault that came and went).
            st.ins("s_nop 4")
            st.ins("global_load_lds_dwordx4 {0}, {1}", (f"{'voff' if isv else 'koff'}{j}", "r"),
                   ("DMAV" if isv else "DMAK", "rs"))
            if os.environ.get("CA_A4_GEN_SKIP"):
                st.ins(".Lca4_nodma%=:")
        for rd in reads[s]:
            if rd[0] == "k":
                st.read_k(rd[1], rd[2], rd[3], rd[4])
            else:
                st.read_v(rd[1], rd[2], rd[3], rd[4], rd[5], rd[6])
        st.flush()      # statement A: wait, MFMA, LDS reads, DMA.  Statement B: the VALU fillers -- with a statement
                        # between two MFMAs of one accumulation chain hipcc has no reason to pad an s_nop between them
        seen = set()
        for (kb, qb, sk, j) in cvt_of[s]:
            assert (kb, qb, sk) not in seen        # one written element per vector variable and statement
            seen.add((kb, qb, sk))
            st.ins("v_cvt_pk_bf16_f32 {0}, {1}, {2}", (f"P{kb}{qb}{sk}[{j}]", "w"),
                   (f"{S(kb, qb)}[{8 * sk + 2 * j}]", "r"), (f"{S(kb, qb)}[{8 * sk + 2 * j + 1}]", "r"))
        assert len(exps_of[s]) == 1
        exp_and_add(exps_of[s][0])
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
    # ---- O -> VGPRs (epilogue), one query block at a time: of[db] f32x16
    for qb in range(2):
        L.append(f"#define CA_A4_READ_O{qb}(of) do {{ asm volatile(\"s_nop 15\\n\\ts_nop 7\"); \\")
        for db in range(4):
            for r in range(16):
                L.append(f'  asm volatile("v_accvgpr_read_b32 %0, a{AO + 16 * (qb * 4 + db) + r}" : "=v"(of[{db}][{r}])); \\')
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
           "//   float l0,l0b,l1,l1b (row sums, two accumulators per query block), uint32_t ka0..ka7 (K fragment addresses), va00..va13 (V fragment addresses,",
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
