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
OUT = os.environ.get("CA_A4_OUT") or os.path.join(ROOT, "conceptattention_amd", "csrc", "ca_attn4_sched.inc")
# timing-only knock-outs (tools/attn4_knockouts.py; the results of such a build are wrong): a comma list of
#   nop4 (s_nop 0 in place of the SGPR->VMEM s_nop 4), exp (v_mov for v_exp), add, cvt, lds (no LDS reads / waits), dma
KO = set(filter(None, os.environ.get("CA_A4_KO", "").split(",")))
# placement inside a gap (A/B aid): "split" = [MFMA, memory] | [VALU] as two statements (shipped), "valu_first" /
# "mem_first" = one statement per slot
ORDER = os.environ.get("CA_A4_ORDER", "split")

AO, AQ, AK, AV = 0, 128, 192, 224
QK_T = '" CA_A4_QK_T "'   # spliced into the asm string literal: ..."v_mfma_f32_32x32x16_" CA_A4_QK_T " %0, ..."
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
        # marker of a stream statement (a comment in the emitted assembly): build.py's audit fails the build on any
        # vector / memory instruction hipcc places BETWEEN two marked statements of one basic block
        text[0] += " ; a4s"
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
        if "lds" in KO:
            self.tag[name] = self.lds
            return
        self.ins(f"ds_read_b128 {areg(AK + 4 * ring, 4)}, {{0}} offset:{off}", (f"ka{ks}", "r"))
        self.lds += 1
        self.tag[name] = self.lds

    def read_v(self, ring, vslot_off, kb, sk, db, name):
        row = (32 * kb + 16 * sk) * 256
        if "lds" in KO:
            self.tag[name] = self.lds
            return
        self.ins(f"ds_read_b64_tr_b16 {areg(AV + 4 * ring, 2)}, {{0}} offset:{vslot_off + row}", (f"va0{db}", "r"))
        self.ins(f"ds_read_b64_tr_b16 {areg(AV + 4 * ring + 2, 2)}, {{0}} offset:{vslot_off + row + 2048}",
                 (f"va1{db}", "r"))
        self.lds += 2
        self.tag[name] = self.lds

    def wait_for(self, name):
        assert name in self.tag, f"{name} waited for before it was issued"
        if "lds" in KO:
            return
        self.ins(f"s_waitcnt lgkmcnt({min(self.lds - self.tag[name], 15)})")   # the counter field has 4 bits


def S(kb, qb):
    return f"S{kb}{qb}"


def gen_iteration(r):
    """Iteration t with t % 3 == r: K(t+1) in K slot (r+1)%3, K(t+2) in (r+2)%3, V(t) in V slot r; the LDS-DMA of this
    iteration fills K slot r with K(t+3) and V slot (r+1)%3 with V(t+1).

    MFMA slots: 0..31 S(t+1) = K(t+1) Q^T with the two query blocks' chains INTERLEAVED (slot s: key block s>>4, K
    fragment (s>>1)&7, query block s&1 -- a chain's MFMAs are two slots apart, so none waits for its predecessor's
    result; back to back they cost 3.75 cycles each), 32..63 O^T += V(t)^T P(t)^T (fragment (s-32)>>1, query block s&1).
    """
    k1, k2, v0 = ((r + 1) % 3) * TILE, ((r + 2) % 3) * TILE, r * TILE
    kdst, vdst = r * TILE, 3 * TILE + ((r + 1) % 3) * TILE     # LDS byte offsets of the DMA destinations (wave part added
                                                               # through LWK / LWV)
    st = Stream()
    # ---- the exponentials: ONE per MFMA gap.  S(kb, qb) is complete after slot 14 + qb + 16 kb; its 16 values are
    # exponentiated in the 16 gaps that start two slots later: S00 in gaps 16..31, S01 32..47, S10 48..63, S11 64..79 =
    # gaps 0..15 of the NEXT iteration (S11 is not written again before slot 17).  So at the top of an iteration S11
    # still holds the previous pass's raw (score - reference) values and P101 / P110 / P111 are not packed yet:
    # ca_attn4.hip hands the loop exactly that state and completes it after the loop ("pending" in the kernel).
    order = [(kb, qb, i) for kb, qb in ((0, 0), (0, 1), (1, 0), (1, 1)) for i in range(16)]
    # hipcc gives every statement AT MOST ONE written element per vector variable for free (a second tied or defined
    # sub-register of the same tuple costs it a copy out and back), and pads an s_nop behind a statement whose output the
    # NEXT statement reads.  So one exponential per statement, and the row-sum add of a value rides two exponentials
    # later (also >= the one wait state a transcendental needs), into two accumulators per query block in turn
    # (l0 / l0b, l1 / l1b): consecutive statements then never read each other's output.
    exps_of, adds_of = {}, {g: [] for g in range(64)}
    for n, e in enumerate(order):
        a = order[(n - 2) % 64]
        g = (16 + n) % 64
        exps_of[g] = e
        # (the last value of S11 is added one gap early, together with its predecessor: the S11 chain restarts at slot
        # 17, and an add in gap 17 would read S11[15] while that MFMA is in flight -- correct only as long as nothing
        # delays the add past the MFMA's write-back: a result that came and went with the LDS queue's state)
        adds_of[g - 1 if a == (1, 1, 15) else g].append(a + ("" if n % 2 == 0 else "b",))
    # ---- packs: P[kb][qb][sk] dword j <- (S[kb][qb][8sk+2j], S[kb][qb][8sk+2j+1]); after the values' exponentials (at
    # least one gap later) and after the last P.V MFMA of tile t that reads P[kb][qb][sk] (slots 32 + 16 kb + 8 sk ..+7);
    # at most one pack per vector variable and statement; on gaps without V reads where there is a choice.
    cvt_of = {s_: [] for s_ in range(64)}
    def cvts(slots, kb, qb, sk):
        for j, s_ in enumerate(slots):
            cvt_of[s_ % 64].append((kb, qb, sk, j))
    cvts([40, 42, 44, 46], 0, 0, 0)    # S00[0..7]  exponentiated by gap 23; P000 last read at slot 38
    cvts([40, 42, 44, 46], 0, 1, 0)    # S01[0..7]  by 39;                   P010 at 39
    cvts([48, 50, 52, 54], 0, 0, 1)    # S00[8..15] by 31;                   P001 at 46
    cvts([48, 50, 52, 54], 0, 1, 1)    # S01[8..15] by 47;                   P011 at 47
    cvts([56, 57, 58, 59], 1, 0, 0)    # S10[0..7]  by 55;                   P100 at 54
    cvts([64, 65, 66, 67], 1, 0, 1)    # S10[8..15] by 63;                   P101 at 62 -> gaps 0..3 of the next iteration
    cvts([72, 73, 74, 75], 1, 1, 0)    # S11[0..7]  by 71;                   P110 at 55 -> gaps 8..11
    cvts([74, 76, 78, 80], 1, 1, 1)    # S11[8..15]: pair j by 73 + 2j;      P111 at 63 -> gaps 10, 12, 14, 16 (the S11
                                       # chain restarts at slot 17)
    # ---- LDS reads (the LDS returns data in order: a counted wait names the reads issued after the one needed)
    reads = {s_: [] for s_ in range(64)}
    for ks in range(8):
        reads[2 + 2 * ks].append(("k", ks, k1 + 8192, ks, f"k1_{ks}"))   # K(t+1) key block 1: ring entry ks is free after
                                                                         # slot 2 ks + 1, needed at slot 16 + 2 ks
        reads[18 + 2 * ks].append(("k", ks, k2, ks, f"k2_{ks}"))         # K(t+2) key block 0 (for the NEXT iteration; no
                                                                         # tile t+2: a stale slot is read, unused)
    for f in range(16):
        kb, sk, db = f >> 3, (f >> 2) & 1, f & 3
        g = 17 + 2 * f if f < 8 else 35 + 2 * (f - 8)     # V(t) fragment f -> ring entry f & 7 (free after slot 17 + 2 f)
        reads[g].append(("v", f & 7, v0, kb, sk, db, f"v_{f}"))
    waits = {16: "k1_3", 24: "k1_7", 32: "v_3", 40: "v_7", 48: "v_11", 56: "v_15"}
    # ---- LDS-DMA: this wave's 4 + 4 pieces (1 KiB each) of K(t+3) / V(t+1) in the odd gaps 1..15 (the even ones carry
    # the K reads): buffer_load ... lds with M0 = the piece's LDS address, a buffer descriptor per matrix (DSK / DSV: the
    # key segment's base, so that byte offset = key index x row bytes in both segments), the tile's byte offset as the
    # scalar offset (SOK / SOV) and the lane's offset inside the tile (koff / voff).  For a tile that does not exist, or
    # that ca_attn4.hip stages the general way, the descriptor is the null one (0 records: nothing is read) and LWK / LWV
    # point into the dump page.  hipcc itself never uses M0 in this kernel (build.py checks it), so it is neither
    # saved nor restored.  SGPR hazards (an SGPR written by SALU / v_readlane needs 5 wait states before a VMEM
    # instruction reads it; M0 one state before the LDS-DMA): the M0 write sits in front of the MFMA, and build.py audits
    # the emitted code for writes of the descriptor / offset registers within 5 instructions of a piece.
    dma_of = {1 + 2 * i: (i >> 1, i & 1) for i in range(8)}       # gap -> (piece j, is_v): K0 V0 K1 V1 ...

    def exp_and_add(g):
        kb, qb, i = exps_of[g]
        st.ins(("v_mov_b32 {0}, {0}" if "exp" in KO else "v_exp_f32 {0}, {0}"), (f"{S(kb, qb)}[{i}]", "rw"))
        for akb, aqb, ai, sfx in adds_of[g]:
            if "add" not in KO:
                st.ins("v_add_f32 {0}, {0}, {1}", (f"l{aqb}{sfx}", "rw"), (f"{S(akb, aqb)}[{ai}]", "r"))

    st.comment(f"---- iteration variant r = {r}: K(t+1) slot {(r + 1) % 3}, K(t+2) slot {(r + 2) % 3}, V(t) slot {r}")
    for s in range(64):
        dma = dma_of.get(s) if "dma" not in KO else None
        if dma:
            j, isv = dma
            st.ins(f"s_add_u32 m0, {{0}}, {(vdst if isv else kdst) + 1024 * j}", ("LWV" if isv else "LWK", "rs"))
        if s in waits:
            st.wait_for(waits[s])
        if s < 32:
            kb, ks, qb = s >> 4, (s >> 1) & 7, s & 1
            a = areg(AK + 4 * ks, 4)
            q = areg(AQ + 4 * (qb * 8 + ks), 4)
            # (the K Q^T MFMAs take their operand type from the including kernel: CA_A4_QK_T = "bf16", or "f16" for the
            # variant whose q / k rows were written as IEEE half by the qkv epilogue -- same layout, same rate)
            if ks == 0:
                st.ins(f"v_mfma_f32_32x32x16_{QK_T} {{0}}, {a}, {q}, {{1}}", (S(kb, qb), "w"), (f"NM{qb}", "r"))
            else:
                st.ins(f"v_mfma_f32_32x32x16_{QK_T} {{0}}, {a}, {q}, {{0}}", (S(kb, qb), "rw"))
        else:
            f, qb = (s - 32) >> 1, (s - 32) & 1
            kb, sk, db = f >> 3, (f >> 2) & 1, f & 3
            o = areg(AO + 16 * (qb * 4 + db), 16)
            v = areg(AV + 4 * (f & 7), 4)
            st.ins(f"v_mfma_f32_32x32x16_bf16 {o}, {v}, {{0}}, {o}", (f"P{kb}{qb}{sk}", "r"))
        def mem_ops():
            if dma:
                j, isv = dma
                st.ins("buffer_load_dwordx4 {0}, {1}, {2} offen lds", (f"{'voff' if isv else 'koff'}{j}", "r"),
                       ("DSV" if isv else "DSK", "rs"), ("SOV" if isv else "SOK", "rs"))
            for rd in reads[s]:
                if rd[0] == "k":
                    st.read_k(rd[1], rd[2], rd[3], rd[4])
                else:
                    st.read_v(rd[1], rd[2], rd[3], rd[4], rd[5], rd[6])

        def valu_ops():
            seen = set()
            for (kb, qb, sk, j) in cvt_of[s]:
                assert (kb, qb, sk) not in seen        # one written element per vector variable and statement
                seen.add((kb, qb, sk))
                if "cvt" in KO:
                    continue
                st.ins("v_cvt_pk_bf16_f32 {0}, {1}, {2}", (f"P{kb}{qb}{sk}[{j}]", "w"),
                       (f"{S(kb, qb)}[{8 * sk + 2 * j}]", "r"), (f"{S(kb, qb)}[{8 * sk + 2 * j + 1}]", "r"))
            exp_and_add(s)

        if ORDER == "split":
            mem_ops()
            st.flush()  # statement A: M0, wait, MFMA, DMA, LDS reads.  Statement B: the VALU fillers
            valu_ops()
        elif ORDER == "valu_first":     # one statement per slot: M0, wait, MFMA, the VALU fillers, then DMA / LDS reads
            valu_ops()
            mem_ops()
        else:                           # "mem_first": one statement per slot, memory operations before the VALU fillers
            mem_ops()
            valu_ops()
        st.flush()
    check_schedule(exps_of, adds_of, cvt_of, reads, waits)
    return st.lines


def check_schedule(exps_of, adds_of, cvt_of, reads, waits):
    """The placement rules of the stream, asserted (gap g = the instructions behind MFMA slot g; gaps of the next
    iteration count 64 + g).  Times are in slots; an MFMA issued in slot s has written its result before gap s + 2."""
    done_at = lambda kb, qb: 14 + qb + 16 * kb                      # slot of the chain's last MFMA
    exp_gap = {}
    for g, (kb, qb, i) in exps_of.items():
        first_write = 16 * kb + qb                                   # slot of the chain's first MFMA
        gg = g if g >= done_at(kb, qb) + 2 else g + 64               # this iteration, or lapped into the next one
        assert gg >= done_at(kb, qb) + 2 and gg < 64 + first_write, (kb, qb, i, g)
        exp_gap[(kb, qb, i)] = gg
    for g, lst in adds_of.items():
        for (kb, qb, i, _) in lst:       # after the value's exponential (>= 1 gap), before the chain's next first MFMA
            e = exp_gap[(kb, qb, i)]
            gg = g if g > e else g + 64
            assert e < gg < 64 + 16 * kb + qb, ("row-sum add outside its value's life", kb, qb, i, g)
    for g, lst in cvt_of.items():
        for (kb, qb, sk, j) in lst:
            last_read = 32 + 16 * kb + 8 * sk + 6 + qb               # last P.V MFMA of tile t reading P[kb][qb][sk]
            e = max(exp_gap[(kb, qb, 8 * sk + 2 * j)], exp_gap[(kb, qb, 8 * sk + 2 * j + 1)])
            gg = g if (g > last_read and g > e) else g + 64
            assert gg > last_read and gg > e, ("pack before its exponentials / its last reader", kb, qb, sk, j, g)
            assert gg < 64 + 16 * kb + qb, ("pack after the chain restarted", kb, qb, sk, j)
            assert gg < 64 + 32 + 16 * kb + 8 * sk, ("pack after its first reader of the next tile", kb, qb, sk, j)
    for g, lst in reads.items():
        for rd in lst:
            if rd[0] == "k":
                ks, name = rd[1], rd[4]
                if name.startswith("k1_"):       # overwrites K(t+1) kb 0 fragment ks (last read slot 2ks+1), used at 16+2ks
                    assert g > 2 * ks + 1 and g + 6 <= 16 + 2 * ks, name
                else:                            # overwrites kb 1 fragment ks (last read slot 17+2ks), used at 64+2ks
                    assert g > 17 + 2 * ks, name
            else:
                f = int(rd[6].split("_")[1])
                assert g + 6 <= 32 + 2 * f, rd[6]
                if f >= 8:
                    assert g > 33 + 2 * (f - 8), rd[6]   # ring entry f & 7 held fragment f - 8


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
    L.append("#define CA_A4_SCALE_O(al0, al1) do { float t_; asm volatile(\"s_nop 15\\n\\ts_nop 7\" : : \"v\"(al0), \"v\"(al1)); \\")
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
                        L.append(f'  asm volatile("v_mfma_f32_32x32x16_{QK_T} %0, {a}, {q}, 0" : "=v"({S(kb, qb)})); \\')
                    elif ks == 0:
                        L.append(f'  asm volatile("v_mfma_f32_32x32x16_{QK_T} %0, {a}, {q}, %1" : "=&v"({S(kb, qb)}) : "v"(NM{qb})); \\')
                    else:
                        L.append(f'  asm volatile("v_mfma_f32_32x32x16_{QK_T} %0, {a}, {q}, %0" : "+v"({S(kb, qb)})); \\')
            if kb == 0:   # the ring is re-filled for key block 1: the MFMAs above must have read their operands
                L.append('  asm volatile("s_nop 7"); \\')
        L.append('  asm volatile("s_nop 15\\n\\ts_nop 7" : "+v"(S00), "+v"(S01), "+v"(S10), "+v"(S11)); } while (0)')
        L.append("")
    # ---- plain O^T += V^T P^T of one tile: vbase = V slot byte offset (runtime).  The first 8 V fragments are read up
    # front, fragment f + 7 into ring entry f - 1 behind the MFMAs of fragment f (one fragment = 2 MFMAs after the entry's
    # last reader was issued); counted waits (the LDS returns data in order).
    # (its first statement takes every P fragment through an s_nop: hipcc pads no hazard INTO an asm statement, and a
    # pack -- VALU write -- right in front of the MFMA that reads it as B needs 2 wait states; without the operands
    # hipcc is free to sink the packs below a bare s_nop, and query block 0's first MFMA then read stale registers)
    allp = ", ".join(f'"+v"(P{kb}{qb}{sk})' for kb in range(2) for qb in range(2) for sk in range(2))
    L.append(f"#define CA_A4_PV_PLAIN(vbase) do {{ asm volatile(\"s_nop 7\" : {allp}); \\")
    def v_read(f):
        kb, sk, db = f >> 3, (f >> 2) & 1, f & 3
        row = (32 * kb + 16 * sk) * 256
        ring = f & 7
        L.append(f'  asm volatile("ds_read_b64_tr_b16 {areg(AV + 4 * ring, 2)}, %0 offset:{row}" : : "v"(va0{db} + (vbase)) : "memory"); \\')
        L.append(f'  asm volatile("ds_read_b64_tr_b16 {areg(AV + 4 * ring + 2, 2)}, %0 offset:{row + 2048}" : : "v"(va1{db} + (vbase)) : "memory"); \\')
    issued = []                       # fragments in issue order; a wait for fragment f allows 2 x (reads issued behind it)
    for f in range(8):
        v_read(f)
        issued.append(f)
    for f in range(16):
        kb, sk, db = f >> 3, (f >> 2) & 1, f & 3
        ring = f & 7
        L.append(f'  asm volatile("s_waitcnt lgkmcnt({2 * (len(issued) - 1 - issued.index(f))})" ::: "memory"); \\')
        for qb in range(2):
            o = areg(AO + 16 * (qb * 4 + db), 16)
            L.append(f'  asm volatile("v_mfma_f32_32x32x16_bf16 {o}, {areg(AV + 4 * ring, 4)}, %0, {o}" : : "v"(P{kb}{qb}{sk})); \\')
        if 1 <= f <= 8:               # ring entry f - 1 is free: its readers were issued one fragment (2 MFMAs) ago
            v_read(f + 7)
            issued.append(f + 7)
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
