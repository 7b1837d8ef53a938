"""The generated instruction stream of ca_attn4_kernel and the audits that guard it (no GPU needed).

* the committed conceptattention_amd/csrc/ca_attn4_sched.inc is what tools/gen_attn4_schedule.py generates (the generator
  asserts its own placement rules while it runs: every exponential / row-sum add / pack inside the life of its score
  value, every LDS read early enough and behind the last reader of the ring entry it overwrites);
* csrc/build.py's audits flag the hazards hipcc does not pad inside asm statements -- each of them was a real bug of
  round 3: a VALU-written SGPR read by an LDS-DMA piece too early, M0 written right in front of
  its LDS-DMA, a VALU-written VGPR read by an MFMA within 2 wait states.
"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "conceptattention_amd", "csrc", "ca_attn4_sched.inc")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_schedule_is_the_generated_one(tmp_path):
    out = tmp_path / "sched.inc"
    env = dict(os.environ, CA_A4_OUT=str(out))
    env.pop("CA_A4_KO", None)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn4_schedule.py")], check=True, env=env,
                   capture_output=True)
    assert out.read_text() == open(INC).read(), "regenerate with: python tools/gen_attn4_schedule.py"


def test_stream_shape():
    """Per iteration variant: 64 MFMAs, 64 exponentials, 64 row-sum adds, 32 packs, 16 + 32 LDS fragment reads, 8 LDS-DMA
    pieces with their 8 M0 writes, and only counted waits."""
    text = open(INC).read()
    sched = text[text.index("#ifdef CA_A4_SCHEDULE"):]
    variants = sched.split("---- iteration variant")[1:]
    assert len(variants) == 3
    for v in variants:
        assert v.count("v_mfma_f32_32x32x16_") == 64
        assert v.count('v_mfma_f32_32x32x16_" CA_A4_QK_T "') == 32 and v.count("v_mfma_f32_32x32x16_bf16") == 32
        assert v.count("v_exp_f32") == 64 and v.count("v_add_f32") == 64 and v.count("v_cvt_pk_bf16_f32") == 32
        assert v.count("ds_read_b128") == 16 and v.count("ds_read_b64_tr_b16") == 32
        assert v.count("buffer_load_dwordx4") == 8 and v.count("s_add_u32 m0") == 8
        assert "s_nop" not in v and "lgkmcnt(0)" in v and v.count("s_waitcnt") == 6


def test_generator_rejects_a_misplaced_add():
    """The rule that was violated in round 3: a row-sum add behind the slot in which its chain restarts."""
    gen = _load(os.path.join(ROOT, "tools", "gen_attn4_schedule.py"), "gen_a4")
    order = [(kb, qb, i) for kb, qb in ((0, 0), (0, 1), (1, 0), (1, 1)) for i in range(16)]
    exps_of = {(16 + n) % 64: e for n, e in enumerate(order)}
    adds_of = {g: [] for g in range(64)}
    for n, e in enumerate(order):
        adds_of[(16 + n) % 64].append(order[(n - 2) % 64] + ("",))      # S11[15] lands in gap 17: one gap too late
    empty = {g: [] for g in range(64)}
    try:
        gen.check_schedule(exps_of, adds_of, empty, empty, {})
    except AssertionError as e:
        assert "row-sum add" in str(e.args[0])
    else:
        raise AssertionError("the misplaced add was accepted")


def test_build_audits_flag_the_three_hazards():
    build = _load(os.path.join(ROOT, "conceptattention_amd", "csrc", "build.py"), "ca_build")
    ok = """\t;;#ASMSTART
\ts_add_u32 m0, s1, 0
\tv_mfma_f32_32x32x16_bf16 v[0:15], a[0:3], a[4:7], v[0:15]
\tbuffer_load_dwordx4 v133, s[20:23], s31 offen lds
\t;;#ASMEND
"""
    assert build.audit_sgpr_hazards(ok) == []
    sgpr = "\tv_readlane_b32 s21, v180, 4\n\ts_add_u32 s5, s5, 1\n" + ok
    assert any("v_readlane_b32 s21" in b for b in build.audit_sgpr_hazards(sgpr))
    m0 = "\t;;#ASMSTART\n\ts_mov_b32 m0, s1\n\tbuffer_load_dwordx4 v133, s[40:43], s31 offen lds\n\t;;#ASMEND\n"
    assert any("M0" in b for b in build.audit_sgpr_hazards(m0))
    valu = ("\tv_cvt_pk_bf16_f32 v5, v21, v83\n\t;;#ASMSTART\n"
            "\tv_mfma_f32_32x32x16_bf16 a[0:15], a[224:227], v[2:5], a[0:15]\n\t;;#ASMEND\n")
    assert any("v_cvt_pk_bf16_f32 v5" in b for b in build.audit_sgpr_hazards(valu))
    far = valu.replace("\t;;#ASMSTART", "\ts_nop 1\n\t;;#ASMSTART")
    assert build.audit_sgpr_hazards(far) == []


def test_build_audit_flags_an_mfma_result_touched_early():
    """The opposite hazard (round-3 advisory): the RESULT of an asm MFMA read by a VALU / LDS / VMEM instruction, or by
    another MFMA as A / B, before the matrix pipe has written it back (8 passes: 11 wait states).  hipcc pads nothing
    around an MFMA it cannot see, inside or outside the asm statements."""
    build = _load(os.path.join(ROOT, "conceptattention_amd", "csrc", "build.py"), "ca_build")
    mfma = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 v[34:49], a[192:195], a[128:131], v[2:17]\n\t;;#ASMEND\n"
    # a compiler-placed copy right behind the statement (what commit 863ff6e's fault looked like from the other side)
    early = mfma + "\tv_mov_b32_e32 v200, v40\n"
    assert any("v_mov_b32_e32 v200, v40" in b for b in build.audit_mfma_result_hazards(early))
    # the same copy two MFMA slots later is what the generated stream relies on: clean
    other = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 v[50:65], a[192:195], a[160:163], v[18:33]\n\t;;#ASMEND\n"
    other2 = other.replace("v[18:33]", "v[50:65]")     # the next MFMA of that chain (accumulates: interlocked)
    assert build.audit_mfma_result_hazards(mfma + other + other2 + "\tv_mov_b32_e32 v200, v40\n") == []
    # ... and one slot later is not (8 passes of the next MFMA + 1 < 11 + 1)
    assert build.audit_mfma_result_hazards(mfma + other + "\tv_exp_f32 v40, v40\n") != []
    # enough s_nop in between: clean; one state short: flagged
    assert build.audit_mfma_result_hazards(mfma + "\ts_nop 10\n\tv_mov_b32_e32 v200, v40\n") == []
    assert build.audit_mfma_result_hazards(mfma + "\ts_nop 9\n\tv_mov_b32_e32 v200, v40\n") != []
    # an LDS write, a store and an MFMA taking the result as B are readers too; accumulating onto it is interlocked
    assert build.audit_mfma_result_hazards(mfma + "\tds_write_b32 v1, v35\n") != []
    assert build.audit_mfma_result_hazards(mfma + "\tglobal_store_dword v[0:1], v49, off\n") != []
    asb = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 a[0:15], a[224:227], v[34:37], a[0:15]\n\t;;#ASMEND\n"
    assert build.audit_mfma_result_hazards(mfma + asb) != []
    acc = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 v[34:49], a[196:199], a[132:135], v[34:49]\n\t;;#ASMEND\n"
    assert build.audit_mfma_result_hazards(mfma + acc) == []
    # the accumulator file: O^T read back too early (CA_A4_SCALE_O / READ_O start with 24 wait states for this)
    pv = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 a[0:15], a[224:227], v[2:5], a[0:15]\n\t;;#ASMEND\n"
    assert build.audit_mfma_result_hazards(pv + "\t;;#ASMSTART\n\tv_accvgpr_read_b32 v9, a3\n\t;;#ASMEND\n") != []
    assert build.audit_mfma_result_hazards(pv + "\t;;#ASMSTART\n\ts_nop 15\n\tv_accvgpr_read_b32 v9, a3\n\t;;#ASMEND\n") == []


def test_build_audit_carries_the_history_across_branches():
    """The state at a branch travels to its target: the tile loop's R = 2 -> R = 0 back-edge, and a forward branch that
    skips the padding which makes the fall-through path legal."""
    build = _load(os.path.join(ROOT, "conceptattention_amd", "csrc", "build.py"), "ca_build")
    mfma = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 v[34:49], a[192:195], a[128:131], v[2:17]\n\t;;#ASMEND\n"
    loop = ".LBB0_1:\n\tv_add_f32_e32 v40, v40, v41\n\ts_nop 15\n" + mfma + "\ts_cbranch_scc1 .LBB0_1\n"
    assert any("v_add_f32_e32 v40" in b for b in build.audit_mfma_result_hazards(loop))
    fwd = mfma + "\ts_cbranch_vccnz .LBB0_9\n\ts_nop 15\n.LBB0_9:\n\tv_add_f32_e32 v40, v40, v41\n"
    assert any("v_add_f32_e32 v40" in b for b in build.audit_mfma_result_hazards(fwd))
    ok = ".LBB0_1:\n\ts_nop 15\n\tv_add_f32_e32 v40, v40, v41\n" + mfma + "\ts_cbranch_scc1 .LBB0_1\n"
    assert build.audit_mfma_result_hazards(ok) == []


def test_build_audit_allows_only_scalar_code_inside_the_generated_stream():
    build = _load(os.path.join(ROOT, "conceptattention_amd", "csrc", "build.py"), "ca_build")
    a = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 v[34:49], a[192:195], a[128:131], v[2:17] ; a4s\n\t;;#ASMEND\n"
    b = "\t;;#ASMSTART\n\tv_exp_f32 v66, v66 ; a4s\n\tv_add_f32 v143, v143, v96\n\t;;#ASMEND\n"
    a2 = a.replace("v[2:17]", "v[34:49]")              # the chain's next MFMA
    assert build.audit_mfma_result_hazards(a + "\ts_nop 0\n" + b + "\ts_add_i32 s0, s18, 0x400\n" + a2) == []
    moved = a + "\tv_mov_b32_e32 v201, v96\n" + b
    assert any("inside the generated stream" in x for x in build.audit_mfma_result_hazards(moved))
    # the same instruction between the stream and an unmarked helper statement is hipcc's business
    helper = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\t;;#ASMEND\n"
    assert build.audit_mfma_result_hazards(b + helper + "\tv_mov_b32_e32 v201, v96\n" + b) == []


def test_shipped_stream_is_marked_for_the_audit():
    text = open(INC).read()
    sched = text[text.index("#ifdef CA_A4_SCHEDULE"):]
    assert sched.count("; a4s") == 3 * 128   # two statements per MFMA slot, three iteration variants


def test_build_audit_classifies_every_hot_asm_statement():
    """Outside the generated stream an asm statement with an MFMA / LDS / memory instruction must be one of the known
    helper shapes; a new one fails the build until it has been looked at (housekeeping)."""
    build = _load(os.path.join(ROOT, "conceptattention_amd", "csrc", "build.py"), "ca_build")
    known = ("\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 v[0:15], a[0:3], a[4:7], 0\n\t;;#ASMEND\n"
             "\t;;#ASMSTART\n\ts_mov_b32 m0, s3\n\ts_nop 4\n\tbuffer_load_dwordx4 v1, s[4:7], s9 offen lds\n\t;;#ASMEND\n"
             "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\t;;#ASMEND\n")
    assert build.audit_asm_statements(known) == []
    marked = "\t;;#ASMSTART\n\tv_exp_f32 v1, v1 ; a4s\n\tds_write_b32 v2, v1\n\t;;#ASMEND\n"
    assert build.audit_asm_statements(marked) == []          # the generated stream has its own rules
    new = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 v[0:15], a[0:3], a[4:7], 0\n\tds_read_b128 a[0:3], v9\n\t;;#ASMEND\n"
    assert any("unclassified" in b for b in build.audit_asm_statements(new))
