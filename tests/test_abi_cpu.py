"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/conceptattn.h declares, and rejects bad arguments without a GPU."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as entry
from conceptattention_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    entry.build()
    return L.load()


def test_header_symbols_all_exported(lib):
    text = open(os.path.join(ROOT, "include", "conceptattn.h")).read()
    declared = set(re.findall(r"\b(ca_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations parsed"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_error_string(lib):
    assert lib.ca_version() == L.CA_VERSION
    assert isinstance(lib.ca_last_error(), bytes)


def test_struct_layouts_match_header():
    # sizes the C compiler sees (LP64): pointers 8, int32 4, no implicit padding inside
    assert ctypes.sizeof(L.GemmProblem) == 14 * 8 + 20 * 4   # 16 int32 + qpre_f32 + float q_out_scale + qk_f16 + pad
    assert ctypes.sizeof(L.AttnProblem) == 11 * 8 + 11 * 4 + 4  # (hm_con / hm_part since round 5) trailing pad to 8 bytes
    assert ctypes.sizeof(L.ModSegment) == 24 and ctypes.sizeof(L.NormSegment) == 24
    assert ctypes.sizeof(L.HeatmapProblem) == 5 * 8 + 4 * 4 + 2 * 4   # ca_heatmap_problem: 64 bytes


def _header_struct_fields(name):
    """(type, field) pairs of a `typedef struct { ... } name;` in include/conceptattn.h, in declaration order."""
    text = open(os.path.join(ROOT, "include", "conceptattn.h")).read()
    end = re.search(r"\}\s*" + name + r"\s*;", text).start()
    body = text[text.rfind("typedef struct {", 0, end) + len("typedef struct {"):end]   # (the LAST opener before the name)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(?:const\s+)?(void|float|int32_t)\s*(\*?)\s*(.*)$", decl, re.S)
        assert m, decl
        ctype = "ptr" if m.group(2) or m.group(3).lstrip().startswith("*") else m.group(1)
        for f in m.group(3).split(","):
            out.append((ctype, f.strip().lstrip("*").strip()))
    return out


_CT = {"ptr": ctypes.c_void_p, "int32_t": ctypes.c_int32, "float": ctypes.c_float}


def test_binding_fields_match_header_field_for_field():
    """Names, order and C types of ca_gemm_problem in the header == conceptattention_amd/_lib.py's ctypes struct."""
    hdr = _header_struct_fields("ca_gemm_problem")
    assert [f for _, f in hdr] == [f for f, _ in L.GemmProblem._fields_]
    assert [_CT[t] for t, _ in hdr] == [t for _, t in L.GemmProblem._fields_]


def test_attn_problem_fields_match_header():
    hdr = _header_struct_fields("ca_attn_problem")
    assert [f for _, f in hdr] == [f for f, _ in L.AttnProblem._fields_]
    assert [_CT[t] for t, _ in hdr] == [t for _, t in L.AttnProblem._fields_]


def test_heatmap_problem_fields_match_header_and_arguments_are_checked_without_a_gpu(lib):
    """ca_heatmap_problem (round 5): header fields == ctypes fields; ca_heatmap_fused rejects bad arguments before any
    launch (no GPU needed for that)."""
    hdr = _header_struct_fields("ca_heatmap_problem")
    assert [f for _, f in hdr] == [f for f, _ in L.HeatmapProblem._fields_]
    assert [_CT[t] for t, _ in hdr] == [t for _, t in L.HeatmapProblem._fields_]
    arr = (L.HeatmapProblem * 1)()
    for bad in (dict(n=0), dict(C=9), dict(dim=12), dict(norm=7), dict()):   # the last: null pointers in the problem
        rc = lib.ca_heatmap_fused(arr, bad.get("n", 1), 64, bad.get("C", 4), bad.get("dim", 256), bad.get("norm", 0), None)
        assert rc == -1, bad
        assert b"ca_heatmap_fused" in lib.ca_last_error()


def test_integration_md_stub_is_the_current_abi():
    """INTEGRATION.md section B shows the ctypes stub a maintainer copies; a stale copy hands the library a struct
    it reads past (round 3's document was two fields short).  Parse the struct out of the document and compare names,
    order, types and sizeof with the binding and the header."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"class GemmProblem\(C\.Structure\):.*?_fields_ = \[(.*?)\]\s*(?:#[^\n]*)?\n\n", doc, re.S).group(1)
    fields = re.findall(r'\("(\w+)",\s*C\.(c_\w+)\)', block)
    assert fields, "no fields parsed from INTEGRATION.md"
    assert [(f, getattr(ctypes, t)) for f, t in fields] == list(L.GemmProblem._fields_)
    Doc = type("Doc", (ctypes.Structure,), {"_fields_": [(f, getattr(ctypes, t)) for f, t in fields]})
    assert ctypes.sizeof(Doc) == ctypes.sizeof(L.GemmProblem)
    assert [f for f, _ in fields] == [f for _, f in _header_struct_fields("ca_gemm_problem")]
    m = re.search(r"ABI version (\d+)", doc)
    assert m and int(m.group(1)) == L.CA_VERSION


def test_argument_rejection_needs_no_gpu(lib):
    assert lib.ca_gemm_bf16(None, 1, 0, None) == -1
    assert b"n_problems" in lib.ca_last_error() or b"ca_gemm" in lib.ca_last_error()
    p = (L.GemmProblem * 1)()
    p[0].M, p[0].N, p[0].K = 16, 256, 100  # K % 64 != 0, null pointers
    assert lib.ca_gemm_bf16(p, 1, L.TILE_256x256, None) == -1
    a = (L.AttnProblem * 1)()
    assert lib.ca_attn_fwd_bf16(a, 1, 24, 0.1, None) == -1
    assert lib.ca_attn_fwd_bf16(a, 3, 24, 0.1, None) == -1
    assert lib.ca_axpy_bf16(None, None, 1.0, 0, None) == -1
    assert lib.ca_gemv_bf16(None, 9, 0, None, None, None, 0, 0, 0, 0, 0, None) == -1


def test_auto_tile_picks_the_256x256_ping_pong_tile_for_the_models_launches(lib):
    """ca_gemm_auto_tile needs no GPU.  Round 4's trace of the one-item forward found proj and mlp.2 on 256x128 tiles
    (two rounds of half-width tiles, 130 / 416 us against 62 / 222 us): the chooser had priced the thin-row launch of the
    4 concept rows at 0.3 round although those rows fit into the CUs the 204 main tiles leave idle.  Pin the choice for
    the grouped launches of the model at 1 and 5 items per forward."""
    H = 3072

    def tile(probs):
        arr = (L.GemmProblem * len(probs))()
        for i, (M, N, K, epi, n_split) in enumerate(probs):
            arr[i].M, arr[i].N, arr[i].K, arr[i].epilogue, arr[i].n_split = M, N, K, epi, n_split
        return lib.ca_gemm_auto_tile(arr, len(probs))
    for B in (1, 5):
        img, ctx, rows = B * 4096, B * 260, B * 4352
        launches = {"qkv": [(img, 3 * H, H, L.EPI_QKV_NORM_ROPE, 3 * H), (ctx, 3 * H, H, L.EPI_QKV_NORM_ROPE, 3 * H)],
                    "proj": [(img, H, H, L.EPI_GATE_RESIDUAL, 0), (ctx, H, H, L.EPI_GATE_RESIDUAL, 0)],
                    "mlp.0": [(img, 4 * H, H, L.EPI_GELU_TANH, 0), (ctx, 4 * H, H, L.EPI_GELU_TANH, 0)],
                    "mlp.2": [(img, H, 4 * H, L.EPI_GATE_RESIDUAL, 0), (ctx, H, 4 * H, L.EPI_GATE_RESIDUAL, 0)],
                    "linear1": [(rows, 7 * H, H, L.EPI_QKV_NORM_ROPE, 3 * H)],
                    "linear2": [(rows, H, 5 * H, L.EPI_GATE_RESIDUAL, 0)]}
        for name, probs in launches.items():
            assert tile(probs) == L.TILE_PP_256x256, (B, name, tile(probs))
    # a narrow problem still gets a narrower tile (N = 128 does not divide by 256)
    assert tile([(4096, 128, 3072, L.EPI_BIAS, 0)]) in (L.TILE_PP_256x128, L.TILE_256x64)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.ConceptAttnError):
        L.load()
