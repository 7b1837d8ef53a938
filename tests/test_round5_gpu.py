"""Round 5: the fused heat-map launch (ca_heatmap_fused) against the three-launch form it replaces, bit for bit, and
against an fp32 reference of concept_attention_pipeline.py:57-82."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from conceptattention_amd import _lib as L  # noqa: E402
from conceptattention_amd import ops  # noqa: E402

DEV = "cuda"


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("C,Lp,dim", [(1, 257, 256), (2, 4096, 3072), (3, 100, 512), (4, 4096, 3072), (5, 333, 1024),
                                      (8, 1025, 3072), (4, 1, 3072), (8, 2, 128)])
@pytest.mark.parametrize("norm", [L.NORM_SOFTMAX, L.NORM_SPARSEMAX, L.NORM_ENTMAX15])
def test_heatmap_fused_is_bit_identical_to_the_three_launch_form(C, Lp, dim, norm):
    """Two accumulators with different weights, fp32 and bf16 image vectors, fp32 and bf16 concept vectors, several
    problems per launch, ragged L (odd patch counts, one patch), C not a multiple of 4; logits scaled so that the
    weighting is not near-uniform."""
    probs, refs = [], []
    for i, (img_dt, con_dt) in enumerate([(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                          (torch.bfloat16, torch.bfloat16)]):
        img = _rand((Lp, dim), 10 + i, 0.2).to(img_dt)
        con = _rand((C, dim), 20 + i, 0.4).to(con_dt)
        acc, acc2 = _rand((C, Lp), 30 + i), _rand((C, Lp), 40 + i)
        r_acc, r_acc2, lg = acc.clone(), acc2.clone(), torch.empty(C, Lp, device=DEV)
        ops.heatmap_logits(img, con, lg)
        ops.heatmap_softmax_accumulate(lg, r_acc, 0.25, norm)
        ops.heatmap_softmax_accumulate(lg, r_acc2, 1.0 / 3.0, norm)
        lg_f = torch.zeros(C, Lp, device=DEV)
        probs.append(ops.Heatmap(img, con, acc, 0.25, acc2, 1.0 / 3.0, lg_f))
        refs.append((r_acc, r_acc2, lg))
    ops.heatmap_fused(probs, norm)
    torch.cuda.synchronize()
    for h, (r_acc, r_acc2, lg) in zip(probs, refs):
        assert torch.equal(h.logits, lg)
        assert torch.equal(h.acc, r_acc)
        assert torch.equal(h.acc2, r_acc2)
    # one accumulator only, either slot
    a = torch.zeros(C, Lp, device=DEV)
    b = torch.zeros(C, Lp, device=DEV)
    ops.heatmap_fused([ops.Heatmap(probs[0].img_vec, probs[0].con_vec, a, 0.5),
                       ops.Heatmap(probs[0].img_vec, probs[0].con_vec, None, 0.0, b, 0.5)], norm)
    assert torch.equal(a, b)
    # against an fp32 einsum + softmax (concept_attention_pipeline.py:57-65); fp32 accumulation order differs: 1e-5
    if norm == L.NORM_SOFTMAX:
        ref = torch.softmax(torch.einsum("pd,cd->cp", probs[0].img_vec.double(), probs[0].con_vec.double()), 0) * 0.5
        assert (a.double() - ref).abs().max().item() < 1e-5


def test_heatmap_fused_rejects_bad_arguments():
    img, con = _rand((64, 256), 1), _rand((9, 256), 2)
    acc = torch.zeros(9, 64, device=DEV)
    assert not ops.heatmap_fused_fits(9, 256)
    with pytest.raises(ValueError):
        ops.heatmap_fused([ops.Heatmap(img, con, acc, 1.0)])            # C > 8: the caller uses the three-launch form
    con4 = con[:4].contiguous()
    with pytest.raises(ValueError):
        ops.heatmap_fused([ops.Heatmap(img, con4)])                     # no accumulator at all
    with pytest.raises(ValueError):
        ops.heatmap_fused([ops.Heatmap(img, con4, acc[:4].t().contiguous().t(), 1.0)])   # not contiguous [C,L]
    with pytest.raises(ValueError):
        ops.heatmap_fused([ops.Heatmap(img, con4, torch.zeros(4, 64, device=DEV), 1.0)] * 17)


def _tiny_forward_kwargs(B):
    from conceptattention_amd.flux_dit import HipFluxDiT
    from conceptattention_amd.params import tiny_params
    from conceptattention_amd.weights import synthetic_inputs, synthetic_state_dict
    from conceptattention_amd import sampling
    p = tiny_params()
    m = HipFluxDiT(p, DEV)
    m.load_state_dict({k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=1).items()})
    items = [synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=3, seed=2 + j, dtype=torch.bfloat16) for j in range(B)]
    cat = lambda k: torch.cat([it[k] for it in items], 0).to(DEV)   # noqa: E731
    kw = dict(img=sampling.patchify(cat("latent")), img_ids=cat("img_ids"), txt=cat("txt"), txt_ids=cat("txt_ids"),
              concepts=cat("concepts"), concept_ids=cat("concept_ids"), concept_vec=cat("concept_vec"), y=cat("vec"),
              timesteps=torch.full((B,), 0.6, device=DEV), guidance=torch.zeros(B, device=DEV),
              stop_after_multimodal_attentions=True, return_vectors=False)
    return m, p, kw


def test_model_fused_heatmaps_equal_the_three_launch_form_and_shared_accumulators_stay_ordered():
    """The model's capture path (HipFluxDiT._capture) with fused_heatmaps on / off: bit-identical accumulators and
    per-layer tables for a 3-item forward; and two requests that SHARE their accumulators (a caller summing over items)
    are split into separate launches -- within one launch the updates of two problems to one tensor would race."""
    from conceptattention_amd.flux_dit import HeatmapRequest
    B, C, Lp = 3, 3, 256
    m, p, kw = _tiny_forward_kwargs(B)
    m.epilogue_logits = False   # (the fp32-rows route on both sides: the accumulator route sums in another order)
    res = {}
    for fused in (True, False):
        m.fused_heatmaps = fused
        acc = torch.zeros(B, 2, C, Lp, device=DEV)
        tab = torch.zeros(B, 2, p.depth, C, Lp, device=DEV)
        reqs = [HeatmapRequest(tuple(range(p.depth)), 0.5, acc[j, 0], acc[j, 1], per_layer_out=tab[j, 0],
                               per_layer_cross=tab[j, 1], per_layer_weight=0.25) for j in range(B)]
        m(heatmaps=reqs, **kw)
        torch.cuda.synchronize()
        res[fused] = (acc, tab)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert float(res[True][0].abs().sum()) > 0
    # shared accumulators: every item adds into the SAME two tensors; fused == three-launch, run after run
    sums = []
    for fused in (True, True, False):
        m.fused_heatmaps = fused
        a, b = torch.zeros(C, Lp, device=DEV), torch.zeros(C, Lp, device=DEV)
        m(heatmaps=[HeatmapRequest((1,), 1.0, a, b) for _ in range(B)], **kw)
        torch.cuda.synchronize()
        sums.append((a, b))
    assert all(torch.equal(sums[0][i], s[i]) for s in sums[1:] for i in (0, 1))
    assert (sums[0][0].sum(0) - B).abs().max().item() < 1e-4      # each item contributes one softmax over the concepts
    m.fused_heatmaps = True


def test_more_than_eight_concepts_take_the_three_launch_form():
    """ca_heatmap_fused holds all C concept vectors of a problem in LDS (C <= 8); a forward with more concepts must fall
    back to the logits + weighting launches by itself and still produce normalised maps equal to the stacked route's."""
    from conceptattention_amd.flux_dit import HeatmapRequest, HipFluxDiT
    from conceptattention_amd.heatmaps import compute_heatmaps_from_vectors
    from conceptattention_amd.params import tiny_params
    from conceptattention_amd.weights import synthetic_inputs, synthetic_state_dict
    from conceptattention_amd import sampling
    p = tiny_params()
    C, Lp = 10, 256
    assert not ops.heatmap_fused_fits(C, p.hidden_size)
    m = HipFluxDiT(p, DEV)
    m.load_state_dict({k: v.bfloat16().float() for k, v in synthetic_state_dict(p, seed=1).items()})
    it = synthetic_inputs(p, 256, 256, n_txt=8, n_concepts=C, seed=4, dtype=torch.bfloat16)
    d = {k: v.to(DEV) for k, v in it.items()}
    kw = dict(img=sampling.patchify(d["latent"]), img_ids=d["img_ids"], txt=d["txt"], txt_ids=d["txt_ids"],
              concepts=d["concepts"], concept_ids=d["concept_ids"], concept_vec=d["concept_vec"], y=d["vec"],
              timesteps=torch.full((1,), 0.6, device=DEV), guidance=torch.zeros(1, device=DEV),
              stop_after_multimodal_attentions=True)
    req = HeatmapRequest((0, 1), 0.5, torch.zeros(C, Lp, device=DEV), torch.zeros(C, Lp, device=DEV))
    m(heatmaps=req, return_vectors=False, **kw)
    _, dd = m(return_vectors=True, **kw)
    ho = compute_heatmaps_from_vectors(dd["output_space_image_vectors"][None], dd["output_space_concept_vectors"][None],
                                       [0, 1], [0])
    assert (req.out_space.sum(0) - 1).abs().max().item() < 1e-5
    # (the stacked dict carries bf16 vectors, the fused route fp32 concept rows: 2e-3, as tests/test_pipeline_gpu.py)
    assert (req.out_space.view(1, C, 16, 16).cpu() - ho.float().cpu()).abs().max().item() < 2e-3


def test_heatmap_fused_shape_fuzz():
    """40 random launches (C 1..8, L 1..700, dim a multiple of 8 up to 1 280 -- below and above one 512-element pass of a
    lane --, 1..6 problems of mixed vector types, every norm, one or two accumulators) against the three-launch form,
    bit for bit."""
    import random
    rng = random.Random(5)
    for case in range(40):
        C, Lp, dim = rng.randint(1, 8), rng.randint(1, 700), 8 * rng.randint(1, 160)
        norm = rng.choice([L.NORM_SOFTMAX, L.NORM_SPARSEMAX, L.NORM_ENTMAX15])
        probs, refs = [], []
        for i in range(rng.randint(1, 6)):
            img_dt, con_dt = rng.choice([(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                         (torch.bfloat16, torch.bfloat16)])
            img = _rand((Lp, dim), 1000 * case + i, 0.3).to(img_dt)
            con = _rand((C, dim), 2000 * case + i, 0.3).to(con_dt)
            two = rng.random() < 0.5
            acc, acc2 = _rand((C, Lp), 3000 * case + i), (_rand((C, Lp), 4000 * case + i) if two else None)
            w, w2 = rng.random(), rng.random()
            r_acc, r_acc2 = acc.clone(), (acc2.clone() if two else None)
            lg = torch.empty(C, Lp, device=DEV)
            ops.heatmap_logits(img, con, lg)
            ops.heatmap_softmax_accumulate(lg, r_acc, w, norm)
            if two:
                ops.heatmap_softmax_accumulate(lg, r_acc2, w2, norm)
            probs.append(ops.Heatmap(img, con, acc, w, acc2, w2))
            refs.append((r_acc, r_acc2))
        ops.heatmap_fused(probs, norm)
        torch.cuda.synchronize()
        for h, (r_acc, r_acc2) in zip(probs, refs):
            assert torch.equal(h.acc, r_acc), (case, C, Lp, dim, norm)
            assert r_acc2 is None or torch.equal(h.acc2, r_acc2), (case, C, Lp, dim, norm)


@pytest.mark.parametrize("qk_f16", [False, True])
@pytest.mark.parametrize("T,Limg,C,NH", [(64, 300, 4, 3), (8, 256, 1, 2), (256, 700, 8, 24)])
def test_attention_epilogue_partial_logits(T, Limg, C, NH, qk_f16):
    """ca_attn_problem.hm_con / hm_part: per head the dot products of the second query segment's rows (the fp32
    accumulators, before their bf16 rounding) with the C concept rows, against an fp32 attention of the same bf16 inputs;
    summed over the heads by ca_heatmap_fused (img_f32 = 2) they are the logits the vector form computes from fp32 rows.
    Rows of the first segment and rows past the problem leave nothing; the bf16 outputs are untouched by the feature."""
    H = NH * 128
    n = T + Limg
    g = torch.Generator(device="cpu").manual_seed(T + Limg)
    q = (torch.randn(n, H, generator=g) * 0.35).to(DEV)
    k = torch.randn(n, H, generator=g).to(DEV)
    v = torch.randn(n, H, generator=g).to(DEV)
    con = (torch.randn(C, H, generator=g) * 0.5).to(DEV)
    cast = (lambda t: t.half().view(torch.bfloat16)) if qk_f16 else (lambda t: t.bfloat16())
    back = (lambda t: t.view(torch.float16).float()) if qk_f16 else (lambda t: t.float())
    qb, kb, vb = cast(q), cast(k), v.bfloat16()
    out = torch.zeros(n, H, device=DEV, dtype=torch.bfloat16)
    out_plain = torch.zeros_like(out)
    part = torch.full((NH, Limg, 8), -7.0, device=DEV)
    ops.attention([ops.Attn(qb[:T], out[:T], kb[:T], vb[:T], kb[T:], vb[T:], q1=qb[T:], out1=out[T:], hm_con=con,
                            hm_part=part)], NH, q_prescaled=True, qk_f16=qk_f16)
    ops.attention([ops.Attn(qb[:T], out_plain[:T], kb[:T], vb[:T], kb[T:], vb[T:], q1=qb[T:], out1=out_plain[T:])], NH,
                  q_prescaled=True, qk_f16=qk_f16)
    torch.cuda.synchronize()
    assert torch.equal(out, out_plain)
    qf, kf, vf = back(qb).view(n, NH, 128), back(kb).view(n, NH, 128), vb.float().view(n, NH, 128)
    s = torch.einsum("qhd,khd->hqk", qf, kf)                                  # q carries softmax_scale * log2(e)
    pr = torch.exp2(s - s.max(-1, keepdim=True).values)
    o = torch.einsum("hqk,khd->qhd", pr / pr.sum(-1, keepdim=True), vf)      # fp32 attention of the same inputs
    ref = torch.einsum("qhd,chd->hqc", o[T:], con.view(C, NH, 128))           # [heads, image rows, C]
    assert torch.all(part[:, :, C:] == -7.0)                                  # columns past C are not written
    err = (part[:, :, :C] - ref).abs().max().item()
    assert err < 6e-3 * max(1.0, ref.abs().max().item()), (err, ref.abs().max().item())   # P is bf16 inside the kernel
    # summed over the heads by the heat-map launch == the vector form on the fp32 rows, to fp32 summation order
    acc_p, acc_v = torch.zeros(C, Limg, device=DEV), torch.zeros(C, Limg, device=DEV)
    lg_p, lg_v = torch.zeros(C, Limg, device=DEV), torch.zeros(C, Limg, device=DEV)
    o32 = torch.zeros(n, H, device=DEV)
    ops.attention([ops.Attn(qb[:T], out_plain[:T], kb[:T], vb[:T], kb[T:], vb[T:], q1=qb[T:], out1=out_plain[T:],
                            out_f32=o32)], NH, q_prescaled=True, qk_f16=qk_f16)
    ops.heatmap_fused([ops.Heatmap(None, None, acc_p, 1.0, logits=lg_p, part=part),
                       ops.Heatmap(o32[T:], con, acc_v, 1.0, logits=lg_v)])
    torch.cuda.synchronize()
    assert (lg_p - lg_v).abs().max().item() < 2e-5 * max(1.0, lg_v.abs().max().item())
    assert (acc_p - acc_v).abs().max().item() < 2e-5
    with pytest.raises(ValueError):   # the old kernel (a softmax scale passed) does not have the feature
        ops.attention([ops.Attn(qb[:T], out[:T], kb[:T], vb[:T], kb[T:], vb[T:], q1=qb[T:], out1=out[T:], hm_con=con,
                                hm_part=part)], NH, scale=0.088)


def test_model_epilogue_logits_equal_the_fp32_row_route():
    """HipFluxDiT.epilogue_logits (the output-space logits from the attention kernel's accumulators, per head) against
    the fp32-rows route it replaces (ATTI32): same arithmetic, another summation order -- maps within 2e-6; the cross
    space, the streams and the latent do not change by one bit."""
    from conceptattention_amd.flux_dit import HeatmapRequest
    B, C, Lp = 3, 3, 256
    m, p, kw = _tiny_forward_kwargs(B)
    kw = dict(kw, stop_after_multimodal_attentions=False)
    res = {}
    for on in (True, False):
        m.epilogue_logits = on
        acc = torch.zeros(B, 2, C, Lp, device=DEV)
        reqs = [HeatmapRequest(tuple(range(p.depth)), 0.5, acc[j, 0], acc[j, 1]) for j in range(B)]
        pred, _ = m(heatmaps=reqs, **kw)
        torch.cuda.synchronize()
        res[on] = (acc, pred)
    assert torch.equal(res[True][1], res[False][1])
    assert torch.equal(res[True][0][:, 1], res[False][0][:, 1])
    d = (res[True][0][:, 0] - res[False][0][:, 0]).abs().max().item()
    assert 0 <= d < 2e-6, d
    m.epilogue_logits = True
