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
