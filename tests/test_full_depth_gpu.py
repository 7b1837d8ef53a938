"""Full-size, full-depth parity on the GPU (BASELINE.json configs[1], configs[0]'s shape at H=3072, configs[4]).

The 57-block flux-schnell geometry with the seeded weights / inputs of oracle/make_full_goldens.py, NOT teacher-forced:
  * 4 diffusion steps, per-(step, layer) concept maps of the fused product path against the fp32 oracle's maps
    (tests/golden/full_depth_schnell.npz).  Yardstick: the REFERENCE'S OWN modules run in bf16 (its production
    dtype) on the same weights and inputs differ from those fp32 maps by the amounts stored in
    tests/golden/full_depth_refbf16.npz; the HIP path must be at least as close:
        err_hip <= max(1e-3, err_reference_bf16)            per step and layer, both spaces, and for the final maps.
  * config 1's shape (256x256 image = 256 tokens, 256 text tokens, ONE concept, one step) through all 57 blocks at
    the full hidden size: a different tile regime of the GEMM (M = 513 rows).
  * the fp8 per-layer x noise-level sweep at full size against the bf16 sweep of the same kernels (no fp8 exists in
    the reference: bound stated from measurement).
One module-scoped model: drawing 11.9 G seeded weights on the host takes about a minute.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conceptattention_amd import ops, sampling  # noqa: E402
from conceptattention_amd.flux_dit import HeatmapRequest  # noqa: E402
from conceptattention_amd.params import configs  # noqa: E402
from conceptattention_amd.pipeline import ConceptAttentionFluxPipeline  # noqa: E402
from conceptattention_amd.weights import synthetic_inputs  # noqa: E402

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}
# Absolute bounds on the max-abs map error vs the fp32 oracle, each <= 1.5 x the value measured on MI355X
# (profiles/r03_full_depth_parity.json).  The north star's bound is 1e-3 on the maps generate_image returns.
# Cross space: with the same input as the oracle (step 0 of a generation, the encode path) a map is within 4.5e-4 -- the
# q vectors come from the unrounded LayerNorm output (HipFluxDiT.split_q_capture); at steps >= 1 of a generation the
# latent itself has moved (pred: 0.24-0.37 % rms off the fp32 trajectory after 57 bf16 blocks per step) and the
# cross-space logits amplify that: 2-4e-3 per map, 1.07e-3 for the 16-map mean.
# Round 4 (profiles/r04_full_depth_parity.json): the attention's q of the captured layers comes from the unrounded
# LayerNorm output as well and q / k are stored as IEEE half: every single (step, layer) map of every configuration is
# now inside the north star's 1e-3 in BOTH spaces (output space: 1.80e-3 -> 5.6e-4).
FINAL_OUT_BOUND = 1.4e-4          # schnell 4 steps: 7.9e-5 capturing all 19 layers, 9.4e-5 as generate_image returns them
                                  # (round 3: 3.15e-4); dev 2 steps: 5.1e-5
FINAL_CROSS_BOUND = 2.7e-4        # schnell 4 steps: 1.77e-4 (bf16 Euler state: 1.05e-3); dev 2 steps: 1.03e-4; encode: 1.5e-4
SINGLE_OUT_BOUND = 8.4e-4         # any (step, layer): <= 5.6e-4 (round 3: 1.80e-3); encode layer 0: 5.3e-4
SINGLE_OUT_STEP0_BOUND = 5.7e-4   # step 0, all 19 layers: 2.2e-4 - 3.8e-4 (round 3: 1.0e-3 - 1.6e-3)
CAPTURE_SET_BOUND = 8.5e-4        # a (step, layer) map of layers 15-18 when ALL 19 layers are captured vs only 15-18, over the
                                  # four steps of a generation (the two trajectories part): <= 5.5e-4; one forward: 3e-4
# capture_independent_image = True (round 5, opt-in): final maps 2.24e-4 / 1.65e-4, worst single map 1.62e-3 / 7.7e-4
# (output space at round 3's level: the map-side attention reads bf16 q / k -- k is shared with the image path)
INDEP_FINAL_OUT_BOUND = 3.4e-4
INDEP_SINGLE_OUT_BOUND = 2.4e-3
# final bf16 latent of a 4-step generation between two capture sets (none / layers 15-18 / all 19), round 5: rel rms
# 9.6e-4 - 9.9e-4, max-abs 0.03125 = ONE bf16 ulp of the largest latent values (5.6); each set is 1.68e-3 rel rms from the
# fp32 oracle's latent
LATENT_CAPTURE_SET_REL_RMS_BOUND = 1.5e-3
LATENT_CAPTURE_SET_MAX_ABS_BOUND = 0.047
SINGLE_CROSS_BOUND = 1.15e-3      # any (step, layer): <= 7.7e-4 (bf16 Euler state: 4.3e-3)
SINGLE_CROSS_SAME_INPUT_BOUND = 6.3e-4   # step 0 / encode path (the oracle's own input): <= 4.2e-4
ENCODE_FINAL_OUT_BOUND = 3.3e-4   # one forward, mean of 4 layers: 2.2e-4 (round 3: 6.9e-4)
ENCODE_FINAL_CROSS_BOUND = 2.5e-4 # 1.6e-4
# fp8 mode (opt-in, never the headline) against the fp32 oracle; bounds to be read next to the measured values
# (round 4, with the qkv projection of the captured layers kept in bf16: output space <= 3.1e-3, cross space <= 8.6e-3,
# arg-max concept agrees on >= 99.7 % of the patches; with every projection in e4m3: 4.0e-2 / 9.5e-2 / 95.3 %)
FP8_OUT_VS_ORACLE_BOUND = 4.6e-3
FP8_CROSS_VS_ORACLE_BOUND = 1.3e-2
FP8_ARGMAX_VS_ORACLE_BOUND = 0.99


def bf_inputs(p, size, T, C):
    return {k: (v.bfloat16().float() if v.is_floating_point() else v)
            for k, v in synthetic_inputs(p, size, size, T, C, seed=5).items()}


@pytest.fixture(scope="module")
def pipe():
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    p = configs["flux-schnell"]
    pl = ConceptAttentionFluxPipeline("flux-schnell", device=DEV, weights=None)
    pl.model.weights.init_synthetic(seed=0, on_device=False)   # the host generator: same numbers as the goldens
    yield pl
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(REPORT, open(os.path.join(ROOT, "gpurun_out", "full_depth_parity.json"), "w"), indent=1)


def run_steps(pl, inp, steps, per_layer=True, ts=None, guidance=0.0, layers=None):
    """The Euler loop of sampling.denoise_steps with one per-layer table per step (fused heat-map path).
    ``ts``: explicit schedule (steps + 1 values), e.g. the head of flux-dev's shifted 50-step schedule.
    ``layers``: the double blocks whose maps are requested (default all); row i of a step's tables then holds the
    maps of layers[i]."""
    m, p = pl.model, pl.params
    d = {k: v.to(DEV) for k, v in inp.items()}
    x = d["latent"].to(torch.bfloat16)
    con, con_ids, con_vec = sampling.concept_inputs(d["concepts"].to(torch.bfloat16), d["vec"].to(torch.bfloat16))
    prep = sampling.prepare_from_embeddings(x, d["txt"].to(torch.bfloat16), d["vec"].to(torch.bfloat16))
    img = prep["img"].to(torch.bfloat16).contiguous().clone()
    L_, C = img.shape[1], con.shape[1]
    ts = sampling.get_schedule(steps, L_, shift=False) if ts is None else list(ts)
    out = torch.zeros(steps, p.depth, C, L_, device=DEV)
    cross = torch.zeros(steps, p.depth, C, L_, device=DEV)
    preds = []
    lat32 = img.float() if getattr(m, "fp32_latent", False) else None    # the Euler state as sampling.denoise_steps keeps it
    m.precompute_conditioning(ts[:-1], prep["vec"], con_vec, guidance)
    for s, (tc, tp) in enumerate(zip(ts[:-1], ts[1:])):
        req = HeatmapRequest(tuple(range(p.depth) if layers is None else layers), 0.0, torch.zeros(C, L_, device=DEV),
                             torch.zeros(C, L_, device=DEV), per_layer_out=out[s], per_layer_cross=cross[s],
                             per_layer_weight=1.0)
        pred, _ = m(img=img if lat32 is None else lat32, img_ids=prep["img_ids"], txt=prep["txt"], txt_ids=prep["txt_ids"], concepts=con,
                    concept_ids=con_ids, concept_vec=con_vec, y=prep["vec"],
                    timesteps=torch.full((1,), tc, device=DEV), guidance=torch.full((1,), guidance, device=DEV),
                    return_vectors=False, heatmaps=req, cond_slot=s)
        preds.append(pred.float().cpu())
        if lat32 is None:
            ops.axpy(img, pred.contiguous(), tp - tc)
        else:
            ops.axpy_f32(lat32, pred.contiguous(), tp - tc)
            img = lat32.to(torch.bfloat16)
    torch.cuda.synchronize()
    return out.cpu().numpy(), cross.cpu().numpy(), preds, img.float().cpu()


def test_four_steps_full_depth_vs_fp32_golden_with_reference_bf16_yardstick(pipe, golden):
    g, y = golden("full_depth_schnell.npz"), golden("full_depth_refbf16.npz")
    p = pipe.params
    inp = bf_inputs(p, 1024, 256, 4)
    out, cross, preds, img = run_steps(pipe, inp, 4)
    rep = {"out": {}, "cross": {}}
    worst = []
    for s in range(4):
        layers = range(p.depth) if s == 0 else range(15, 19)
        for l in layers:
            go = g["out_step0"][l] if s == 0 else g["out_late"][s - 1, l - 15]
            gc = g["cross_step0"][l] if s == 0 else g["cross_late"][s - 1, l - 15]
            eo, ec = float(np.abs(out[s, l] - go).max()), float(np.abs(cross[s, l] - gc).max())
            rep["out"][f"step{s}_layer{l}"] = [eo, float(y["err_out"][s, l])]
            rep["cross"][f"step{s}_layer{l}"] = [ec, float(y["err_cross"][s, l])]
            worst.append((eo / max(1e-3, y["err_out"][s, l]), "out", s, l, eo, float(y["err_out"][s, l])))
            worst.append((ec / max(1e-3, y["err_cross"][s, l]), "cross", s, l, ec, float(y["err_cross"][s, l])))
            agree = float((cross[s, l].argmax(0) == gc.argmax(0)).mean())
            assert agree >= min(0.985, float(y["agree_cross"][s, l])), (s, l, agree)
    fo = float(np.abs(out[:, 15:19].mean((0, 1)) - g["final_out"]).max())
    fc = float(np.abs(cross[:, 15:19].mean((0, 1)) - g["final_cross"]).max())
    rep["final"] = {"out": [fo, float(y["final_err_out_fp32reduce"]), float(y["final_err_out_own_bf16_reduce"])],
                    "cross": [fc, float(y["final_err_cross_fp32reduce"]), float(y["final_err_cross_own_bf16_reduce"])]}
    rows = g["sample_rows"]
    rep["pred_rel_rms"] = [float((preds[s][0, rows] - torch.from_numpy(g["pred_rows"][s])).pow(2).mean().sqrt()
                                 / torch.from_numpy(g["pred_rows"][s]).pow(2).mean().sqrt()) for s in range(4)]
    ref_img = torch.from_numpy(g["final_img_rows"])
    rep["final_latent_rel_rms"] = float((img[0, rows] - ref_img).pow(2).mean().sqrt() / ref_img.pow(2).mean().sqrt())
    rep["reference_bf16_final_latent_rel_rms"] = float(
        (torch.from_numpy(y["final_img_rows"]) - ref_img).pow(2).mean().sqrt() / ref_img.pow(2).mean().sqrt())
    REPORT["four_steps"] = rep
    worst.sort(reverse=True)
    print("worst (err_hip / max(1e-3, err_reference_bf16)):", worst[:6])
    print("final maps:", rep["final"], "pred rel rms per step:", rep["pred_rel_rms"])
    # the gate: at least as close to fp32 as the reference's own bf16 run, per (step, layer) and for the final maps
    for ratio, space, s, l, e, yref in worst:
        assert e <= max(1e-3, yref), (space, s, l, e, yref)
    assert fo <= max(1e-3, float(y["final_err_out_fp32reduce"]))
    assert fc <= max(1e-3, float(y["final_err_cross_fp32reduce"]))
    # ... and in absolute terms (bounds and measured values at the top of this file)
    assert fo <= FINAL_OUT_BOUND, fo
    assert fc <= FINAL_CROSS_BOUND, fc
    assert max(v[0] for k, v in rep["out"].items() if k.startswith("step0")) <= SINGLE_OUT_STEP0_BOUND
    assert max(v[0] for k, v in rep["cross"].items() if k.startswith("step0")) <= SINGLE_CROSS_SAME_INPUT_BOUND
    assert max(v[0] for v in rep["out"].values()) <= SINGLE_OUT_BOUND
    assert max(v[0] for v in rep["cross"].values()) <= SINGLE_CROSS_BOUND
    assert rep["final_latent_rel_rms"] <= max(0.02, rep["reference_bf16_final_latent_rel_rms"])
    # the product entry point gives the same final maps as per-layer tables of the SAME captured layers (same kernels,
    # same order).  The captured set matters at rounding level since round 4: in a captured layer the attention's q of the
    # image and concept rows is formed from the unrounded LayerNorm output (HipFluxDiT.split_q_attention), so a run that
    # captures all 19 layers carries slightly different residual rows into layers 15-18 than one that captures only those.
    d = {k: v.to(DEV) for k, v in inp.items()}
    _, hm, cm = pipe.generate_on_device(d["latent"], d["txt"].bfloat16(), d["vec"].bfloat16(), d["concepts"].bfloat16())
    out4, cross4, _, _ = run_steps(pipe, inp, 4, layers=range(15, 19))
    out4, cross4 = out4[:, :4], cross4[:, :4]       # (rows 0..3 = layers 15..18)
    assert np.abs(hm[0].reshape(4, -1).cpu().numpy() - out4.mean((0, 1))).max() < 1e-5
    assert np.abs(cm[0].reshape(4, -1).cpu().numpy() - cross4.mean((0, 1))).max() < 1e-5
    dep = [float(np.abs(out4 - out[:, 15:19]).max()), float(np.abs(cross4 - cross[:, 15:19]).max())]
    REPORT["four_steps"]["capture_set_dependence_layers_15_18(out,cross)"] = dep
    assert max(dep) <= CAPTURE_SET_BOUND, dep
    # ... and the maps of the product's own capture set are inside the same bounds
    fo4 = float(np.abs(out4.mean((0, 1)) - g["final_out"]).max())
    fc4 = float(np.abs(cross4.mean((0, 1)) - g["final_cross"]).max())
    REPORT["four_steps"]["final_capturing_15_18_only"] = {"out": fo4, "cross": fc4}
    assert fo4 <= FINAL_OUT_BOUND and fc4 <= FINAL_CROSS_BOUND, (fo4, fc4)
    for s_ in range(4):
        for l in range(15, 19):
            go = g["out_step0"][l] if s_ == 0 else g["out_late"][s_ - 1, l - 15]
            gc = g["cross_step0"][l] if s_ == 0 else g["cross_late"][s_ - 1, l - 15]
            assert float(np.abs(out4[s_, l - 15] - go).max()) <= SINGLE_OUT_BOUND, (s_, l)
            assert float(np.abs(cross4[s_, l - 15] - gc).max()) <= SINGLE_CROSS_BOUND, (s_, l)


def test_latent_dependence_on_the_captured_layer_set(pipe, golden):
    """In the reference the concept stream is a read-only side computation (modified_double_stream_block.py:105-119 vs
    :120-168): the generated image does not depend on which maps are asked for.  On this path a layer whose maps are
    requested forms the attention's q of its image rows from the unrounded LayerNorm output and stores q / k as IEEE
    half (HipFluxDiT.split_q_attention, qk_f16 = "captured"), so the residual stream -- and the latent generate_image
    returns -- moves at rounding level with ``layer_indices``.  This pins HOW MUCH: the final
    latent of a 4-step generation capturing no layer, layers 15-18 (the default) and all 19, against each other and
    against the fp32 oracle's latent.  Bounds = 1.5 x measured (profiles/r05_full_depth_parity.json);
    CA_SPLIT_Q_ATTENTION=0 CA_QK_F16=0 removes the dependence (at 7.8e-4 instead of 2.2e-4 per single output-space map)."""
    g = golden("full_depth_schnell.npz")
    p = pipe.params
    inp = bf_inputs(p, 1024, 256, 4)
    rows = g["sample_rows"]
    ref = torch.from_numpy(g["final_img_rows"])
    lat = {}
    for tag, layers in (("none", ()), ("15_18", tuple(range(15, 19))), ("all19", tuple(range(p.depth)))):
        lat[tag] = run_steps(pipe, inp, 4, layers=layers)[3]

    def rel_rms(a, b):
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    rep = {"vs_fp32_oracle_rel_rms": {t: rel_rms(v[0, rows], ref) for t, v in lat.items()}}
    for a, b in (("15_18", "none"), ("all19", "none"), ("all19", "15_18")):
        rep[f"{a}_vs_{b}"] = {"max_abs": float((lat[a] - lat[b]).abs().max()), "rel_rms": rel_rms(lat[a], lat[b]),
                              "latent_max_abs": float(lat[b].abs().max())}
    REPORT["latent_vs_capture_set"] = rep
    print("latent vs captured layer set:", rep)
    assert not torch.equal(lat["15_18"], lat["none"])     # (the dependence exists; if it ever vanishes, drop this test)
    for k in ("15_18_vs_none", "all19_vs_none", "all19_vs_15_18"):
        assert rep[k]["rel_rms"] <= LATENT_CAPTURE_SET_REL_RMS_BOUND, (k, rep[k])
        assert rep[k]["max_abs"] <= LATENT_CAPTURE_SET_MAX_ABS_BOUND, (k, rep[k])
    # every capture set is as close to the fp32 trajectory as any other: the dependence is inside the path's own error
    assert max(rep["vs_fp32_oracle_rel_rms"].values()) <= 0.0026


def test_capture_independent_mode_makes_latent_and_maps_independent_of_the_captured_set(pipe, golden):
    """HipFluxDiT.capture_independent_image = True restores the reference's property exactly: the latent of a 4-step
    generation is bit-identical whether no layer, layers 15-18 or all 19 are captured, and a (step, layer) map does not
    depend on which OTHER layers are captured (every stream that feeds the residuals is formed as in an uncaptured
    layer; the maps come from separate attention problems with the accurate q).  Its maps against the fp32 oracle are
    reported and gated too: the map-side attention then reads bf16 q / k (k is shared with the image path)."""
    g = golden("full_depth_schnell.npz")
    p = pipe.params
    inp = bf_inputs(p, 1024, 256, 4)
    m = pipe.model
    m.capture_independent_image = True
    try:
        o_none, _, _, lat_none = run_steps(pipe, inp, 4, layers=())
        o4, c4, _, lat4 = run_steps(pipe, inp, 4, layers=tuple(range(15, 19)))
        o19, c19, _, lat19 = run_steps(pipe, inp, 4)
    finally:
        m.capture_independent_image = False
    assert torch.equal(lat_none, lat4) and torch.equal(lat_none, lat19)
    assert np.array_equal(o4[:, :4], o19[:, 15:19]) and np.array_equal(c4[:, :4], c19[:, 15:19])
    rep = {"out": {}, "cross": {}}
    for s in range(4):
        for l in (range(p.depth) if s == 0 else range(15, 19)):
            go = g["out_step0"][l] if s == 0 else g["out_late"][s - 1, l - 15]
            gc = g["cross_step0"][l] if s == 0 else g["cross_late"][s - 1, l - 15]
            rep["out"][f"step{s}_layer{l}"] = float(np.abs(o19[s, l] - go).max())
            rep["cross"][f"step{s}_layer{l}"] = float(np.abs(c19[s, l] - gc).max())
    fo = float(np.abs(o19[:, 15:19].mean((0, 1)) - g["final_out"]).max())
    fc = float(np.abs(c19[:, 15:19].mean((0, 1)) - g["final_cross"]).max())
    rep["final"] = {"out": fo, "cross": fc}
    rep["worst_single"] = {"out": max(rep["out"].values()), "cross": max(rep["cross"].values())}
    rows = g["sample_rows"]
    ref = torch.from_numpy(g["final_img_rows"])
    rep["final_latent_rel_rms_vs_fp32_oracle"] = float((lat19[0, rows] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    REPORT["capture_independent_mode"] = rep
    print("capture-independent mode:", rep["final"], rep["worst_single"], rep["final_latent_rel_rms_vs_fp32_oracle"])
    assert fo <= INDEP_FINAL_OUT_BOUND and fc <= FINAL_CROSS_BOUND, (fo, fc)
    assert rep["worst_single"]["out"] <= INDEP_SINGLE_OUT_BOUND and rep["worst_single"]["cross"] <= SINGLE_CROSS_BOUND
    assert rep["final_latent_rel_rms_vs_fp32_oracle"] <= 0.0026


def dev_items(p, n, size=1024, T=256, C=4, first_seed=5):
    """Work items for the batched entry points: item j = synthetic_inputs(seed=first_seed + j), bf16, on the device."""
    items = []
    for j in range(n):
        inp = synthetic_inputs(p, size, size, T, C, seed=first_seed + j, dtype=torch.bfloat16)
        items.append({k: inp[k].to(DEV) for k in ("latent", "txt", "vec", "concepts")})
    return items


def test_five_items_per_forward_at_full_size_equal_single_items_and_the_golden(pipe, golden):
    """The configuration bench.py measures (BASELINE.json configs[1] with 5 work items in every launch: 21 780 rows,
    per-item gate strides, the thin-row launch for the 20 concept rows, 10 attention problems and 15 LayerNorm
    segments per launch, multi-round persistent tile walks) at H = 3072 / 57 blocks / 4 steps:
      (i)   item 0 is the golden's work item: its final maps against the fp32 oracle's, same gates as the B = 1 test;
      (ii)  EVERY item's (latent, heat maps, cross maps) is bit-identical to its own single-item call
            (the reference's call is one item: concept_attention_pipeline.py:115-202);
      (iii) the same for the encode path (stop_after_multimodal_attentions, 5 images per forward)."""
    g, y = golden("full_depth_schnell.npz"), golden("full_depth_refbf16.npz")
    p = pipe.params
    items = dev_items(p, 5)
    many = pipe.generate_many_on_device(items, batch=5)
    torch.cuda.synchronize()
    assert len(many) == 5
    img0, hm0, cm0 = many[0]
    fo = float(np.abs(hm0[0].reshape(4, -1).cpu().numpy() - g["final_out"]).max())
    fc = float(np.abs(cm0[0].reshape(4, -1).cpu().numpy() - g["final_cross"]).max())
    REPORT["batch5"] = {"item0_final_out_err": fo, "item0_final_cross_err": fc}
    assert fo <= max(1e-3, float(y["final_err_out_fp32reduce"])) and fo <= FINAL_OUT_BOUND, fo
    assert fc <= max(1e-3, float(y["final_err_cross_fp32reduce"])) and fc <= FINAL_CROSS_BOUND, fc
    rows = g["sample_rows"]
    ref_img = torch.from_numpy(g["final_img_rows"])
    rel = float((img0[0, rows].float().cpu() - ref_img).pow(2).mean().sqrt() / ref_img.pow(2).mean().sqrt())
    assert rel <= 0.02, rel
    for j, it in enumerate(items):
        one = pipe.generate_on_device(it["latent"], it["txt"], it["vec"], it["concepts"])
        for a, b, what in zip(many[j], one, ("latent", "heat maps", "cross maps")):
            assert torch.equal(a, b), (j, what, float((a.float() - b.float()).abs().max()))
    # the items differ from each other (a batched path that broadcast item 0 would pass the loop above otherwise)
    assert not torch.equal(many[0][1], many[1][1])
    enc = pipe.encode_many_on_device(items, batch=5, seed=11)
    for j, it in enumerate(items):
        one = pipe.encode_many_on_device([it], batch=1, seed=11)[0]
        assert torch.equal(enc[j][0], one[0]) and torch.equal(enc[j][1], one[1]), j
    REPORT["batch5"]["items_bit_identical_to_single"] = True


def test_encode_path_full_size_vs_fp32_golden(pipe, golden):
    """BASELINE.json configs[3]'s unit of work at H = 3072: add_noise_to_image + ONE forward of the 19 double blocks
    (stop_after_multimodal_attentions, y = concept_vec = 0) with two concepts, against the fp32 oracle
    (tests/golden/encode_full.npz) with the reference's own bf16 run as the yardstick (encode_full_refbf16.npz).
    concept_attention_pipeline.py:204-357, segmentation.py:85-113, modified_flux_dit.py:152-153."""
    g, y = golden("encode_full.npz"), golden("encode_full_refbf16.npz")
    p = pipe.params
    C = 2
    inp = bf_inputs(p, 1024, 256, C)
    noise = torch.randn(inp["latent"].shape, generator=torch.Generator().manual_seed(int(g["noise_seed"]))).bfloat16()
    assert abs(noise.float().sum().item() - float(g["noise_checksum"])) < 1e-2   # the golden's host noise stream
    d = {k: v.to(DEV) for k, v in inp.items()}
    item = {"latent": d["latent"].bfloat16(), "txt": d["txt"].bfloat16(), "vec": d["vec"].bfloat16(),
            "concepts": d["concepts"].bfloat16()}
    # per-layer tables through the same code path encode_image takes (layer_noise_sweep shares _encode's forward);
    # here directly: one model call with a per-layer request
    m = pipe.model
    t = float(g["t"])
    x = (t * noise.to(DEV).float() + (1.0 - t) * item["latent"].float()).to(torch.bfloat16)
    assert np.abs(x[0, :, ::16, ::16].float().cpu().numpy() - g["x_rows"]).max() == 0.0   # same noised latent, bit for bit
    con, con_ids, con_vec = sampling.concept_inputs(item["concepts"], item["vec"])
    prep = sampling.prepare_from_embeddings(x, item["txt"], item["vec"])
    L_ = prep["img"].shape[1]
    out = torch.zeros(p.depth, C, L_, device=DEV)
    cross = torch.zeros(p.depth, C, L_, device=DEV)
    req = HeatmapRequest(tuple(range(p.depth)), 0.0, torch.zeros(C, L_, device=DEV), torch.zeros(C, L_, device=DEV),
                         per_layer_out=out, per_layer_cross=cross, per_layer_weight=1.0)
    pred, _ = m(img=prep["img"], img_ids=prep["img_ids"], txt=prep["txt"], txt_ids=prep["txt_ids"], concepts=con,
                concept_ids=con_ids, concept_vec=con_vec, y=con_vec, timesteps=torch.full((1,), t, device=DEV),
                guidance=torch.zeros(1, device=DEV), stop_after_multimodal_attentions=True, return_vectors=False,
                heatmaps=req)
    assert pred is None
    torch.cuda.synchronize()
    out, cross = out.cpu().numpy(), cross.cpu().numpy()
    rep = {"out": {}, "cross": {}}
    for li, l in list(enumerate(range(15, 19))) + [(0, 0), (1, 9)]:
        go, gc = (g["out_layers"][li], g["cross_layers"][li]) if l >= 15 else (g["out_early"][li], g["cross_early"][li])
        eo, ec = float(np.abs(out[l] - go).max()), float(np.abs(cross[l] - gc).max())
        rep["out"][f"layer{l}"] = [eo, float(y["err_out"][0, l])]
        rep["cross"][f"layer{l}"] = [ec, float(y["err_cross"][0, l])]
        assert eo <= max(1e-3, float(y["err_out"][0, l])), (l, eo)
        assert ec <= max(1e-3, float(y["err_cross"][0, l])), (l, ec)
        assert eo <= SINGLE_OUT_BOUND and ec <= SINGLE_CROSS_SAME_INPUT_BOUND, (l, eo, ec)
    # the product entry point (encode_image's device core) on the same noise: the mean over layers 15..18
    ho, hc = pipe.encode_many_on_device([item], batch=1, noise=[noise])[0]
    fo = float(np.abs(ho[0].reshape(C, -1).cpu().numpy() - g["final_out"]).max())
    fc = float(np.abs(hc[0].reshape(C, -1).cpu().numpy() - g["final_cross"]).max())
    rep["final"] = {"out": [fo, float(y["final_err_out_fp32reduce"])], "cross": [fc, float(y["final_err_cross_fp32reduce"])]}
    REPORT["encode_full"] = rep
    print("encode full size:", rep)
    # (against tables of the same captured layers: see the four-step test for why the captured set matters)
    out4 = torch.zeros(p.depth, C, L_, device=DEV)
    cross4 = torch.zeros(p.depth, C, L_, device=DEV)
    req4 = HeatmapRequest(tuple(range(15, 19)), 0.0, torch.zeros(C, L_, device=DEV), torch.zeros(C, L_, device=DEV),
                          per_layer_out=out4, per_layer_cross=cross4, per_layer_weight=1.0)
    m(img=prep["img"], img_ids=prep["img_ids"], txt=prep["txt"], txt_ids=prep["txt_ids"], concepts=con,
      concept_ids=con_ids, concept_vec=con_vec, y=con_vec, timesteps=torch.full((1,), t, device=DEV),
      guidance=torch.zeros(1, device=DEV), stop_after_multimodal_attentions=True, return_vectors=False, heatmaps=req4)
    torch.cuda.synchronize()
    out4 = out4[:4].cpu().numpy()                   # (rows 0..3 = layers 15..18)
    assert np.abs(ho[0].reshape(C, -1).cpu().numpy() - out4.mean(0)).max() < 1e-5
    assert np.abs(out4 - out[15:19]).max() <= CAPTURE_SET_BOUND
    for li in range(4):
        assert float(np.abs(out4[li] - g["out_layers"][li]).max()) <= SINGLE_OUT_BOUND, li
    assert fo <= max(1e-3, float(y["final_err_out_fp32reduce"])) and fo <= ENCODE_FINAL_OUT_BOUND
    assert fc <= max(1e-3, float(y["final_err_cross_fp32reduce"])) and fc <= ENCODE_FINAL_CROSS_BOUND


@pytest.fixture(scope="module")
def dev_pipe(pipe):
    """flux-dev = the schnell geometry + guidance_in (flux/util.py:46 vs :78): the seeded tensors are keyed by name, so
    every shared tensor is copied from the schnell model on the device and only guidance_in is drawn on the host."""
    from conceptattention_amd.weights import synthetic_state_dict
    p = configs["flux-dev"]
    pl = ConceptAttentionFluxPipeline("flux-dev", device=DEV, weights=None)
    missing, unexpected = pl.model.weights.load_state_dict(pipe.model.state_dict(), strict=False)
    assert unexpected == [] and all(k.startswith("guidance_in.") for k in missing), (missing, unexpected)
    pl.model.weights.load_state_dict(synthetic_state_dict(p, seed=0, prefix="guidance_in."), strict=False)
    yield pl
    del pl
    torch.cuda.empty_cache()


def test_flux_dev_two_steps_full_depth_vs_fp32_golden(dev_pipe, golden):
    """BASELINE.json configs[2]'s geometry at full depth: flux-dev (guidance embedding, guidance 3.5), 512 text tokens,
    8 concepts (two passes of the 4-concept heat-map kernels), the first two steps of the shifted 50-step schedule,
    all 57 blocks, not teacher-forced; fp32 oracle maps (full_depth_dev.npz), reference-in-bf16 yardstick
    (full_depth_dev_refbf16.npz).  modified_flux_dit.py:99-104, flux/sampling.py:67-94."""
    g, y = golden("full_depth_dev.npz"), golden("full_depth_dev_refbf16.npz")
    p = dev_pipe.params
    assert p.guidance_embed
    inp = bf_inputs(p, 1024, 512, 8)
    ts = sampling.get_schedule(50, 4096, shift=True)[:3]
    assert np.abs(np.array(ts) - g["schedule"]).max() < 1e-6
    out, cross, preds, img = run_steps(dev_pipe, inp, 2, ts=ts, guidance=float(g["guidance"]))
    rep = {"out": {}, "cross": {}}
    cells = [(0, int(l), g["out_step0"][i], g["cross_step0"][i]) for i, l in enumerate(g["layers_step0"])] + \
            [(1, l, g["out_step1"][l - 15], g["cross_step1"][l - 15]) for l in range(15, 19)]
    for s, l, go, gc in cells:
        eo, ec = float(np.abs(out[s, l] - go).max()), float(np.abs(cross[s, l] - gc).max())
        rep["out"][f"step{s}_layer{l}"] = [eo, float(y["err_out"][s, l])]
        rep["cross"][f"step{s}_layer{l}"] = [ec, float(y["err_cross"][s, l])]
        assert eo <= max(1e-3, float(y["err_out"][s, l])), (s, l, eo, float(y["err_out"][s, l]))
        assert ec <= max(1e-3, float(y["err_cross"][s, l])), (s, l, ec, float(y["err_cross"][s, l]))
        assert eo <= SINGLE_OUT_STEP0_BOUND and ec <= (SINGLE_CROSS_SAME_INPUT_BOUND if s == 0 else SINGLE_CROSS_BOUND), \
            (s, l, eo, ec)
        agree = float((cross[s, l].argmax(0) == gc.argmax(0)).mean())
        assert agree >= min(0.985, float(y["agree_cross"][s, l])), (s, l, agree)
    fo = float(np.abs(out[:, 15:19].mean((0, 1)) - g["final_out"]).max())
    fc = float(np.abs(cross[:, 15:19].mean((0, 1)) - g["final_cross"]).max())
    rows = g["sample_rows"]
    rep["final"] = {"out": [fo, float(y["final_err_out_fp32reduce"])], "cross": [fc, float(y["final_err_cross_fp32reduce"])]}
    rep["pred_rel_rms"] = [float((preds[s][0, rows] - torch.from_numpy(g["pred_rows"][s])).pow(2).mean().sqrt()
                                 / torch.from_numpy(g["pred_rows"][s]).pow(2).mean().sqrt()) for s in range(2)]
    REPORT["flux_dev"] = rep
    print("flux-dev:", rep)
    assert fo <= max(1e-3, float(y["final_err_out_fp32reduce"])) and fo <= FINAL_OUT_BOUND
    assert fc <= max(1e-3, float(y["final_err_cross_fp32reduce"])) and fc <= FINAL_CROSS_BOUND
    assert max(rep["pred_rel_rms"]) < 0.02


def test_config1_shape_at_full_hidden_size(pipe, golden):
    """256x256, one concept, one step, all 57 blocks (M = 256 + 256 + 1 = 513 rows): with C = 1 the softmax over
    concepts is identically 1, so the comparison is on the logits themselves and on `pred`."""
    g = golden("cfg1_full_hidden.npz")
    p = pipe.params
    inp = bf_inputs(p, 256, 256, 1)
    m = pipe.model
    d = {k: v.to(DEV) for k, v in inp.items()}
    prep = sampling.prepare_from_embeddings(d["latent"].to(torch.bfloat16), d["txt"].bfloat16(), d["vec"].bfloat16())
    con, con_ids, con_vec = sampling.concept_inputs(d["concepts"].bfloat16(), d["vec"].bfloat16())
    pred, dd = m(img=prep["img"], img_ids=prep["img_ids"], txt=prep["txt"], txt_ids=prep["txt_ids"], concepts=con,
                 concept_ids=con_ids, concept_vec=con_vec, y=prep["vec"], timesteps=torch.ones(1, device=DEV),
                 guidance=torch.zeros(1, device=DEV), return_vectors=True)
    torch.cuda.synchronize()
    assert dd["output_space_image_vectors"].shape == (19, 1, 256, 3072)
    assert dd["cross_attention_concept_vectors"].shape == (19, 1, 24, 1, 128)
    iv, cv = dd["output_space_image_vectors"].float().cpu(), dd["output_space_concept_vectors"].float().cpu()
    lo = torch.einsum("lbpd,lbcd->lcp", iv, cv).numpy()
    iq = dd["cross_attention_image_vectors"].float().cpu().permute(0, 1, 3, 2, 4).reshape(19, 256, 3072)
    cq = dd["cross_attention_concept_vectors"].float().cpu().permute(0, 1, 3, 2, 4).reshape(19, 1, 3072)
    lc = torch.einsum("lpd,lcd->lcp", iq, cq).numpy()
    rep = {"logits_out_err": [], "logits_cross_err": []}
    for l in range(19):
        so, sc = np.abs(g["logits_out"][l]).max(), np.abs(g["logits_cross"][l]).max()
        rep["logits_out_err"].append([float(np.abs(lo[l] - g["logits_out"][l]).max()), float(so)])
        rep["logits_cross_err"].append([float(np.abs(lc[l] - g["logits_cross"][l]).max()), float(sc)])
    gp = torch.from_numpy(g["pred"])
    rep["pred_rel_rms"] = float((pred[0].float().cpu() - gp).pow(2).mean().sqrt() / gp.pow(2).mean().sqrt())
    rep["attn_rows_maxabs"] = float(np.abs(iv[:, 0, ::16].numpy() - g["img_attn_rows"]).max())
    REPORT["config1"] = rep
    print("config1:", rep)
    for (e, s) in rep["logits_out_err"]:
        assert e <= 2e-2 * max(s, 1.0)
    for (e, s) in rep["logits_cross_err"]:
        assert e <= 3e-2 * max(s, 1.0)
    assert rep["pred_rel_rms"] < 0.03
    assert rep["attn_rows_maxabs"] < 1e-2


def test_fp8_forward_full_size_vs_fp32_oracle(pipe, golden):
    """BASELINE.json configs[4] names fp8; the reference has none, so round 3 only compared the fp8 sweep with the bf16
    sweep of the same kernels.  tests/golden/encode_full.npz holds the fp32 ORACLE's maps of layers 0, 9, 15-18 for
    exactly such a forward (one noised image, the 19 double blocks, C = 2): the same forward with every projection in
    e4m3 against them.  fp8 is the opt-in reduced-precision mode, not the parity path: the bound states what was
    measured (profiles/r04_full_depth_parity.json, "fp8_vs_oracle"), the bf16 path's figure is printed beside it."""
    g = golden("encode_full.npz")
    p = pipe.params
    C = 2
    inp = bf_inputs(p, 1024, 256, C)
    noise = torch.randn(inp["latent"].shape, generator=torch.Generator().manual_seed(int(g["noise_seed"]))).bfloat16()
    d = {k: v.to(DEV) for k, v in inp.items()}
    t = float(g["t"])
    x = (t * noise.to(DEV).float() + (1.0 - t) * d["latent"].bfloat16().float()).to(torch.bfloat16)
    con, con_ids, con_vec = sampling.concept_inputs(d["concepts"].bfloat16(), d["vec"].bfloat16())
    prep = sampling.prepare_from_embeddings(x, d["txt"].bfloat16(), d["vec"].bfloat16())
    L_ = prep["img"].shape[1]
    m = pipe.model

    def tables():
        out = torch.zeros(p.depth, C, L_, device=DEV)
        cross = torch.zeros(p.depth, C, L_, device=DEV)
        req = HeatmapRequest(tuple(range(p.depth)), 0.0, torch.zeros(C, L_, device=DEV), torch.zeros(C, L_, device=DEV),
                             per_layer_out=out, per_layer_cross=cross, per_layer_weight=1.0)
        m(img=prep["img"], img_ids=prep["img_ids"], txt=prep["txt"], txt_ids=prep["txt_ids"], concepts=con,
          concept_ids=con_ids, concept_vec=con_vec, y=con_vec, timesteps=torch.full((1,), t, device=DEV),
          guidance=torch.zeros(1, device=DEV), stop_after_multimodal_attentions=True, return_vectors=False, heatmaps=req)
        torch.cuda.synchronize()
        return out.cpu().numpy(), cross.cpu().numpy()
    b_out, b_cross = tables()
    m.set_precision("fp8")
    try:
        f_out, f_cross = tables()
    finally:
        m.set_precision("bf16")
    rep = {}
    for li, l in list(enumerate(range(15, 19))) + [(0, 0), (1, 9)]:
        go, gc = (g["out_layers"][li], g["cross_layers"][li]) if l >= 15 else (g["out_early"][li], g["cross_early"][li])
        rep[f"layer{l}"] = {"fp8_out": float(np.abs(f_out[l] - go).max()), "fp8_cross": float(np.abs(f_cross[l] - gc).max()),
                            "bf16_out": float(np.abs(b_out[l] - go).max()), "bf16_cross": float(np.abs(b_cross[l] - gc).max()),
                            "fp8_argmax_agree_out": float((f_out[l].argmax(0) == go.argmax(0)).mean()),
                            "fp8_argmax_agree_cross": float((f_cross[l].argmax(0) == gc.argmax(0)).mean())}
    REPORT["fp8_vs_oracle"] = rep
    print("fp8 forward vs the fp32 oracle:", rep)
    assert np.isfinite(f_out).all() and np.isfinite(f_cross).all()
    assert max(v["fp8_out"] for v in rep.values()) < FP8_OUT_VS_ORACLE_BOUND
    assert max(v["fp8_cross"] for v in rep.values()) < FP8_CROSS_VS_ORACLE_BOUND
    assert min(v["fp8_argmax_agree_out"] for v in rep.values()) > FP8_ARGMAX_VS_ORACLE_BOUND


def test_fp8_layer_noise_sweep_full_size_vs_bf16(pipe):
    """BASELINE.json configs[4]: all 19 double blocks x noise levels, fp8 projections, against the bf16 sweep."""
    p = pipe.params
    inp = bf_inputs(p, 1024, 256, 4)
    d = {k: v.to(DEV) for k, v in inp.items()}
    args = (d["latent"].bfloat16(), d["txt"].bfloat16(), d["vec"].bfloat16(), d["concepts"].bfloat16())
    kw = dict(noise_levels=[10, 40], num_steps=50)
    b_out, b_cross = pipe.layer_noise_sweep_on_device(*args, **kw)
    pipe.model.set_precision("fp8")
    try:
        f_out, f_cross = pipe.layer_noise_sweep_on_device(*args, **kw)
        f2_out, _ = pipe.layer_noise_sweep_on_device(*args, **kw)
    finally:
        pipe.model.set_precision("bf16")
    torch.cuda.synchronize()
    assert f_out.shape == (2, 19, 4, 64, 64) and torch.equal(f_out, f2_out)
    e_out = (f_out - b_out).abs().amax((2, 3, 4)).cpu().numpy()
    e_cross = (f_cross - b_cross).abs().amax((2, 3, 4)).cpu().numpy()
    agree = (f_out.argmax(2) == b_out.argmax(2)).float().mean().item()
    REPORT["fp8_sweep"] = {"out_maxabs_per_level_layer": e_out.tolist(), "cross_maxabs_per_level_layer": e_cross.tolist(),
                           "argmax_agree": agree}
    print("fp8 sweep: out", e_out.max(), "cross", e_cross.max(), "argmax agree", agree)
    assert torch.isfinite(f_out).all() and abs(f_out.sum(2).mean().item() - 1) < 1e-4
    # (round 4: 2.5e-3 / 7.4e-3 / 99.8 % -- the qkv projection of a captured layer stays bf16; round 3: 3.1e-2 / 8.4e-2 / 96.6 %)
    assert e_out.max() < 3.8e-3 and e_cross.max() < 1.2e-2 and agree > 0.99
