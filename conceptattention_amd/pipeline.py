"""ConceptAttentionFluxPipeline on MI355X: same public surface as the reference's
``concept_attention/concept_attention_pipeline.py:94-357`` (constructor arguments, ``generate_image``
and ``encode_image`` signatures and defaults, ``ConceptAttentionPipelineOutput``), running the DiT
and the heat-map reduction in the gfx950 kernels.

Out of scope: the T5/CLIP text encoders and the VAE need checkpoints that
are not available offline.  They are injectable (``text_encoder`` / ``autoencoder``); without them
the pipeline runs in *synthetic conditioning* mode -- prompt and concept strings are mapped to
seeded N(0,1) embeddings of the right shapes and ``image`` is returned as the unpacked latent --
which is exactly the configuration BASELINE.json measures.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib, ops, sampling
from .flux_dit import HeatmapRequest, HipFluxDiT, on_own_device
from .heatmaps import compute_heatmaps_from_vectors, resolve_norm
from .params import configs


@dataclass
class ConceptAttentionPipelineOutput:
    """concept_attention_pipeline.py:20-24."""
    image: object  # PIL.Image.Image | np.ndarray
    concept_heatmaps: list
    cross_attention_maps: list


class SyntheticTextEncoder:
    """Stand-in for HFEmbedder (flux/modules/conditioner.py): deterministic N(0,1) embeddings
    keyed by the text, T5-like (1,T,4096) and CLIP-like pooled (1,768)."""

    def __init__(self, n_tokens: int, context_dim: int = 4096, vec_dim: int = 768, device="cuda:0"):
        self.n_tokens, self.context_dim, self.vec_dim, self.device = n_tokens, context_dim, vec_dim, device

    def _gen(self, text: str, salt: int):
        g = torch.Generator(device="cpu")
        g.manual_seed((zlib.crc32(text.encode()) * 31 + salt) & 0x7FFFFFFF)
        return g

    def t5(self, text: str) -> torch.Tensor:
        return torch.randn(1, self.n_tokens, self.context_dim, generator=self._gen(text, 1)).to(
            self.device, torch.bfloat16)

    def clip(self, text: str) -> torch.Tensor:
        return torch.randn(1, self.vec_dim, generator=self._gen(text, 2)).to(self.device, torch.bfloat16)


def colorize_heatmaps(heatmaps: np.ndarray, cmap: str = "plasma") -> list:
    """Global min-max over ALL concepts, matplotlib colormap, uint8 RGB -> PIL
    (concept_attention_pipeline.py:174-196)."""
    import matplotlib.pyplot as plt
    import PIL.Image
    lo, hi = heatmaps.min(), heatmaps.max()
    out = []
    for hm in heatmaps:
        hm = (hm - lo) / (hi - lo)
        rgb = (plt.get_cmap(cmap)(hm)[:, :, :3] * 255).astype(np.uint8)
        out.append(PIL.Image.fromarray(rgb))
    return out


class ConceptAttentionFluxPipeline:
    def __init__(self, model_name: str = "flux-schnell", offload_model: bool = False, device="cuda:0",
                 weights="synthetic", weight_seed: int = 0, text_encoder=None, autoencoder=None,
                 params=None, n_text_tokens: Optional[int] = None, precision: str = "bf16",
                 residual_dtype=torch.float32, capture_independent_image: bool = False):
        """model_name / offload_model / device as in the reference (:100-113).  ``weights`` is
        "synthetic" (seeded random init), a path to a flux1-*.safetensors file, or a state dict.
        ``precision="fp8"`` runs the large projections on e4m3 operands (HipFluxDiT.set_precision);
        ``residual_dtype`` is the storage type of the residual streams (fp32 by default, HipFluxDiT.__init__);
        ``capture_independent_image=True`` makes the returned latent (and every map) independent of ``layer_indices``
        bit for bit, as in the reference, for the image rows' attention twice in the captured layers
        (HipFluxDiT.capture_independent_image)."""
        if params is None and model_name not in configs:
            raise KeyError(model_name)
        self.model_name = model_name
        self.offload_model = offload_model
        self.device = torch.device(device)
        self.is_schnell = model_name == "flux-schnell"
        self.params = params if params is not None else configs[model_name]
        # the generator owns model / encoders / autoencoder, as in the reference (:104-113)
        from .image_generator import FluxGenerator
        self.flux_generator = FluxGenerator(model_name=model_name, offload=offload_model, device=self.device,
                                            weights=weights, weight_seed=weight_seed, text_encoder=text_encoder,
                                            autoencoder=autoencoder, params=self.params,
                                            n_text_tokens=n_text_tokens, residual_dtype=residual_dtype)
        self.model = self.flux_generator.model
        self.model.set_precision(precision)
        self.model.capture_independent_image = bool(capture_independent_image)
        self.fp8_keep_heatmap_layers = True  # generate path only; the sweeps/encode path run every block in fp8
        self._replicas = [self.model]  # activation sets that share self.model's weights (one per stream)
        self._streams = []
        self.text_encoder = self.flux_generator.text_encoder
        self.autoencoder = autoencoder

    # ------------------------------------------------------------------ conditioning
    def _embed(self, prompt: str, concepts: Sequence[str]):
        return self.flux_generator.embed(prompt, concepts)

    def _finish(self, image, concept_heatmaps, cross_attention_maps, return_pil_heatmaps, cmap):
        concept_heatmaps = concept_heatmaps.to(torch.float32).detach().cpu().numpy()[0]
        cross_attention_maps = cross_attention_maps.to(torch.float32).detach().cpu().numpy()[0]
        if return_pil_heatmaps:
            concept_heatmaps = colorize_heatmaps(concept_heatmaps, cmap)
            cross_attention_maps = colorize_heatmaps(cross_attention_maps, cmap)
        return ConceptAttentionPipelineOutput(image=image, concept_heatmaps=concept_heatmaps,
                                              cross_attention_maps=cross_attention_maps)

    def _decode(self, x: torch.Tensor, height: int, width: int):
        return self.flux_generator.decode(x, height, width)

    # ------------------------------------------------------------------ generate_image (:115-202)
    @torch.no_grad()
    @on_own_device
    def generate_image(self, prompt: str, concepts: list, width: int = 1024, height: int = 1024,
                       return_cross_attention=False, layer_indices=list(range(15, 19)),
                       return_pil_heatmaps=True, seed: int = 0, num_inference_steps: int = 4,
                       guidance: float = 0.0, timesteps=None, softmax: bool = True,
                       attention_norm: str = "sparsemax", cmap="plasma", latent: Optional[torch.Tensor] = None,
                       fused: bool = True) -> ConceptAttentionPipelineOutput:
        """``latent`` (1,16,h/8,w/8) overrides get_noise (device RNG differs between platforms);
        ``fused=False`` takes the reference's route (stack the per-layer vectors, then reduce)."""
        assert return_cross_attention is False, "Not supported yet"
        assert all([0 <= li < self.params.depth for li in layer_indices]), "Invalid layer index"
        assert height == width, "Height and width must be the same for now"
        norm = resolve_norm(softmax, attention_norm)  # ValueError on an unknown name, as the reference (:70-71)
        if timesteps is None:
            timesteps = list(range(num_inference_steps))
        x = latent if latent is not None else sampling.get_noise(1, height, width, self.device, torch.bfloat16, seed)
        txt, vec, con, con_ids, con_vec = self._embed(prompt, concepts)
        img, concept_heatmaps, cross_attention_maps = self.generate_on_device(
            x, txt, vec, con, layer_indices=layer_indices, num_inference_steps=num_inference_steps,
            guidance=guidance, timesteps=timesteps, fused=fused, norm=norm)
        image = self._decode(img, height, width)
        return self._finish(image, concept_heatmaps, cross_attention_maps, return_pil_heatmaps, cmap)

    @torch.no_grad()
    @on_own_device
    def generate_many_on_device(self, items, n_streams: int = 1, batch: int = 1, **kw):
        """Throughput mode for independent work items (dicts with latent/txt/vec/concepts, each with a leading
        batch dimension of 1 and equal shapes).

        ``batch`` items at a time go through ONE forward (one launch per kernel for all of them): the launches then
        hold 5 x 17 = 85 row tiles / 5 x 408 attention workgroups, which fill the 256 CUs to 99.6 % in their last
        round, where a single item leaves 15-20 % of the chip idle in every launch.
        ``n_streams`` groups are kept in flight on separate HIP streams, each with its own activation set and the
        shared weights.  Every item's result is bit-identical to ``generate_on_device`` on that item alone.
        Returns [(img, heat, cross), ...] in item order."""
        batch = max(1, min(batch, _lib.ATTN_MAX_PROBLEMS // 2, _lib.MAX_SEGMENTS // 3))  # launch limits: 5 items
        groups = [list(range(g0, min(g0 + batch, len(items)))) for g0 in range(0, len(items), batch)]
        n_streams = max(1, min(n_streams, len(groups)))
        while len(self._replicas) < n_streams:
            self._replicas.append(HipFluxDiT(self.params, self.device, weights=self.model.weights,
                                             precision=self.model.precision,
                                             residual_dtype=self.model.residual_dtype)
                                  .set_precision(self.model.precision, self.model.keep_bf16_layers))
        for r in self._replicas:
            r.capture_independent_image = self.model.capture_independent_image
        while len(self._streams) < n_streams:
            self._streams.append(torch.cuda.Stream(device=self.device))
        cur = torch.cuda.current_stream(self.device)
        self.model.materialize()  # shared fp8 weight images: built on `cur`, which every side stream waits on
        results = [None] * len(items)

        def cat(idx, key):
            return torch.cat([items[i][key] for i in idx], 0) if len(idx) > 1 else items[idx[0]][key]
        for w0 in range(0, len(groups), n_streams):
            wave = groups[w0:w0 + n_streams]
            gens = {}
            for slot, idx in enumerate(wave):
                st = self._streams[slot] if n_streams > 1 else cur
                if n_streams > 1:
                    st.wait_stream(cur)
                with torch.cuda.stream(st):
                    gens[slot] = self._generate_steps(self._replicas[slot], cat(idx, "latent"), cat(idx, "txt"),
                                                      cat(idx, "vec"), cat(idx, "concepts"), **kw)
            alive = dict(gens)
            done = {}
            while alive:
                for slot in list(alive):
                    with torch.cuda.stream(self._streams[slot] if n_streams > 1 else cur):
                        try:
                            next(alive[slot])
                        except StopIteration as stop:
                            done[slot] = stop.value
                            del alive[slot]
            for slot, idx in enumerate(wave):
                img, hm, cm = done[slot]
                if n_streams > 1:
                    cur.wait_stream(self._streams[slot])
                    for t in (img, hm, cm):  # allocated on the side stream, consumed on the caller's stream
                        t.record_stream(cur)
                for k, i in enumerate(idx):
                    results[i] = (img[k:k + 1], hm[k:k + 1], cm[k:k + 1])
        return results

    @torch.no_grad()
    @on_own_device
    def generate_on_device(self, latent, txt, vec, concept_embeddings, layer_indices=list(range(15, 19)),
                           num_inference_steps: int = 4, guidance: float = 0.0, timesteps=None, fused: bool = True,
                           norm: int = 0):
        gen = self._generate_steps(self.model, latent, txt, vec, concept_embeddings, layer_indices,
                                   num_inference_steps, guidance, timesteps, fused, norm)
        while True:
            try:
                next(gen)
            except StopIteration as stop:
                return stop.value

    def _generate_steps(self, model, latent, txt, vec, concept_embeddings, layer_indices=list(range(15, 19)),
                        num_inference_steps: int = 4, guidance: float = 0.0, timesteps=None, fused: bool = True,
                        norm: int = 0):
        """The device-resident core of generate_image for B work items at once (B = 1 from the public API):
        latent (B,16,h/8,w/8), txt (B,T,4096), vec (B,768), concept_embeddings (B,C,4096) already in HBM ->
        (final latent tokens, concept heat maps fp32 [B,C,side,side], cross-attention maps fp32
        [B,C,side,side]) on the device, no host synchronisation."""
        if timesteps is None:
            timesteps = list(range(num_inference_steps))
        x = latent.to(self.device, torch.bfloat16)
        schedule = sampling.get_schedule(num_inference_steps, x.shape[-1] * x.shape[-2] // 4,
                                         shift=(not self.is_schnell))
        con, con_ids, con_vec = sampling.concept_inputs(concept_embeddings, vec)
        inp = sampling.prepare_from_embeddings(x, txt, vec)
        C, n_patches = con.shape[1], inp["img"].shape[1]
        # the reference indexes the stacked maps with `timesteps` (concept_attention_pipeline.py:76): an index
        # outside [-steps, steps) is an IndexError there too; negative indices count from the end
        ts = []
        for t in timesteps:
            t = int(t)
            if not -num_inference_steps <= t < num_inference_steps:
                raise IndexError(f"timestep index {t} is out of range for {num_inference_steps} inference steps")
            ts.append(t % num_inference_steps)
        ls = [int(l) for l in layer_indices]
        keep_before = model.keep_bf16_layers
        if model.precision == "fp8" and self.fp8_keep_heatmap_layers:
            # the double blocks whose attention outputs become heat maps stay in bf16: 4 of 57 blocks, no
            # measurable cost, and the maps move from 1.1e-2 to 1.8e-3 of the bf16 path
            model.set_precision("fp8", keep_bf16_layers=ls)
        try:
            return (yield from self._generate_steps_inner(model, x, inp, con, con_ids, con_vec, schedule, ts, ls,
                                                          C, n_patches, guidance, layer_indices, timesteps, fused,
                                                          norm))
        finally:
            model.keep_bf16_layers = keep_before  # this call's choice does not leak into later sweeps / encodes

    def _generate_steps_inner(self, model, x, inp, con, con_ids, con_vec, schedule, ts, ls, C, n_patches, guidance,
                              layer_indices, timesteps, fused, norm=0):
        names = {v: k for k, v in _lib.NORMS.items()}
        # repeated indices weigh a (step, layer) pair repeatedly in the reference's fancy indexing
        # (concept_attention_pipeline.py:76-77); the fused accumulation covers distinct pairs only
        if fused and (len(set(ts)) != len(ts) or len(set(ls)) != len(ls)):
            fused = False
        B = x.shape[0]
        if fused:
            acc_o = torch.zeros(B, C, n_patches, device=self.device)
            acc_c = torch.zeros(B, C, n_patches, device=self.device)
            reqs = [HeatmapRequest(tuple(ls), 1.0 / (len(ts) * len(ls)), acc_o[j], acc_c[j], norm=norm)
                    for j in range(B)]
            img, _, _ = yield from sampling.denoise_steps(
                model, **inp, timesteps=schedule, guidance=guidance, concepts=con, concept_ids=con_ids,
                concept_vec=con_vec, return_intermediate_images=False, return_vectors=False, heatmaps=reqs,
                heatmap_timesteps=ts)
            side = int(round(n_patches ** 0.5))
            return img, acc_o.view(B, C, side, side), acc_c.view(B, C, side, side)
        if B != 1:
            raise NotImplementedError("the stacked (fused=False) route takes one work item at a time")
        img, _, d = yield from sampling.denoise_steps(
            model, **inp, timesteps=schedule, guidance=guidance, concepts=con, concept_ids=con_ids,
            concept_vec=con_vec, return_intermediate_images=False)
        cross_attention_maps = compute_heatmaps_from_vectors(
            d["cross_attention_image_vectors"], d["cross_attention_concept_vectors"],
            layer_indices=layer_indices, timesteps=timesteps, softmax=False, attention_norm=names[norm])
        concept_heatmaps = compute_heatmaps_from_vectors(
            d["output_space_image_vectors"], d["output_space_concept_vectors"],
            layer_indices=layer_indices, timesteps=timesteps, softmax=False, attention_norm=names[norm])
        return img, concept_heatmaps, cross_attention_maps

    # ------------------------------------------------------------------ per-layer x per-noise-level tables
    @torch.no_grad()
    @on_own_device
    def layer_noise_sweep_on_device(self, latent, txt, vec, concept_embeddings, noise_levels, num_steps: int = 50,
                                    layer_indices=None, seed: int = 0, num_samples: int = 1,
                                    rank: int = 0, world: int = 1, batch: int = 1):
        """Concept maps per (noise level, double block): the workload of the reference's per-layer /
        per-timestep segmentation sweeps (experiments/per_layer_segmentation/test_segmentations_per_layer.py:
        104-114,164-189; experiments/per_timestep_segmentation/test_segmentations_per_time.py:75-104) without
        ever materialising the vector stacks.  Every noise level is an independent forward of the 19 double
        blocks from x = t*noise + (1-t)*latent (concept_attention/segmentation.py:85-113), so the levels shard
        over ranks: this rank computes levels rank, rank+world, ...; rows of other ranks stay
        zero and one all_reduce(sum) (distributed.allreduce_sum_) or all_gather completes the table.
        ``batch`` levels at a time share one forward (each level is a work item with its own timestep; per level
        bit-identical to batch = 1).
        noise_levels: indices into get_schedule(num_steps).  Returns (out_space, cross_space), each
        fp32 [len(noise_levels), len(layer_indices), C, side, side]."""
        layer_indices = list(range(self.params.depth)) if layer_indices is None else [int(l) for l in layer_indices]
        latent = latent.to(self.device, torch.bfloat16)
        n_patches = (latent.shape[-1] // 2) * (latent.shape[-2] // 2)
        side = int(round(n_patches ** 0.5))
        con, con_ids, con_vec = sampling.concept_inputs(concept_embeddings, vec)
        C = con.shape[1]
        schedule = sampling.get_schedule(num_steps, n_patches, shift=(not self.is_schnell))
        nl, nlay = len(noise_levels), len(layer_indices)
        out = torch.zeros(nl, nlay, C, n_patches, device=self.device)
        cross = torch.zeros(nl, nlay, C, n_patches, device=self.device)
        height, width = latent.shape[-2] * 8, latent.shape[-1] * 8
        batch = max(1, min(batch, _lib.ATTN_MAX_PROBLEMS // 2, _lib.MAX_SEGMENTS // 3))
        mine = list(range(rank, nl, world))
        for g0 in range(0, len(mine), batch):
            grp = mine[g0:g0 + batch]
            B = len(grp)
            ts = [schedule[int(noise_levels[li])] for li in grp]
            reqs = [HeatmapRequest(tuple(layer_indices), 0.0, None, None, per_layer_out=out[li],
                                   per_layer_cross=cross[li], per_layer_weight=1.0 / num_samples) for li in grp]

            def rep(t):   # the B levels are B work items of the same image
                return t if B == 1 else t.expand(B, *t.shape[1:]).contiguous()
            for s_i in range(num_samples):
                noise = sampling.get_noise(1, height, width, self.device, torch.bfloat16, seed + s_i)
                x = torch.cat([(t * noise.float() + (1.0 - t) * latent.float()).to(torch.bfloat16) for t in ts], 0)
                inp = sampling.prepare_from_embeddings(x, rep(txt), rep(vec))
                cB, idB, vB = sampling.concept_inputs(rep(con), rep(vec))
                self.model(img=inp["img"], img_ids=inp["img_ids"], txt=inp["txt"], txt_ids=inp["txt_ids"],
                           concepts=cB, concept_ids=idB, concept_vec=vB, y=vB,
                           timesteps=ops.host_values(ts, self.device),
                           guidance=torch.zeros(B, device=self.device), stop_after_multimodal_attentions=True,
                           return_vectors=False, heatmaps=reqs)
        return out.view(nl, nlay, C, side, side), cross.view(nl, nlay, C, side, side)

    # ------------------------------------------------------------------ encode_image (:204-357)
    @torch.no_grad()
    @on_own_device
    def encode_image(self, image, concepts: list, prompt: str = "", width: int = 1024, height: int = 1024,
                     layer_indices=list(range(15, 19)), num_samples: int = 1, num_steps: int = 4,
                     noise_timestep: int = 2, device: str = "cuda:0", return_pil_heatmaps: bool = True,
                     seed: int = 0, cmap="plasma", stop_after_multi_modal_attentions=True,
                     attention_norm: str = "sparsemax", softmax=True,
                     joint_attention_kwargs=None, noise=None) -> ConceptAttentionPipelineOutput:
        """``image``: a latent tensor (1,16,h/8,w/8), or a PIL image when an autoencoder was injected.
        One forward of the 19 double blocks per noise sample (stop_after_multimodal_attentions).
        ``joint_attention_kwargs`` (not in the reference's signature, which hard-codes None at :296) lets the
        segmentation harness select the concept cross/self-attention ablations; ``noise`` (a list of num_samples
        tensors shaped like the latent) overrides get_noise, whose device RNG stream differs between platforms."""
        assert all([0 <= li < self.params.depth for li in layer_indices]), "Invalid layer index"
        assert height == width, "Height and width must be the same for now"
        norm = resolve_norm(softmax, attention_norm)
        if isinstance(image, torch.Tensor):
            latent = image.to(self.device, torch.bfloat16)
        elif self.autoencoder is not None:
            arr = torch.from_numpy(np.asarray(image.convert("RGB"))).permute(2, 0, 1).float() / 255.0
            arr = torch.nn.functional.interpolate((2.0 * arr - 1.0)[None].to(self.device), (height, width))
            latent = self.autoencoder.encode(arr).to(torch.bfloat16)
        else:
            raise ValueError("encode_image needs a latent tensor or an injected autoencoder")
        txt, vec, con, con_ids, con_vec = self._embed(prompt, concepts)
        out_space, cross_space = self._encode_maps(self.model, latent, txt, vec, con, con_ids, con_vec, layer_indices,
                                                   num_samples, num_steps, noise_timestep, seed,
                                                   stop_after_multi_modal_attentions, joint_attention_kwargs, norm,
                                                   noise=noise)
        return self._finish(image, out_space, cross_space, return_pil_heatmaps, cmap)

    def _encode_maps(self, model, latent, txt, vec, con, con_ids, con_vec, layer_indices, num_samples, num_steps,
                     noise_timestep, seed, stop_after_multi_modal_attentions=True, joint_attention_kwargs=None,
                     norm: int = 0, noise=None):
        """Device core of encode_image for B images at once (latent (B,16,h,w), txt (B,T,4096), ...): one forward
        per noise sample on ``model``; returns the two fp32 maps [B, C, side, side].  ``noise``: optional list of
        num_samples tensors (1 or B,16,h,w) used instead of get_noise(seed + i)."""
        if noise is not None and len(noise) != num_samples:
            raise ValueError("noise: one tensor per noise sample")
        B, C = con.shape[0], con.shape[1]
        n_patches = (latent.shape[-1] // 2) * (latent.shape[-2] // 2)
        height, width = latent.shape[-2] * 8, latent.shape[-1] * 8
        # the reference indexes the stacked samples with the float schedule values
        # (concept_attention_pipeline.py:311), which selects sample 0 only unless
        # a value >= 1; here every sample contributes equally (identical at num_samples=1).
        acc_o = torch.zeros(B, C, n_patches, device=self.device)
        acc_c = torch.zeros(B, C, n_patches, device=self.device)
        req = [HeatmapRequest(tuple(int(l) for l in layer_indices), 1.0 / (num_samples * len(layer_indices)),
                              acc_o[j], acc_c[j], norm=norm) for j in range(B)]
        schedule = sampling.get_schedule(num_steps, n_patches, shift=(not self.is_schnell))
        for i in range(num_samples):
            # add_noise_to_image (concept_attention/segmentation.py:85-113)
            nz = (noise[i].to(self.device, torch.bfloat16) if noise is not None
                  else sampling.get_noise(1, height, width, self.device, torch.bfloat16, seed + i))
            t = schedule[noise_timestep]
            x = (t * nz.float() + (1.0 - t) * latent.float()).to(torch.bfloat16)
            inp = sampling.prepare_from_embeddings(x, txt, vec)
            t_vec = torch.full((B,), schedule[noise_timestep], device=self.device)
            model(img=inp["img"], img_ids=inp["img_ids"], txt=inp["txt"], txt_ids=inp["txt_ids"],
                  concepts=con, concept_ids=con_ids, concept_vec=con_vec, y=con_vec, timesteps=t_vec,
                  guidance=torch.zeros(B, device=self.device),
                  stop_after_multimodal_attentions=stop_after_multi_modal_attentions,
                  joint_attention_kwargs=joint_attention_kwargs, return_vectors=False, heatmaps=req)
        side = int(round(n_patches ** 0.5))
        return acc_o.view(B, C, side, side), acc_c.view(B, C, side, side)

    @torch.no_grad()
    @on_own_device
    def encode_many_on_device(self, items, n_streams: int = 1, batch: int = 1, layer_indices=list(range(15, 19)),
                              num_samples: int = 1, num_steps: int = 4, noise_timestep: int = 2, seed: int = 0,
                              noise=None):
        """Batch form of encode_image for independent images (the loop of
        experiments/imagenet_segmentation/run_experiment.py:137; BASELINE.json configs[3]): ``items`` are dicts with
        latent (1,16,h/8,w/8), txt (1,T,4096), vec (1,768), concepts (1,C,4096) already on the device; they are
        dealt round-robin to ``n_streams`` HIP streams, each with its own activation set (the same throughput
        mode as generate_many_on_device).  Returns [(heat [1,C,s,s], cross [1,C,s,s]), ...], identical to
        encoding the items one by one."""
        batch = max(1, min(batch, _lib.ATTN_MAX_PROBLEMS // 2, _lib.MAX_SEGMENTS // 3))
        groups = [list(range(g0, min(g0 + batch, len(items)))) for g0 in range(0, len(items), batch)]
        n_streams = max(1, min(n_streams, len(groups)))
        while len(self._replicas) < n_streams:
            self._replicas.append(HipFluxDiT(self.params, self.device, weights=self.model.weights,
                                             precision=self.model.precision,
                                             residual_dtype=self.model.residual_dtype)
                                  .set_precision(self.model.precision, self.model.keep_bf16_layers))
        for r in self._replicas:
            r.capture_independent_image = self.model.capture_independent_image
        while len(self._streams) < n_streams:
            self._streams.append(torch.cuda.Stream(device=self.device))
        cur = torch.cuda.current_stream(self.device)
        self.model.materialize()  # shared fp8 weight images: built on `cur`, which every side stream waits on
        for st in self._streams[:n_streams]:
            st.wait_stream(cur)
        results = [None] * len(items)
        for gi, idx in enumerate(groups):   # ``batch`` images share every launch of a forward (see generate_many)
            slot = gi % n_streams
            with torch.cuda.stream(self._streams[slot]):
                def cat(key):
                    return torch.cat([items[i][key] for i in idx], 0) if len(idx) > 1 else items[idx[0]][key]
                latent = cat("latent").to(self.device, torch.bfloat16)
                con, con_ids, con_vec = sampling.concept_inputs(cat("concepts"), cat("vec"))
                ho, hc = self._encode_maps(self._replicas[slot], latent, cat("txt"), cat("vec"), con, con_ids, con_vec,
                                           layer_indices, num_samples, num_steps, noise_timestep, seed, noise=noise)
                for k, i in enumerate(idx):
                    results[i] = (ho[k:k + 1], hc[k:k + 1])
        for st in self._streams[:n_streams]:
            cur.wait_stream(st)
        for pair in results:
            for t in pair:
                t.record_stream(cur)
        return results
