"""FluxGenerator: host-side mirror of the reference's generator object
(``concept_attention/image_generator.py:64-205``) for the HIP path.

Same constructor arguments (``model_name, device, offload, attention_block_class, dit_class``) and the
same ``generate_image(width, height, num_steps, guidance, seed, prompt, concepts, ...)`` ->
``(image, concept_attention_dict)`` contract, where the dict holds the four vector stacks
``[steps, 19, 1, ...]`` that ``compute_heatmaps_from_vectors`` consumes.  What is NOT re-stated: the
HuggingFace downloads (`load_t5/load_clip/load_ae/hf_hub_download`, `:19-62`; unavailable offline) -- the
text encoder and autoencoder are injectable, with synthetic stand-ins by default -- and the
`model.cpu()` / `.to(device)` round trip of the 23.8 GB weights on every call (`:183,194`), which a
288 GB device does not need.
"""
from __future__ import annotations

import time

import torch

from . import sampling
from .flux_dit import HipFluxDiT, on_own_device
from .params import T5_TOKENS, configs


def load_flow_model(name: str, device="cuda", hf_download: bool = True, attention_block_class=None,
                    dit_class=HipFluxDiT, weights="synthetic", weight_seed: int = 0, params=None,
                    residual_dtype=torch.float32):
    """Counterpart of load_flow_model (image_generator.py:19-47): builds ``dit_class(params)`` and fills
    it from ``weights``: "synthetic", a flux1-*.safetensors path (env FLUX_SCHNELL / FLUX_DEV are honoured
    like flux/util.py:33,65), or a state dict.  Nothing is downloaded."""
    import os
    p = params if params is not None else configs[name]
    model = (dit_class(p, device, residual_dtype=residual_dtype) if dit_class is HipFluxDiT
             else dit_class(p, attention_block_class=attention_block_class))
    env = {"flux-schnell": "FLUX_SCHNELL", "flux-dev": "FLUX_DEV"}.get(name)
    if isinstance(weights, str) and weights == "synthetic" and env and os.getenv(env):
        weights = os.getenv(env)
    if isinstance(weights, str) and weights == "synthetic":
        model.weights.init_synthetic(weight_seed)
    elif isinstance(weights, str):
        from safetensors.torch import load_file
        model.load_state_dict(load_file(weights, device=str(device)), strict=False)
    elif weights is not None:
        model.load_state_dict(weights, strict=False)
    return model


class FluxGenerator:
    def __init__(self, model_name: str, device, offload: bool = False, attention_block_class=None,
                 dit_class=HipFluxDiT, weights="synthetic", weight_seed: int = 0, text_encoder=None,
                 autoencoder=None, params=None, n_text_tokens=None, residual_dtype=torch.float32):
        from .pipeline import SyntheticTextEncoder
        self.device = torch.device(device)
        self.offload = offload
        self.model_name = model_name
        self.is_schnell = model_name == "flux-schnell"
        self.params = params if params is not None else configs[model_name]
        self.model = load_flow_model(model_name, self.device, attention_block_class=attention_block_class,
                                     dit_class=dit_class, weights=weights, weight_seed=weight_seed,
                                     params=self.params, residual_dtype=residual_dtype)
        n_tok = n_text_tokens or T5_TOKENS.get(model_name, 256)
        enc = text_encoder or SyntheticTextEncoder(n_tok, self.params.context_in_dim, self.params.vec_in_dim,
                                                   self.device)
        self.text_encoder = enc
        self.t5, self.clip = enc.t5, enc.clip
        self.ae = autoencoder
        self.nsfw_classifier = None

    def embed(self, prompt: str, concepts):
        """prepare()'s text side + embed_concepts (flux/sampling.py:47-55, concept_attention/utils.py:6-33)."""
        txt, vec = self.t5(prompt), self.clip(prompt)
        con = torch.stack([self.t5(c)[0, 0, :] for c in concepts]).unsqueeze(0)
        con, con_ids, con_vec = sampling.concept_inputs(con, vec)
        return txt, vec, con, con_ids, con_vec

    def decode(self, x: torch.Tensor, height: int, width: int):
        """unpack + VAE decode + PIL (image_generator.py:189-204); without an autoencoder the unpacked
        latent is returned as a numpy array."""
        lat = sampling.unpack(x.float(), height, width)
        if self.ae is None:
            return lat[0].cpu().numpy()
        import PIL.Image
        img = self.ae.decode(lat.to(torch.float32)).clamp(-1, 1)[0].permute(1, 2, 0)
        return PIL.Image.fromarray((127.5 * (img + 1.0)).cpu().byte().numpy())

    @torch.no_grad()  # (the reference uses inference_mode; the resident workspace is reused across calls)
    @on_own_device
    def generate_image(self, width, height, num_steps, guidance, seed, prompt, concepts, init_image=None,
                       image2image_strength=0.0, add_sampling_metadata=True, restrict_clip_guidance=False,
                       joint_attention_kwargs=None, latent=None):
        """image_generator.py:87-205.  ``latent`` overrides get_noise (device RNG differs per platform)."""
        seed = int(seed)
        if seed == -1:
            seed = torch.Generator(device="cpu").seed()
        t0 = time.perf_counter()
        x = latent if latent is not None else sampling.get_noise(1, height, width, self.device, torch.bfloat16, seed)
        x = x.to(self.device, torch.bfloat16)
        timesteps = sampling.get_schedule(num_steps, x.shape[-1] * x.shape[-2] // 4, shift=(not self.is_schnell))
        if init_image is not None:  # image-to-image start (:121-158)
            t_idx = int((1 - image2image_strength) * num_steps)
            t = timesteps[t_idx]
            timesteps = timesteps[t_idx:]
            x = (t * x.float() + (1.0 - t) * init_image.to(self.device).float()).to(torch.bfloat16)
        txt, vec, con, con_ids, con_vec = self.embed("" if restrict_clip_guidance else prompt, concepts)
        if restrict_clip_guidance:
            txt = self.t5(prompt)
        inp = sampling.prepare_from_embeddings(x, txt, vec)
        x, _, concept_attention_dict = sampling.denoise(
            self.model, **inp, timesteps=timesteps, guidance=guidance, concepts=con, concept_ids=con_ids,
            concept_vec=con_vec, joint_attention_kwargs=joint_attention_kwargs)
        img = self.decode(x, height, width)
        self.last_seconds = time.perf_counter() - t0
        return img, concept_attention_dict
