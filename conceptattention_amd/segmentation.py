"""Zero-shot segmentation on top of the concept heat maps, and the scores the reference's evaluation
harness reports for it.

What is mirrored
  * ``add_noise_to_image``        concept_attention/segmentation.py:85-113
  * mean-threshold masks           concept_attention/segmentation.py:55-81 (``SegmentationAbstractClass.__call__``)
  * ``batch_pix_accuracy`` / ``batch_intersection_union`` / ``get_ap_scores``
                                   concept_attention/utils.py:48-108
  * the per-image scoring + running pixAcc / mIoU / mAP of
                                   experiments/imagenet_segmentation/run_experiment.py:166-235
The heat maps themselves come from ``ConceptAttentionFluxPipeline.encode_image`` (one forward of the 19
double blocks on the HIP path); everything in this file is small host-side arithmetic on (C, side, side)
maps.  The image loop of the harness (`run_experiment.py:137`) is the natural multi-GPU shard: every rank
scores its own images and ``SegmentationScores.all_reduce`` sums the five counters once at the end.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch

from . import sampling


# ------------------------------------------------------------------------------------------ noise
def add_noise_to_image(encoded_image: torch.Tensor, num_steps: int = 50, noise_timestep: int = 49, seed: int = 63,
                       width: int = 1024, height: int = 1024, device="cuda", is_schnell: bool = True,
                       noise: torch.Tensor | None = None):
    """x = t*noise + (1-t)*latent at schedule position ``noise_timestep``; returns (x, remaining timesteps).
    ``noise`` overrides get_noise (device RNG streams differ between platforms)."""
    x = noise if noise is not None else sampling.get_noise(1, height, width, device, torch.bfloat16, seed)
    timesteps = sampling.get_schedule(num_steps, x.shape[-1] * x.shape[-2] // 4, shift=(not is_schnell))
    t = timesteps[noise_timestep]
    x = t * x + (1.0 - t) * encoded_image.to(x.dtype)
    return x, timesteps[noise_timestep:]


# ------------------------------------------------------------------------------------------ masks
def mean_threshold_masks(coefficients: torch.Tensor) -> torch.Tensor:
    """(C, h, w) coefficients -> bool masks, each concept thresholded at its own spatial mean."""
    return coefficients > coefficients.mean(dim=(1, 2), keepdim=True)


def target_mask(coefficients: torch.Tensor, index: int, mean_value_threshold: bool = True) -> torch.Tensor:
    c = coefficients[index]
    return c > (c.mean() if mean_value_threshold else 0.0)


def _resize_nearest(x: torch.Tensor, size) -> torch.Tensor:
    return torch.nn.functional.interpolate(x[None, None].float(), size=size, mode="nearest")[0, 0]


def prepare_for_scoring(coefficients, mask, size: int = 224, downscale_for_eval: bool = False):
    """min-max rescale the target concept's map, optional 14x14 round trip, nearest resize of map and mask
    to the label resolution (run_experiment.py:177-204)."""
    c = torch.as_tensor(np.asarray(coefficients), dtype=torch.float32)
    c = (c - c.min()) / (c.max() - c.min())
    if downscale_for_eval:
        c = _resize_nearest(c, (14, 14))
    c = _resize_nearest(c, (size, size))
    m = _resize_nearest(torch.as_tensor(np.asarray(mask), dtype=torch.float32), (size, size))
    return c, m


# ------------------------------------------------------------------------------------------ metrics
def _np(x) -> np.ndarray:
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def batch_pix_accuracy(predict, target):
    """(correct, labelled) pixel counts; labels < 0 are unlabelled."""
    p, t = _np(predict) + 1, _np(target) + 1
    labelled = int(np.sum(t > 0))
    correct = int(np.sum((p == t) & (t > 0)))
    assert correct <= labelled, "Correct area should be smaller than Labeled"
    return correct, labelled


def batch_intersection_union(predict, target, nclass: int):
    """per-class (intersection, union) areas, classes 0..nclass-1, labels < 0 ignored."""
    p, t = _np(predict) + 1, _np(target) + 1
    p = p * (t > 0).astype(p.dtype)
    hit = p * (p == t)
    rng = (1, nclass)
    inter, _ = np.histogram(hit, bins=nclass, range=rng)
    a_p, _ = np.histogram(p, bins=nclass, range=rng)
    a_t, _ = np.histogram(t, bins=nclass, range=rng)
    union = a_p + a_t - inter
    assert (inter <= union).all(), "Intersection area should be smaller than Union area"
    return inter, union


def average_precision(y_true: np.ndarray, y_score: np.ndarray) -> float:
    """AP = sum_n (R_n - R_{n-1}) P_n over the distinct score thresholds, descending (the definition
    sklearn.metrics.average_precision_score uses, which utils.py:63 calls)."""
    y_true = np.asarray(y_true).astype(np.float64).ravel()
    y_score = np.asarray(y_score).astype(np.float64).ravel()
    n_pos = y_true.sum()
    if y_true.size == 0 or n_pos == 0:
        return float("nan") if y_true.size == 0 else 0.0
    order = np.argsort(-y_score, kind="stable")
    s, y = y_score[order], y_true[order]
    last_of_run = np.r_[np.nonzero(np.diff(s))[0], s.size - 1]
    tp = np.cumsum(y)[last_of_run]
    precision = tp / (last_of_run + 1.0)
    recall = tp / n_pos
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def get_ap_scores(predict, target, ignore_index: int = -1):
    """per-sample AP of (K, h, w) class scores against an (h, w) integer label map."""
    out = []
    for pred, tgt in zip(predict, target):
        pred, tgt = _np(pred), _np(tgt)
        k = pred.shape[0]
        lab = np.broadcast_to(tgt[None], pred.shape)
        onehot = (np.arange(k).reshape(k, *([1] * tgt.ndim)) == np.clip(tgt, 0, None).astype(np.int64)[None])
        keep = lab.reshape(-1) != ignore_index
        score = np.nan_to_num(pred.reshape(-1))[keep]
        out.append(float(np.nan_to_num(average_precision(onehot.reshape(-1)[keep], score))))
    return out


@dataclass
class SegmentationScores:
    """Running pixAcc / mIoU / mAP over images (run_experiment.py:134-232)."""
    correct: float = 0.0
    labelled: float = 0.0
    inter: np.ndarray = field(default_factory=lambda: np.zeros(2))
    union: np.ndarray = field(default_factory=lambda: np.zeros(2))
    ap_sum: float = 0.0
    n: int = 0

    def update(self, mask, coefficients, labels) -> dict:
        """mask, coefficients: (h, w) at the label resolution (see prepare_for_scoring); labels: (h, w) bool."""
        m = torch.as_tensor(_np(mask), dtype=torch.float32)
        y = torch.as_tensor(_np(labels).astype(np.float32))
        c = torch.as_tensor(_np(coefficients), dtype=torch.float32)
        m2, y2 = torch.stack((1 - m, m)), torch.stack((1 - y, y))
        cor, lab = batch_pix_accuracy(m2, y2)
        inter, union = batch_intersection_union(m2, y2, nclass=2)
        ap = get_ap_scores(torch.stack((1 - c, c))[None], y[None])[0]
        self.correct += cor
        self.labelled += lab
        self.inter = self.inter + inter
        self.union = self.union + union
        self.ap_sum += ap
        self.n += 1
        return {"correct": cor, "labelled": lab, "inter": inter, "union": union, "ap": ap}

    def all_reduce(self) -> "SegmentationScores":
        """Sum the counters over ranks (no-op without an initialised process group)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .distributed import allreduce_sum_
            v = torch.tensor([self.correct, self.labelled, *self.inter, *self.union, self.ap_sum, float(self.n)],
                             dtype=torch.float64)
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else v.device
            v = allreduce_sum_(v.to(dev)).cpu()
            self.correct, self.labelled = float(v[0]), float(v[1])
            self.inter, self.union = v[2:4].numpy().copy(), v[4:6].numpy().copy()
            self.ap_sum, self.n = float(v[6]), int(round(float(v[7])))
        return self

    def result(self) -> dict:
        eps = np.spacing(1, dtype=np.float64)
        iou = self.inter / (eps + self.union)
        return {"pixAcc": float(self.correct / (eps + self.labelled)), "mIoU": float(iou.mean()),
                "mAP": float(self.ap_sum / max(self.n, 1)), "n": self.n}


# ------------------------------------------------------------------------------------------ model wrapper
class ConceptAttentionSegmentationModel:
    """Callable with the reference's segmentation-model contract (segmentation.py:34-81):
    ``model(images, target_concepts, concepts, captions, mean_value_threshold=True, joint_attention_kwargs=None,
    apply_blur=False, **kwargs) -> (all_masks, all_coefficients, reconstructed_images)``.
    ``images`` are latents (1,16,h/8,w/8) or, with an autoencoder injected into the pipeline, PIL images."""

    def __init__(self, pipeline):
        self.pipeline = pipeline

    @torch.no_grad()
    def segment_individual_image(self, image, concepts, caption, layers=list(range(15, 19)), num_samples: int = 1,
                                 num_steps: int = 4, noise_timestep: int = 2, seed: int = 0, height: int = 1024,
                                 width: int = 1024, target_space: str = "output", joint_attention_kwargs=None,
                                 **_unused):
        out = self.pipeline.encode_image(image, concepts, prompt=caption, width=width, height=height,
                                         layer_indices=layers, num_samples=num_samples, num_steps=num_steps,
                                         noise_timestep=noise_timestep, seed=seed, return_pil_heatmaps=False,
                                         joint_attention_kwargs=joint_attention_kwargs)
        maps = out.concept_heatmaps if target_space == "output" else out.cross_attention_maps
        return torch.as_tensor(np.asarray(maps), dtype=torch.float32), None

    def __call__(self, images, target_concepts, concepts, captions, mean_value_threshold: bool = True,
                 joint_attention_kwargs=None, apply_blur: bool = False, **kwargs):
        if not isinstance(images, (list, tuple)):
            images = [images]
        all_masks, all_coefficients, reconstructed = [], [], []
        for i, image in enumerate(images):
            coeff, recon = self.segment_individual_image(image, concepts, captions[i],
                                                         joint_attention_kwargs=joint_attention_kwargs, **kwargs)
            if apply_blur:
                coeff = gaussian_blur3(coeff)
            if target_concepts is None:
                all_masks.append(mean_threshold_masks(coeff))
                all_coefficients.append(coeff)
            else:
                k = concepts.index(target_concepts[i])
                all_masks.append(target_mask(coeff, k, mean_value_threshold).numpy())
                all_coefficients.append(coeff[k].numpy())
            reconstructed.append(recon)
        return all_masks, all_coefficients, reconstructed


def gaussian_blur3(x: torch.Tensor, sigma: float = 1.0) -> torch.Tensor:
    """3x3 Gaussian blur with reflect padding of a (C, h, w) map (torchvision's gaussian_blur(kernel_size=3,
    sigma=1.0), used at segmentation.py:61)."""
    k = torch.exp(-0.5 * (torch.tensor([-1.0, 0.0, 1.0]) / sigma) ** 2)
    k = (k / k.sum()).to(x.dtype)
    w = (k[:, None] * k[None, :])[None, None]
    xp = torch.nn.functional.pad(x[:, None], (1, 1, 1, 1), mode="reflect")
    return torch.nn.functional.conv2d(xp, w)[:, 0]
