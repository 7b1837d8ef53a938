"""Image-sharded zero-shot segmentation scoring (the loop of experiments/imagenet_segmentation/run_experiment.py
on the MI355X path): every rank encodes its share of the images, thresholds the target concept's heat map at its
mean and accumulates pixAcc / mIoU / mAP; one all_reduce of seven numbers at the end.

    python examples/segmentation_eval.py                      # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/segmentation_eval.py

No dataset can be fetched here, so the 'images' are seeded random latents and the 'labels' random blobs: the
scores are meaningless, the data flow is the real one."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from conceptattention_amd import ConceptAttentionFluxPipeline
from conceptattention_amd import distributed as D
from conceptattention_amd.segmentation import ConceptAttentionSegmentationModel, SegmentationScores, prepare_for_scoring

rank, world, local = D.init_from_env()
dev = torch.device(os.environ.get("CA_BENCH_DEVICE") or f"cuda:{local}")
torch.cuda.set_device(dev)
n_images = int(os.environ.get("N_IMAGES", 8))
pipe = ConceptAttentionFluxPipeline("flux-schnell", device=dev)
seg = ConceptAttentionSegmentationModel(pipe)
scores = SegmentationScores()
background = ["background", "floor", "grass", "tree", "sky"]
for j in D.shard_items(n_images, rank, world):
    g = torch.Generator().manual_seed(100 + j)
    latent = torch.randn(1, 16, 128, 128, generator=g)
    label = torch.nn.functional.interpolate(torch.rand(1, 1, 7, 7, generator=g), size=(224, 224)) [0, 0] > 0.5
    masks, coeffs, _ = seg(latent, target_concepts=["object"], concepts=["object"] + background,
                           captions=["a object"], layers=list(range(19)), num_samples=1, num_steps=4, noise_timestep=2)
    c, m = prepare_for_scoring(coeffs[0], masks[0], size=224)
    scores.update(m, c, label.numpy())
result = scores.all_reduce().result()
if rank == 0:
    print(f"images {result['n']}: pixAcc {result['pixAcc']:.4f}  mIoU {result['mIoU']:.4f}  mAP {result['mAP']:.4f}")
D.barrier()
if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()
