"""Concept heat maps of an existing image -- the reference's example_encoded_image.py on the MI355X path.

The image is given as a VAE latent (1, 16, H/8, W/8); with an autoencoder injected into the pipeline a PIL image
works as in the reference.  One forward of the 19 double blocks per noise sample."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from conceptattention_amd import ConceptAttentionFluxPipeline

pipeline = ConceptAttentionFluxPipeline(model_name="flux-schnell", device="cuda:0")

latent = torch.randn(1, 16, 128, 128, generator=torch.Generator().manual_seed(0))  # stand-in for ae.encode(image)
concepts = ["dragon", "rock", "sky", "sun", "clouds"]

pipeline_output = pipeline.encode_image(image=latent, concepts=concepts, prompt="A fire breathing dragon.",
                                        width=1024, height=1024)

out_dir = sys.argv[1] if len(sys.argv) > 1 else "results"
os.makedirs(out_dir, exist_ok=True)
for concept, heatmap in zip(concepts, pipeline_output.concept_heatmaps):
    heatmap.save(os.path.join(out_dir, f"encoded_{concept}.png"))
print("wrote", len(concepts), "heat maps to", out_dir)
