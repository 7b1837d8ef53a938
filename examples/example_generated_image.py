"""Concept heat maps for a generated image -- the reference's example_generated_image.py on the MI355X path.

Without Flux/T5/CLIP/VAE checkpoints (none can be fetched here) the pipeline runs on seeded random-init weights
and synthetic text embeddings, so the pictures are noise; the call sequence, shapes and outputs are the real ones.
Pass weights="/path/to/flux1-schnell.safetensors" plus text_encoder= / autoencoder= objects to run the real model."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conceptattention_amd import ConceptAttentionFluxPipeline

pipeline = ConceptAttentionFluxPipeline(model_name="flux-schnell", device="cuda:0")

prompt = "A cat in a park on the grass by a tree"
concepts = ["cat", "grass", "sky", "tree"]

pipeline_output = pipeline.generate_image(prompt=prompt, concepts=concepts, width=1024, height=1024)

out_dir = sys.argv[1] if len(sys.argv) > 1 else "results"
os.makedirs(out_dir, exist_ok=True)
for concept, heatmap in zip(concepts, pipeline_output.concept_heatmaps):
    heatmap.save(os.path.join(out_dir, f"{concept}.png"))
for concept, heatmap in zip(concepts, pipeline_output.cross_attention_maps):
    heatmap.save(os.path.join(out_dir, f"cross_attention_{concept}.png"))
print("wrote", len(concepts) * 2, "heat maps to", out_dir)
