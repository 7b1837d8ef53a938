"""conceptattention_amd: MI355X-native ConceptAttention hot path.

Public surface mirrors the reference package (`concept_attention/__init__.py:2`):
``from conceptattention_amd import ConceptAttentionFluxPipeline``.  Importing the package does not
touch the GPU; the HIP library is loaded on first use and its absence is a hard error.
"""
from .params import FluxParams, configs, tiny_params  # noqa: F401


def __getattr__(name):  # lazy: keep `import conceptattention_amd` light for host-only tools
    if name in ("ConceptAttentionFluxPipeline", "ConceptAttentionPipelineOutput"):
        from . import pipeline
        return getattr(pipeline, name)
    if name in ("HipFluxDiT", "FluxWeights", "HeatmapRequest"):
        from . import flux_dit
        return getattr(flux_dit, name)
    if name in ("FluxGenerator", "load_flow_model"):
        from . import image_generator
        return getattr(image_generator, name)
    if name == "compute_heatmaps_from_vectors":
        from .heatmaps import compute_heatmaps_from_vectors
        return compute_heatmaps_from_vectors
    raise AttributeError(name)
