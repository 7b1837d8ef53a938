"""conceptattention_amd: MI355X-native ConceptAttention hot path."""
from .params import FluxParams, configs, tiny_params  # noqa: F401
