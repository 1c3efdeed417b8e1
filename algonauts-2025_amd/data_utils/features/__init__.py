"""Feature-cache helpers on the hot path (layer grouping of cached extractor states)."""
from .layers import aggregate_layers  # noqa: F401
