"""data_utils (MI355X build): only the pieces of the reference package that sit on the
trimodal-encode hot path -- the `SegmentData` batch type (dataloader.py) and the
feature-cache layer grouping (features/)."""
from . import dataloader  # noqa: F401
from .dataloader import SegmentData  # noqa: F401
