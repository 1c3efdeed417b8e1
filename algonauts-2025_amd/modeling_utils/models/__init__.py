from .common import MlpConfig, SubjectLayers  # noqa: F401
from .transformer import TransformerEncoderConfig  # noqa: F401
