from .dit_backbone import DiTBackbone  # noqa: F401
from .dit_encoder import DiTEncoder, DiTEncoderOutput  # noqa: F401
