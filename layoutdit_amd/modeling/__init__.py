from .dit_backbone import DiTBackbone  # noqa: F401
from .dit_encoder import DiTEncoder, DiTEncoderOutput  # noqa: F401
from .dit_fpn import DiTWithFPN  # noqa: F401
from .detector_input import DetectorInputTransform, ImageList  # noqa: F401
