"""layoutdit_amd: MI355X-native ViT (DiT / BEiT) encoder forward for LayoutDiT's document-layout detector.

Only the hot path ``hs = self.dit(x).hidden_states`` (ref src/layoutdit/modeling/dit_backbone.py:47) lives here:
hand-written gfx950 HIP kernels behind a C ABI (``include/ldit.h`` -> ``libldit_hip.so``) and the thin PyTorch-ROCm
host mirror of the reference's module surface.  Importing the package does not load the shared object; the first
compute call does, and fails loudly if it is missing (there is no CPU fallback).
"""
from .config import DiTConfig, vit_base, vit_large, vit_micro, vit_tiny  # noqa: F401


def __getattr__(name):
    if name == "DiTEncoder":
        from .modeling.dit_encoder import DiTEncoder
        return DiTEncoder
    if name == "DiTBackbone":
        from .modeling.dit_backbone import DiTBackbone
        return DiTBackbone
    if name == "DiTWithFPN":
        from .modeling.dit_fpn import DiTWithFPN
        return DiTWithFPN
    if name == "DetectorInputTransform":
        from .modeling.detector_input import DetectorInputTransform
        return DetectorInputTransform
    raise AttributeError(name)
