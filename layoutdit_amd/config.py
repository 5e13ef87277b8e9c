"""Encoder geometry for the DiT / BEiT ViT encoder behind LayoutDiT's ``DiTBackbone``.

Mirrors the handful of ``BeitConfig`` fields the hot path reads
(transformers ``models/beit/configuration_beit.py:72-102``; DiT-base = BEiT-base with
``use_absolute_position_embeddings=True``, ``use_mask_token=True``) and the two attributes
``DiTBackbone.__init__`` reads from ``config``
(ref ``src/layoutdit/modeling/dit_backbone.py:33-36``: ``num_hidden_layers``, ``hidden_size``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class DiTConfig:
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    patch_size: int = 16
    image_size: int = 224          # side of the square grid the position table was built for
    num_channels: int = 3
    layer_norm_eps: float = 1e-12  # configuration_beit.py:81
    layer_scale_init_value: float = 0.1
    drop_path_rate: float = 0.1    # stochastic depth in train mode, layer i drops with rate * i / (L-1) (configuration_beit.py:90)
    output_hidden_states: bool = True
    # which hidden states the caller consumes; None -> DiTBackbone's [d/3, d/2, 2d/3, d]
    taps: List[int] = field(default_factory=list)

    def __post_init__(self):
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be a multiple of num_attention_heads")
        if not self.taps:
            d = self.num_hidden_layers
            # ref dit_backbone.py:33-34
            self.taps = [d // 3, d // 2, 2 * d // 3, d]

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def num_patches(self) -> int:
        g = self.image_size // self.patch_size
        return g * g

    def tokens(self, height: int, width: int) -> int:
        return (height // self.patch_size) * (width // self.patch_size) + 1

    def flops_per_image(self, height: int | None = None, width: int | None = None) -> int:
        """Algorithmic matmul FLOPs (2 x MAC), SURVEY.md 8(d): pooler, softmax, LN, GELU excluded."""
        height = height or self.image_size
        width = width or self.image_size
        P = (height // self.patch_size) * (width // self.patch_size)
        N = P + 1
        C, F, L = self.hidden_size, self.intermediate_size, self.num_hidden_layers
        k_patch = self.num_channels * self.patch_size * self.patch_size
        macs = P * k_patch * C + L * (4 * N * C * C + 2 * N * C * F + 2 * N * N * C)
        return 2 * macs


# Named geometries used by BASELINE.json's configs.
def vit_micro() -> DiTConfig:   # golden G0: bit-for-bit debuggable
    return DiTConfig(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=512,
                     image_size=64)


def vit_tiny() -> DiTConfig:
    return DiTConfig(hidden_size=192, num_hidden_layers=12, num_attention_heads=3, intermediate_size=768)


def vit_base() -> DiTConfig:
    return DiTConfig()


def vit_large(image_size: int = 224) -> DiTConfig:
    return DiTConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                     image_size=image_size)


GEOMETRIES = {"micro": vit_micro, "tiny": vit_tiny, "base": vit_base, "large": vit_large}
