"""The producer of ``x``: the detector's input transform as LayoutDiT configures it.

The reference builds ``FasterRCNN(backbone, ..., min_size=224, max_size=224, fixed_size=(224, 224), image_mean=(0.5,)*3,
image_std=(0.5,)*3)`` (ref ``src/layoutdit/modeling/model.py:45-55``); torchvision's ``GeneralizedRCNNTransform`` then turns
the ``List[Tensor[3, h, w]]`` that ``LayoutDetectionModel.forward(images, targets)`` receives (ref ``model.py:87-88``; fp32 in
[0, 1] from the evaluator, fp16 from the trainer, ref ``trainer.py:153-155``) into the NCHW batch the backbone sees:
per image ``(img - mean) / std``, bilinear resize (``align_corners=False``, no antialias) to the fixed size, stack; boxes
of the targets are scaled by the same ratios, and detections are scaled back afterwards.

``torchvision`` is not available offline, so this mirrors that contract from its documented behaviour;
the pixel arithmetic is ONE launch of ``ldit_preprocess_f32`` / ``_f16`` for the whole ragged list (no per-image host loop,
no intermediate normalised copy).  Parity is pinned against ``torch.nn.functional.interpolate`` only ("parity unpinned" with
respect to torchvision itself).  Box arithmetic is a handful of scalars per image and stays in torch.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import ops


class ImageList:
    """Same two fields as ``torchvision.models.detection.image_list.ImageList``."""

    def __init__(self, tensors: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, device) -> "ImageList":
        return ImageList(self.tensors.to(device), self.image_sizes)


def resize_boxes(boxes: torch.Tensor, original_size: Sequence[int], new_size: Sequence[int]) -> torch.Tensor:
    """``xyxy`` boxes of an ``original_size = (h, w)`` image in the coordinates of its ``new_size`` resize."""
    rh, rw = float(new_size[0]) / float(original_size[0]), float(new_size[1]) / float(original_size[1])
    scale = torch.tensor([rw, rh, rw, rh], dtype=boxes.dtype, device=boxes.device)
    return boxes * scale


class DetectorInputTransform(nn.Module):
    def __init__(self, fixed_size: Tuple[int, int] = (224, 224), image_mean: Sequence[float] = (0.5, 0.5, 0.5),
                 image_std: Sequence[float] = (0.5, 0.5, 0.5), size_divisible: int = 32):
        """``fixed_size`` is ``(width, height)`` as in torchvision (it interpolates to ``size=(fixed_size[1], fixed_size[0])``)."""
        super().__init__()
        if len(set(image_mean)) != 1 or len(set(image_std)) != 1:
            raise NotImplementedError("per-channel mean / std: the kernel takes one mean and one std (LayoutDiT uses 0.5 / 0.5)")
        self.out_h, self.out_w = int(fixed_size[1]), int(fixed_size[0])
        if self.out_h % size_divisible or self.out_w % size_divisible:
            raise NotImplementedError(f"fixed size {self.out_h}x{self.out_w} is not a multiple of {size_divisible}: batching "
                                      "would pad (LayoutDiT's 224x224 does not)")
        self.mean, self.std = float(image_mean[0]), float(image_std[0])

    def forward(self, images: List[torch.Tensor], targets: Optional[List[Dict[str, torch.Tensor]]] = None):
        if len(images) == 0:
            raise ValueError("empty image list")
        for img in images:
            if img.dim() != 3:
                raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], got {tuple(img.shape)}")
        if targets is not None and len(targets) != len(images):
            raise ValueError("one target per image")
        batch = ops.preprocess([img.contiguous() for img in images], size=(self.out_h, self.out_w), mean=self.mean, std=self.std)
        out_targets = None
        if targets is not None:
            out_targets = []
            for img, t in zip(images, targets):
                t = dict(t)
                if "boxes" in t:
                    t["boxes"] = resize_boxes(t["boxes"], img.shape[-2:], (self.out_h, self.out_w))
                out_targets.append(t)
        return ImageList(batch, [(self.out_h, self.out_w)] * len(images)), out_targets

    def encode(self, encoder, images: List[torch.Tensor], taps: Optional[Sequence[int]] = None):
        """``encoder(self(images)[0].tensors)`` in eval mode with the transform FUSED into the encoder's patch-embedding load
        (``DiTEncoder.forward_image_list`` -> ``ldit_vit_forward_images``): same taps bit for bit, and the bf16 /
        fp8 / split-fp32 builds skip the fp32 ``[B, 3, H, W]`` batch altogether.  For callers that own the loop; torchvision's
        ``GeneralizedRCNN.forward`` hands the backbone the already transformed batch, which stays the two-step path."""
        return encoder.forward_image_list([img.contiguous() for img in images], size=(self.out_h, self.out_w), mean=self.mean,
                                          std=self.std, taps=taps)

    def postprocess(self, result: List[Dict[str, torch.Tensor]], image_shapes: List[Tuple[int, int]],
                    original_image_sizes: List[Tuple[int, int]]) -> List[Dict[str, torch.Tensor]]:
        """Detections back in the coordinates of the original images (what the evaluator reads, ref evaluator.py:237-258)."""
        for pred, shape, orig in zip(result, image_shapes, original_image_sizes):
            if "boxes" in pred:
                pred["boxes"] = resize_boxes(pred["boxes"], shape, orig)
        return result
