"""Shared helpers for the test-suite (metrics, fingerprints, host-side position resampling)."""
import numpy as np


def rel_l2(a, b) -> float:
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_rel(a, b, floor: float = 1.0) -> float:
    """max |a-b| / max(|b|, floor): the element-wise gate of SURVEY.md 8(d) (floor 1)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def weight_fingerprint(weights) -> float:
    tot = 0.0
    for k in sorted(weights):
        tot += float(np.sum(weights[k].astype(np.float64) * (1.0 + (len(k) % 7))))
    return tot


def resample_pos(pos_table, grid: int, gh: int, gw: int) -> np.ndarray:
    """Bicubic (align_corners=False) resample of the patch part of a ``[1, 1+grid*grid, C]`` position table to
    ``gh x gw``; the arithmetic of TF:models/beit/modeling_beit.py:130-151 restated with the same torch op."""
    import torch
    import torch.nn.functional as F
    t = torch.from_numpy(np.ascontiguousarray(pos_table, dtype=np.float32)).reshape(1, -1, pos_table.shape[-1])
    cls, patch = t[:, :1], t[:, 1:]
    C = t.shape[-1]
    patch = patch.reshape(1, grid, grid, C).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False)
    patch = patch.permute(0, 2, 3, 1).reshape(1, gh * gw, C)
    return torch.cat((cls, patch), dim=1)[0].numpy()
