"""BEiT ``state_dict`` key layouts.

The reference pins ``transformers==4.49.0`` (ref ``uv.lock:1771-1772``), whose ``BeitModel`` names its tensors
``encoder.layer.{i}.attention.attention.query.weight`` etc.; transformers 5.x renamed them
(``layers.{i}.attention.q_proj.weight`` ..., map at transformers ``conversion_mapping.py:338-346``).  Checkpoints the
reference writes (ref ``src/layoutdit/modeling/model.py:90-121``) additionally carry the detector prefix
``backbone.backbone.dit.`` (or ``model.backbone.backbone.dit.``).  The encoder module owns parameters under the 4.49
names and accepts all of these on load.
"""
from __future__ import annotations

import re
from typing import Dict, Mapping

_V5_TO_V4 = [
    (r"^layers\.(\d+)\.attention\.q_proj\.", r"encoder.layer.\1.attention.attention.query."),
    (r"^layers\.(\d+)\.attention\.k_proj\.", r"encoder.layer.\1.attention.attention.key."),
    (r"^layers\.(\d+)\.attention\.v_proj\.", r"encoder.layer.\1.attention.attention.value."),
    (r"^layers\.(\d+)\.attention\.o_proj\.", r"encoder.layer.\1.attention.output.dense."),
    (r"^layers\.(\d+)\.mlp\.fc1\.", r"encoder.layer.\1.intermediate.dense."),
    (r"^layers\.(\d+)\.mlp\.fc2\.", r"encoder.layer.\1.output.dense."),
    (r"^layers\.(\d+)\.", r"encoder.layer.\1."),
]

_V4_TO_V5 = [
    (r"^encoder\.layer\.(\d+)\.attention\.attention\.query\.", r"layers.\1.attention.q_proj."),
    (r"^encoder\.layer\.(\d+)\.attention\.attention\.key\.", r"layers.\1.attention.k_proj."),
    (r"^encoder\.layer\.(\d+)\.attention\.attention\.value\.", r"layers.\1.attention.v_proj."),
    (r"^encoder\.layer\.(\d+)\.attention\.output\.dense\.", r"layers.\1.attention.o_proj."),
    (r"^encoder\.layer\.(\d+)\.intermediate\.dense\.", r"layers.\1.mlp.fc1."),
    (r"^encoder\.layer\.(\d+)\.output\.dense\.", r"layers.\1.mlp.fc2."),
    (r"^encoder\.layer\.(\d+)\.", r"layers.\1."),
]

_PREFIXES = ("model.backbone.backbone.dit.", "backbone.backbone.dit.", "backbone.dit.", "dit.", "beit.")


def _apply(key: str, rules) -> str:
    for pat, rep in rules:
        new, n = re.subn(pat, rep, key)
        if n:
            return new
    return key


def to_v4(key: str) -> str:
    """Any accepted spelling -> the transformers-4.49 key the module owns."""
    for p in _PREFIXES:
        if key.startswith(p):
            key = key[len(p):]
            break
    return _apply(key, _V5_TO_V4)


def to_v5(key: str) -> str:
    return _apply(key, _V4_TO_V5)


def remap_state_dict(sd: Mapping[str, object], target: str = "v4") -> Dict[str, object]:
    f = to_v4 if target == "v4" else (lambda k: to_v5(to_v4(k)))
    return {f(k): v for k, v in sd.items()}
