"""Inputs of the per-op known-answer tests, rebuilt from seeds (shared by make_golden.py and the tests so that the
large q/k/v tensors need not be committed; only the expected outputs are)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from layoutdit_amd import synth  # noqa: E402


def attn_inputs(n_tok: int, heads: int = 3, dim: int = 64, batch: int = 1):
    shape = (batch, n_tok, heads * dim)
    n = batch * n_tok * heads * dim
    q = synth.normal(12, n_tok, n).reshape(shape).astype(np.float32)
    k = synth.normal(13, n_tok, n).reshape(shape).astype(np.float32)
    v = synth.normal(14, n_tok, n).reshape(shape).astype(np.float32)
    q[0, 5] *= 12.0    # large logits on one query row
    k[0, 17] *= 9.0    # one dominant key
    return q, k, v


def ln_inputs(C: int = 768):
    rows = synth.normal(11, 1, 16 * C).reshape(16, C)
    rows[1] = 1000.0 + 1e-3 * rows[1]   # large mean, tiny variance (eps = 1e-12 matters)
    rows[2] *= 50.0                     # large-magnitude channels
    rows[3, :] = 0.25                   # (almost) constant row
    rows[3, 0] = 0.25000003
    rows[4, ::97] += 200.0              # outlier channels
    g = (1.0 + 0.1 * synth.normal(11, 2, C)).astype(np.float32)
    b = (0.1 * synth.normal(11, 3, C)).astype(np.float32)
    return rows.astype(np.float32), g, b
