"""Deterministic seeded weights shared by tests, golden generation and bench.py.

The trained checkpoints (model/regressor0.pt, model/classifier1.pt) are not available, so
parity and throughput are measured on weights drawn from numpy's frozen legacy stream
`RandomState(seed)`: keys visited in SORTED order, U(-b, b) with b = 1/sqrt(fan_in) for
matrices (PyG/torch Linear default scale; `lin_edge.weight [96,1]` gets b = 1) and b = 0.1
for vectors, times `scale`.  Works on any module exposing the reference's state_dict keys.
"""
import math

import numpy as np
import torch


def seeded_state_dict(shapes, seed: int, scale: float = 1.0):
    """shapes: mapping key -> shape (e.g. {k: tuple(v.shape) for k, v in sd.items()})."""
    rs = np.random.RandomState(seed)
    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        bound = 1.0 / math.sqrt(shape[-1]) if len(shape) >= 2 else 0.1
        w = rs.uniform(-bound, bound, size=shape).astype(np.float32) * np.float32(scale)
        out[key] = torch.from_numpy(w)
    return out


def load_seeded(module, seed: int, scale: float = 1.0):
    sd = module.state_dict()
    new = seeded_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed, scale)
    module.load_state_dict(new)
    return module
