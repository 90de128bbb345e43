"""Seeded, machine-independent parameter draws for fixtures whose weights are too large to commit.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  numpy's PCG64 stream is specified to be identical on
every platform, so ``oracle/make_golden.py`` (which loads these draws INTO the reference's modules before
recording their outputs) and the tests (which rebuild the same state on the GPU box) agree bit for bit;
the fixture stores ``abs_sum`` to prove it.
"""
from __future__ import annotations

import numpy as np


def seeded_state(shapes: dict, seed: int) -> dict:
    """shapes: ordered {key: shape}.  2-D tensors: U(-1,1)/sqrt(fan_in); 1-D ``*.weight`` (LayerNorm gain):
    1 + 0.1 N(0,1); every bias: 0.05 N(0,1); ``*.weight_g`` (weight-norm gain, (out,1)): 0.577 (1 + 0.2 N(0,1)),
    i.e. the norm of a default-initialised row, perturbed."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, shp in shapes.items():
        if k.endswith("weight_g"):
            w = 0.577 * (1.0 + 0.2 * rng.standard_normal(shp))
        elif len(shp) == 2:
            w = rng.uniform(-1.0, 1.0, size=shp) / np.sqrt(shp[1])
        elif k.endswith(".weight"):
            w = 1.0 + 0.1 * rng.standard_normal(shp)
        else:
            w = 0.05 * rng.standard_normal(shp)
        out[k] = w.astype(np.float32)
    return out


def subsample(g: np.ndarray) -> np.ndarray:
    """What the large fixture keeps of a gradient: all of a vector, rows ::7 / columns ::5 of a matrix."""
    return g if g.ndim < 2 else np.ascontiguousarray(g[::7, ::5])
