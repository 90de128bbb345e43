"""Small conditional MLP vector field of the fully-connected FM model (BASELINE config 1, "two moons").

Mirrors particle_fm/models/components/mlp.py:5-68 (``MLP``, ``small_cond_MLP_model``): four MLP stages that each
re-concatenate ``[t ; x ; cond]``.  Plain PyTorch on purpose: SURVEY.md §8 row a13 scopes this configuration as
CPU plumbing (no HIP); it exists so that the reference's notebook-style 2-D demo runs against this package.
"""
from __future__ import annotations

import torch
from torch import nn


class MLP(nn.Sequential):
    def __init__(self, in_features: int, out_features: int, hidden_features=(64, 64), activation: str = "ELU"):
        hidden = list(hidden_features)
        layers = []
        for a, b in zip([in_features] + hidden, hidden + [out_features]):
            layers.extend([nn.Linear(a, b), getattr(nn, activation)()])
        super().__init__(*layers[:-1])


class small_cond_MLP_model(nn.Module):
    def __init__(self, in_features: int, out_features: int, activation: str = "ELU", dim_t: int = 6, dim_cond: int = 1):
        super().__init__()
        e = dim_t + dim_cond
        self.mlp1 = MLP(in_features + e, 64, [64, 64], activation)
        self.mlp2 = MLP(64 + e, 256, [256, 256], activation)
        self.mlp3 = MLP(256 + e, 256, [256, 256], activation)
        self.mlp4 = MLP(256 + e, out_features, [64, 64], activation)

    def forward(self, t, x, cond):
        for stage in (self.mlp1, self.mlp2, self.mlp3, self.mlp4):
            x = stage(torch.cat([t, x, cond], dim=-1))
        return x
