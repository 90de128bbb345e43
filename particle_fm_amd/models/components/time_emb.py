"""Host-side mirror of particle_fm/models/components/time_emb.py:25-96 (CosineEncoding).

On the HIP path the embedding is evaluated inside the kernels (csrc/epic_nfe.h::epic_time_embedding)
from the frequency table the layout fixes (layout.EpicLayout.default_freqs).  This module keeps the
reference's callable for code that wants the embedding itself; it uses the same table.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class GaussianFourierProjection(nn.Module):
    """time_emb.py:9-22: Gaussian random features of the time, [sin(2 pi W t) ; cos(2 pi W t)], W a frozen parameter drawn
    N(0, scale^2) (in state_dict as ``W``).  Front of the t_emb="gaussian" embedding network of CNF (flow_matching_module.py:178-181):
    a per-jet vector of hidden_dim numbers, O(B * hidden) work -- host-side torch ops on the device, not a kernel."""

    def __init__(self, embed_dim, scale=30.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embed_dim // 2) * scale, requires_grad=False)

    def forward(self, x):
        x_proj = x[..., None] * self.W[None, ...] * 2 * math.pi
        return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


def exp_frequencies(outp_dim: int) -> torch.Tensor:
    """exp(0..D-1) correctly rounded to fp32 (time_emb.py:90; see EpicLayout.default_freqs)."""
    return torch.arange(outp_dim, dtype=torch.float64).exp().to(torch.float32)


def cosine_encoding(x: torch.Tensor, outp_dim: int = 32, min_value: float = 0.0, max_value: float = 1.0,
                    frequency_scaling: str = "exponential") -> torch.Tensor:
    if x.shape[-1] != 1 or x.dim() == 1:
        x = x.unsqueeze(-1)
    if frequency_scaling == "exponential":
        freqs = exp_frequencies(outp_dim).to(x.device)
    elif frequency_scaling == "linear":
        freqs = torch.arange(1, outp_dim + 1, device=x.device)
    else:
        raise RuntimeError(f"Unrecognised frequency scaling: {frequency_scaling}")
    return torch.cos((x + min_value) * freqs * math.pi / (max_value + min_value))


class CosineEncoding:
    def __init__(self, outp_dim: int = 32, min_value: float = 0.0, max_value: float = 1.0,
                 frequency_scaling: str = "exponential") -> None:
        self.outp_dim = outp_dim
        self.min_value = min_value
        self.max_value = max_value
        self.frequency_scaling = frequency_scaling

    def __call__(self, inpt: torch.Tensor) -> torch.Tensor:
        return cosine_encoding(inpt, self.outp_dim, self.min_value, self.max_value, self.frequency_scaling)
