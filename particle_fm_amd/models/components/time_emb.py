"""Host-side mirror of particle_fm/models/components/time_emb.py:25-96 (CosineEncoding).

On the HIP path the embedding is evaluated inside the kernels (csrc/epic_nfe.h::epic_time_embedding)
from the frequency table the layout fixes (layout.EpicLayout.default_freqs).  This module keeps the
reference's callable for code that wants the embedding itself; it uses the same table.
"""
from __future__ import annotations

import math

import torch


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
