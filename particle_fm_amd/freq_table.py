"""Which frequency table the cosine time embedding multiplies t by.

``cosine_encoding`` (particle_fm/models/components/time_emb.py:89-96) uses ``torch.arange(T).exp()`` in fp32 and arguments up
to e^31 * pi: cos() of 1e13-sized numbers, so ONE ulp of a frequency changes that embedding component by O(1).  The fp32 ``exp``
is not the same on every host (measured: element 15 differs by 1 ulp between an Intel and an AMD EPYC build of the same torch
wheel), i.e. a checkpoint is tied to the table of the machine that trained it.  The kernels read the table from the weight blob;
this module decides what goes there:

  "float64-rounded"  (default) exp in float64, rounded once to fp32: the same on every host
  "torch"            ``torch.arange(T).exp()`` as THIS host computes it in fp32: what the reference would use here
  a tensor (T,)      the table a checkpoint was trained with (e.g. recorded on the training host)

Only t_emb="cosine" has such a table (sincos uses the module's ``frequencies`` buffer, exact powers of two times pi).
"""
from __future__ import annotations

from typing import Optional, Union

import torch

FreqSpec = Union[None, str, torch.Tensor]


def resolve_freq_table(spec: FreqSpec, t_dim: int, t_emb: str = "cosine") -> Optional[torch.Tensor]:
    """None = let the layout use its default (float64-rounded / the sincos buffer); otherwise the fp32 table (T,)."""
    if spec is None or t_emb != "cosine":
        return None
    if isinstance(spec, str):
        if spec == "float64-rounded":
            return None
        if spec == "torch":
            return torch.arange(t_dim).exp()  # fp32 on this host, exactly time_emb.py:90
        raise ValueError(f"freq_table={spec!r}: expected 'float64-rounded', 'torch' or a tensor of {t_dim} frequencies")
    t = torch.as_tensor(spec, dtype=torch.float32).detach().reshape(-1).cpu()
    if t.numel() != t_dim:
        raise ValueError(f"freq_table has {t.numel()} entries, the time embedding has {t_dim}")
    return t.clone()


class FreqTableMixin:
    """``set_freq_table`` for the network classes: the table is picked up the next time the weights are packed."""

    freq_table: FreqSpec = None
    _freq_version: int = 0

    def set_freq_table(self, spec: FreqSpec = "float64-rounded") -> None:
        t_dim = 2 * int(getattr(self, "frequencies"))
        resolve_freq_table(spec, t_dim, getattr(self, "t_emb", "cosine"))  # validate now
        self.freq_table = spec
        self._freq_version += 1  # packed-weight caches keyed on it (engine.FusedFMTrainer) start over

    def freq_tensor(self) -> Optional[torch.Tensor]:
        return resolve_freq_table(self.freq_table, 2 * int(getattr(self, "frequencies")), getattr(self, "t_emb", "cosine"))
