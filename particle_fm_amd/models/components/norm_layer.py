"""Drop-in for particle_fm/models/components/norm_layer.py::IterativeNormLayer (the ``use_normaliser=True`` pre-processing of
SetFlowMatchingLitModule, flow_matching_module.py:467-473, 514-518, 666-677).

Same constructor, buffers (``means``, ``vars``, ``n``, ``m2`` -> same state_dict) and methods (``fit``, ``forward``, ``reverse``,
``update``); the statistics and the mapping run in libpfm_hip.so (pfm_norm_update / pfm_norm_apply), on (rows, features) inputs
with an optional boolean row mask.  ``extra_dims`` (statistics shared over more axes) is not implemented; CPU tensors raise.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Union

import torch
import torch.nn as nn

from ... import _lib
from ...hip_ops import _ptr, _stream_ptr


class IterativeNormLayer(nn.Module):
    def __init__(self, inpt_dim: Union[torch.Tensor, tuple, int], means: Optional[torch.Tensor] = None,
                 vars: Optional[torch.Tensor] = None, n: int = 0, max_n: int = 5_00_000,
                 extra_dims: Union[tuple, int] = ()) -> None:
        super().__init__()
        if (means is None) ^ (vars is None):
            raise ValueError("Only one of 'means' and 'vars' is defined. Either both or neither must be defined")  # :49-53
        if isinstance(inpt_dim, int):
            inpt_dim = (inpt_dim,)
        if isinstance(extra_dims, int):
            extra_dims = [extra_dims]
        if len(tuple(extra_dims)) or len(tuple(inpt_dim)) != 1:
            raise NotImplementedError("IterativeNormLayer: the HIP path implements per-feature statistics of (.., features) inputs "
                                      "(inpt_dim = (features,), no extra_dims), the way SetFlowMatchingLitModule builds it")
        if int(inpt_dim[0]) > 16:
            raise NotImplementedError("IterativeNormLayer: at most 16 features")
        if isinstance(n, int):
            n = torch.tensor(n)
        self.extra_dims = []
        self.max_n = max_n
        self.inpt_dim = list(inpt_dim)
        self.stat_dim = [1] + list(inpt_dim)
        self.register_buffer("means", torch.zeros(self.stat_dim) if means is None else means)
        self.register_buffer("vars", torch.ones(self.stat_dim) if vars is None else vars)
        self.register_buffer("n", n)
        self.register_buffer("m2", torch.ones(self.stat_dim) if vars is None else vars)
        self.frozen = means is not None

    # ---- helpers ----------------------------------------------------------------------------------------------------
    def _rows(self, inpt: torch.Tensor, mask: Optional[torch.Tensor]):
        if not inpt.is_cuda:
            raise RuntimeError("IterativeNormLayer: the HIP path needs tensors on a ROCm device; there is no CPU fallback")
        F = self.inpt_dim[0]
        if inpt.shape[-1] != F:
            raise ValueError(f"input has {inpt.shape[-1]} features, the layer was built for {F}")
        x = inpt.to(torch.float32).contiguous()
        m = None
        if mask is not None:
            if tuple(mask.shape) != tuple(inpt.shape[:-1]):
                raise ValueError("mask must have the input's shape without the feature axis")
            m = mask.to(torch.float32).contiguous()
        for name in ("means", "vars", "m2", "n"):
            b = getattr(self, name)
            if b.device != inpt.device:
                raise RuntimeError(f"IterativeNormLayer.{name} lives on {b.device}, the input on {inpt.device}")
        return x, m, x.numel() // F, F

    def _stat_update(self, x, m, rows, F, max_n):
        rc = _lib.load().pfm_norm_update(_ptr(x), _ptr(m), ctypes.c_int64(rows), F, _ptr(self.n), _ptr(self.means), _ptr(self.vars),
                                         _ptr(self.m2), ctypes.c_int64(max_n), _stream_ptr(x.device))
        _lib.check(rc, "pfm_norm_update")

    def _map(self, x, m, rows, F, reverse):
        out = torch.empty_like(x)
        rc = _lib.load().pfm_norm_apply(_ptr(out), _ptr(x), _ptr(m), ctypes.c_int64(rows), F, _ptr(self.means), _ptr(self.vars),
                                        int(reverse), _stream_ptr(x.device))
        _lib.check(rc, "pfm_norm_apply")
        return out

    # ---- reference surface ------------------------------------------------------------------------------------------
    def fit(self, inpt: torch.Tensor, mask: Optional[torch.Tensor] = None, freeze: bool = True) -> None:
        """Set the stats given a population of data (:98-104)."""
        x, m, rows, F = self._rows(inpt, mask)
        self.n.zero_()
        self._stat_update(x, m, rows, F, torch.iinfo(torch.int64).max)
        self.frozen = freeze

    def update(self, inpt: torch.Tensor, mask: Optional[torch.Tensor] = None) -> None:
        """Update the running stats using a batch of data (:137-155).  The freeze-at-max_n test runs on the device (the kernel
        leaves the buffers alone once n >= max_n), so ``frozen`` stays a host-side hint and no launch waits on the counter."""
        x, m, rows, F = self._rows(inpt, mask)
        self._stat_update(x, m, rows, F, self.max_n)

    def forward(self, inpt: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        x, m, rows, F = self._rows(inpt, mask)
        with torch.no_grad():
            if not self.frozen and self.training:
                self._stat_update(x, m, rows, F, self.max_n)
            return self._map(x, m, rows, F, False).view_as(inpt)

    def reverse(self, inpt: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        x, m, rows, F = self._rows(inpt, mask)
        return self._map(x, m, rows, F, True).view_as(inpt)
