"""Tensor-level wrappers over include/pfm_epicw.h (EPiC at widths beyond the jet-resident kernel).  No CPU path."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .hip_ops import _dev_f32, _prep_common, _ptr, _stream_ptr, midpoint_grid
from .layout_wide import EpicWideLayout


def workspace(layout: EpicWideLayout, n_jets: int, device) -> torch.Tensor:
    lib = _lib.load()
    cache = layout.__dict__.setdefault("_ws", {})
    key = (n_jets, str(device))
    if key not in cache:
        n = lib.pfm_ew_workspace_floats(ctypes.byref(layout.desc), n_jets, 0)
        if n < 0:
            _lib.check(1, "pfm_ew_workspace_floats")
        cache.clear()
        cache[key] = torch.empty(n, device=device, dtype=torch.float32)
    return cache[key]


def ew_forward(layout: EpicWideLayout, blob, t, x, cond=None, mask=None) -> torch.Tensor:
    """v = EPiC(t, x, cond, mask).  t: (B,) one time per jet, or a single element for one shared time."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep_common(layout, blob, x, cond, mask)
    t = _dev_f32("t", t.reshape(-1), dev)
    if t.numel() not in (1, B):
        raise ValueError(f"t has {t.numel()} elements, expected 1 or {B}")
    v = torch.empty_like(x)
    rc = lib.pfm_ew_forward(ctypes.byref(layout.desc), _ptr(blob), _ptr(t), 1 if (t.numel() == B and B > 1) else 0, _ptr(x),
                            _ptr(cond), _ptr(mask), _ptr(v), B, _ptr(workspace(layout, B, dev)), _stream_ptr(dev))
    _lib.check(rc, "pfm_ew_forward")
    return v


def ew_sample_midpoint(layout: EpicWideLayout, blob, z, cond=None, mask=None, ode_steps: int = 100, premask: bool = True):
    lib = _lib.load()
    dev, B, blob, z, cond, mask = _prep_common(layout, blob, z, cond, mask)
    if ode_steps < 2:
        raise ValueError("ode_steps must be >= 2")
    ts, dts = midpoint_grid(ode_steps)
    ts, dts = ts.to(dev), dts.to(dev)
    out = torch.empty_like(z)
    state = torch.empty(2 * z.numel(), device=dev, dtype=torch.float32)
    rc = lib.pfm_ew_sample_midpoint(ctypes.byref(layout.desc), _ptr(blob), _ptr(ts), _ptr(dts), ode_steps - 1, _ptr(z),
                                    _ptr(cond), _ptr(mask), _ptr(out), B, int(bool(premask and mask is not None)),
                                    _ptr(state), _ptr(workspace(layout, B, dev)), _stream_ptr(dev))
    _lib.check(rc, "pfm_ew_sample_midpoint")
    return out
