"""Tensor-level wrappers over the cross-attention part of the C ABI (include/pfm_ca.h).

PyTorch owns device memory and the stream; every number comes out of libpfm_hip.so.  No CPU path."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .hip_ops import _dev_f32, _ptr, _stream_ptr, midpoint_grid, rk_grid, rk_tableau
from .hip_ops_tf import _KINDS, _prep, _temb_table, _time_arg, temb_given
from .layout_ca import CaLayout


def ca_backward_dtemb(layout, blob, B: int, dev) -> torch.Tensor:
    """d loss / d temb (B, T) of the loss backward that has just run for this layout and batch size (pfm_ca_backward_dtemb reads that
    backward's scratch: call it right behind ca_fm_loss_backward, same stream)."""
    lib = _lib.load()
    scratch = layout.__dict__["_bscratch"][(B, str(dev))]
    out = torch.empty(B, layout.cfg.t_dim, device=dev, dtype=torch.float32)
    rc = lib.pfm_ca_backward_dtemb(ctypes.byref(layout.desc), _ptr(blob), _ptr(scratch), B, _ptr(out), _stream_ptr(dev))
    _lib.check(rc, "pfm_ca_backward_dtemb")
    return out


def workspace(layout: CaLayout, n_jets: int, device, train: bool = False) -> torch.Tensor:
    """Activation workspace; cached per (n_jets, train) on the layout (the kernels fully overwrite what they read)."""
    lib = _lib.load()
    if train:
        # The loss forward hands this tensor to its autograd node as the saved activations: it must be private to the call.  (A
        # cached one would be overwritten by a second forward of the same batch size before the first backward -- two
        # micro-batches summed before .backward(), a no_grad validation loss between a forward and its backward -- and the
        # backward would silently return wrong gradients.)  torch's caching allocator hands the block back in steady state.
        n = lib.pfm_ca_workspace_floats(ctypes.byref(layout.desc), n_jets, 1)
        if n < 0:
            _lib.check(1, "pfm_ca_workspace_floats")
        return torch.empty(n, device=device, dtype=torch.float32)
    cache = layout.__dict__.setdefault("_ws", {})
    # one workspace per stream: samples queued on two streams may run at the same time
    key = (n_jets, bool(train), str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in cache:
        n = lib.pfm_ca_workspace_floats(ctypes.byref(layout.desc), n_jets, int(train))
        if n < 0:
            _lib.check(1, "pfm_ca_workspace_floats")
        for k in [k for k in cache if k[1] == key[1] and k[2] == key[2] and k[0] != key[0]]:
            del cache[k]
        cache[key] = torch.empty(n, device=device, dtype=torch.float32)
    return cache[key]


def ca_forward(layout: CaLayout, blob, t, x, cond=None, mask=None) -> torch.Tensor:
    """v = FullCrossAttentionEncoder(t, x, cond, mask).  t: (B,) one time per jet, or 0-dim / (1,) for one shared time."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep(layout, blob, x, cond, mask)
    t, t_per_jet = _time_arg(layout, t, B, dev)
    v = torch.empty_like(x)
    ws = workspace(layout, B, dev)
    rc = lib.pfm_ca_forward(ctypes.byref(layout.desc), _ptr(blob), _ptr(t), t_per_jet, _ptr(x),
                            _ptr(cond), _ptr(mask), _ptr(v), B, _ptr(ws), _stream_ptr(dev))
    _lib.check(rc, "pfm_ca_forward")
    return v


def ca_sample_midpoint(layout: CaLayout, blob, z, cond=None, mask=None, ode_steps: int = 100,
                       premask: bool = True, temb_fn=None) -> torch.Tensor:
    """x(0) from x(1) = z (*mask) by ode_steps-1 explicit-midpoint intervals (2 NFE each), all launches queued on
    the current stream without a host sync."""
    lib = _lib.load()
    dev, B, blob, z, cond, mask = _prep(layout, blob, z, cond, mask)
    if ode_steps < 2:
        raise ValueError("ode_steps must be >= 2")
    ts, dts = midpoint_grid(ode_steps, dev)
    if temb_given(layout):  # the table of embeddings replaces the time grid (PFM_*_F_TEMB_GIVEN)
        ts = _temb_table(temb_fn, ts, dev)
    out = torch.empty_like(z)
    state = torch.empty(2 * z.numel(), device=dev, dtype=torch.float32)
    ws = workspace(layout, B, dev)
    rc = lib.pfm_ca_sample_midpoint(ctypes.byref(layout.desc), _ptr(blob), _ptr(ts), _ptr(dts), ode_steps - 1, _ptr(z),
                                    _ptr(cond), _ptr(mask), _ptr(out), B, int(bool(premask and mask is not None)),
                                    _ptr(state), _ptr(ws), _stream_ptr(dev))
    _lib.check(rc, "pfm_ca_sample_midpoint")
    return out


def ca_sample_rk(layout: CaLayout, blob, z, cond=None, mask=None, ode_steps: int = 100, solver: str = "rk4",
                 premask: bool = True, t0: float = 1.0, t1: float = 0.0, temb_fn=None) -> torch.Tensor:
    """x(t1) from x(t0) = z (*mask) with the fixed-step explicit Runge-Kutta scheme ``solver`` ("euler", "midpoint", "rk4" =
    torchdyn's 3/8 rule) over linspace(t0, t1, ode_steps); all launches queued on the current stream."""
    lib = _lib.load()
    dev, B, blob, z, cond, mask = _prep(layout, blob, z, cond, mask)
    if ode_steps < 2:
        raise ValueError("ode_steps must be >= 2")
    tab = rk_tableau(solver)
    ts, dts = rk_grid(ode_steps, solver, t0, t1)
    ts, dts = ts.to(dev), dts.to(dev)
    if temb_given(layout):
        ts = _temb_table(temb_fn, ts, dev)
    out = torch.empty_like(z)
    state = torch.empty((2 + tab.stages) * z.numel(), device=dev, dtype=torch.float32)
    rc = lib.pfm_ca_sample_rk(ctypes.byref(layout.desc), _ptr(blob), ctypes.byref(tab), _ptr(ts), _ptr(dts), ode_steps - 1,
                              _ptr(z), _ptr(cond), _ptr(mask), _ptr(out), B, int(bool(premask and mask is not None)),
                              _ptr(state), _ptr(workspace(layout, B, dev)), _stream_ptr(dev))
    _lib.check(rc, "pfm_ca_sample_rk")
    return out


def ca_fm_loss_forward(layout: CaLayout, blob, x, t, a, cond=None, mask=None, sigma: float = 1e-4, kind: str = "FM-OT",
                       eps: Optional[torch.Tensor] = None):
    """Loss forward with the draws given (a = z for FM-OT / droid; a = x0, eps for CFM).
    Returns (sums (2,) = [sum (v-u)^2, sum mask], saved = (y, u, v, workspace))."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep(layout, blob, x, cond, mask)
    if kind not in _KINDS:
        raise NotImplementedError(f"loss kind {kind} has no HIP kernel")
    t, _ = _time_arg(layout, t, B, dev, per_jet=True)
    a = _dev_f32("a", a, dev, tuple(x.shape))
    if kind == "CFM":
        if eps is None:
            raise ValueError("CFM needs the second noise draw eps")
        if mask is None:
            raise ValueError("CFM loss needs a mask (losses.py:119)")
        eps = _dev_f32("eps", eps, dev, tuple(x.shape))
    y, u, v = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    sums = torch.zeros(2, device=dev, dtype=torch.float32)
    ws = workspace(layout, B, dev, train=True)
    rc = lib.pfm_ca_fm_loss_forward(ctypes.byref(layout.desc), _ptr(blob), _KINDS[kind], float(sigma), _ptr(t), _ptr(x),
                                    _ptr(a), _ptr(eps), _ptr(cond), _ptr(mask), _ptr(y), _ptr(u), _ptr(v), _ptr(sums), B,
                                    _ptr(ws), _stream_ptr(dev))
    _lib.check(rc, "pfm_ca_fm_loss_forward")
    return sums, (y, u, v, ws)


def ca_fm_loss_backward(layout: CaLayout, blob, cond, mask, saved, gscale: torch.Tensor, d_y=None) -> torch.Tensor:
    """Gradient blob (layout.blob_total floats); gscale: 0-dim device tensor grad_output / sum(mask).  d_y: (B, N, F) tensor that receives
    d loss / d y * gscale (pfm_ca_fm_loss_backward_dx: chains of flows)."""
    lib = _lib.load()
    y, u, v, ws = saved
    dev, B = y.device, y.shape[0]
    cache = layout.__dict__.setdefault("_bscratch", {})
    key = (B, str(dev))
    if key not in cache:
        n = lib.pfm_ca_backward_scratch_floats(ctypes.byref(layout.desc), B)
        cache.clear()
        cache[key] = torch.empty(n, device=dev, dtype=torch.float32)
    gblob = torch.zeros(layout.blob_total, device=dev, dtype=torch.float32)
    gs = gscale.to(device=dev, dtype=torch.float32).reshape(1).contiguous()
    if d_y is not None:
        rc = lib.pfm_ca_fm_loss_backward_dx(ctypes.byref(layout.desc), _ptr(blob), _ptr(cond), _ptr(mask), _ptr(y), _ptr(u), _ptr(v),
                                            _ptr(gs), _ptr(gblob), _ptr(d_y), B, _ptr(ws), _ptr(cache[key]), _stream_ptr(dev))
        _lib.check(rc, "pfm_ca_fm_loss_backward_dx")
        return gblob
    rc = lib.pfm_ca_fm_loss_backward(ctypes.byref(layout.desc), _ptr(blob), _ptr(cond), _ptr(mask), _ptr(y), _ptr(u), _ptr(v),
                                     _ptr(gs), _ptr(gblob), B, _ptr(ws), _ptr(cache[key]), _stream_ptr(dev))
    _lib.check(rc, "pfm_ca_fm_loss_backward")
    return gblob
