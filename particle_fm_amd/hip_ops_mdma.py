"""Tensor-level wrappers over the MDMA part of the C ABI (include/pfm_mdma.h).

PyTorch owns device memory and the stream; every number comes out of libpfm_hip.so.  No CPU path.  The field has ONE output per
particle (mdma.py:136); the library hands it back broadcast over the features, (B, N, F), which is what the reference's loss
and solver arithmetic turn it into."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .hip_ops import _dev_f32, _ptr, _stream_ptr, rk_grid, rk_tableau
from .hip_ops_tf import _temb_table, _time_arg, temb_given
from .hip_ops_tf import _KINDS
from .layout_mdma import MdmaLayout


def _cond_arg(layout: MdmaLayout, cond, B: int, dev):
    """The conditional variant's ONE value per jet (desc.c_cat; mdma.py:157-169) as a (B,) device tensor; None otherwise (never read)."""
    if not layout.cfg.needs_cond:
        return None
    if cond is None:
        raise ValueError("this MDMA configuration (global_cond_dim / local_cat_cond / global_cat_cond) needs the condition, one value per jet")
    return _dev_f32("cond", cond.reshape(-1), dev, (B,))


def _prep(layout: MdmaLayout, blob, x, mask):
    cfg = layout.cfg
    if not x.is_cuda:
        raise RuntimeError("HIP backend needs tensors on a ROCm device; there is no CPU fallback")
    dev = x.device
    B, N, F = x.shape
    if N != cfg.num_particles or F != cfg.features:
        raise ValueError(f"x has shape {tuple(x.shape)}, model expects (*, {cfg.num_particles}, {cfg.features})")
    blob = _dev_f32("blob", blob, dev, (layout.blob_total,))
    x = _dev_f32("x", x, dev)
    if mask is None:
        raise ValueError("MDMA needs a mask (MDMA.forward indexes with it, mdma.py:151)")
    mask = _dev_f32("mask", mask.reshape(B, N), dev, (B, N))
    return dev, B, blob, x, mask


def workspace(layout: MdmaLayout, n_jets: int, device, train: bool = False) -> torch.Tensor:
    """Activation workspace.  Train: private to the call (it becomes the autograd node's saved activations); inference: cached
    per (n_jets, device, stream) on the layout."""
    lib = _lib.load()
    if train:
        n = lib.pfm_mdma_workspace_floats(ctypes.byref(layout.desc), n_jets, 1)
        if n < 0:
            _lib.check(1, "pfm_mdma_workspace_floats")
        return torch.empty(n, device=device, dtype=torch.float32)
    cache = layout.__dict__.setdefault("_ws", {})
    key = (n_jets, str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in cache:
        n = lib.pfm_mdma_workspace_floats(ctypes.byref(layout.desc), n_jets, 0)
        if n < 0:
            _lib.check(1, "pfm_mdma_workspace_floats")
        for k in [k for k in cache if k[1:] == key[1:] and k[0] != key[0]]:
            del cache[k]
        cache[key] = torch.empty(n, device=device, dtype=torch.float32)
    return cache[key]


def mdma_forward(layout: MdmaLayout, blob, t, x, mask, cond=None) -> torch.Tensor:
    """(B, N, F) broadcast of MDMA(t, x, mask).  t: (B,) one time per jet, or 0-dim / (1,) for one shared time; a layout with
    t_emb="gaussian" (PFM_MDMA_F_TEMB_GIVEN) takes the time EMBEDDING rows (B, T) / one shared row instead."""
    lib = _lib.load()
    dev, B, blob, x, mask = _prep(layout, blob, x, mask)
    t, t_per_jet = _time_arg(layout, t, B, dev)
    cond = _cond_arg(layout, cond, B, dev)
    v = torch.empty_like(x)
    rc = lib.pfm_mdma_forward(ctypes.byref(layout.desc), _ptr(blob), _ptr(t), t_per_jet, _ptr(x), _ptr(cond),
                              _ptr(mask), _ptr(v), B, _ptr(workspace(layout, B, dev)), _stream_ptr(dev))
    _lib.check(rc, "pfm_mdma_forward")
    return v


def mdma_backward_dtemb(layout, B: int, dev) -> torch.Tensor:
    """d loss / d temb (B, T) of the loss backward that has just run for this layout and batch size (pfm_mdma_backward_dtemb reads that
    backward's scratch: call it right behind mdma_fm_loss_backward, same stream)."""
    lib = _lib.load()
    scratch = layout.__dict__["_bscratch"][(B, str(dev))]
    out = torch.empty(B, layout.cfg.t_dim, device=dev, dtype=torch.float32)
    rc = lib.pfm_mdma_backward_dtemb(ctypes.byref(layout.desc), _ptr(scratch), B, _ptr(out), _stream_ptr(dev))
    _lib.check(rc, "pfm_mdma_backward_dtemb")
    return out


def mdma_sample_rk(layout: MdmaLayout, blob, z, mask, ode_steps: int = 100, solver: str = "midpoint", premask: bool = True,
                   t0: float = 1.0, t1: float = 0.0, temb_fn=None, cond=None) -> torch.Tensor:
    """x(t1) from x(t0) = z (*mask) with the fixed-step explicit Runge-Kutta scheme ``solver`` ("euler", "midpoint", "rk4" =
    torchdyn's 3/8 rule) over linspace(t0, t1, ode_steps); all launches queued on the current stream."""
    lib = _lib.load()
    dev, B, blob, z, mask = _prep(layout, blob, z, mask)
    if ode_steps < 2:
        raise ValueError("ode_steps must be >= 2")
    tab = rk_tableau(solver)
    ts, dts = rk_grid(ode_steps, solver, t0, t1)
    ts, dts = ts.to(dev), dts.to(dev)
    if temb_given(layout):  # the table of embeddings replaces the time grid (PFM_MDMA_F_TEMB_GIVEN)
        ts = _temb_table(temb_fn, ts, dev)
    cond = _cond_arg(layout, cond, B, dev)
    out = torch.empty_like(z)
    state = torch.empty((2 + tab.stages) * z.numel(), device=dev, dtype=torch.float32)
    rc = lib.pfm_mdma_sample_rk(ctypes.byref(layout.desc), _ptr(blob), ctypes.byref(tab), _ptr(ts), _ptr(dts), ode_steps - 1,
                                _ptr(z), _ptr(cond), _ptr(mask), _ptr(out), B, int(bool(premask)), _ptr(state),
                                _ptr(workspace(layout, B, dev)), _stream_ptr(dev))
    _lib.check(rc, "pfm_mdma_sample_rk")
    return out


def mdma_fm_loss_forward(layout: MdmaLayout, blob, x, t, a, mask, sigma: float = 1e-4, kind: str = "FM-OT",
                         eps: Optional[torch.Tensor] = None, cond=None):
    """Loss forward with the draws given (a = z for FM-OT / droid; a = x0, eps for CFM).
    Returns (sums (2,) = [sum (v-u)^2 over (B, N, F), sum mask], saved = (y, u, v, workspace))."""
    lib = _lib.load()
    dev, B, blob, x, mask = _prep(layout, blob, x, mask)
    if kind not in _KINDS:
        raise NotImplementedError(f"loss kind {kind} has no HIP kernel")
    if temb_given(layout):  # (the interpolation must not depend on t: fm_field.py's forward-with-saved-activations, kind "droid", a = 0)
        t = _dev_f32("temb", t, dev, (B, layout.cfg.t_dim))
    else:
        t = _dev_f32("t", t, dev, (B,))
    a = _dev_f32("a", a, dev, tuple(x.shape))
    if kind == "CFM":
        if eps is None:
            raise ValueError("CFM needs the second noise draw eps")
        eps = _dev_f32("eps", eps, dev, tuple(x.shape))
    cond = _cond_arg(layout, cond, B, dev)
    y, u, v = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    sums = torch.zeros(2, device=dev, dtype=torch.float32)
    ws = workspace(layout, B, dev, train=True)
    rc = lib.pfm_mdma_fm_loss_forward(ctypes.byref(layout.desc), _ptr(blob), _KINDS[kind], float(sigma), _ptr(t), _ptr(x), _ptr(a),
                                      _ptr(eps), _ptr(cond), _ptr(mask), _ptr(y), _ptr(u), _ptr(v), _ptr(sums), B, _ptr(ws), _stream_ptr(dev))
    _lib.check(rc, "pfm_mdma_fm_loss_forward")
    return sums, (y, u, v, ws, mask)


def mdma_fm_loss_backward(layout: MdmaLayout, blob, saved, gscale: torch.Tensor) -> torch.Tensor:
    """Gradient blob (layout.blob_total floats); gscale: 0-dim device tensor grad_output / sum(mask)."""
    lib = _lib.load()
    y, u, v, ws, mask = saved
    dev, B = y.device, y.shape[0]
    cache = layout.__dict__.setdefault("_bscratch", {})
    key = (B, str(dev))
    if key not in cache:
        n = lib.pfm_mdma_backward_scratch_floats(ctypes.byref(layout.desc), B)
        cache.clear()
        cache[key] = torch.empty(n, device=dev, dtype=torch.float32)
    gblob = torch.zeros(layout.blob_total, device=dev, dtype=torch.float32)
    gs = gscale.to(device=dev, dtype=torch.float32).reshape(1).contiguous()
    rc = lib.pfm_mdma_fm_loss_backward(ctypes.byref(layout.desc), _ptr(blob), _ptr(mask), _ptr(y), _ptr(u), _ptr(v), _ptr(gs),
                                       _ptr(gblob), B, _ptr(ws), _ptr(cache[key]), _stream_ptr(dev))
    _lib.check(rc, "pfm_mdma_fm_loss_backward")
    return gblob
