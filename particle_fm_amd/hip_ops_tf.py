"""Tensor-level wrappers over the transformer part of the C ABI (include/pfm_tf.h).

PyTorch owns device memory and the stream; every number comes out of libpfm_hip.so.  No CPU path."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from .hip_ops import _dev_f32, _ptr, _stream_ptr, midpoint_grid, rk_grid, rk_tableau
from .layout_tf import TfLayout


def _prep(layout: TfLayout, blob, x, cond, mask):
    cfg = layout.cfg
    if not x.is_cuda:
        raise RuntimeError("HIP backend needs tensors on a ROCm device; there is no CPU fallback")
    dev = x.device
    B, N, F = x.shape
    if N != cfg.num_particles or F != cfg.features:
        raise ValueError(f"x has shape {tuple(x.shape)}, model expects (*, {cfg.num_particles}, {cfg.features})")
    blob = _dev_f32("blob", blob, dev, (layout.blob_total,))
    x = _dev_f32("x", x, dev)
    if cfg.global_cond_dim > 0:
        if cond is None:
            raise ValueError("global_cond_dim > 0 but no cond given")
        cond = _dev_f32("cond", cond, dev, (B, cfg.global_cond_dim))
    else:
        cond = None
    if mask is not None:
        mask = _dev_f32("mask", mask.reshape(B, N), dev, (B, N))
    return dev, B, blob, x, cond, mask


def temb_given(layout) -> bool:
    """PFM_*_F_TEMB_GIVEN (64 in pfm_tf.h / pfm_ca.h / pfm_epicw.h): the entry points take the time embedding through `t`."""
    return bool(layout.desc.flags & 64)


def _time_arg(layout, t, B, dev, per_jet: bool = False):
    """The `t` argument of the entry points: times -- (B,) one per jet, or 0-dim / (1,) shared -- or, for a layout with
    t_emb="gaussian" (PFM_*_F_TEMB_GIVEN), the time EMBEDDING rows (B, T) or one shared row (T,) / (1, T).  Returns (tensor, t_stride)."""
    T = layout.cfg.t_dim
    if temb_given(layout):
        t = _dev_f32("temb", t.reshape(-1, T), dev)
        if t.shape[0] not in (1, B) or (per_jet and t.shape[0] != B):
            raise ValueError(f"the time embedding has {t.shape[0]} rows, expected {'' if per_jet else '1 or '}{B}")
        return t, 1 if (t.shape[0] == B and (B > 1 or per_jet)) else 0
    t = _dev_f32("t", t.reshape(-1), dev)
    if t.numel() not in (1, B) or (per_jet and t.numel() != B):
        raise ValueError(f"t has {t.numel()} elements, expected {'' if per_jet else '1 or '}{B}")
    return t, 1 if (t.numel() == B and (B > 1 or per_jet)) else 0


def _temb_table(temb_fn, ts: torch.Tensor, dev) -> torch.Tensor:
    """The samplers' embedding table for a PFM_*_F_TEMB_GIVEN layout: temb_fn(ts) -> (n_evaluations, T), handed over TRANSPOSED
    ([T][n_evaluations]: evaluation e starts at table + e, like the time grid it replaces)."""
    tab = temb_fn(ts.to(dev))
    return tab.to(torch.float32).t().contiguous()


def tf_backward_dtemb(layout, blob, B: int, dev) -> torch.Tensor:
    """d loss / d temb (B, T) of the loss backward that has just run for this layout and batch size (pfm_tf_backward_dtemb reads that
    backward's scratch: call it right behind tf_fm_loss_backward, same stream)."""
    lib = _lib.load()
    scratch = layout.__dict__["_bscratch"][(B, str(dev))]
    out = torch.empty(B, layout.cfg.t_dim, device=dev, dtype=torch.float32)
    rc = lib.pfm_tf_backward_dtemb(ctypes.byref(layout.desc), _ptr(blob), _ptr(scratch), B, _ptr(out), _stream_ptr(dev))
    _lib.check(rc, "pfm_tf_backward_dtemb")
    return out


def workspace(layout: TfLayout, n_jets: int, device, train: bool = False) -> torch.Tensor:
    """Activation workspace; cached per (n_jets, train) on the layout (the kernels fully overwrite what they read)."""
    lib = _lib.load()
    if train:
        # The loss forward hands this tensor to its autograd node as the saved activations: it must be private to the call.  (A
        # cached one would be overwritten by a second forward of the same batch size before the first backward -- two
        # micro-batches summed before .backward(), a no_grad validation loss between a forward and its backward -- and the
        # backward would silently return wrong gradients.)  torch's caching allocator hands the block back in steady state.
        n = lib.pfm_tf_workspace_floats(ctypes.byref(layout.desc), n_jets, 1)
        if n < 0:
            _lib.check(1, "pfm_tf_workspace_floats")
        return torch.empty(n, device=device, dtype=torch.float32)
    cache = layout.__dict__.setdefault("_ws", {})
    # one workspace per stream: samples queued on two streams may run at the same time
    key = (n_jets, bool(train), str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in cache:
        n = lib.pfm_tf_workspace_floats(ctypes.byref(layout.desc), n_jets, int(train))
        if n < 0:
            _lib.check(1, "pfm_tf_workspace_floats")
        for k in [k for k in cache if k[1] == key[1] and k[2] == key[2] and k[0] != key[0]]:
            del cache[k]
        cache[key] = torch.empty(n, device=device, dtype=torch.float32)
    return cache[key]


def tf_forward(layout: TfLayout, blob, t, x, cond=None, mask=None) -> torch.Tensor:
    """v = FullTransformer(t, x, cond, mask).  t: (B,) one time per jet, or 0-dim / (1,) for one shared time."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep(layout, blob, x, cond, mask)
    t, t_per_jet = _time_arg(layout, t, B, dev)
    v = torch.empty_like(x)
    ws = workspace(layout, B, dev)
    rc = lib.pfm_tf_forward(ctypes.byref(layout.desc), _ptr(blob), _ptr(t), t_per_jet, _ptr(x),
                            _ptr(cond), _ptr(mask), _ptr(v), B, _ptr(ws), _stream_ptr(dev))
    _lib.check(rc, "pfm_tf_forward")
    return v


def tf_sample_midpoint(layout: TfLayout, blob, z, cond=None, mask=None, ode_steps: int = 100,
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
    rc = lib.pfm_tf_sample_midpoint(ctypes.byref(layout.desc), _ptr(blob), _ptr(ts), _ptr(dts), ode_steps - 1, _ptr(z),
                                    _ptr(cond), _ptr(mask), _ptr(out), B, int(bool(premask and mask is not None)),
                                    _ptr(state), _ptr(ws), _stream_ptr(dev))
    _lib.check(rc, "pfm_tf_sample_midpoint")
    return out


def tf_sample_rk(layout: TfLayout, blob, z, cond=None, mask=None, ode_steps: int = 100, solver: str = "rk4",
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
    rc = lib.pfm_tf_sample_rk(ctypes.byref(layout.desc), _ptr(blob), ctypes.byref(tab), _ptr(ts), _ptr(dts), ode_steps - 1,
                              _ptr(z), _ptr(cond), _ptr(mask), _ptr(out), B, int(bool(premask and mask is not None)),
                              _ptr(state), _ptr(workspace(layout, B, dev)), _stream_ptr(dev))
    _lib.check(rc, "pfm_tf_sample_rk")
    return out


_KINDS = {"FM-OT": 0, "CFM": 1, "droid": 2}


def tf_fm_loss_forward(layout: TfLayout, blob, x, t, a, cond=None, mask=None, sigma: float = 1e-4, kind: str = "FM-OT",
                       eps: Optional[torch.Tensor] = None):
    """Loss forward with the draws given (a = z for FM-OT; a = x0, eps for CFM).
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
    rc = lib.pfm_tf_fm_loss_forward(ctypes.byref(layout.desc), _ptr(blob), _KINDS[kind], float(sigma), _ptr(t), _ptr(x),
                                    _ptr(a), _ptr(eps), _ptr(cond), _ptr(mask), _ptr(y), _ptr(u), _ptr(v), _ptr(sums), B,
                                    _ptr(ws), _stream_ptr(dev))
    _lib.check(rc, "pfm_tf_fm_loss_forward")
    return sums, (y, u, v, ws)


def tf_fm_loss_backward(layout: TfLayout, blob, t, cond, mask, saved, gscale: torch.Tensor, d_y=None) -> torch.Tensor:
    """Gradient blob (layout.blob_total floats) of loss * gscale-normalisation; gscale: 0-dim device tensor
    grad_output / sum(mask).  d_y: (B, N, F) tensor that receives d loss / d y * gscale (pfm_tf_fm_loss_backward_dx: chains of flows)."""
    lib = _lib.load()
    y, u, v, ws = saved
    dev, B = y.device, y.shape[0]
    cache = layout.__dict__.setdefault("_bscratch", {})
    key = (B, str(dev))
    if key not in cache:
        n = lib.pfm_tf_backward_scratch_floats(ctypes.byref(layout.desc), B)
        cache.clear()
        cache[key] = torch.empty(n, device=dev, dtype=torch.float32)
    gblob = torch.zeros(layout.blob_total, device=dev, dtype=torch.float32)
    gs = gscale.to(device=dev, dtype=torch.float32).reshape(1).contiguous()
    if d_y is not None:
        rc = lib.pfm_tf_fm_loss_backward_dx(ctypes.byref(layout.desc), _ptr(blob), _ptr(cond), _ptr(mask), _ptr(y), _ptr(u), _ptr(v),
                                            _ptr(gs), _ptr(gblob), _ptr(d_y), B, _ptr(ws), _ptr(cache[key]), _stream_ptr(dev))
        _lib.check(rc, "pfm_tf_fm_loss_backward_dx")
        return gblob
    rc = lib.pfm_tf_fm_loss_backward(ctypes.byref(layout.desc), _ptr(blob), _ptr(t), _ptr(cond), _ptr(mask), _ptr(y), _ptr(u),
                                     _ptr(v), _ptr(gs), _ptr(gblob), B, _ptr(ws), _ptr(cache[key]), _stream_ptr(dev))
    _lib.check(rc, "pfm_tf_fm_loss_backward")
    return gblob
