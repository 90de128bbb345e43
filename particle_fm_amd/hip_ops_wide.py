"""Tensor-level wrappers over include/pfm_epicw.h (EPiC at widths beyond the jet-resident kernel).  No CPU path."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .hip_ops import _dev_f32, _prep_common, _ptr, _stream_ptr, midpoint_grid, rk_grid, rk_tableau
from .hip_ops_tf import _temb_table, _time_arg, temb_given
from .layout_wide import EpicWideLayout


def workspace(layout: EpicWideLayout, n_jets: int, device, train: bool = False) -> torch.Tensor:
    lib = _lib.load()
    if train:
        # The loss forward hands this tensor to its autograd node as the saved activations: it must be private to the call.  (A
        # cached one would be overwritten by a second forward of the same batch size before the first backward -- two
        # micro-batches summed before .backward(), a no_grad validation loss between a forward and its backward -- and the
        # backward would silently return wrong gradients.)  torch's caching allocator hands the block back in steady state.
        n = lib.pfm_ew_workspace_floats(ctypes.byref(layout.desc), n_jets, 1)
        if n < 0:
            _lib.check(1, "pfm_ew_workspace_floats")
        return torch.empty(n, device=device, dtype=torch.float32)
    cache = layout.__dict__.setdefault("_ws", {})
    # one workspace per stream: samples queued on two streams may run at the same time
    key = (n_jets, bool(train), str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in cache:
        n = lib.pfm_ew_workspace_floats(ctypes.byref(layout.desc), n_jets, int(train))
        if n < 0:
            _lib.check(1, "pfm_ew_workspace_floats")
        for k in [k for k in cache if k[1] == key[1] and k[0] != key[0]]:
            del cache[k]
        cache[key] = torch.empty(n, device=device, dtype=torch.float32)
    return cache[key]


def ew_forward(layout: EpicWideLayout, blob, t, x, cond=None, mask=None) -> torch.Tensor:
    """v = EPiC(t, x, cond, mask).  t: (B,) one time per jet, or a single element for one shared time; a layout with t_emb="gaussian"
    (PFM_EW_F_TEMB_GIVEN) takes the time EMBEDDING rows (B, T) / one shared row instead."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep_common(layout, blob, x, cond, mask)
    t, t_per_jet = _time_arg(layout, t, B, dev)
    v = torch.empty_like(x)
    rc = lib.pfm_ew_forward(ctypes.byref(layout.desc), _ptr(blob), _ptr(t), t_per_jet, _ptr(x),
                            _ptr(cond), _ptr(mask), _ptr(v), B, _ptr(workspace(layout, B, dev)), _stream_ptr(dev))
    _lib.check(rc, "pfm_ew_forward")
    return v


def ew_backward_dtemb(layout, B: int, dev) -> torch.Tensor:
    """d loss / d temb (B, T) of the loss backward that has just run for this layout and batch size (pfm_ew_backward_dtemb reads that
    backward's scratch: call it right behind ew_fm_loss_backward, same stream)."""
    lib = _lib.load()
    scratch = layout.__dict__["_bscratch"][(B, str(dev))]
    out = torch.empty(B, layout.cfg.t_dim, device=dev, dtype=torch.float32)
    rc = lib.pfm_ew_backward_dtemb(ctypes.byref(layout.desc), _ptr(scratch), B, _ptr(out), _stream_ptr(dev))
    _lib.check(rc, "pfm_ew_backward_dtemb")
    return out


def ew_sample_midpoint(layout: EpicWideLayout, blob, z, cond=None, mask=None, ode_steps: int = 100, premask: bool = True, temb_fn=None):
    lib = _lib.load()
    dev, B, blob, z, cond, mask = _prep_common(layout, blob, z, cond, mask)
    if ode_steps < 2:
        raise ValueError("ode_steps must be >= 2")
    ts, dts = midpoint_grid(ode_steps, dev)
    if temb_given(layout):  # the table of embeddings replaces the time grid (PFM_EW_F_TEMB_GIVEN)
        ts = _temb_table(temb_fn, ts, dev)
    out = torch.empty_like(z)
    state = torch.empty(2 * z.numel(), device=dev, dtype=torch.float32)
    rc = lib.pfm_ew_sample_midpoint(ctypes.byref(layout.desc), _ptr(blob), _ptr(ts), _ptr(dts), ode_steps - 1, _ptr(z),
                                    _ptr(cond), _ptr(mask), _ptr(out), B, int(bool(premask and mask is not None)),
                                    _ptr(state), _ptr(workspace(layout, B, dev)), _stream_ptr(dev))
    _lib.check(rc, "pfm_ew_sample_midpoint")
    return out


def ew_sample_rk(layout: EpicWideLayout, blob, z, cond=None, mask=None, ode_steps: int = 100, solver: str = "rk4", diff_config=None,
                 premask: bool = True, t0: float = 1.0, t1: float = 0.0, temb_fn=None) -> torch.Tensor:
    """x(t1) from x(t0) = z (*mask) with the fixed-step explicit Runge-Kutta scheme ``solver`` ("euler", "midpoint", "rk4" =
    torchdyn's 3/8 rule) over linspace(t0, t1, ode_steps); all launches queued on the current stream."""
    lib = _lib.load()
    dev, B, blob, z, cond, mask = _prep_common(layout, blob, z, cond, mask)
    if ode_steps < 2:
        raise ValueError("ode_steps must be >= 2")
    tab = rk_tableau(solver)
    ts, dts = rk_grid(ode_steps, solver, t0, t1)
    ts, dts = ts.to(dev), dts.to(dev)
    out = torch.empty_like(z)
    state = torch.empty((2 + tab.stages) * z.numel(), device=dev, dtype=torch.float32)
    if temb_given(layout):
        if diff_config is not None:
            raise NotImplementedError("loss_type='diffusion' with t_emb='gaussian' has no HIP sampler on the row-matrix EPiC path")
        ts = _temb_table(temb_fn, ts, dev)
    if diff_config is not None:  # loss_type="diffusion": the probability-flow ODE of a noise-predicting network
        from .hip_ops import diffusion_schedule
        _, nr, beta = diffusion_schedule(ts, **diff_config)
        rhs = torch.stack([-0.5 * beta, nr], dim=1).contiguous()
        rc = lib.pfm_ew_sample_rk_rhs(ctypes.byref(layout.desc), _ptr(blob), ctypes.byref(tab), _ptr(ts), _ptr(dts), ode_steps - 1,
                                      _ptr(z), _ptr(cond), _ptr(mask), _ptr(out), B, int(bool(premask and mask is not None)),
                                      _ptr(state), _ptr(workspace(layout, B, dev)), _ptr(rhs), _stream_ptr(dev))
        _lib.check(rc, "pfm_ew_sample_rk_rhs")
        return out
    rc = lib.pfm_ew_sample_rk(ctypes.byref(layout.desc), _ptr(blob), ctypes.byref(tab), _ptr(ts), _ptr(dts), ode_steps - 1,
                              _ptr(z), _ptr(cond), _ptr(mask), _ptr(out), B, int(bool(premask and mask is not None)),
                              _ptr(state), _ptr(workspace(layout, B, dev)), _stream_ptr(dev))
    _lib.check(rc, "pfm_ew_sample_rk")
    return out


_KINDS = {"FM-OT": 0, "CFM": 1, "droid": 2}


def ew_fm_loss_forward(layout: EpicWideLayout, blob, x, t, a, cond=None, mask=None, sigma: float = 1e-4, kind: str = "FM-OT",
                       eps=None):
    """Loss forward with the draws given.  Returns (sums (2,) = [sum (v-u)^2, sum mask], saved = (y, u, v, workspace, mask))."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep_common(layout, blob, x, cond, mask)
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
        if mask is None:
            raise ValueError("CFM loss needs a mask (losses.py:119)")
        eps = _dev_f32("eps", eps, dev, tuple(x.shape))
    y, u, v = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    sums = torch.zeros(2, device=dev, dtype=torch.float32)
    ws = workspace(layout, B, dev, train=True)
    rc = lib.pfm_ew_fm_loss_forward(ctypes.byref(layout.desc), _ptr(blob), _KINDS[kind], float(sigma), _ptr(t), _ptr(x),
                                    _ptr(a), _ptr(eps), _ptr(cond), _ptr(mask), _ptr(y), _ptr(u), _ptr(v), _ptr(sums), B,
                                    _ptr(ws), _stream_ptr(dev))
    _lib.check(rc, "pfm_ew_fm_loss_forward")
    return sums, (y, u, v, ws, mask)


def ew_diffusion_loss_forward(layout: EpicWideLayout, blob, x, t, z, rates, jet_w, cond=None, mask=None, criterion: str = "huber"):
    """DiffusionLoss forward with the draws given (see hip_ops.epic_diffusion_loss_forward): rates (B,2), jet_w (B,).
    Returns (sums (2,) = [sum_b w_b sum criterion(v - z), sum mask], saved)."""
    lib = _lib.load()
    dev, B, blob, x, cond, mask = _prep_common(layout, blob, x, cond, mask)
    t = _dev_f32("t", t, dev, (B,))
    z = _dev_f32("z", z, dev, tuple(x.shape))
    rates = _dev_f32("rates", rates, dev, (B, 2))
    jet_w = _dev_f32("jet_w", jet_w, dev, (B,))
    y, u, v = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    sums = torch.zeros(2, device=dev, dtype=torch.float32)
    ws = workspace(layout, B, dev, train=True)
    rc = lib.pfm_ew_diffusion_loss_forward(ctypes.byref(layout.desc), _ptr(blob), {"mse": 0, "huber": 1}[criterion], _ptr(rates),
                                           _ptr(jet_w), _ptr(t), _ptr(x), _ptr(z), _ptr(cond), _ptr(mask), _ptr(y), _ptr(u),
                                           _ptr(v), _ptr(sums), B, _ptr(ws), _stream_ptr(dev))
    _lib.check(rc, "pfm_ew_diffusion_loss_forward")
    return sums, (y, u, v, ws, mask)


def ew_fm_loss_backward(layout: EpicWideLayout, blob, saved, gscale: torch.Tensor, criterion=None, jet_w=None, d_y=None) -> torch.Tensor:
    """Gradient blob (layout.blob_total floats); gscale: 0-dim device tensor grad_output / sum(mask).
    criterion / jet_w: the diffusion loss's (None: the flow-matching losses).  d_y: (B, N, F) tensor that receives d loss / d y * gscale
    (pfm_ew_fm_loss_backward_dx; flow-matching losses only)."""
    lib = _lib.load()
    y, u, v, ws, mask = saved
    dev, B = y.device, y.shape[0]
    cache = layout.__dict__.setdefault("_bscratch", {})
    key = (B, str(dev))
    if key not in cache:
        n = lib.pfm_ew_backward_scratch_floats(ctypes.byref(layout.desc), B)
        cache.clear()
        cache[key] = torch.empty(n, device=dev, dtype=torch.float32)
    gblob = torch.zeros(layout.blob_total, device=dev, dtype=torch.float32)
    gs = gscale.to(device=dev, dtype=torch.float32).reshape(1).contiguous()
    if criterion is not None:
        jw = _dev_f32("jet_w", jet_w, dev, (B,))
        rc = lib.pfm_ew_diffusion_loss_backward(ctypes.byref(layout.desc), _ptr(blob), {"mse": 0, "huber": 1}[criterion], _ptr(jw),
                                                _ptr(mask), _ptr(y), _ptr(u), _ptr(v), _ptr(gs), _ptr(gblob), B, _ptr(ws),
                                                _ptr(cache[key]), _stream_ptr(dev))
        _lib.check(rc, "pfm_ew_diffusion_loss_backward")
        return gblob
    if d_y is not None:
        rc = lib.pfm_ew_fm_loss_backward_dx(ctypes.byref(layout.desc), _ptr(blob), _ptr(mask), _ptr(y), _ptr(u), _ptr(v), _ptr(gs),
                                            _ptr(gblob), _ptr(d_y), B, _ptr(ws), _ptr(cache[key]), _stream_ptr(dev))
        _lib.check(rc, "pfm_ew_fm_loss_backward_dx")
        return gblob
    rc = lib.pfm_ew_fm_loss_backward(ctypes.byref(layout.desc), _ptr(blob), _ptr(mask), _ptr(y), _ptr(u), _ptr(v), _ptr(gs),
                                     _ptr(gblob), B, _ptr(ws), _ptr(cache[key]), _stream_ptr(dev))
    _lib.check(rc, "pfm_ew_fm_loss_backward")
    return gblob
