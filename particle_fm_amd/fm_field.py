"""The vector field v = f(t, x) of the row-matrix models (Full-Transformer, cross-attention, wide EPiC, MDMA) as ONE differentiable
function of the parameters, for the objectives that have no fused loss kernel on these paths (loss_type="diffusion":
losses.py:207-290 builds DiffusionLoss for any `model`, flow_matching_module.py:452-458).

No new kernels: the loss entry points of each path are used as a forward-with-saved-activations and a backward-from-an-upstream-gradient.
  forward : pfm_*_fm_loss_forward with kind "droid" and z = 0  ->  y = x + t * 0 = x, u = 0, v = f(t, x) and the saved activations
  backward: the loss backward differentiates sum (v - u)^2 * gscale, i.e. starts from dv = 2 (v - u) gscale (times the output mask
            where the network masks its output: wide EPiC, MDMA).  With u' = v - G / 2 and gscale = 1 that IS the upstream gradient G
            (MDMA sums it over the broadcast features, as autograd does for its (B, N, 1) output), so the same call returns
            d <G, v> / d parameters.
Whatever is built on v -- criterion, masks, per-jet weights -- is a handful of element-wise torch ops on (B, N, F) device tensors whose
autograd ends in this node.  The shipped FM-OT / CFM / droid objectives keep their fused loss kernels (fm_loss_*.py)."""
from __future__ import annotations

from typing import Optional

import torch

from . import hip_ops_ca, hip_ops_mdma, hip_ops_tf, hip_ops_wide


def _freq_table(layout, freqs, default_freqs):
    f = default_freqs(layout.cfg.t_dim, layout.cfg.t_emb) if freqs is None else freqs
    if layout.cfg.t_emb == "sincos" and f.numel() == layout.cfg.frequencies:
        f = torch.cat([f, f])
    return f


def _upstream(saved, G):
    """saved (y, u, v, ...) with u replaced so that the loss backward starts from the upstream gradient G"""
    y, u, v = saved[:3]
    return (y, v - 0.5 * G.to(v.dtype).reshape(v.shape)) + tuple(saved[2:])


class TfFieldFn(torch.autograd.Function):
    """Full-Transformer (ops = hip_ops_tf) / cross-attention (ops = hip_ops_ca) field over the flat parameter vector.  ``t``: the times
    (B,), or -- a layout with t_emb="gaussian" (PFM_*_F_TEMB_GIVEN) -- the time EMBEDDING rows (B, T), then a differentiable input: the
    backward also returns d / d temb (pfm_*_backward_dtemb), from which autograd continues into the CNF's embedding network.  Also
    differentiable w.r.t. the particle input x (pfm_{tf,ca}_fm_loss_backward_dx): chains of flows."""

    @staticmethod
    def forward(ctx, flat_params, t, layout, freqs, ops, x, cond, mask):
        dev = x.device
        src = torch.cat([flat_params.to(torch.float32), freqs.to(dev, torch.float32), torch.zeros(1, device=dev)])
        blob = src[layout.index_map_on(dev)]
        fwd = ops.tf_fm_loss_forward if ops is hip_ops_tf else ops.ca_fm_loss_forward
        _, saved = fwd(layout, blob, x, t.detach(), torch.zeros_like(x), cond, mask, 0.0, "droid", None)
        ctx.layout, ctx.saved, ctx.blob, ctx.ops, ctx.t, ctx.cond = layout, saved, blob, ops, t.detach(), cond
        ctx.mask = None if mask is None else mask.reshape(x.shape[0], -1).to(torch.float32).contiguous()
        ctx.temb_given = hip_ops_tf.temb_given(layout)
        return saved[2].clone()

    @staticmethod
    def backward(ctx, G):
        lay, ops = ctx.layout, ctx.ops
        one = torch.ones((), device=G.device)
        d_x = torch.empty_like(G) if ctx.needs_input_grad[5] else None
        if ops is hip_ops_tf:
            gblob = ops.tf_fm_loss_backward(lay, ctx.blob, ctx.t, ctx.cond, ctx.mask, _upstream(ctx.saved, G.contiguous()), one, d_y=d_x)
        else:
            gblob = ops.ca_fm_loss_backward(lay, ctx.blob, ctx.cond, ctx.mask, _upstream(ctx.saved, G.contiguous()), one, d_y=d_x)
        d_t = None
        if ctx.temb_given and ctx.needs_input_grad[1]:
            dtemb = ops.tf_backward_dtemb if ops is hip_ops_tf else ops.ca_backward_dtemb
            d_t = dtemb(lay, ctx.blob, G.shape[0], G.device).reshape(ctx.t.shape)
        return (gblob[lay.grad_pos_on(gblob.device)], d_t, None, None, None, d_x, None, None)


def tf_field(layout, flat_params, t, x, cond=None, mask=None, freqs: Optional[torch.Tensor] = None):
    from .layout_tf import default_freqs
    return TfFieldFn.apply(flat_params, t, layout, _freq_table(layout, freqs, default_freqs), hip_ops_tf, x, cond, mask)


def ca_field(layout, flat_params, t, x, cond=None, mask=None, freqs: Optional[torch.Tensor] = None):
    from .layout_ca import default_freqs
    return TfFieldFn.apply(flat_params, t, layout, _freq_table(layout, freqs, default_freqs), hip_ops_ca, x, cond, mask)


class MdmaFieldFn(torch.autograd.Function):
    """MDMA field, broadcast over the features as the reference's loss arithmetic does with its (B, N, 1) output.  ``t``: the times
    (B,), or -- a layout with t_emb="gaussian" (PFM_MDMA_F_TEMB_GIVEN) -- the time EMBEDDING rows (B, T), then a differentiable input."""

    @staticmethod
    def forward(ctx, flat_params, layout, freqs, x, t, mask, cond=None):
        dev = x.device
        src = torch.cat([flat_params.to(torch.float32), freqs.to(dev, torch.float32), torch.zeros(1, device=dev)])
        blob = src[layout.index_map_on(dev)]
        _, saved = hip_ops_mdma.mdma_fm_loss_forward(layout, blob, x, t.detach(), torch.zeros_like(x), mask, 0.0, "droid", None, cond=cond)
        ctx.layout, ctx.saved, ctx.blob = layout, saved, blob
        ctx.temb_shape = tuple(t.shape) if hip_ops_tf.temb_given(layout) else None
        return saved[2].clone()

    @staticmethod
    def backward(ctx, G):
        lay = ctx.layout
        gblob = hip_ops_mdma.mdma_fm_loss_backward(lay, ctx.blob, _upstream(ctx.saved, G.contiguous()), torch.ones((), device=G.device))
        d_t = None
        if ctx.temb_shape is not None and ctx.needs_input_grad[4]:
            d_t = hip_ops_mdma.mdma_backward_dtemb(lay, G.shape[0], G.device).reshape(ctx.temb_shape)
        return (gblob[lay.grad_pos_on(gblob.device)], None, None, None, d_t, None, None)


def mdma_field(layout, flat_params, t, x, mask, freqs: Optional[torch.Tensor] = None, cond=None):
    from .layout_mdma import default_freqs
    return MdmaFieldFn.apply(flat_params, layout, _freq_table(layout, freqs, default_freqs), x, t, mask, cond)


class EpicWideFieldFn(torch.autograd.Function):
    """Wide (row-matrix) EPiC field over the layout's source vector (autograd continues through the weight-norm reparametrisation).
    ``t``: the times (B,), or -- a layout with t_emb="gaussian" (PFM_EW_F_TEMB_GIVEN) -- the time EMBEDDING rows (B, T), then a
    differentiable input (pfm_ew_backward_dtemb).  Also differentiable w.r.t. the particle input x (pfm_ew_fm_loss_backward_dx): chains."""

    @staticmethod
    def forward(ctx, src, layout, x, t, cond, mask):
        from .fm_loss_wide import pack_blob_from_source
        blob = pack_blob_from_source(layout, src)
        _, saved = hip_ops_wide.ew_fm_loss_forward(layout, blob, x, t.detach(), torch.zeros_like(x), cond, mask, 0.0, "droid", None)
        ctx.layout, ctx.saved, ctx.blob, ctx.n_source = layout, saved, blob, src.numel()
        ctx.temb_shape = tuple(t.shape) if hip_ops_tf.temb_given(layout) else None
        return saved[2].clone()

    @staticmethod
    def backward(ctx, G):
        from .fm_loss_wide import _maps
        lay = ctx.layout
        d_x = torch.empty_like(G) if ctx.needs_input_grad[2] else None
        gblob = hip_ops_wide.ew_fm_loss_backward(lay, ctx.blob, _upstream(ctx.saved, G.contiguous()), torch.ones((), device=G.device), d_y=d_x)
        gpos = _maps(lay, gblob.device)[1]
        d_src = torch.zeros(ctx.n_source, device=gblob.device, dtype=torch.float32)
        d_src[: gpos.numel()] = gblob[gpos]
        d_t = None
        if ctx.temb_shape is not None and ctx.needs_input_grad[3]:
            d_t = hip_ops_wide.ew_backward_dtemb(lay, G.shape[0], G.device).reshape(ctx.temb_shape)
        return (d_src, None, d_x, d_t, None, None)


def epic_wide_field(layout, src, t, x, cond=None, mask=None):
    return EpicWideFieldFn.apply(src, layout, x, t, cond, mask)


class EpicFieldFn(torch.autograd.Function):
    """Jet-resident EPiC field over the layout's source vector, differentiable w.r.t. the parameters AND the particle input x
    (pfm_epic_fm_loss_backward_dx): the building block of n_transforms > 1 (losses.py:66-69 feeds each flow's output to the next).
    The saved record of a jet starts with y | v | u (pfm_hip.h): the upstream gradient goes in as u := v - G / 2.
    ``temb`` (B, T): a caller-supplied time embedding (t_emb="gaussian"), then a differentiable input too
    (pfm_epic_fm_loss_backward_temb / _dx_temb return d / d temb)."""

    @staticmethod
    def forward(ctx, src, x, layout, t, cond, mask, temb=None):
        from .fm_loss import pack_blob_from_source
        from . import hip_ops
        blob = pack_blob_from_source(layout, src)
        _, _, saved = hip_ops.epic_fm_loss_forward(layout, blob, x, t, torch.zeros_like(x), cond, mask, 0.0, "droid", None,
                                                   temb=None if temb is None else temb.detach())
        B, N, F = x.shape
        ctx.layout, ctx.cond, ctx.mask, ctx.n_source, ctx.shape = layout, cond, mask, src.numel(), (B, N, F)
        ctx.temb_shape = None if temb is None else tuple(temb.shape)
        ctx.save_for_backward(blob, saved)
        r4 = (N * F + 3) & ~3
        v = saved[:, r4:r4 + N * F].reshape(B, N, F)
        # rows behind a jet's last valid particle are never computed (nor written): the field is 0 there (epic.py:391)
        return v.clone() if mask is None else torch.where(mask.reshape(B, N, 1) != 0, v, torch.zeros((), device=v.device))

    @staticmethod
    def backward(ctx, G):
        from .fm_loss import _Maps
        from . import hip_ops
        layout = ctx.layout
        blob, saved = ctx.saved_tensors
        B, N, F = ctx.shape
        dev = blob.device
        r4 = (N * F + 3) & ~3
        sv = saved.clone()  # (the node may be differentiated twice: leave the forward's record alone)
        sv[:, 2 * r4:2 * r4 + N * F] = sv[:, r4:r4 + N * F] - 0.5 * G.reshape(B, N * F).to(torch.float32)
        cond = None if (ctx.cond is None or layout.cfg.global_cond_dim == 0) else ctx.cond.to(torch.float32).contiguous()
        maskf = None if ctx.mask is None else ctx.mask.reshape(B, -1).to(torch.float32).contiguous()
        one = torch.ones(1, device=dev)
        gblob = torch.zeros_like(blob)
        d_temb = None if ctx.temb_shape is None else torch.empty(B, layout.cfg.t_dim, device=dev, dtype=torch.float32)
        d_y = torch.empty(B, N, F, device=dev, dtype=torch.float32) if (ctx.temb_shape is None or ctx.needs_input_grad[1]) else None
        hip_ops.epic_loss_backward(layout, blob, cond, maskf, sv, one, one, gblob, d_temb=d_temb, d_y=d_y)
        if d_temb is not None:
            d_temb = d_temb.reshape(ctx.temb_shape)
        _, gpos, _ = _Maps.get(layout, dev)
        d_src = torch.zeros(ctx.n_source, device=dev, dtype=torch.float32)
        d_src[: gpos.numel()] = gblob[gpos]
        return d_src, d_y, None, None, None, None, d_temb


def epic_field(layout, src, t, x, cond=None, mask=None, temb=None):
    return EpicFieldFn.apply(src, x, layout, t, cond, mask, temb)


def fm_loss_from_field(field, kind: str, x, t, a, eps, mask, sigma: float):
    """FM-OT / CFM / droid (losses.py:56-76, 115-130, 332-341) around a differentiable field(y) -> v: interpolation, target and the
    squared error as element-wise device ops in the reference's own expressions.  Used where no fused loss kernel applies (a
    caller-supplied time embedding on the transformer paths; chained flows: models/components/losses.py)."""
    tt = t.to(x.device, torch.float32).view(-1, 1, 1)
    m = torch.ones_like(x[..., :1]) if mask is None else mask.to(x.dtype)
    if kind == "FM-OT":
        y = (1 - tt) * x + (sigma + (1 - sigma) * tt) * a
        u = ((1 - sigma) * a - x) * m
    elif kind == "CFM":
        y = (1 - tt) * x + tt * a + sigma * eps
        u = (a - x) * m
    elif kind == "droid":
        y = x + tt * a
        u = a * m
    else:
        raise NotImplementedError(f"loss kind {kind}")
    return (field(y) - u).square().sum() / m.sum()


def diffusion_loss_from_field(v, z, mask, t, criterion: str, diff_config, mle_loss_weight: float = 0.001):
    """DiffusionLoss.forward behind the network call (losses.py:272-288): v = predicted noise, z = the (masked) noise.
    loss = sum crit(z, v) mask (1 + w beta / noise_rate) / sum mask, the rates in the reference's fp32 op order (hip_ops)."""
    from .hip_ops import diffusion_schedule
    _, nr, beta = diffusion_schedule(t.to(torch.float32), **dict(diff_config))
    if criterion == "mse":
        simple = (z - v).square()
    elif criterion == "huber":
        simple = torch.nn.functional.huber_loss(z, v, reduction="none")
    else:
        raise NotImplementedError(f"criterion {criterion} not supported")
    simple = simple * mask
    msum = mask.sum()
    out = simple.sum() / msum
    if mle_loss_weight:
        out = out + mle_loss_weight * ((beta / nr).view(-1, 1, 1) * simple).sum() / msum
    return out
