"""Flow-matching loss of the EPiC vector field as a torch.autograd.Function over the HIP kernels.

Mirrors FlowMatchingLoss / ConditionalFlowMatchingLoss of the reference
(particle_fm/models/components/losses.py:38-77, 101-136) with the random draws passed in.  The
differentiable input is the layout's *source vector* (effective weights | biases | freqs | 0,
layout.EpicLayout.source_vector); the gather into the kernel blob happens inside forward(), and
backward() gathers the kernel's gradient blob back with one index op.  Autograd then continues
through the weight-norm reparametrisation to weight_g / weight_v / bias, so DDP / Lightning hooks
see ordinary .grad accumulation on the real parameters.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib, hip_ops
from .layout import EpicLayout


class _Maps:
    """device copies of the (constant) gather maps, cached on the layout object per device"""

    @classmethod
    def get(cls, layout: EpicLayout, device):
        cache = layout.__dict__.setdefault("_device_maps", {})
        key = str(device)
        ent = cache.get(key)
        if ent is None:
            ent = (
                torch.from_numpy(layout.index_map).to(device),
                torch.from_numpy(layout.src_gpos.astype("int64")).to(device),
                layout.desc_tail().to(device),
            )
            cache[key] = ent
        return ent


def pack_blob_from_source(layout: EpicLayout, src: torch.Tensor) -> torch.Tensor:
    imap, _, tail = _Maps.get(layout, src.device)
    return layout.finish_blob(torch.cat([src.detach()[imap], tail]))  # + the bf16 copies of the particle blocks, if the layout wants them


class EpicFMLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, layout, x, t, z, eps, cond, mask, sigma, kind, temb=None):
        blob = pack_blob_from_source(layout, src)
        parts, count, saved = hip_ops.epic_fm_loss_forward(layout, blob, x, t, z, cond, mask, sigma, kind, eps, temb=temb)
        ctx.has_temb = temb is not None
        total = count.sum()
        loss = parts.sum() / total  # losses.py:75-76 / :130
        ctx.layout = layout
        ctx.mask = mask
        ctx.cond = cond
        ctx.save_for_backward(blob, saved, total)
        ctx.n_source = src.numel()
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        layout = ctx.layout
        blob, saved, total = ctx.saved_tensors
        dev = blob.device
        lib = _lib.load()
        B = saved.shape[0]
        gblob = torch.zeros_like(blob)
        inv_total = (1.0 / total).reshape(1).contiguous()
        gscale = grad_loss.to(torch.float32).reshape(1).contiguous()
        cond, mask = ctx.cond, ctx.mask
        if layout.cfg.global_cond_dim == 0:
            cond = None
        cond = None if cond is None else cond.to(torch.float32).contiguous()
        maskf = None if mask is None else mask.reshape(B, -1).to(torch.float32).contiguous()
        d_temb = torch.empty(B, layout.cfg.t_dim, device=dev, dtype=torch.float32) if ctx.has_temb else None
        # (the caller's embedding network trains too: d loss / d temb comes back with the weight gradient)
        hip_ops.epic_loss_backward(layout, blob, cond, maskf, saved, inv_total, gscale, gblob, d_temb=d_temb)
        _, gpos, _ = _Maps.get(layout, dev)
        # every weight / bias has exactly one gradient slot in the blob (layout.src_gpos): a plain gather;
        # freqs and the zero pad get no gradient
        d_src = torch.zeros(ctx.n_source, device=dev, dtype=torch.float32)
        d_src[: gpos.numel()] = gblob[gpos]
        return d_src, None, None, None, None, None, None, None, None, None, d_temb


def epic_fm_loss(layout: EpicLayout, src: torch.Tensor, x: torch.Tensor, t: torch.Tensor, z: torch.Tensor,
                 cond: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None, sigma: float = 1e-4,
                 kind: str = "FM-OT", eps: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None) -> torch.Tensor:
    """loss = sum((v - u)^2) / sum(mask) with v = EPiC(t, y); differentiable w.r.t. ``src`` and, if given, the caller-supplied
    time embedding ``temb`` (B,T)."""
    return EpicFMLossFn.apply(src, layout, x, t, z, eps, cond, mask, float(sigma), kind, temb)


class EpicDiffusionLossFn(torch.autograd.Function):
    """DiffusionLoss (losses.py:207-290) with the draws given: loss = sum_b w_b * sum_n,f criterion(v - z) / sum(mask),
    w_b = 1 + mle_weight * beta(t_b) / noise_rate(t_b)."""

    @staticmethod
    def forward(ctx, src, layout, x, t, z, rates, jet_w, cond, mask, criterion):
        blob = pack_blob_from_source(layout, src)
        parts, count, saved = hip_ops.epic_diffusion_loss_forward(layout, blob, x, t, z, rates, cond, mask, criterion)
        total = count.sum()
        ctx.layout, ctx.mask, ctx.cond, ctx.criterion = layout, mask, cond, criterion
        ctx.save_for_backward(blob, saved, total, jet_w)
        ctx.n_source = src.numel()
        return (parts * jet_w).sum() / total

    @staticmethod
    def backward(ctx, grad_loss):
        layout = ctx.layout
        blob, saved, total, jet_w = ctx.saved_tensors
        dev = blob.device
        lib = _lib.load()
        B = saved.shape[0]
        gblob = torch.zeros_like(blob)
        inv_total = (1.0 / total).reshape(1).contiguous()
        gscale = grad_loss.to(torch.float32).reshape(1).contiguous()
        cond = None if (ctx.cond is None or layout.cfg.global_cond_dim == 0) else ctx.cond.to(torch.float32).contiguous()
        maskf = None if ctx.mask is None else ctx.mask.reshape(B, -1).to(torch.float32).contiguous()
        hip_ops.epic_loss_backward(layout, blob, cond, maskf, saved, inv_total, gscale, gblob, criterion=ctx.criterion,
                                   jet_w=jet_w.contiguous())
        _, gpos, _ = _Maps.get(layout, dev)
        d_src = torch.zeros(ctx.n_source, device=dev, dtype=torch.float32)
        d_src[: gpos.numel()] = gblob[gpos]
        return (d_src,) + (None,) * 9


MLE_LOSS_WEIGHT = 0.001  # losses.py:226


def epic_diffusion_loss(layout: EpicLayout, src, x, t, z, cond=None, mask=None, criterion: str = "huber",
                        diff_config=None) -> torch.Tensor:
    """z must already be multiplied by the mask (losses.py:244).  diff_config: {"max_sr", "min_sr"} of VPDiffusionSchedule."""
    dc = dict(diff_config or {"max_sr": 1, "min_sr": 1e-8})
    sr, nr, beta = hip_ops.diffusion_schedule(t.to(torch.float32), **dc)
    rates = torch.stack([sr, nr], dim=1).contiguous()
    jet_w = (1.0 + MLE_LOSS_WEIGHT * (beta / nr)).contiguous()
    return EpicDiffusionLossFn.apply(src, layout, x, t, z, rates, jet_w, cond, mask, criterion)
