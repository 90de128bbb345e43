"""Flow-matching losses of the wide EPiC path as one autograd node over the layout's source vector
(effective weights | biases | freqs | 0), like fm_loss.EpicFMLossFn for the jet-resident kernel: autograd continues
through the weight-norm reparametrisation to weight_g / weight_v / bias.  Reference: losses.py:38-77, 101-136."""
from __future__ import annotations

from typing import Optional

import torch

from . import hip_ops_wide
from .layout_wide import EpicWideLayout


def _maps(layout: EpicWideLayout, device):
    cache = layout.__dict__.setdefault("_device_maps", {})
    key = str(device)
    if key not in cache:
        cache[key] = (torch.from_numpy(layout.index_map).to(device), torch.from_numpy(layout.grad_pos).to(device))
    return cache[key]


def pack_blob_from_source(layout: EpicWideLayout, src: torch.Tensor) -> torch.Tensor:
    return src.detach()[_maps(layout, src.device)[0]]


class EpicWideFMLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, layout, x, t, a, eps, cond, mask, sigma, kind):
        blob = pack_blob_from_source(layout, src)
        sums, saved = hip_ops_wide.ew_fm_loss_forward(layout, blob, x, t, a, cond, mask, sigma, kind, eps)
        ctx.layout, ctx.saved, ctx.blob, ctx.n_source = layout, saved, blob, src.numel()
        ctx.inv = 1.0 / sums[1]
        return sums[0] * ctx.inv

    @staticmethod
    def backward(ctx, grad_out):
        lay = ctx.layout
        gblob = hip_ops_wide.ew_fm_loss_backward(lay, ctx.blob, ctx.saved, grad_out * ctx.inv)
        gpos = _maps(lay, gblob.device)[1]
        d_src = torch.zeros(ctx.n_source, device=gblob.device, dtype=torch.float32)
        d_src[: gpos.numel()] = gblob[gpos]
        return (d_src,) + (None,) * 9


def epic_wide_fm_loss(layout: EpicWideLayout, src: torch.Tensor, x, t, a, cond=None, mask=None, sigma: float = 1e-4,
                      kind: str = "FM-OT", eps: Optional[torch.Tensor] = None) -> torch.Tensor:
    return EpicWideFMLossFn.apply(src, layout, x, t, a, eps, cond, mask, float(sigma), kind)


class EpicWideDiffusionLossFn(torch.autograd.Function):
    """DiffusionLoss (losses.py:207-290) on the row-matrix path, draws given (see fm_loss.EpicDiffusionLossFn)."""

    @staticmethod
    def forward(ctx, src, layout, x, t, z, rates, jet_w, cond, mask, criterion):
        blob = pack_blob_from_source(layout, src)
        sums, saved = hip_ops_wide.ew_diffusion_loss_forward(layout, blob, x, t, z, rates, jet_w, cond, mask, criterion)
        ctx.layout, ctx.saved, ctx.blob, ctx.n_source, ctx.criterion, ctx.jet_w = layout, saved, blob, src.numel(), criterion, jet_w
        ctx.inv = 1.0 / sums[1]
        return sums[0] * ctx.inv

    @staticmethod
    def backward(ctx, grad_out):
        lay = ctx.layout
        gblob = hip_ops_wide.ew_fm_loss_backward(lay, ctx.blob, ctx.saved, grad_out * ctx.inv, ctx.criterion, ctx.jet_w)
        gpos = _maps(lay, gblob.device)[1]
        d_src = torch.zeros(ctx.n_source, device=gblob.device, dtype=torch.float32)
        d_src[: gpos.numel()] = gblob[gpos]
        return (d_src,) + (None,) * 9


def epic_wide_diffusion_loss(layout: EpicWideLayout, src, x, t, z, cond=None, mask=None, criterion: str = "huber",
                             diff_config=None) -> torch.Tensor:
    """z must already be multiplied by the mask (losses.py:244).  diff_config: {"max_sr", "min_sr"} of VPDiffusionSchedule."""
    from .fm_loss import MLE_LOSS_WEIGHT
    from .hip_ops import diffusion_schedule
    dc = dict(diff_config or {"max_sr": 1, "min_sr": 1e-8})
    sr, nr, beta = diffusion_schedule(t.to(torch.float32), **dc)
    rates = torch.stack([sr, nr], dim=1).contiguous()
    jet_w = (1.0 + MLE_LOSS_WEIGHT * (beta / nr)).contiguous()
    return EpicWideDiffusionLossFn.apply(src, layout, x, t, z, rates, jet_w, cond, mask, criterion)
