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
