"""Flow-matching losses of the MDMA field as one autograd node over the flat parameter vector (same scheme as fm_loss_tf.py).
Reference: losses.py:38-77, 101-136, 304-342 with the (B, N, 1) field broadcast over the features."""
from __future__ import annotations

from typing import Optional

import torch

from . import hip_ops_mdma
from .layout_mdma import MdmaLayout, default_freqs


class MdmaFMLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat_params, layout: MdmaLayout, freqs, x, t, a, eps, mask, sigma, kind, cond=None):
        dev = x.device
        src = torch.cat([flat_params.to(torch.float32), freqs.to(dev, torch.float32), torch.zeros(1, device=dev)])
        blob = src[layout.index_map_on(dev)]
        sums, saved = hip_ops_mdma.mdma_fm_loss_forward(layout, blob, x, t, a, mask, sigma, kind, eps, cond=cond)
        ctx.layout, ctx.saved, ctx.blob = layout, saved, blob
        ctx.inv = 1.0 / sums[1]
        return sums[0] * ctx.inv

    @staticmethod
    def backward(ctx, grad_out):
        lay = ctx.layout
        gblob = hip_ops_mdma.mdma_fm_loss_backward(lay, ctx.blob, ctx.saved, grad_out * ctx.inv)
        return (gblob[lay.grad_pos_on(gblob.device)],) + (None,) * 10


def mdma_fm_loss(layout: MdmaLayout, flat_params: torch.Tensor, x, t, a, mask, sigma: float = 1e-4, kind: str = "FM-OT",
                 eps: Optional[torch.Tensor] = None, freqs: Optional[torch.Tensor] = None, cond=None):
    """flat_params: concatenation of the parameters in layout.keys() order (requires_grad as the caller wishes); cond: the conditional
    variant's one value per jet (layout.cfg.needs_cond)."""
    f = default_freqs(layout.cfg.t_dim, layout.cfg.t_emb) if freqs is None else freqs
    if layout.cfg.t_emb == "sincos" and f.numel() == layout.cfg.frequencies:
        f = torch.cat([f, f])
    return MdmaFMLossFn.apply(flat_params, layout, f, x, t, a, eps, mask, float(sigma), kind, cond)
