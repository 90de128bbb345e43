"""Flow-matching losses of the cross-attention field as one autograd node over the flat parameter vector
(same scheme as fm_loss_tf.py).  Reference: losses.py:38-77, 101-136, 139-176."""
from __future__ import annotations

from typing import Optional

import torch

from . import hip_ops_ca
from .layout_ca import CaLayout, default_freqs


class CaFMLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat_params, layout: CaLayout, freqs, x, t, a, eps, cond, mask, sigma, kind):
        dev = x.device
        src = torch.cat([flat_params.to(torch.float32), freqs.to(dev, torch.float32), torch.zeros(1, device=dev)])
        blob = src[layout.index_map_on(dev)]
        sums, saved = hip_ops_ca.ca_fm_loss_forward(layout, blob, x, t, a, cond, mask, sigma, kind, eps)
        ctx.layout, ctx.saved, ctx.blob, ctx.cond = layout, saved, blob, cond
        ctx.mask = None if mask is None else mask.reshape(x.shape[0], -1).to(torch.float32).contiguous()
        ctx.inv = 1.0 / sums[1]
        return sums[0] * ctx.inv

    @staticmethod
    def backward(ctx, grad_out):
        lay = ctx.layout
        gblob = hip_ops_ca.ca_fm_loss_backward(lay, ctx.blob, ctx.cond, ctx.mask, ctx.saved, grad_out * ctx.inv)
        return (gblob[lay.grad_pos_on(gblob.device)],) + (None,) * 10


def ca_fm_loss(layout: CaLayout, flat_params: torch.Tensor, x, t, a, cond=None, mask=None, sigma: float = 1e-4,
               kind: str = "FM-OT", eps: Optional[torch.Tensor] = None, freqs: Optional[torch.Tensor] = None):
    """flat_params: concatenation of the parameters in layout.keys() order (requires_grad as the caller wishes)."""
    f = default_freqs(layout.cfg.t_dim, layout.cfg.t_emb) if freqs is None else freqs
    if layout.cfg.t_emb == "sincos" and f.numel() == layout.cfg.frequencies:
        f = torch.cat([f, f])
    return CaFMLossFn.apply(flat_params, layout, f, x, t, a, eps, cond, mask, float(sigma), kind)
