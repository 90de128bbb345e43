"""Flow-matching losses of the Full-Transformer field as one autograd node over the flat parameter vector.

forward : blob = source[index_map] (gather), then pfm_tf_fm_loss_forward  -> loss = sum (v-u)^2 / sum mask
backward: pfm_tf_fm_loss_backward fills a gradient blob; every parameter element reads its own slot
          (layout.grad_pos), so d loss / d params is one gather.
Reference: FlowMatchingLoss.forward / ConditionalFlowMatchingLoss.forward, losses.py:38-77, 101-136."""
from __future__ import annotations

from typing import Optional

import torch

from . import hip_ops_tf
from .layout_tf import TfLayout


class TfFMLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat_params, layout: TfLayout, freqs, x, t, a, eps, cond, mask, sigma, kind):
        dev = x.device
        src = torch.cat([flat_params.to(torch.float32), freqs.to(dev, torch.float32), torch.zeros(1, device=dev)])
        blob = src[layout.index_map_on(dev)]
        sums, saved = hip_ops_tf.tf_fm_loss_forward(layout, blob, x, t, a, cond, mask, sigma, kind, eps)
        ctx.layout, ctx.saved, ctx.blob = layout, saved, blob
        ctx.t, ctx.cond = t, cond
        ctx.mask = None if mask is None else mask.reshape(x.shape[0], -1).to(torch.float32).contiguous()
        ctx.inv = 1.0 / sums[1]
        return sums[0] * ctx.inv

    @staticmethod
    def backward(ctx, grad_out):
        lay = ctx.layout
        gblob = hip_ops_tf.tf_fm_loss_backward(lay, ctx.blob, ctx.t, ctx.cond, ctx.mask, ctx.saved, grad_out * ctx.inv)
        g = gblob[lay.grad_pos_on(gblob.device)]
        return (g,) + (None,) * 10


def tf_fm_loss(layout: TfLayout, flat_params: torch.Tensor, x, t, a, cond=None, mask=None, sigma: float = 1e-4,
               kind: str = "FM-OT", eps: Optional[torch.Tensor] = None, freqs: Optional[torch.Tensor] = None):
    """flat_params: concatenation of the parameters in layout.keys() order (requires_grad as the caller wishes)."""
    from .layout_tf import default_freqs

    f = default_freqs(layout.cfg.t_dim, layout.cfg.t_emb) if freqs is None else freqs
    if layout.cfg.t_emb == "sincos" and f.numel() == layout.cfg.frequencies:
        f = torch.cat([f, f])
    return TfFMLossFn.apply(flat_params, layout, f, x, t, a, eps, cond, mask, float(sigma), kind)
