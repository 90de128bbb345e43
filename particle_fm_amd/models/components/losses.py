"""Flow-matching losses behind the reference's class names.

Mirrors particle_fm/models/components/losses.py:16-136: ``FlowMatchingLoss`` ("FM-OT") and
``ConditionalFlowMatchingLoss`` ("CFM") with the constructor ``(flows, sigma, criterion)`` and
``forward(x, mask=None, cond=None) -> scalar``.  The random draws are made exactly where and how the
reference makes them (t from the CPU generator via ``rand_like(ones(B))``, z / x0 / eps by ``randn_like`` on
x's device, in the same order), then the whole loss -- interpolation, network, masked MSE -- runs in one HIP
launch (forward) and one (backward).
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _to_device(t_cpu: torch.Tensor, x: torch.Tensor, land=None) -> torch.Tensor:
    if land is not None and x.is_cuda and x.dtype == torch.float32:
        return land(t_cpu)
    return t_cpu.type_as(x)


def _chained_diffusion_loss(flows, x, t, z, mask, cond, criterion: str, diff_config) -> torch.Tensor:
    """DiffusionLoss with n_transforms > 1 (losses.py:264-267: the noisy particles pass through every flow at the same t; the last
    output is the predicted noise): the flows as differentiable fields (see _chained_loss), rates / criterion / weights as device ops."""
    from ... import fm_field, hip_ops
    dc = dict(diff_config)
    tt = t.to(x.device, torch.float32)
    sr, nr, _ = hip_ops.diffusion_schedule(tt, **dc)
    temp = sr.view(-1, 1, 1) * x + nr.view(-1, 1, 1) * z
    for f in flows:
        temp = f.field(tt, temp, cond=cond, mask=mask)
    return fm_field.diffusion_loss_from_field(temp, z, mask, tt, criterion, dc)


def _chained_loss(flows, kind: str, x, t, a, eps, mask, cond, sigma: float) -> torch.Tensor:
    """n_transforms > 1 (flow_matching_module.py:421-443): the reference feeds each flow's output to the next one at the SAME time t
    (losses.py:66-69, 125-128, 337-340) and compares the last output with the target.  No fused loss kernel for a chain: every flow is
    the differentiable field of fm_field.py (one HIP forward with saved activations; its backward returns the gradient w.r.t. the
    parameters AND w.r.t. the particle input, which the flow in front of it consumes); interpolation, target and the squared error
    are element-wise device ops in the reference's own expressions."""
    from ... import fm_field

    def chain(y):
        temp = y
        for f in flows:
            temp = f.field(t, temp, cond=cond, mask=mask)
        return temp

    return fm_field.fm_loss_from_field(chain, kind, x, t, a, eps, mask, sigma)


class FlowMatchingLoss(nn.Module):
    def __init__(self, flows: nn.ModuleList, sigma: float = 1e-4, criterion: str = "mse"):
        super().__init__()
        self.flows = flows
        self.sigma = sigma
        # losses.py:31-36 builds MSELoss / HuberLoss, but forward (:74-76) never calls it: the loss is the
        # squared error for both spellings.  Anything else raises, as in the reference.
        if criterion not in ("mse", "huber"):
            raise NotImplementedError(f"criterion {criterion} not supported")
        self.criterion = criterion

    def draw(self, x: torch.Tensor, land=None):
        """losses.py:46-53: t ~ U(0,1) per jet from the CPU generator, z ~ N(0,1) like x.  ``land`` (optional): how the host tensor
        reaches x's device (the fused trainer passes an asynchronous pinned-memory copy; default: ``.type_as(x)`` like the reference,
        which synchronises the stream)."""
        t = _to_device(torch.rand_like(torch.ones(x.shape[0])), x, land)
        z = torch.randn_like(x)
        return t, z

    def forward(self, x: torch.Tensor, mask: torch.Tensor = None, cond: torch.Tensor = None) -> torch.Tensor:
        if x.dim() != 3:
            raise NotImplementedError("the HIP loss handles set data (B, N, F)")
        t, z = self.draw(x)
        if len(self.flows) > 1:
            return _chained_loss(self.flows, "FM-OT", x, t, z, None, mask, cond, self.sigma)
        return self.flows[0].fm_loss(x, t, z, mask=mask, cond=cond, sigma=self.sigma, kind="FM-OT")


class ConditionalFlowMatchingLoss(nn.Module):
    def __init__(self, flows: nn.ModuleList, sigma: float = 1e-4, criterion: str = "mse"):
        super().__init__()
        self.flows = flows
        self.sigma = sigma
        if criterion not in ("mse",):
            raise NotImplementedError(f"criterion {criterion} not supported on the HIP path")

    def draw(self, x: torch.Tensor, land=None):
        """losses.py:104, 108, 116: t, x_0, then the noise added to mu_t."""
        t = _to_device(torch.rand_like(torch.ones(x.shape[0])), x, land)
        x0 = torch.randn_like(x)
        eps = torch.randn_like(x)
        return t, x0, eps

    def forward(self, x: torch.Tensor, mask: torch.Tensor = None, cond: torch.Tensor = None) -> torch.Tensor:
        if mask is None:
            raise TypeError("ConditionalFlowMatchingLoss needs a mask (losses.py:119 multiplies by it)")
        t, x0, eps = self.draw(x)
        if len(self.flows) > 1:
            return _chained_loss(self.flows, "CFM", x, t, x0, eps, mask, cond, self.sigma)
        return self.flows[0].fm_loss(x, t, x0, mask=mask, cond=cond, sigma=self.sigma, kind="CFM", eps=eps)


class DroidLoss(nn.Module):
    """losses.py:304-342: y = x + t z, target u = z * mask, sum of squares / sum(mask) (the criterion object is built but,
    as in the reference, never called)."""

    def __init__(self, flows: nn.ModuleList, sigma: float = 1e-4, criterion: str = "mse"):
        super().__init__()
        self.flows = flows
        self.sigma = sigma
        if criterion not in ("mse", "huber"):
            raise NotImplementedError(f"criterion {criterion} not supported")

    def draw(self, x: torch.Tensor, land=None):
        t = _to_device(torch.rand_like(torch.ones(x.shape[0])), x, land)  # :330
        z = torch.randn_like(x)                                            # :335
        return t, z

    def forward(self, x: torch.Tensor, mask: torch.Tensor = None, cond: torch.Tensor = None) -> torch.Tensor:
        if mask is None:
            raise TypeError("DroidLoss needs a mask (losses.py:339 multiplies by it)")
        t, z = self.draw(x)
        if len(self.flows) > 1:
            return _chained_loss(self.flows, "droid", x, t, z, None, mask, cond, self.sigma)
        return self.flows[0].fm_loss(x, t, z, mask=mask, cond=cond, sigma=self.sigma, kind="droid")


class DiffusionLoss(nn.Module):
    """losses.py:207-290 (PC-JeDi style noise prediction): noisy = signal_rate(t) x + noise_rate(t) z with z = randn * mask,
    loss = sum criterion(z, net(t, noisy)) * mask * (1 + 0.001 beta(t) / noise_rate(t)) / sum(mask)."""

    def __init__(self, flows: nn.ModuleList, sigma: float = 1e-4, criterion: str = "huber",
                 diff_config={"max_sr": 1, "min_sr": 1e-8}):
        super().__init__()
        self.flows = flows
        self.sigma = sigma
        self.mle_loss_weight = 0.001
        self.diff_config = dict(diff_config)
        if criterion not in ("mse", "huber"):
            raise NotImplementedError(f"criterion {criterion} not supported")
        self.criterion = criterion

    def draw(self, x: torch.Tensor, mask: torch.Tensor, land=None):
        t = _to_device(torch.rand_like(torch.ones(x.shape[0])), x, land)  # :241-243
        z = torch.randn_like(x) * mask                                     # :247
        return t, z

    def forward(self, x: torch.Tensor, mask: torch.Tensor = None, cond: torch.Tensor = None) -> torch.Tensor:
        if mask is None:
            raise TypeError("DiffusionLoss needs a mask (losses.py:247 multiplies the noise by it)")
        t, z = self.draw(x, mask)
        if len(self.flows) > 1:
            return _chained_diffusion_loss(self.flows, x, t, z, mask, cond, self.criterion, self.diff_config)
        return self.flows[0].diffusion_loss(x, t, z, mask=mask, cond=cond, criterion=self.criterion, diff_config=self.diff_config)
