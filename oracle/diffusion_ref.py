"""CPU restatement of the reference's diffusion objective and samplers (loss_type="diffusion", configs/model/diffusion.yaml).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PINNED: checked against vectors recorded from the reference's own
DiffusionLoss / ode_wrapper / ddim_sampler / euler_maruyama_sampler (tests/golden/epic_diffusion.npz, oracle/make_golden.py).

Follows:
  * models/components/diffusion.py:9-62      VPDiffusionSchedule (cosine signal / noise rates, betas)
  * models/components/losses.py:207-290      DiffusionLoss.forward (noise prediction, "none"-reduced criterion * mask, MLE weight)
  * models/flow_matching_module.py:62-71     ode_wrapper.forward for loss_type="diffusion" (probability-flow ODE right-hand side)
  * models/components/solver.py:12-143       ddim_predict, ddim_sampler, euler_maruyama_sampler
"""
from __future__ import annotations

import math
from typing import Callable, Mapping, Sequence

import torch
import torch.nn.functional as F

MLE_LOSS_WEIGHT = 0.001  # losses.py:226


def schedule(t: torch.Tensor, max_sr: float = 1.0, min_sr: float = 1e-2):
    """diffusion.py:21-52: (signal_rate, noise_rate) = (cos, sin)(start + t (end - start))."""
    start, end = math.acos(max_sr), math.acos(min_sr)
    ang = start + t * (end - start)
    return torch.cos(ang), torch.sin(ang)


def betas(t: torch.Tensor, max_sr: float = 1.0, min_sr: float = 1e-2):
    """diffusion.py:55-62."""
    start, end = math.acos(max_sr), math.acos(min_sr)
    return 2 * (end - start) * torch.tan(start + t * (end - start))


def diffusion_loss(vf: Callable, x, mask, cond, t, z, criterion: str = "huber", diff_config: Mapping = None):
    """DiffusionLoss.forward with the draws given: t (B,) uniform, z (B,N,F) ALREADY multiplied by the mask (losses.py:244).
    Returns (loss, noisy input, predicted noise)."""
    dc = dict(diff_config or {"max_sr": 1, "min_sr": 1e-8})
    tt = t.unsqueeze(-1).repeat_interleave(x.shape[1], dim=1)  # (B,N), what the network is called with (:267)
    sr, nr = schedule(t.view(-1, 1, 1), **dc)
    noisy = sr * x + nr * z  # :260
    pred = vf(tt, noisy, mask=mask, cond=cond)
    crit = F.mse_loss if criterion == "mse" else F.huber_loss
    simple = crit(z, pred, reduction="none") * mask  # :272 (criterion(noises, pred_noises))
    w = betas(t.view(-1, 1, 1), **dc) / nr  # :276-277
    return simple.sum() / mask.sum() + MLE_LOSS_WEIGHT * (w * simple).sum() / mask.sum(), noisy, pred


def diffusion_rhs(vf: Callable, t, x, cond, mask, diff_config: Mapping):
    """ode_wrapper.forward (flow_matching_module.py:62-69): -0.5 beta(t) (x - model(t, x) / noise_rate(t)), t 0-dim."""
    shape = [-1] + [1] * (x.dim() - 1)
    _, nr = schedule(t.view(shape), **diff_config)
    b = betas(t.view(shape), **diff_config)
    return -0.5 * b * (x - vf(t, x, mask=mask, cond=cond) / nr)


def ddim_sample(vf: Callable, z, cond, mask, n_steps: int, diff_config: Mapping):
    """ddim_sampler (solver.py:22-96) -> the final predicted data."""
    B = z.shape[0]
    shape = [-1] + [1] * (z.dim() - 1)
    step = 1 / n_steps
    noisy = z
    times = torch.ones(B)
    nsr, nnr = schedule(times.view(shape), **diff_config)
    pred_data = None
    with torch.no_grad():
        for _ in range(n_steps):
            sr, nr = nsr, nnr
            pred = vf(times[0], noisy, mask=mask, cond=cond)
            pred_data = (noisy - nr * pred) / sr
            times = times - step
            nsr, nnr = schedule(times.view(shape), **diff_config)
            noisy = nsr * pred_data + nnr * pred
    return pred_data


def em_sample(vf: Callable, z, cond, mask, n_steps: int, diff_config: Mapping, noises: Sequence[torch.Tensor]):
    """euler_maruyama_sampler (solver.py:99-143) with the per-step normal draws given."""
    B = z.shape[0]
    shape = [-1] + [1] * (z.dim() - 1)
    delta = 1 / n_steps
    x = z.clone()
    t = torch.ones(B)
    with torch.no_grad():
        for k in range(n_steps):
            pred = vf(t[0], x, mask=mask, cond=cond)
            _, nr = schedule(t.view(shape), **diff_config)
            s = -pred / nr
            b = betas(t.view(shape), **diff_config)
            x = x + 0.5 * b * (x + 2 * s) * delta
            x = x + (b * delta).sqrt() * noises[k]
            t = t - delta
    return x
