"""Fully-connected flow matching (no sets): drop-in for particle_fm/models/flow_matching_no_sets.py.

BASELINE config 1 ("FC FM on 2-D two-moons, CPU, batch 512") -- plumbing, deliberately plain PyTorch on the CPU
(SURVEY.md §8 row a13).  Same classes / keywords / state_dict keys as the reference (:16-38 ode_wrapper, :41-113 CNF,
:116-238 FLowMatchingNoSetsLitModule); the FM loss is the non-set branch of losses.py:38-77 (t per sample, mask of
ones) and the sampler is the same fixed-step midpoint the set model uses (torchdyn restated, hip_ops.midpoint_grid).
"""
from __future__ import annotations

import torch
from torch import nn

from ..hip_ops import midpoint_grid
from .components.mlp import small_cond_MLP_model
from .flow_matching_module import _LitBase


class ode_wrapper(nn.Module):
    def __init__(self, model: nn.Module, mask: torch.Tensor = None, cond: torch.Tensor = None):
        super().__init__()
        self.model, self.mask, self.cond = model, mask, cond

    def forward(self, t, x, *args, **kwargs):
        return self.model(t, x, mask=self.mask, cond=self.cond)


class CNF(nn.Module):
    def __init__(self, features: int, freqs: int = 3, activation: str = "Tanh"):
        super().__init__()
        self.net = small_cond_MLP_model(features, features, dim_t=2 * freqs, dim_cond=1, activation=activation)
        self.register_buffer("freqs", torch.arange(1, freqs + 1) * torch.pi)

    def forward(self, t: torch.Tensor, x: torch.Tensor, mask: torch.Tensor = None, cond: torch.Tensor = None):
        t = self.freqs * t[..., None]  # flow_matching_no_sets.py:62-64
        t = torch.cat((t.cos(), t.sin()), dim=-1)
        t = t.expand(*x.shape[:-1], -1)
        return self.net(t, x, cond=cond)

    def decode(self, z: torch.Tensor, cond: torch.Tensor, mask: torch.Tensor = None, ode_solver: str = "midpoint",
               ode_steps: int = 100) -> torch.Tensor:
        if ode_solver != "midpoint":
            raise NotImplementedError(f"Solver {ode_solver} not implemented")  # :93
        ts, dts = midpoint_grid(ode_steps)
        x = z
        for k in range(ode_steps - 1):  # Midpoint.step: k1 = f(t,x); x <- x + dt f(t + dt/2, x + dt/2 k1)
            t0, tm, dt = ts[2 * k].to(z.device), ts[2 * k + 1].to(z.device), dts[k].to(z.device)
            k1 = self(t0, x, mask=mask, cond=cond)
            x = x + dt * self(tm, x + 0.5 * dt * k1, mask=mask, cond=cond)
        return x

    def encode(self, *a, **k):
        raise NotImplementedError("CNF.encode is not implemented (the reference's own version is not callable either)")


def fm_loss_no_sets(flow: nn.Module, x: torch.Tensor, cond, t: torch.Tensor, z: torch.Tensor, sigma: float):
    """losses.py:41-77, non-set branch: t (B,1) per sample, mask = ones -> loss = sum((v-u)^2) / B."""
    y = (1 - t) * x + (sigma + (1 - sigma) * t) * z
    u = (1 - sigma) * z - x
    v = flow(t.squeeze(-1), y, mask=None, cond=cond)
    return (v - u).square().sum() / x.shape[0]


class FLowMatchingNoSetsLitModule(_LitBase):
    def __init__(self, optimizer: torch.optim.Optimizer = None, scheduler: torch.optim.lr_scheduler = None,
                 features: int = 10, n_transforms: int = 1, sigma: float = 1e-4, activation: str = "ELU", freqs: int = 3):
        super().__init__()
        self.save_hyperparameters(logger=False)
        if n_transforms != 1:
            raise NotImplementedError("n_transforms > 1")
        self.flows = nn.ModuleList([CNF(features, freqs=freqs, activation=activation)])
        self.sigma = sigma

    def loss(self, x: torch.Tensor, cond: torch.Tensor = None) -> torch.Tensor:
        t = torch.rand_like(x[..., 0]).unsqueeze(-1)  # losses.py:49
        z = torch.randn_like(x)                       # losses.py:53
        return fm_loss_no_sets(self.flows[0], x, cond, t, z, self.sigma)

    def forward(self, x, cond=None, mask=None, reverse: bool = False, ode_solver: str = "midpoint", ode_steps: int = 100):
        if not reverse:
            raise NotImplementedError("forward (encode) direction")
        for f in reversed(self.flows):
            x = f.decode(x, cond, mask, ode_solver=ode_solver, ode_steps=ode_steps)
        return x

    def training_step(self, batch, batch_idx):
        x, mask, cond = batch
        loss = self.loss(x, cond=cond)
        self.log("train/loss", loss, on_step=False, on_epoch=True, prog_bar=True)
        return {"loss": loss}

    def on_validation_epoch_start(self) -> None:
        torch.manual_seed(9999)

    def on_validation_epoch_end(self) -> None:
        torch.manual_seed(torch.seed())

    def validation_step(self, batch, batch_idx: int):
        x, mask, cond = batch
        loss = self.loss(x, cond=cond)
        self.log("val/loss", loss, on_step=False, on_epoch=True, prog_bar=True)
        return {"loss": loss}

    def configure_optimizers(self):
        optimizer = self.hparams.optimizer(params=self.parameters())
        if self.hparams.scheduler is not None:
            return {"optimizer": optimizer, "lr_scheduler": {"scheduler": self.hparams.scheduler(optimizer=optimizer),
                                                             "monitor": "val/loss", "interval": "epoch", "frequency": 1}}
        return {"optimizer": optimizer}

    @torch.no_grad()
    def sample(self, n_samples: int, mask=None, cond=None, ode_solver: str = "midpoint", ode_steps: int = 100):
        z = torch.randn(n_samples, self.hparams.features).to(self.device)
        if cond is not None:
            cond = cond.to(self.device)
        return self.forward(z, cond=cond, mask=mask, reverse=True, ode_solver=ode_solver, ode_steps=ode_steps)
