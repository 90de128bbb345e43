"""Eager-PyTorch CPU restatement of the flow-matching wrapper around the network.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Follows:
  * particle_fm/models/components/time_emb.py:49-96     (cosine_encoding)
  * particle_fm/models/flow_matching_module.py:191-233  (CNF.forward / time_embedding)
  * particle_fm/models/flow_matching_module.py:62-71    (ode_wrapper.forward, FM branch)
  * particle_fm/models/flow_matching_module.py:245-259, 283-287 (CNF.decode, "midpoint")
  * particle_fm/models/flow_matching_module.py:637-677  (SetFlowMatchingLitModule.sample)
  * particle_fm/models/components/losses.py:38-77       (FlowMatchingLoss.forward)
  * particle_fm/models/components/losses.py:101-136     (ConditionalFlowMatchingLoss.forward)
  * torchdyn (requirements.txt:25, unpinned, NOT vendored): fixed-step ``odeint`` driver and
    ``Midpoint.step`` restated from the published algorithm of torchdyn 1.0.x:
        k1 = f(t, x); x_mid = x + 0.5*dt*k1; x_new = x + dt * f(t + 0.5*dt, x_mid)
        t <- t + dt; dt <- t_span[k+1] - t          (t, dt 0-dim fp32 from torch.linspace)
"""
from __future__ import annotations

import math
from typing import Callable, Mapping, Optional

import torch

from .epic_ref import epic_encoder


def cosine_encoding(x: torch.Tensor, outp_dim: int = 32, min_value: float = 0.0, max_value: float = 1.0,
                    freqs: Optional[torch.Tensor] = None):
    """time_emb.py:79-96, exponential frequencies; exact fp32 op order
    ``((x + min) * exp(arange(D))) * pi / (max + min)``.

    ``freqs`` overrides ``torch.arange(D).exp()``: that fp32 ``exp`` is NOT the same on every host
    (measured: element 15 differs by 1 ulp between an Intel Xeon and an AMD EPYC build of the same
    torch), and because the arguments reach 1e13 a 1-ulp change of a frequency changes cos() by O(1).
    Golden vectors carry the table of the machine that recorded them."""
    if x.shape[-1] != 1 or x.dim() == 1:
        x = x.unsqueeze(-1)
    if freqs is None:
        freqs = torch.arange(outp_dim, device=x.device).exp()
    return torch.cos((x + min_value) * freqs * math.pi / (max_value + min_value))


def time_embedding_cosine(t: torch.Tensor, x: torch.Tensor, t_dim: int, freqs=None) -> torch.Tensor:
    """flow_matching_module.py:223-228.  t is (B,N) in training, 0-dim in sampling."""
    if t.dim() == 0:
        t = t.unsqueeze(0)
    emb = cosine_encoding(t, t_dim, freqs=freqs)
    return emb.expand(*x.shape[:-1], -1)


def gaussian_time_embedding(t: torch.Tensor, x: torch.Tensor, state: Mapping[str, torch.Tensor], cnf_prefix: str,
                            activation: str = "leaky_relu") -> torch.Tensor:
    """t_emb="gaussian" (flow_matching_module.py:178-181, 213-221): GaussianFourierProjection (time_emb.py:9-22: frozen W,
    x_proj = t W 2 pi, cat(sin, cos)) -> Linear -> activation -> Linear(2 * frequencies), one row per jet, expanded over particles.
    The four Linear tensors are trainable parameters of the CNF (``embed.1.*``, ``linear.*``), W is ``embed.0.W``."""
    if t.dim() == 2:
        t = t[:, 0]  # :217-218 "different shape for training"
    W = state[cnf_prefix + "embed.0.W"]
    x_proj = t[..., None] * W[None, ...] * 2 * math.pi
    e = torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)
    e = torch.nn.functional.linear(e, state[cnf_prefix + "embed.1.weight"], state[cnf_prefix + "embed.1.bias"])
    e = getattr(torch.nn.functional, activation, lambda v: v)(e)
    e = torch.nn.functional.linear(e, state[cnf_prefix + "linear.weight"], state[cnf_prefix + "linear.bias"]).unsqueeze(1)
    return e.expand(*x.shape[:-1], -1)


def time_embedding(t: torch.Tensor, x: torch.Tensor, hp: Mapping, freqs=None) -> torch.Tensor:
    """CNF.time_embedding (flow_matching_module.py:206-233) for t_emb in {"cosine", "sincos"}.
    sincos (:208-211): t = frequencies * t[..., None]; cat(cos, sin); frequencies = 2**arange(F) * pi (:172) -- pass the
    module's buffer as ``freqs`` (F values) or leave None to rebuild it."""
    kind = hp.get("t_emb", "cosine")
    if kind == "cosine":
        return time_embedding_cosine(t, x, 2 * hp["frequencies"], freqs)
    if kind != "sincos":
        raise NotImplementedError(kind)
    f = 2 ** torch.arange(hp["frequencies"]) * torch.pi if freqs is None else freqs[: hp["frequencies"]]  # ([f ; f] tables too)
    a = f * t[..., None]
    return torch.cat((a.cos(), a.sin()), dim=-1).expand(*x.shape[:-1], -1)


class EpicVectorField:
    """CNF.forward for model="epic", t_emb="cosine" (flow_matching_module.py:191-204)."""

    def __init__(self, state: Mapping[str, torch.Tensor], prefix: str, hp: Mapping, freqs=None):
        self.state = state
        self.prefix = prefix
        self.hp = dict(hp)
        self.freqs = freqs

    def __call__(self, t, x, cond=None, mask=None):
        hp = self.hp
        t_dim = 2 * hp["frequencies"]
        if hp.get("t_emb", "cosine") == "gaussian":
            cnf_prefix = self.prefix[: -len("net")] if self.prefix.endswith("net") else self.prefix + "."
            temb = gaussian_time_embedding(t, x, self.state, cnf_prefix, hp.get("activation", "leaky_relu"))
        else:
            temb = time_embedding(t, x, hp, self.freqs)
        if hp.get("add_time_to_input", False):
            x = torch.cat((temb, x), dim=-1)  # :199-200
        return epic_encoder(
            self.state,
            self.prefix,
            temb,
            x,
            cond,
            mask,
            layers=hp["layers"],
            t_local_cat=hp.get("t_local_cat", True),
            t_global_cat=hp.get("t_global_cat", True),
            global_cond_dim=hp.get("global_cond_dim", 0),
            local_cond_dim=hp.get("local_cond_dim", 0),
            sum_scale=hp.get("sum_scale", 1e-2),
            activation=hp.get("activation", "leaky_relu"),
        )


def fm_ot_targets(x, mask, t, z, sigma: float):
    """losses.py:41-62 with the random draws (t per jet, z) made explicit."""
    if mask is None:
        mask = torch.ones_like(x[..., 0]).unsqueeze(-1)
    tt = t.unsqueeze(-1).repeat_interleave(x.shape[1], dim=1).unsqueeze(-1).type_as(x)  # :47,50
    y = (1 - tt) * x + (sigma + (1 - sigma) * tt) * z  # :56
    u_t = ((1 - sigma) * z - x) * mask  # :61-62
    return tt, y, u_t, mask


def fm_ot_loss(vf: Callable, x, mask, cond, t, z, sigma: float = 1e-4):
    """FlowMatchingLoss.forward (losses.py:38-77) for set data, draws given.
    Returns (loss, y, u_t, v_t)."""
    tt, y, u_t, m = fm_ot_targets(x, mask, t, z, sigma)
    v_t = vf(tt.squeeze(-1), y, mask=m, cond=cond)  # :66-69 (single flow)
    loss = (v_t - u_t).square().sum() / m.sum()  # :75-76
    return loss, y, u_t, v_t


def cfm_loss(vf: Callable, x, mask, cond, t, x0, eps, sigma: float = 1e-4):
    """ConditionalFlowMatchingLoss.forward (losses.py:101-136), draws given
    (x0 = prior sample, eps = the second randn_like).  mask must not be None (:119)."""
    tt = t.unsqueeze(-1).repeat_interleave(x.shape[1], dim=1).unsqueeze(-1).type_as(x)
    mu_t = (1 - tt) * x + tt * x0
    y = mu_t + sigma * eps
    u_t = (x0 - x) * mask
    v_t = vf(tt.squeeze(-1), y, mask=mask, cond=cond)
    loss = torch.nn.functional.mse_loss(v_t, u_t, reduction="sum") / mask.sum()
    return loss, y, u_t, v_t


def droid_loss(vf: Callable, x, mask, cond, t, z):
    """DroidLoss.forward (losses.py:326-342), draws given: y = x + t z, u = z mask."""
    tt = t.unsqueeze(-1).repeat_interleave(x.shape[1], dim=1).unsqueeze(-1).type_as(x)
    y = x + tt * z
    u_t = z * mask
    v_t = vf(tt.squeeze(-1), y, mask=mask, cond=cond)
    return (v_t - u_t).square().sum() / mask.sum(), y, u_t, v_t


def midpoint_trajectory_end(f: Callable, x: torch.Tensor, t_span: torch.Tensor) -> torch.Tensor:
    """Fixed-step explicit midpoint over ``t_span`` (torchdyn 1.0.x semantics, restated).
    Returns the final state only (the reference takes ``traj[-1]``, :285-287)."""
    t = t_span[0]
    dt = t_span[1] - t
    steps = len(t_span)
    for k in range(1, steps):
        k1 = f(t, x)
        x_mid = x + 0.5 * dt * k1
        x = x + dt * f(t + 0.5 * dt, x_mid)
        t = t + dt
        if k < steps - 1:
            dt = t_span[k + 1] - t
    return x


# torchdyn's fixed-step solvers as explicit Runge-Kutta tableaus (c, rows of a, b).  torchdyn is not vendored (requirements.txt:25):
# restated from its published solver steps -- Euler.step (x + dt*k1), Midpoint.step, RungeKutta4.step with construct_rk4 = the
# 3/8 rule -- and UNPINNED at the torchdyn boundary like the midpoint driver above.
RK_TABLEAUS = {
    "euler": ([0.0], [[]], [1.0]),
    "midpoint": ([0.0, 0.5], [[], [0.5]], [0.0, 1.0]),
    "rk4": ([0.0, 1 / 3, 2 / 3, 1.0], [[], [1 / 3], [-1 / 3, 1.0], [1.0, -1.0, 1.0]], [1 / 8, 3 / 8, 3 / 8, 1 / 8]),
}


def rk_trajectory_end(f: Callable, x: torch.Tensor, t_span: torch.Tensor, solver: str) -> torch.Tensor:
    """Fixed-step explicit Runge-Kutta over ``t_span`` with the driver of ``midpoint_trajectory_end``; a stage input is
    x + dt * (a_0 k_0 + a_1 k_1 + ...), the update x + dt * (b_0 k_0 + ...), fp32 tableau entries, sums left to right."""
    c, a, b = (RK_TABLEAUS[solver][0], RK_TABLEAUS[solver][1], RK_TABLEAUS[solver][2])
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)
    t = t_span[0]
    dt = t_span[1] - t
    steps = len(t_span)
    for k in range(1, steps):
        ks = []
        for s in range(len(b)):
            if s == 0:
                ks.append(f(t, x))
                continue
            acc = f32(a[s][0]) * ks[0]
            for j in range(1, s):
                acc = acc + f32(a[s][j]) * ks[j]
            ks.append(f(t + f32(c[s]) * dt, x + dt * acc))
        acc = f32(b[0]) * ks[0]
        for j in range(1, len(b)):
            acc = acc + f32(b[j]) * ks[j]
        x = x + dt * acc
        t = t + dt
        if k < steps - 1:
            dt = t_span[k + 1] - t
    return x


def sample_fixed_step(vf: Callable, z, cond, mask, ode_steps: int = 100, solver: str = "rk4", t0: float = 1.0, t1: float = 0.0):
    """CNF.decode with ode_solver "euler" / "rk4" / "midpoint" (flow_matching_module.py:261-287) behind sample's z * mask
    (:668-671); t0 = 0, t1 = 1, solver "rk4", ode_steps 100, cond None = CNF.encode (:235-243)."""
    if mask is not None:
        z = z * mask
    t_span = torch.linspace(t0, t1, ode_steps)
    with torch.no_grad():
        return rk_trajectory_end(lambda t, x: vf(t, x, mask=mask, cond=cond), z, t_span, solver)


def midpoint_time_grid(ode_steps: int):
    """The 2*(ode_steps-1) evaluation times (t_k, t_k + dt_k/2) and the dt_k the
    restated driver visits, as fp32 tensors -- the same arithmetic as above."""
    t_span = torch.linspace(1.0, 0.0, ode_steps)
    t = t_span[0]
    dt = t_span[1] - t
    ts, dts = [], []
    for k in range(1, ode_steps):
        ts.append(t.clone())
        ts.append(t + 0.5 * dt)
        dts.append(dt.clone())
        t = t + dt
        if k < ode_steps - 1:
            dt = t_span[k + 1] - t
    return torch.stack(ts), torch.stack(dts)


def sample_midpoint(vf: Callable, z, cond, mask, ode_steps: int = 100):
    """SetFlowMatchingLitModule.sample + CNF.decode("midpoint") with z given
    (flow_matching_module.py:659-674, 253-259, 283-287)."""
    if mask is not None:
        z = z * mask  # :668-671
    t_span = torch.linspace(1.0, 0.0, ode_steps)
    with torch.no_grad():
        return midpoint_trajectory_end(lambda t, x: vf(t, x, mask=mask, cond=cond), z, t_span)


def generate_epilogue(x, mask, normalized_data, normalize_sigma, means, stds, log_pt, pt_standardization, variable_set_sizes):
    """The per-batch post-processing of generate_data (utils/data_generation.py:94-123) on a CPU tensor, op for op:
    inverse_normalize_tensor (data/components/utils.py:183-199), log_pt through numpy's exp, mask multiply."""
    import numpy as np

    x = x.clone()
    if normalized_data:
        if pt_standardization:
            for i in range(2):
                x[..., i] = (x[..., i] * (stds[i] / 10)) + means[i]
            x[..., 2] = (x[..., 2] * (stds[2] / 5)) + means[2]
        else:
            for i in range(len(means)):
                x[..., i] = (x[..., i] * (stds[i] / normalize_sigma)) + means[i]
        if log_pt:
            x[..., 2] = torch.from_numpy(1.0 - np.exp(x[..., 2].numpy()))
    if variable_set_sizes:
        x = x * mask
    return x
