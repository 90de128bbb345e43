"""Eager-PyTorch CPU restatement of the reference EPiC vector-field network.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Functional style: every
function takes an explicit ``state`` mapping with the reference's state_dict
key names (``fc_l1.weight_g`` ...), so the same golden weights drive the
reference, this oracle and the HIP path.

Deliberately keeps the reference's inefficiencies (weight-norm recomputed every
call, concatenated t/cond columns, per-particle time embedding) so that timing
it is an honest stand-in for "reference PyTorch CPU path" (BASELINE.md §3).

Follows:
  * particle_fm/models/components/epic.py:85-203   (EPiC_layer.forward)
  * particle_fm/models/components/epic.py:304-391  (EPiC_encoder.forward)
  * torch.nn.utils.weight_norm (old style): W = g * v / ||v||_2 per output row
"""
from __future__ import annotations

from typing import Mapping, Optional

import torch
import torch.nn.functional as F


def wn_linear(state: Mapping[str, torch.Tensor], prefix: str, inp: torch.Tensor) -> torch.Tensor:
    """Weight-normalised Linear, as ``nn.utils.weight_norm(nn.Linear)`` evaluates it
    (epic.py:66-81, 262-300): ``W[o, :] = g[o] * v[o, :] / ||v[o, :]||``."""
    v = state[prefix + ".weight_v"]
    g = state[prefix + ".weight_g"]
    b = state[prefix + ".bias"]
    w = v * (g / v.norm(dim=1, keepdim=True))
    return F.linear(inp, w, b)


def _act(x: torch.Tensor, activation: str = "leaky_relu") -> torch.Tensor:
    # epic.py:180 -- getattr(F, activation, lambda x: x): "leaky_relu" (default slope 0.01) in every shipped config; a name torch.nn.functional
    # does not know means NO activation
    return getattr(F, activation, lambda v: v)(x)


def epic_layer(
    state: Mapping[str, torch.Tensor],
    prefix: str,
    t_local: Optional[torch.Tensor],
    t_global: Optional[torch.Tensor],
    x_global: torch.Tensor,
    x_local: torch.Tensor,
    cond_g: Optional[torch.Tensor],
    cond_l: Optional[torch.Tensor],
    mask: torch.Tensor,
    sum_scale: float,
    activation: str = "leaky_relu",
):
    """One EPiC layer (epic.py:159-203).  Returns (x_global, x_local)."""
    n = x_local.shape[1]
    # masked mean/sum pooling, order (mean, sum*scale, global)   epic.py:160-171
    pooled_sum = (x_local * mask).sum(1)
    pooled_mean = pooled_sum / mask.sum(1)
    pooled_sum = pooled_sum * sum_scale
    parts = [p for p in (t_global, pooled_mean, pooled_sum, x_global, cond_g) if p is not None]
    g1 = _act(wn_linear(state, prefix + ".fc_global1", torch.cat(parts, -1)), activation)  # :180-182
    parts = [p for p in (t_global, g1, cond_g) if p is not None]
    x_global = _act(wn_linear(state, prefix + ".fc_global2", torch.cat(parts, -1)) + x_global, activation)  # :184-186
    g2l = x_global.unsqueeze(1).expand(-1, n, -1)  # :189 (repeat_interleave over points)
    parts = [p for p in (t_local, x_local, g2l, cond_l) if p is not None]
    l1 = _act(wn_linear(state, prefix + ".fc_local1", torch.cat(parts, -1)), activation)  # :194-196
    parts = [p for p in (t_local, l1, cond_l) if p is not None]
    x_local = _act(wn_linear(state, prefix + ".fc_local2", torch.cat(parts, -1)) + x_local, activation)  # :198-200
    return x_global, x_local


def epic_encoder(
    state: Mapping[str, torch.Tensor],
    prefix: str,
    t_emb: torch.Tensor,
    x: torch.Tensor,
    cond: Optional[torch.Tensor],
    mask: Optional[torch.Tensor],
    *,
    layers: int,
    t_local_cat: bool = True,
    t_global_cat: bool = True,
    global_cond_dim: int = 0,
    local_cond_dim: int = 0,
    sum_scale: float = 1e-2,
    activation: str = "leaky_relu",
) -> torch.Tensor:
    """EPiC_encoder.forward (epic.py:304-391).

    t_emb (B,N,T), x (B,N,D_in), cond (B,Cg) or None, mask (B,N,1) or None -> (B,N,F).
    ``prefix`` is the state_dict prefix of the net (e.g. ``"flows.0.net"``) or "".
    """
    p = (prefix + ".") if prefix else ""
    if x is None:
        raise ValueError("x_local is None")
    if cond is None and (global_cond_dim > 0 or local_cond_dim > 0):
        raise ValueError("conditioning dims set but no global_cond given")  # epic.py:313-317
    if t_emb is None and (t_local_cat or t_global_cat):
        raise ValueError("t_local_cat/t_global_cat set but no t given")  # epic.py:318-322
    n = x.shape[1]
    if mask is None:  # epic.py:328-329
        mask = torch.ones_like(x[:, :, 0]).unsqueeze(-1)
    mask = mask.to(x.dtype) if not mask.is_floating_point() else mask
    t_local = t_emb if t_local_cat else None
    t_global = t_emb[:, 0, :] if t_global_cat else None  # epic.py:342
    cond_g = cond if global_cond_dim > 0 else None
    cond_l = cond.unsqueeze(1).expand(-1, n, -1) if local_cond_dim > 0 else None  # :353-354

    parts = [q for q in (t_local, x, cond_l) if q is not None]
    h = _act(wn_linear(state, p + "fc_l1", torch.cat(parts, -1)), activation)  # :360-362
    parts = [q for q in (t_local, h, cond_l) if q is not None]
    h = _act(wn_linear(state, p + "fc_l2", torch.cat(parts, -1)) + h, activation)  # :364-366

    z_sum = (h * mask).sum(1)  # :369
    z_mean = z_sum / mask.sum(1)  # :370  (NaN for an all-masked jet, as the reference)
    z_sum = z_sum * sum_scale  # :371
    parts = [q for q in (t_global, z_sum, z_mean, cond_g) if q is not None]  # (sum, mean) order :373
    g = _act(wn_linear(state, p + "fc_g1", torch.cat(parts, -1)), activation)  # :375-377
    parts = [q for q in (t_global, g, cond_g) if q is not None]
    g = _act(wn_linear(state, p + "fc_g2", torch.cat(parts, -1)), activation)  # :378-380

    for k in range(layers):  # :382-385
        g, h = epic_layer(
            state, f"{p}nn_list.{k}", t_local, t_global, g, h, cond_g, cond_l, mask, sum_scale, activation
        )

    parts = [q for q in (t_local, h, cond_l) if q is not None]
    out = _act(wn_linear(state, p + "fc_l3", torch.cat(parts, -1)), activation)  # :387-389
    return out * mask  # :391


def epic_param_shapes(
    *,
    features: int,
    input_dim: int,
    hidden: int,
    latent: int,
    layers: int,
    t_dim_local: int,
    t_dim_global: int,
    global_cond_dim: int = 0,
    local_cond_dim: int = 0,
):
    """(key, shape) list in the reference's registration order (epic.py:264-300, 67-81):
    for each weight-normed Linear ``bias, weight_g, weight_v``."""
    H, L = hidden, latent

    def lin(name, i, o):
        return [(f"{name}.bias", (o,)), (f"{name}.weight_g", (o, 1)), (f"{name}.weight_v", (o, i))]

    out = []
    out += lin("fc_l1", input_dim + t_dim_local + local_cond_dim, H)
    out += lin("fc_l2", H + t_dim_local + local_cond_dim, H)
    out += lin("fc_g1", 2 * H + t_dim_global + global_cond_dim, H)
    out += lin("fc_g2", H + t_dim_global + global_cond_dim, L)
    for k in range(layers):
        out += lin(f"nn_list.{k}.fc_global1", 2 * H + L + t_dim_global + global_cond_dim, H)
        out += lin(f"nn_list.{k}.fc_global2", H + t_dim_global + global_cond_dim, L)
        out += lin(f"nn_list.{k}.fc_local1", H + L + t_dim_local + local_cond_dim, H)
        out += lin(f"nn_list.{k}.fc_local2", H + t_dim_local + local_cond_dim, H)
    out += lin("fc_l3", H + t_dim_local + local_cond_dim, features)
    return out
