"""Eager-PyTorch CPU restatement of the MDMA vector field (model="mdma", configs/model/flow_matching_mdma.yaml).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PINNED: checked against vectors recorded from the reference's own
modules (tests/golden/mdma_*.npz, written by oracle/make_golden.py).

Follows (local_cat_cond = global_cat_cond = False, global_cond_dim = 0 as shipped; t_local_cat / t_global_cat either way -- the
yaml has them False, MDMA.__init__'s own defaults are True: mdma.py:101-102, 56-59, 71-78, 155-156; the conditional variant --
net_config.global_cond_dim = 1, local_cat_cond, global_cat_cond, all off in the yaml: mdma.py:60-63, 79-82, 157-174 -- likewise):
  * mdma.py:142-176  MDMA.forward: embed + LeakyReLU, padded particles zeroed, class token from (sum / avg_n, count) through
                     embbed_cls, gated (F.glu) by cond(count); the blocks; out(LeakyReLU(x)) * mask -- ONE output per particle
  * mdma.py:53-84    Block.forward: x = fc0(act(x)); x_cls = ln(fc0_cls(act(x_cls))); x_cls = attn(x_cls, x, x, padded keys masked);
                     x_cls = fc1_cls(cat(x_cls, cond)); x_cls = fc2_cls(x_cls); x = fc1(cat(x, x_cls.expand)) + res
  * torch.nn.MultiheadAttention (batch_first, one query): q / k / v = rows 0..H / H..2H / 2H..3H of in_proj, heads of
                     H / num_heads columns, softmax(q k^T / sqrt(head_dim)) v, out_proj
  * flow_matching_module.py:191-204 CNF.forward: x = cat(time embedding, x) when add_time_to_input
The shape-(B, N, 1) output is what the reference's losses (losses.py:64-75: ``(v_t - u_t).square()`` broadcasts) and solvers
(``x + dt * f``) consume; ``broadcast_field`` expands it to (B, N, F) as they implicitly do.
"""
from __future__ import annotations

import math
from typing import Mapping

import torch
import torch.nn.functional as F

from .fm_ref import gaussian_time_embedding, time_embedding

NEG_SLOPE = 0.01  # nn.LeakyReLU() default (mdma.py:46, 138)


def _lin(state, key, x):
    return F.linear(x, state[key + ".weight"], state[key + ".bias"])


def one_query_attention(state, key: str, q_in, kv_in, key_pad, num_heads: int):
    """nn.MultiheadAttention(H, num_heads, batch_first=True)(q_in, kv_in, kv_in, key_padding_mask=key_pad); q_in (B,1,H)."""
    B, S, H = kv_in.shape
    hd = H // num_heads
    W, b = state[key + ".in_proj_weight"], state[key + ".in_proj_bias"]
    q = F.linear(q_in, W[:H], b[:H]).view(B, 1, num_heads, hd).transpose(1, 2)
    k = F.linear(kv_in, W[H:2 * H], b[H:2 * H]).view(B, S, num_heads, hd).transpose(1, 2)
    v = F.linear(kv_in, W[2 * H:], b[2 * H:]).view(B, S, num_heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-2, -1)) / math.sqrt(hd)
    s = s.masked_fill(key_pad[:, None, None, :], -float("inf"))
    a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, 1, H)
    return _lin(state, key + ".out_proj", a)


def mdma_block(state, key: str, x, x_cls, cond, mask_bool, num_heads: int, t_in=None, t_local: bool = False, t_global: bool = False,
               local_cat_cond: bool = False, global_cat_cond: bool = False):
    """Block.forward (mdma.py:53-84); mask_bool (B,N,1); t_in (B,N,T) the time embedding (t_local_cat / t_global_cat); cond (B,1,1 or 2):
    the particle count and, behind it, the jet's condition -- the *_cat_cond switches append its LAST entry."""
    res = x
    if t_local:  # :56-57 (the activation below covers the concatenated embedding too)
        x = torch.cat((x, t_in), dim=-1)
    if t_global:  # :58-59
        x_cls = torch.cat((x_cls, t_in[:, :1, :]), dim=-1)
    if global_cat_cond:  # :60-61
        x_cls = torch.cat((x_cls, cond[..., -1:]), dim=-1)
    if local_cat_cond:  # :62-63
        x = torch.cat((x, cond[..., -1:].expand(-1, x.shape[1], 1)), dim=-1)
    x = _lin(state, key + ".fc0", F.leaky_relu(x, NEG_SLOPE))
    x_cls = F.layer_norm(_lin(state, key + ".fc0_cls", F.leaky_relu(x_cls, NEG_SLOPE)), (x.shape[-1],),
                         state[key + ".ln.weight"], state[key + ".ln.bias"], 1e-5)
    x_cls = one_query_attention(state, key + ".attn", x_cls, x, ~mask_bool.squeeze(-1), num_heads)
    x_cls = torch.cat((x_cls, cond) + ((t_in[:, :1, :],) if t_global else ()), dim=-1)  # :70-74
    x_cls = _lin(state, key + ".fc1_cls", x_cls)
    if t_global:  # :78
        x_cls = torch.cat((x_cls, t_in[:, :1, :]), dim=-1)
    if global_cat_cond:  # :79
        x_cls = torch.cat((x_cls, cond[..., -1:]), dim=-1)
    x_cls = _lin(state, key + ".fc2_cls", x_cls)
    if local_cat_cond:  # :81-82
        x = torch.cat((x, cond[..., -1:].expand(-1, x.shape[1], 1)), dim=-1)
    x = _lin(state, key + ".fc1", torch.cat((x, x_cls.expand(-1, x.shape[1], -1)), dim=-1)) + res
    return x, x_cls


def mdma_forward(state: Mapping[str, torch.Tensor], prefix: str, x, mask, *, num_layers: int, num_heads: int, avg_n: float, t_in=None,
                 t_local: bool = False, t_global: bool = False, global_cond_in=None, global_cond: bool = False,
                 local_cat_cond: bool = False, global_cat_cond: bool = False):
    """MDMA.forward (mdma.py:142-176) on the already time-concatenated input; returns (B, N, 1).  global_cond_in (B, 1): the jet's
    condition (global_cond = net_config.global_cond_dim > 0, and the two *_cat_cond switches)."""
    p = prefix
    mb = mask.bool()
    if t_local:  # :155-156
        x = torch.cat((x, t_in), dim=-1)
    if local_cat_cond:  # :157-158
        x = torch.cat((x, global_cond_in.unsqueeze(-1).expand(-1, x.shape[1], 1)), dim=-1)
    x = F.leaky_relu(_lin(state, p + "embed", x), NEG_SLOPE)
    x = x * mb.to(x.dtype)  # x[~mask] = 0
    n_valid = mask.sum(1, keepdim=True).reshape(-1, 1, 1).to(x.dtype)
    x_cls = torch.cat((x.sum(1, keepdim=True) / avg_n, n_valid), dim=-1)
    if global_cat_cond or global_cond:  # :164-165
        x_cls = torch.cat((x_cls, global_cond_in.unsqueeze(-1)), dim=-1)
    x_cls = _lin(state, p + "embbed_cls", x_cls)
    cond = n_valid
    if global_cond or global_cat_cond:  # :168-169
        cond = torch.cat((cond, global_cond_in.unsqueeze(-1)), dim=-1)
    x_cls = F.glu(torch.cat((x_cls, _lin(state, p + "cond", cond)), dim=-1))
    for l in range(num_layers):
        x, x_cls = mdma_block(state, f"{p}encoder.{l}", x, x_cls, cond, mb, num_heads, t_in, t_local, t_global, local_cat_cond,
                              global_cat_cond)
    if local_cat_cond:  # :173-174
        x = torch.cat((x, global_cond_in.unsqueeze(-1).expand(-1, x.shape[1], 1)), dim=-1)
    return _lin(state, p + "out", F.leaky_relu(x, NEG_SLOPE)) * mask


class MdmaVectorField:
    """CNF.forward for model="mdma" (flow_matching_module.py:163-167, 191-204): returns (B, N, 1)."""

    def __init__(self, state, prefix: str, hp: Mapping, freqs=None):
        self.state, self.prefix, self.hp, self.freqs = state, prefix, dict(hp), freqs

    def __call__(self, t, x, cond=None, mask=None):
        hp = self.hp
        nc = hp.get("net_config") or {}
        if hp.get("t_emb", "cosine") == "gaussian":  # flow_matching_module.py:178-181, 213-221: a trainable embedding network of the CNF
            temb = gaussian_time_embedding(t, x, self.state, self.prefix, hp.get("activation", "leaky_relu"))
        else:
            temb = time_embedding(t, x, hp, self.freqs)
        if hp.get("add_time_to_input", True):
            x = torch.cat((temb, x), dim=-1)
        return mdma_forward(self.state, self.prefix + "net.", x, mask, num_layers=int(nc.get("layers", 16)),
                            num_heads=int(nc.get("num_heads", 8)), avg_n=float(nc.get("avg_n", 30)), t_in=temb,
                            t_local=bool(nc.get("t_local_cat", True)), t_global=bool(nc.get("t_global_cat", True)),
                            global_cond_in=cond, global_cond=int(nc.get("global_cond_dim", 0)) > 0,
                            local_cat_cond=bool(nc.get("local_cat_cond", False)), global_cat_cond=bool(nc.get("global_cat_cond", False)))


def broadcast_field(vf):
    """The (B, N, 1) field as the (B, N, F) array the reference's loss and solver arithmetic broadcast it to."""
    def f(t, x, cond=None, mask=None):
        return vf(t, x, cond, mask).expand(-1, -1, x.shape[-1])
    return f
