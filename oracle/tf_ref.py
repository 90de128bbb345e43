"""Eager-PyTorch CPU restatement of the Full-Transformer vector field (BASELINE cfg 4).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this.  PINNED: checked against vectors recorded from the
reference's own modules (tests/golden/tf_*.npz, written by oracle/make_golden.py).

Follows, function by function:
  * particle_fm/models/components/droid_transformer.py:16-52    merge_masks (kv padding only, queries unmasked)
  * droid_transformer.py:211-284   MultiHeadedAttentionBlock.forward (do_selfattn, torch SDPA, optional
                                   LayerNorm before out_linear)
  * droid_transformer.py:331-344   TransformerEncoderLayer.forward (x + MHA(norm1 x); x + dense(norm2 x, ctxt))
  * droid_transformer.py:433-437   TransformerEncoder.forward (layers, final_norm)
  * droid_transformer.py:529-548   FullTransformerEncoder.forward (ctxt = ctxt_emdb(cat(t[:,0], cond)); node_embd; te; outp_embd)
  * droid_transformer.py:793-813   MLPBlock.forward (cat(inpt, ctxt) -> Linear -> act -> norm)
  * droid_transformer.py:958-981   DenseNetwork.forward (input block, hidden blocks, output block)
  * droid_transformer.py:1014-1051 get_act / get_nrm ("lrlu" = LeakyReLU(0.1), "layer" = LayerNorm)
  * particle_fm/models/flow_matching_module.py:191-204  CNF.forward (cosine embedding, add_time_to_input)

State is the reference's ``state_dict`` (keys ``<prefix>net.…``); nothing is an nn.Module here.
"""
from __future__ import annotations

import math
from typing import Mapping, Optional

import torch
import torch.nn.functional as F

from .fm_ref import gaussian_time_embedding, time_embedding

LRLU_SLOPE = 0.1  # droid_transformer.py:1022


def _lin(state, key, x):
    return F.linear(x, state[key + ".weight"], state[key + ".bias"])


def _ln(state, key, x):
    w = state[key + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, state[key + ".bias"], 1e-5)


def mlp_block(state, key, x, ctxt, has_act: bool, has_nrm: bool):
    """MLPBlock with n_layers=1 (droid_transformer.py:793-813): Linear, [act], [norm]."""
    h = torch.cat([x, ctxt], dim=-1) if ctxt is not None else x
    h = _lin(state, key + ".block.0", h)
    i = 1
    if has_act:
        h = F.leaky_relu(h, LRLU_SLOPE)
        i += 1
    if has_nrm:
        h = _ln(state, f"{key}.block.{i}", h)
    return h


def dense_network(state, key, x, ctxt=None, nrm: bool = True):
    """DenseNetwork with one hidden block, act_h "lrlu", act_o "none" (droid_transformer.py:958-981).
    The context is broadcast over the set dimension and concatenated to the INPUT block only."""
    if ctxt is not None and x.dim() > ctxt.dim():
        ctxt = ctxt.unsqueeze(1).expand(*x.shape[:-1], -1)
    h = mlp_block(state, key + ".input_block", x, ctxt, True, nrm)
    return mlp_block(state, key + ".output_block", h, None, False, False)


def mha(state, key, x, kv_mask, num_heads: int, do_layer_norm: bool):
    """Self-attention block (droid_transformer.py:231-284).  kv_mask (B,N) bool; every query row is
    computed (merge_masks leaves queries unmasked, :16-52)."""
    B, N, D = x.shape
    hd = D // num_heads
    q, k, v = _lin(state, key + ".all_linear", x).chunk(3, -1)
    shape = (B, -1, num_heads, hd)
    q, k, v = (a.view(shape).transpose(1, 2) for a in (q, k, v))
    s = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
    if kv_mask is not None:
        s = s + torch.zeros(B, 1, 1, N).masked_fill(~kv_mask[:, None, None, :], -float("inf"))
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).contiguous().view(B, -1, D)
    if do_layer_norm:
        a = _ln(state, key + ".layer_norm", a)
    return _lin(state, key + ".out_linear", a)


def full_transformer(state: Mapping[str, torch.Tensor], prefix: str, temb, x, cond, mask, *, num_layers: int,
                     num_heads: int, do_layer_norm: bool = True, nrm: bool = True, intermediates: Optional[dict] = None):
    """FullTransformerEncoder.forward (droid_transformer.py:529-548).
    temb (B,N,T) expanded time embedding, x (B,N,D_in) (already time-concatenated by CNF.forward),
    cond (B,Cg) or None, mask (B,N,1) -> (B,N,outp)."""
    p = prefix
    kv = mask.squeeze(-1).bool()  # :539
    ctxt = temb[:, 0] if cond is None else torch.cat([temb[:, 0], cond], dim=-1)  # :541
    ctxt = dense_network(state, p + "ctxt_emdb", ctxt, None, nrm)  # :542
    h = dense_network(state, p + "node_embd", x, ctxt, nrm)  # :545
    if intermediates is not None:
        intermediates["ctxt"], intermediates["x0"] = ctxt, h
    for k in range(num_layers):  # :434-436
        lp = f"{p}te.layers.{k}."
        h = h + mha(state, lp + "self_attn", _ln(state, lp + "norm1", h), kv, num_heads, do_layer_norm)  # :340-342
        h = h + dense_network(state, lp + "dense", _ln(state, lp + "norm2", h), ctxt, nrm)  # :343
        if intermediates is not None:
            intermediates[f"x{k + 1}"] = h
    h = _ln(state, p + "te.final_norm", h)  # :437
    return dense_network(state, p + "outp_embd", h, ctxt, nrm)  # :547


class TransformerVectorField:
    """CNF.forward for model="droid_fulltransformer", t_emb="cosine" (flow_matching_module.py:191-204)."""

    def __init__(self, state, prefix: str, hp: Mapping, freqs=None):
        self.state, self.prefix, self.hp, self.freqs = state, prefix, dict(hp), freqs

    def __call__(self, t, x, cond=None, mask=None, intermediates=None):
        hp = self.hp
        if hp.get("t_emb", "cosine") == "gaussian":  # flow_matching_module.py:178-181, 213-221: a trainable embedding network of the CNF
            temb = gaussian_time_embedding(t, x, self.state, self.prefix, hp.get("activation", "leaky_relu"))
        else:
            temb = time_embedding(t, x, hp, self.freqs)
        if hp.get("add_time_to_input", True):
            x = torch.cat((temb, x), dim=-1)
        te = hp["net_config"]["te_config"]
        return full_transformer(
            self.state, self.prefix + "net.", temb, x, cond, mask,
            num_layers=te["num_layers"], num_heads=te["mha_config"]["num_heads"],
            do_layer_norm=te["mha_config"].get("do_layer_norm", False),
            nrm=te["dense_config"].get("nrm", "none") == "layer", intermediates=intermediates)
