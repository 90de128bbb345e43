"""Eager-PyTorch CPU restatement of the cross-attention vector field (model="droid_fullcrossattention",
configs/model/fm_droid_crossattention.yaml).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PINNED: checked against vectors recorded from the reference's own
modules (tests/golden/ca_*.npz, written by oracle/make_golden.py).

Follows:
  * droid_transformer.py:347-397   TransformerCrossAttentionLayer.forward
        q = q + cross_attn(norm1(q), norm0(kv), kv_mask);  q = q + dense(norm2(q), ctxt)
  * droid_transformer.py:231-284   MultiHeadedAttentionBlock.forward with do_selfattn=False (q_linear / k_linear / v_linear)
  * droid_transformer.py:442-472   CrossAttentionEncoder.forward (global tokens <- sequence (masked), sequence <- tokens)
  * droid_transformer.py:685-711   FullCrossAttentionEncoder.forward (ctxt_emdb, node_embd, cae, outp_embd)
  * DenseNetwork / MLPBlock / get_act / get_nrm as in oracle/tf_ref.py
"""
from __future__ import annotations

import math
from typing import Mapping

import torch

from .fm_ref import gaussian_time_embedding, time_embedding
from .tf_ref import _lin, _ln, dense_network


def cross_mha(state, key, q_in, kv_in, kv_mask, num_heads: int, do_layer_norm: bool):
    """q_in (B,Lq,D), kv_in (B,S,D), kv_mask (B,S) bool or None."""
    B, Lq, D = q_in.shape
    hd = D // num_heads
    q = _lin(state, key + ".q_linear", q_in)
    k = _lin(state, key + ".k_linear", kv_in)
    v = _lin(state, key + ".v_linear", kv_in)
    shape = (B, -1, num_heads, hd)
    q, k, v = (a.view(shape).transpose(1, 2) for a in (q, k, v))
    s = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
    if kv_mask is not None:
        s = s + torch.zeros(B, 1, 1, kv_in.shape[1]).masked_fill(~kv_mask[:, None, None, :], -float("inf"))
    a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).contiguous().view(B, -1, D)
    if do_layer_norm:
        a = _ln(state, key + ".layer_norm", a)
    return _lin(state, key + ".out_linear", a)


def cross_layer(state, key, q_seq, kv_seq, kv_mask, ctxt, num_heads, do_layer_norm):
    q_seq = q_seq + cross_mha(state, key + ".cross_attn", _ln(state, key + ".norm1", q_seq), _ln(state, key + ".norm0", kv_seq),
                              kv_mask, num_heads, do_layer_norm)
    return q_seq + dense_network(state, key + ".dense", _ln(state, key + ".norm2", q_seq), ctxt, True)


def full_cross_attention(state: Mapping[str, torch.Tensor], prefix: str, temb, x, cond, mask, *, num_layers: int, num_heads: int,
                         do_layer_norm: bool = True):
    p = prefix
    kv = mask.squeeze(-1).bool()
    ctxt = temb[:, 0] if cond is None else torch.cat([temb[:, 0], cond], dim=-1)
    ctxt = dense_network(state, p + "ctxt_emdb", ctxt, None, True)
    seq = dense_network(state, p + "node_embd", x, ctxt, True)
    tok = state[p + "cae.global_tokens"].expand(seq.shape[0], -1, -1)
    for l in range(num_layers):
        tok = cross_layer(state, f"{p}cae.from_layers.{l}", tok, seq, kv, ctxt, num_heads, do_layer_norm)
        seq = cross_layer(state, f"{p}cae.to_layers.{l}", seq, tok, None, ctxt, num_heads, do_layer_norm)
    return dense_network(state, p + "outp_embd", seq, ctxt, True)


class CrossAttentionVectorField:
    """CNF.forward for model="droid_fullcrossattention" (flow_matching_module.py:159-165, 191-204)."""

    def __init__(self, state, prefix: str, hp: Mapping, freqs=None):
        self.state, self.prefix, self.hp, self.freqs = state, prefix, dict(hp), freqs

    def __call__(self, t, x, cond=None, mask=None):
        hp = self.hp
        if hp.get("t_emb", "cosine") == "gaussian":  # flow_matching_module.py:178-181, 213-221: a trainable embedding network of the CNF
            temb = gaussian_time_embedding(t, x, self.state, self.prefix, hp.get("activation", "leaky_relu"))
        else:
            temb = time_embedding(t, x, hp, self.freqs)
        if hp.get("add_time_to_input", True):
            x = torch.cat((temb, x), dim=-1)
        cae = hp["net_config"]["cae_config"]
        return full_cross_attention(self.state, self.prefix + "net.", temb, x, cond, mask, num_layers=cae["num_layers"],
                                    num_heads=cae["mha_config"]["num_heads"],
                                    do_layer_norm=cae["mha_config"].get("do_layer_norm", False))
