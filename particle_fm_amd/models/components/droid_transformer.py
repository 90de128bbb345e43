"""Drop-in for the classes of particle_fm/models/components/droid_transformer.py that the models
"droid_fulltransformer" and "droid_fullcrossattention" instantiate: ``FullTransformerEncoder``,
``FullCrossAttentionEncoder`` and their parameter containers.

Same class names, constructor keywords, parameter names / shapes / registration order (hence the same
``state_dict`` keys and, under a fixed seed, bit-identical default initialisation incl. ``init_zeros`` /
``output_init_zeros``) as the reference (droid_transformer.py:117-229, 287-397, 400-527, 551-700, 714-1051).  The sub-blocks
own parameters only: one evaluation of the whole encoder is a fixed sequence of HIP launches
(``FullTransformerEncoder.forward`` / ``vector_field``), there is no per-block PyTorch compute and no CPU fallback.
Configurations the kernels do not cover raise NotImplementedError at construction.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Mapping, Optional, Union

import torch
import torch.nn as nn

from ... import hip_ops_ca, hip_ops_tf
from ...layout_ca import CaConfig, CaLayout
from ...freq_table import FreqTableMixin
from ...layout_tf import TfConfig, TfLayout, default_freqs


def _container_only(name):
    def forward(self, *a, **k):
        raise RuntimeError(f"{name} is evaluated inside the fused FullTransformerEncoder HIP path, not on its own")
    return forward


def get_act(name: str) -> nn.Module:
    if name == "lrlu":
        return nn.LeakyReLU(0.1)  # droid_transformer.py:1022
    raise NotImplementedError(f"activation {name!r} has no HIP path in this build (only 'lrlu')")


def get_nrm(name: str, outp_dim: int) -> nn.Module:
    if name == "layer":
        return nn.LayerNorm(outp_dim)
    raise NotImplementedError(f"normalisation {name!r} has no HIP path in this build (only 'layer')")


class MLPBlock(nn.Module):
    """droid_transformer.py:714-827: ``block`` = [Linear, act, norm] (n_layers = 1)."""

    def __init__(self, inpt_dim: int, outp_dim: int, ctxt_dim: int = 0, n_layers: int = 1, act: str = "lrlu",
                 nrm: str = "none", drp: float = 0, do_res: bool = False, init_zeros: bool = False) -> None:
        super().__init__()
        if n_layers != 1 or drp > 0 or do_res:
            raise NotImplementedError("MLPBlock: the HIP path implements n_layers=1, no dropout, no residual")
        self.inpt_dim, self.outp_dim, self.ctxt_dim = inpt_dim, outp_dim, ctxt_dim
        self.block = nn.ModuleList([nn.Linear(inpt_dim + ctxt_dim, outp_dim)])
        if init_zeros:
            self.block[-1].weight.data.fill_(0)
            self.block[-1].bias.data.fill_(0)
        if act != "none":
            self.block.append(get_act(act))
        if nrm != "none":
            self.block.append(get_nrm(nrm, outp_dim))

    forward = _container_only("MLPBlock")


class DenseNetwork(nn.Module):
    """droid_transformer.py:830-1011 with one hidden block: input_block (Linear, act_h, nrm) -> output_block (Linear)."""

    def __init__(self, inpt_dim: int, outp_dim: int = 0, ctxt_dim: int = 0, hddn_dim: Union[int, list] = 32,
                 num_blocks: int = 1, n_lyr_pbk: int = 1, act_h: str = "lrlu", act_o: str = "none", do_out: bool = True,
                 nrm: str = "none", drp: float = 0, drp_on_output: bool = False, nrm_on_output: bool = False,
                 do_res: bool = False, ctxt_in_inpt: bool = True, ctxt_in_hddn: bool = False,
                 output_init_zeros: bool = False) -> None:
        super().__init__()
        if not isinstance(hddn_dim, int) or num_blocks != 1 or act_o != "none" or not do_out or drp > 0 \
                or nrm_on_output or not ctxt_in_inpt or ctxt_in_hddn or nrm != "layer" or act_h != "lrlu":
            raise NotImplementedError(
                "DenseNetwork: the HIP path implements one hidden block (int hddn_dim), act_h='lrlu', nrm='layer', "
                "act_o='none', context in the input block only, no dropout (configs/model/fm_droid_transformer.yaml)")
        self.inpt_dim, self.hddn_dim, self.num_blocks = inpt_dim, [hddn_dim], 1
        self.outp_dim = outp_dim or inpt_dim
        self.ctxt_dim, self.do_out = ctxt_dim, do_out
        self.hidden_features = hddn_dim
        self.input_block = MLPBlock(inpt_dim=inpt_dim, outp_dim=hddn_dim, ctxt_dim=ctxt_dim, act=act_h, nrm=nrm)
        self.hidden_blocks = []
        self.output_block = MLPBlock(inpt_dim=hddn_dim, outp_dim=self.outp_dim, act=act_o, init_zeros=output_init_zeros)

    forward = _container_only("DenseNetwork")


class MultiHeadedAttentionBlock(nn.Module):
    """droid_transformer.py:117-284: all_linear (self-attention) or q_linear / k_linear / v_linear (cross-attention),
    layer_norm, out_linear."""

    def __init__(self, model_dim: int, num_heads: int = 1, drp: float = 0, init_zeros: bool = False,
                 do_selfattn: bool = False, do_layer_norm: bool = False, attn_act=None) -> None:
        super().__init__()
        if not do_layer_norm or drp > 0 or attn_act is not None:
            raise NotImplementedError("MultiHeadedAttentionBlock: the HIP path implements do_layer_norm=True, softmax, no dropout")
        self.model_dim, self.num_heads, self.head_dim = model_dim, num_heads, model_dim // num_heads
        if self.head_dim * num_heads != model_dim:
            raise ValueError("Model dimension must be divisible by number of heads!")  # droid_transformer.py:191
        self.do_selfattn, self.drp, self.do_layer_norm, self.attn_act = do_selfattn, drp, do_layer_norm, attn_act
        if do_selfattn:
            self.all_linear = nn.Linear(model_dim, 3 * model_dim)
        else:
            self.q_linear = nn.Linear(model_dim, model_dim)
            self.k_linear = nn.Linear(model_dim, model_dim)
            self.v_linear = nn.Linear(model_dim, model_dim)
        self.layer_norm = nn.LayerNorm(model_dim)
        self.out_linear = nn.Linear(model_dim, model_dim)
        if init_zeros:
            self.out_linear.weight.data.fill_(0)
            self.out_linear.bias.data.fill_(0)

    forward = _container_only("MultiHeadedAttentionBlock")


class TransformerEncoderLayer(nn.Module):
    """droid_transformer.py:287-344."""

    def __init__(self, model_dim: int, mha_config: Mapping | None = None, dense_config: Mapping | None = None,
                 ctxt_dim: int = 0) -> None:
        super().__init__()
        self.model_dim, self.ctxt_dim = model_dim, ctxt_dim
        self.self_attn = MultiHeadedAttentionBlock(model_dim, do_selfattn=True, **(mha_config or {}))
        self.dense = DenseNetwork(model_dim, outp_dim=model_dim, ctxt_dim=ctxt_dim, **(dense_config or {}))
        self.norm1 = nn.LayerNorm(model_dim)
        self.norm2 = nn.LayerNorm(model_dim)

    forward = _container_only("TransformerEncoderLayer")


class TransformerEncoder(nn.Module):
    """droid_transformer.py:400-437."""

    def __init__(self, model_dim: int = 64, num_layers: int = 3, mha_config: Mapping | None = None,
                 dense_config: Mapping | None = None, ctxt_dim: int = 0) -> None:
        super().__init__()
        self.model_dim, self.num_layers = model_dim, num_layers
        self.layers = nn.ModuleList([TransformerEncoderLayer(model_dim, mha_config, dense_config, ctxt_dim)
                                     for _ in range(num_layers)])
        self.final_norm = nn.LayerNorm(model_dim)

    forward = _container_only("TransformerEncoder")


class TransformerCrossAttentionLayer(nn.Module):
    """droid_transformer.py:347-397: q <- q + cross_attn(norm1 q, norm0 kv); q <- q + dense(norm2 q, ctxt)."""

    def __init__(self, model_dim: int, mha_config: Mapping | None = None, dense_config: Mapping | None = None,
                 ctxt_dim: int = 0) -> None:
        super().__init__()
        self.model_dim, self.ctxt_dim = model_dim, ctxt_dim
        self.cross_attn = MultiHeadedAttentionBlock(model_dim, do_selfattn=False, **(mha_config or {}))
        self.dense = DenseNetwork(model_dim, outp_dim=model_dim, ctxt_dim=ctxt_dim, **(dense_config or {}))
        self.norm0 = nn.LayerNorm(model_dim)
        self.norm1 = nn.LayerNorm(model_dim)
        self.norm2 = nn.LayerNorm(model_dim)

    forward = _container_only("TransformerCrossAttentionLayer")


class CrossAttentionEncoder(nn.Module):
    """droid_transformer.py:551-618: learned global tokens, ``num_layers`` there-and-back cross-attention layer pairs."""

    def __init__(self, model_dim: int = 64, num_tokens: int = 4, num_layers: int = 5, mha_config: Mapping | None = None,
                 dense_config: Mapping | None = None, ctxt_dim: int = 0) -> None:
        super().__init__()
        self.model_dim, self.num_layers, self.num_tokens = model_dim, num_layers, num_tokens
        self.global_tokens = nn.Parameter(torch.randn((1, num_tokens, model_dim)))
        self.from_layers = nn.ModuleList([TransformerCrossAttentionLayer(model_dim, mha_config, dense_config, ctxt_dim)
                                          for _ in range(num_layers)])
        self.to_layers = nn.ModuleList([TransformerCrossAttentionLayer(model_dim, mha_config, dense_config, ctxt_dim)
                                        for _ in range(num_layers)])

    forward = _container_only("CrossAttentionEncoder")


class _FusedEncoder(FreqTableMixin, nn.Module):
    """What the two full encoders share: the kernel layout per (num_points, precision), the flat parameter vector and
    the packed weight blob.  Subclasses provide ``config()``, ``_LAYOUT`` and ``_forward_op``."""

    _LAYOUT = None
    _forward_op = None
    _GRAPH_FLAG = 0  # descriptor flag of the hipGraph replay of the sampler's step body (0: the path has none)
    _ONE_STREAM_FLAG = 0  # descriptor flag that keeps the midpoint sampler on the caller's stream (0: the path never splits a call)

    def _init_fused(self, num_points, frequencies, add_time_to_input, t_emb):
        # what the kernels need to know beyond the reference's own arguments
        self.num_points, self.frequencies, self.add_time_to_input, self.t_emb = num_points, frequencies, add_time_to_input, t_emb
        self._layouts = {}
        self.mfma_dtype = "fp32"  # "f16x3": every Linear on split-fp16 operands, fp32-grade accuracy; "bf16": bf16 operands (PFM_TF_F_BF16)
        # inference evaluates the valid particles only (PFM_*_F_VALID_ROWS): same numbers at valid particles, the reference's
        # unmasked values at padded positions (which every consumer multiplies by the mask) are not produced
        self.valid_rows_only = False
        # the midpoint sampler replays its step body as a hipGraph (PFM_CA_F_GRAPH_STEPS): for callers that keep several sampler
        # calls in flight from one thread and are bound by the host's launch rate; same kernels, same results
        self.graph_replay = False
        # a midpoint call on >= 64 jets runs as two half-batches on two internal streams (Full-Transformer: PFM_TF_F_ONE_STREAM switches
        # it off); callers that keep several calls in flight themselves (generate_data's pipeline) turn it off: same results either way
        self.stream_split = True
        self.cfg = self.config(num_points or 1)
        self._LAYOUT(self.cfg)  # rejects unsupported sizes at construction

    def layout(self, num_points: Optional[int] = None):
        n = num_points or self.num_points
        flags = {"fp32": 0, "f16x3": 1, "bf16": 32}[self.mfma_dtype] | (4 if self.valid_rows_only else 0) | (self._GRAPH_FLAG if self.graph_replay else 0) | (0 if self.stream_split else self._ONE_STREAM_FLAG)
        lay = self._layouts.get((n, flags))
        if lay is None:
            lay = self._layouts[(n, flags)] = self._LAYOUT(self.config(n), flags=flags)
        return lay

    def set_valid_rows_only(self, on: bool = True) -> None:
        """Sampling / forward skip padded particles (training is unaffected: the reference's loss includes padded rows)."""
        self.valid_rows_only = bool(on)

    def set_stream_split(self, on: bool = True) -> None:
        """A single midpoint call may fan out over two internal streams (paths that have it: Full-Transformer)."""
        self.stream_split = bool(on)

    def set_graph_replay(self, on: bool = True) -> None:
        """The midpoint sampler captures its step body once per call and replays it (paths that have it: cross-attention)."""
        self.graph_replay = bool(on)

    def set_precision(self, precision) -> None:
        """Accepts Lightning's spellings: "bf16", "bf16-mixed", "bf16-true" -> the Linears (forward and dX, inference and training) on bf16
        operands with fp32 accumulate (PFM_TF_F_BF16 / PFM_CA_F_BF16; LayerNorm, softmax, attention products and the dW GEMMs stay fp32);
        "f16x3" -> split-fp16 Linears (fp32-grade); anything else fp32."""
        p = str(precision)
        self.mfma_dtype = "bf16" if p.startswith("bf16") else ("f16x3" if p == "f16x3" else "fp32")

    def flat_parameters(self, layout=None) -> torch.Tensor:
        """All parameters in the layout's (= state_dict) order as one differentiable vector."""
        lay = layout or self.layout()
        named = dict(self.named_parameters())
        return torch.cat([named[k[len("net."):]].reshape(-1) for k in lay.keys()])

    def packed_weights(self, num_points: Optional[int] = None) -> torch.Tensor:
        """Kernel blob of the current parameter values (no autograd); rebuilt on every call so it can never go stale
        after an optimizer step, load_state_dict or an EMA swap, and never appears in state_dict()."""
        lay = self.layout(num_points)
        with torch.no_grad():
            flat = self.flat_parameters(lay)
            f = self.freq_tensor()
            if f is None:
                f = default_freqs(lay.cfg.t_dim, lay.cfg.t_emb)
            src = torch.cat([flat.float(), f.to(flat.device), torch.zeros(1, device=flat.device)])
            return src[lay.index_map_on(flat.device)]

    def vector_field(self, t: torch.Tensor, x: torch.Tensor, cond: torch.Tensor = None, mask: torch.Tensor = None,
                     blob: torch.Tensor = None) -> torch.Tensor:
        lay = self.layout(x.shape[1])
        if blob is None:
            blob = self.packed_weights(x.shape[1])
        return type(self)._forward_op(lay, blob, t, x, cond, mask)

    def forward(self, t: torch.Tensor, x: torch.Tensor, ctxt: torch.Tensor | None = None,
                mask: Optional[torch.Tensor] = None, attn_bias=None, attn_mask=None) -> torch.Tensor:
        raise RuntimeError(
            f"{type(self).__name__}.forward(t_emb, x_cat, ...) of the reference takes the already embedded time; the HIP "
            "path embeds in-kernel: call vector_field(t, x, cond, mask) (CNF.forward does)")


class FullTransformerEncoder(_FusedEncoder):
    """droid_transformer.py:440-548.  ``forward(t, x, ctxt, mask)`` keeps the reference's call (t = the (B,N,T)
    time embedding, x already time-concatenated); ``vector_field(t, x, cond, mask)`` takes the time itself (B,)
    and the bare particle features and lets the kernels embed (what CNF.forward uses)."""

    _ONE_STREAM_FLAG = 16  # PFM_TF_F_ONE_STREAM

    def __init__(self, inpt_dim: int, outp_dim: int, edge_dim: int = 0, ctxt_dim: int = 0,
                 te_config: Mapping | None = None, node_embd_config: Mapping | None = None,
                 outp_embd_config: Mapping | None = None, edge_embd_config: Mapping | None = None,
                 ctxt_embd_config: Mapping | None = None, *, num_points: int = 0, frequencies: int = 0,
                 add_time_to_input: bool = True, t_emb: str = "cosine") -> None:
        super().__init__()
        if edge_dim:
            raise NotImplementedError("edge features (attn_bias) have no HIP path in this build")
        if not ctxt_dim:
            raise NotImplementedError("the HIP transformer path needs the context network (ctxt_dim > 0: it always is, "
                                      "CNF passes global_cond_dim + 2*frequencies)")
        self.inpt_dim, self.outp_dim, self.ctxt_dim, self.edge_dim = inpt_dim, outp_dim, ctxt_dim, edge_dim
        te_config = deepcopy(te_config) or {}
        node_embd_config = deepcopy(node_embd_config) or {}
        outp_embd_config = deepcopy(outp_embd_config) or {}
        ctxt_embd_config = deepcopy(ctxt_embd_config) or {}
        te_config.setdefault("dense_config", {})
        if "model_dim" in te_config:  # droid_transformer.py:478-488: dense nets default to twice the width
            model_dim = te_config["model_dim"]
            for cfg in (node_embd_config, ctxt_embd_config, outp_embd_config, te_config["dense_config"]):
                cfg.setdefault("hddn_dim", 2 * model_dim)
        self.ctxt_emdb = DenseNetwork(inpt_dim=self.ctxt_dim, **ctxt_embd_config)
        self.ctxt_out = self.ctxt_emdb.outp_dim
        self.te = TransformerEncoder(**te_config, ctxt_dim=self.ctxt_out)
        self.model_dim = self.te.model_dim
        self.node_embd = DenseNetwork(inpt_dim=self.inpt_dim, outp_dim=self.model_dim, ctxt_dim=self.ctxt_out,
                                      **node_embd_config)
        self.outp_embd = DenseNetwork(inpt_dim=self.model_dim, outp_dim=self.outp_dim, ctxt_dim=self.ctxt_out,
                                      **outp_embd_config)
        self._init_fused(num_points, frequencies, add_time_to_input, t_emb)

    # -- layout / weights --------------------------------------------------------------------------
    def config(self, num_points: Optional[int] = None) -> TfConfig:
        t_dim = 2 * self.frequencies
        dense = self.te.layers[0].dense
        hid = {dense.hddn_dim[0], self.node_embd.hddn_dim[0], self.outp_embd.hddn_dim[0]}
        if len(hid) != 1:
            raise NotImplementedError("the HIP transformer path needs one hddn_dim for node_embd / dense / outp_embd")
        return TfConfig(num_particles=num_points or self.num_points, features=self.outp_dim, model_dim=self.model_dim,
                        num_layers=self.te.num_layers, num_heads=self.te.layers[0].self_attn.num_heads, hidden=hid.pop(),
                        ctxt_hidden=self.ctxt_emdb.hddn_dim[0], ctxt_dim=self.ctxt_out, frequencies=self.frequencies,
                        global_cond_dim=self.ctxt_dim - t_dim, add_time_to_input=self.add_time_to_input, t_emb=self.t_emb)

    _LAYOUT = TfLayout
    _forward_op = staticmethod(hip_ops_tf.tf_forward)


class FullCrossAttentionEncoder(_FusedEncoder):
    """droid_transformer.py:620-711; evaluation as FullTransformerEncoder (``vector_field``)."""

    _GRAPH_FLAG = 8  # PFM_CA_F_GRAPH_STEPS

    def __init__(self, inpt_dim: int, outp_dim: int, ctxt_dim: int = 0, cae_config: Mapping | None = None,
                 node_embd_config: Mapping | None = None, outp_embd_config: Mapping | None = None,
                 ctxt_embd_config: Mapping | None = None, *, num_points: int = 0, frequencies: int = 0,
                 add_time_to_input: bool = True, t_emb: str = "cosine") -> None:
        super().__init__()
        if not ctxt_dim:
            raise NotImplementedError("the HIP cross-attention path needs the context network (ctxt_dim > 0: it always is, "
                                      "CNF passes global_cond_dim + 2*frequencies)")
        self.inpt_dim, self.outp_dim, self.ctxt_dim = inpt_dim, outp_dim, ctxt_dim
        cae_config = deepcopy(cae_config) or {}
        node_embd_config = deepcopy(node_embd_config) or {}
        outp_embd_config = deepcopy(outp_embd_config) or {}
        ctxt_embd_config = deepcopy(ctxt_embd_config) or {}
        cae_config.setdefault("dense_config", {})
        if "model_dim" in cae_config:  # droid_transformer.py:660-669: dense nets default to twice the width
            model_dim = cae_config["model_dim"]
            for cfg in (node_embd_config, ctxt_embd_config, outp_embd_config, cae_config["dense_config"]):
                cfg.setdefault("hddn_dim", 2 * model_dim)
        self.ctxt_emdb = DenseNetwork(inpt_dim=self.ctxt_dim, **ctxt_embd_config)
        self.ctxt_out = self.ctxt_emdb.outp_dim
        self.cae = CrossAttentionEncoder(**cae_config, ctxt_dim=self.ctxt_out)
        self.model_dim = self.cae.model_dim
        self.node_embd = DenseNetwork(inpt_dim=self.inpt_dim, outp_dim=self.model_dim, ctxt_dim=self.ctxt_out,
                                      **node_embd_config)
        self.outp_embd = DenseNetwork(inpt_dim=self.model_dim, outp_dim=self.outp_dim, ctxt_dim=self.ctxt_out,
                                      **outp_embd_config)
        self._init_fused(num_points, frequencies, add_time_to_input, t_emb)

    def config(self, num_points: Optional[int] = None) -> CaConfig:
        t_dim = 2 * self.frequencies
        dense = self.cae.from_layers[0].dense
        hid = {dense.hddn_dim[0], self.node_embd.hddn_dim[0], self.outp_embd.hddn_dim[0]}
        if len(hid) != 1:
            raise NotImplementedError("the HIP cross-attention path needs one hddn_dim for node_embd / dense / outp_embd")
        return CaConfig(num_particles=num_points or self.num_points, features=self.outp_dim, model_dim=self.model_dim,
                        num_layers=self.cae.num_layers, num_heads=self.cae.from_layers[0].cross_attn.num_heads,
                        num_tokens=self.cae.num_tokens, hidden=hid.pop(), ctxt_hidden=self.ctxt_emdb.hddn_dim[0],
                        ctxt_dim=self.ctxt_out, frequencies=self.frequencies, global_cond_dim=self.ctxt_dim - t_dim,
                        add_time_to_input=self.add_time_to_input, t_emb=self.t_emb)

    _LAYOUT = CaLayout
    _forward_op = staticmethod(hip_ops_ca.ca_forward)
