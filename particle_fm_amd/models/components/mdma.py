"""Drop-in for particle_fm/models/components/mdma.py: ``MDMA`` and its ``Block``.

Same class names, constructor keywords, parameter names / shapes / registration order (hence the same ``state_dict`` keys and,
under a fixed seed, bit-identical default initialisation) as the reference (mdma.py:7-45, 87-140).  The blocks own parameters
only: one evaluation of the network is a fixed sequence of HIP launches (``vector_field``; include/pfm_mdma.h), there is no
per-block PyTorch compute and no CPU fallback.  The time concatenations (t_local_cat, t_global_cat: off in
configs/model/flow_matching_mdma.yaml, on by MDMA's own defaults) and the conditional variant (global_cond_dim = 1, local_cat_cond,
global_cat_cond: one condition value per jet, off in the yaml) have kernels; `dropout` is ignored as in the reference (mdma.py:121-135 builds the blocks with dropout=0).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ... import hip_ops_mdma
from ...layout_mdma import MdmaConfig, MdmaLayout
from .droid_transformer import _FusedEncoder, _container_only


class Block(nn.Module):
    """mdma.py:7-51 (parameters in the reference's registration order; ``cond_cls`` is constructed and never used there too)."""

    def __init__(self, embed_dim, num_heads, hidden, dropout, weightnorm=True, glu=False, critic=True, t_local_cat=False,
                 t_global_cat=False, global_cond_dim=0, frequencies=8, local_cat_cond=False, global_cat_cond=False):
        super().__init__()
        self.fc0 = nn.Linear(hidden + 2 * frequencies * t_local_cat + local_cat_cond, hidden)
        self.fc0_cls = nn.Linear(embed_dim + 2 * frequencies * t_global_cat + global_cat_cond, hidden)
        self.fc1 = nn.Linear(hidden + embed_dim + local_cat_cond, hidden)
        self.glu = False
        self.fc1_cls = nn.Linear(hidden + 1 + global_cond_dim + 2 * frequencies * t_global_cat, embed_dim)
        self.fc2_cls = nn.Linear(embed_dim + 2 * frequencies * t_global_cat + global_cat_cond, embed_dim)
        self.cond_cls = nn.Linear(global_cond_dim, hidden)
        self.attn = nn.MultiheadAttention(hidden, num_heads, batch_first=True)
        self.act = nn.LeakyReLU()
        self.ln = nn.LayerNorm(hidden)
        self.t_local_cat, self.t_global_cat = t_local_cat, t_global_cat
        self.local_cat_cond, self.global_cat_cond = local_cat_cond, global_cat_cond

    forward = _container_only("Block")


class MDMA(_FusedEncoder):
    """mdma.py:87-176.  ``vector_field(t, x, cond, mask)`` takes the time itself (B,) and the bare particle features and lets
    the kernels embed (what CNF.forward uses); it returns (B, N, 1) like the reference's forward.

    ``_cnf`` (keyword, filled in by CNF): what the kernels need to know beyond the reference's own arguments -- the CNF's
    ``features`` (net_config's ``feats`` is not used by MDMA), ``num_particles``, the time embedding's ``frequencies``,
    ``add_time_to_input`` and ``t_emb``."""

    def __init__(self, latent: int = 16, input_dim: int = 3, hidden_dim: int = 256, feats: int = 128, layers: int = 16,
                 global_cond_dim: int = 0, local_cond_dim: int = 0, activation: str = "leaky_relu", wrapper_func: str = "",
                 frequencies: int = 6, num_points: int = 30, t_local_cat: bool = True, t_global_cat: bool = True,
                 dropout: float = 0.0, sum_scale: float = 1e-2, avg_n: int = 30, num_heads: int = 8,
                 local_cat_cond: bool = False, global_cat_cond: bool = False, **kwargs):
        cnf = dict(kwargs.pop("_cnf", None) or {})
        super().__init__()
        if global_cond_dim not in (0, 1) or (global_cat_cond and global_cond_dim != 1):
            # MDMA.forward appends global_cond_in.unsqueeze(-1) -- ONE value per jet (mdma.py:157-169) -- and sizes embbed_cls by global_cond_dim
            raise NotImplementedError("MDMA(global_cond_dim) must be 0 or 1 (1 with global_cat_cond): the reference's own shapes")
        # (`dropout` is accepted and ignored, as in the reference: MDMA builds its blocks with dropout=0 and Block never reads it, mdma.py:121-135)
        self.t_local_cat, self.t_global_cat = t_local_cat, t_global_cat
        self.embed = nn.Linear(input_dim + 2 * frequencies * t_local_cat + local_cat_cond, hidden_dim)
        self.embbed_cls = nn.Linear(hidden_dim + 1 + global_cond_dim, latent)
        self.encoder = nn.ModuleList([
            Block(embed_dim=latent, num_heads=num_heads, hidden=hidden_dim, weightnorm=False, dropout=0, glu=False, critic=False,
                  t_local_cat=t_local_cat, t_global_cat=t_global_cat, global_cond_dim=global_cond_dim, frequencies=frequencies,
                  local_cat_cond=local_cat_cond, global_cat_cond=global_cat_cond) for _ in range(layers)])
        self.out = nn.Linear(hidden_dim + local_cat_cond, 1)
        self.act = nn.LeakyReLU()
        self.avg_n = avg_n
        self.local_cat_cond, self.global_cat_cond = local_cat_cond, global_cat_cond
        self.cond = nn.Linear(global_cond_dim + 1, latent)
        self.global_cond = global_cond_dim > 0
        self.global_cond_dim = global_cond_dim
        self.latent, self.hidden_dim, self.num_layers, self.num_heads = latent, hidden_dim, layers, num_heads
        add_time = bool(cnf.get("add_time_to_input", False))
        cnf_freq = int(cnf.get("frequencies", 0))
        self.features = int(cnf.get("features", input_dim - (2 * cnf_freq if add_time else 0)))
        if self.features + (2 * cnf_freq if add_time else 0) != input_dim:
            raise ValueError(f"input_dim {input_dim} is not features {self.features} + time columns {2 * cnf_freq if add_time else 0}")
        if (t_local_cat or t_global_cat) and cnf and frequencies != cnf_freq:
            raise ValueError(f"MDMA(t_local_cat / t_global_cat): frequencies {frequencies} sizes the Linears, the CNF's {cnf_freq} the "
                             "embedding they receive (mdma.py:25-36, flow_matching_module.py:208-221): the two must agree")
        self._init_fused(int(cnf.get("num_particles", num_points)), cnf_freq, add_time, str(cnf.get("t_emb", "cosine")))

    def config(self, num_points: Optional[int] = None) -> MdmaConfig:
        return MdmaConfig(num_particles=num_points or self.num_points, features=self.features, hidden=self.hidden_dim,
                          latent=self.latent, num_layers=self.num_layers, num_heads=self.num_heads, avg_n=float(self.avg_n),
                          frequencies=self.frequencies, add_time_to_input=self.add_time_to_input, t_emb=self.t_emb,
                          t_local_cat=bool(self.t_local_cat), t_global_cat=bool(self.t_global_cat),
                          global_cond_dim=int(self.global_cond_dim), global_cat_cond=bool(self.global_cat_cond),
                          local_cat_cond=bool(self.local_cat_cond))

    def layout(self, num_points: Optional[int] = None):
        n = num_points or self.num_points
        flags = 32 if getattr(self, "mfma_dtype", "fp32") == "bf16" else 0  # PFM_MDMA_F_BF16
        lay = self._layouts.get((n, flags))
        if lay is None:
            lay = self._layouts[(n, flags)] = MdmaLayout(self.config(n), flags=flags)
        return lay

    def set_precision(self, precision) -> None:
        """Lightning's "bf16", "bf16-mixed", "bf16-true" -> the particle-stream Linears (forward and dX) on bf16 operands with fp32
        accumulate (PFM_MDMA_F_BF16); the class token, the one-query attention and the dW GEMMs stay fp32.  Anything else: fp32."""
        self.mfma_dtype = "bf16" if str(precision).startswith("bf16") else "fp32"

    _LAYOUT = MdmaLayout

    @staticmethod
    def _forward_op(lay, blob, t, x, cond, mask):
        # `cond` is read by the conditional variant only (global_cond_dim = 1 / the *_cat_cond switches: one value per jet, mdma.py:157-169)
        return hip_ops_mdma.mdma_forward(lay, blob, t, x, mask, cond=cond if lay.cfg.needs_cond else None)[..., :1].contiguous()
