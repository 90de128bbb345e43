"""Host-side weight layout of the cross-attention vector field for libpfm_hip.so (formats: include/pfm_ca.h).

Reference parameters: ``net.{ctxt_emdb, cae.global_tokens, cae.from_layers.*, cae.to_layers.*, node_embd, outp_embd}``
(droid_transformer.py:560-618 CrossAttentionEncoder.__init__, :626-700 FullCrossAttentionEncoder.__init__).  Same gather-map
scheme as layout_tf.TfLayout (whose helpers this reuses); the k_linear and v_linear of a layer are packed as ONE [2D][D]
matrix so that keys and values come out of a single GEMM.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Mapping, Tuple

import numpy as np

from .layout_tf import TfLayout, TfLin, TfNorm, default_freqs  # noqa: F401  (default_freqs re-exported)

PFM_CA_ABI_VERSION = 1
PFM_CA_MAX_LAYERS = 16
PFM_CA_MAX_TOKENS = 8
PFM_CA_F_F16X3 = 1
PFM_CA_F_TEMB_SINCOS = 2
PFM_CA_F_VALID_ROWS = 4
PFM_CA_F_GRAPH_STEPS = 8


class CaLayer(ctypes.Structure):
    _fields_ = [("norm0", TfNorm), ("norm1", TfNorm), ("norm2", TfNorm), ("attn_norm", TfNorm), ("d_norm", TfNorm),
                ("q", TfLin), ("kv", TfLin), ("out", TfLin), ("d1", TfLin), ("d2", TfLin)]


class CaDesc(ctypes.Structure):
    """ctypes mirror of ``pfm_ca_desc`` (include/pfm_ca.h)."""

    _fields_ = [
        ("abi_version", ctypes.c_int32), ("n_points", ctypes.c_int32), ("features", ctypes.c_int32),
        ("model_dim", ctypes.c_int32), ("hidden", ctypes.c_int32), ("layers", ctypes.c_int32),
        ("heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("tokens", ctypes.c_int32), ("t_dim", ctypes.c_int32),
        ("cond_dim", ctypes.c_int32), ("ctxt_dim", ctypes.c_int32), ("ctxt_hidden", ctypes.c_int32),
        ("time_in_input", ctypes.c_int32), ("flags", ctypes.c_int32), ("pad_", ctypes.c_int32),
        ("neg_slope", ctypes.c_float), ("ln_eps", ctypes.c_float),
        ("blob_floats", ctypes.c_int64), ("freqs", ctypes.c_int64), ("global_tokens", ctypes.c_int64),
        ("c1", TfLin), ("c_norm", TfNorm), ("c2", TfLin),
        ("n1", TfLin), ("n_norm", TfNorm), ("n2", TfLin),
        ("from_layer", CaLayer * PFM_CA_MAX_LAYERS), ("to_layer", CaLayer * PFM_CA_MAX_LAYERS),
        ("o1", TfLin), ("o_norm", TfNorm), ("o2", TfLin),
    ]


@dataclass(frozen=True)
class CaConfig:
    """The hyper-parameters of CNF(model="droid_fullcrossattention") that shape the network
    (flow_matching_module.py:159-165, configs/model/fm_droid_crossattention.yaml:15-40)."""

    num_particles: int
    features: int = 3
    model_dim: int = 128
    num_layers: int = 8       # from/to layer pairs
    num_heads: int = 16
    num_tokens: int = 4
    hidden: int = 256         # hddn_dim of node_embd / dense / outp_embd (reference default 2 * model_dim)
    ctxt_hidden: int = 256
    ctxt_dim: int = 64
    frequencies: int = 16
    global_cond_dim: int = 0
    add_time_to_input: bool = True
    t_emb: str = "cosine"

    @property
    def t_dim(self) -> int:
        return 2 * self.frequencies

    @property
    def head_dim(self) -> int:
        return self.model_dim // self.num_heads

    @staticmethod
    def from_hparams(hp: Mapping) -> "CaConfig":
        nc = hp.get("net_config") or {}
        cae = dict(nc.get("cae_config") or {})
        D = int(cae.get("model_dim", 64))
        mha = dict(cae.get("mha_config") or {})
        dense = dict(cae.get("dense_config") or {})
        node = dict(nc.get("node_embd_config") or {})
        outp = dict(nc.get("outp_embd_config") or {})
        ctxt = dict(nc.get("ctxt_embd_config") or {})
        hid = {int(c.get("hddn_dim", 2 * D)) for c in (dense, node, outp)}
        if len(hid) != 1:
            raise NotImplementedError("the HIP cross-attention path needs one hddn_dim for node_embd / dense / outp_embd")
        for c, who in ((dense, "dense_config"), (node, "node_embd_config"), (outp, "outp_embd_config"), (ctxt, "ctxt_embd_config")):
            if c.get("act_h", "lrlu") != "lrlu" or c.get("nrm", "none") != "layer" or c.get("num_blocks", 1) != 1 \
                    or not isinstance(c.get("hddn_dim", 0), int) or c.get("drp", 0):
                raise NotImplementedError(f"{who}: the HIP path implements act_h='lrlu', nrm='layer', one hidden block, no dropout")
        if not mha.get("do_layer_norm", False) or mha.get("drp", 0) or mha.get("attn_act") is not None:
            raise NotImplementedError("mha_config: the HIP path implements do_layer_norm=True, softmax attention, no dropout")
        if not ctxt.get("outp_dim"):
            raise NotImplementedError("ctxt_embd_config.outp_dim must be given")
        return CaConfig(
            num_particles=int(hp["num_particles"]), features=int(hp.get("features", 3)), model_dim=D,
            num_layers=int(cae.get("num_layers", 5)), num_heads=int(mha.get("num_heads", 1)),
            num_tokens=int(cae.get("num_tokens", 4)), hidden=hid.pop(), ctxt_hidden=int(ctxt.get("hddn_dim", 2 * D)),
            ctxt_dim=int(ctxt["outp_dim"]), frequencies=int(hp.get("frequencies", 6)),
            global_cond_dim=int(hp.get("global_cond_dim", 0)), add_time_to_input=bool(hp.get("add_time_to_input", False)),
            t_emb=str(hp.get("t_emb", "cosine")),
        )

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """(key, shape) in the reference's state_dict order: ctxt_emdb, cae (global_tokens, from_layers, to_layers),
        node_embd, outp_embd; inside a layer cross_attn, dense, norm0, norm1, norm2 (droid_transformer.py:366-378)."""
        D, Hd, CO, CH, T = self.model_dim, self.hidden, self.ctxt_dim, self.ctxt_hidden, self.t_dim
        d_in = self.features + (T if self.add_time_to_input else 0)
        out: List[Tuple[str, Tuple[int, ...]]] = []

        def lin(k, o, i):
            out.extend([(k + ".weight", (o, i)), (k + ".bias", (o,))])

        def ln(k, n):
            out.extend([(k + ".weight", (n,)), (k + ".bias", (n,))])

        def dense(k, i, h, o, ctxt):
            lin(k + ".input_block.block.0", h, i + ctxt)
            ln(k + ".input_block.block.2", h)
            lin(k + ".output_block.block.0", o, h)

        dense("net.ctxt_emdb", T + self.global_cond_dim, CH, CO, 0)
        out.append(("net.cae.global_tokens", (1, self.num_tokens, D)))
        for group in ("from_layers", "to_layers"):
            for l in range(self.num_layers):
                p = f"net.cae.{group}.{l}."
                for nm in ("q_linear", "k_linear", "v_linear"):
                    lin(p + "cross_attn." + nm, D, D)
                ln(p + "cross_attn.layer_norm", D)
                lin(p + "cross_attn.out_linear", D, D)
                dense(p + "dense", D, Hd, D, CO)
                ln(p + "norm0", D)
                ln(p + "norm1", D)
                ln(p + "norm2", D)
        dense("net.node_embd", d_in, Hd, D, CO)
        dense("net.outp_embd", D, Hd, self.features, CO)
        return out

    def param_count(self) -> int:
        return sum(int(np.prod(s)) for _, s in self.param_shapes())


class CaLayout(TfLayout):
    """Descriptor + gather maps for one CaConfig."""

    def __init__(self, cfg: CaConfig, flags: int = 0):
        self.flags = flags
        D, Hd = cfg.model_dim, cfg.hidden
        if D % 128 or Hd % 128 or D > 512 or Hd > 512:
            raise NotImplementedError("the HIP kernels need model_dim and hddn_dim to be multiples of 128, at most 512")
        if cfg.num_heads * cfg.head_dim != D or cfg.head_dim not in (8, 16) or cfg.num_heads > 64:
            raise NotImplementedError("the HIP cross-attention kernels are specialised for head_dim 8 and 16")
        if not 1 <= cfg.num_tokens <= PFM_CA_MAX_TOKENS or cfg.num_tokens * cfg.head_dim > 64:
            raise NotImplementedError("num_tokens must be in 1..8 with num_tokens * head_dim <= 64")
        if not 1 <= cfg.num_layers <= PFM_CA_MAX_LAYERS:
            raise NotImplementedError(f"num_layers must be in 1..{PFM_CA_MAX_LAYERS}")
        if cfg.ctxt_dim % 4 or cfg.ctxt_dim > 64 or cfg.ctxt_hidden % 4 or cfg.ctxt_hidden > 512:
            raise NotImplementedError("ctxt_emdb: outp_dim must be a multiple of 4 (<= 64), hddn_dim a multiple of 4 (<= 512)")
        if cfg.features > 16 or cfg.global_cond_dim > 16 or cfg.t_dim > 64:
            raise NotImplementedError("limits of this build: features <= 16, global_cond_dim <= 16, frequencies <= 32")
        self._init_params(cfg)
        self._build()

    def _mfma_ak_stack(self, keys, K: int, transposed: bool = False) -> int:
        """MFMA_AK of the row-wise stack of the Linears ``keys`` (each NO x K), or of its transpose."""
        each = self._shape[keys[0] + ".weight"][0]
        NO = each * len(keys)
        base = np.array([self.p_off[k + ".weight"] for k in keys], dtype=np.int64)
        rows, red = (K, NO) if transposed else (NO, K)
        ob = np.arange(rows // 16)[:, None, None, None, None]
        kc = np.arange(red // 128)[None, :, None, None, None]
        kt = np.arange(8)[None, None, :, None, None]
        lane = np.arange(64)[None, None, None, :, None]
        r = np.arange(4)[None, None, None, None, :]
        i = 16 * ob + (lane & 15) + 0 * (kc + kt + r)
        k = 128 * kc + 16 * kt + 4 * (lane >> 4) + r + 0 * ob
        o, c = (k, i) if transposed else (i, k)  # element W_stack[o][c]
        return self._put(base[o // each] + (o % each) * K + c, primary=not transposed)

    def _layer(self, L: CaLayer, p: str):
        cfg = self.cfg
        D, Hd, CO = cfg.model_dim, cfg.hidden, cfg.ctxt_dim
        L.norm0, L.norm1, L.norm2 = self._norm(p + "norm0"), self._norm(p + "norm1"), self._norm(p + "norm2")
        L.attn_norm = self._norm(p + "cross_attn.layer_norm")
        k = p + "cross_attn.q_linear"
        L.q = TfLin(self._mfma_ak(k, 0, D), -1, -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, D, True))
        ks = [p + "cross_attn.k_linear", p + "cross_attn.v_linear"]
        bias = self._put(np.concatenate([self.p_off[k + ".bias"] + np.arange(D) for k in ks]))
        L.kv = TfLin(self._mfma_ak_stack(ks, D), -1, -1, bias, self._mfma_ak_stack(ks, D, True))
        k = p + "cross_attn.out_linear"
        L.out = TfLin(self._mfma_ak(k, 0, D), -1, -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, D, True))
        k = p + "dense.input_block.block.0"
        L.d1 = TfLin(self._mfma_ak(k, 0, D), self._kmajor(k, D, CO), -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, D, True))
        L.d_norm = self._norm(p + "dense.input_block.block.2")
        k = p + "dense.output_block.block.0"
        L.d2 = TfLin(self._mfma_ak(k, 0, Hd), -1, -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, Hd, True))

    def _build(self):
        cfg = self.cfg
        D, Hd, CO, T, F = cfg.model_dim, cfg.hidden, cfg.ctxt_dim, cfg.t_dim, cfg.features
        d = CaDesc()
        d.abi_version = PFM_CA_ABI_VERSION
        d.n_points, d.features, d.model_dim, d.hidden, d.layers = cfg.num_particles, F, D, Hd, cfg.num_layers
        d.heads, d.head_dim, d.tokens, d.t_dim, d.cond_dim = cfg.num_heads, cfg.head_dim, cfg.num_tokens, T, cfg.global_cond_dim
        d.ctxt_dim, d.ctxt_hidden, d.time_in_input = CO, cfg.ctxt_hidden, int(cfg.add_time_to_input)
        d.flags = self.flags | (PFM_CA_F_TEMB_SINCOS if cfg.t_emb == "sincos" else 0) | (64 if cfg.t_emb == "gaussian" else 0)  # 64: PFM_*_F_TEMB_GIVEN
        d.neg_slope, d.ln_eps = 0.1, 1e-5
        d.freqs = self._put(self.freq_off + np.arange(T), primary=False)
        d.global_tokens = self._put(self.p_off["net.cae.global_tokens"] + np.arange(cfg.num_tokens * D))

        k = "net.ctxt_emdb.input_block.block.0"
        d.c1 = TfLin(self._kmajor(k, 0, T + cfg.global_cond_dim), -1, -1, self._vec(k + ".bias"), -1)
        d.c_norm = self._norm("net.ctxt_emdb.input_block.block.2")
        k = "net.ctxt_emdb.output_block.block.0"
        d.c2 = TfLin(self._kmajor(k, 0, cfg.ctxt_hidden), -1, -1, self._vec(k + ".bias"), -1)

        k = "net.node_embd.input_block.block.0"
        t0 = T if cfg.add_time_to_input else 0
        d.n1 = TfLin(self._kmajor(k, t0, F), self._kmajor(k, t0 + F, CO),
                     self._kmajor(k, 0, T) if cfg.add_time_to_input else -1, self._vec(k + ".bias"), -1)
        d.n_norm = self._norm("net.node_embd.input_block.block.2")
        k = "net.node_embd.output_block.block.0"
        d.n2 = TfLin(self._mfma_ak(k, 0, Hd), -1, -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, Hd, True))

        for l in range(cfg.num_layers):
            self._layer(d.from_layer[l], f"net.cae.from_layers.{l}.")
            self._layer(d.to_layer[l], f"net.cae.to_layers.{l}.")

        k = "net.outp_embd.input_block.block.0"
        d.o1 = TfLin(self._mfma_ak(k, 0, D), self._kmajor(k, D, CO), -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, D, True))
        d.o_norm = self._norm("net.outp_embd.input_block.block.2")
        k = "net.outp_embd.output_block.block.0"
        d.o2 = TfLin(self._put(self._w(k, np.arange(F)[:, None], np.arange(Hd)[None, :])), -1, -1, self._vec(k + ".bias"), -1)
        self._finish(d)
