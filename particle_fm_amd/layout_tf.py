"""Host-side weight layout of the Full-Transformer vector field for libpfm_hip.so (formats: include/pfm_tf.h).

The reference stores the network as plain ``nn.Linear`` / ``nn.LayerNorm`` parameters under
``net.{ctxt_emdb,te,node_embd,outp_embd}.*`` (droid_transformer.py:440-527; 68 tensors for three layers).  The
kernels want every Linear split into the block that multiplies per-particle activations (MFMA operand order) and
the columns that multiply per-jet vectors (context, time embedding; K-major).  As for EPiC (layout.py) this module
computes one int64 gather map per configuration so that ``blob = source[index_map]`` with ``source`` the
concatenation of all parameters in state_dict order, the frequency table and a zero; the gradient of the blob goes
back to the parameters through ``grad_pos`` (every parameter element has exactly one primary blob slot).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Mapping, Tuple

import numpy as np
import torch

PFM_TF_ABI_VERSION = 1
PFM_TF_MAX_LAYERS = 12
HEAD_DIM = 16
PFM_TF_F_F16X3 = 1
PFM_TF_F_TEMB_SINCOS = 2
PFM_TF_F_VALID_ROWS = 4
PFM_TF_F_ONE_STREAM = 16  # the midpoint sampler stays on the caller's stream (callers with several calls in flight)


class TfNorm(ctypes.Structure):
    _fields_ = [("gamma", ctypes.c_int64), ("beta", ctypes.c_int64)]


class TfLin(ctypes.Structure):
    _fields_ = [("W", ctypes.c_int64), ("Wc", ctypes.c_int64), ("Wt", ctypes.c_int64), ("b", ctypes.c_int64),
                ("WT", ctypes.c_int64)]


class TfLayer(ctypes.Structure):
    _fields_ = [("norm1", TfNorm), ("qkv", TfLin), ("attn_norm", TfNorm), ("out", TfLin), ("norm2", TfNorm),
                ("d1", TfLin), ("d_norm", TfNorm), ("d2", TfLin)]


class TfDesc(ctypes.Structure):
    """ctypes mirror of ``pfm_tf_desc`` (include/pfm_tf.h)."""

    _fields_ = [
        ("abi_version", ctypes.c_int32), ("n_points", ctypes.c_int32), ("features", ctypes.c_int32),
        ("model_dim", ctypes.c_int32), ("hidden", ctypes.c_int32), ("layers", ctypes.c_int32),
        ("heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("t_dim", ctypes.c_int32),
        ("cond_dim", ctypes.c_int32), ("ctxt_dim", ctypes.c_int32), ("ctxt_hidden", ctypes.c_int32),
        ("time_in_input", ctypes.c_int32), ("flags", ctypes.c_int32),
        ("neg_slope", ctypes.c_float), ("ln_eps", ctypes.c_float),
        ("blob_floats", ctypes.c_int64), ("freqs", ctypes.c_int64),
        ("c1", TfLin), ("c_norm", TfNorm), ("c2", TfLin),
        ("n1", TfLin), ("n_norm", TfNorm), ("n2", TfLin),
        ("layer", TfLayer * PFM_TF_MAX_LAYERS),
        ("final_norm", TfNorm), ("o1", TfLin), ("o_norm", TfNorm), ("o2", TfLin),
    ]


@dataclass(frozen=True)
class TfConfig:
    """The hyper-parameters of CNF(model="droid_fulltransformer") that shape the network
    (flow_matching_module.py:152-158, configs/model/fm_droid_transformer.yaml:15-44)."""

    num_particles: int
    features: int = 3
    model_dim: int = 256
    num_layers: int = 3
    num_heads: int = 16
    hidden: int = 512         # hddn_dim of node_embd / dense / outp_embd (reference default 2 * model_dim)
    ctxt_hidden: int = 512    # hddn_dim of ctxt_emdb (same default)
    ctxt_dim: int = 64        # ctxt_embd_config.outp_dim
    frequencies: int = 16
    global_cond_dim: int = 0
    add_time_to_input: bool = True
    t_emb: str = "cosine"   # or "sincos" (flow_matching_module.py:208-211)

    @property
    def t_dim(self) -> int:
        return 2 * self.frequencies

    @staticmethod
    def from_hparams(hp: Mapping) -> "TfConfig":
        nc = hp.get("net_config") or {}
        te = dict(nc.get("te_config") or {})
        D = int(te.get("model_dim", 64))
        mha = dict(te.get("mha_config") or {})
        dense = dict(te.get("dense_config") or {})
        node = dict(nc.get("node_embd_config") or {})
        outp = dict(nc.get("outp_embd_config") or {})
        ctxt = dict(nc.get("ctxt_embd_config") or {})
        hid = {int(c.get("hddn_dim", 2 * D)) for c in (dense, node, outp)}
        if len(hid) != 1:
            raise NotImplementedError("the HIP transformer path needs one hddn_dim for node_embd / dense / outp_embd")
        for c, who in ((dense, "dense_config"), (node, "node_embd_config"), (outp, "outp_embd_config"), (ctxt, "ctxt_embd_config")):
            if c.get("act_h", "lrlu") != "lrlu" or c.get("nrm", "none") != "layer" or c.get("num_blocks", 1) != 1 \
                    or not isinstance(c.get("hddn_dim", 0), int) or c.get("drp", 0):
                raise NotImplementedError(f"{who}: the HIP path implements act_h='lrlu', nrm='layer', one hidden block, no dropout")
        if not mha.get("do_layer_norm", False) or mha.get("drp", 0) or mha.get("attn_act") is not None:
            raise NotImplementedError("mha_config: the HIP path implements do_layer_norm=True, softmax attention, no dropout")
        if not ctxt.get("outp_dim"):
            raise NotImplementedError("ctxt_embd_config.outp_dim must be given")
        return TfConfig(
            num_particles=int(hp["num_particles"]), features=int(hp.get("features", 3)), model_dim=D,
            num_layers=int(te.get("num_layers", 3)), num_heads=int(mha.get("num_heads", 1)), hidden=hid.pop(),
            ctxt_hidden=int(ctxt.get("hddn_dim", 2 * D)), ctxt_dim=int(ctxt["outp_dim"]),
            frequencies=int(hp.get("frequencies", 6)), global_cond_dim=int(hp.get("global_cond_dim", 0)),
            add_time_to_input=bool(hp.get("add_time_to_input", False)), t_emb=str(hp.get("t_emb", "cosine")),
        )

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """(key, shape) in the reference's state_dict order (module registration order of
        FullTransformerEncoder.__init__, droid_transformer.py:494-526: ctxt_emdb, te, node_embd, outp_embd)."""
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
        for l in range(self.num_layers):
            p = f"net.te.layers.{l}."
            lin(p + "self_attn.all_linear", 3 * D, D)
            ln(p + "self_attn.layer_norm", D)
            lin(p + "self_attn.out_linear", D, D)
            dense(p + "dense", D, Hd, D, CO)
            ln(p + "norm1", D)
            ln(p + "norm2", D)
        ln("net.te.final_norm", D)
        dense("net.node_embd", d_in, Hd, D, CO)
        dense("net.outp_embd", D, Hd, self.features, CO)
        return out

    def param_count(self) -> int:
        return sum(int(np.prod(s)) for _, s in self.param_shapes())


def default_freqs(t_dim: int, t_emb: str = "cosine") -> torch.Tensor:
    """cosine: exp(0..T-1) of time_emb.py:90 as correctly rounded fp32 (see EpicLayout.default_freqs: the reference's fp32
    ``arange(T).exp()`` is host-dependent in the last bit, so the product fixes the table).
    sincos: [f ; f] with f = 2^k pi, the module buffer of flow_matching_module.py:172."""
    if t_emb == "sincos":
        f = (2 ** torch.arange(t_dim // 2) * torch.pi).to(torch.float32)
        return torch.cat([f, f])
    return torch.arange(t_dim, dtype=torch.float64).exp().to(torch.float32)


class TfLayout:
    """Descriptor + gather maps for one TfConfig."""

    def __init__(self, cfg: TfConfig, flags: int = 0):
        self.flags = flags
        D, Hd = cfg.model_dim, cfg.hidden
        if D % 128 or Hd % 128 or D > 512 or Hd > 512:
            raise NotImplementedError("the HIP transformer kernels need model_dim and hddn_dim to be multiples of 128, at most 512")
        if cfg.num_heads * HEAD_DIM != D:
            raise NotImplementedError(f"the HIP attention kernel is specialised for head_dim {HEAD_DIM} (num_heads = model_dim / 16)")
        if cfg.num_layers > PFM_TF_MAX_LAYERS:
            raise NotImplementedError(f"num_layers > {PFM_TF_MAX_LAYERS}")
        if cfg.ctxt_dim % 4 or cfg.ctxt_dim > 64 or cfg.ctxt_hidden % 4 or cfg.ctxt_hidden > 512:
            raise NotImplementedError("ctxt_emdb: outp_dim must be a multiple of 4 (<= 64), hddn_dim a multiple of 4 (<= 512)")
        if cfg.num_particles > 512 or cfg.features > 16 or cfg.global_cond_dim > 16 or cfg.t_dim > 64:
            raise NotImplementedError("limits of this build: num_particles <= 512, features <= 16, global_cond_dim <= 16, frequencies <= 32")
        self._init_params(cfg)
        self._build()

    def _init_params(self, cfg):
        """Offsets of the parameters in the source vector (state_dict order), then the frequency table and a zero."""
        self.cfg = cfg
        self.shapes = cfg.param_shapes()
        self.p_off: Dict[str, int] = {}
        o = 0
        for k, s in self.shapes:
            self.p_off[k] = o
            o += int(np.prod(s))
        self.n_params = o
        self.freq_off = o
        o += cfg.t_dim
        self.zero_off = o
        self.n_source = o + 1
        self._shape = dict(self.shapes)
        self._cursor = 0
        self._segments: List[Tuple[int, np.ndarray, bool]] = []

    # ---- helpers ------------------------------------------------------------------------------
    def _alloc(self, n: int) -> int:
        off = self._cursor
        self._cursor += (n + 3) & ~3
        return off

    def _put(self, idx: np.ndarray, primary: bool = True) -> int:
        flat = np.asarray(idx, dtype=np.int64).reshape(-1)
        off = self._alloc(flat.size)
        self._segments.append((off, flat, primary))
        return off

    def _w(self, key: str, rows, cols):
        return self.p_off[key + ".weight"] + rows * self._shape[key + ".weight"][1] + cols

    def _vec(self, key: str) -> int:
        n = self._shape[key][0]
        return self._put(self.p_off[key] + np.arange(n))

    def _norm(self, key: str) -> TfNorm:
        return TfNorm(self._vec(key + ".weight"), self._vec(key + ".bias"))

    def _mfma_ak(self, key: str, c0: int, K: int, transposed: bool = False) -> int:
        """MFMA_AK of W[:, c0:c0+K] (NO x K), or of its transpose (K x NO) when ``transposed``."""
        NO = self._shape[key + ".weight"][0]
        rows, red = (K, NO) if transposed else (NO, K)
        assert rows % 16 == 0 and red % 128 == 0, (key, rows, red)
        ob = np.arange(rows // 16)[:, None, None, None, None]
        kc = np.arange(red // 128)[None, :, None, None, None]
        kt = np.arange(8)[None, None, :, None, None]
        lane = np.arange(64)[None, None, None, :, None]
        r = np.arange(4)[None, None, None, None, :]
        i = 16 * ob + (lane & 15) + 0 * (kc + kt + r)
        k = 128 * kc + 16 * kt + 4 * (lane >> 4) + r + 0 * ob
        src = self._w(key, k, c0 + i) if transposed else self._w(key, i, c0 + k)
        return self._put(src, primary=not transposed)

    def _kmajor(self, key: str, c0: int, K: int) -> int:
        NO = self._shape[key + ".weight"][0]
        return self._put(self._w(key, np.arange(NO)[None, :], c0 + np.arange(K)[:, None]))

    def _build(self):
        cfg = self.cfg
        D, Hd, CO, T, F = cfg.model_dim, cfg.hidden, cfg.ctxt_dim, cfg.t_dim, cfg.features
        d = TfDesc()
        d.abi_version = PFM_TF_ABI_VERSION
        d.n_points, d.features, d.model_dim, d.hidden, d.layers = cfg.num_particles, F, D, Hd, cfg.num_layers
        d.heads, d.head_dim, d.t_dim, d.cond_dim = cfg.num_heads, HEAD_DIM, T, cfg.global_cond_dim
        d.ctxt_dim, d.ctxt_hidden, d.time_in_input, d.flags = CO, cfg.ctxt_hidden, int(cfg.add_time_to_input), self.flags | (PFM_TF_F_TEMB_SINCOS if cfg.t_emb == "sincos" else 0) | (64 if cfg.t_emb == "gaussian" else 0)  # 64: PFM_*_F_TEMB_GIVEN
        d.neg_slope, d.ln_eps = 0.1, 1e-5
        d.freqs = self._put(self.freq_off + np.arange(T), primary=False)

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
            p = f"net.te.layers.{l}."
            L = d.layer[l]
            L.norm1 = self._norm(p + "norm1")
            k = p + "self_attn.all_linear"
            L.qkv = TfLin(self._mfma_ak(k, 0, D), -1, -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, D, True))
            L.attn_norm = self._norm(p + "self_attn.layer_norm")
            k = p + "self_attn.out_linear"
            L.out = TfLin(self._mfma_ak(k, 0, D), -1, -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, D, True))
            L.norm2 = self._norm(p + "norm2")
            k = p + "dense.input_block.block.0"
            L.d1 = TfLin(self._mfma_ak(k, 0, D), self._kmajor(k, D, CO), -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, D, True))
            L.d_norm = self._norm(p + "dense.input_block.block.2")
            k = p + "dense.output_block.block.0"
            L.d2 = TfLin(self._mfma_ak(k, 0, Hd), -1, -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, Hd, True))

        d.final_norm = self._norm("net.te.final_norm")
        k = "net.outp_embd.input_block.block.0"
        d.o1 = TfLin(self._mfma_ak(k, 0, D), self._kmajor(k, D, CO), -1, self._vec(k + ".bias"), self._mfma_ak(k, 0, D, True))
        d.o_norm = self._norm("net.outp_embd.input_block.block.2")
        k = "net.outp_embd.output_block.block.0"
        d.o2 = TfLin(self._put(self._w(k, np.arange(F)[:, None], np.arange(Hd)[None, :])), -1, -1, self._vec(k + ".bias"), -1)

        self._finish(d)

    def _finish(self, d):
        d.blob_floats = self._cursor
        self.desc = d
        idx = np.full(self._cursor, self.zero_off, dtype=np.int64)
        gpos = np.full(self.n_params, -1, dtype=np.int64)
        for off, flat, primary in self._segments:
            idx[off:off + flat.size] = flat
            if primary:
                assert (gpos[flat] == -1).all(), "parameter element with two primary slots"
                gpos[flat] = off + np.arange(flat.size)
        assert (gpos >= 0).all(), "parameter element without a blob slot"
        self.index_map = idx
        self.grad_pos = gpos  # blob slot holding d loss / d param element
        del self._segments

    # ---- packing ------------------------------------------------------------------------------
    @property
    def blob_total(self) -> int:
        return int(self.desc.blob_floats)

    def keys(self, prefix: str = "") -> List[str]:
        return [prefix + k for k, _ in self.shapes]

    def source_vector(self, state: Mapping[str, torch.Tensor], prefix: str = "", freqs=None) -> torch.Tensor:
        parts = [state[prefix + k].reshape(-1).to(torch.float32) for k, _ in self.shapes]
        dev = parts[0].device
        f = default_freqs(self.cfg.t_dim, self.cfg.t_emb) if freqs is None else freqs
        if self.cfg.t_emb == "sincos" and f.numel() == self.cfg.frequencies:
            f = torch.cat([f, f])  # the module buffer holds f once
        parts.append(f.to(device=dev, dtype=torch.float32).reshape(-1))
        parts.append(torch.zeros(1, device=dev))
        return torch.cat(parts)

    def index_map_on(self, device) -> torch.Tensor:
        cache = self.__dict__.setdefault("_imap", {})
        key = str(device)
        if key not in cache:
            cache[key] = torch.from_numpy(self.index_map).to(device)
        return cache[key]

    def grad_pos_on(self, device) -> torch.Tensor:
        cache = self.__dict__.setdefault("_gpos", {})
        key = str(device)
        if key not in cache:
            cache[key] = torch.from_numpy(self.grad_pos).to(device)
        return cache[key]

    def pack_blob(self, state: Mapping[str, torch.Tensor], prefix: str = "", freqs=None) -> torch.Tensor:
        src = self.source_vector(state, prefix, freqs)
        return src[self.index_map_on(src.device)]
