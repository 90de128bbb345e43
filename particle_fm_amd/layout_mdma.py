"""Host-side weight layout of the MDMA vector field for libpfm_hip.so (formats: include/pfm_mdma.h).

Reference parameters: ``net.{embed, embbed_cls, encoder.<i>.{fc0, fc0_cls, fc1, fc1_cls, fc2_cls, cond_cls, attn, ln}, out, cond}``
(mdma.py:113-140 MDMA.__init__, :24-45 Block.__init__).  Same gather-map scheme as layout_tf.TfLayout (whose helpers this
reuses): the blob is ``source[index_map]`` with source = parameters in state_dict order | frequency table | 0.

``attn.in_proj_weight`` ([3H][H], rows q | k | v) is split: the query rows are a KMAJOR block for the per-jet token kernel, the
key | value rows one MFMA_AK [2H][H] matrix for the particle-row GEMM.  ``fc1.weight`` ([H][H + L]) likewise: the particle
columns MFMA_AK, the token columns KMAJOR.  ``cond_cls`` (Linear(global_cond_dim = 0, H)) is never used by Block.forward
(``self.glu = False``, mdma.py:30): its bias gets a blob slot only so that every parameter element has one.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Mapping, Tuple

import numpy as np

from .layout_tf import TfLayout, TfLin, default_freqs  # noqa: F401  (default_freqs re-exported)

PFM_MDMA_ABI_VERSION = 3
PFM_MDMA_MAX_LAYERS = 16
PFM_MDMA_F_TEMB_SINCOS = 2


class MdmaBlock(ctypes.Structure):
    _fields_ = [("fc0", TfLin), ("kv", TfLin), ("fc1", TfLin)] + [
        (n, ctypes.c_int64) for n in ("fc0c_W", "fc0c_b", "ln_g", "ln_b", "q_W", "q_b", "o_W", "o_b", "fc1c_W", "fc1c_b", "fc2c_W", "fc2c_b")]


class MdmaDesc(ctypes.Structure):
    """ctypes mirror of ``pfm_mdma_desc`` (include/pfm_mdma.h)."""

    _fields_ = [
        ("abi_version", ctypes.c_int32), ("n_points", ctypes.c_int32), ("features", ctypes.c_int32), ("hidden", ctypes.c_int32),
        ("latent", ctypes.c_int32), ("layers", ctypes.c_int32), ("heads", ctypes.c_int32), ("head_dim", ctypes.c_int32),
        ("t_dim", ctypes.c_int32), ("time_in_input", ctypes.c_int32), ("flags", ctypes.c_uint32), ("t_cat", ctypes.c_int32),
        ("neg_slope", ctypes.c_float), ("ln_eps", ctypes.c_float), ("avg_n", ctypes.c_float), ("c_cat", ctypes.c_int32),
        ("blob_floats", ctypes.c_int64), ("freqs", ctypes.c_int64),
        ("emb_Wx", ctypes.c_int64), ("emb_Wt", ctypes.c_int64), ("emb_b", ctypes.c_int64), ("emb_Wt2", ctypes.c_int64),
        ("emb_Wc", ctypes.c_int64), ("out_Wc", ctypes.c_int64),
        ("ecls_W", ctypes.c_int64), ("ecls_b", ctypes.c_int64), ("cond_W", ctypes.c_int64), ("cond_b", ctypes.c_int64),
        ("out_W", ctypes.c_int64), ("out_b", ctypes.c_int64),
        ("block", MdmaBlock * PFM_MDMA_MAX_LAYERS),
    ]


@dataclass(frozen=True)
class MdmaConfig:
    """The hyper-parameters of CNF(model="mdma") that shape the network (flow_matching_module.py:163-167 passes
    ``input_dim`` and ``**net_config`` to MDMA; configs/model/flow_matching_mdma.yaml:15-30)."""

    num_particles: int
    features: int = 3
    hidden: int = 256
    latent: int = 16
    num_layers: int = 16
    num_heads: int = 8
    avg_n: float = 30.0
    frequencies: int = 6            # of the CNF's time embedding (the net_config entry of the same name is unused here)
    add_time_to_input: bool = True
    t_emb: str = "cosine"
    t_local_cat: bool = False       # mdma.py:56-57, 155-156: the time embedding concatenated to the inputs of embed and of every Block.fc0
    t_global_cat: bool = False      # mdma.py:58-59, 71-78: ... to the class-token Linears fc0_cls, fc1_cls, fc2_cls
    global_cond_dim: int = 0        # net_config's (0 or 1): one condition value per jet behind the particle count (mdma.py:164-169, 70-74)
    global_cat_cond: bool = False   # mdma.py:60-61, 79: cond[..., -1:] appended to the inputs of fc0_cls and fc2_cls
    local_cat_cond: bool = False    # mdma.py:62-63, 81-82, 157-158, 173-174: ... of fc0 and fc1 (the condition itself for embed and out)

    @property
    def t_dim(self) -> int:
        return 2 * self.frequencies

    @property
    def head_dim(self) -> int:
        return self.hidden // self.num_heads

    @property
    def input_dim(self) -> int:
        return self.features + (self.t_dim if self.add_time_to_input else 0)

    @property
    def t_l(self) -> int:
        return self.t_dim if self.t_local_cat else 0

    @property
    def t_g(self) -> int:
        return self.t_dim if self.t_global_cat else 0

    @property
    def needs_cond(self) -> bool:
        return bool(self.global_cond_dim or self.global_cat_cond or self.local_cat_cond)

    @staticmethod
    def from_hparams(hp: Mapping) -> "MdmaConfig":
        nc = dict(hp.get("net_config") or {})
        gcd, gcc, lcc = int(nc.get("global_cond_dim", 0)), bool(nc.get("global_cat_cond", False)), bool(nc.get("local_cat_cond", False))
        if gcd not in (0, 1) or (gcc and gcd != 1):
            # MDMA.forward appends global_cond_in.unsqueeze(-1), ONE value per jet (mdma.py:164-169), and sizes embbed_cls by global_cond_dim
            raise NotImplementedError("MDMA net_config.global_cond_dim must be 0 or 1 (1 with global_cat_cond): the reference's own shapes")
        if (gcd or gcc or lcc) and int(hp.get("global_cond_dim", 0)) != 1:
            raise ValueError("the conditional MDMA reads ONE condition value per jet (global_cond_in.unsqueeze(-1), mdma.py:158-169): "
                             "the model's global_cond_dim must be 1")
        # MDMA.__init__'s own defaults (mdma.py:101-102): the time concatenations default to True; its Linears are sized by
        # net_config.frequencies, the embedding by the CNF's frequencies: the reference itself only runs when the two agree
        tl, tg = bool(nc.get("t_local_cat", True)), bool(nc.get("t_global_cat", True))
        if (tl or tg) and int(nc.get("frequencies", 6)) != int(hp.get("frequencies", 6)):
            raise ValueError("MDMA with t_local_cat / t_global_cat: net_config.frequencies must equal the model's frequencies "
                             "(mdma.py:25-36 sizes the Linears by the former, flow_matching_module.py:208-221 the embedding by the latter)")
        return MdmaConfig(
            num_particles=int(hp["num_particles"]), features=int(hp.get("features", 3)), hidden=int(nc.get("hidden_dim", 256)),
            latent=int(nc.get("latent", 16)), num_layers=int(nc.get("layers", 16)), num_heads=int(nc.get("num_heads", 8)),
            avg_n=float(nc.get("avg_n", 30)), frequencies=int(hp.get("frequencies", 6)),
            add_time_to_input=bool(hp.get("add_time_to_input", True)), t_emb=str(hp.get("t_emb", "sincos")),
            t_local_cat=tl, t_global_cat=tg, global_cond_dim=gcd, global_cat_cond=gcc, local_cat_cond=lcc,
        )

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """(key, shape) in the reference's state_dict order (module registration order of mdma.py:113-140, :24-45)."""
        H, L = self.hidden, self.latent
        out: List[Tuple[str, Tuple[int, ...]]] = []

        def lin(k, o, i):
            out.extend([(k + ".weight", (o, i)), (k + ".bias", (o,))])

        Tl, Tg = self.t_l, self.t_g
        gcd, gcc, lcc = self.global_cond_dim, int(self.global_cat_cond), int(self.local_cat_cond)
        lin("net.embed", H, self.input_dim + Tl + lcc)
        lin("net.embbed_cls", L, H + 1 + gcd)
        for l in range(self.num_layers):
            p = f"net.encoder.{l}."
            lin(p + "fc0", H, H + Tl + lcc)
            lin(p + "fc0_cls", H, L + Tg + gcc)
            lin(p + "fc1", H, H + L + lcc)
            lin(p + "fc1_cls", L, H + 1 + gcd + Tg)
            lin(p + "fc2_cls", L, L + Tg + gcc)
            lin(p + "cond_cls", H, gcd)
            out.extend([(p + "attn.in_proj_weight", (3 * H, H)), (p + "attn.in_proj_bias", (3 * H,))])
            lin(p + "attn.out_proj", H, H)
            out.extend([(p + "ln.weight", (H,)), (p + "ln.bias", (H,))])
        lin("net.out", 1, H + lcc)
        lin("net.cond", L, 1 + gcd)
        return out

    def param_count(self) -> int:
        return sum(int(np.prod(s)) for _, s in self.param_shapes())


class MdmaLayout(TfLayout):
    """Descriptor + gather maps for one MdmaConfig."""

    def __init__(self, cfg: MdmaConfig, flags: int = 0):
        self.flags = flags
        H, L = cfg.hidden, cfg.latent
        if H % 128 or H > 512:
            raise NotImplementedError("the HIP MDMA kernels need hidden_dim to be a multiple of 128, at most 512")
        if cfg.num_heads * cfg.head_dim != H or cfg.head_dim not in (8, 16) or cfg.num_heads > 64:
            raise NotImplementedError("the HIP attention kernels are specialised for head_dim 8 and 16 (hidden_dim / num_heads)")
        if L % 4 or not 4 <= L <= 64:
            raise NotImplementedError("latent must be a multiple of 4 in 4..64")
        if not 1 <= cfg.num_layers <= PFM_MDMA_MAX_LAYERS:
            raise NotImplementedError(f"layers must be in 1..{PFM_MDMA_MAX_LAYERS}")
        if cfg.features > 16 or cfg.t_dim > 64:
            raise NotImplementedError("limits of this build: features <= 16, frequencies <= 32")
        self._init_params(cfg)
        self._build()

    # element (r, c) of a 2-D parameter
    def _el(self, key: str, rows, cols):
        return self.p_off[key] + rows * self._shape[key][1] + cols

    def _ak(self, key: str, r0: int, NO: int, c0: int, K: int, transposed: bool = False) -> int:
        """MFMA_AK of param[r0:r0+NO, c0:c0+K] (NO x K), or of its transpose (K x NO)."""
        rows, red = (K, NO) if transposed else (NO, K)
        assert rows % 16 == 0 and red % 128 == 0, (key, rows, red)
        ob = np.arange(rows // 16)[:, None, None, None, None]
        kc = np.arange(red // 128)[None, :, None, None, None]
        kt = np.arange(8)[None, None, :, None, None]
        lane = np.arange(64)[None, None, None, :, None]
        r = np.arange(4)[None, None, None, None, :]
        i = 16 * ob + (lane & 15) + 0 * (kc + kt + r)
        k = 128 * kc + 16 * kt + 4 * (lane >> 4) + r + 0 * ob
        src = self._el(key, r0 + k, c0 + i) if transposed else self._el(key, r0 + i, c0 + k)
        return self._put(src, primary=not transposed)

    def _km(self, key: str, r0: int, NO: int, c0: int, K: int) -> int:
        """KMAJOR [K][NO] of param[r0:r0+NO, c0:c0+K]."""
        return self._put(self._el(key, r0 + np.arange(NO)[None, :], c0 + np.arange(K)[:, None]))

    def _v(self, key: str, r0: int = 0, n: int = None) -> int:
        n = self._shape[key][0] - r0 if n is None else n
        return self._put(self.p_off[key] + r0 + np.arange(n))

    def _build(self):
        cfg = self.cfg
        H, L, T, F = cfg.hidden, cfg.latent, cfg.t_dim, cfg.features
        d = MdmaDesc()
        d.abi_version = PFM_MDMA_ABI_VERSION
        d.n_points, d.features, d.hidden, d.latent, d.layers = cfg.num_particles, F, H, L, cfg.num_layers
        d.heads, d.head_dim, d.t_dim, d.time_in_input = cfg.num_heads, cfg.head_dim, T, int(cfg.add_time_to_input)
        d.t_cat = int(cfg.t_local_cat) | (2 * int(cfg.t_global_cat))
        Tl, Tg = cfg.t_l, cfg.t_g
        gcd, gcc, lcc = cfg.global_cond_dim, int(cfg.global_cat_cond), int(cfg.local_cat_cond)
        d.c_cat = gcd | (2 * gcc) | (4 * lcc)
        d.flags = self.flags | (PFM_MDMA_F_TEMB_SINCOS if cfg.t_emb == "sincos" else 0) | (64 if cfg.t_emb == "gaussian" else 0)  # 64: PFM_*_F_TEMB_GIVEN
        d.neg_slope, d.ln_eps, d.avg_n = 0.01, 1e-5, cfg.avg_n
        d.freqs = self._put(self.freq_off + np.arange(T), primary=False)
        t0 = T if cfg.add_time_to_input else 0  # x = cat(temb, x): the time columns come first (flow_matching_module.py:201)
        d.emb_Wx = self._km("net.embed.weight", 0, H, t0, F)
        d.emb_Wt = self._km("net.embed.weight", 0, H, 0, T) if cfg.add_time_to_input else -1
        d.emb_Wt2 = self._km("net.embed.weight", 0, H, t0 + F, T) if Tl else -1  # x = cat(x, t_in) (mdma.py:155-156)
        d.emb_Wc = self._km("net.embed.weight", 0, H, t0 + F + Tl, 1) if lcc else -1
        d.emb_b = self._v("net.embed.bias")
        d.ecls_W = self._km("net.embbed_cls.weight", 0, L, 0, H + 1 + gcd)
        d.ecls_b = self._v("net.embbed_cls.bias")
        d.cond_W = self._km("net.cond.weight", 0, L, 0, 1 + gcd)
        d.cond_b = self._v("net.cond.bias")
        d.out_W = self._v("net.out.weight", 0, H)  # [1][H] row
        d.out_Wc = self._v("net.out.weight", H, 1) if lcc else -1
        d.out_b = self._v("net.out.bias")
        for l in range(cfg.num_layers):
            p = f"net.encoder.{l}."
            B = d.block[l]
            k = p + "fc0.weight"
            B.fc0 = TfLin(self._ak(k, 0, H, 0, H), self._km(k, 0, H, H + Tl, 1) if lcc else -1, self._km(k, 0, H, H, T) if Tl else -1,
                          self._v(p + "fc0.bias"), self._ak(k, 0, H, 0, H, True))
            k = p + "attn.in_proj_weight"
            B.kv = TfLin(self._ak(k, H, 2 * H, 0, H), -1, -1, self._v(p + "attn.in_proj_bias", H, 2 * H), self._ak(k, H, 2 * H, 0, H, True))
            B.q_W = self._km(k, 0, H, 0, H)
            B.q_b = self._v(p + "attn.in_proj_bias", 0, H)
            k = p + "fc1.weight"
            B.fc1 = TfLin(self._ak(k, 0, H, 0, H), self._km(k, 0, H, H + lcc, L), self._km(k, 0, H, H, 1) if lcc else -1, self._v(p + "fc1.bias"),
                          self._ak(k, 0, H, 0, H, True))
            B.fc0c_W, B.fc0c_b = self._km(p + "fc0_cls.weight", 0, H, 0, L + Tg + gcc), self._v(p + "fc0_cls.bias")
            B.ln_g, B.ln_b = self._v(p + "ln.weight"), self._v(p + "ln.bias")
            B.o_W, B.o_b = self._km(p + "attn.out_proj.weight", 0, H, 0, H), self._v(p + "attn.out_proj.bias")
            B.fc1c_W, B.fc1c_b = self._km(p + "fc1_cls.weight", 0, L, 0, H + 1 + gcd + Tg), self._v(p + "fc1_cls.bias")
            B.fc2c_W, B.fc2c_b = self._km(p + "fc2_cls.weight", 0, L, 0, L + Tg + gcc), self._v(p + "fc2_cls.bias")
            self._v(p + "cond_cls.bias")  # unused by the network (see the module docstring)
            if gcd:
                self._km(p + "cond_cls.weight", 0, H, 0, gcd)  # likewise (a slot for every parameter element)
        self._finish(d)
