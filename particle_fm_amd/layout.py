"""Host-side weight layout for libpfm_hip.so (see include/pfm_hip.h for the formats).

The reference stores every Linear of the EPiC network weight-normalised
(``weight_g (out,1)``, ``weight_v (out,in)``, ``bias (out)``; epic.py:66-81,
262-300) with the input columns in concatenation order ``[t ; payload ; cond]``
(SURVEY.md Appendix A).  The kernels want the *effective* matrices
``W = g * v / ||v||`` split into the block that multiplies per-particle
activations (MFMA operand order) and the blocks that multiply per-jet vectors
(K-major, folded into a per-jet bias).  This module computes, once per model
configuration and in numpy, the descriptor (offsets) and one int64 gather map
so that ``blob = source[index_map]`` where ``source`` is the concatenation of
all effective matrices, all biases, the time-embedding frequencies and a zero.
Building the blob is therefore a handful of differentiable torch ops
(``pack_blob``), and the gradient of the blob flows back to
``weight_g / weight_v / bias`` through the same map.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Mapping, Sequence, Tuple

import numpy as np
import torch

PFM_ABI_VERSION = 3
PFM_MAX_LAYERS = 24
PFM_HIDDEN = 128
PFM_F_SKIP_MASKED_TAIL = 1
PFM_F_BF16_MFMA = 2
PFM_F_F16X3_MFMA = 4
PFM_F_TEMB_SINCOS = 8
PFM_F_PACK_JETS = 16
PFM_F_GENERIC_SAMPLER = 32  # keep the generic sampler kernel where the lean evaluation (csrc/epic_fast.h) would run
PFM_F_QUAD_JETS = 64  # four jets per workgroup in fixed 32-row slots on a 128-row descriptor (set by hip_ops for padded short sets)


class LocalLin(ctypes.Structure):
    _fields_ = [("A", ctypes.c_int64), ("AT", ctypes.c_int64), ("We", ctypes.c_int64), ("b", ctypes.c_int64), ("A16", ctypes.c_int64)]


class DenseLin(ctypes.Structure):
    _fields_ = [("W", ctypes.c_int64), ("b", ctypes.c_int64)]


class EpicLayer(ctypes.Structure):
    _fields_ = [("gl1", DenseLin), ("gl2", DenseLin), ("lc1", LocalLin), ("lc2", LocalLin)]


class EpicDesc(ctypes.Structure):
    """ctypes mirror of ``pfm_epic_desc`` (include/pfm_hip.h)."""

    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("n_points", ctypes.c_int32),
        ("features", ctypes.c_int32),
        ("hidden", ctypes.c_int32),
        ("latent", ctypes.c_int32),
        ("layers", ctypes.c_int32),
        ("t_dim", ctypes.c_int32),
        ("cond_global", ctypes.c_int32),
        ("cond_local", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("sum_scale", ctypes.c_float),
        ("neg_slope", ctypes.c_float),
        ("blob_floats", ctypes.c_int64),
        ("freqs", ctypes.c_int64),
        ("l1x", DenseLin),
        ("l1_We", ctypes.c_int64),
        ("l1_b", ctypes.c_int64),
        ("l2", LocalLin),
        ("g1", DenseLin),
        ("g2", DenseLin),
        ("layer", EpicLayer * PFM_MAX_LAYERS),
        ("l3_W", ctypes.c_int64),
        ("l3_We", ctypes.c_int64),
        ("l3_b", ctypes.c_int64),
        ("l3_A", ctypes.c_int64),
        ("l3_A16", ctypes.c_int64),
        ("q_g1", ctypes.c_int64),
        ("q_g2", ctypes.c_int64),
        ("q_gl1", ctypes.c_int64 * PFM_MAX_LAYERS),
        ("q_gl2", ctypes.c_int64 * PFM_MAX_LAYERS),
        ("q_we1", ctypes.c_int64 * PFM_MAX_LAYERS),
        ("b_g1", ctypes.c_int64),
        ("b_g2", ctypes.c_int64),
        ("b_gl1", ctypes.c_int64 * PFM_MAX_LAYERS),
        ("b_gl2", ctypes.c_int64 * PFM_MAX_LAYERS),
        ("b_we1", ctypes.c_int64 * PFM_MAX_LAYERS),
    ]


def _round4(x: int) -> int:
    return (x + 3) & ~3


def saved_layout(N: int, F: int, layers: int, H: int = PFM_HIDDEN) -> Dict[str, int]:
    """Mirror of ``make_saved`` (csrc/epic_nfe.h): float offsets inside one jet's saved-activation
    record written by pfm_epic_fm_loss_forward."""
    s: Dict[str, int] = {}
    o = 0
    for k in ("y", "v", "u"):
        s[k] = o
        o += _round4(N * F)
    s["x1"] = o; o += N * H
    s["x2"] = o; o += N * H
    s["l1"] = o; s["xo"] = o + N * H; s["lstride"] = 2 * N * H; o += layers * 2 * N * H
    s["gstem1"] = o; o += H
    s["gstem"] = o; o += 16
    s["glayer"] = o; s["gstride"] = H + 16; o += layers * (H + 16)
    s["pool"] = o; s["pstride"] = H; o += (layers + 1) * H
    s["temb"] = o; o += 64
    s["total"] = o
    return s


@dataclass
class EpicConfig:
    """The subset of SetFlowMatchingLitModule kwargs that shapes the EPiC network
    (flow_matching_module.py:100-148)."""

    num_particles: int
    features: int = 3
    hidden_dim: int = 128
    latent: int = 16
    layers: int = 8
    frequencies: int = 6
    t_local_cat: bool = False
    t_global_cat: bool = False
    global_cond_dim: int = 0
    local_cond_dim: int = 0
    sum_scale: float = 1e-2
    neg_slope: float = 0.01  # F.leaky_relu default (epic.py:180)
    t_emb: str = "cosine"   # or "sincos" (flow_matching_module.py:208-211)
    # CNF.forward concatenates the time embedding to the particle features (flow_matching_module.py:199-200: x = cat(t, x)), so the
    # reference's fc_l1 has 2 * frequencies more input columns: [t_l ; temb ; x ; c_l].  Both time blocks multiply the SAME per-jet
    # vector, so the kernels see ONE time block whose weights are the sum of the two (source_vector folds them; the gradient of the
    # folded block flows back to both): no kernel knows about the switch.
    add_time_to_input: bool = False

    @property
    def t_local(self) -> int:
        return 2 * self.frequencies if self.t_local_cat else 0

    @property
    def t_l1(self) -> int:
        """time columns of fc_l1 as the kernels see it (after the folding of add_time_to_input)"""
        return 2 * self.frequencies if (self.t_local_cat or self.add_time_to_input) else 0

    @property
    def t_input(self) -> int:
        return 2 * self.frequencies if self.add_time_to_input else 0

    @property
    def t_global(self) -> int:
        return 2 * self.frequencies if self.t_global_cat else 0

    @property
    def t_dim(self) -> int:
        return 2 * self.frequencies if (self.t_local_cat or self.t_global_cat or self.add_time_to_input) else 0

    def linear_shapes(self) -> List[Tuple[str, int, int]]:
        """(name, in, out) of every Linear in the reference's registration order, as the kernels see them (fc_l1 with its two
        time blocks folded into one under add_time_to_input; the stored parameter has ``t_input`` more columns)."""
        H, L, F = self.hidden_dim, self.latent, self.features
        Tl, Tg, Cg, Cl = self.t_local, self.t_global, self.global_cond_dim, self.local_cond_dim
        out = [
            ("fc_l1", F + self.t_l1 + Cl, H),
            ("fc_l2", H + Tl + Cl, H),
            ("fc_g1", 2 * H + Tg + Cg, H),
            ("fc_g2", H + Tg + Cg, L),
        ]
        for k in range(self.layers):
            out += [
                (f"nn_list.{k}.fc_global1", 2 * H + L + Tg + Cg, H),
                (f"nn_list.{k}.fc_global2", H + Tg + Cg, L),
                (f"nn_list.{k}.fc_local1", H + L + Tl + Cl, H),
                (f"nn_list.{k}.fc_local2", H + Tl + Cl, H),
            ]
        out.append(("fc_l3", H + Tl + Cl, F))
        return out

    def param_count(self) -> int:
        return sum(o * i + 2 * o for _, i, o in self.linear_shapes()) + (self.t_local + self.t_input - self.t_l1) * self.hidden_dim


class EpicLayout:
    """Descriptor + gather map for one EpicConfig."""

    def __init__(self, cfg: EpicConfig, with_backward: bool = True, flags: int = 0):
        if cfg.hidden_dim != PFM_HIDDEN:
            raise NotImplementedError(
                f"the HIP kernels of this build are specialised for hidden_dim={PFM_HIDDEN}, got {cfg.hidden_dim}"
            )
        if cfg.layers > PFM_MAX_LAYERS:
            raise NotImplementedError(f"layers > {PFM_MAX_LAYERS}")
        if cfg.local_cond_dim not in (0, cfg.global_cond_dim):
            raise NotImplementedError("local_cond_dim must be 0 or equal to global_cond_dim (epic.py:122,354)")
        self.cfg = cfg
        self.with_backward = with_backward
        self.linears = cfg.linear_shapes()
        # ---- source vector: [W_eff of every linear (row-major) | biases | freqs | 0] ----
        self.w_off: Dict[str, int] = {}
        self.b_off: Dict[str, int] = {}
        o = 0
        for name, i, oo in self.linears:
            self.w_off[name] = o
            o += i * oo
        self.n_weight = o
        for name, i, oo in self.linears:
            self.b_off[name] = o
            o += oo
        self.freq_off = o
        o += cfg.t_dim
        self.zero_off = o
        o += 1
        self.n_source = o
        self._in = {name: i for name, i, _ in self.linears}
        self._out = {name: oo for name, _, oo in self.linears}
        self._build()
        self.desc.flags = flags | (PFM_F_TEMB_SINCOS if cfg.t_emb == "sincos" else 0)

    # -- helpers -------------------------------------------------------------------------------
    def _w(self, name: str, rows: np.ndarray, cols: np.ndarray) -> np.ndarray:
        """source indices of W[name][rows, cols] (broadcast); cols < 0 -> the zero slot."""
        idx = self.w_off[name] + rows * self._in[name] + np.maximum(cols, 0)
        return np.where(cols < 0, self.zero_off, idx)

    def _alloc(self, n: int) -> int:
        off = self._cursor
        self._cursor += (n + 3) & ~3
        return off

    def _put(self, off: int, idx: np.ndarray):
        flat = np.asarray(idx, dtype=np.int64).reshape(-1)
        self._segments.append((off, flat))

    def _kmajor(self, name: str, cols: Sequence[int]) -> int:
        """Per-jet GEMV block: element [k][o] = W[o][cols[k]].
        OUT = 128 -> KM16 (16-row panels, float ((k>>4)*32 + (o>>2))*64 + (k&15)*4 + (o&3)), rows zero-padded to 16;
        OUT <= 16 -> KP16 ([K16][16], outputs zero-padded to 16 columns)."""
        OUT = self._out[name]
        cols = list(cols)
        cols += [-1] * ((-len(cols)) % 16)
        cols = np.asarray(cols, dtype=np.int64)
        K16 = len(cols)
        if OUT == PFM_HIDDEN:
            k = np.arange(K16)[:, None]
            o = np.arange(OUT)[None, :]
            pos = ((k >> 4) * 32 + (o >> 2)) * 64 + (k & 15) * 4 + (o & 3)
            src = self._w(name, o + 0 * k, cols[:, None] + 0 * o)
            flat = np.empty(K16 * OUT, dtype=np.int64)
            flat[pos.reshape(-1)] = src.reshape(-1)
            off = self._alloc(K16 * OUT)
            self._put(off, flat)
            return off
        assert OUT <= 16
        flat = np.full((K16, 16), self.zero_off, dtype=np.int64)
        flat[:, :OUT] = self._w(name, np.arange(OUT)[None, :], cols[:, None])
        off = self._alloc(K16 * 16)
        self._put(off, flat)
        return off

    def _kq16(self, name: str, cols: Sequence[int]) -> int:
        """KQ16 (include/pfm_hip.h): OUT = 128, 16-row panels [o][16 k]; a forward-only second copy (no gradient slot)."""
        OUT = self._out[name]
        assert OUT == PFM_HIDDEN
        cols = list(cols)
        cols += [-1] * ((-len(cols)) % 16)
        cols = np.asarray(cols, dtype=np.int64)
        K16 = len(cols)
        k = np.arange(K16)[:, None]
        o = np.arange(OUT)[None, :]
        pos = (k >> 4) * 2048 + o * 16 + (k & 15)
        src = self._w(name, o + 0 * k, cols[:, None] + 0 * o)
        flat = np.empty(K16 * OUT, dtype=np.int64)
        flat[pos.reshape(-1)] = src.reshape(-1)
        off = self._alloc(K16 * OUT)
        self._put(off, flat)
        self._fwd_only.append((off, K16 * OUT))
        self._ch16_rows[off] = K16  # rows the block really has (a CH16 copy pads to a multiple of 32 with zeros)
        return off

    def _wq16(self, name: str, cols: Sequence[int]) -> int:
        """WQ16 (include/pfm_hip.h): OUT <= 16, K = 128: float (k>>4)*256 + o*16 + (k&15); forward-only second copy."""
        OUT = self._out[name]
        cols = np.asarray(list(cols), dtype=np.int64)
        assert OUT <= 16 and len(cols) == PFM_HIDDEN
        flat = np.full(PFM_HIDDEN * 16, self.zero_off, dtype=np.int64)
        k = np.arange(PFM_HIDDEN)[:, None]
        o = np.arange(OUT)[None, :]
        pos = (k >> 4) * 256 + o * 16 + (k & 15)
        flat[pos.reshape(-1)] = self._w(name, o + 0 * k, cols[:, None] + 0 * o).reshape(-1)
        off = self._alloc(PFM_HIDDEN * 16)
        self._put(off, flat)
        self._fwd_only.append((off, PFM_HIDDEN * 16))
        return off

    def _ch16(self, src_off: int, kind: str, K: int) -> int:
        """CH16 (include/pfm_hip.h): bf16 A-operand copy of a KQ16 (kind "kq", OUT = 128) or WQ16 (kind "wq", OUT <= 16) block with K
        real rows; like MFMA_A16 not part of the gather map: finish_blob / pfm_epic_pack_a16 fill it from the fp32 copy."""
        nw, nk = (8 if kind == "kq" else 1), (K + 31) // 32
        off = self._alloc(nw * nk * 64 * 4)
        self._ch16_blocks.append((int(src_off), int(off), kind, nk))
        return off

    def _plain_kmajor(self, name: str, cols: Sequence[int]) -> int:
        """plain K-major [K][OUT] (fc_l1's particle block, fc_l3's extras)"""
        OUT = self._out[name]
        cols = np.asarray(list(cols), dtype=np.int64)
        off = self._alloc(len(cols) * OUT)
        self._put(off, self._w(name, np.arange(OUT)[None, :], cols[:, None]))
        return off

    def _bias(self, name: str) -> int:
        OUT = self._out[name]
        n = max(OUT, 16) if OUT <= 16 else OUT  # small outputs padded to 16 (read as float4 groups)
        idx = np.full(n, self.zero_off, dtype=np.int64)
        idx[:OUT] = self.b_off[name] + np.arange(OUT)
        off = self._alloc(n)
        self._put(off, idx)
        return off

    def _mfma_a(self, name: str, c0: int, transposed: bool) -> int:
        w = np.arange(8)[:, None, None, None]
        kt = np.arange(8)[None, :, None, None]
        lane = np.arange(64)[None, None, :, None]
        r = np.arange(4)[None, None, None, :]
        i = 16 * w + (lane & 15)
        k = 16 * kt + 4 * (lane >> 4) + r
        off = self._alloc(PFM_HIDDEN * PFM_HIDDEN)
        if not transposed:
            self._put(off, self._w(name, i + 0 * k, c0 + k + 0 * i))
        else:
            self._put(off, self._w(name, k + 0 * i, c0 + i + 0 * k))
        return off

    def _local(self, name: str, tcols, xc0: int, ccols, gcols=()) -> LocalLin:
        ll = LocalLin()
        ll.A = self._mfma_a(name, xc0, False)
        ll.AT = self._mfma_a(name, xc0, True) if self.with_backward else -1
        # bf16 copy of the A block (MFMA_A16, include/pfm_hip.h): not part of the gather map -- filled from the fp32 block by
        # finish_blob() / pfm_epic_pack_a16 when the descriptor asks for bf16 operands
        ll.A16 = self._alloc(PFM_HIDDEN * PFM_HIDDEN // 2)
        self._local_blocks.append((name, xc0, int(ll.A), int(ll.AT)))
        self._a16_blocks.append((int(ll.A), int(ll.A16), 8))
        ll.We = self._kmajor(name, list(tcols) + list(ccols) + list(gcols))
        ll.b = self._bias(name)
        return ll

    def _build(self):
        cfg = self.cfg
        H, L, F = cfg.hidden_dim, cfg.latent, cfg.features
        self._H = H
        T, Tl, Tg, Cg, Cl = cfg.t_dim, cfg.t_local, cfg.t_global, cfg.global_cond_dim, cfg.local_cond_dim
        self._cursor = 0
        self._segments: List[Tuple[int, np.ndarray]] = []
        self._local_blocks: List[Tuple[str, int, int, int]] = []
        self._a16_blocks: List[Tuple[int, int, int]] = []  # (fp32 MFMA_A offset, MFMA_A16 offset, output slices w)
        self._ch16_rows: Dict[int, int] = {}
        self._ch16_blocks: List[Tuple[int, int, str, int]] = []  # (fp32 KQ16 / WQ16 offset, CH16 offset, "kq" | "wq", K tiles)
        self._fwd_only: List[Tuple[int, int]] = []  # (offset, floats) of second copies that carry no gradient (KQ16 / WQ16)
        d = EpicDesc()
        d.abi_version = PFM_ABI_VERSION
        d.n_points, d.features, d.hidden, d.latent, d.layers = cfg.num_particles, F, H, L, cfg.layers
        d.t_dim, d.cond_global, d.cond_local = T, Cg, Cl
        d.sum_scale, d.neg_slope = cfg.sum_scale, cfg.neg_slope

        def tcols(present: int):  # the T temb columns, or the zero slot if this group gets no time
            return list(range(T)) if present else [-1] * T

        d.freqs = self._alloc(T)
        self._put(d.freqs, self.freq_off + np.arange(T))
        # fc_l1: [t_l ; x(F) ; c_l]  (T1 = its time columns after the folding of add_time_to_input, EpicConfig.t_l1)
        T1 = cfg.t_l1
        d.l1x.W = self._plain_kmajor("fc_l1", range(T1, T1 + F))
        d.l1x.b = -1
        d.l1_We = self._kmajor("fc_l1", tcols(T1) + list(range(T1 + F, T1 + F + Cl)))
        d.l1_b = self._bias("fc_l1")
        # fc_l2: [t_l ; x(H) ; c_l]
        d.l2 = self._local("fc_l2", tcols(Tl), Tl, range(Tl + H, Tl + H + Cl))
        # fc_g1: reference order [t_g ; sum ; mean ; c_g] -> kernel order [temb ; cond ; mean ; sum]
        d.g1.W = self._kmajor(
            "fc_g1",
            tcols(Tg) + list(range(Tg + 2 * H, Tg + 2 * H + Cg)) + list(range(Tg + H, Tg + 2 * H)) + list(range(Tg, Tg + H)),
        )
        d.g1.b = self._bias("fc_g1")
        # fc_g2: [t_g ; g(H) ; c_g] -> [temb ; cond ; g1]
        d.g2.W = self._kmajor("fc_g2", tcols(Tg) + list(range(Tg + H, Tg + H + Cg)) + list(range(Tg, Tg + H)))
        d.g2.b = self._bias("fc_g2")
        # the lean sampler's copies (KQ16 / WQ16) without the time / conditioning rows: [mean ; sum], g1
        d.q_g1 = self._kq16("fc_g1", list(range(Tg + H, Tg + 2 * H)) + list(range(Tg, Tg + H)))
        d.q_g2 = self._wq16("fc_g2", range(Tg, Tg + H))
        d.b_g1 = self._ch16(d.q_g1, "kq", 2 * H)
        d.b_g2 = self._ch16(d.q_g2, "wq", H)
        for k in range(cfg.layers):
            p = f"nn_list.{k}."
            ly = d.layer[k]
            # fc_global1: [t_g ; mean ; sum ; g(L) ; c_g] -> [temb ; cond ; mean ; sum ; g]
            ly.gl1.W = self._kmajor(
                p + "fc_global1",
                tcols(Tg) + list(range(Tg + 2 * H + L, Tg + 2 * H + L + Cg)) + list(range(Tg, Tg + 2 * H + L)),
            )
            ly.gl1.b = self._bias(p + "fc_global1")
            ly.gl2.W = self._kmajor(
                p + "fc_global2", tcols(Tg) + list(range(Tg + H, Tg + H + Cg)) + list(range(Tg, Tg + H))
            )
            ly.gl2.b = self._bias(p + "fc_global2")
            # fc_local1: [t_l ; x(H) ; g(L) ; c_l] -> A | extras [temb ; cond_l ; g]
            ly.lc1 = self._local(
                p + "fc_local1", tcols(Tl), Tl, range(Tl + H + L, Tl + H + L + Cl), range(Tl + H, Tl + H + L)
            )
            # fc_local2: [t_l ; l1(H) ; c_l]
            ly.lc2 = self._local(p + "fc_local2", tcols(Tl), Tl, range(Tl + H, Tl + H + Cl))
            d.q_gl1[k] = self._kq16(p + "fc_global1", range(Tg, Tg + 2 * H + L))   # [mean ; sum ; g]
            d.q_gl2[k] = self._wq16(p + "fc_global2", range(Tg, Tg + H))
            d.q_we1[k] = self._kq16(p + "fc_local1", range(Tl + H, Tl + H + L))     # the g rows of the extras
            d.b_gl1[k] = self._ch16(d.q_gl1[k], "kq", 2 * H + 16)
            d.b_gl2[k] = self._ch16(d.q_gl2[k], "wq", H)
            d.b_we1[k] = self._ch16(d.q_we1[k], "kq", 16)
        # fc_l3: [t_l ; x(H) ; c_l]; particle block row-major [F][H]
        d.l3_W = self._alloc(F * H)
        self._put(d.l3_W, self._w("fc_l3", np.arange(F)[:, None], Tl + np.arange(H)[None, :]))
        d.l3_We = self._plain_kmajor("fc_l3", tcols(Tl) + list(range(Tl + H, Tl + H + Cl)))
        d.l3_b = self._bias("fc_l3")
        # the same particle block once more as a single 16-row MFMA_A panel (the head runs on the matrix cores too)
        kt = np.arange(8)[:, None, None]
        lane = np.arange(64)[None, :, None]
        r = np.arange(4)[None, None, :]
        f = (lane & 15) + 0 * (kt + r)
        k = 16 * kt + 4 * (lane >> 4) + r
        idx = self._w("fc_l3", np.minimum(f, F - 1), Tl + k + 0 * f)
        idx = np.where(f < F, idx, self.zero_off)
        d.l3_A = self._alloc(2048)
        self._put(d.l3_A, idx)
        d.l3_A16 = self._alloc(1024)
        self._a16_blocks.append((int(d.l3_A), int(d.l3_A16), 1))
        d.blob_floats = self._cursor
        index_map = np.full(self._cursor, self.zero_off, dtype=np.int64)
        for off, flat in self._segments:
            index_map[off : off + flat.size] = flat
        self.index_map = index_map
        self.desc = d
        del self._segments
        # Gradient blob (written by pfm_epic_fm_loss_backward): same offsets, but every 128x128 block is in
        # the accumulator-native GRAD_D order  float ((w*8 + it)*4 + r)*64 + lane  <->
        # dW[16w + 4(lane>>4) + r][c0 + 8(lane&15) + it]; the transposed copies receive nothing.
        gmap = index_map.copy()
        w = np.arange(8)[:, None, None, None]
        it = np.arange(8)[None, :, None, None]
        r = np.arange(4)[None, None, :, None]
        lane = np.arange(64)[None, None, None, :]
        rows = 16 * w + 4 * (lane >> 4) + r + 0 * it
        cols = 8 * (lane & 15) + it + 0 * (w + r)
        for name, c0, offA, offAT in self._local_blocks:
            gmap[offA : offA + H * H] = self._w(name, rows, c0 + cols).reshape(-1)
            if offAT >= 0:
                gmap[offAT : offAT + H * H] = self.zero_off
        gmap[d.l3_A : d.l3_A + 2048] = self.zero_off  # forward-only copy of fc_l3's particle block
        for off, n in self._fwd_only:  # KQ16 / WQ16 copies
            gmap[off : off + n] = self.zero_off
        self.grad_index_map = gmap
        # inverse maps, per source element (weights then biases): where it lands in the blob (one or two places:
        # MFMA_A and MFMA_AT), and the single place of the gradient blob that carries its gradient
        n_wb = self.freq_off
        d1 = np.full(n_wb, -1, dtype=np.int32)
        d2 = np.full(n_wb, -1, dtype=np.int32)
        pos = np.nonzero(index_map < n_wb)[0]
        order = np.argsort(index_map[pos], kind="stable")
        pos, srcs = pos[order], index_map[pos][order]
        first = np.ones(len(srcs), dtype=bool)
        first[1:] = srcs[1:] != srcs[:-1]
        d1[srcs[first]] = pos[first]
        second = ~first
        d2[srcs[second]] = pos[second]
        assert np.all(np.bincount(srcs, minlength=n_wb) <= 2) and np.all(d1 >= 0)
        gp = np.full(n_wb, -1, dtype=np.int32)
        gpos = np.nonzero(gmap < n_wb)[0]
        assert len(np.unique(gmap[gpos])) == len(gpos) == n_wb, "every weight/bias has exactly one gradient slot"
        gp[gmap[gpos]] = gpos
        self.src_dst1, self.src_dst2, self.src_gpos = d1, d2, gp

    # -- torch side ------------------------------------------------------------------------------
    def default_freqs(self) -> torch.Tensor:
        """exp(0..T-1) of time_emb.py:90 as correctly rounded fp32 (via float64).  The reference's own
        fp32 ``torch.arange(T).exp()`` differs between hosts by 1 ulp in some elements, which the
        1e13-sized cosine arguments amplify to O(1); the product therefore fixes the table."""
        if self.cfg.t_emb == "sincos":  # [f ; f] with f = 2^k pi, exactly the module buffer (flow_matching_module.py:172)
            f = 2 ** torch.arange(self.cfg.frequencies) * torch.pi
            return torch.cat([f, f]).to(torch.float32)
        return torch.arange(self.cfg.t_dim, dtype=torch.float64).exp().to(torch.float32)

    def source_vector(self, state: Mapping[str, torch.Tensor], prefix: str = "", freqs: torch.Tensor = None) -> torch.Tensor:
        """[W_eff ... | bias ... | freqs | 0] from reference-named tensors (differentiable).
        W_eff[o,:] = g[o] * v[o,:] / ||v[o,:]||  (old-style nn.utils.weight_norm, dim=0)."""
        ws, bs = [], []
        any_t = None
        for name, _, _ in self.linears:
            v = state[prefix + name + ".weight_v"]
            g = state[prefix + name + ".weight_g"]
            w = v * (g / v.norm(dim=1, keepdim=True))
            if name == "fc_l1" and self.cfg.add_time_to_input:
                # reference columns [t_l (Tl) ; temb (T) ; x ; c_l] -> kernel columns [t ; x ; c_l]: the two time blocks multiply the
                # same embedding, their weights add (autograd sends the folded block's gradient to both)
                T, Tl = self.cfg.t_input, self.cfg.t_local
                w = torch.cat([w[:, :T] + w[:, T:2 * T], w[:, 2 * T:]], dim=1) if Tl else w
            ws.append(w.reshape(-1))
            bs.append(state[prefix + name + ".bias"].reshape(-1))
            any_t = v
        if freqs is None:
            freqs = self.default_freqs()
        elif self.cfg.t_emb == "sincos" and freqs.numel() == self.cfg.frequencies:
            freqs = torch.cat([freqs, freqs])  # the module buffer holds f once
        freqs = freqs.to(device=any_t.device, dtype=any_t.dtype)
        zero = torch.zeros(1, device=any_t.device, dtype=any_t.dtype)
        return torch.cat(ws + bs + [freqs, zero])

    @property
    def desc_floats(self) -> int:
        """PFM_DESC_FLOATS of include/pfm_hip.h"""
        return (ctypes.sizeof(EpicDesc) + 15) // 16 * 4

    @property
    def blob_total(self) -> int:
        return int(self.desc.blob_floats) + self.desc_floats

    def padded(self, num_particles: int, extra_flags: int = 0) -> "EpicLayout":
        """The same network on a larger set size (same blob offsets: only desc.n_points -- and ``extra_flags`` -- differ); cached."""
        cache = self.__dict__.setdefault("_padded", {})
        key = (int(num_particles), int(extra_flags))
        lay = cache.get(key)
        if lay is None:
            import dataclasses
            lay = cache[key] = EpicLayout(dataclasses.replace(self.cfg, num_particles=int(num_particles)),
                                          with_backward=self.with_backward,
                                          flags=(int(self.desc.flags) & ~PFM_F_TEMB_SINCOS) | int(extra_flags))
            assert int(lay.desc.blob_floats) == int(self.desc.blob_floats)
        return lay

    def desc_tail(self) -> torch.Tensor:
        """The descriptor's bytes as fp32 words: the tail of every blob (read by the kernels)."""
        raw = bytes(self.desc)
        raw += b"\0" * (self.desc_floats * 4 - len(raw))
        return torch.from_numpy(np.frombuffer(raw, dtype=np.float32).copy())

    def pack_blob(self, state: Mapping[str, torch.Tensor], prefix: str = "", index_map: torch.Tensor = None,
                  freqs: torch.Tensor = None):
        src = self.source_vector(state, prefix, freqs)
        if index_map is None:
            index_map = torch.from_numpy(self.index_map).to(src.device)
        return self.finish_blob(torch.cat([src[index_map], self.desc_tail().to(src.device)]))

    def finish_blob(self, blob: torch.Tensor) -> torch.Tensor:
        """What a freshly gathered blob still lacks: the bf16 copies of the 128x128 particle blocks (MFMA_A16) that the lean bf16
        sampler streams instead of the fp32 ones.  Only blobs of bf16 descriptors (PFM_F_BF16_MFMA) carry them -- the region is zero
        otherwise.  In place; a device blob is converted by one launch (pfm_epic_pack_a16 on the current stream), a host blob here.
        Rounding: round-to-nearest-even, what v_cvt_pk_bf16_f32 does to the same value in the kernels that convert on the fly."""
        if not (int(self.desc.flags) & PFM_F_BF16_MFMA):
            return blob
        if blob.is_cuda:
            from . import _lib
            rc = _lib.load().pfm_epic_pack_a16(ctypes.byref(self.desc), ctypes.c_void_p(blob.data_ptr()),
                                               ctypes.c_void_p(torch.cuda.current_stream(blob.device).cuda_stream))
            _lib.check(rc, "pfm_epic_pack_a16")
            return blob
        with torch.no_grad():
            for offA, off16, nw in self._a16_blocks:
                a = blob[offA: offA + nw * 2048].detach().reshape(nw, 4, 2, 64, 4)   # [w][kt2][h][lane][r]
                a16 = a.permute(0, 1, 3, 2, 4).contiguous().to(torch.bfloat16)      # [w][kt2][lane][h][r]: 8 bf16 = one 16-byte unit
                blob[off16: off16 + nw * 1024] = a16.reshape(-1, 2).view(torch.float32).reshape(-1)
            for src, dst, kind, nk in self._ch16_blocks:
                nw = 8 if kind == "kq" else 1
                w = torch.arange(nw).view(nw, 1, 1, 1)
                kt = torch.arange(nk).view(1, nk, 1, 1)
                lane = torch.arange(64).view(1, 1, 64, 1)
                e = torch.arange(8).view(1, 1, 1, 8)
                o = 16 * w + (lane & 15)
                k = 32 * kt + 8 * (lane >> 4) + e
                if kind == "kq":   # KQ16: float (k >> 4) * 2048 + o * 16 + (k & 15); panels beyond the block's own read as zero
                    idx = (k >> 4) * 2048 + o * 16 + (k & 15)
                    size = self._ch16_rows[src]
                    val = torch.where(k < size, blob[src + torch.where(k < size, idx, torch.zeros_like(idx))].detach(), torch.zeros(()))
                else:              # WQ16: float (k >> 4) * 256 + o * 16 + (k & 15), K = 128
                    idx = (k >> 4) * 256 + o * 16 + (k & 15)
                    val = blob[src + idx].detach()
                b16 = (val + torch.zeros(nw, nk, 64, 8)).to(torch.bfloat16).contiguous()
                blob[dst: dst + nw * nk * 256] = b16.reshape(-1, 2).view(torch.float32).reshape(-1)
        return blob
