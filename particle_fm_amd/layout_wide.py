"""Host-side weight layout of the EPiC network at widths beyond the jet-resident kernel (include/pfm_epicw.h).

Same source vector as ``layout.EpicLayout`` -- ``[W_eff of every Linear (row-major) | biases | freqs | 0]`` with
``W_eff = g * v / ||v||`` -- and the same idea (one int64 gather map, ``blob = source[index_map]``), but every
matrix is an fp32-MFMA GEMM operand (MFMA_AK, pfm_tf.h) zero-padded to multiples of 64, and the columns that
multiply per-jet vectors are regrouped to address the per-jet row ``P = [temb | cond | 0.. (128) ; g | 0.. (128) ;
g1 (Hp)]`` and the pooled row ``Q = [mean (Hp) | sum*scale (Hp)]`` (column orders of the reference: SURVEY.md
Appendix A; epic.py:63-81, 259-300).
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence

import numpy as np

from .layout import EpicConfig, EpicLayout

PFM_EW_ABI_VERSION = 1
PFM_EW_MAX_LAYERS = 24
PFM_EW_F_F16X3 = 1
PFM_EW_F_TEMB_SINCOS = 2
PFM_EW_F_TEMB_GIVEN = 64


class EwLin(ctypes.Structure):
    _fields_ = [("W", ctypes.c_int64), ("b", ctypes.c_int64), ("WT", ctypes.c_int64)]


class EwLayer(ctypes.Structure):
    _fields_ = [("g1", EwLin), ("g2", EwLin), ("jb", EwLin), ("l1", EwLin), ("l2", EwLin)]


class EwDesc(ctypes.Structure):
    """ctypes mirror of ``pfm_ew_desc``."""

    _fields_ = [
        ("abi_version", ctypes.c_int32), ("n_points", ctypes.c_int32), ("features", ctypes.c_int32),
        ("hidden", ctypes.c_int32), ("hidden_pad", ctypes.c_int32), ("latent", ctypes.c_int32),
        ("layers", ctypes.c_int32), ("t_dim", ctypes.c_int32), ("cond_global", ctypes.c_int32),
        ("cond_local", ctypes.c_int32), ("flags", ctypes.c_int32), ("pad_", ctypes.c_int32),
        ("sum_scale", ctypes.c_float), ("neg_slope", ctypes.c_float),
        ("blob_floats", ctypes.c_int64), ("freqs", ctypes.c_int64), ("l1x", ctypes.c_int64), ("l3", ctypes.c_int64),
        ("sjb", EwLin), ("l2", EwLin), ("sg1", EwLin), ("sg2", EwLin),
        ("layer", EwLayer * PFM_EW_MAX_LAYERS),
    ]


class EpicWideLayout(EpicLayout):
    """Descriptor + gather maps for one EpicConfig of any hidden_dim <= 512."""

    def __init__(self, cfg: EpicConfig, with_backward: bool = True, flags: int = 0):
        self.flags = flags
        if cfg.hidden_dim > 512:
            raise NotImplementedError("hidden_dim > 512 is beyond this build's Linear kernel (K <= 512 per segment)")
        if cfg.layers > PFM_EW_MAX_LAYERS:
            raise NotImplementedError(f"layers > {PFM_EW_MAX_LAYERS}")
        if cfg.local_cond_dim not in (0, cfg.global_cond_dim):
            raise NotImplementedError("local_cond_dim must be 0 or equal to global_cond_dim (epic.py:122,354)")
        if cfg.t_dim + cfg.global_cond_dim > 128 or cfg.latent > 128 or cfg.features > 16:
            raise NotImplementedError("limits of this build: 2*frequencies + global_cond_dim <= 128, latent <= 128, features <= 16")
        self.cfg = cfg
        self.with_backward = with_backward
        self.linears = cfg.linear_shapes()
        self.w_off, self.b_off = {}, {}
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
        self.n_source = o + 1
        self._in = {name: i for name, i, _ in self.linears}
        self._out = {name: oo for name, _, oo in self.linears}
        self.Hp = (cfg.hidden_dim + 63) // 64 * 64
        self._build()

    # ---- helpers ------------------------------------------------------------------------------
    def _put(self, idx: np.ndarray, primary: bool = True) -> int:
        flat = np.asarray(idx, dtype=np.int64).reshape(-1)
        off = self._cursor
        self._cursor += (flat.size + 3) & ~3
        self._segs.append((off, flat, primary))
        return off

    def _src(self, name: str, rows: np.ndarray, cols: np.ndarray) -> np.ndarray:
        """source index of W[name][rows, cols]; rows >= out or cols < 0 -> the zero slot (broadcasts)."""
        rows, cols = np.broadcast_arrays(np.asarray(rows), np.asarray(cols))
        ok = (rows < self._out[name]) & (cols >= 0)
        idx = self.w_off[name] + np.minimum(rows, self._out[name] - 1) * self._in[name] + np.maximum(cols, 0)
        return np.where(ok, idx, self.zero_off)

    def _mfma_ak(self, rowsrc, NO: int, K: int) -> np.ndarray:
        """MFMA_AK order of the NO x K matrix whose element (o, k) has source index rowsrc(o, k)."""
        assert NO % 32 == 0 and K % 64 == 0, (NO, K)
        ob = np.arange(NO // 16)[:, None, None, None, None]
        st = np.arange(K // 64)[None, :, None, None, None]  # 64-wide k steps (two per 128-wide chunk of pfm_tf.h)
        kt = np.arange(4)[None, None, :, None, None]
        lane = np.arange(64)[None, None, None, :, None]
        r = np.arange(4)[None, None, None, None, :]
        o = 16 * ob + (lane & 15) + 0 * (st + kt + r)
        k = 64 * st + 16 * kt + 4 * (lane >> 4) + r + 0 * ob
        return rowsrc(o, k)

    def _lin(self, blocks: Sequence, NO: int, K: int, bias_rows: Sequence) -> EwLin:
        """blocks: list of (name, row0, nrows, colmap) -- output rows [row0, row0+nrows) of the padded matrix come from
        Linear ``name`` (its row o - row0) with input column k taken from source column colmap[k] (-1: zero).
        bias_rows: list of (name, row0) giving the padded bias vector."""
        colmaps = []
        for name, row0, nrows, colmap in blocks:
            cm = np.full(K, -1, dtype=np.int64)
            cm[: len(colmap)] = colmap
            colmaps.append((name, row0, nrows, cm))

        def rowsrc(o, k):
            out = np.full(np.broadcast(o, k).shape, self.zero_off, dtype=np.int64)
            for name, row0, nrows, cm in colmaps:
                sel = (o >= row0) & (o < row0 + nrows)
                src = self._src(name, o - row0, cm[k])
                out = np.where(sel, src, out)
            return out

        def rowsrc_T(i, j):  # the transposed matrix (K x NO): element (i, j) = W[j][i]
            return rowsrc(j, i)

        lin = EwLin()
        lin.W = self._put(self._mfma_ak(rowsrc, NO, K))
        lin.WT = self._put(self._mfma_ak(rowsrc_T, K, NO), primary=False) if self.with_backward else -1
        if bias_rows:
            b = np.full(NO, self.zero_off, dtype=np.int64)
            for name, row0 in bias_rows:
                n = self._out[name]
                b[row0:row0 + n] = self.b_off[name] + np.arange(n)
            lin.b = self._put(b)
        else:
            lin.b = -1
        return lin

    def _pcols(self, tcols: Sequence[int], ccols: Sequence[int], gcols: Sequence[int] = ()) -> np.ndarray:
        """column map over the first 256 columns of P: [temb (T) | cond | 0.. ; g | 0..]"""
        T = self.cfg.t_dim
        cm = np.full(256, -1, dtype=np.int64)
        cm[: len(tcols)] = tcols
        cm[T:T + len(ccols)] = ccols
        cm[128:128 + len(gcols)] = gcols
        return cm

    def _build(self):
        cfg = self.cfg
        H, Hp, L, F = cfg.hidden_dim, self.Hp, cfg.latent, cfg.features
        T, Tl, Tg, Cg, Cl = cfg.t_dim, cfg.t_local, cfg.t_global, cfg.global_cond_dim, cfg.local_cond_dim
        self._cursor = 0
        self._segs = []
        d = EwDesc()
        d.abi_version = PFM_EW_ABI_VERSION
        d.n_points, d.features, d.hidden, d.hidden_pad, d.latent, d.layers = cfg.num_particles, F, H, Hp, L, cfg.layers
        d.t_dim, d.cond_global, d.cond_local, d.flags = (
            T, Cg, Cl, self.flags | (PFM_EW_F_TEMB_SINCOS if cfg.t_emb == "sincos" else 0) | (PFM_EW_F_TEMB_GIVEN if cfg.t_emb == "gaussian" else 0))
        d.sum_scale, d.neg_slope = cfg.sum_scale, cfg.neg_slope
        d.freqs = self._put(self.freq_off + np.arange(T), primary=False)
        ar = np.arange

        def t(n):  # the time columns of a Linear that takes them first, or "absent"
            return list(range(n)) if n else []

        # fc_l1 [t_l ; x(F) ; c_l]: particle columns K-major [F][Hp].  T1 = its time columns as source_vector hands them over: under
        # add_time_to_input the reference's two time blocks [t_l ; temb] folded into one (layout.EpicConfig.t_l1), so T1 may be T with Tl = 0
        T1 = cfg.t_l1
        d.l1x = self._put(self._src("fc_l1", ar(Hp)[None, :], (T1 + ar(F))[:, None]))
        # fc_l3 [t_l ; x(H) ; c_l]: particle block row-major [F][Hp]
        kk = ar(Hp)[None, :]
        d.l3 = self._put(self._src("fc_l3", ar(16)[:, None], np.where(kk < H, Tl + kk, -1)))  # [16][Hp], rows >= F zero
        ext_l1 = self._pcols(t(T1), list(range(T1 + F, T1 + F + Cl)))
        ext_h = self._pcols(t(Tl), list(range(Tl + H, Tl + H + Cl)))  # fc_l2 / fc_local2 / fc_l3 share [t ; x(H) ; c]
        d.sjb = self._lin([("fc_l1", 0, Hp, ext_l1), ("fc_l2", Hp, Hp, ext_h), ("fc_l3", 2 * Hp, 128, ext_h)],
                          2 * Hp + 128, 256, [("fc_l1", 0), ("fc_l2", Hp), ("fc_l3", 2 * Hp)])
        xblock = list(range(Tl, Tl + H))
        d.l2 = self._lin([("fc_l2", 0, Hp, xblock)], Hp, Hp, [])
        # fc_g1 [t_g ; sum ; mean ; c_g] over [P(256) | mean (Hp) | sum (Hp)]
        cm = np.full(256 + 2 * Hp, -1, dtype=np.int64)
        cm[:256] = self._pcols(t(Tg), list(range(Tg + 2 * H, Tg + 2 * H + Cg)))
        cm[256:256 + H] = Tg + H + ar(H)
        cm[256 + Hp:256 + Hp + H] = Tg + ar(H)
        d.sg1 = self._lin([("fc_g1", 0, Hp, cm)], Hp, 256 + 2 * Hp, [("fc_g1", 0)])

        def g2map():  # fc_g2 / fc_global2 [t_g ; g1(H) ; c_g] over [P(256) | g1 (Hp)]
            m = np.full(256 + Hp, -1, dtype=np.int64)
            m[:256] = self._pcols(t(Tg), list(range(Tg + H, Tg + H + Cg)))
            m[256:256 + H] = Tg + ar(H)
            return m

        d.sg2 = self._lin([("fc_g2", 0, 128, g2map())], 128, 256 + Hp, [("fc_g2", 0)])
        for k in range(cfg.layers):
            p = f"nn_list.{k}."
            ly = d.layer[k]
            # fc_global1 [t_g ; mean ; sum ; g(L) ; c_g]
            cm = np.full(256 + 2 * Hp, -1, dtype=np.int64)
            cm[:256] = self._pcols(t(Tg), list(range(Tg + 2 * H + L, Tg + 2 * H + L + Cg)), list(range(Tg + 2 * H, Tg + 2 * H + L)))
            cm[256:256 + H] = Tg + ar(H)
            cm[256 + Hp:256 + Hp + H] = Tg + H + ar(H)
            ly.g1 = self._lin([(p + "fc_global1", 0, Hp, cm)], Hp, 256 + 2 * Hp, [(p + "fc_global1", 0)])
            ly.g2 = self._lin([(p + "fc_global2", 0, 128, g2map())], 128, 256 + Hp, [(p + "fc_global2", 0)])
            # fc_local1 [t_l ; x(H) ; g(L) ; c_l], fc_local2 [t_l ; l1(H) ; c_l]
            e1 = self._pcols(t(Tl), list(range(Tl + H + L, Tl + H + L + Cl)), list(range(Tl + H, Tl + H + L)))
            ly.jb = self._lin([(p + "fc_local1", 0, Hp, e1), (p + "fc_local2", Hp, Hp, ext_h)], 2 * Hp, 256,
                              [(p + "fc_local1", 0), (p + "fc_local2", Hp)])
            ly.l1 = self._lin([(p + "fc_local1", 0, Hp, xblock)], Hp, Hp, [])
            ly.l2 = self._lin([(p + "fc_local2", 0, Hp, xblock)], Hp, Hp, [])
        d.blob_floats = self._cursor
        self.desc = d
        idx = np.full(self._cursor, self.zero_off, dtype=np.int64)
        n_wb = self.freq_off
        gpos = np.full(n_wb, -1, dtype=np.int64)
        for off, flat, primary in self._segs:
            idx[off:off + flat.size] = flat
            if primary:
                sel = flat < n_wb
                assert (gpos[flat[sel]] == -1).all(), "weight / bias element with two primary slots"
                gpos[flat[sel]] = off + np.nonzero(sel)[0]
        assert (gpos >= 0).all(), "weight / bias element without a blob slot"
        self.index_map = idx
        self.grad_pos = gpos
        del self._segs

    # ---- packing ------------------------------------------------------------------------------
    @property
    def blob_total(self) -> int:
        return int(self.desc.blob_floats)

    def pack_blob(self, state, prefix: str = "", index_map=None, freqs=None):
        import torch

        src = self.source_vector(state, prefix, freqs)
        if index_map is None:
            index_map = torch.from_numpy(self.index_map).to(src.device)
        return src[index_map]
